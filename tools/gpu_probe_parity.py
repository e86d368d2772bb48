import sys, time; sys.path.insert(0, '.')
import numpy as np
from oracle import oracle as O
from rdis_amd import capi, problems as P
ctx = capi.Context(0)
for name, pp in (("5/30", P.load_bal(ncams=5, npts=30).single_component()), ("full", P.load_bal().single_component())):
    for stale in (1, 0):
        g = capi.Problem(ctx, pp); plan = capi.Plan(g)
        plan.set_option("factor_rounding", 1); plan.set_option("emulate_stale_cache", stale)
        plan.set_start(pp.x0); plan.solve(25, 3e-8); r = plan.fetch()
        ms = plan.last_kernel_ms()
        t = time.time(); w = O.OracleProblem.device_parity(pp, emulate_stale_cache=bool(stale)).cgd(x=pp.x0, maxiters=25); dt = time.time() - t
        print(name, "stale", stale, "device", repr(float(r.fret[0])), int(r.iters[0]), int(r.nfeval[0]), int(r.ngeval[0]), "kernel ms", ms,
              "| oracle", repr(w.fret), w.iters, w.nfeval, w.ngeval, f"{dt:.2f}s", "| equal x:", r.x.tobytes() == w.x.tobytes())
