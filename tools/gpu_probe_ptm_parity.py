"""the default point-major streaming path (one workgroup of 768 lanes a component) against the oracle's restatement of its sums"""
import sys, time; sys.path.insert(0, '.')
import numpy as np
from oracle import oracle as O
from rdis_amd import capi, problems as P
ctx = capi.Context(0)
L3 = P.make_synthetic_ba(3, 49, 7776, obs_per_pt=4)
cases = [("49x7776x4", L3, {"ptm_group": 1}, 768, 1),
         ("16x2000x4 forced", P.make_synthetic_ba(3, 16, 2000, obs_per_pt=4), {"ptm_group": 1, "ptm_stream": 2}, 768, 1),
         ("8x700x3 forced", P.make_synthetic_ba(2, 8, 700, obs_per_pt=3), {"ptm_group": 1, "ptm_stream": 2}, 768, 1),
         ("49x7776x4 pairs", L3, {"ptm_group": 2}, 512, 2),
         ("49x7776x4 fours", L3, {"ptm_group": 4}, 512, 4)]
for name, pp, opts, threads, K in cases:
    g = capi.Problem(ctx, pp); plan = capi.Plan(g)
    plan.set_option("coop_min_factors", 1 << 40)   # (a few components: keep them off the cooperative solver)
    for k, v in opts.items(): plan.set_option(k, v)
    plan.set_start(pp.x0); plan.solve(25, 3e-8); r = plan.fetch()
    print(name, {k: plan.info(k) for k in ("components_point_major", "point_major_group")})
    for c in range(pp.ncomp):
        fv, fc = pp.component(c)
        t = time.time()
        w = O.OracleProblem.device_ptm_default(pp, fac=fc, threads=threads, group=K).cgd(free_vid=fv, fac=fc, x=pp.x0[fv], maxiters=25)
        print(" comp", c, "device", repr(float(r.fret[c])), int(r.iters[c]), int(r.nfeval[c]), int(r.ngeval[c]), "| oracle", repr(w.fret), w.iters, w.nfeval, w.ngeval,
              f"{time.time() - t:.1f}s", "==" if float(r.fret[c]) == w.fret else "DIFFERENT")
