"""Where the time of a cached tiny optimize() call goes: a 3-variable point component of ladybug 5/30 on a
resident plan, set_start / solve / fetch timed separately (ctypes binding included), kernel time from the events."""
import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
from rdis_amd import capi, problems as P
ctx = capi.Context(0)
pp = P.load_bal(ncams=5, npts=30)
g = capi.Problem(ctx, pp)
for which, gmin in (("point", 256), ("separator", 256), ("point", 1), ("separator", 1), ("separator", -1)):
    if which == "point":
        fv = np.arange(45 + 3, 48 + 3, dtype=np.int64)
        fi = np.where(pp.pt_vid0 == 48)[0].astype(np.int64)
    else:
        fv = np.arange(48, dtype=np.int64)
        fi = np.arange(pp.nfac, dtype=np.int64)
    plan = capi.Plan(g, np.array([0, len(fv)], np.int64), fv, np.array([0, len(fi)], np.int64), fi)
    x0 = np.ascontiguousarray(pp.x0[fv])
    if gmin > 0:
        plan.set_option('coop_group_min_factors', gmin)
    else:
        plan.set_option('coop_group_min_factors', 1); plan.set_option('coop_pipeline', 0)
    n = 1000
    t = np.zeros(4)
    kms = 0.0
    for rep in range(n + 50):
        a = time.perf_counter(); plan.set_start(x0)
        b = time.perf_counter(); plan.solve(25, 3e-8)
        c = time.perf_counter(); r = plan.fetch()
        d = time.perf_counter()
        if rep >= 50:
            t += (b - a, c - b, d - c, d - a)
            kms += plan.last_kernel_ms()[0]
    print("[group_min %d] " % gmin + "%s (%d variables, %d factors): set_start %.1f us, solve (launch) %.1f, fetch (wait + copy) %.1f, total %.1f; kernel %.1f us; %d f-evals"
          % (which, len(fv), len(fi), *(t / n * 1e6), kms / n * 1e3, int(r.nfeval[0])) + " fret %.9f" % r.fret[0], file=sys.stderr)
