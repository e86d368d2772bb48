"""exploratory: quad solver (4 lanes per tiny component) vs workgroup per component, by number of components"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rdis_amd import problems as P, capi
ctx = capi.Context(0)
for nblk in (1, 4, 16, 64):
    pp = P.make_synthetic_ba(nblk, 49, 7776, obs_per_pt=4)
    g = capi.Problem(ctx, pp)
    a = np.zeros(pp.nvars, np.uint8); a[np.arange(pp.nvars) % 23769 < 441] = 1
    comps = g.components(a)
    for q in (0, 4):
        plan = capi.Plan(g, *comps)
        plan.set_option("quad_max_vars", q)
        best = 1e9
        for rep in range(3):
            g.set_x(pp.x0); plan.set_start(None)
            plan.solve(25, 3e-8); r = plan.fetch()
            ms, nl = plan.last_kernel_ms(); best = min(best, ms)
        print("%7d point components, quad_max_vars %d: kernel %.3f ms, %d iterations (%.3g it/s), fret sum %.9g" % (
            len(comps[0]) - 1, q, best, int((r.iters + 1).sum()), (r.iters + 1).sum() / best * 1e3, r.fret.sum()))
        plan.close()
