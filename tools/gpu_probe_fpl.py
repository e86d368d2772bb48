"""exploratory: the pipelined groups with two factors per lane (solver_pipe.hpp, FPL = 2) -- ladybug's 49 camera
components (245 workgroups instead of 273: one launch) and a single component of 40 000 factors -- against the plain
cooperative solver"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rdis_amd import problems as P, capi
ctx = capi.Context(0)
lb = P.load_bal()
cams, pts = P.ba_alternation_plans(lb)
big = P.make_synthetic_ba(1, 49, 10000, obs_per_pt=4)
whole = (np.array([0, big.nvars]), np.arange(big.nvars, dtype=np.int64), np.array([0, big.nfac]), np.arange(big.nfac, dtype=np.int64))
for name, pp, comps in (("ladybug's 49 camera components", lb, cams), ("1 x (49 cameras, 10000 points): %d factors" % big.nfac, big, whole)):
    g = capi.Problem(ctx, pp)
    for label, opts in (("pipelined where it fits", {}), ("plain cooperative", {"coop_pipeline": 0})):
        plan = capi.Plan(g, *comps)
        for k, v in opts.items(): plan.set_option(k, v)
        best = 1e9
        for rep in range(4):
            g.set_x(pp.x0); plan.set_start(None); plan.solve(25, 3e-8); r = plan.fetch()
            ms, nl = plan.last_kernel_ms(); best = min(best, ms)
        print("%-50s %-26s %.3f ms, %d launch(es), objective %.6f, evaluations %d, sync timeouts %d" % (
            name, label, best, nl, r.fret.sum(), r.nfeval.sum(), int(np.sum((r.status & 0xFF) == 7))))
        plan.close()
