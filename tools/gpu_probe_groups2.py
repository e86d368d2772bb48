"""exploratory: the point-major streaming solver on 125 / 250 / 500 / 1000 components of ladybug's size (one rank's share of
1000 at 8 / 4 / 2 / 1 ranks): workgroup size and workgroups per component"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rdis_amd import capi, problems as P
ctx = capi.Context(0)
ncam, npt, obs = 49, 7776, 4
for ncomp, variants in ((125, [{"ptm_group": 4, "ptm_threads": 256}, {"ptm_group": 3, "ptm_threads": 256}, {"ptm_group": 0}]),
                        (250, [{"ptm_group": 1}, {"ptm_group": 2, "ptm_threads": 256}, {"ptm_group": 1, "ptm_threads": 512}, {"ptm_group": 0}]),
                        (500, [{"ptm_group": 1}, {"ptm_group": 1, "ptm_threads": 768}, {"ptm_group": 1, "ptm_threads": 512}, {"ptm_group": 0}]),
                        (1000, [{"ptm_group": 1}, {"ptm_group": 1, "ptm_threads": 768}, {"ptm_group": 1, "ptm_threads": 512}, {"ptm_group": 0}])):
    pp = P.make_synthetic_ba(ncomp, ncam, npt, obs_per_pt=obs)
    csr = (pp.comp_free_ptr, pp.comp_free_vid, pp.comp_fac_ptr, pp.comp_fac_id)
    g = capi.Problem(ctx, pp)
    for opts in variants:
        plan = capi.Plan(g, *csr)
        for k, v in opts.items(): plan.set_option(k, v)
        plan.set_start(pp.x0[csr[1]])
        best = 1e9
        for rep in range(2):
            plan.solve(25, 3e-8); r = plan.fetch(); best = min(best, plan.last_kernel_ms()[0])
        print("%4d comps  %-40s kernel %9.3f ms  group %d  objective %.8g  evals mean %.0f max %d" % (
            ncomp, opts, best, plan.info("point_major_group"), r.fret.sum(), r.nfeval.mean(), r.nfeval.max()), flush=True)
        plan.close()
    g.close()
    del pp
