"""exploratory: the strong-scaling block's decomposition (1000 components of 2048 factors) under the batch solver's
workgroup sizes (option block_threads): 22.6-23.9 ms for 128..768 lanes, 35 ms for 1024 -- the launch is bound by
instruction issue (about 60 % of it), not by how the lanes are grouped"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rdis_amd import problems as P, capi
ctx = capi.Context(0)
pp = P.make_synthetic_ba(1000, 8, 512, obs_per_pt=4)
g = capi.Problem(ctx, pp)
comps = (pp.comp_free_ptr, pp.comp_free_vid, pp.comp_fac_ptr, pp.comp_fac_id)
for bt in (0, 128, 256, 512, 1024):
    plan = capi.Plan(g, *comps)
    plan.set_option("coop_min_factors", 0); plan.set_option("coop_group_min_factors", 0)
    if bt: plan.set_option("block_threads", bt)
    best = 1e9
    for rep in range(3):
        g.set_x(pp.x0); plan.set_start(None); plan.solve(25, 3e-8); r = plan.fetch()
        best = min(best, plan.last_kernel_ms()[0])
    print("block_threads %4d: %.3f ms, objective %.6f, evals %d" % (bt, best, r.fret.sum(), r.nfeval.sum()))
    plan.close()
