import sys, os
sys.path.insert(0, '/root/repo')
import numpy as np
from rdis_amd import problems as P, capi
ctx = capi.Context(0)
pp = P.load_bal()
cams, pts = P.ba_alternation_plans(pp)
g = capi.Problem(ctx, pp)
for name, opts in (("default (plain groups, one launch)", {}),
                   ("pipelined groups, packed launches", {"coop_group_min_factors": 0, "coop_min_factors": 300, "coop_max_components": 49}),
                   ("plain groups via the few-large rule", {"coop_group_min_factors": 0, "coop_min_factors": 300, "coop_max_components": 49, "coop_pipeline": 0}),
                   ("workgroup each", {"coop_group_min_factors": 0, "coop_min_factors": 0})):
    plan = capi.Plan(g, *cams)
    for k, v in opts.items(): plan.set_option(k, v)
    best = 1e9
    for rep in range(5):
        g.set_x(pp.x0); plan.set_start(None); plan.solve(25, 3e-8); r = plan.fetch()
        ms, nl = plan.last_kernel_ms(); best = min(best, ms)
    print("%-45s %.3f ms, %d launches, iterations %d, objective %.6f, evals %d" % (name, best, nl, int(r.iters.sum()) + 49, r.fret.sum(), r.nfeval.sum()))
    plan.close()
