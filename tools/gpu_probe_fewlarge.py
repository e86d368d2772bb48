"""exploratory: a few components of ladybug's size -- cooperative groups packed into launches of
what is resident at once, against one workgroup per component.  Sets coop_max_components."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rdis_amd import problems as P, capi
ctx = capi.Context(0)
for ncomp in (2, 4, 8, 16, 32, 48):
    pp = P.make_synthetic_ba(ncomp, 49, 7776, obs_per_pt=4)
    g = capi.Problem(ctx, pp)
    comps = (pp.comp_free_ptr, pp.comp_free_vid, pp.comp_fac_ptr, pp.comp_fac_id)
    row = []
    for label, opts in (("cooperative groups", {"coop_max_components": 64}), ("workgroup each", {"coop_min_factors": 0})):
        plan = capi.Plan(g, *comps)
        for k, v in opts.items(): plan.set_option(k, v)
        best = 1e9
        for rep in range(2):
            g.set_x(pp.x0); plan.set_start(None)
            plan.solve(25, 3e-8); r = plan.fetch(want_x=False)
            ms, nl = plan.last_kernel_ms(); best = min(best, ms)
        row.append("%s %.2f ms (%d launches)" % (label, best, nl))
        plan.close()
    print("%3d components of %d factors: %s" % (ncomp, pp.nfac // ncomp, "   ".join(row)))
    g.close()
