"""exploratory: point components (3 variables, cameras fixed) by count -- a workgroup each,
sixteen lanes each, four lanes each.  Sets row_min_components / quad_min_components."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rdis_amd import problems as P, capi
ctx = capi.Context(0)
BIG = 1 << 40
def sweep(name, g, pp, comps, counts):
    fp, fv, cp, fi = comps
    for ncomp in counts:
        sub = (fp[:ncomp + 1], fv[:fp[ncomp]], cp[:ncomp + 1], fi[:cp[ncomp]])
        row = []
        for label, opts in (("workgroup", {"row_min_components": BIG, "quad_min_components": BIG}),
                            ("row16", {"row_min_components": 1, "quad_min_components": BIG}),
                            ("quad", {"quad_min_components": 1})):
            plan = capi.Plan(g, *sub)
            for k, v in opts.items(): plan.set_option(k, v)
            best = 1e9
            for rep in range(3):
                g.set_x(pp.x0); plan.set_start(None)
                plan.solve(25, 3e-8); r = plan.fetch()
                best = min(best, plan.last_kernel_ms()[0])
            row.append("%s %.3f ms" % (label, best))
            plan.close()
        print("%-12s %7d components: %s" % (name, ncomp, "   ".join(row)))
pp = P.load_bal()
g = capi.Problem(ctx, pp)
sweep("ladybug", g, pp, P.ba_alternation_plans(pp)[1], (512, 1024, 2048, 4096, 7776))
g.close()
pp = P.make_synthetic_ba(4, 49, 7776, obs_per_pt=4)
g = capi.Problem(ctx, pp)
a = np.zeros(pp.nvars, np.uint8); a[np.arange(pp.nvars) % 23769 < 441] = 1
sweep("synthetic", g, pp, g.components(a), (2048, 4096, 8192, 16384, 31104))
