import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rdis_amd import problems as P, capi
ctx = capi.Context(0)
def run(name, g, pp, comps, opts):
    for rec in (0, 2):
        plan = capi.Plan(g, *comps)
        for k, v in opts.items(): plan.set_option(k, v)
        plan.set_option("camera_records", rec)
        best = 1e9
        for rep in range(4):
            g.set_x(pp.x0); plan.set_start(None)
            plan.solve(25, 3e-8); r = plan.fetch()
            best = min(best, plan.last_kernel_ms()[0])
        print("%-22s %-22s records %d: %.3f ms  fret sum %.15g" % (name, opts, rec, best, r.fret.sum()))
        plan.close()
pp = P.load_bal(); g = capi.Problem(ctx, pp)
cams, pts = P.ba_alternation_plans(pp)
for th in (0, 256, 512, 768, 1024):
    run("ladybug cameras", g, pp, cams, {"block_threads": th})
g.close()
pp = P.make_synthetic_ba(1000, 3, 40); g = capi.Problem(ctx, pp)
run("synthetic-S", g, pp, (pp.comp_free_ptr, pp.comp_free_vid, pp.comp_fac_ptr, pp.comp_fac_id), {})
