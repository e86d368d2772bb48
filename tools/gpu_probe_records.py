"""exploratory: per-camera rotation records on / off (option "camera_records"):
   * launches of point components (cameras constant: records read only) -- workgroup per component and quad solver;
   * launches with free cameras (records rewritten per trial point) -- ladybug's camera components,
     synthetic-S, whole ladybug-sized components"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rdis_amd import problems as P, capi
ctx = capi.Context(0)

def run(name, g, pp, comps, opts, reps=3):
    for rec in (0, 1):
        plan = capi.Plan(g, *comps)
        for k, v in opts.items(): plan.set_option(k, v)
        plan.set_option("camera_records", rec)
        best = 1e9
        for rep in range(reps):
            g.set_x(pp.x0); plan.set_start(None)
            plan.solve(25, 3e-8); r = plan.fetch()
            ms, nl = plan.last_kernel_ms(); best = min(best, ms)
        print("%-28s %7d comps %-28s records %d: kernel %8.3f ms, %d iterations, fret sum %.15g" % (
            name, len(comps[0]) - 1, opts, rec, best, int((r.iters + 1).sum()), r.fret.sum()))
        plan.close()

pp = P.load_bal()
g = capi.Problem(ctx, pp)
cams, pts = P.ba_alternation_plans(pp)
run("ladybug points", g, pp, pts, {"quad_min_components": 1 << 40, "row_min_components": 1 << 40})
run("ladybug points", g, pp, pts, {"quad_min_components": 1 << 40, "row_min_components": 1})
run("ladybug points", g, pp, pts, {"quad_min_components": 1})
run("ladybug cameras", g, pp, cams, {})
g.close()
pp = P.make_synthetic_ba(1000, 3, 40)
g = capi.Problem(ctx, pp)
run("synthetic-S", g, pp, (pp.comp_free_ptr, pp.comp_free_vid, pp.comp_fac_ptr, pp.comp_fac_id), {})
g.close()
for nblk in (4, 64):
    pp = P.make_synthetic_ba(nblk, 49, 7776, obs_per_pt=4)
    g = capi.Problem(ctx, pp)
    a = np.zeros(pp.nvars, np.uint8); a[np.arange(pp.nvars) % 23769 < 441] = 1
    comps = g.components(a)
    run("synthetic x%d points" % nblk, g, pp, comps, {"quad_min_components": 1 << 40, "row_min_components": 1 << 40})
    run("synthetic x%d points" % nblk, g, pp, comps, {"quad_min_components": 1 << 40, "row_min_components": 1})
    run("synthetic x%d points" % nblk, g, pp, comps, {"quad_min_components": 1})
    if nblk == 64:
        g.set_x(pp.x0)
        run("synthetic x64 whole", g, pp, (pp.comp_free_ptr, pp.comp_free_vid, pp.comp_fac_ptr, pp.comp_fac_id), {"coop_min_factors": 0}, reps=2)
    g.close()
