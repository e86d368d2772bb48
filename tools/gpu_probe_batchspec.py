"""exploratory (needs tools/experiments/r02_batch_speculation.patch applied: option batch_speculate): the batch solver with and without guessed trial steps on decompositions of
small and middling components: kernel time, and whether every result and call count is the same bits"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rdis_amd import problems as P, capi
ctx = capi.Context(0)
cases = [("1000 x (3 cams, 40 pts)", P.make_synthetic_ba(1000, 3, 40, obs_per_pt=3), {}),
         ("256 x (6 cams, 128 pts, 4 obs)", P.make_synthetic_ba(256, 6, 128, obs_per_pt=4), {"coop_min_factors": 0, "coop_group_min_factors": 0}),
         ("1000 x (8 cams, 512 pts, 4 obs) at 512 lanes", P.make_synthetic_ba(1000, 8, 512, obs_per_pt=4), {"coop_min_factors": 0, "coop_group_min_factors": 0, "block_threads": 512})]
lb = P.load_bal()
cams, pts = P.ba_alternation_plans(lb)
for name, pp, opts in cases + [("ladybug's 49 camera components, a workgroup each", lb, {"coop_min_factors": 0, "coop_group_min_factors": 0, "_comps": cams}),
                                ("ladybug's 7776 point components, a workgroup each", lb, {"row_min_components": 1 << 40, "quad_min_components": 1 << 40, "_comps": pts})]:
    comps = opts.pop("_comps", None) or (pp.comp_free_ptr, pp.comp_free_vid, pp.comp_fac_ptr, pp.comp_fac_id)
    g = capi.Problem(ctx, pp)
    out = {}
    for spec in (0, 1):
        plan = capi.Plan(g, *comps)
        for k, v in opts.items(): plan.set_option(k, v)
        plan.set_option("batch_speculate", spec)
        best = 1e9
        for rep in range(4):
            g.set_x(pp.x0); plan.set_start(None); plan.solve(25, 3e-8); r = plan.fetch()
            best = min(best, plan.last_kernel_ms()[0])
        out[spec] = (best, r, g.get_x())
        plan.close()
    (t0, r0, x0), (t1, r1, x1) = out[0], out[1]
    same = np.array_equal(r0.fret, r1.fret) and np.array_equal(r0.x, r1.x) and np.array_equal(x0, x1) and np.array_equal(r0.nfeval, r1.nfeval) and np.array_equal(r0.ngeval, r1.ngeval) and np.array_equal(r0.status, r1.status)
    print("%-52s plain %.3f ms, with guesses %.3f ms (%.0f %%), same bits: %s, iterations %d" % (name, t0, t1, 100.0 * t1 / t0, same, int(r0.iters.sum() + len(r0.iters))))
