"""the LDS-resident batch solver (solver_lds.hpp) against the plain one (solver_wg.hpp): the strong-scaling
block's rank-0 launch at 1 and 8 ranks (emulated on one GPU), synthetic-S, ladybug's camera and point
components -- kernel time per option set, and whether the bits agree with the plain solver's"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
bench.STRONG.update(bench.STRONG_SIZES["small"])   # (round 2's block: 2048-factor components, the LDS-resident solver's case)
from rdis_amd import capi, problems as P
ctx = capi.Context(0)


def run(g, csr, x0, opts, reps=3, iters=25):
    plan = capi.Plan(g, *csr)
    for k, v in opts.items():
        plan.set_option(k, v)
    plan.set_start(x0)
    best = 1e9
    for _ in range(reps):
        plan.solve(iters, 3e-8); r = plan.fetch(); best = min(best, plan.last_kernel_ms()[0])
    nl = plan.last_kernel_ms()[1]
    plan.close()
    return best, r, nl


VARIANTS = [("plain wg", {"lds_resident": 0}),
            ("lds auto", {}), ("lds vector form", {"lds_matrix": 0}),
            ("lds rot0", {"lds_rot": 0}), ("lds rot1", {"lds_rot": 1}),
            ("lds 512 rot0", {"lds_threads": 512, "lds_rot": 0}), ("lds 512 rot1", {"lds_threads": 512, "lds_rot": 1}),
            ("lds 1024 rot0", {"lds_threads": 1024, "lds_rot": 0}), ("lds 1024 rot1", {"lds_threads": 1024, "lds_rot": 1}),
            ("lds 256 rot1", {"lds_threads": 256, "lds_rot": 1})]
which = sys.argv[1:] or ["strong", "small", "cams", "points"]
if "strong" in which:
    for world in (1, 8):
        pp, csr, mine, loads = bench.strong_scaling_shard(0, world)
        g = capi.Problem(ctx, pp)
        ref = None
        for name, o in VARIANTS:
            ms, r, nl = run(g, csr, pp.x0[csr[1]], o)
            if ref is None: ref = r
            same = np.array_equal(ref.fret, r.fret) and np.array_equal(ref.x, r.x) and np.array_equal(ref.nfeval, r.nfeval)
            print("strong world %d (%4d comps) %-14s %8.3f ms  %d launch(es)  objective %.9g  evals mean %.0f max %d  bits==plain %s" % (
                world, len(mine), name, ms, nl, r.fret.sum(), r.nfeval.mean(), r.nfeval.max(), same), flush=True)
        g.close()
if "small" in which:
    pp = P.make_synthetic_ba(1000, 3, 40)
    g = capi.Problem(ctx, pp)
    csr = (pp.comp_free_ptr, pp.comp_free_vid, pp.comp_fac_ptr, pp.comp_fac_id)
    ref = None
    for name, o in [("plain wg", {"lds_resident": 0}), ("lds auto", {}), ("lds vector", {"lds_matrix": 0}), ("lds rot1", {"lds_rot": 1}), ("lds 64", {"lds_threads": 64}), ("lds 256", {"lds_threads": 256})]:
        ms, r, nl = run(g, csr, pp.x0[csr[1]], o)
        if ref is None: ref = r
        same = np.array_equal(ref.fret, r.fret) and np.array_equal(ref.x, r.x)
        print("synthetic-S 1000 x (3,40) %-12s %8.3f ms  objective %.9g  iters %d  bits==plain %s" % (name, ms, r.fret.sum(), int((r.iters + 1).sum()), same), flush=True)
    g.close()
if "cams" in which or "points" in which:
    pp = P.load_bal()
    cams, pts = P.ba_alternation_plans(pp)
    g = capi.Problem(ctx, pp)
    for label, csr, base in (("ladybug 49 camera comps", cams, {"coop_group_min_factors": 0, "coop_min_factors": 0}), ("ladybug 7776 point comps", pts, {"row_min_components": 1 << 30, "quad_min_components": 1 << 30})):
        if ("cams" in label and "cams" not in which) or ("point" in label and "points" not in which): continue
        ref = None
        for name, o in [("plain wg", {"lds_resident": 0}), ("lds auto", {}), ("lds vector", {"lds_matrix": 0}), ("lds 1024", {"lds_threads": 1024}), ("lds 512", {"lds_threads": 512})]:
            g.set_x(pp.x0)
            ms, r, nl = run(g, csr, pp.x0[csr[1]], {**base, **o})
            if ref is None: ref = r
            same = np.array_equal(ref.fret, r.fret) and np.array_equal(ref.x, r.x)
            print("%s %-10s %8.3f ms  %d launch(es) objective %.9g  bits==plain %s" % (label, name, ms, nl, r.fret.sum(), same), flush=True)
    g.close()
