"""exploratory: K workgroups per component in the point-major streaming solver (cgd_ptmg_kernel).  One rank's share
of a 1000-component decomposition at 1 and 8 ranks (1000 / 125 components, emulated on one GPU) for several component
sizes: kernel time by ptm_group, and what that means for the 1 -> 8 speed-up."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rdis_amd import capi, problems as P
ctx = capi.Context(0)
sizes = [(8, 512, 4), (16, 2048, 4), (49, 7776, 4)]
if len(sys.argv) > 1: sizes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
for (ncam, npt, obs) in sizes:
    res = {}
    for ncomp in (125, 1000):
        t0 = time.time()
        pp = P.make_synthetic_ba(ncomp, ncam, npt, obs_per_pt=obs)
        csr = (pp.comp_free_ptr, pp.comp_free_vid, pp.comp_fac_ptr, pp.comp_fac_id)
        g = capi.Problem(ctx, pp)
        tb = time.time() - t0
        for opts in ({"ptm_group": 1}, {"ptm_group": 0}, {"ptm_group": 2, "ptm_threads": 768}, {"ptm_group": 2, "ptm_threads": 512}, {"ptm_group": 4, "ptm_threads": 256}):
            if ncomp == 1000 and opts.get("ptm_group", 0) > 1: continue
            plan = capi.Plan(g, *csr)
            for k, v in opts.items(): plan.set_option(k, v)
            plan.set_start(pp.x0[csr[1]])
            best = 1e9
            for rep in range(2):
                plan.solve(25, 3e-8); r = plan.fetch(); best = min(best, plan.last_kernel_ms()[0])
            info = {k: plan.info(k) for k in ("components_point_major", "point_major_group", "components_lds")}
            print("%2d x %4d x %d (%6d factors)  %4d comps  %-40s kernel %9.3f ms  group %d  (ptm %d, lds %d)  objective %.8g  evals mean %.0f max %d  [build %.1f s]" % (
                ncam, npt, obs, pp.nfac // ncomp, ncomp, opts, best, info["point_major_group"], info["components_point_major"], info["components_lds"],
                r.fret.sum(), r.nfeval.mean(), r.nfeval.max(), tb), flush=True)
            res[(ncomp, tuple(sorted(opts.items())))] = best
            plan.close()
        g.close()
        del pp
    one = res[(1000, (("ptm_group", 1),))]
    for k, v in res.items():
        if k[0] == 125: print("    1 -> 8 ranks with %s at 8 ranks: %.2f x" % (dict(k[1]), one / v))
