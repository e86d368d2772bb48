"""cycle stamps of the point-major streaming solver (a -DRDIS_COOP_TIMING build, RDIS_PROBE_LIB), alone and with K
workgroups per component: where a trial point's time goes in the launch's first workgroup -- the cameras' trial
point, this workgroup's factors, the sums (K > 1: the exchange) -- and what a gradient costs"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rdis_amd import capi, problems as P
if os.environ.get("RDIS_PROBE_LIB"):
    capi.LIB_PATH = os.path.abspath(os.environ["RDIS_PROBE_LIB"])
ctx = capi.Context(0)
sizes = [(16, 2048, 4), (49, 7776, 4)]
for (ncam, npt, obs) in sizes:
    pp = P.make_synthetic_ba(125, ncam, npt, obs_per_pt=obs)
    csr = (pp.comp_free_ptr, pp.comp_free_vid, pp.comp_fac_ptr, pp.comp_fac_id)
    g = capi.Problem(ctx, pp)
    for opts in ({"ptm_group": 1}, {"ptm_group": 1, "ptm_threads": 256}, {"ptm_group": 2, "ptm_threads": 768}, {"ptm_group": 2, "ptm_threads": 512}, {"ptm_group": 4, "ptm_threads": 256}):
        plan = capi.Plan(g, *csr)
        for k, v in opts.items(): plan.set_option(k, v)
        plan.set_start(pp.x0[csr[1]])
        for _ in range(2):
            plan.solve(25, 3e-8); r = plan.fetch()
        ms = plan.last_kernel_ms()[0]
        tm = plan.debug_counters()
        print("%d x %d x %d, 125 comps, %s: kernel %.3f ms (group %d); first workgroup %.3f ms at 2.4 GHz, its component %d evaluations" % (
            ncam, npt, obs, opts, ms, plan.info("point_major_group"), tm[7] / 2.4e6, r.nfeval[np.argmax(np.diff(csr[2]))]))
        n = max(int(tm[3]), 1)
        print("   value+slope trials %d: cameras %.0f, factors %.0f, sums %.0f cycles each" % (tm[3], tm[0] / n, tm[1] / n, tm[2] / n))
        steps = max(int(tm[22] + tm[23] + tm[24] + tm[27]), 1)
        print("   control step %.0f cycles, hand-over %.0f (x%d requests)" % (tm[8] / steps, tm[9] / steps, steps))
        ng = max(int(tm[10]), 1)
        print("   gradient (x%d): until the chunks are done %.0f, after %.0f cycles" % (tm[10], tm[4] / ng, tm[5] / ng))
        print("   per request kind: " + "  ".join("%s %d x %.0f" % (nm, tm[22 + i], tm[12 + i] / max(int(tm[22 + i]), 1)) for i, nm in ((0, "value"), (1, "value+slope"), (2, "gradient"), (5, "line end"))))
        if tm[28]: print("   exchanges %d: publish %.0f, sweep %.0f, tail %.0f cycles each" % (tm[28], tm[29] / tm[28], tm[30] / tm[28], tm[31] / tm[28]))
        plan.close()
    g.close()
