"""cycle stamps of the LDS-resident batch solver (a -DRDIS_COOP_TIMING build, RDIS_PROBE_LIB): where a
trial point's time goes -- phase A, phase B, reduction, control step, hand-over -- for the first component of
the strong-scaling block's rank-0 launch at 1 and 8 ranks"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
bench.STRONG.update(bench.STRONG_SIZES["small"])   # (round 2's block: 2048-factor components, the LDS-resident solver's case)
from rdis_amd import capi, problems as P
if os.environ.get("RDIS_PROBE_LIB"):
    capi.LIB_PATH = os.path.abspath(os.environ["RDIS_PROBE_LIB"])
ctx = capi.Context(0)
for world in (8, 1):
    pp, csr, mine, loads = bench.strong_scaling_shard(0, world)
    g = capi.Problem(ctx, pp)
    for opts in ({}, {"lds_threads": 768}, {"lds_threads": 256}, {"lds_rot": 0}):
        plan = capi.Plan(g, *csr)
        for k, v in opts.items():
            plan.set_option(k, v)
        plan.set_start(pp.x0[csr[1]])
        for _ in range(2):
            plan.solve(25, 3e-8); r = plan.fetch()
        ms = plan.last_kernel_ms()[0]
        tm = plan.debug_counters()
        c0 = int(np.argmax(np.diff(csr[2])))   # (all the same size: the first of the launch order)
        print("world %d %s: kernel %.3f ms; first workgroup: %d cycles in all (%.3f ms at 2.4 GHz)" % (world, opts, ms, tm[7], tm[7] / 2.4e6))
        n = max(int(tm[3]), 1)
        print("   value+slope trials %d: phase A %.0f, phase B %.0f, reduction %.0f cycles each" % (tm[3], tm[0] / n, tm[1] / n, tm[2] / n))
        steps = max(int(tm[22] + tm[23] + tm[24] + tm[27]), 1)
        print("   control step %.0f cycles, hand-over %.0f (x%d requests)" % (tm[8] / steps, tm[9] / steps, steps))
        ng = max(int(tm[10]), 1)
        print("   gradient (x%d): trial point + partials + scatter %.0f, per-variable sums %.0f, long runs %.0f cycles" % (tm[10], tm[4] / ng, tm[5] / ng, tm[6] / ng))
        print("   per request kind: " + "  ".join("%s %d x %.0f" % (nm, tm[22 + i], tm[12 + i] / max(int(tm[22 + i]), 1)) for i, nm in ((0, "value"), (1, "value+slope"), (2, "gradient"), (5, "line end"))))
        plan.close()
    g.close()
