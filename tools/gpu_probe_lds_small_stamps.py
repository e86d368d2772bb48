import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np
from rdis_amd import capi, problems as P
capi.LIB_PATH = os.path.abspath("build_ab/tm/rdis_amd/lib/librdis_hip.so")
ctx = capi.Context(0)
pp = P.make_synthetic_ba(1000, 3, 40)
g = capi.Problem(ctx, pp)
for opts in ({}, {"lds_threads": 64}, {"lds_threads": 256}):
    plan = capi.Plan(g)
    for k, v in opts.items(): plan.set_option(k, v)
    plan.set_start(pp.x0)
    for _ in range(2):
        plan.solve(25, 3e-8); r = plan.fetch()
    ms = plan.last_kernel_ms()[0]
    tm = plan.debug_counters()
    print("synthetic-S %s: kernel %.3f ms; first workgroup %.3f ms at 2.4 GHz; evals of its component %d iters %d" % (opts, ms, tm[7] / 2.4e6, r.nfeval[0], r.iters[0] + 1))
    n = max(int(tm[3]), 1)
    print("   value+slope trials %d: phase A %.0f, phase B %.0f, reduction %.0f cycles each" % (tm[3], tm[0] / n, tm[1] / n, tm[2] / n))
    steps = max(int(tm[22] + tm[23] + tm[24] + tm[27]), 1)
    print("   control step %.0f cycles, hand-over %.0f (x%d requests)" % (tm[8] / steps, tm[9] / steps, steps))
    ng = max(int(tm[10]), 1)
    print("   gradient (x%d): partials %.0f, sums %.0f" % (tm[10], tm[4] / ng, tm[5] / ng))
    print("   per request kind: " + "  ".join("%s %d x %.0f" % (nm, tm[22 + i], tm[12 + i] / max(int(tm[22 + i]), 1)) for i, nm in ((0, "value"), (1, "value+slope"), (2, "gradient"), (5, "line end"))))
    plan.close()
