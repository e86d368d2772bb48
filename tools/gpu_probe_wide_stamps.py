"""cycle stamps of a wide point-major group (a -DRDIS_COOP_TIMING build: tools/build_timing.sh, RDIS_PROBE_LIB): where a trial
point's time goes in the group's first workgroup -- cameras' records, its factors, the exchange -- and what a gradient costs

    RDIS_PROBE_LIB=build_ab/librdis_hip_timing.so python tools/gpu_probe_wide_stamps.py [cameras] [points] [obs] [iters] [K]
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rdis_amd import problems as P, capi
if os.environ.get("RDIS_PROBE_LIB"):
    capi.LIB_PATH = os.path.abspath(os.environ["RDIS_PROBE_LIB"])
C_, Pn, K_, IT, G = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((1, 64), (2, 2000000), (3, 4), (4, 3), (5, 0)))
ctx = capi.Context(0)
big = P.make_synthetic_ba(1, C_, Pn, obs_per_pt=K_)
g = capi.Problem(ctx, big)
plan = capi.Plan(g)
if G: plan.set_option("ptm_group", G)
for _ in range(2):
    plan.set_start(big.x0); plan.solve(IT, 3e-8); r = plan.fetch(want_x=False)
ms = plan.last_kernel_ms()[0]
tm = plan.debug_counters()
print("%d x %d x %d: kernel %.3f ms, K = %d (wide %d); first workgroup %.3f ms at 2.4 GHz; %d evaluations" % (
    C_, Pn, K_, ms, plan.info("point_major_group"), plan.info("point_major_wide"), tm[7] / 2.4e6, r.nfeval[0]))
n = max(int(tm[3]), 1)
print("   value+slope trials %d: cameras %.0f, factors %.0f (slowest wave %.0f, fastest %.0f, mean %.0f), sums %.0f cycles each" % (
    tm[3], tm[0] / n, tm[1] / n, tm[20] / n, tm[6] / n, tm[21] / n, tm[2] / n))
steps = max(int(tm[22] + tm[23] + tm[24] + tm[27]), 1)
print("   control step %.0f cycles, hand-over %.0f (x%d requests)" % (tm[8] / steps, tm[9] / steps, steps))
ng = max(int(tm[10]), 1)
print("   gradient (x%d): until the chunks are done %.0f, after %.0f cycles" % (tm[10], tm[4] / ng, tm[5] / ng))
nr = max(int(tm[10]), 1)
print("   gradient rounds, cycles per gradient of the first wave: a round's factor %.0f, wait for the others %.0f, staging + table %.0f, wait %.0f, sums %.0f" % (
    tm[15] / nr, tm[16] / nr, tm[18] / nr, tm[19] / nr, tm[11] / nr))
print("   per request kind: " + "  ".join("%s %d x %.0f" % (nm, tm[22 + i], tm[12 + i] / max(int(tm[22 + i]), 1)) for i, nm in ((0, "value"), (1, "value+slope"), (2, "gradient"), (5, "line end"))))
if tm[28]: print("   exchanges %d: publish %.0f, sweep %.0f, tail %.0f cycles each" % (tm[28], tm[29] / tm[28], tm[30] / tm[28], tm[31] / tm[28]))
