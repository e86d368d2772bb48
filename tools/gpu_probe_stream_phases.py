"""the streaming grid solver (solver_stream.hpp) on one large synthetic component: where the first workgroup's cycles go --
the factor loop of an evaluation (tm[0]), the grid exchanges and barriers (tm[4], tm[5] of them: publish tm[2], sweep tm[3]),
the whole kernel (tm[7]); shader clock at 100 MHz x 24 on this device family is NOT assumed: fractions are what counts.

    python tools/gpu_probe_stream_phases.py [cameras] [points]
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rdis_amd import problems as P, capi
if os.environ.get("RDIS_PROBE_LIB"):   # (the cycle counters need a -DRDIS_COOP_TIMING build)
    capi.LIB_PATH = os.path.abspath(os.environ["RDIS_PROBE_LIB"])
C_, Pn = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (64, 2000000)
ctx = capi.Context(0)
big = P.make_synthetic_ba(1, C_, Pn, obs_per_pt=4)
g = capi.Problem(ctx, big)
plan = capi.Plan(g)
plan.set_start(big.x0)
for rep in range(2):
    plan.solve(3, 3e-8); r = plan.fetch(want_x=False)
ms = plan.last_kernel_ms()[0]
tm = plan.debug_counters()
nf, ng = int(r.nfeval[0]), int(r.ngeval[0])
tot = max(int(tm[7]), 1)
print("%d factors, %d variables: kernel %.3f ms, %d evaluations (%d gradients): %.1f us per evaluation" % (big.nfac, big.nvars, ms, nf, ng, ms * 1e3 / nf))
print("first workgroup: factor loops %.1f %% of its cycles, exchanges + barriers %.1f %% (%d of them: publish %.1f %%, sweep %.1f %%), the rest (assignments, gradient sums, control) %.1f %%" % (
    100.0 * tm[0] / tot, 100.0 * tm[4] / tot, tm[5], 100.0 * tm[2] / tot, 100.0 * tm[3] / tot, 100.0 * (tot - tm[0] - tm[4]) / tot))
print("info:", {k: plan.info(k) for k in ("components_cooperative", "components_point_major", "components_plain")})
print("raw counters:", [int(v) for v in tm[:12]])
