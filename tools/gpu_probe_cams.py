"""exploratory: the 49 camera components of ladybug (points fixed): time vs iterations, block sizes"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rdis_amd import problems as P, capi
if os.environ.get("RDIS_PROBE_LIB"):
    capi.LIB_PATH = os.path.abspath(os.environ["RDIS_PROBE_LIB"])
ctx = capi.Context(0)
pp = P.load_bal()
cams, pts = P.ba_alternation_plans(pp)
g = capi.Problem(ctx, pp)
for threads in (0, 512, 768):
    plan = capi.Plan(g, *cams)
    if threads: plan.set_option("block_threads", threads)
    rows = []
    for mi in (1, 2, 4, 8, 16, 25):
        best = 1e9
        for rep in range(3):
            g.set_x(pp.x0); plan.set_start(None)
            plan.solve(mi, 3e-8); r = plan.fetch()
            ms, nl = plan.last_kernel_ms(); best = min(best, ms)
        rows.append((mi, int(r.iters.max()) + 1, int(r.nfeval.max()), best))
    A = np.array([[1.0, it, nf] for _, it, nf, _ in rows]); y = np.array([t for *_, t in rows])
    coef, *_ = np.linalg.lstsq(A, y, rcond=None)
    print("threads %4d: 25 iters %.3f ms (max nfeval %d); fit vs the slowest component: fixed %.0f us + %.1f us/iter + %.2f us/eval" % (
        threads, rows[-1][3], rows[-1][2], coef[0] * 1e3, coef[1] * 1e3, coef[2] * 1e3))
    plan.close()
