"""exploratory: cooperative solver time vs number of CG iterations / evaluations (fixed vs marginal cost)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rdis_amd import problems as P, capi
if os.environ.get("RDIS_PROBE_LIB"):
    capi.LIB_PATH = os.path.abspath(os.environ["RDIS_PROBE_LIB"])
ctx = capi.Context(0)
pp = P.load_bal().single_component()
g = capi.Problem(ctx, pp)
plan = capi.Plan(g)
rows = []
for mi in (1, 2, 3, 5, 8, 12, 16, 20, 25):
    plan.set_start(pp.x0)
    best = 1e9
    for rep in range(4):
        plan.solve(mi, 3e-8); r = plan.fetch()
        ms, nl = plan.last_kernel_ms()
        best = min(best, ms)
    rows.append((mi, int(r.iters[0]) + 1, int(r.nfeval[0]), best))
    print("maxiters %2d: iters %2d nfeval %4d kernel %.3f ms" % rows[-1])
A = np.array([[1.0, it, nf] for _, it, nf, _ in rows])
y = np.array([t for *_, t in rows])
coef, *_ = np.linalg.lstsq(A, y, rcond=None)
print("fit: fixed %.1f us + %.2f us per CG iteration + %.3f us per evaluation; residual %.3f ms" % (
    coef[0] * 1e3, coef[1] * 1e3, coef[2] * 1e3, float(np.abs(A @ coef - y).max())))
