"""exploratory: the tiny-component regime (every point of ladybug a component, cameras fixed)"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from rdis_amd import problems as P, capi
from oracle import oracle as O
ctx = capi.Context(0)
pp = P.load_bal()
ncam = 49
npts = 7776
order = np.argsort(pp.pt_vid0, kind="stable")
fac_sorted = order.astype(np.int64)
counts = np.bincount((pp.pt_vid0 - 9 * ncam) // 3, minlength=npts)
fac_ptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
free_ptr = (np.arange(npts + 1) * 3).astype(np.int64)
free_vid = (9 * ncam + np.arange(3 * npts)).astype(np.int64)
g = capi.Problem(ctx, pp)
t = time.time(); plan = capi.Plan(g, free_ptr, free_vid, fac_ptr, fac_sorted); print("plan create %.2f ms" % ((time.time() - t) * 1e3))
for bt in (0,):
    for rep in range(3):
        plan.set_start(pp.x0[free_vid])
        t = time.time(); plan.solve(25, 3e-8); r = plan.fetch(); dt = time.time() - t
    ms, nl = plan.last_kernel_ms()
    print("7776 point components: wall %.3f ms kernel %.3f ms launches %d; iters %d, nfeval %d, status hist %s" % (
        dt * 1e3, ms, nl, int((r.iters + 1).sum()), int(r.nfeval.sum()), np.bincount(r.status & 0xff)))
t = time.time()
o = O.OracleProblem(pp)
tot = 0.0
for c in range(0, npts, 8):
    fv = free_vid[3 * c:3 * c + 3]; fc = fac_sorted[fac_ptr[c]:fac_ptr[c + 1]]
    tot += o.cgd(free_vid=fv, fac=fc, x=pp.x0[fv], maxiters=25).fret
dt = time.time() - t
print("oracle: every 8th component %.3f s => all ~%.2f s" % (dt, dt * 8))
# one-call-at-a-time cost through cgd_batch (what a per-call drop-in pays)
g2 = capi.Problem(ctx, pp)
t = time.time()
for c in range(200):
    fv = free_vid[3 * c:3 * c + 3]; fc = fac_sorted[fac_ptr[c]:fac_ptr[c + 1]]
    g2.cgd_batch(np.array([0, 3]), fv, np.array([0, len(fc)]), fc, pp.x0[fv], 25, 3e-8)
print("200 single calls: %.3f ms per call" % ((time.time() - t) / 200 * 1e3))
