"""exploratory: throughput regime -- many large components, one workgroup each (cgd_wg_kernel)"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rdis_amd import problems as P, capi
if os.environ.get("RDIS_PROBE_LIB"):
    capi.LIB_PATH = os.path.abspath(os.environ["RDIS_PROBE_LIB"])
ncomp = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ncams = int(sys.argv[2]) if len(sys.argv) > 2 else 49
npts = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
t = time.time()
pp = P.make_synthetic_ba(ncomp, ncams, npts, obs_per_pt=4)
print("built %d comps, %d factors, %d vars in %.1f s" % (pp.ncomp, pp.nfac, pp.nvars, time.time() - t))
ctx = capi.Context(0)
g = capi.Problem(ctx, pp)
for threads in (512, 768, 1024):
    plan = capi.Plan(g)
    plan.set_option("coop_max_components", 0)
    plan.set_option("block_threads", threads)
    plan.set_start(pp.x0[pp.comp_free_vid])
    for rep in range(2):
        plan.solve(25, 3e-8); r = plan.fetch()
    ms, nl = plan.last_kernel_ms()
    nfe, nge = int(r.nfeval.sum()), int(r.ngeval.sum())
    fpc = pp.nfac // pp.ncomp
    fe = float((r.nfeval * fpc).sum())   # factor evaluations
    byts = float(((24 * fpc + 8 * (pp.nvars // pp.ncomp) + 8) * r.nfeval).sum())
    print("threads %4d: kernel %.2f ms, %d iters (%.0f it/s), %.3g factor evals/s, algorithmic %.1f GB/s, fret sum %.6g, status %s" % (
        threads, ms, int((r.iters + 1).sum()), (r.iters + 1).sum() / ms * 1e3, fe / ms * 1e3, byts / ms / 1e6, r.fret.sum(),
        np.unique(r.status & 0xFF, return_counts=True)))
    plan.close()
