"""The public gradient entry point in its streaming regime: 256 x (49 cameras, 7776 points) = 8.0e6 factors, 6.1e6
variables (OptimizableFunction::computeGradient; algorithmic bytes 24 F + 16 N + 8).  Host clock of
rdis_hip_eval_grad (gradient copied to the host) and of rdis_hip_eval_grad_device + a wait for the stream."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rdis_amd import problems as P, capi
if os.environ.get("RDIS_PROBE_LIB"):
    capi.LIB_PATH = os.path.abspath(os.environ["RDIS_PROBE_LIB"])

ncomp = int(sys.argv[1]) if len(sys.argv) > 1 else 256
REPS = 10
ctx = capi.Context(0)
huge = P.make_synthetic_ba(ncomp, 49, 7776, obs_per_pt=4)
g = capi.Problem(ctx, huge)
g.set_x(huge.x0)
t0 = time.perf_counter(); f0, g0 = g.eval_grad(); t1 = time.perf_counter()
print("first call (tables built): %.1f ms" % ((t1 - t0) * 1e3))
alg = 24 * huge.nfac + 16 * huge.nvars + 8
gbuf = np.empty(huge.nvars)
for name, fn in (("rdis_hip_eval (value)", g.eval), ("rdis_hip_eval_grad (g copied to the host, a fresh array per call)", g.eval_grad),
                 ("rdis_hip_eval_grad (g copied to the host, the caller's array reused)", lambda: g.eval_grad(out=gbuf))):
    t0 = time.perf_counter()
    for _ in range(REPS): fn()
    dt = (time.perf_counter() - t0) / REPS
    print("%s: %.3f ms per call (host clock)" % (name, dt * 1e3))
ctx.synchronize()
t0 = time.perf_counter()
for _ in range(REPS): g.eval_grad_device()
t_issue = (time.perf_counter() - t0) / REPS
ctx.synchronize()
dt = (time.perf_counter() - t0) / REPS
print("rdis_hip_eval_grad_device: %.3f ms per call to issue, %.3f ms per call with the final wait (host clock); algorithmic %.0f MB -> %.0f GB/s"
      % (t_issue * 1e3, dt * 1e3, alg / 1e6, alg / dt / 1e9))
fd, gd = g.eval_grad_device()
if hasattr(ctx.lib, "rdis_hip_debug_grad_stamps"):   # (a -DRDIS_GRAD_STAMPS build: cycles of the first workgroup's first wave)
    import ctypes
    st = (ctypes.c_longlong * 8)()
    ctx.lib.rdis_hip_debug_grad_stamps(st, 1)
    g.eval_grad_device(); ctx.synchronize()
    ctx.lib.rdis_hip_debug_grad_stamps(st, 0)
    v = list(st)
    print("stamps of one call, first wave of the first workgroup: %d chunks; per chunk: factors %.0f, to the first barrier %.0f, segment sums %.0f, "
          "second barrier %.0f cycles; prologue + epilogue %d" % (v[4], v[0] / max(v[4], 1), v[1] / max(v[4], 1), v[2] / max(v[4], 1), v[3] / max(v[4], 1), v[5]))
gg = np.frombuffer(ctx.copy_to_host(gd, 8 * huge.nvars), dtype=np.float64)
print("same bits as the host variant:", bool(np.array_equal(gg, g0)), " f =", f0, " |g|_inf =", float(np.max(np.abs(g0))))
# a sub-list in another order: tables of its own
sub = np.random.default_rng(1).permutation(huge.nfac)[:1000000].astype(np.int64)
t0 = time.perf_counter(); g.eval_grad(sub); t1 = time.perf_counter(); g.eval_grad(sub); t2 = time.perf_counter()
print("1e6 scattered factors: first call %.1f ms, second %.1f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))
