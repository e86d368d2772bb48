"""profile target: Levenberg-Marquardt (pixel residuals) on full ladybug, 3 x 25 iterations
   rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/lm -o lm -- python tools/gpu_lm_profile.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rdis_amd import problems as P, capi
ctx = capi.Context(0)
pp = P.load_bal()
g = capi.Problem(ctx, pp)
for rep in range(3):
    g.set_x(pp.x0)
    t = time.time(); r = g.lm_optimize(maxiters=25, model=2); dt = time.time() - t
print("full ladybug, pixel residuals: f %.9g -> %.9g iters %d nsolve %d wall %.1f ms" % (r.fret - r.delta, r.fret, r.iters, r.nsolve, dt * 1e3))
