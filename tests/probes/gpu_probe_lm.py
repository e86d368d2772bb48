"""exploratory: device LM vs the dense numpy oracle, step by step"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from rdis_amd import problems as P, capi
from oracle import oracle as O, lm_oracle as LM
ctx = capi.Context(0)
for nc, npt in ((5, 30), (49, 300)):
    pp = P.load_bal(ncams=nc, npts=npt)
    g = capi.Problem(ctx, pp)
    t = time.time(); r = g.lm_optimize(maxiters=25); dt = time.time() - t
    o = O.OracleProblem(pp, emulate_stale_cache=False)
    ro = LM.lm_optimize(o, maxiters=25)
    print("%d/%d: device f %.9g -> %.9g iters %d stop %d nsolve %d mu %.4g blocks %d/%d (%.1f ms)" % (nc, npt, r.fret - r.delta, r.fret, r.iters, r.stop, r.nsolve, r.mu, r.camera_blocks, r.point_blocks, dt * 1e3))
    print("        oracle f %.9g -> %.9g iters %d stop %d nsolve %d mu %.4g" % (ro.finit, ro.fret, ro.iters, ro.stop, ro.nsolve, ro.mu))
    for i in range(min(6, len(ro.history), len(r.history))):
        print("   ", ["%.10g" % v for v in r.history[i]], "|", ["%.10g" % float(v) for v in ro.history[i]])
pp = P.load_bal()
g = capi.Problem(ctx, pp)
for rep in range(2):
    g.set_x(pp.x0)
    t = time.time(); r = g.lm_optimize(maxiters=25); dt = time.time() - t
print("full ladybug: f %.9g -> %.9g iters %d stop %d nsolve %d mu %.4g blocks %d/%d wall %.1f ms" % (r.fret - r.delta, r.fret, r.iters, r.stop, r.nsolve, r.mu, r.camera_blocks, r.point_blocks, dt * 1e3))
print("   oracle objective at the device's point: %.9g" % (lambda o: (o.assign(None, r.x), o.eval())[1])(O.OracleProblem(pp)))
for rep in range(2):
    g.set_x(pp.x0)
    t = time.time(); r = g.lm_optimize(maxiters=25, model=2); dt = time.time() - t
print("full ladybug, pixel residuals: f %.9g -> %.9g iters %d stop %d nsolve %d wall %.1f ms; trial objectives %s" % (r.fret - r.delta, r.fret, r.iters, r.stop, r.nsolve, dt * 1e3, ["%.5g" % h[2] for h in r.history[:12]]))
