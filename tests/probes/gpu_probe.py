"""exploratory GPU run: kernel parity vs the oracle + solver timings"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from rdis_amd import problems as P, capi
from oracle import oracle as O
np.set_printoptions(precision=17, linewidth=200)
ctx = capi.Context(0)
print("devices", ctx.lib.rdis_hip_device_count())
for name, pp in [("5/30", P.load_bal(ncams=5, npts=30)), ("full", P.load_bal()), ("poly", P.load_poly()), ("sin", P.make_high_dim_sinusoid())]:
    if pp.kind == 1:
        pp.x0 = np.random.default_rng(1).uniform(-3, 3, pp.nvars)
    o = O.OracleProblem(pp)
    g = capi.Problem(ctx, pp)
    fo, fg = o.eval(), g.eval()
    print(name, "f oracle", repr(fo), "gpu", repr(fg), "rel", abs(fo - fg) / abs(fo))
    f2, gg = g.eval_grad()
    go = o.gradient()
    print(name, "grad maxerr rel-to-inf-norm", np.max(np.abs(gg - go)) / np.max(np.abs(go)), "f2", f2 == fg)
    e_o, e_g = o.eval_each(), g.eval_each()
    print(name, "each rel", np.max(np.abs(e_o - e_g) / np.maximum(np.abs(e_o), 1e-300)))
    if pp.kind == 0:
        a, b = o.grad_each_ba(), g.grad_each_ba()
        print(name, "grad_each rel-to-row-max", np.max(np.abs(a - b) / np.max(np.abs(a), axis=1, keepdims=True)))
    for mit in (1, 5, 25):
        plan = capi.Plan(g)
        plan.set_start(pp.x0)
        t = time.time(); plan.solve(mit, 3e-8); r = plan.fetch(); dt = time.time() - t
        ms, nl = plan.last_kernel_ms()
        oo = O.OracleProblem(pp); ro = oo.cgd(maxiters=mit)
        print(name, "cgd", mit, "gpu fret", repr(r.fret[0]), "oracle", repr(ro.fret), "rel", abs(r.fret[0] - ro.fret) / abs(ro.fret),
              "iters", r.iters[0], ro.iters, "status", r.status[0], ro.status, "nf/ng", r.nfeval[0], r.ngeval[0], ro.nfeval, ro.ngeval,
              "wall %.3f ms kernel %.3f ms" % (dt * 1e3, ms))
        g.set_x(pp.x0)
        plan.close()
# batch of synthetic components
pp = P.make_synthetic_ba(1000, 3, 40)
g = capi.Problem(ctx, pp)
plan = capi.Plan(g)
for rep in range(3):
    plan.set_start(pp.x0)
    t = time.time(); plan.solve(25, 3e-8); r = plan.fetch(); dt = time.time() - t
    ms, nl = plan.last_kernel_ms()
    print("synth 1000x(3x40): wall %.3f ms kernel %.3f ms, sum iters %d, sum fret %.6f, status hist %s" % (dt * 1e3, ms, int(r.iters.sum() + len(r.iters)), r.fret.sum(), np.bincount(r.status & 0xff)))
o = O.OracleProblem(pp)
t = time.time()
fr = []
for c in range(20):
    sl = slice(pp.comp_free_ptr[c], pp.comp_free_ptr[c + 1]); fs = slice(pp.comp_fac_ptr[c], pp.comp_fac_ptr[c + 1])
    ro = o.cgd(free_vid=pp.comp_free_vid[sl], fac=pp.comp_fac_id[fs], x=pp.x0[sl], maxiters=25)
    fr.append(ro.fret)
print("oracle 20 comps %.3f s" % (time.time() - t), "fret rel diff first 20", np.abs(np.array(fr) - r.fret[:20]) / np.array(fr))
