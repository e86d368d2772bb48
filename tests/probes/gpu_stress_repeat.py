"""stress: the concurrent paths (cooperative groups side by side, batched launch overlapping
cooperative launches, persistent tiny-component groups) must give the same bits every time"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from rdis_amd import problems as P, capi
ctx = capi.Context(0)
pp = P.load_bal()
g = capi.Problem(ctx, pp)
cams, pts = P.ba_alternation_plans(pp)
a = np.zeros(pp.nvars, np.uint8); a[:9 * 46] = 1
mixed = g.components(a)
whole = (np.array([0, pp.nvars]), np.arange(pp.nvars, dtype=np.int64), np.array([0, pp.nfac]), np.arange(pp.nfac, dtype=np.int64))
cases = [("whole ladybug, pipelined group", whole, {}, 150), ("whole ladybug, pipelined, no guesses", whole, {"coop_speculate": 0}, 30),
         ("camera groups", cams, {}, 60), ("points, row16", pts, {}, 60), ("points, quad", pts, {"quad_min_components": 1}, 40),
         ("cooperative + batch overlap", mixed, {}, 60)]
bad = 0
for name, comps, opts, reps in cases:
    plan = capi.Plan(g, *comps)
    for k, v in opts.items(): plan.set_option(k, v)
    ref = None
    for rep in range(reps):
        g.set_x(pp.x0); plan.set_start(None)
        plan.solve(25, 3e-8); r = plan.fetch()
        key = (r.fret.tobytes(), r.x.tobytes(), r.iters.tobytes(), r.status.tobytes(), g.get_x().tobytes())
        if ref is None: ref = key
        elif key != ref:
            bad += 1; print("MISMATCH", name, "repetition", rep)
        if np.any((r.status & 0xFF) == 7): bad += 1; print("SYNC TIMEOUT", name, rep)
        if np.any((r.status & 0xFF) == 5): bad += 1; print("NAN", name, rep)
    print("%-30s %d repetitions, %d launches, identical" % (name, reps, plan.last_kernel_ms()[1]))
    plan.close()
# several pipelined groups side by side (each its own exchange state), repeated
syn = P.make_synthetic_ba(6, 3, 900, obs_per_pt=3)
gs = capi.Problem(ctx, syn)
plan = capi.Plan(gs, syn.comp_free_ptr, syn.comp_free_vid, syn.comp_fac_ptr, syn.comp_fac_id)
plan.set_option("coop_min_factors", 1000)
ref = None
for rep in range(100):
    gs.set_x(syn.x0); plan.set_start(None); plan.solve(25, 3e-8); r = plan.fetch()
    key = (r.fret.tobytes(), r.x.tobytes(), r.iters.tobytes(), r.status.tobytes(), r.nfeval.tobytes())
    if ref is None: ref = key
    elif key != ref: bad += 1; print("MISMATCH six pipelined groups, repetition", rep)
    if np.any((r.status & 0xFF) == 7): bad += 1; print("SYNC TIMEOUT six pipelined groups", rep)
print("%-30s %d repetitions, %d launches, identical" % ("six pipelined groups", 100, plan.last_kernel_ms()[1]))
plan.close()
print("FAILED" if bad else "ok")
