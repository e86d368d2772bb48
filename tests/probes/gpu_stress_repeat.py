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
# the pipelined group under timing changes: every poll delay, guesses on and off, tracing on and off, several
# group sizes -- the bits must not depend on any of it (RDIS_STRESS_SOAK=k: k x 1260 solves)
if os.environ.get("RDIS_STRESS_SOAK"):
    for label, prob in (("ladybug 49/7776", pp), ("ladybug 49/500", P.load_bal(ncams=49, npts=500)), ("ladybug 5/30", P.load_bal(ncams=5, npts=30))):
        gp = capi.Problem(ctx, prob)
        wh = (np.array([0, prob.nvars]), np.arange(prob.nvars, dtype=np.int64), np.array([0, prob.nfac]), np.arange(prob.nfac, dtype=np.int64))
        ref = None
        n = 0
        for delay in (0, 4, 16, 48):
            for spec in (1, 0):
                for tr in (0, 2048):
                    plan = capi.Plan(gp, *wh)
                    plan.set_option("coop_min_factors", 64); plan.set_option("coop_poll_delay", delay); plan.set_option("coop_speculate", spec)
                    if tr: plan.set_option("trace_records", tr)
                    for rep in range((60 if spec and not tr else 15) * max(1, int(os.environ.get("RDIS_STRESS_SOAK", "1")))):
                        gp.set_x(prob.x0); plan.set_start(None); plan.solve(25, 3e-8); r = plan.fetch(); n += 1
                        key = (r.fret.tobytes(), r.x.tobytes(), r.iters.tobytes(), r.status.tobytes(), r.nfeval.tobytes(), r.ngeval.tobytes())
                        if ref is None: ref = key
                        elif key != ref: bad += 1; print("MISMATCH", label, "delay", delay, "guesses", spec, "trace", tr, "repetition", rep)
                        if np.any((r.status & 0xFF) == 7): bad += 1; print("SYNC TIMEOUT", label, delay, spec, tr, rep)
                    plan.close()
        print("%-30s %d solves under 16 timing variants, identical" % (label + " (soak)", n))
        gp.close()
print("FAILED" if bad else "ok")
