"""exploratory: cooperative solver on full ladybug -- kernel time, per-evaluation cost, speculation
hit counts (timing build), and the replay check (bit-identical decisions against the oracle)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from rdis_amd import problems as P, capi
from oracle import oracle as O
if os.environ.get("RDIS_PROBE_LIB"):
    capi.LIB_PATH = os.path.abspath(os.environ["RDIS_PROBE_LIB"])
ctx = capi.Context(0)
which = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else "full"
pp = (P.load_bal() if which == "full" else P.load_bal(ncams=49, npts=500) if which == "49_500" else P.load_bal(ncams=5, npts=30)).single_component()
g = capi.Problem(ctx, pp)
plan = capi.Plan(g)
if which != "full":
    plan.set_option("coop_min_factors", 64)
for a in sys.argv[1:]:
    if a.startswith("--opt="):
        k, v = a[6:].split(":")
        plan.set_option(k, int(v))
plan.set_start(pp.x0)
best = 1e9
for rep in range(5):
    plan.solve(25, 3e-8); r = plan.fetch()
    ms, nl = plan.last_kernel_ms(); best = min(best, ms)
print("%s: kernel %.3f ms (best of 5), fret %.6f nfeval %d ngeval %d iters %d status %d -> %.2f us per evaluation, %.0f it/s" % (
    which, best, r.fret[0], r.nfeval[0], r.ngeval[0], r.iters[0] + 1, r.status[0], best * 1e3 / r.nfeval[0], 25e3 / best))
tm = plan.debug_counters()
if tm[7] > 0:
    nx = max(int(tm[5]), 1)
    print("   timing build: exchanges %d sweeps %d; spec hits %d, guesses evaluated %d" % (tm[5], tm[6], tm[18], tm[19]))
    print("   cycles/exchange: compute %.0f publish %.0f sweep %.0f | step %.0f hand-over %.0f" % (tm[0] / nx, tm[2] / nx, tm[3] / nx, tm[8] / nx, tm[9] / nx))
    names = ["F", "FD", "GRAD", "-", "-", "LINE_END"]
    print("   handlers: " + "  ".join("%s %d x %.0f" % (nm, tm[22 + i], tm[12 + i] / max(int(tm[22 + i]), 1)) for i, nm in enumerate(names) if nm != "-"))
    ng = max(int(tm[24]), 1)
    print("   gradient: partials+scatter %.0f, barrier %.0f, gather %.0f cycles; whole kernel %d cycles" % (tm[20] / ng, tm[21] / ng, tm[30] / ng, tm[7]))
if "--replay" in sys.argv:
    plan.set_option("trace_records", 8192); plan.set_option("dump_iters", 25)
    plan.set_start(pp.x0); plan.solve(25, 3e-8); r2 = plan.fetch()
    tr, n = plan.get_trace(0, 8192)
    for d in ("refchain", "adjoint"):
        rep = O.OracleProblem(pp, derivative=d).replay(tr, x=pp.x0, maxiters=25, vdump=plan.get_vectors(0, 25))
        print("   replay (%s): records %d consumed %d step_mismatches %d tag_mismatches %d underrun %d fret_equal %s max_f_rel_near %.2e max_slope_rel_near %.2e" % (
            d, n, rep.consumed, rep.step_mismatches, rep.tag_mismatches, rep.underrun, rep.fret == r2.fret[0], rep.max_f_rel_near, rep.max_slope_rel_near))
    print("   same result with tracing on:", r2.fret[0] == r.fret[0], int(r2.nfeval[0]) == int(r.nfeval[0]))
