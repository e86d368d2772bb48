"""exploratory: pipelined cooperative solver (solver_pipe.hpp) on full ladybug -- kernel time and, with a
-DRDIS_COOP_TIMING build (RDIS_PROBE_LIB), the cycle counters of the control wave and of a lane wave"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rdis_amd import problems as P, capi
if os.environ.get("RDIS_PROBE_LIB"):
    capi.LIB_PATH = os.path.abspath(os.environ["RDIS_PROBE_LIB"])
ctx = capi.Context(0)
pp = P.load_bal().single_component()
g = capi.Problem(ctx, pp)
plan = capi.Plan(g)
for a in sys.argv[1:]:
    if a.startswith("--opt="):
        k, v = a[6:].split(":")
        plan.set_option(k, int(v))
plan.set_start(pp.x0)
best = 1e9
for rep in range(7):
    plan.solve(25, 3e-8); r = plan.fetch()
    ms, nl = plan.last_kernel_ms(); best = min(best, ms)
print("%s kernel %.3f ms (best of 7), fret %.6f nfeval %d ngeval %d status %d -> %.0f it/s" % (
    " ".join(sys.argv[1:]), best, r.fret[0], r.nfeval[0], r.ngeval[0], r.status[0], 25e3 / best))
tm = plan.debug_counters()
if tm[7] > 0:
    nh, nm = max(int(tm[16]), 1), max(int(tm[17]), 1)
    print("   stepper: value+slope steps %d guessed (waits %.0f cycles for the sums), %d fresh (waits %.0f) | step+post %.0f cycles | guesses posted %d" % (
        tm[16], tm[20] / nh, tm[17], tm[21] / nm, tm[8] / max(int(tm[16] + tm[17] + tm[22] + tm[24]), 1), tm[19]))
    names = ["value", "value+slope", "gradient", "line start"]
    print("   stepper, cycles serving a request after its post: " + "  ".join("%s %d x %.0f" % (nm_, tm[22 + i], tm[12 + i] / max(int(tm[22 + i]), 1)) for i, nm_ in enumerate(names)))
    print("   collector: %d sweeps completed (%.0f cycles each, %d polls in all), %d given up" % (tm[5], tm[1] / max(int(tm[5]), 1), tm[6], tm[4]))
    ne = max(int(tm[10] + tm[11]), 1)
    print("   lanes: %d requests + %d guesses evaluated: arithmetic %.0f, reduce+publish %.0f cycles each; waiting for requests %d cycles of %d" % (
        tm[10], tm[11], tm[0] / ne, tm[2] / ne, tm[9], tm[7]))
    print("   collector: waited %d times for its own lanes (%.0f cycles each); polling memory %.0f, final reduction %.0f cycles per sweep" % (
        tm[27], tm[26] / max(int(tm[27]), 1), tm[28] / max(int(tm[5]), 1), tm[29] / max(int(tm[5]), 1)))
    print("   lanes gradient: partials+scatter %.0f, barrier %.0f, sums %.0f (x%d)" % (tm[3] / max(int(tm[31]), 1), tm[18] / max(int(tm[31]), 1), tm[30] / max(int(tm[31]), 1), tm[31]))
