"""exploratory: cooperative solver timing vs configuration + replay report on ladybug full"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from rdis_amd import problems as P, capi
from oracle import oracle as O
if os.environ.get("RDIS_PROBE_LIB"):
    capi.LIB_PATH = os.path.abspath(os.environ["RDIS_PROBE_LIB"])  # A/B builds
if os.environ.get("RDIS_PROBE_LIB"):
    capi.LIB_PATH = os.path.abspath(os.environ["RDIS_PROBE_LIB"])  # A/B builds
ctx = capi.Context(0)
pp = P.load_bal().single_component()
g = capi.Problem(ctx, pp)
for opts in [{"coop_poll_delay": 8}, {"coop_poll_delay": 12}, {"coop_poll_delay": 16}, {"coop_poll_delay": 20}, {"coop_poll_delay": 24}, {"coop_threads": 128}]:
    plan = capi.Plan(g)
    for k, v in opts.items():
        plan.set_option(k, v)
    plan.set_start(pp.x0)
    for rep in range(3):
        t = time.time(); plan.solve(25, 3e-8); r = plan.fetch(); dt = time.time() - t
    ms, nl = plan.last_kernel_ms()
    print(opts, "wall %.3f ms kernel %.3f ms" % (dt * 1e3, ms), "fret", r.fret[0], "nf/ng", r.nfeval[0], r.ngeval[0], "status", r.status[0],
          "us/eval %.2f" % (ms * 1e3 / r.nfeval[0]))
    if opts.get("coop_min_factors", 1) != 0:
        tm = plan.debug_counters()
        nx = max(int(tm[5]), 1)
        print("   total ticks %d => %.1f MHz tick rate; accounted %.0f%%" % (tm[7], tm[7] / (ms * 1e3), 100.0 * tm[:5].sum() / max(tm[7], 1)))
        print("   publish: combine waves %.0f, release %.0f cycles" % (tm[10] / nx, tm[11] / nx))
        names = ["F", "FD", "GRAD", "CG_START", "LINE_BEGIN", "LINE_END", "CG_REDUCE", "CG_UPDATE"]
        print("   handlers: " + "  ".join("%s %d x %.0f" % (nm, tm[22 + i], tm[12 + i] / max(int(tm[22 + i]), 1)) for i, nm in enumerate(names)))
        ng = max(int(tm[24]), 1)
        print("   gradient: partials+scatter %.0f, barrier %.0f, gather %.0f cycles" % (tm[20] / ng, tm[21] / ng, tm[30] / ng))
        print("   state machine step %.0f, hand-over %.0f cycles per exchange" % (tm[8] / nx, tm[9] / nx))
        print("   cycles/exchange: compute %.0f local-reduce %.0f publish %.0f sweep %.0f tail %.0f | exchanges %d sweeps %d (%.1f per exchange)" % (
            tm[0] / nx, tm[1] / nx, tm[2] / nx, tm[3] / nx, tm[4] / nx, tm[5], tm[6], tm[6] / nx))
    plan.close()
if "--replay" in sys.argv:
    plan = capi.Plan(g)
    plan.set_option("trace_records", 8192); plan.set_option("dump_iters", 25)
    plan.set_start(pp.x0); plan.solve(25, 3e-8); r = plan.fetch()
    tr, n = plan.get_trace(0, 8192)
    rep = O.OracleProblem(pp).replay(tr, x=pp.x0, maxiters=25, vdump=plan.get_vectors(0, 25))
    print(rep)
