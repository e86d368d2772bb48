"""the point-major streaming solver (solver_ptm.hpp) on N components of ladybug's size (49 cameras x 7776 points x 4
observations, bench.py's synthetic-L / strong-scaling shape): kernel time per option set, the algorithmic HBM figure,
and the objective -- for A/B runs of two builds (RDIS_PROBE_LIB=path/to/other/librdis_hip.so).

    python tools/gpu_probe_ptm.py [ncomp] [--set k=v,k=v ...]      (one solve chain per --set; none = the defaults)
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rdis_amd import problems as P, capi
if os.environ.get("RDIS_PROBE_LIB"):
    capi.LIB_PATH = os.path.abspath(os.environ["RDIS_PROBE_LIB"])
args = [a for a in sys.argv[1:] if not a.startswith("--")]
ncomp = int(args[0]) if args else 256
sets = []
it = iter(sys.argv[1:])
for a in it:
    if a == "--set":
        kv = next(it)
        sets.append(dict((k, int(v)) for k, v in (x.split("=") for x in kv.split(",") if x)))
if not sets:
    sets = [{}]
t = time.time()
pp = P.make_synthetic_ba(ncomp, 49, 7776, obs_per_pt=4)
print("lib %s: %d comps, %d factors, %d vars (built in %.1f s)" % (capi.LIB_PATH if hasattr(capi, "LIB_PATH") else "?", pp.ncomp, pp.nfac, pp.nvars, time.time() - t), flush=True)
ctx = capi.Context(0)
g = capi.Problem(ctx, pp)
F, N = pp.nfac // pp.ncomp, pp.nvars // pp.ncomp
for opts in sets:
    plan = capi.Plan(g)
    try:
        for k, v in opts.items():
            plan.set_option(k, v)
        plan.set_start(pp.x0[pp.comp_free_vid])
        best = 1e9
        for rep in range(3):
            plan.solve(25, 3e-8); r = plan.fetch(); best = min(best, plan.last_kernel_ms()[0])
        ab = float(np.sum((r.nfeval - r.ngeval).clip(0) * (24 * F + 8 * N + 8) + r.ngeval * (24 * F + 16 * N + 8)))
        print("%-40s kernel %8.3f ms  %6.1f GB/s algorithmic (%.3f of 8 TB/s)  group %d  evals/comp %.1f  objective %.10g  exits %s" % (
            opts, best, ab / best / 1e6, ab / best / 1e6 / 8000.0, plan.info("point_major_group"), r.nfeval.mean(), r.fret.sum(),
            dict(zip(*np.unique(r.status & 0xFF, return_counts=True)))), flush=True)
        if "--waves" in sys.argv:    # a -DRDIS_COOP_TIMING=2 build: per wave, cycles of its factors and from the trial's start to the barrier
            tm = plan.debug_counters()
            print("   factors by wave (kilo-cycles over the solve): " + " ".join("%d" % (v // 1000) for v in tm[:16] if v))
            print("   start -> barrier by wave:                      " + " ".join("%d" % (v // 1000) for v in tm[16:] if v), flush=True)
        if "--stamps" in sys.argv:   # a -DRDIS_COOP_TIMING build: cycle stamps of the launch's first workgroup (solver_ptm.hpp)
            tm = plan.debug_counters()
            n = max(int(tm[3]), 1)
            print("   first workgroup %.3f ms at 2.4 GHz; value+slope trials %d: cameras %.0f, factors %.0f, sums %.0f cycles each" % (tm[7] / 2.4e6, tm[3], tm[0] / n, tm[1] / n, tm[2] / n))
            print("   the factor phase by wave: fastest %.0f, mean %.0f, slowest %.0f cycles" % (tm[6] / n, tm[21] / n, tm[20] / n))
            steps = max(int(tm[22] + tm[23] + tm[24] + tm[27]), 1)
            ng = max(int(tm[10]), 1)
            print("   control step %.0f cycles, hand-over %.0f (x%d requests); gradient (x%d): rounds %.0f, after %.0f cycles" % (tm[8] / steps, tm[9] / steps, steps, tm[10], tm[4] / ng, tm[5] / ng))
            nr = 44   # (rounds of a gradient at the bench's shape, 768 lanes: the longest wave's slots)
            print("   a gradient's round (first wave, %d rounds a gradient): its factor %.0f, wait for the others %.0f, staging + table %.0f, wait %.0f, sums %.0f cycles" % ((nr,) + tuple(tm[i] / max(ng * nr, 1) for i in (15, 16, 18, 19, 11))))
            print("   per request kind: " + "  ".join("%s %d x %.0f" % (nm, tm[22 + i], tm[12 + i] / max(int(tm[22 + i]), 1)) for i, nm in ((0, "value"), (1, "value+slope"), (2, "gradient"), (5, "line end"))), flush=True)
    except Exception as e:   # an option this build does not know
        print("%-40s failed: %s" % (opts, e), flush=True)
    plan.close()
