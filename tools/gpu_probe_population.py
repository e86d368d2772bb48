"""end values of 25 CG iterations on full ladybug over one-ulp-perturbed starts: the device's solvers, one population each,
against the oracle's committed sample (tests/golden/end_values.json) -- quartiles, two-sample Kolmogorov-Smirnov and
Mann-Whitney.  Which solver runs is chosen by plan options; the starts are those of the distribution test.

    python tools/gpu_probe_population.py [n] [--5_30]
"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from scipy import stats
from rdis_amd import capi, problems as P
if os.environ.get("RDIS_PROBE_LIB"):
    capi.LIB_PATH = os.path.abspath(os.environ["RDIS_PROBE_LIB"])
args = [a for a in sys.argv[1:] if not a.startswith("--")]
n = int(args[0]) if args else 320
small = "--5_30" in sys.argv
key = "ladybug_5_30" if small else "ladybug_full"
fx = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "end_values.json")))
oe = np.array(fx[key]["end_values"])
oc = np.array(fx[key]["end_values_contracted"])
osf = np.array(fx[key]["end_values_slope_by_factor"])   # (tests/golden/make_end_values_slope.py)
pp = P.load_bal(ncams=5, npts=30) if small else P.load_bal()


def start(k):
    rng = np.random.default_rng([fx["seed"], 100000 + k])
    return np.nextafter(pp.x0, np.where(rng.random(pp.x0.shape) < 0.5, -np.inf, np.inf))


ctx = capi.Context(0)
g = capi.Problem(ctx, pp)
fv, fc = np.arange(pp.nvars, dtype=np.int64), np.arange(pp.nfac, dtype=np.int64)
q = lambda v: ("q25 %.4f median %.4f q75 %.4f [%.3f, %.3f]" if small else "q25 %.1f median %.1f q75 %.1f [%.1f, %.1f]") % (*np.quantile(v, [0.25, 0.5, 0.75]), v.min(), v.max())
print("%s: oracle fixture n %d: %s" % (key, len(oe), q(oe)))
print("%s: oracle compiled with contraction:  %s   KS against the oracle %.3f" % (key, q(oc), stats.ks_2samp(oc, oe).statistic))
print("%s: oracle, slope factor by factor:    %s   KS against the oracle %.3f" % (key, q(osf), stats.ks_2samp(osf, oe).statistic))
SETS = {"default (pipelined cooperative)": {},
        "reference rounding (factor_rounding 1)": {"factor_rounding": 1},
        "cooperative, not pipelined": {"coop_pipeline": 0},
        "... with reference rounding": {"coop_pipeline": 0, "factor_rounding": 1},
        "one workgroup, streaming (ptm)": {"coop_min_factors": 0, "coop_group_min_factors": 0, "ptm_stream": 2},
        "one workgroup, plain": {"coop_min_factors": 0, "coop_group_min_factors": 0, "ptm_stream": 0, "lds_resident": 0}}
if small:
    SETS = {"default (LDS-resident)": {}, "reference rounding (factor_rounding 1)": {"factor_rounding": 1},
            "plain batch solver": {"lds_resident": 0}}
if "--only" in sys.argv:
    want = sys.argv[sys.argv.index("--only") + 1]
    SETS = {k: v for k, v in SETS.items() if want in k}
for name, opts in SETS.items():
    plan = capi.Plan(g, np.array([0, len(fv)]), fv, np.array([0, len(fc)]), fc)
    for k, v in opts.items():
        plan.set_option(k, v)
    de, nfe, kms = [], [], 0.0
    for k in range(n):
        plan.set_start(start(k))
        plan.solve(25, 3e-8)
        r = plan.fetch()
        de.append(r.fret[0]); nfe.append(r.nfeval[0]); kms += plan.last_kernel_ms()[0]
    info = {k: plan.info(k) for k in ("components_cooperative", "components_lds", "components_point_major", "components_plain", "pipelined")}
    plan.close()
    de = np.array(de)
    ks = stats.ks_2samp(de, oe)
    kc = stats.ks_2samp(de, oc)
    mw = stats.mannwhitneyu(de, oe)
    kf = stats.ks_2samp(de, osf)
    if os.environ.get("RDIS_PROBE_SAMPLE"):   # (another sample of end values, a JSON list: an experiment's oracle run)
        for path in os.environ["RDIS_PROBE_SAMPLE"].split(":"):
            ox = np.array(json.load(open(path)))
            kx = stats.ks_2samp(de, ox)
            print("   against %s (n %d: %s): KS %.3f (p %.3f); that sample against the oracle's: KS %.3f" % (path, len(ox), q(ox), kx.statistic, kx.pvalue, stats.ks_2samp(ox, oe).statistic))
    print("%-40s n %d: %s  evals %.0f  kernel %.3f ms  KS vs oracle %.3f (p %.3f), vs contracted oracle %.3f (p %.3f), vs by-factor oracle %.3f (p %.3f)  MWU p %.3f  %s" % (
        name, n, q(de), np.mean(nfe), kms / n, ks.statistic, ks.pvalue, kc.statistic, kc.pvalue, kf.statistic, kf.pvalue, mw.pvalue, info), flush=True)
