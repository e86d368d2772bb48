"""Where does the device's population of end values part from the oracle's (ladybug 5 / 30: KS 0.15 at n = 512, with every
rounding switch tried -- DESIGN.md section 6)?  For N one-ulp starts, the objective after k = 1 .. 25 CG iterations on the
device and by the oracle FROM THE SAME START: per k the median and the quartiles of (f_dev - f_orc) / f_orc and the share of
starts where the device ends higher.  A rounding-level difference is symmetric (share ~ 0.5, median ~ 0) and grows with k; a
systematic one shows as a one-sided shift from some k on."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from concurrent.futures import ThreadPoolExecutor
from oracle import oracle as O
from rdis_amd import capi, problems as P
if os.environ.get("RDIS_PROBE_LIB"):
    capi.LIB_PATH = os.path.abspath(os.environ["RDIS_PROBE_LIB"])
N = int(sys.argv[1]) if len(sys.argv) > 1 else 96
pp = P.load_bal(ncams=5, npts=30)
fx = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "golden", "end_values.json")))


def start(k):
    rng = np.random.default_rng([fx["seed"], 100000 + k])
    return np.nextafter(pp.x0, np.where(rng.random(pp.x0.shape) < 0.5, -np.inf, np.inf))


ctx = capi.Context(0)
g = capi.Problem(ctx, pp)
fv, fc = np.arange(pp.nvars, dtype=np.int64), np.arange(pp.nfac, dtype=np.int64)
plan = capi.Plan(g, np.array([0, len(fv)]), fv, np.array([0, len(fc)]), fc)
for kv in sys.argv[2:]:
    k, v = kv.split("=")
    plan.set_option(k, int(v))
starts = [start(k) for k in range(N)]
dev = np.zeros((N, 25)); itd = np.zeros((N, 25), int)
for i, x in enumerate(starts):
    for k in range(1, 26):
        plan.set_start(x); plan.solve(k, 3e-8); r = plan.fetch()
        dev[i, k - 1] = r.fret[0]; itd[i, k - 1] = r.iters[0]


def orc(i):
    return [O.OracleProblem(pp, emulate_stale_cache=False).cgd(x=starts[i], maxiters=k, ftol=3e-8).fret for k in range(1, 26)]


with ThreadPoolExecutor(8) as ex:
    ora = np.array(list(ex.map(orc, range(N))))
rel = (dev - ora) / ora
print("k   median rel diff   q25        q75        share dev > orc   median |rel|   f_orc median   f_dev median")
for k in range(25):
    d = rel[:, k]
    print("%2d  %+.3e   %+.3e  %+.3e   %.2f             %.2e     %.6f   %.6f" % (
        k + 1, np.median(d), np.quantile(d, 0.25), np.quantile(d, 0.75), np.mean(d > 0), np.median(np.abs(d)), np.median(ora[:, k]), np.median(dev[:, k])))
