"""ladybug 5 / 30, the first line minimisations from x0: the device's trace next to the oracle's own run -- record by record
(tag, step, value, slope) -- to see where the two part at the 1e-8 level (tests/probes/gpu_probe_prefix_population.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import oracle as O
from rdis_amd import capi, problems as P
if os.environ.get("RDIS_PROBE_LIB"):
    capi.LIB_PATH = os.path.abspath(os.environ["RDIS_PROBE_LIB"])
K = int(sys.argv[1]) if len(sys.argv) > 1 else 2
SK = int(os.environ.get("START", "-1"))   # START=k: the k-th one-ulp start of the population probes instead of x0
pp = P.load_bal(ncams=5, npts=30)
if SK >= 0:
    import json
    fx = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "golden", "end_values.json")))
    rng = np.random.default_rng([fx["seed"], 100000 + SK])
    pp.x0 = np.nextafter(pp.x0, np.where(rng.random(pp.x0.shape) < 0.5, -np.inf, np.inf))
ctx = capi.Context(0)
g = capi.Problem(ctx, pp)
fv, fc = np.arange(pp.nvars, dtype=np.int64), np.arange(pp.nfac, dtype=np.int64)
plan = capi.Plan(g, np.array([0, len(fv)]), fv, np.array([0, len(fc)]), fc)
plan.set_option("trace_records", 4096)
for kv in sys.argv[2:]:
    k, v = kv.split("=")
    plan.set_option(k, int(v))
plan.set_start(pp.x0); plan.solve(K, 3e-8); r = plan.fetch()
td = plan.get_trace(0, 4096)[0]
o = O.OracleProblem(pp, emulate_stale_cache=False)
to, _n = o.record(x=pp.x0, maxiters=K)
print("device fret %.15g nfeval %d; oracle records %d, device records %d" % (r.fret[0], r.nfeval[0], len(to), len(td)))
n = max(len(to), len(td))
if os.environ.get("COMPACT"):
    rel = lambda x, y: abs(x - y) / max(abs(y), 1e-300)
    for i in range(min(len(to), len(td))):
        a, b = td[i], to[i]
        print("%3d tag %d/%d a %+.6e  rel diff: a %.1e  f %.1e  slope %.1e" % (i, a[0], b[0], b[1], rel(a[1], b[1]), rel(a[2], b[2]), rel(a[3], b[3]) if b[0] == 2 and a[0] == 2 else float("nan")))
    sys.exit(0)
for i in range(n):
    a = td[i] if i < len(td) else [np.nan] * 4
    b = to[i] if i < len(to) else [np.nan] * 4
    flag = "" if (a[0] == b[0] and a[1] == b[1]) else "   <-- step differs"
    print("%3d  dev tag %2.0f a %+.17e f %.15e s %+.6e | orc tag %2.0f a %+.17e f %.15e s %+.6e%s" % (i, a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3], flag))
