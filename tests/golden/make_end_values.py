#!/usr/bin/env python3
"""Generates tests/golden/end_values.json: the end value of CGDSubspaceOptimizer::optimize (SSmaxit 25,
ftol 3e-8) on BASELINE configs 3 and 4 from starts that differ from the BAL file's by one unit in the last
place per variable, computed by the reference-faithful CPU oracle (oracle/rdis_oracle.c: the reference's
forward-chain derivative, sequential sums, the stale-cache rule of Variable::assign -- it reproduces the
reference's recorded runs bit for bit, tests/test_oracle.py).  25 unconverged CG iterations are a chaotic
map of the start, so these values are a sample of the DISTRIBUTION the reference itself draws from; the
GPU tests compare the device's sample with it (two-sample Kolmogorov-Smirnov), bench.py prints both.

Start k (k >= 1) = nextafter(x0, +-inf) per variable with signs from numpy's default_rng([20260929, k]);
start 0 is x0 itself.  Also written: the objective after k = 1 .. 25 iterations from x0 (the growth curve
test compares the device's prefix values with these), once more with the oracle's OTHER derivative formula
(the adjoint sweep: per-factor rows agree with the reference's forward chain to 2e-14 of the row's largest
entry), and from the one-ulp starts 1 .. 8 -- how far two correct roundings of the same algorithm part,
iteration by iteration.  (With the derivative exchanged the VALUES f stay bit-identical and only slopes
move: that curve is calm.  From a one-ulp start the values differ in the last place too, as they do
between the oracle and the device, and Brent's comparisons of nearly equal values flip: on the full
problem the trajectories are 1e-3 .. 1e-2 apart after three iterations.)

Also written (round 4): "end_values_contracted" -- the same starts through the same oracle COMPILED WITH CONTRACTION
(oracle/Makefile: liboracle_contracted.so, -ffp-contract=fast -mfma; every a * b + c the compiler sees becomes one fused
operation, as in the device's factor arithmetic).  Same algorithm, same sums, an equally valid rounding -- and a
different distribution: on full ladybug the two samples (320 each) part with a two-sample Kolmogorov-Smirnov statistic
of 0.21 (p = 1e-5).  The distribution of end values over one-ulp starts is a property of the evaluator's rounding, not
of the algorithm alone; the device tests measure the device's distance from the oracle against this distance of the
oracle from itself.  (Measured and not the cause: the order of the objective's sum -- ro_set_sum_order: end values equal
to 1e-15 --, the stale-cache rule, the derivative formula: KS 0.05 / 0.09, p 0.9 / 0.2.)

    python tests/golden/make_end_values.py            # ~12 minutes on 8 cores
"""
import json
import os
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O          # noqa: E402
from rdis_amd import problems as P      # noqa: E402

SEED = 20260929
CONFIGS = {"ladybug_5_30": dict(ncams=5, npts=30, n=512), "ladybug_full": dict(ncams=None, npts=None, n=320)}


def start(x0, k):
    if k == 0:
        return x0
    rng = np.random.default_rng([SEED, k])
    return np.nextafter(x0, np.where(rng.random(x0.shape) < 0.5, -np.inf, np.inf))


def contracted_oracle():
    """a second instance of the oracle module bound to liboracle_contracted.so"""
    import importlib.util
    import subprocess
    odir = os.path.join(ROOT, "oracle")
    subprocess.check_call(["make", "-C", odir, "liboracle_contracted.so"], stdout=subprocess.DEVNULL)
    spec = importlib.util.spec_from_file_location("oracle_contracted", os.path.join(odir, "oracle.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules["oracle_contracted"] = mod
    spec.loader.exec_module(mod)
    mod._LIB = os.path.join(odir, "liboracle_contracted.so")
    mod.build = lambda force=False: None
    return mod


def main():
    OC = contracted_oracle()
    out = {"seed": SEED, "maxiters": 25, "ftol": 3e-8,
           "what": "oracle end values from one-ulp-perturbed starts; start(k) as in tests/golden/make_end_values.py"}
    threads = max(1, min(8, len(os.sched_getaffinity(0))))
    for key, c in CONFIGS.items():
        pp = P.load_bal(ncams=c["ncams"], npts=c["npts"]) if c["ncams"] else P.load_bal()

        def run(k):
            r = O.OracleProblem(pp).cgd(x=start(pp.x0, k), maxiters=25, ftol=3e-8)
            return r.fret, r.iters, r.status, r.nfeval

        def prefix(k):
            return O.OracleProblem(pp).cgd(x=pp.x0, maxiters=k, ftol=3e-8).fret

        def prefix_adjoint(k):   # the same oracle with its second derivative formula: rows agree to 2e-14, nothing else changes
            return O.OracleProblem(pp, derivative="adjoint").cgd(x=pp.x0, maxiters=k, ftol=3e-8).fret
        def run_contracted(k):
            return OC.OracleProblem(pp).cgd(x=start(pp.x0, k), maxiters=25, ftol=3e-8).fret
        with ThreadPoolExecutor(threads) as ex:
            res = list(ex.map(run, range(c["n"])))
            resc = list(ex.map(run_contracted, range(c["n"])))
            pre = list(ex.map(prefix, range(1, 26)))
            pre2 = list(ex.map(prefix_adjoint, range(1, 26)))
            # ... and from eight one-ulp starts (the rounding of f itself differs then, as it does on the device)
            pre3 = list(ex.map(lambda a: O.OracleProblem(pp).cgd(x=start(pp.x0, a[0]), maxiters=a[1], ftol=3e-8).fret,
                               [(s_, k) for s_ in range(1, 9) for k in range(1, 26)]))
        out[key] = {"end_values": [float(r[0]) for r in res], "end_values_contracted": [float(v) for v in resc],
                    "iters": [int(r[1]) for r in res],
                    "status": [int(r[2]) for r in res], "nfeval": [int(r[3]) for r in res],
                    "prefix_values_from_x0": [float(v) for v in pre],
                    "prefix_values_from_x0_adjoint_derivative": [float(v) for v in pre2],
                    "prefix_values_from_ulp_starts_1_to_8": [[float(v) for v in pre3[25 * i:25 * i + 25]] for i in range(8)]}
        v = np.array(out[key]["end_values"])
        print(key, "n", len(v), "min/q25/median/q75/max", v.min(), *np.quantile(v, [0.25, 0.5, 0.75]), v.max(), "unperturbed", v[0])
    with open(os.path.join(ROOT, "tests", "golden", "end_values.json"), "w") as fh:
        json.dump(out, fh, indent=0)


if __name__ == "__main__":
    main()
