#!/usr/bin/env python3
"""Adds "end_values_slope_by_factor" to tests/golden/end_values.json: the oracle's end values over the fixture's own one-ulp
starts (tests/golden/make_end_values.py: start(k)) with ONE thing changed -- a trial's slope added factor by factor,
sum_f (sum_k partial_fk xi_k), the association the device's solvers use, instead of the reference's gradient times direction,
sum_v (sum_f partial_fv) xi_v (Df1dim::df, minimize_nrc.h:439-447).  Switch: ro_set_experiment(2), oracle/rdis_oracle.c; values,
gradients and every decision rule are the reference's.  The GPU test compares the device's population under the reference's
rounding with THIS sample at the plain two-sample critical value -- no allowance -- on full ladybug, where the cooperative
solvers keep their own association (tests/test_gpu_solver.py::test_end_values_distribution_matches_oracle).

And "end_values_slope_by_factor_tree" (full ladybug): the same association with the factors' terms -- and the objective's --
added as a balanced tree instead of in list order (ro_set_experiment(6), sum_order "pairwise"): the shape of a parallel
reduction.  Only the ORDER of two sums differs from the sample above, and the two part with KS 0.19: on this problem the
population of end values follows the last-place rounding of a trial's slope whatever its source (reference association
against list order 0.10, against the tree 0.18, against contraction 0.21) -- the oracle family's own spread is what a
parallel implementation's sample is measured against.

    python tests/golden/make_end_values_slope.py      # ~14 minutes on 8 cores; the other entries of the file are kept
"""
import json
import os
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from oracle import oracle as O                       # noqa: E402
from rdis_amd import problems as P                   # noqa: E402
from make_end_values import CONFIGS, start           # noqa: E402


def main():
    path = os.path.join(ROOT, "tests", "golden", "end_values.json")
    with open(path) as fh:
        out = json.load(fh)
    threads = max(1, min(8, len(os.sched_getaffinity(0))))
    try:
        for name, flags, order, keys in (("end_values_slope_by_factor", 2, "list", list(CONFIGS)),
                                         ("end_values_slope_by_factor_tree", 6, "pairwise", ["ladybug_full"])):
            O.lib().ro_set_experiment(flags)
            for key in keys:
                c = CONFIGS[key]
                pp = P.load_bal(ncams=c["ncams"], npts=c["npts"]) if c["ncams"] else P.load_bal()
                with ThreadPoolExecutor(threads) as ex:
                    res = list(ex.map(lambda k: O.OracleProblem(pp, sum_order=order).cgd(x=start(pp.x0, k), maxiters=25, ftol=3e-8).fret, range(c["n"])))
                out[key][name] = [float(v) for v in res]
                v, ref = np.array(res), np.array(out[key]["end_values"])
                print(key, name, "n", len(v), "quartiles", np.quantile(v, [0.25, 0.5, 0.75]), "reference association", np.quantile(ref, [0.25, 0.5, 0.75]))
    finally:
        O.lib().ro_set_experiment(0)
    with open(path, "w") as fh:
        json.dump(out, fh, indent=0)


if __name__ == "__main__":
    main()
