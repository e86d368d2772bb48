"""Generates tests/golden/sinusoid_start.json: a start state for BASELINE config 2 (the default
high-dimensional sinusoid, 121 variables) drawn uniformly over the FULL variable domain
(+-62.83...), as the reference's optSinusoid does (src/optimize_sinusoid.cpp:154-165, there with
boost::mt19937(834725927), which is not available here: this fixture pins the start instead).

    python tests/golden/make_sinusoid_start.py
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rdis_amd import problems as P  # noqa: E402

pp = P.make_high_dim_sinusoid()
rng = np.random.Generator(np.random.PCG64(834725927))
x0 = pp.lo + rng.random(pp.nvars) * (pp.hi - pp.lo)
out = {"_provenance": "numpy PCG64(834725927), uniform over [lo, hi] of make_high_dim_sinusoid(); see make_sinusoid_start.py",
       "nvars": int(pp.nvars), "lo": float(pp.lo[0]), "hi": float(pp.hi[0]), "x0": [float(v) for v in x0]}
with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "sinusoid_start.json"), "w") as fh:
    json.dump(out, fh, indent=0)
print("wrote", len(x0), "values in [%.4f, %.4f]" % (x0.min(), x0.max()))
