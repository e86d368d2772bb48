"""Writes tests/golden/parity_end_values.json: what the CPU oracle returns on BASELINE configs 3 and 4 (25 CG iterations from
x0) with its three named switches for the device's factor arithmetic on (oracle/rdis_oracle.h: RO_ARITH_RECIPROCAL,
RO_ARITH_SINCOS_ANGLE, RO_BA_DERIV_ADJOINT_DEVICE), with and without the reference's stale factor cache.

A fourth switch, RO_SUM_TOPOLOGY_COOPERATIVE (the device's sum trees), makes the oracle return what the DEFAULT cooperative path
-- the benchmarked one -- returns (the *_default_path entries).

With the switches on no C-library transcendental is on the path (sqrt, division and fma are exactly rounded), so the numbers
do not depend on the machine: the CPU suite pins them (tests/test_oracle.py), and under -m gpu the device's parity option
(plan option factor_rounding = 1) must return the same bits, run live against the oracle (tests/test_gpu_parity.py).

    python tests/golden/make_parity_end_values.py
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O               # noqa: E402
from rdis_amd import problems as P           # noqa: E402

out = {"_what": __doc__.split("\n\n")[0].replace("\n", " ")}
for key, (nc, npt) in (("ladybug_5_30", (5, 30)), ("ladybug_full", (0, 0))):
    for stale in (True, False):
        pp = P.load_bal(ncams=nc, npts=npt).single_component()
        r = O.OracleProblem.device_parity(pp, emulate_stale_cache=stale).cgd(x=pp.x0, maxiters=25)
        out[f"{key}{'_stale_cache' if stale else ''}"] = {
            "ncams": nc, "npts": npt, "maxiters": 25, "emulate_stale_cache": stale, "fret": r.fret, "delta": r.delta,
            "iters": r.iters, "status": r.status, "nfeval": r.nfeval, "ngeval": r.ngeval, "x_0_2": list(r.x[:3]),
            "x_last": float(r.x[-1])}
# ... and the DEFAULT cooperative path (the benchmarked headline solve): the fourth switch on too -- the device's sum trees
# (RO_SUM_TOPOLOGY_COOPERATIVE) -- and no stale cache
for key, (nc, npt) in (("ladybug_full", (0, 0)), ("ladybug_49_500", (49, 500))):
    pp = P.load_bal(ncams=nc, npts=npt).single_component()
    r = O.OracleProblem.device_default(pp).cgd(x=pp.x0, maxiters=25)
    out[f"{key}_default_path"] = {
        "ncams": nc, "npts": npt, "maxiters": 25, "emulate_stale_cache": False, "sum_topology": "cooperative", "fret": r.fret, "delta": r.delta,
        "iters": r.iters, "status": r.status, "nfeval": r.nfeval, "ngeval": r.ngeval, "x_0_2": list(r.x[:3]), "x_last": float(r.x[-1])}
with open(os.path.join(ROOT, "tests", "golden", "parity_end_values.json"), "w") as fh:
    json.dump(out, fh, indent=1)
print(json.dumps(out, indent=1))
