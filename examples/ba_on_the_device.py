#!/usr/bin/env python3
"""What a user of the reference's optBA gets from this library on an MI355X, on the BAL file the
reference ships (ladybug-49-7776), through the C ABI:

  1. the reference's own inner call: CGD over all variables and factors (SSmaxit 25);
  2. the decomposition RDIS reaches: fix a separator, find the connected
     components on the device, solve all of them in one launch -- here as camera / point alternation;
  3. the Levenberg-Marquardt subspace solver with pixel residuals.

  python examples/ba_on_the_device.py [path/to/problem.txt(.gz)]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rdis_amd import capi, problems as P  # noqa: E402


def main():
    pp = P.load_bal(sys.argv[1]) if len(sys.argv) > 1 else P.load_bal()
    nc = int(pp.meta["ncams"])
    ctx = capi.Context(0)
    g = capi.Problem(ctx, pp)
    f0 = g.eval()
    print(f"{nc} cameras, {(pp.nvars - 9 * nc) // 3} points, {pp.nfac} observations; f(x0) = {f0:.6f}")

    # 1. one CGD call over everything (BASELINE config 4)
    plan = capi.Plan(g)
    plan.set_start(pp.x0)
    t = time.perf_counter(); plan.solve(25, 3e-8); r = plan.fetch(); dt = time.perf_counter() - t
    print(f"CGD, all variables, 25 iterations: f = {r.fret[0]:.6f}   ({dt * 1e3:.2f} ms, {int(r.nfeval[0])} evaluations)")

    # 2. alternation through device-side component labelling
    g.set_x(pp.x0)
    fixed_pts = np.zeros(pp.nvars, np.uint8); fixed_pts[9 * nc:] = 1
    fixed_cams = np.zeros(pp.nvars, np.uint8); fixed_cams[:9 * nc] = 1
    plans = [capi.Plan(g, *g.components(a)) for a in (fixed_pts, fixed_cams)]
    t = time.perf_counter()
    for rnd in range(10):
        for pl in plans:
            pl.set_start(None)          # from the currently assigned values
            pl.solve(25, 3e-8)
            res = pl.fetch()
        print(f"  alternation round {rnd + 1:2d}: f = {g.eval():.6f}")
    print(f"10 rounds (49 camera components, then {len(plans[1].fetch().fret)} point components): {(time.perf_counter() - t) * 1e3:.1f} ms")

    # 3. Levenberg-Marquardt, pixel residuals
    g.set_x(pp.x0)
    t = time.perf_counter(); lm = g.lm_optimize(maxiters=25, model=2); dt = time.perf_counter() - t
    print(f"LM (pixel residuals), 25 iterations: f = {lm.fret:.6f}   ({dt * 1e3:.2f} ms, {lm.nsolve} damped solves)")


if __name__ == "__main__":
    main()
