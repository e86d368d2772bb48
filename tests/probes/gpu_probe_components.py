"""exploratory: connected-components labelling, device vs CPU oracle, wall time per call"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from rdis_amd import problems as P, capi
from oracle import oracle as O
ctx = capi.Context(0)
for name, pp in (("ladybug", P.load_bal()), ("synthetic 64x(49,7776)", P.make_synthetic_ba(64, 49, 7776, obs_per_pt=4))):
    g = capi.Problem(ctx, pp); o = O.OracleProblem(pp)
    nc = int(pp.meta.get("ncams", 0)) if pp.ncomp == 1 else 0
    for label, a in (("cameras fixed", None), ("nothing fixed", np.zeros(pp.nvars, np.uint8))):
        if a is None:
            a = np.zeros(pp.nvars, np.uint8)
            if pp.ncomp == 1: a[:9 * nc] = 1
            else: a[np.arange(pp.nvars) % 23769 < 441] = 1
        g.components(a)
        t = time.perf_counter(); reps = 5
        for _ in range(reps): r = g.components(a)
        dt = (time.perf_counter() - t) / reps
        t = time.perf_counter(); ro = o.components(a); dto = time.perf_counter() - t
        print("%-24s %-14s N %9d F %9d -> %7d components: device %.2f ms (incl. mask upload, list download), CPU oracle %.2f ms, equal %s" % (
            name, label, pp.nvars, pp.nfac, len(r[0]) - 1, dt * 1e3, dto * 1e3, all(np.array_equal(x, y) for x, y in zip(r, ro))))
