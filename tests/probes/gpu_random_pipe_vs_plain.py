"""campaign: random single-component shapes, pipelined group (solver_pipe.hpp) against the plain cooperative solver --
values, points, call counts and statuses must be the same bits (shapes are drawn so that both layouts have a wave for
every variable fed by many partials, the one place where their orders of summation could differ)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from rdis_amd import problems as P, capi
ctx = capi.Context(0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 7)
bad = 0
ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 40
for case in range(ncase):
    cams = int(rng.integers(2, 5))
    obs = int(rng.integers(2, cams + 1))
    pts = int(rng.integers(max(300, 2400 // obs), 4000))
    iters = int(rng.integers(2, 9))
    pp = P.make_synthetic_ba(1, cams, pts, obs_per_pt=obs, first_comp=int(rng.integers(0, 100000)))
    comps = (np.array([0, pp.nvars]), np.arange(pp.nvars, dtype=np.int64), np.array([0, pp.nfac]), np.arange(pp.nfac, dtype=np.int64))
    g = capi.Problem(ctx, pp)
    out = []
    for pipe in (0, 1):
        plan = capi.Plan(g, *comps)
        plan.set_option("coop_min_factors", 256); plan.set_option("coop_pipeline", pipe)
        g.set_x(pp.x0); plan.set_start(None); plan.solve(iters, 3e-8); r = plan.fetch()
        out.append((r, g.get_x()))
        plan.close()
    (ra, xa), (rb, xb) = out
    same = (np.array_equal(ra.fret, rb.fret) and np.array_equal(ra.x, rb.x) and np.array_equal(xa, xb) and np.array_equal(ra.nfeval, rb.nfeval)
            and np.array_equal(ra.ngeval, rb.ngeval) and np.array_equal(ra.status, rb.status) and np.array_equal(ra.iters, rb.iters))
    if not same or np.any((rb.status & 0xFF) == 7):
        bad += 1
        print("DIFFERENT: %d cameras, %d points, %d observations each, %d iterations: %.12g / %.12g, evaluations %d / %d, status %s / %s" % (
            cams, pts, obs, iters, ra.fret[0], rb.fret[0], ra.nfeval[0], rb.nfeval[0], ra.status, rb.status))
    g.close()
print("%d shapes, %d different" % (ncase, bad))
