"""Connected components of the residual factor graph on the device (rdis_hip_components, the
step before the solver: Component::createChildren) against the CPU oracle.  Index work: exact."""
import numpy as np
import pytest

from oracle import oracle as O
from rdis_amd import capi, problems as P

pytestmark = pytest.mark.gpu


def _same(dev, orc):
    return all(np.array_equal(a, b) for a, b in zip(dev, orc))


@pytest.mark.parametrize("case", ["cameras_fixed", "points_fixed", "46_cameras", "random20", "random60", "random95", "none", "all"])
def test_ladybug_components_equal_oracle(case, gctx):
    pp = P.load_bal()
    rng = np.random.default_rng(17)
    a = np.zeros(pp.nvars, np.uint8)
    if case == "cameras_fixed": a[:441] = 1
    elif case == "points_fixed": a[441:] = 1
    elif case == "46_cameras": a[:9 * 46] = 1            # the separator PaToH found
    elif case.startswith("random"): a[rng.random(pp.nvars) < int(case[6:]) / 100.0] = 1
    elif case == "all": a[:] = 1
    dev = capi.Problem(gctx, pp).components(a)
    orc = O.OracleProblem(pp).components(a)
    assert _same(dev, orc)
    cams, pts = P.ba_alternation_plans(pp)
    if case == "cameras_fixed": assert _same(dev, pts)   # the classic alternation: 7776 point components
    if case == "points_fixed": assert _same(dev, cams)
    if case == "none": assert len(dev[0]) == 2 and dev[0][1] == pp.nvars and dev[2][1] == pp.nfac
    if case == "all": assert len(dev[0]) == 1 and len(dev[1]) == 0 and len(dev[3]) == 0
    if case == "46_cameras":
        # three cameras and every point they see form one component; all other points are alone
        sizes = np.diff(dev[0])
        assert sizes[-1] > 27 and np.all(sizes[:-1] == 3) and (sizes[-1] - 27) % 3 == 0


def test_synthetic_decomposition_is_recovered(gctx):
    pp = P.make_synthetic_ba(300, 3, 40)
    g = capi.Problem(gctx, pp)
    dev = g.components(np.zeros(pp.nvars, np.uint8))
    assert _same(dev, O.OracleProblem(pp).components(np.zeros(pp.nvars, np.uint8)))
    assert len(dev[0]) - 1 == 300 and np.all(np.diff(dev[0]) == 147)
    # the generator's own decomposition, component by component (equal sizes: ordered by smallest id)
    assert np.array_equal(dev[1], pp.comp_free_vid) and np.array_equal(dev[3], pp.comp_fac_id)


def test_nonlinear_products_and_isolated_variables(gctx):
    rng = np.random.default_rng(3)
    pp = P.make_high_dim_sinusoid()
    for frac in (0.0, 0.3, 0.8):
        a = (rng.random(pp.nvars) < frac).astype(np.uint8)
        assert _same(capi.Problem(gctx, pp).components(a), O.OracleProblem(pp).components(a))
    terms = [(2.0, [(0, 1.0, 0.0, 0), (1, 1.0, 0.0, 0)]), (1.0, [(3, 2.0, 0.0, 0)]), (-7.0, [])]
    q = P._pack_nlp(terms, np.zeros(4), np.full(4, -5.0), np.full(4, 5.0), {})
    dev = capi.Problem(gctx, q).components(np.zeros(4, np.uint8))
    assert list(dev[0]) == [0, 1, 2, 4] and list(dev[1]) == [2, 3, 0, 1] and list(dev[2]) == [0, 0, 1, 2] and list(dev[3]) == [1, 0]


def test_components_feed_the_solver(gctx):
    """labelling -> plan -> one launch: the same result as the hand-built alternation plan, bit for bit"""
    pp = P.load_bal(ncams=49, npts=500)
    g = capi.Problem(gctx, pp)
    a = np.zeros(pp.nvars, np.uint8); a[:441] = 1
    comps = g.components(a)
    r1 = _solve(g, pp, comps)
    r2 = _solve(g, pp, P.ba_alternation_plans(pp)[1])
    assert np.array_equal(r1.fret, r2.fret) and np.array_equal(r1.x, r2.x) and np.array_equal(r1.iters, r2.iters)
    # a mixed separator: components of very different sizes in one plan (isolated-variable components included)
    rng = np.random.default_rng(1)
    a = (rng.random(pp.nvars) < 0.5).astype(np.uint8)
    comps = g.components(a)
    r = _solve(g, pp, comps)
    assert np.all(r.delta <= 0) and np.all((r.status & 0xFF) != 5)
    empty = np.diff(comps[2]) == 0
    assert np.all((r.status[empty] & 0xFF) == 6) and np.all(r.fret[empty] == 0.0)   # empty factor list => 0 (.cpp:26-29)


def _solve(g, pp, comps):
    g.set_x(pp.x0)
    plan = capi.Plan(g, *comps)
    plan.set_start(None)
    plan.solve(25, 3e-8)
    return plan.fetch()
