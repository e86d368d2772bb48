"""Parity of the batched factor kernels (through the C ABI) with the CPU oracle
and with the reference's golden values.  fp64; tolerances are stated per test."""
import numpy as np
import pytest

from oracle import oracle as O
from rdis_amd import capi, problems as P

pytestmark = pytest.mark.gpu


def _cases():
    rng = np.random.default_rng(7)
    poly = P.load_poly()
    poly.x0 = np.array([1.0, -2.0])
    sin = P.make_high_dim_sinusoid()
    sin.x0 = rng.uniform(-6.28, 6.28, sin.nvars)
    return {"ladybug_5_30": P.load_bal(ncams=5, npts=30), "ladybug_49_500": P.load_bal(ncams=49, npts=500),
            "ladybug_full": P.load_bal(), "testpoly": poly, "sinusoid": sin,
            "synthetic": P.make_synthetic_ba(20, 3, 40)}


CASES = _cases()


def test_golden_factor0_and_full_gradient(golden, gctx):
    pp = CASES["ladybug_full"]
    g = capi.Problem(gctx, pp)
    gl, g0 = golden["ladybug_full"], golden["ba_factor0"]
    each = g.eval_each(np.array([0, 1, 31842]))
    # |E_dev - E_ref| <= 1e-13 * (E + |obs|^2): the residual is a difference of pixel-scale numbers
    scale = each + np.sum(pp.obs[[0, 1, 31842]] ** 2, axis=1)
    assert np.all(np.abs(each - [g0["E"], gl["E_factor1"], gl["E_factor31842"]]) <= 1e-13 * scale)
    row = g.grad_each_ba(np.array([0]))[0]
    assert np.max(np.abs(row - g0["grad"])) <= 1e-12 * np.max(np.abs(g0["grad"]))
    f, grad = g.eval_grad()
    assert abs(f - gl["f_xinit"]) <= 1e-12 * gl["f_xinit"]
    gmax = np.max(np.abs(grad))
    assert abs(np.linalg.norm(grad) - gl["grad_norm2"]) <= 1e-12 * gl["grad_norm2"]
    assert np.max(np.abs(grad[0:9] - gl["grad_0_8"])) <= 1e-10 * gmax          # 1e-10 rel-to-inf-norm
    assert np.max(np.abs(grad[441:444] - gl["grad_441_443"])) <= 1e-10 * gmax
    assert abs(grad[23768] - gl["grad_23768"]) <= 1e-10 * gmax


@pytest.mark.parametrize("name", list(CASES))
def test_sum_gradient_and_per_factor_parity(name, gctx):
    pp = CASES[name]
    o, g = O.OracleProblem(pp), capi.Problem(gctx, pp)
    fo, go = o.eval(), o.gradient()
    eo = o.eval_each()
    fsum = np.sum(np.abs(eo))
    assert abs(g.eval() - fo) <= 1e-12 * fsum                                  # sum: 1e-12 rel
    f2, gg = g.eval_grad()
    assert f2 == g.eval()                                                       # same kernel, same order
    assert np.max(np.abs(gg - go)) <= 1e-12 * max(np.max(np.abs(go)), 1e-300)   # entries: 1e-12 rel-to-inf-norm
    eg = g.eval_each()
    if pp.kind == 0:
        scale = eo + np.sum(pp.obs ** 2, axis=1)
        assert np.all(np.abs(eg - eo) <= 1e-13 * scale)
        a, b = o.grad_each_ba(), g.grad_each_ba()
        # 12 partials per factor: 1e-12 of the row's largest entry, or -- for the few badly conditioned
        # rows (near-cancelling terms of the rotation derivative) -- a few times what one-ulp changes of
        # the inputs do to the oracle's own row (up to 8e-9 on ladybug's factor 30771)
        rng, allv, noise = np.random.default_rng(1), np.arange(pp.nvars, dtype=np.int64), np.zeros(len(a))
        rowmax = np.max(np.abs(a), axis=1)
        for _ in range(6):
            o.assign(allv, pp.x0 * (1.0 + rng.choice([-1.0, 0.0, 1.0], pp.nvars) * 2.0 ** -52))
            noise = np.maximum(noise, np.max(np.abs(a - o.grad_each_ba()), axis=1) / rowmax)
        o.assign(allv, pp.x0)
        dev = np.max(np.abs(a - b), axis=1) / rowmax
        assert np.all(dev <= np.maximum(1e-12, 8.0 * noise)), (np.argmax(dev / np.maximum(1e-12, 8.0 * noise)), np.max(dev))
    else:
        assert np.max(np.abs(eg - eo)) <= 1e-13 * max(np.max(np.abs(eo)), 1.0)


def test_factor_sublists_and_linearity(gctx):
    pp = CASES["ladybug_full"]
    o, g = O.OracleProblem(pp), capi.Problem(gctx, pp)
    rng = np.random.default_rng(11)
    sub = np.sort(rng.choice(pp.nfac, 5000, replace=False)).astype(np.int64)
    rest = np.setdiff1d(np.arange(pp.nfac, dtype=np.int64), sub)
    f_sub, g_sub = g.eval_grad(sub)
    f_rest, g_rest = g.eval_grad(rest)
    f_all, g_all = g.eval_grad()
    assert abs(f_sub - o.eval(sub)) <= 1e-12 * abs(f_sub)
    assert np.max(np.abs(g_sub - o.gradient(sub))) <= 1e-12 * np.max(np.abs(g_sub))
    # size-independent property: the objective and gradient are additive over a partition of the factors
    assert abs((f_sub + f_rest) - f_all) <= 1e-12 * f_all
    assert np.max(np.abs((g_sub + g_rest) - g_all)) <= 1e-12 * np.max(np.abs(g_all))
    # a permuted list is the same set: only the summation order changes
    perm = rng.permutation(sub)
    assert abs(g.eval(perm) - f_sub) <= 1e-12 * f_sub
    # empty list
    assert g.eval(np.zeros(0, dtype=np.int64)) == 0.0


def test_assign_constants_and_determinism(gctx):
    pp = CASES["ladybug_49_500"]
    g = capi.Problem(gctx, pp)
    o = O.OracleProblem(pp)
    rng = np.random.default_rng(5)
    vid = np.sort(rng.choice(pp.nvars, 300, replace=False)).astype(np.int64)
    val = pp.x0[vid] * (1 + 1e-3 * rng.standard_normal(300))
    g.set_x(val, vid)
    o.assign(vid, val)
    assert np.array_equal(g.get_x(vid), val)
    assert np.array_equal(g.get_x(), o.get_x())
    f1, g1 = g.eval_grad()
    f2, g2 = g.eval_grad()
    assert f1 == f2 and np.array_equal(g1, g2)                                  # bit-reproducible
    assert abs(f1 - o.eval()) <= 1e-12 * f1


def test_theta_zero_and_nlp_edge_cases(gctx):
    # a camera with zero rotation takes the first-order branch
    pp = P.make_synthetic_ba(1, 2, 5)
    pp.x0[0:3] = 0.0
    # (the reference's forward chain divides by theta without a guard, BundleAdjustmentFactor.cpp:376,
    # 407-409: its oracle restatement returns NaN there like the reference; the independent adjoint
    # derivation has the theta = 0 branch the device implements)
    assert np.any(np.isnan(O.OracleProblem(pp).gradient()[0:3]))
    o, g = O.OracleProblem(pp, derivative="adjoint"), capi.Problem(gctx, pp)
    fo, go = o.eval(), o.gradient()
    f, gg = g.eval_grad()
    assert np.isfinite(f) and np.all(np.isfinite(gg))
    assert abs(f - fo) <= 1e-12 * fo and np.max(np.abs(gg - go)) <= 1e-12 * np.max(np.abs(go))
    # NLP with constants, a general exponent, exponent 0-free terms and a constant-only factor
    terms = [(2.0, [(0, 3.0, 0.5, 0), (1, 1.0, 0.0, 1)]), (-7.0, []), (1.5, [(1, 2.0, -1.0, 1)]),
             (0.25, [(0, 1.0, 0.0, 0), (1, 1.0, 0.0, 0), (2, 2.0, 0.0, 0)])]
    q = P._pack_nlp(terms, np.array([1.3, -0.7, 0.9]), np.full(3, -5.0), np.full(3, 5.0), {})
    o, g = O.OracleProblem(q), capi.Problem(gctx, q)
    f, gg = g.eval_grad()
    assert abs(f - o.eval()) <= 1e-13 * 10 and np.max(np.abs(gg - o.gradient())) <= 1e-13 * 10


def test_nlp_exponential_values_and_refusals(gctx):
    """useExponential of NonlinearProductFactor (src/NonlinearProductFactor.cpp:140, 204): flagged factors evaluate to
    coeff * exp(-product), like the oracle; a gradient or a solve over a list holding one is refused (the
    reference's computeGradient asserts the flag off, :110), over a list without one it runs as before."""
    s = P.make_high_dim_sinusoid()
    rng = np.random.default_rng(3)
    x = rng.uniform(-3.0, 3.0, s.nvars)
    o, g = O.OracleProblem(s), capi.Problem(gctx, s)
    o.assign(None, x)
    g.set_x(x)
    idx = np.arange(s.nfac, dtype=np.int64)
    plain = g.eval_each(idx)
    flags = (rng.random(s.nfac) < 0.4).astype(np.uint8)
    o.set_exponential(flags)
    g.set_exponential(flags)
    e_dev, e_orc = g.eval_each(idx), o.eval_each(idx)
    assert np.array_equal(e_dev[flags == 0], plain[flags == 0])
    assert np.max(np.abs(e_dev - e_orc) / np.maximum(np.abs(e_orc), 1e-300)) <= 8e-16
    assert abs(g.eval() - o.eval()) <= 1e-13 * np.sum(np.abs(e_orc))
    sub = np.flatnonzero(flags)[:7].astype(np.int64)
    assert abs(g.eval(sub) - float(np.sum(e_orc[sub]))) <= 1e-13 * np.sum(np.abs(e_orc[sub]))
    with pytest.raises(capi.RdisHipError, match="exponential"):
        g.eval_grad()
    clean = np.flatnonzero(flags == 0).astype(np.int64)
    f_c, g_c = g.eval_grad(clean)                       # no flagged factor in the list: allowed
    o.set_exponential(None)
    g_o = o.gradient(clean)
    assert np.max(np.abs(g_c - g_o)) <= 1e-13 * max(1.0, np.max(np.abs(g_o)))
    comps = (np.array([0, s.nvars]), np.arange(s.nvars), np.array([0, s.nfac]), idx)    # one component: everything
    with pytest.raises(capi.RdisHipError, match="exponential"):
        capi.Plan(g, *comps)
    with pytest.raises(capi.RdisHipError, match="exponential"):
        g.cgd_batch(*comps, x, maxiters=5, ftol=3e-8)
    g.set_exponential(None)
    assert np.array_equal(g.eval_each(idx), plain)
    f2, g2 = g.eval_grad()
    assert np.all(np.isfinite(g2))
    capi.Plan(g, *comps).close()


def test_rotation_angles_over_the_whole_domain(gctx):
    """The device evaluates sin/cos of the rotation angle with its own routine (factors.hpp
    sincos_angle).  Rotation vectors from 1e-9 rad to the edge of their domain (1000 pi,
    BundleAdjustmentFunction.cpp:419-471), all quadrants, and beyond the routine's exact range
    where it defers to the library: per-factor values and partials against the oracle's libm."""
    rng = np.random.default_rng(11)
    pp = P.make_synthetic_ba(1, 40, 60, obs_per_pt=6)
    ncam = 40
    mags = np.concatenate([10.0 ** rng.uniform(-9, 0, 10), rng.uniform(0.0, 2 * np.pi, 14),
                           np.arange(1, 9) * (np.pi / 4) + rng.uniform(-1e-9, 1e-9, 8),   # quadrant edges
                           rng.uniform(6.0, 3141.0, 6), [3.0e6, 1.0e9]])
    assert len(mags) == ncam
    for c in range(ncam):
        d = rng.normal(size=3)
        pp.x0[9 * c:9 * c + 3] = mags[c] * d / np.linalg.norm(d)
    pp.lo[:], pp.hi[:] = -np.inf, np.inf
    o, g = O.OracleProblem(pp), capi.Problem(gctx, pp)
    idx = np.arange(pp.nfac)
    e_dev, e_orc = g.eval_each(idx), o.eval_each(idx)
    g_dev, g_orc = g.grad_each_ba(idx), o.grad_each_ba(idx)
    # the residual is a difference of pixel-scale numbers: scale with E + |projection|^2 ~ E + |obs|^2
    scale = e_orc + np.sum(pp.obs ** 2, axis=1) + 1.0
    ok = np.isfinite(e_orc)
    assert ok.sum() >= 0.9 * pp.nfac
    assert np.all(np.abs(e_dev - e_orc)[ok] <= 1e-12 * scale[ok])
    gs = np.max(np.abs(g_orc), axis=1, keepdims=True) + 1.0
    assert np.all(np.abs(g_dev - g_orc)[ok] <= 1e-11 * gs[ok])


def test_abi_error_codes(gctx):
    pp = CASES["ladybug_5_30"]
    g = capi.Problem(gctx, pp)
    with pytest.raises(capi.RdisHipError) as e:
        g.eval(np.array([pp.nfac]))                 # factor id out of range
    assert e.value.code == -1
    with pytest.raises(capi.RdisHipError):
        g.set_x(np.array([1.0]), np.array([-1]))
    bad = P.load_bal(ncams=5, npts=30)
    bad.pt_vid0 = bad.pt_vid0.copy()
    bad.pt_vid0[3] = bad.nvars - 1                  # point block would run past the last variable
    with pytest.raises(capi.RdisHipError):
        capi.Problem(gctx, bad)


def test_fused_gradient_on_any_list_and_on_the_device(gctx):
    """rdis_hip_eval_grad's one-pass form (rdis_amd/csrc/grad_fused.hpp) builds per-list tables: lists in another
    order than the loader's point-major one, with repeated factors, cut at odd lengths, a single factor; several
    lists in turn (the problem keeps the tables of up to 64 lists within a byte budget; a list of at most one chunk needs
    none: rdis_hip.hip eval_grad_short), and the device-resident variant."""
    pp = CASES["ladybug_full"]
    o, g = O.OracleProblem(pp), capi.Problem(gctx, pp)
    rng = np.random.default_rng(3)
    lists = [rng.permutation(pp.nfac)[:20000].astype(np.int64),                      # scattered: every block is staged
             np.sort(rng.choice(pp.nfac, 777, replace=False)).astype(np.int64),
             np.concatenate([np.arange(1000), np.arange(500, 1500)]).astype(np.int64),  # 500 factors listed twice
             np.array([31842], dtype=np.int64),
             np.argsort(pp.cam_vid0, kind="stable").astype(np.int64),                  # camera-major
             np.arange(513, dtype=np.int64)]                                           # one entry past a chunk
    for rep in range(2):                                                               # second round: tables rebuilt or found
        for fl in lists:
            f, gg = g.eval_grad(fl)
            go = o.gradient(fl)
            assert f == g.eval(fl)
            assert abs(f - o.eval(fl)) <= 1e-12 * np.sum(np.abs(o.eval_each(fl)))
            assert np.max(np.abs(gg - go)) <= 1e-12 * np.max(np.abs(go))
            assert np.all(gg[go == 0.0] == 0.0)                                        # untouched variables: exactly zero
            f2, g2 = g.eval_grad(fl)
            assert f2 == f and np.array_equal(g2, gg)                                  # fixed order of every sum
    f, gg = g.eval_grad()
    fd, gd = g.eval_grad_device()
    assert np.frombuffer(gctx.copy_to_host(fd, 8), dtype=np.float64)[0] == f
    assert np.array_equal(np.frombuffer(gctx.copy_to_host(gd, 8 * pp.nvars), dtype=np.float64), gg)
    # constants changed between calls: the tables are index work only, the camera records are formed per call
    x1 = pp.x0 * (1 + 1e-3 * rng.standard_normal(pp.nvars))
    g.set_x(x1)
    o.assign(np.arange(pp.nvars, dtype=np.int64), x1)
    f, gg = g.eval_grad()
    go = o.gradient()
    assert np.max(np.abs(gg - go)) <= 1e-12 * np.max(np.abs(go)) and abs(f - o.eval()) <= 1e-12 * f


def test_fused_gradient_with_more_cameras_than_a_chunk_holds(gctx):
    """grad_fused_kernel keeps a chunk's first 56 distinct cameras in its fast rows and sends the rest through the overflow path
    (cur.extra); ladybug has 49 cameras, so only the point side of that path was exercised.  A synthetic problem with 96
    cameras, its factors listed in a camera-SCATTERED order (every chunk of 512 entries meets nearly all 96), and a list that
    cycles through many short and long lists in turn (more than the four tables round 5 kept): value and gradient against
    the oracle, the same bits twice, zeros where no listed factor reads."""
    pp = P.make_synthetic_ba(1, 96, 6000, obs_per_pt=4)
    o, g = O.OracleProblem(pp), capi.Problem(gctx, pp)
    rng = np.random.default_rng(17)
    scattered = rng.permutation(pp.nfac).astype(np.int64)
    assert len(np.unique(pp.cam_vid0[scattered[:512]])) > 56
    lists = [scattered, scattered[:5000], scattered[:513], scattered[:512], scattered[:40], scattered[7:8]]
    lists += [rng.choice(pp.nfac, int(n), replace=False).astype(np.int64) for n in rng.integers(600, 3000, size=8)]   # > 4 lists in turn
    for rep in range(2):
        for fl in lists:
            f, gg = g.eval_grad(fl)
            go = o.gradient(fl)
            assert f == g.eval(fl)
            assert abs(f - o.eval(fl)) <= 1e-12 * np.sum(np.abs(o.eval_each(fl)))
            assert np.max(np.abs(gg - go)) <= 1e-12 * np.max(np.abs(go))
            assert np.all(gg[go == 0.0] == 0.0)
            f2, g2 = g.eval_grad(fl)
            assert f2 == f and np.array_equal(g2, gg)
    # a short list's table-free path gives the bits of the fused pass: the same 512 entries as the head of a longer list whose
    # other entries read other blocks
    far = np.setdiff1d(np.arange(pp.nfac), scattered[:512])
    far = far[~np.isin(pp.pt_vid0[far], pp.pt_vid0[scattered[:512]])][:100]
    _, g_short = g.eval_grad(scattered[:512])
    _, g_long = g.eval_grad(np.concatenate([scattered[:512], far]).astype(np.int64))
    pts = np.unique(pp.pt_vid0[scattered[:512]])
    idx = (pts[:, None] + np.arange(3)[None, :]).ravel()
    assert np.array_equal(g_short[idx], g_long[idx])


def test_a_plan_made_before_a_factor_became_exponential_is_refused(gctx):
    """rdis_hip_nlp_set_exponential after rdis_hip_plan_create: the reference's gradient asserts the flag off
    (NonlinearProductFactor.cpp:110), so the stale plan's solve is refused like plan_create and eval_grad refuse (advisor,
    round 5: its value-only trials would have used coeff * exp(-product), its slope trials the plain product)"""
    pp = P.make_high_dim_sinusoid()
    g = capi.Problem(gctx, pp)
    plan = capi.Plan(g)
    plan.set_start(pp.x0)
    plan.solve(2, 3e-8)
    plan.fetch()
    use = np.zeros(pp.nfac, dtype=np.uint8)
    use[3] = 1
    g.set_exponential(use)
    with pytest.raises(capi.RdisHipError):
        plan.solve(2, 3e-8)
    g.set_exponential(None)
    plan.solve(2, 3e-8)
    plan.fetch()
