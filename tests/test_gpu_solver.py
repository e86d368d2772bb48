"""Parity of the device-resident subspace solver (through the C ABI) with the CPU
oracle, i.e. with CGDSubspaceOptimizer::optimize of the reference.

Why several kinds of check: 25 unconverged CG iterations are a chaotic map of
the start point (tests/test_oracle.py::test_cgd_is_chaotic -- a 1e-15 relative
perturbation moves the end value by percents, for the reference itself too), so
"end value equal to 1e-6" is only meaningful where the solve converges.  Hence:
  * distribution: end values over one-ulp-perturbed starts, device against oracle
               (configs 3 and 4); the oracle's unperturbed run IS the reference's
               recorded one, bit for bit.
  * config 5 : at its stated size (1000 components in one launch), every component
               against an INDEPENDENT oracle run; measured: no converged regime exists
               for BA under this algorithm (the oracle differs from itself by 11 %
               after a one-ulp change), so populations are compared.
  * replay   : the device records every value its control logic saw; the oracle,
               fed those values, must ask for bit-identical step lengths, and
               its own evaluations at those points must agree to rounding.
  * prefix   : one line minimisation agrees to Brent's tolerance.
  * converged: minima of converging problems agree tightly (golden values).
  * contract : post-conditions of optimize() (clamping, constants, empty list,
               rollback, counters, independence of batch members).
"""
import numpy as np
import pytest

from oracle import oracle as O
from rdis_amd import capi, problems as P

pytestmark = pytest.mark.gpu


def solve(gctx, pp, maxiters=25, ftol=3e-8, free_vid=None, fac_id=None, x=None, trace=0, opts=None):
    g = capi.Problem(gctx, pp)
    if free_vid is None:
        fv, fc = np.arange(pp.nvars, dtype=np.int64), np.arange(pp.nfac, dtype=np.int64)
    else:
        fv, fc = np.asarray(free_vid, dtype=np.int64), np.asarray(fac_id, dtype=np.int64)
    plan = capi.Plan(g, np.array([0, len(fv)]), fv, np.array([0, len(fc)]), fc)
    if trace:
        plan.set_option("trace_records", trace)
        plan.set_option("dump_iters", maxiters)
    for k, v in (opts or {}).items():
        plan.set_option(k, v)
    plan.set_start(pp.x0[fv] if x is None else x)
    plan.solve(maxiters, ftol)
    r = plan.fetch()
    tr = (plan.get_trace(0, trace)[0], plan.get_vectors(0, maxiters)) if trace else None
    return g, r, tr


def check_replay(pp, trv, r, maxiters, ftol=3e-8, free_vid=None, fac_id=None, x=None, iter_tol=1e-11, far_tol=1e-6, near_scale=1.0, vec_tol=1e-8):
    """The oracle re-runs the solve, (a) fed the scalars the device's control logic saw and
    (b) restarted at every line search from the device's own point and direction, so both
    sides evaluate at bit-identical points.  Then over the WHOLE run:
      - every step length the oracle asks for is bit-identical to the device's (same algorithm),
      - the oracle's own objective / slope / CG reductions agree with the device's to rounding,
      - between re-syncs the oracle's own p, xi stay within 1e-8 of the device's."""
    tr, vd = trv
    x = pp.x0 if x is None and free_vid is None else x
    rep = O.OracleProblem(pp).replay(tr, free_vid=free_vid, fac=fac_id, x=x, maxiters=maxiters, ftol=ftol, vdump=vd)
    print("replay: f_rel %.2e / ordinary %.2e  slope_rel %.2e / ordinary %.2e  in eps x bound: f %.2f slope %.2f" % (
        rep.max_f_rel, rep.max_f_rel_near, rep.max_slope_rel, rep.max_slope_rel_near, rep.max_f_bound, rep.max_slope_bound))
    assert rep.underrun == 0 and rep.tag_mismatches == 0, rep
    assert rep.step_mismatches == 0 and rep.first_mismatch == -1, rep      # bit-identical decisions
    assert rep.consumed == len(tr), rep                                    # and nothing left over
    assert rep.reason == (r.status[0] & 0xFF) and rep.iters == r.iters[0], rep
    assert rep.synced_iters == r.iters[0] + 1, rep
    # objective: 1e-12 of sum|factor values|, slope: 1e-11 of sum|g_j xi_j| at every ordinary trial
    # point (|f| <= 4|f(x0)|+1 and first-order error amplification below 1e5, rdis_oracle.h).  The
    # far-out bracketing steps (f up to 1e4 f(x0)) and steps next to a projection's pole (z is what
    # is left of O(1) terms cancelling to 1e-4..1e-8) are ill-conditioned in the factor arithmetic
    # itself; two correct fp64 evaluations differ there by up to ~1e-7 (observed 1e-11..1e-7
    # depending on where the chaotic trajectory happens to step) and 1e-6 is the bar.
    # rep.max_f_bound / max_slope_bound report every difference in units of eps x a first-order
    # rounding bound of the sum at that point (below 1 at ordinary points, up to ~1e3 far out).
    assert rep.max_f_rel_near <= 1e-12 * near_scale and rep.max_slope_rel_near <= 1e-11 * near_scale, rep
    assert rep.max_f_rel <= far_tol and rep.max_slope_rel <= far_tol, rep
    assert rep.max_iter_rel <= iter_tol, rep    # gg, dgg, gradient test
    assert rep.max_vec_rel <= vec_tol, rep      # one iteration of drift in p / xi (inf-norm relative)
    if not (r.status[0] & capi.STATUS_ROLLED_BACK):
        assert rep.fret == r.fret[0]
    # without the re-sync the decisions are still bit-identical (they only depend on the scalars)
    rep2 = O.OracleProblem(pp).replay(tr, free_vid=free_vid, fac=fac_id, x=x, maxiters=maxiters, ftol=ftol)
    assert rep2.step_mismatches == 0 and rep2.tag_mismatches == 0 and rep2.consumed == len(tr), rep2
    return rep


def check_prefix_endpoints(gctx, pp, ks, free_vid=None, fac_id=None):
    """the value returned after k iterations is the oracle's objective at the returned point
    (1e-12), for every prefix length k: value parity along the device's whole trajectory"""
    fv = np.arange(pp.nvars, dtype=np.int64) if free_vid is None else free_vid
    for k in ks:
        _, r, _ = solve(gctx, pp, maxiters=k, free_vid=free_vid, fac_id=fac_id)
        o = O.OracleProblem(pp, emulate_stale_cache=False)
        o.assign(fv, r.x)
        fo = o.eval(fac_id)
        assert abs(fo - r.fret[0]) <= 1e-12 * abs(fo), (k, fo, r.fret[0])


REPLAY_CASES = {
    "ladybug_5_30": (lambda: P.load_bal(ncams=5, npts=30), 25),
    "ladybug_49_500": (lambda: P.load_bal(ncams=49, npts=500), 8),
    "testpoly": (lambda: _with_x0(P.load_poly(), [1.0, -2.0]), 50),
    "sinusoid": (lambda: _with_x0(P.make_high_dim_sinusoid(), np.random.default_rng(2).uniform(-6, 6, 121)), 25),
    "synthetic_3x40": (lambda: P.make_synthetic_ba(1, 3, 40), 25),
    # BASELINE config 2 the way optSinusoid starts it: uniform over the FULL domain +-62.83
    # (src/optimize_sinusoid.cpp:154-165); committed start vector, tests/golden/sinusoid_start.json
    "sinusoid_full_domain": (lambda: _with_x0(P.make_high_dim_sinusoid(), _sinusoid_start()), 25),
}


def _sinusoid_start():
    import json
    import os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sinusoid_start.json")) as fh:
        return np.array(json.load(fh)["x0"])


def _with_x0(pp, x0):
    pp.x0 = np.asarray(x0, dtype=float)
    return pp


@pytest.mark.parametrize("name", list(REPLAY_CASES))
def test_replay_device_decisions_equal_oracle(name, gctx):
    make, mit = REPLAY_CASES[name]
    pp = make()
    _, r, trv = solve(gctx, pp, maxiters=mit, trace=1 << 14)
    assert len(trv[0]) > 10
    check_replay(pp, trv, r, mit)
    check_prefix_endpoints(gctx, pp, [k for k in (1, 2, 3, 5, 10, 25) if k <= mit])


def test_config2_sinusoid_from_the_full_domain_start(gctx):
    """config 2 from the committed full-domain start: the first line searches take steps far beyond the
    domain (a bound becomes active: the objective is evaluated at the clamped point while CG keeps
    the unclamped iterate, CGDSubspaceOptimizer.cpp:165-168), the solve descends from 17126 by an
    order of magnitude, and the returned point is inside the domain with a variable ON a bound"""
    pp = _with_x0(P.make_high_dim_sinusoid(), _sinusoid_start())
    assert np.max(np.abs(pp.x0)) > 50 and np.all(pp.x0 >= pp.lo) and np.all(pp.x0 <= pp.hi)
    g, r, _ = solve(gctx, pp, maxiters=25)
    o = O.OracleProblem(pp, emulate_stale_cache=False)
    finit = o.eval()
    assert abs((r.fret[0] - r.delta[0]) - finit) <= 1e-12 * abs(finit) and abs(finit - 17126.136253546265) < 1e-6
    assert r.fret[0] < 0.2 * finit and np.all(r.x >= pp.lo) and np.all(r.x <= pp.hi)
    assert np.any((r.x == pp.lo) | (r.x == pp.hi))                     # a bound is active at the end
    o.assign(None, r.x)
    assert abs(o.eval() - r.fret[0]) <= 1e-12 * abs(r.fret[0])


@pytest.mark.parametrize("name", ["ladybug_5_30", "ladybug_49_500", "sinusoid"])
def test_first_line_minimisation_prefix(name, gctx):
    make, _ = REPLAY_CASES[name]
    pp = make()
    _, r, _ = solve(gctx, pp, maxiters=1)
    ro = O.OracleProblem(pp).cgd(maxiters=1)
    # SURVEY 8c proposed 1e-10 (the value is second order in the step at a line minimum).  Measured: two correct
    # roundings of the SAME algorithm -- the oracle against the oracle started one unit in the last place away,
    # or with its second derivative formula -- end one line minimisation 1e-14 .. 1e-7 apart with (nearly)
    # identical evaluation counts: Dbrent places its steps by slopes and compares nearly equal values, and
    # their rounding noise near the minimum moves the last trial points.  The bar is therefore a few times
    # Brent's 3e-8, and the device must be no further from the oracle than the oracle is from itself (x 10).
    rng = np.random.default_rng(7)
    own = max(abs(O.OracleProblem(pp).cgd(x=ulp_perturbed(pp.x0, rng), maxiters=1).fret - ro.fret) / abs(ro.fret) for _ in range(8))
    dev = abs(r.fret[0] - ro.fret) / abs(ro.fret)
    print("%s: one line minimisation, device against oracle %.2e, oracle against itself (8 one-ulp starts, max) %.2e" % (name, dev, own))
    assert dev <= 3e-7 and dev <= 10.0 * max(own, 3e-8), (dev, own)
    assert abs(r.delta[0] - ro.delta) <= 3e-7 * abs(ro.delta)
    assert abs(int(r.nfeval[0]) - ro.nfeval) <= 3 and r.iters[0] == ro.iters == 0


def test_testpoly_converged_minima_match_golden(golden, gctx):
    t = golden["testpoly"]
    pp = P.load_poly()
    for case in t["cgd"]:
        _, r, _ = solve(gctx, pp, maxiters=t["cgd_maxiters"], x=np.array(case["start"]))
        assert abs(r.fret[0] - case["fret"]) <= 1e-10 * abs(case["fret"])        # converged: 1e-10 rel
        assert (r.status[0] & 0xFF) in (0, 1, 2)
        if "x" in case:
            assert np.max(np.abs(r.x - case["x"])) <= 1e-5                      # minimiser known to sqrt(eps)-ish
    # the reference's documented global minimum (data/testpoly.txt:17-22)
    _, r, _ = solve(gctx, pp, maxiters=50, x=np.array([0.0, 0.0]))
    assert abs(r.fret[0] - (-168.2721)) < 1e-4 and np.max(np.abs(r.x - (-4.6601))) < 1e-4


def ulp_perturbed(x0, rng):
    """every variable moved by one unit in the last place, up or down"""
    return np.nextafter(x0, np.where(rng.random(x0.shape) < 0.5, -np.inf, np.inf))


def oracle_end_values(pp, starts, maxiters, threads=8):
    """independent oracle solves (reference-faithful rounding), in parallel: ctypes releases the GIL"""
    from concurrent.futures import ThreadPoolExecutor

    def run(x):
        return O.OracleProblem(pp).cgd(x=x, maxiters=maxiters, ftol=3e-8).fret
    with ThreadPoolExecutor(threads) as ex:
        return np.array(list(ex.map(run, starts)))


def device_end_values(gctx, pp, starts, maxiters, opts=None):
    g = capi.Problem(gctx, pp)
    fv, fc = np.arange(pp.nvars, dtype=np.int64), np.arange(pp.nfac, dtype=np.int64)
    plan = capi.Plan(g, np.array([0, len(fv)]), fv, np.array([0, len(fc)]), fc)
    for k, v in (opts or {}).items():
        plan.set_option(k, v)
    out = []
    for x in starts:
        plan.set_start(x)
        plan.solve(maxiters, 3e-8)
        r = plan.fetch()
        assert (r.status[0] & 0xFF) in (0, 1, 3)      # the iteration limit as a rule; now and then the 3e-8 test fires first
        out.append(r.fret[0])
    return np.array(out)


def _end_value_fixture():
    import json
    import os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "end_values.json")) as fh:
        return json.load(fh)


def _ks2(a, b, c_alpha=1.358):
    """two-sample Kolmogorov-Smirnov statistic and its critical value c(alpha) sqrt((n + m) / (n m)); c(0.05) = 1.358"""
    a, b = np.sort(a), np.sort(b)
    allv = np.concatenate([a, b])
    d = np.max(np.abs(np.searchsorted(a, allv, side="right") / len(a) - np.searchsorted(b, allv, side="right") / len(b)))
    return d, c_alpha * np.sqrt((len(a) + len(b)) / (len(a) * len(b)))


@pytest.mark.parametrize("key", ["ladybug_5_30", "ladybug_full"])
def test_end_values_distribution_matches_oracle(key, golden, gctx):
    """BASELINE configs 3 and 4.  25 unconverged CG iterations are a chaotic map of the start (a one-ulp
    change of x0 moves the end value by percents, for the reference itself too), so the end value of ONE
    run cannot be compared to 1e-6 between implementations that round differently.  What can: the
    DISTRIBUTION of end values over one-ulp-perturbed starts -- and that distribution is a property of the
    evaluator's ROUNDING, not of the algorithm alone.  The committed fixture (tests/golden/end_values.json,
    tests/golden/make_end_values.py) holds two samples of the reference-faithful oracle over the same 320 / 512
    starts: compiled like the reference (no fused multiply-add: its unperturbed entry IS the reference's
    recorded end value) and compiled with contraction -- the same algorithm, an equally valid rounding.  On full
    ladybug the two part with KS 0.21 (p = 1e-5).  Two things move the device's population, both found by
    changing one thing at a time (round 4: in the oracle; round 5: on the device, profiles/r05_*_population*.txt):
      * fused multiply-adds in the factor arithmetic (full ladybug: device median 89130 with them, 87820 without,
        oracle 87978);
      * the ASSOCIATION of a trial's slope (ladybug 5 / 30): the solvers add factor by factor, sum_f (sum_k partial_fk
        xi_k), the reference forms the gradient and then gradient times direction, sum_v (sum_f partial_fv) xi_v
        (Df1dim::df).  Last-place differences -- but Dbrent takes secant steps between trial points 1e-17 apart, the
        difference of the two slopes cancels ten digits, and the association moves every such step the same way: the
        first line minimisation ends 6e-9 lower on 88 % of the starts, and after 25 iterations the lower quartile
        sits 0.17 % higher (KS 0.15 at n = 512).  The ORACLE with the device's association (ro_set_experiment(2))
        draws the device's population (quartiles 25.152 / 25.219 / 25.412 against the device's 25.156 / 25.222 / 25.427);
        sincos, reciprocals, forward-mode slope, contraction: none of them does (KS 0.14 .. 0.16 each).
    Plan option factor_rounding = 1 runs the solvers these configs reach with the reference's rounding AND its slope: the
    gradient at the trial point, every variable's partials in factor-list order, then gradient times direction over the
    variables in list order -- on full ladybug one sequential sum of 23 769 terms by one wave (solver_coop.hpp:
    slope_reference; 0.18 s a solve instead of 2.5 ms: a parity option).  Asserted for it, with as many device draws as the
    fixture has oracle draws, from OTHER one-ulp starts, at alpha = 0.05:
      * the PLAIN two-sample test, KS(device, oracle) <= the critical value, on both configs (measured: 0.045 on 5 / 30,
        p = 0.7; 0.056 on full ladybug, p = 0.7 -- round 5; before the slope's order was the reference's: 0.125 .. 0.133);
      * every quartile within 1.5 % of the oracle's, the median between the 25 % and 75 % quantiles of both oracle samples,
        the reference's recorded value inside the device's range, the sample spread (chaos on the device's side too).
    The default arithmetic is a different population by the causes named above.  On 5 / 30 (fused multiply-adds and the
    factor-by-factor slope in the batch solvers): the locations and a looser bound that covers the 0.15 measured.  On full
    ladybug (the pipelined cooperative solver: the reference's rounding, a parallel reduction's slope): the device must be
    no further from the reference-faithful oracle than the oracle's own variants -- slope by factor in list order, slope by
    factor as a tree: 0.10 .. 0.19 from one another, the last two differing in nothing but the ORDER of one sum -- are from
    one another (the contracted oracle is not one of them: this path does not contract).  Since round 6 the parity option is
    also compared END TO END with ==, no populations: tests/test_gpu_parity.py."""
    c = golden["cgd"][key]
    fx = _end_value_fixture()
    oe, oc = np.array(fx[key]["end_values"]), np.array(fx[key]["end_values_contracted"])
    assert oe[0] == c["fret"] and len(oe) >= 256 and len(oc) == len(oe)   # the unperturbed oracle run IS the reference's
    pp = P.load_bal(ncams=c["ncams"], npts=c["npts"])
    n = len(oe)
    dstarts = [ulp_perturbed(pp.x0, np.random.default_rng([fx["seed"], 100000 + k])) for k in range(n)]
    q = lambda v: np.quantile(v, [0.25, 0.5, 0.75])
    from scipy import stats
    (d_self, crit_full) = _ks2(oc, oe)
    for name, opts, slack in (("reference rounding", {"factor_rounding": 1}, 0.0), ("default", {}, 0.07)):
        crit = crit_full
        # (the parity option on full ladybug is 0.3 s a solve and, since round 6, pinned with == in tests/test_gpu_parity.py: 96 draws
        # of it here -- the two-sample test's critical value grows accordingly --, every other sample in full)
        few = slack == 0.0 and key == "ladybug_full"
        de = device_end_values(gctx, pp, dstarts[:96] if few else dstarts, c["maxiters"], opts)
        (d_o, crit_o), (d_c, _) = _ks2(de, oe), _ks2(de, oc)
        if few:
            crit = crit_o
        print("%s, %s: end values after %d iterations (n = %d each): oracle quartiles %s, oracle contracted %s, device %s; KS device-oracle %.3f, "
              "device-contracted %.3f, oracle-contracted %.3f (critical at 0.05: %.3f); Mann-Whitney p device-oracle %.3f; "
              "reference %.6g, device range [%.6g, %.6g]" % (key, name, c["maxiters"], n, q(oe), q(oc), q(de), d_o, d_c, d_self, crit,
                                                             stats.mannwhitneyu(de, oe).pvalue, c["fret"], de.min(), de.max()))
        assert de.max() - de.min() > 1e-4 * de.min()                  # chaos on the device's side too
        if slack == 0.0:
            # the plain two-sample test against the reference-faithful oracle: under this option the solvers these configs
            # reach (LDS-resident; cooperative, plain layout) form a trial's slope the reference's way -- gradient times
            # direction, every sum in the reference's order (solver_lds.hpp / solver_coop.hpp: slope_reference)
            assert d_o <= crit, (name, key, d_o, crit)
        elif key == "ladybug_5_30":
            assert min(d_o, d_c) <= d_self + crit + slack, (name, d_o, d_c, d_self, crit)
        else:
            # Full ladybug by default: the pipelined cooperative solver rounds like the reference and adds a trial's slope
            # factor by factor, as a tree over lanes and workgroups.  The ORACLE's population moves with exactly that
            # (tests/golden/make_end_values_slope.py): the same association in list order is 0.10 from the reference's, as a
            # tree 0.18, list order against tree 0.19 -- two samples that differ in nothing but the ORDER of one sum.
            # Asserted: the device is no further from the reference-faithful oracle than the oracle's own variants are from
            # one another (no allowance on top), and its median sits inside theirs.
            # (The contracted oracle is NOT in this family: the default cooperative path rounds like the reference, so only the
            # variants that differ from the reference-faithful sample in the association / order of the slope's sum are.)
            fam = [oe, np.array(fx[key]["end_values_slope_by_factor"]), np.array(fx[key]["end_values_slope_by_factor_tree"])]
            spread = max(_ks2(a, b)[0] for i, a in enumerate(fam) for b in fam[i + 1:])
            d_fam = [_ks2(de, m)[0] for m in fam]
            print("%s, %s: KS against the oracle family (reference, slope by factor, ... as a tree) %s; the family's own spread %.3f" % (
                key, name, np.round(d_fam, 3), spread))
            assert d_o <= spread, (name, key, d_o, spread)
            assert min(q(m)[1] for m in fam) * (1 - 2e-3) <= q(de)[1] <= max(q(m)[1] for m in fam) * (1 + 2e-3), (name, [q(m)[1] for m in fam], q(de)[1])
        assert np.all(np.abs(q(de) - q(oe)) <= (0.03 if few else 0.015) * q(oe)), (name, q(de), q(oe))
        lo, hi = min(q(oe)[0], q(oc)[0]), max(q(oe)[2], q(oc)[2])
        assert lo <= q(de)[1] <= hi, (name, q(oe), q(oc), q(de))
        assert few or de.min() <= c["fret"] <= de.max(), (name, c["fret"], de.min(), de.max())


def test_prefix_values_part_from_the_oracle_at_the_rate_of_chaos(gctx):
    """The curve that makes "chaos, not a bug" falsifiable on the device itself: |f_k(device) - f_k(oracle)| / f_k
    for k = 1 .. 25 CG iterations from the SAME start, next to the oracle's curves against ITSELF (committed
    fixture, tests/golden/make_end_values.py): from eight starts one unit in the last place away -- there the
    VALUES f differ in the last place, as they do between oracle and device -- and with only its derivative
    formula exchanged (values bit-identical, slopes 2e-14 apart: a calm curve, printed for contrast).
    Measured: one line minimisation is reproducible to 1e-9 .. 1e-7 between two correct roundings, not to the
    1e-14 a second-order argument suggests (Dbrent places its steps by slopes and compares nearly equal
    values); on the full problem the oracle is 4e-5 .. 5e-3 from itself after TWO iterations and saturated at
    1e-2 after three (device: 4e-4, 4e-2).  Asserted: at every k the device's envelope (running maximum) is
    within a factor 10 of the oracle family's -- the device is as far from the oracle as the oracle is from
    itself; an error in an evaluation would show at k = 1 --, and the curves do saturate (so the distribution
    test is the right comparison at k = 25)."""
    fx = _end_value_fixture()
    for key, kw in (("ladybug_5_30", dict(ncams=5, npts=30)), ("ladybug_full", {})):
        pp = P.load_bal(**kw)
        of = np.array(fx[key]["prefix_values_from_x0"])
        calm = np.abs(of - np.array(fx[key]["prefix_values_from_x0_adjoint_derivative"])) / np.abs(of)
        own = np.max(np.abs(of - np.array(fx[key]["prefix_values_from_ulp_starts_1_to_8"])) / np.abs(of), axis=0)
        g = capi.Problem(gctx, pp)
        fv, fc = np.arange(pp.nvars, dtype=np.int64), np.arange(pp.nfac, dtype=np.int64)
        plan = capi.Plan(g, np.array([0, len(fv)]), fv, np.array([0, len(fc)]), fc)
        rel = []
        for k in range(1, 26):
            plan.set_start(pp.x0)
            plan.solve(k, 3e-8)
            rel.append(abs(plan.fetch().fret[0] - of[k - 1]) / abs(of[k - 1]))
        plan.close()
        rel = np.array(rel)
        env, env_own = np.maximum.accumulate(rel), np.maximum.accumulate(own)
        print(key, "device against oracle, k = 1..25:           ", " ".join("%.1e" % v for v in rel))
        print(key, "oracle against itself (8 one-ulp starts, max):", " ".join("%.1e" % v for v in own))
        print(key, "oracle against itself (derivative exchanged): ", " ".join("%.1e" % v for v in calm))
        assert rel[0] <= 3e-7, rel[0]
        for k in range(25):
            assert env[k] <= 10.0 * max(env_own[k], 3e-8), (key, k + 1, env[k], env_own[k])
        assert env[-1] >= 1e-6 and env_own[-1] >= 1e-6      # the trajectories do separate: chaos, on both sides


def test_ladybug_full_objective(golden, gctx):
    """BASELINE config 4: CGD over all 23769 variables / 31843 factors, SSmaxit 25"""
    c = golden["cgd"]["ladybug_full"]
    pp = P.load_bal()
    g, r, tr = solve(gctx, pp, maxiters=25, trace=1 << 13)
    assert (r.status[0] & 0xFF) == 3 and r.iters[0] == 24
    assert abs((r.fret[0] - r.delta[0]) - 850912.46068083902) <= 1e-12 * 850912.46068083902   # initialFval (golden)
    # (end value: test_end_values_distribution_matches_oracle; evaluation counts vary with the trajectory)
    assert abs(int(r.nfeval[0]) - c["nfeval"]) <= 0.15 * c["nfeval"]
    # the returned value IS the objective at the returned point (size-independent check)
    assert np.array_equal(g.get_x(), r.x)                                       # variables left assigned
    assert abs(g.eval() - r.fret[0]) <= 1e-12 * r.fret[0]
    assert np.all(r.x >= pp.lo) and np.all(r.x <= pp.hi)
    # the whole run replays against the oracle: bit-identical decisions, values to rounding
    check_replay(pp, tr, r, 25)
    ro = O.OracleProblem(pp, emulate_stale_cache=False)
    ro.assign(None, r.x)
    assert abs(ro.eval() - r.fret[0]) <= 1e-12 * r.fret[0]                      # oracle's objective at the device's point


def test_reference_slope_option_replays_on_full_ladybug(gctx):
    """factor_rounding = 1 on BASELINE config 4: the plain cooperative layout with a trial's slope formed the reference's way
    (solver_coop.hpp: slope_reference -- the gradient at the trial point, variable sums in factor-list order, gradient times
    direction over the variables in list order by one wave).  Five iterations replay against the oracle like every other
    solver's, the value is the objective at the returned point, and two runs agree bit for bit."""
    pp = P.load_bal()
    opts = {"factor_rounding": 1}
    g, r, tr = solve(gctx, pp, maxiters=5, trace=1 << 12, opts=opts)
    assert (r.status[0] & 0xFF) == 3 and r.iters[0] == 4
    assert abs(g.eval() - r.fret[0]) <= 1e-12 * r.fret[0]
    check_replay(pp, tr, r, 5)
    g2, r2, _ = solve(gctx, pp, maxiters=5, opts=opts)
    assert r2.fret[0] == r.fret[0] and np.array_equal(r2.x, r.x)
    # ... and it is another trajectory than the default's from the first line minimisation on (the slopes differ in the last place)
    g3, r3, _ = solve(gctx, pp, maxiters=5)
    assert r3.fret[0] != r.fret[0] and abs(r3.fret[0] - r.fret[0]) <= 0.05 * r.fret[0]


def _ks(a, b):
    """two-sample Kolmogorov-Smirnov statistic"""
    allv = np.sort(np.concatenate([a, b]))
    return float(np.max(np.abs(np.searchsorted(np.sort(a), allv, side="right") / len(a) -
                               np.searchsorted(np.sort(b), allv, side="right") / len(b))))


def test_config5_synthetic_1000_components(gctx):
    """BASELINE config 5 at its stated size: the 1000-component synthetic decomposition (3 cameras x 40
    points each: 147 variables, 120 factors) in ONE launch, every component compared with an
    INDEPENDENT oracle run of CGDSubspaceOptimizer::optimize (own trajectory, reference-faithful
    rounding) -- not a replay.

    What "the same result" can mean here was measured, not assumed: most components leave by the
    function tolerance (2|df| <= 3e-8 (|f| + |fp|)) after 10-25 iterations, but that test fires on a
    plateau of the CG descent, not at a minimum -- the ORACLE ITSELF, restarted from a start moved by
    one unit in the last place, ends 11 % away (median over components; a quarter of them 23 % and
    more).  There is no converged regime for bundle adjustment under this algorithm, with the
    reference's own arithmetic; per-component end values are draws from a distribution.  So the test
    compares distributions, with the oracle-against-perturbed-oracle spread as the yardstick:
      * paired differences device / oracle are no larger than oracle / perturbed oracle,
      * the two populations of 1000 end values are indistinguishable (Kolmogorov-Smirnov, quantiles),
      * the top-level objective (the sum the all-reduce carries) agrees within the spread of such sums,
      * exit reasons occur at the same rates.
    The per-evaluation parity of these very solves is pinned bit for bit by the replay tests
    (synthetic_3x40, test_batch_replay_of_members)."""
    pp = P.make_synthetic_ba(1000, 3, 40)
    g = capi.Problem(gctx, pp)
    plan = capi.Plan(g)
    plan.set_start(pp.x0)
    plan.solve(25, 3e-8)
    r = plan.fetch()
    ms, _ = plan.last_kernel_ms()
    from concurrent.futures import ThreadPoolExecutor
    rng = np.random.default_rng(5)
    x1 = ulp_perturbed(pp.x0, rng)

    def run(args):
        c, x, stale = args
        fv, fc = pp.component(c)
        return O.OracleProblem(pp, emulate_stale_cache=stale).cgd(free_vid=fv, fac=fc, x=x[fv], maxiters=25, ftol=3e-8)
    # The comparator recomputes every factor at every trial point, like the device.  The reference keeps
    # a factor's cached value when none of its variables moved by 1e-12 (Variable.cpp:70-76), a quirk the
    # device deliberately does not emulate: it changes values at the 1e-12 level, but on
    # these plateaus that shifts how often the 3e-8 test fires -- 859 tolerance exits of 1000 with the
    # quirk, 939 without (oracle against oracle) -- so the reference-faithful population is reported
    # and bounded separately below.
    with ThreadPoolExecutor(8) as ex:
        ro = list(ex.map(run, [(c, pp.x0, False) for c in range(1000)]))
        rp = list(ex.map(run, [(c, x1, False) for c in range(1000)]))
        rq = list(ex.map(run, [(c, pp.x0, True) for c in range(1000)]))
    of, pf = np.array([q.fret for q in ro]), np.array([q.fret for q in rp])
    ost, pst, dst = np.array([q.status & 0xFF for q in ro]), np.array([q.status & 0xFF for q in rp]), r.status & 0xFF
    finit = np.array([q.finit for q in ro])
    assert np.max(np.abs((r.fret - r.delta) - finit) / finit) <= 1e-12            # same initial values
    d_do = np.abs(r.fret - of) / np.abs(of)          # device against oracle
    d_oo = np.abs(pf - of) / np.abs(of)              # oracle against itself, start moved by one ulp
    qs = [0.1, 0.25, 0.5, 0.75, 0.9]
    qd, qo, qp = np.quantile(r.fret / finit, qs), np.quantile(of / finit, qs), np.quantile(pf / finit, qs)
    ks_do, ks_oo = _ks(r.fret / finit, of / finit), _ks(pf / finit, of / finit)
    print("config 5 (1000 components, %.2f ms on the device): paired |df|/f median / 75%% -- device:oracle %.3f / %.3f, oracle:oracle' %.3f / %.3f; "
          "KS device:oracle %.3f oracle:oracle' %.3f; f_end/f_0 quantiles 10..90%% device %s oracle %s; objective device %.6g oracle %.6g oracle' %.6g; "
          "tolerance exits device %d oracle %d oracle' %d" % (
              ms, np.median(d_do), np.quantile(d_do, 0.75), np.median(d_oo), np.quantile(d_oo, 0.75), ks_do, ks_oo,
              np.array2string(qd, precision=5), np.array2string(qo, precision=5), r.fret.sum(), of.sum(), pf.sum(),
              (dst <= 2).sum(), (ost <= 2).sum(), (pst <= 2).sum()))
    assert np.median(d_oo) > 1e-3                                   # the premise: the oracle does not agree with itself
    assert np.median(d_do) <= 1.5 * np.median(d_oo) and np.quantile(d_do, 0.75) <= 1.5 * np.quantile(d_oo, 0.75)
    assert ks_do <= 0.0872                                          # 1.95 sqrt(2 / 1000): alpha = 0.001
    assert np.all(np.abs(qd - qo) <= 0.1 * qo + 3 * np.abs(qp - qo))
    spread = abs(pf.sum() - of.sum())
    assert abs(r.fret.sum() - of.sum()) <= max(3 * spread, 0.03 * of.sum())
    assert abs(int((dst <= 2).sum()) - int((ost <= 2).sum())) <= 35 and np.all((dst != 5) & (dst != 7))   # 3 sigma of a binomial count
    # against the reference-faithful population (stale factor cache emulated): same distribution up to
    # the documented shift of the exit rate
    qf, qfst = np.array([q.fret for q in rq]), np.array([q.status & 0xFF for q in rq])
    print("   reference-faithful oracle (stale factor cache emulated): tolerance exits %d, objective %.6g, KS device:reference %.3f" % (
        (qfst <= 2).sum(), qf.sum(), _ks(r.fret / finit, qf / finit)))
    assert _ks(r.fret / finit, qf / finit) <= 0.0872 and abs(r.fret.sum() - qf.sum()) <= 0.05 * qf.sum()
    assert np.all(r.delta <= 0)
    o = O.OracleProblem(pp, emulate_stale_cache=False)
    o.assign(None, r.x)
    assert abs(o.eval() - r.fret.sum()) <= 1e-12 * r.fret.sum()          # returned values ARE the objective at the returned point
    # The device with the quirk EMULATED (plan option emulate_stale_cache: a factor keeps its cached value until one of
    # its variables is assigned a value 1e-12 or more away from its previous one, src/Variable.cpp:66-76, src/Factor.h:228-234;
    # every point the reference evaluates is evaluated): now the reference-faithful population is the comparator --
    # exit rates included, which the recomputing device misses by 76 of 1000.
    g.set_x(pp.x0)
    plan_e = capi.Plan(g)
    plan_e.set_option("emulate_stale_cache", 1)
    plan_e.set_start(pp.x0)
    plan_e.solve(25, 3e-8)
    re_ = plan_e.fetch()
    ms_e, _ = plan_e.last_kernel_ms()
    est = re_.status & 0xFF
    d_eq = np.abs(re_.fret - qf) / np.abs(qf)
    print("   device with the cache emulated (%.2f ms): tolerance exits %d (reference-faithful oracle %d, recomputing device %d), objective %.6g (%.6g), "
          "KS device:reference %.3f, paired |df|/f median %.3f, evaluations %d (oracle %d)" % (
              ms_e, (est <= 2).sum(), (qfst <= 2).sum(), (dst <= 2).sum(), re_.fret.sum(), qf.sum(), _ks(re_.fret / finit, qf / finit),
              np.median(d_eq), re_.nfeval.sum(), sum(q.nfeval for q in rq)))
    assert abs(int((est <= 2).sum()) - int((qfst <= 2).sum())) <= 35 and np.all((est != 5) & (est != 7))    # 3 sigma of a binomial count
    assert _ks(re_.fret / finit, qf / finit) <= 0.0872
    assert np.median(d_eq) <= 1.5 * np.median(d_oo) and np.quantile(d_eq, 0.75) <= 1.5 * np.quantile(d_oo, 0.75)
    assert abs(re_.fret.sum() - qf.sum()) <= max(3 * spread, 0.03 * qf.sum())
    assert abs(int(re_.nfeval.sum()) - sum(q.nfeval for q in rq)) <= 0.03 * re_.nfeval.sum()
    assert np.max(np.abs((re_.fret - re_.delta) - finit) / finit) <= 1e-12 and np.all(re_.delta <= 0)
    # short prefixes are not yet chaotic: after two iterations device and emulating oracle agree component by component
    g.set_x(pp.x0)
    plan_e.set_start(pp.x0)
    plan_e.solve(2, 3e-8)
    r2 = plan_e.fetch()
    def run2(c):
        fv, fc = pp.component(c)
        return O.OracleProblem(pp, emulate_stale_cache=True).cgd(free_vid=fv, fac=fc, x=pp.x0[fv], maxiters=2, ftol=3e-8)
    with ThreadPoolExecutor(8) as ex:
        o2 = list(ex.map(run2, range(1000)))
    f2 = np.array([q.fret for q in o2])
    # (the number of Brent steps is already a matter of last bits -- near a line minimum the comparisons of trial values
    # that differ by rounding decide it: a third of the components make exactly the oracle's calls -- the values are not)
    same_counts = np.mean([(q.nfeval == a and q.ngeval == b) for q, a, b in zip(o2, r2.nfeval, r2.ngeval)])
    dcalls = np.array([abs(int(q.nfeval) - int(a)) for q, a in zip(o2, r2.nfeval)])
    rel2 = np.abs(r2.fret - f2) / np.abs(f2)
    print("   two iterations, cache emulated: identical f / df call counts in %.1f %% of the components (median difference %d calls), |df|/f median %.2e, 99 %% %.2e" % (
        100 * same_counts, np.median(dcalls), np.median(rel2), np.quantile(rel2, 0.99)))
    assert np.median(rel2) <= 1e-10 and np.quantile(rel2, 0.99) <= 1e-6 and np.median(dcalls) <= 4
    plan_e.close()
    # the option is refused where it is not implemented (a cooperative group)
    lb = P.load_bal(ncams=5, npts=30).single_component()
    gl = capi.Problem(gctx, lb)
    pl = capi.Plan(gl)
    pl.set_option("emulate_stale_cache", 1)
    pl.set_option("lds_resident", 0)
    pl.set_start(lb.x0)
    with pytest.raises(capi.RdisHipError):
        pl.solve(3, 3e-8)
    pl.close(); gl.close()


def test_plans_survive_a_move_of_the_exchange_state(gctx):
    """a persistent plan with ONE cooperative group, then a call whose 49 groups make the problem's
    exchange-state buffer grow (and move), then the first plan again: same bits, no timeout"""
    pp = P.load_bal()
    g = capi.Problem(gctx, pp)
    fv, fc = np.arange(pp.nvars, dtype=np.int64), np.arange(pp.nfac, dtype=np.int64)
    a = capi.Plan(g, np.array([0, len(fv)]), fv, np.array([0, len(fc)]), fc)
    a.set_start(pp.x0)
    a.solve(3, 3e-8)
    r1 = a.fetch()
    cams, _pts = P.ba_alternation_plans(pp)                     # 49 camera components: one group each
    x = pp.x0[cams[1]]
    rb = g.cgd_batch(cams[0], cams[1], cams[2], cams[3], x, 3, 3e-8)
    assert np.all((rb.status & 0xFF) != 7)
    g.set_x(pp.x0)
    a.solve(3, 3e-8)
    r2 = a.fetch()
    assert (r2.status[0] & 0xFF) != 7
    assert r2.fret[0] == r1.fret[0] and np.array_equal(r2.x, r1.x) and r2.nfeval[0] == r1.nfeval[0]


def test_start_points_of_resident_plans_are_taken_when_the_call_returns(gctx):
    """rdis_hip_plan_set_start on a resident plan copies the caller's values through a pinned staging buffer of
    the problem and does not wait for the stream: the caller may overwrite its array the moment the call returns,
    and the start points of two plans set back to back (one staging buffer, reused) do not mix"""
    pp = P.load_bal(ncams=5, npts=30)
    g = capi.Problem(gctx, pp)
    cams, pts = P.ba_alternation_plans(pp)
    pa, pb = capi.Plan(g, *cams), capi.Plan(g, *pts)
    xa, xb = np.ascontiguousarray(pp.x0[cams[1]]), np.ascontiguousarray(pp.x0[pts[1]])

    def solve(plan, x):
        g.set_x(pp.x0)
        plan.set_start(x)
        plan.solve(5, 3e-8)
        return plan.fetch()
    ra, rb = solve(pa, xa), solve(pb, xb)
    # the same starts from scratch arrays that are overwritten as soon as set_start returns, both plans' starts
    # set before either is solved (the second copy follows the first through the same staging buffer)
    g.set_x(pp.x0)
    ta, tb = xa.copy(), xb.copy()
    pa.set_start(ta); ta[:] = np.nan
    pb.set_start(tb); tb[:] = np.nan
    pa.solve(5, 3e-8)
    r1 = pa.fetch()
    g.set_x(pp.x0)
    pb.solve(5, 3e-8)
    r2 = pb.fetch()
    assert np.array_equal(r1.fret, ra.fret) and np.array_equal(r1.x, ra.x)
    assert np.array_equal(r2.fret, rb.fret) and np.array_equal(r2.x, rb.x)


def test_rdis_separator_block_on_full_ladybug(gctx):
    """the call RDIS spends its time in on ladybug: the separator block -- 46 cameras and
    one point, 417 free variables -- against every factor of those cameras (30285), all other points
    constants.  Cooperative kernel with mostly-constant factor slots; replayed by the oracle."""
    pp = P.load_bal()
    free = np.concatenate([np.arange(9 * 46), np.arange(441, 444)]).astype(np.int64)
    fac = np.where((pp.cam_vid0 // 9 < 46) | (pp.pt_vid0 == 441))[0].astype(np.int64)
    g, r, tr = solve(gctx, pp, maxiters=25, free_vid=free, fac_id=fac, trace=1 << 13)
    assert (r.status[0] & 0xFF) == 3 and r.delta[0] < 0
    o = O.OracleProblem(pp)
    assert abs((r.fret[0] - r.delta[0]) - o.eval(fac)) <= 1e-12 * o.eval(fac)      # f(x_init) over the listed factors
    x_all = g.get_x()
    assert np.array_equal(x_all[free], r.x)                                         # block left assigned
    rest = np.setdiff1d(np.arange(pp.nvars), free)
    assert np.array_equal(x_all[rest], pp.x0[rest])                                 # constants untouched
    o.assign(free, r.x)
    assert abs(o.eval(fac) - r.fret[0]) <= 1e-12 * r.fret[0]
    check_replay(pp, tr, r, 25, free_vid=free, fac_id=fac, x=pp.x0[free])


def test_subfunction_with_constants_and_clamping(gctx):
    # only camera 0 and point 0 free; the other variables are constants of the listed factors
    pp = P.load_bal(ncams=5, npts=30)
    free = np.concatenate([np.arange(0, 9), np.arange(45, 48)]).astype(np.int64)
    touching = np.where((pp.cam_vid0 == 0) | (pp.pt_vid0 == 45))[0].astype(np.int64)
    g, r, trv = solve(gctx, pp, maxiters=10, free_vid=free, fac_id=touching, trace=4096)
    check_replay(pp, trv, r, 10, free_vid=free, fac_id=touching, x=pp.x0[free])
    after = g.get_x()
    mask = np.ones(pp.nvars, bool)
    mask[free] = False
    assert np.array_equal(after[mask], pp.x0[mask])                 # constants untouched
    assert np.array_equal(after[free], r.x) and r.delta[0] < 0
    # a start outside the domain is clamped before the first evaluation (quickAssignVals)
    q = P.load_poly()
    _, r, _ = solve(gctx, q, maxiters=50, x=np.array([100.0, -100.0]))
    o = O.OracleProblem(q)
    o.assign(None, np.array([8.0, -9.0]))
    assert abs((r.fret[0] - r.delta[0]) - o.eval()) <= 1e-12 * abs(o.eval())
    assert np.all(r.x >= q.lo) and np.all(r.x <= q.hi)
    # a domain that cuts off the minimum: the solver ends on the boundary like the oracle
    q2 = P.load_poly()
    q2.lo[:] = -3.0
    _, r2, _ = solve(gctx, q2, maxiters=50, x=np.array([0.0, 0.0]))
    ro = O.OracleProblem(q2).cgd(x=np.array([0.0, 0.0]), maxiters=50)
    assert np.max(np.abs(r2.x - ro.x)) <= 1e-6 and abs(r2.fret[0] - ro.fret) <= 1e-9 * abs(ro.fret)
    assert np.min(r2.x) == -3.0


def test_empty_factor_list_and_nan(gctx):
    pp = P.load_poly()
    pp.x0 = np.array([3.0, 4.0])
    g = capi.Problem(gctx, pp)
    r = g.cgd_batch(np.array([0, 2]), np.array([0, 1]), np.array([0, 0]), np.zeros(0, np.int64),
                    np.array([30.0, 40.0]), 50, 3e-8)
    # returns 0, delta 0, nothing touched (CGDSubspaceOptimizer.cpp:26-29)
    assert (r.fret[0], r.delta[0], r.status[0], r.nfeval[0]) == (0.0, 0.0, 6, 0)
    assert list(r.x) == [30.0, 40.0] and list(g.get_x()) == [3.0, 4.0]
    # NaN objective (sqrt of a negative number): reported, start restored
    terms = [(1.0, [(0, 0.5, 0.0, 0)]), (1.0, [(0, 2.0, 0.0, 0)])]
    q = P._pack_nlp(terms, np.array([-1.0]), np.array([-5.0]), np.array([5.0]), {})
    gq = capi.Problem(gctx, q)
    r = gq.cgd_batch(np.array([0, 1]), np.array([0]), np.array([0, 2]), np.array([0, 1]), np.array([-1.0]), 10, 3e-8)
    assert (r.status[0] & 0xFF) == 5 and r.rolled_back[0] and list(r.x) == [-1.0]


def test_batch_members_are_independent_and_deterministic(gctx):
    pp = P.make_synthetic_ba(64, 3, 40)
    g = capi.Problem(gctx, pp)
    plan = capi.Plan(g)
    plan.set_start(pp.x0)
    plan.solve(25, 3e-8)
    a = plan.fetch()
    g.set_x(pp.x0)
    plan.solve(25, 3e-8)
    b = plan.fetch()
    for k in ("x", "fret", "delta", "iters", "status", "nfeval", "ngeval"):
        assert np.array_equal(getattr(a, k), getattr(b, k)), k                  # bit-reproducible
    assert np.all(a.delta <= 0) and np.all((a.status & 0xFF) != 5)
    # each member alone gives bit-identical results to the same member inside the batch
    for c in (0, 17, 63):
        fv, fc = pp.component(c)
        g2 = capi.Problem(gctx, pp)
        r = g2.cgd_batch(np.array([0, len(fv)]), fv, np.array([0, len(fc)]), fc, pp.x0[fv], 25, 3e-8)
        assert r.fret[0] == a.fret[c] and np.array_equal(r.x, a.x[pp.comp_free_ptr[c]:pp.comp_free_ptr[c + 1]])
        assert r.iters[0] == a.iters[c] and r.nfeval[0] == a.nfeval[c]
    # the batch objective (what the RCCL all-reduce carries) is the sum of the members
    assert abs(plan.objective() - np.sum(a.fret)) <= 1e-12 * np.sum(a.fret)


def test_batch_replay_of_members(gctx):
    pp = P.make_synthetic_ba(8, 3, 40)
    g = capi.Problem(gctx, pp)
    plan = capi.Plan(g)
    plan.set_option("trace_records", 8192)
    plan.set_option("dump_iters", 25)
    plan.set_start(pp.x0)
    plan.solve(25, 3e-8)
    r = plan.fetch()
    for c in range(8):
        fv, fc = pp.component(c)
        tr, n = plan.get_trace(c, 8192)
        assert n == len(tr)
        rep = O.OracleProblem(pp).replay(tr, free_vid=fv, fac=fc, x=pp.x0[fv], maxiters=25,
                                         vdump=plan.get_vectors(c, 25)[:int(r.iters[c]) + 1])
        assert rep.step_mismatches == 0 and rep.tag_mismatches == 0 and rep.underrun == 0, (c, rep)
        assert rep.consumed == n and rep.synced_iters == r.iters[c] + 1, (c, rep)
        assert rep.max_f_rel_near <= 1e-12 and rep.max_slope_rel_near <= 1e-11 and rep.max_iter_rel <= 1e-11, (c, rep)
        assert rep.max_f_rel <= 1e-6 and rep.max_slope_rel <= 1e-6, (c, rep)
        assert rep.fret == r.fret[c] and rep.iters == r.iters[c] and rep.reason == (r.status[c] & 0xFF)


def test_plan_rejects_dependent_components(gctx):
    pp = P.load_bal(ncams=5, npts=30)
    g = capi.Problem(gctx, pp)
    allf = np.arange(pp.nfac, dtype=np.int64)
    with pytest.raises(capi.RdisHipError) as e:          # a variable free in two components
        capi.Plan(g, np.array([0, 9, 18]), np.concatenate([np.arange(9), np.arange(9)]), np.array([0, 1, 2]), allf[:2])
    assert e.value.code == -4
    with pytest.raises(capi.RdisHipError) as e:          # factor 0 reads camera 0, which is free in component 1
        capi.Plan(g, np.array([0, 3, 12]), np.concatenate([np.arange(45, 48), np.arange(9)]),
                  np.array([0, 1, 2]), np.array([0, 5]))
    assert e.value.code == -4
    with pytest.raises(capi.RdisHipError) as e:          # the same factor twice
        capi.Plan(g, np.array([0, 9]), np.arange(9), np.array([0, 2]), np.array([0, 0]))
    assert e.value.code == -4


@pytest.mark.parametrize("threads", [64, 256, 768, 1024])
def test_workgroup_size_does_not_change_the_algorithm(threads, gctx):
    pp = P.load_bal(ncams=5, npts=30)
    _, r, trv = solve(gctx, pp, maxiters=6, trace=4096, opts={"block_threads": threads})
    check_replay(pp, trv, r, 6)


STREAM_CASES = {
    "ladybug_49_500": (lambda: P.load_bal(ncams=49, npts=500), 6, {"coop_min_factors": 1000, "force_stream": 1}),
    "ladybug_full": (lambda: P.load_bal(), 4, {"force_stream": 1}),
    "sinusoid_nlp": (lambda: _with_x0(P.make_high_dim_sinusoid(), np.random.default_rng(4).uniform(-6, 6, 121)), 12,
                     {"coop_min_factors": 100}),
}


@pytest.mark.parametrize("solver", ["point-major", "grid"])
@pytest.mark.parametrize("name", list(STREAM_CASES))
def test_streaming_grid_solver_replays_against_oracle(name, solver, gctx):
    """components too large for the register-resident solver (or not bundle adjustment) are solved by a streaming
    multi-workgroup kernel: same algorithm, checked the same way.  Since round 6 a bundle-adjustment component whose cameras
    fit the LDS streams through the point-major solver (solver_ptm.hpp: a group of workgroups on the one component, wide
    groups for the really large ones); ptm_stream = 0 keeps it on the grid solver it replaces (solver_stream.hpp), which
    still takes nonlinear-product components and components with more cameras than fit."""
    make, mit, opts = STREAM_CASES[name]
    pp = make()
    opts = dict(opts, **({"ptm_stream": 0} if solver == "grid" else {}))
    g, r, trv = solve(gctx, pp, maxiters=mit, trace=1 << 13, opts=opts)
    check_replay(pp, trv, r, mit)
    assert np.array_equal(g.get_x(), r.x)
    o = O.OracleProblem(pp, emulate_stale_cache=False)
    o.assign(None, r.x)
    assert abs(o.eval() - r.fret[0]) <= 1e-12 * abs(r.fret[0])
    # dir is shared by the plans of a problem and must be left zero: a second, different solve on the
    # same problem still replays
    _, r2, trv2 = solve(gctx, pp, maxiters=2, trace=4096, opts={"coop_min_factors": 0})
    check_replay(pp, trv2, r2, 2)


def _plan_of(gctx, pp, opts):
    g = capi.Problem(gctx, pp)
    plan = capi.Plan(g)
    for k, v in opts.items():
        plan.set_option(k, v)
    return g, plan


@pytest.mark.parametrize("case", ["ladybug as 64 workgroups", "24 cameras x 30000 points", "more than 2^20 points"])
def test_wide_point_major_group_on_one_large_component(case, gctx):
    """ONE bundle-adjustment component on a large share of the device (cgd_ptmg_kernel<512, ., true>: consecutive workgroups
    over all XCDs, one exchange entry a workgroup, the partial camera gradients added by shares): the solver that takes what
    is too large for the register-resident cooperative solver when its cameras fit the LDS.  Replayed against the oracle like
    every solver, the same bits twice, the dispatcher's choice reported."""
    if case == "ladybug as 64 workgroups":
        pp, mit, opts, K = P.load_bal(), 4, {"force_stream": 1, "ptm_group": 64}, 64
    elif case == "24 cameras x 30000 points":
        pp, mit, opts, K = P.make_synthetic_ba(1, 24, 30000, obs_per_pt=4), 3, {}, None
    else:
        pp, mit, opts, K = P.make_synthetic_ba(1, 6, (1 << 20) + 4096, obs_per_pt=2), 1, {}, None
    pp.single_component()
    g, plan = _plan_of(gctx, pp, dict(opts, trace_records=1 << 13, dump_iters=mit))
    ends = []
    for _ in range(2):
        plan.set_start(pp.x0)
        plan.solve(mit, 3e-8)
        r = plan.fetch()
        ends.append((r.fret[0], r.x.tobytes()))
    assert ends[0] == ends[1]                                  # fixed order of every sum
    assert plan.info("components_point_major") == 1 and plan.info("components_grid_stream") == 0 and plan.info("point_major_wide") == 1
    assert plan.info("point_major_group") == (K or plan.info("point_major_group")) and plan.info("point_major_group") > 16
    tr, n = plan.get_trace(0, 1 << 13)
    check_replay(pp, (tr[:n], plan.get_vectors(0, mit)), r, mit)
    assert r.delta[0] < 0 and np.array_equal(g.get_x(), r.x)
    # ... and the grid solver it replaces ends where it does after one line minimisation
    g2, plan2 = _plan_of(gctx, pp, {"ptm_stream": 0, **({"force_stream": 1} if "force_stream" in opts else {})})
    plan2.set_start(pp.x0)
    plan2.solve(1, 3e-8)
    r2 = plan2.fetch()
    plan.set_start(pp.x0)
    plan.solve(1, 3e-8)
    r1 = plan.fetch()
    assert plan2.info("components_grid_stream") == 1
    assert abs(r1.fret[0] - r2.fret[0]) <= 1e-6 * abs(r2.fret[0])      # (one line minimisation: Brent's own tolerance)


@pytest.mark.parametrize("case", ["300 cameras x 60000 points", "forced on 24 cameras x 30000 points"])
def test_wide_group_with_local_camera_numbering(case, gctx):
    """A component with more cameras than a compute unit's LDS holds (about 125): the wide group's workgroups each keep only the
    cameras their own chunks meet -- contiguous shares of the camera-sorted chunk order -- under local numbers; the partial camera
    gradients meet through per-camera lists of (rank, local number) (solver_ptm.hpp: LOCAL).  Replayed against the oracle, the
    same bits twice, against the grid solver after one line minimisation; and forced (option ptm_local_cameras = 1) on a
    component whose cameras would fit, where it must agree with the all-cameras form to rounding."""
    if case.startswith("300"):
        pp, opts = P.make_synthetic_ba(1, 300, 60000, obs_per_pt=4), {}
    else:
        pp, opts = P.make_synthetic_ba(1, 24, 30000, obs_per_pt=4), {"ptm_local_cameras": 1}
    pp.single_component()
    mit = 3
    g, plan = _plan_of(gctx, pp, dict(opts, trace_records=1 << 13, dump_iters=mit))
    ends = []
    for _ in range(2):
        plan.set_start(pp.x0)
        plan.solve(mit, 3e-8)
        r = plan.fetch()
        ends.append((r.fret[0], r.x.tobytes()))
    assert ends[0] == ends[1]
    assert plan.info("components_point_major") == 1 and plan.info("point_major_wide") == 1 and plan.info("components_grid_stream") == 0
    held = plan.info("point_major_local_cameras")
    assert 0 < held <= 125 and plan.info("point_major_group") > 16      # (a workgroup's cameras: what its LDS holds)
    tr, n = plan.get_trace(0, 1 << 13)
    check_replay(pp, (tr[:n], plan.get_vectors(0, mit)), r, mit)
    assert r.delta[0] < 0 and np.array_equal(g.get_x(), r.x)
    others = [{"ptm_stream": 0}] + ([{"ptm_local_cameras": 0}] if opts else [])
    plan.set_start(pp.x0)
    plan.solve(1, 3e-8)
    r1 = plan.fetch()
    for o in others:
        g2, plan2 = _plan_of(gctx, pp, o)
        plan2.set_start(pp.x0)
        plan2.solve(1, 3e-8)
        r2 = plan2.fetch()
        assert plan2.info("point_major_local_cameras") == 0
        assert abs(r1.fret[0] - r2.fret[0]) <= 1e-6 * abs(r2.fret[0]), o


@pytest.mark.parametrize("ncams", [24, 300])
def test_wide_group_with_every_camera_constant(ncams, gctx):
    """the points of a large problem as ONE component against constant cameras (a caller may pass that): the wide group without
    gradient rounds or camera exchange (ROT_CAMFIX), with all cameras in the LDS (24) and with local camera numbering (300)"""
    pp = P.make_synthetic_ba(1, ncams, 60000 if ncams > 100 else 30000, obs_per_pt=4)
    fv = np.arange(9 * ncams, pp.nvars, dtype=np.int64)
    fc = np.arange(pp.nfac, dtype=np.int64)
    g, r, trv = solve(gctx, pp, maxiters=3, free_vid=fv, fac_id=fc, trace=1 << 13)
    check_replay(pp, trv, r, 3, free_vid=fv, fac_id=fc, x=pp.x0[fv])
    assert r.delta[0] < 0 and np.array_equal(g.get_x(fv), r.x) and np.array_equal(g.get_x(np.arange(9 * ncams)), pp.x0[:9 * ncams])


def test_component_beyond_register_capacity_uses_streaming(gctx):
    # one synthetic component with more factors than the register-resident solver has lanes (65536)
    pp = P.make_synthetic_ba(1, 12, 20000, obs_per_pt=4)
    assert pp.nfac == 80000
    _, r, trv = solve(gctx, pp, maxiters=3, trace=4096)
    check_replay(pp, trv, r, 3)
    assert r.delta[0] < 0 and (r.status[0] & 0xFF) in (0, 3)


def test_camera_and_point_component_batches(gctx):
    """the component mix RDIS reaches on a BAL problem: all cameras (points fixed) in one launch,
    then all points (cameras fixed) in one launch; members replay against the oracle"""
    pp = P.load_bal(ncams=49, npts=500)
    g = capi.Problem(gctx, pp)
    obj = g.eval()
    for name, (free_ptr, free_vid, fac_ptr, fac_id) in zip(("cameras", "points"), P.ba_alternation_plans(pp)):
        plan = capi.Plan(g, free_ptr, free_vid, fac_ptr, fac_id)
        plan.set_option("trace_records", 4096)
        plan.set_option("dump_iters", 25)
        plan.set_start(None)                      # start from the currently assigned values
        x_before = g.get_x()
        plan.solve(25, 3e-8)
        r = plan.fetch()
        assert np.all(r.delta <= 0) and np.all((r.status & 0xFF) != 5)
        new_obj = g.eval()
        assert abs(new_obj - np.sum(r.fret)) <= 1e-12 * new_obj          # every factor is in exactly one component
        assert abs((obj + np.sum(r.delta)) - new_obj) <= 1e-10 * obj
        obj = new_obj
        ncomp = len(free_ptr) - 1
        for c in (0, ncomp // 3, ncomp - 1):
            fv, fc = free_vid[free_ptr[c]:free_ptr[c + 1]], fac_id[fac_ptr[c]:fac_ptr[c + 1]]
            q = P.load_bal(ncams=49, npts=500)
            q.x0 = x_before                                               # the oracle sees the same constants
            tr, n = plan.get_trace(c, 4096)
            rep = O.OracleProblem(q).replay(tr, free_vid=fv, fac=fc, x=x_before[fv], maxiters=25,
                                            vdump=plan.get_vectors(c, 25)[:int(r.iters[c]) + 1])
            assert rep.step_mismatches == 0 and rep.tag_mismatches == 0 and rep.underrun == 0 and rep.consumed == n, (name, c, rep)
            # one camera against fixed points: the objective is a few hundred residuals of one
            # projection, less averaging than in the full problem -- 1e-11 (observed up to 2e-12)
            assert rep.max_f_rel_near <= 1e-11 and rep.max_slope_rel_near <= 1e-11 and rep.fret == r.fret[c], (name, c, rep)
            assert rep.max_f_rel <= 1e-6 and rep.max_slope_rel <= 1e-6, (name, c, rep)


def test_cooperative_exchange_is_bit_reproducible(gctx):
    """a stale or torn read in the inter-workgroup exchange would change a sum and, through the
    chaotic trajectory, the whole result: 12 repeated solves (register-resident and streaming
    grid solvers, two workgroup counts) must agree to the last bit"""
    pp = P.load_bal().single_component()
    g = capi.Problem(gctx, pp)
    for opts in ({}, {"coop_threads": 128}, {"force_stream": 1}):
        plan = capi.Plan(g)
        for k, v in opts.items():
            plan.set_option(k, v)
        ref = None
        for rep in range(4):
            plan.set_start(pp.x0)
            plan.solve(10, 3e-8)
            r = plan.fetch()
            assert (r.status[0] & 0xFF) == 3, r.status                     # never a sync timeout
            if ref is None:
                ref = r
            else:
                assert r.fret[0] == ref.fret[0] and np.array_equal(r.x, ref.x) and r.nfeval[0] == ref.nfeval[0], opts
        plan.close()


def test_quad_solver_point_components(gctx):
    """four (or sixteen) lanes per tiny component, sixteen (four) machines per wave (solver_quad.hpp), forced on for
    ladybug's point components: same contract, members replay against the oracle, empty and
    single-factor components included"""
    pp = P.load_bal(ncams=49, npts=700)
    g = capi.Problem(gctx, pp)
    a = np.zeros(pp.nvars, np.uint8); a[:441] = 1
    comps = g.components(a)
    free_ptr, free_vid, fac_ptr, fac_id = comps
    results = {}
    for quad in (4, 16, 0):                       # lanes per component; 0: a workgroup each
        g.set_x(pp.x0)
        plan = capi.Plan(g, *comps)
        plan.set_option("quad_min_components", 1 if quad == 4 else 1 << 40)
        plan.set_option("row_min_components", 1 if quad == 16 else 1 << 40)
        plan.set_option("trace_records", 2048)
        plan.set_option("dump_iters", 25)
        plan.set_start(None)
        plan.solve(25, 3e-8)
        r = plan.fetch()
        results[quad] = r
        if not quad:
            continue
        assert plan.last_kernel_ms()[1] == 1                                 # one launch, the group kernel
        assert np.all(r.delta <= 0) and np.all((r.status & 0xFF) != 5)
        new_obj = g.eval()
        assert abs(new_obj - np.sum(r.fret)) <= 1e-12 * new_obj              # every factor in exactly one component
        assert np.array_equal(g.get_x()[free_vid], r.x)                      # variables left assigned
        ncomp = len(free_ptr) - 1
        for c in (0, 1, ncomp // 2, ncomp - 2, ncomp - 1):                   # first / last: lightest and heaviest
            fv, fc = free_vid[free_ptr[c]:free_ptr[c + 1]], fac_id[fac_ptr[c]:fac_ptr[c + 1]]
            tr, n = plan.get_trace(c, 2048)
            rep = O.OracleProblem(pp).replay(tr, free_vid=fv, fac=fc, x=pp.x0[fv], maxiters=25,
                                             vdump=plan.get_vectors(c, 25)[:int(r.iters[c]) + 1])
            assert rep.step_mismatches == 0 and rep.tag_mismatches == 0 and rep.underrun == 0 and rep.consumed == n, (c, rep)
            assert rep.max_f_rel_near <= 1e-11 and rep.max_slope_rel_near <= 1e-11 and rep.fret == r.fret[c], (c, rep)
            assert rep.iters == r.iters[c] and rep.reason == (r.status[c] & 0xFF)
    # against one workgroup per component: same algorithm, sums in a different order
    for lanes in (4, 16):
        _agree(results[lanes], results[0])


def _agree(rq, rw):
    assert np.array_equal(rq.status & 0xFF, rw.status & 0xFF) or np.mean((rq.status & 0xFF) == (rw.status & 0xFF)) > 0.98
    conv = ((rq.status & 0xFF) != 3) & ((rw.status & 0xFF) != 3)
    dev = np.abs(rq.fret[conv] - rw.fret[conv]) / (1.0 + np.abs(rw.fret[conv]))
    assert np.quantile(dev, 0.99) <= 1e-6 and np.max(dev) <= 1e-4           # converged members agree (stop test: 3e-8 on the decrease)


def test_quad_solver_partial_blocks_and_active_bounds(gctx):
    """quad solver on components that are not whole point blocks (1 or 2 free coordinates, the rest
    constants) and with domains tight enough that the clamp is active during the line searches"""
    rng = np.random.default_rng(23)
    pp = P.load_bal(ncams=49, npts=300)
    pp.lo[441:] = pp.x0[441:] - rng.uniform(0.002, 0.05, pp.nvars - 441)
    pp.hi[441:] = pp.x0[441:] + rng.uniform(0.002, 0.05, pp.nvars - 441)
    g = capi.Problem(gctx, pp)
    a = np.ones(pp.nvars, np.uint8)
    keep = rng.random(300) < 0.7
    for p in np.where(keep)[0]:                      # free 1..3 coordinates of 70 % of the points
        k = rng.integers(1, 4)
        a[441 + 3 * p + rng.choice(3, size=k, replace=False)] = 0
    comps = g.components(a)
    free_ptr, free_vid, fac_ptr, fac_id = comps
    assert set(np.diff(free_ptr)) <= {1, 2, 3} and len(set(np.diff(free_ptr))) == 3
    plan = capi.Plan(g, *comps)
    plan.set_option("quad_min_components", 1)
    plan.set_option("trace_records", 2048)
    plan.set_option("dump_iters", 25)
    plan.set_start(None)
    plan.solve(25, 3e-8)
    r = plan.fetch()
    assert plan.last_kernel_ms()[1] == 1
    assert np.all(r.x >= pp.lo[free_vid]) and np.all(r.x <= pp.hi[free_vid])
    assert np.any((r.x == pp.lo[free_vid]) | (r.x == pp.hi[free_vid]))          # some results sit on their bounds
    ncomp = len(free_ptr) - 1
    for c in rng.choice(ncomp, size=12, replace=False):
        fv, fc = free_vid[free_ptr[c]:free_ptr[c + 1]], fac_id[fac_ptr[c]:fac_ptr[c + 1]]
        tr, n = plan.get_trace(c, 2048)
        rep = O.OracleProblem(pp).replay(tr, free_vid=fv, fac=fc, x=pp.x0[fv], maxiters=25,
                                         vdump=plan.get_vectors(c, 25)[:int(r.iters[c]) + 1])
        assert rep.step_mismatches == 0 and rep.tag_mismatches == 0 and rep.underrun == 0 and rep.consumed == n, (c, rep)
        assert rep.max_f_rel_near <= 1e-11 and rep.max_slope_rel_near <= 1e-11, (c, rep)
        if not (r.status[c] & capi.STATUS_ROLLED_BACK):
            assert rep.fret == r.fret[c], (c, rep)


@pytest.mark.parametrize("solver", ["workgroup", "row", "quad"])
def test_camera_rotation_records_change_nothing(gctx, solver):
    """launches whose components leave every camera constant read per-camera rotation records
    (camera_rotations_kernel) instead of redoing the angle / axis / sine / cosine per factor and
    trial point: the arithmetic is the same, so every output is bit-identical with the records off"""
    pp = P.load_bal(ncams=49, npts=900)
    g = capi.Problem(gctx, pp)
    a = np.zeros(pp.nvars, np.uint8); a[:441] = 1
    comps = g.components(a)
    out = {}
    for rec in (1, 0):
        g.set_x(pp.x0)
        plan = capi.Plan(g, *comps)
        plan.set_option("quad_min_components", 1 if solver == "quad" else 1 << 40)
        plan.set_option("row_min_components", 1 if solver == "row" else 1 << 40)
        plan.set_option("camera_records", rec)
        plan.set_option("trace_records", 1024)
        plan.set_start(None)
        plan.solve(25, 3e-8)
        out[rec] = (plan.fetch(), [plan.get_trace(c, 1024) for c in (0, 5, 450, 899)])
        plan.close()
    r1, r0 = out[1][0], out[0][0]
    assert np.array_equal(r1.fret, r0.fret) and np.array_equal(r1.x, r0.x) and np.array_equal(r1.delta, r0.delta)
    assert np.array_equal(r1.iters, r0.iters) and np.array_equal(r1.status, r0.status)
    assert np.array_equal(r1.nfeval, r0.nfeval) and np.array_equal(r1.ngeval, r0.ngeval)
    for (t1, n1), (t0, n0) in zip(out[1][1], out[0][1]):
        assert n1 == n0 and np.array_equal(t1[:n1], t0[:n0])


def test_handles_close_in_any_order(gctx):
    """closing a problem closes its plans first, so wrappers finalised late (the frames of a failed
    test, a garbage cycle) never hand the library a dangling pointer"""
    pp = P.load_bal(ncams=5, npts=30)
    ctx = capi.Context(0)
    g = capi.Problem(ctx, pp)
    plan = capi.Plan(g)
    plan.solve(2, 3e-8)
    g.close()
    assert plan.h is None and g.h is None
    plan.close(); plan.close()
    g2 = capi.Problem(ctx, pp)
    p2 = capi.Plan(g2)
    ctx.close()
    assert p2.h is None and g2.h is None and ctx.h is None
    del p2, g2, plan, g


def test_camera_rotation_records_with_free_cameras(gctx):
    """... and when cameras are free a component rewrites the records of its own cameras at every
    trial point (option 2 forces it; by default only components of more than 2048 factors do):
    bit-identical again -- components with cameras and points free, cameras only, rotations only"""
    pp = P.make_synthetic_ba(12, 3, 40)
    whole = (pp.comp_free_ptr, pp.comp_free_vid, pp.comp_fac_ptr, pp.comp_fac_id)
    lb = P.load_bal(ncams=49, npts=600)
    cams, _ = P.ba_alternation_plans(lb)
    rot_only = (np.arange(0, 3 * 49 + 1, 3), np.concatenate([np.arange(9 * c, 9 * c + 3) for c in range(49)]),
                cams[2], cams[3])
    for prob, comps, opts in ((pp, whole, {"coop_min_factors": 0}), (lb, cams, {"coop_min_factors": 0}),
                              (lb, rot_only, {"coop_min_factors": 0})):
        g = capi.Problem(gctx, prob)
        out = {}
        for rec in (2, 0):
            g.set_x(prob.x0)
            plan = capi.Plan(g, *comps)
            plan.set_option("camera_records", rec)
            for k, v in opts.items():
                plan.set_option(k, v)
            plan.set_option("trace_records", 1024)
            plan.set_start(None)
            plan.solve(12, 3e-8)
            out[rec] = (plan.fetch(), plan.get_trace(0, 1024), g.get_x())
            plan.close()
        (r2, (t2, n2), x2), (r0, (t0, n0), x0) = out[2], out[0]
        assert np.array_equal(r2.fret, r0.fret) and np.array_equal(r2.x, r0.x) and np.array_equal(x2, x0)
        assert np.array_equal(r2.iters, r0.iters) and np.array_equal(r2.nfeval, r0.nfeval) and np.array_equal(r2.status, r0.status)
        assert n2 == n0 and np.array_equal(t2[:n2], t0[:n0])
        g.close()


def test_batched_launch_overlaps_cooperative_launches(gctx):
    """a plan with both kinds of components runs its batched launch on a second stream next to
    the cooperative ones (disjoint components): same results as one after the other"""
    pp = P.load_bal(ncams=49, npts=1500)
    g = capi.Problem(gctx, pp)
    a = np.zeros(pp.nvars, np.uint8); a[:9 * 46] = 1            # 3 cameras stay free: one large component + single points
    comps = g.components(a)
    assert np.diff(comps[2]).max() > 1000 and len(comps[0]) - 1 > 500
    out = {}
    for overlap in (1, 0):
        g.set_x(pp.x0)
        plan = capi.Plan(g, *comps)
        plan.set_option("coop_min_factors", 1000)
        plan.set_option("overlap_batch", overlap)
        plan.set_start(None)
        plan.solve(25, 3e-8)
        out[overlap] = (plan.fetch(), g.get_x(), plan.last_kernel_ms()[1])
        plan.close()
    (r1, x1, n1), (r0, x0, n0) = out[1], out[0]
    assert n1 == n0 == 2                                         # one cooperative launch, one batched launch
    assert np.array_equal(r1.fret, r0.fret) and np.array_equal(r1.x, r0.x) and np.array_equal(x1, x0)
    assert np.array_equal(r1.iters, r0.iters) and np.array_equal(r1.status, r0.status) and np.all((r1.status & 0xFF) != 5)


def test_cooperative_groups_side_by_side(gctx):
    """components of some size each get a cooperative group of workgroups, all groups in one launch
    when they fit the device together (ladybug's 49 camera components: 361..906 factors, 9 variables,
    every one of them a long gradient run): members replay against the oracle; packed into several
    smaller launches the same groups give the same bits"""
    pp = P.load_bal()
    cams, _ = P.ba_alternation_plans(pp)
    free_ptr, free_vid, fac_ptr, fac_id = cams
    g = capi.Problem(gctx, pp)
    out = {}
    for mode in ("one launch", "chunks", "workgroup each"):
        g.set_x(pp.x0)
        plan = capi.Plan(g, *cams)
        if mode == "chunks":
            plan.set_option("coop_workgroups", 16)             # group mode needs all groups resident: off ...
            plan.set_option("coop_min_factors", 300)           # ... the few-large-components rule takes them instead
            plan.set_option("coop_max_components", 49)
        if mode == "workgroup each":
            plan.set_option("coop_group_min_factors", 0)
        plan.set_option("trace_records", 4096)
        plan.set_option("dump_iters", 25)
        plan.set_start(None)
        plan.solve(25, 3e-8)
        r = plan.fetch()
        out[mode] = (r, g.get_x(), plan.last_kernel_ms()[1])
        assert np.all(r.delta <= 0) and np.all((r.status & 0xFF) != 5), mode
        assert np.array_equal(g.get_x()[free_vid], r.x)
        if mode == "one launch":
            for c in (0, 17, 48):
                fv, fc = free_vid[free_ptr[c]:free_ptr[c + 1]], fac_id[fac_ptr[c]:fac_ptr[c + 1]]
                tr, n = plan.get_trace(c, 4096)
                rep = O.OracleProblem(pp).replay(tr, free_vid=fv, fac=fc, x=pp.x0[fv], maxiters=25,
                                                 vdump=plan.get_vectors(c, 25)[:int(r.iters[c]) + 1])
                assert rep.step_mismatches == 0 and rep.tag_mismatches == 0 and rep.underrun == 0 and rep.consumed == n, (c, rep)
                assert rep.max_f_rel_near <= 1e-11 and rep.max_slope_rel_near <= 1e-11 and rep.fret == r.fret[c], (c, rep)
        plan.close()
    (r1, x1, n1), (rc, xc, nc), (rw, xw, nw) = out["one launch"], out["chunks"], out["workgroup each"]
    assert n1 == 1 and nc > 4 and nw == 1
    assert np.array_equal(r1.fret, rc.fret) and np.array_equal(r1.x, rc.x) and np.array_equal(x1, xc) and np.array_equal(r1.nfeval, rc.nfeval)
    assert abs(np.sum(r1.fret) - np.sum(rw.fret)) <= 0.05 * np.sum(rw.fret)     # same algorithm, sums in another order, 25 unconverged iterations


def test_cooperative_groups_side_by_side_with_the_reference_slope(gctx):
    """the parity option (factor_rounding = 1) where several cooperative groups share a launch: every group forms its trials'
    slopes the reference's way in its own part of the exchange arrays (solver_coop.hpp: slope_reference).  Members replay
    against the oracle; the same groups packed into smaller launches give the same bits; and the option is another
    trajectory than the default's."""
    pp = P.load_bal()
    cams, _ = P.ba_alternation_plans(pp)
    free_ptr, free_vid, fac_ptr, fac_id = cams
    g = capi.Problem(gctx, pp)
    out = {}
    for mode in ("one launch", "chunks", "default rounding"):
        g.set_x(pp.x0)
        plan = capi.Plan(g, *cams)
        if mode != "default rounding":
            plan.set_option("factor_rounding", 1)
        plan.set_option("lds_resident", 0)                     # (the cooperative groups, not the LDS-resident batch solver)
        if mode == "chunks":
            plan.set_option("coop_workgroups", 16)
            plan.set_option("coop_min_factors", 300)
            plan.set_option("coop_max_components", 49)
        plan.set_option("trace_records", 4096)
        plan.set_option("dump_iters", 12)
        plan.set_start(None)
        plan.solve(12, 3e-8)
        r = plan.fetch()
        assert plan.info("components_cooperative") == 49, (mode, plan.info("components_cooperative"))
        if mode != "default rounding":
            assert plan.info("pipelined") == 0, mode                     # (the reference's slope exists in the plain layout)
        out[mode] = r
        assert np.all(r.delta <= 0) and np.array_equal(g.get_x()[free_vid], r.x)
        if mode == "one launch":
            for c in (0, 17, 48):
                fv, fc = free_vid[free_ptr[c]:free_ptr[c + 1]], fac_id[fac_ptr[c]:fac_ptr[c + 1]]
                sub = type("R", (), {"status": r.status[c:c + 1], "iters": r.iters[c:c + 1], "fret": r.fret[c:c + 1]})
                # (a camera component's gg / dgg are sums of nine terms that cancel: 1e-8, as for the other solvers' camera members)
                check_replay(pp, (plan.get_trace(c, 4096)[0], plan.get_vectors(c, 12)), sub, 12, free_vid=fv, fac_id=fc, x=pp.x0[fv], iter_tol=1e-8)
        plan.close()
    r1, rc, rd = out["one launch"], out["chunks"], out["default rounding"]
    assert np.array_equal(r1.fret, rc.fret) and np.array_equal(r1.x, rc.x) and np.array_equal(r1.nfeval, rc.nfeval)
    assert not np.array_equal(r1.fret, rd.fret) and abs(np.sum(r1.fret) - np.sum(rd.fret)) <= 0.05 * np.sum(rd.fret)


@pytest.mark.parametrize("lanes", [4, 16])
def test_persistent_groups_take_components_off_the_list(gctx, lanes):
    """the tiny-component kernels are persistent: with the grid capped at one block (64 or 4
    groups) every group works through many components, one after the other -- and gives each of
    them the bits it gets when every component has a group of its own"""
    pp = P.load_bal(ncams=49, npts=1200)
    g = capi.Problem(gctx, pp)
    a = np.zeros(pp.nvars, np.uint8); a[:441] = 1
    comps = g.components(a)
    out = {}
    for cap in (0, 1):
        g.set_x(pp.x0)
        plan = capi.Plan(g, *comps)
        plan.set_option("quad_min_components", 1 if lanes == 4 else 1 << 40)
        plan.set_option("row_min_components", 1 if lanes == 16 else 1 << 40)
        plan.set_option("tiny_max_blocks", cap)
        plan.set_option("trace_records", 256)
        plan.set_start(None)
        plan.solve(25, 3e-8)
        out[cap] = (plan.fetch(), g.get_x(), [plan.get_trace(c, 256) for c in (0, 64, 65, 700, 1199)])
        assert plan.last_kernel_ms()[1] == 1
        plan.close()
    (r0, x0, t0), (r1, x1, t1) = out[0], out[1]
    assert np.array_equal(r0.fret, r1.fret) and np.array_equal(r0.x, r1.x) and np.array_equal(x0, x1)
    assert np.array_equal(r0.iters, r1.iters) and np.array_equal(r0.status, r1.status) and np.array_equal(r0.nfeval, r1.nfeval)
    for (ta, na), (tb, nb) in zip(t0, t1):
        assert na == nb and np.array_equal(ta[:min(na, 256)], tb[:min(nb, 256)])


def test_large_components_packed_into_one_cooperative_launch(gctx):
    """a few large components share a cooperative launch (each its own group of workgroups and its
    own exchange state): same bits as one launch per component"""
    pp = P.make_synthetic_ba(5, 12, 700, obs_per_pt=4)           # 5 components of 2800 factors
    comps = (pp.comp_free_ptr, pp.comp_free_vid, pp.comp_fac_ptr, pp.comp_fac_id)
    g = capi.Problem(gctx, pp)
    out = {}
    for label, opts in (("packed", {"coop_min_factors": 1000}), ("one each", {"coop_min_factors": 1000, "coop_workgroups": 24})):
        g.set_x(pp.x0)
        plan = capi.Plan(g, *comps)
        for k, v in opts.items():
            plan.set_option(k, v)
        plan.set_start(None)
        plan.solve(10, 3e-8)
        out[label] = (plan.fetch(), g.get_x(), plan.last_kernel_ms()[1])
        plan.close()
    (ra, xa, na), (rb, xb, nb) = out["packed"], out["one each"]
    assert na == 1 and nb == 5
    assert np.array_equal(ra.fret, rb.fret) and np.array_equal(ra.x, rb.x) and np.array_equal(xa, xb)
    assert np.array_equal(ra.iters, rb.iters) and np.array_equal(ra.nfeval, rb.nfeval) and np.all((ra.status & 0xFF) != 5)
    assert np.all(ra.delta < 0)


def test_pipelined_groups_give_the_bits_of_the_plain_cooperative_solver(gctx):
    """solver_pipe.hpp (control logic and exchange on waves of their own, guesses at the following trial
    steps evaluated ahead) against solver_coop.hpp: the sums are formed entry by entry in the same order
    and the control logic never sees a guess, so values, call counts and the whole trace are the same
    bits -- with speculation on or off, for one group (full ladybug; a sub-function with constants)
    and for several groups side by side in one launch"""
    lb = P.load_bal()
    sub = P.load_bal(ncams=49, npts=500)
    # (few cameras: both layouts then have a wave for every variable fed by many partials, whose
    # sums are the one place where the order depends on how many waves a group has)
    syn = P.make_synthetic_ba(5, 3, 900, obs_per_pt=3)
    whole = lambda pp: (np.array([0, pp.nvars]), np.arange(pp.nvars, dtype=np.int64), np.array([0, pp.nfac]), np.arange(pp.nfac, dtype=np.int64))
    # the first 441 variables (the cameras) of the sub-problem held constant: only points are free
    cams_const = (np.array([0, sub.nvars - 441]), np.arange(441, sub.nvars, dtype=np.int64), np.array([0, sub.nfac]), np.arange(sub.nfac, dtype=np.int64))
    # bounds that bite: every seventh variable may move 1e-4 of its size up, every eleventh down (the clamp
    # of a trial point is part of the lanes' arithmetic in both solvers)
    import copy
    tight = copy.deepcopy(sub)
    idx = np.arange(tight.nvars)
    tight.hi = np.where(idx % 7 == 0, tight.x0 + 1e-4 * np.abs(tight.x0) + 1e-9, tight.hi)
    tight.lo = np.where(idx % 11 == 0, tight.x0 - 1e-4 * np.abs(tight.x0) - 1e-9, tight.lo)
    cases = (("ladybug", lb, whole(lb), {}, 25),
             ("active bounds", tight, whole(tight), {"coop_min_factors": 1000}, 8),
             ("cameras constant", sub, cams_const, {"coop_min_factors": 1000, "coop_max_components": 4}, 6),
             ("five groups", syn, (syn.comp_free_ptr, syn.comp_free_vid, syn.comp_fac_ptr, syn.comp_fac_id), {"coop_min_factors": 1000}, 10)) + tuple(
        # odd shapes: more variables than factors (the variables decide the number of lanes), a last workgroup
        # that is mostly empty, one factor beyond a full workgroup
        ("%d cameras x %d points x %d" % (c, q, o), sp, whole(sp), {"coop_min_factors": 1000}, 6)
        for c, q, o in ((2, 1500, 2), (4, 1000, 4), (3, 1067, 3), (2, 1153, 2)) for sp in (P.make_synthetic_ba(1, c, q, obs_per_pt=o),))
    for name, pp, comps, opts, iters in cases:
        g = capi.Problem(gctx, pp)
        out = {}
        for label, extra in (("plain", {"coop_pipeline": 0}), ("pipelined", {"coop_pipeline": 1}), ("no guesses", {"coop_pipeline": 1, "coop_speculate": 0})):
            g.set_x(pp.x0)
            plan = capi.Plan(g, *comps)
            for k, v in {**opts, **extra}.items():
                plan.set_option(k, v)
            plan.set_option("trace_records", 4096)
            plan.set_option("dump_iters", iters)
            plan.set_start(None)
            plan.solve(iters, 3e-8)
            r = plan.fetch()
            ncomp = len(comps[0]) - 1
            out[label] = (r, g.get_x(), [plan.get_trace(c, 4096) for c in range(ncomp)], plan.last_kernel_ms()[1])
            assert np.all((r.status & 0xFF) != 7), (name, label)      # no exchange gave up
            plan.close()
        ra, xa, ta, na = out["plain"]
        assert np.all(ra.delta <= 0) and np.all(ra.nfeval > 20), name
        if name == "active bounds":
            assert np.sum(ra.x == tight.hi) + np.sum(ra.x == tight.lo) > 10      # ... and they do bite
        for label in ("pipelined", "no guesses"):
            rb, xb, tb, nb = out[label]
            assert na == nb == 1, (name, label)
            assert np.array_equal(ra.fret, rb.fret) and np.array_equal(ra.x, rb.x) and np.array_equal(xa, xb), (name, label)
            assert np.array_equal(ra.iters, rb.iters) and np.array_equal(ra.status, rb.status), (name, label)
            assert np.array_equal(ra.nfeval, rb.nfeval) and np.array_equal(ra.ngeval, rb.ngeval), (name, label)
            for (tra, ca), (trb, cb) in zip(ta, tb):
                assert ca == cb and np.array_equal(tra[:min(ca, 4096)], trb[:min(cb, 4096)]), (name, label)


def test_pipelined_groups_move_to_the_line_minimum_before_an_ftol_exit(gctx):
    """REQ_LINE_END is the one request the pipelined stepper does not wait for; on an
    ftol exit the next post (REQ_DONE) follows a few hundred cycles later, and a lane wave still busy
    with a guess used to jump to the newest post and skip the move to the line minimum -- x of its
    variables was the point before the step while fret reported the moved one.  The lanes now act on
    every post in order.  Run to EXIT_FTOL under several tolerances (different exits of the line
    searches) and compare every bit with the plain cooperative solver, which has no such hand-over;
    the returned value must also be the function at the returned point."""
    cases = [(P.make_synthetic_ba(1, 3, 900, obs_per_pt=3, first_comp=s), ft) for s in (0, 1, 2) for ft in (1e-2, 1e-4, 1e-6, 3e-8)]
    cases.append((P.load_bal(ncams=49, npts=500), 1e-3))
    n_ftol = 0
    for pp, ftol in cases:
        comps = (np.array([0, pp.nvars]), np.arange(pp.nvars, dtype=np.int64), np.array([0, pp.nfac]), np.arange(pp.nfac, dtype=np.int64))
        g = capi.Problem(gctx, pp)
        out = {}
        for label, extra in (("plain", {"coop_pipeline": 0}), ("pipelined", {"coop_pipeline": 1}), ("slow polls", {"coop_pipeline": 1, "coop_poll_delay": 64})):
            g.set_x(pp.x0)
            plan = capi.Plan(g, *comps)
            for k, v in {"coop_min_factors": 1000, **extra}.items():
                plan.set_option(k, v)
            plan.set_start(None)
            plan.solve(600, ftol)
            out[label] = (plan.fetch(), g.get_x())
            plan.close()
        ra, xa = out["plain"]
        n_ftol += int((ra.status[0] & 0xFF) == 0)
        for label in ("pipelined", "slow polls"):
            rb, xb = out[label]
            assert np.array_equal(ra.status, rb.status) and np.array_equal(ra.iters, rb.iters), (ftol, label)
            assert np.array_equal(ra.fret, rb.fret) and np.array_equal(ra.x, rb.x) and np.array_equal(xa, xb), (ftol, label)
            assert np.array_equal(ra.nfeval, rb.nfeval) and np.array_equal(ra.ngeval, rb.ngeval), (ftol, label)
        o = O.OracleProblem(pp, emulate_stale_cache=False)
        o.assign(np.arange(pp.nvars, dtype=np.int64), out["pipelined"][0].x)
        fo = o.eval()
        assert abs(fo - out["pipelined"][0].fret[0]) <= 1e-12 * abs(fo), (ftol, fo)
    assert n_ftol >= 8, n_ftol      # the exit this test is about


def test_lds_resident_batch_solver_gives_the_bits_of_the_plain_one(gctx):
    """solver_lds.hpp keeps a component's variables in LDS slots instead of forming every trial point in
    global memory (solver_wg.hpp).  The factor arithmetic and the orders of all sums are the same, so when
    the slots are the free variables in their listed order (every block free, cameras before points,
    ascending ids) every bit of the result is the same: synthetic components under three workgroup sizes,
    with per-factor rotations and with records (option lds_camera_sums = 0: by default a camera's gradient
    entries are summed across waves that share the camera -- no memory traffic -- which groups the partials
    differently from the plain solver's strided wave sum; that default is replayed against the oracle).  With constants among the slots (ladybug's camera
    components: points fixed; its point components: cameras fixed; a sub-function with partly free
    blocks) the per-factor values are still the same bits and only the Polak-Ribiere sums run over the
    slots in another grouping: there the replay check against the oracle is the judge."""
    syn = P.make_synthetic_ba(40, 5, 64, obs_per_pt=3)
    csr = (syn.comp_free_ptr, syn.comp_free_vid, syn.comp_fac_ptr, syn.comp_fac_id)
    g = capi.Problem(gctx, syn)

    def run(opts, comps=csr, prob=g, pp=syn, iters=25, trace=0):
        prob.set_x(pp.x0)
        plan = capi.Plan(prob, *comps)
        for k, v in opts.items():
            plan.set_option(k, v)
        if trace:
            plan.set_option("trace_records", trace)
            plan.set_option("dump_iters", iters)
        plan.set_start(None)
        plan.solve(iters, 3e-8)
        r = plan.fetch()
        tr = [(plan.get_trace(c, trace)[0], plan.get_vectors(c, iters)) for c in range(len(comps[0]) - 1)] if trace else None
        out = (r, prob.get_x(), plan.last_kernel_ms()[1], tr)
        plan.close()
        return out
    for threads in (128, 256, 768):
        for rot in (0, 2):
            ra, xa, _, _ = run({"lds_resident": 0, "ptm_stream": 0, "block_threads": threads, "camera_records": rot})
            # (lds_camera_sums 0: a camera variable's gradient entry summed as the plain solver sums it -- by default
            # the LDS solver sums camera partials across waves that share a camera, another grouping)
            # (lds_matrix 0, which is also the default: trials in the vector form per factor, the plain solver's arithmetic; the
            # option's matrix form -- another association of the same sums, measured slower -- is replayed against the oracle below)
            rb, xb, nb, _ = run({"lds_resident": 1, "block_threads": threads, "camera_records": rot, "lds_rot": 1 if rot else 0, "lds_camera_sums": 0, "lds_matrix": 0})
            assert nb == 1
            assert np.array_equal(ra.fret, rb.fret) and np.array_equal(ra.x, rb.x) and np.array_equal(xa, xb), (threads, rot)
            assert np.array_equal(ra.iters, rb.iters) and np.array_equal(ra.status, rb.status), (threads, rot)
            assert np.array_equal(ra.nfeval, rb.nfeval) and np.array_equal(ra.ngeval, rb.ngeval), (threads, rot)
    assert np.all(ra.delta < 0)
    # the default (camera partials summed across waves): same problem, replayed against the oracle; and with the trials in
    # matrix form (option lds_matrix: another association of a factor's sums; built, correct, slower here -- not the default)
    for extra in ({}, {"lds_matrix": 1}):
        rc, xc, _, trc = run({"lds_resident": 1, **extra}, trace=4096)
        assert np.all(rc.delta < 0) and abs(rc.fret.sum() - ra.fret.sum()) <= 0.05 * ra.fret.sum()
        for c in (0, 17, 39):
            fv, fc = syn.component(c)
            sub = type("R", (), {"status": rc.status[c:c + 1], "iters": rc.iters[c:c + 1], "fret": rc.fret[c:c + 1]})
            check_replay(syn, trc[c], sub, 25, free_vid=fv, fac_id=fc, x=syn.x0[fv])
    # constants among the slots
    lb = P.load_bal(ncams=49, npts=500)
    cams, pts = P.ba_alternation_plans(lb)
    gl = capi.Problem(gctx, lb)
    for comps, base in ((cams, {"coop_group_min_factors": 0, "coop_min_factors": 0}), (pts, {"row_min_components": 1 << 30, "quad_min_components": 1 << 30})):
        ra, xa, _, _ = run({**base, "lds_resident": 0, "ptm_stream": 0}, comps, gl, lb, 12)
        rb, xb, _, trb = run({**base, "lds_resident": 1}, comps, gl, lb, 12, trace=2048)
        # same decisions as long as the sums agree to the last bit; they may part in the last place -- population check
        assert np.all(rb.delta <= 0) and abs(rb.fret.sum() - ra.fret.sum()) <= 0.05 * abs(ra.fret.sum())
        for c in (0, len(comps[0]) // 2, len(comps[0]) - 2):
            fv = comps[1][comps[0][c]:comps[0][c + 1]]
            fc = comps[3][comps[2][c]:comps[2][c + 1]]
            sub = type("R", (), {"status": rb.status[c:c + 1], "iters": rb.iters[c:c + 1], "fret": rb.fret[c:c + 1]})
            # (a camera's nine gradient entries are sums of 361..906 partials that cancel to a few digits -- in the last iterations before a tolerance exit to very few: 1e-8; observed 1.6e-9)
            check_replay(lb, trb[c], sub, 12, free_vid=fv, fac_id=fc, x=lb.x0[fv], iter_tol=1e-8)
    # partly free blocks: two of a camera's nine, one of a point's three held constant
    free = np.array([v for v in range(lb.nvars) if not (v < 441 and v % 9 in (6, 8)) and not (v >= 441 and v % 3 == 1 and v % 5 == 0)], dtype=np.int64)
    one = (np.array([0, len(free)]), free, np.array([0, lb.nfac]), np.arange(lb.nfac, dtype=np.int64))
    base = {"coop_min_factors": 0, "coop_group_min_factors": 0}
    ra, xa, _, _ = run({**base, "lds_resident": 0, "ptm_stream": 0}, one, gl, lb, 8)
    rb, xb, _, trb = run({**base, "lds_resident": 1}, one, gl, lb, 8, trace=4096)
    if lb.nvars * 44 < 150 * 1024:     # (fits the LDS: the new solver ran)
        check_replay(lb, trb[0], rb, 8, free_vid=free, fac_id=one[3], x=lb.x0[free])
        const = np.setdiff1d(np.arange(lb.nvars), free)
        assert np.array_equal(xb[const], lb.x0[const])


def test_point_major_streaming_solver_alone_and_in_groups(gctx):
    """solver_ptm.hpp: components too large for the LDS keep their cameras in LDS and stream their point blocks
    (one record per block, a lane per point); with fewer components than compute units K workgroups share a
    component (cgd_ptmg_kernel: partial sums exchanged per trial point, two ordered grid barriers per gradient).
    Every variant is replayed against the oracle (bit-identical decisions, values to rounding); the variants
    differ from each other only in the grouping of their sums, so their first line minimisation agrees to 1e-9
    and their end values are draws of the same family.  Option ptm_stream = 2 sends components here that would
    fit the LDS solver (small enough for the oracle to replay in seconds)."""
    syn = P.make_synthetic_ba(5, 6, 700, obs_per_pt=3)     # 2100 factors, 2154 variables per component
    csr = (syn.comp_free_ptr, syn.comp_free_vid, syn.comp_fac_ptr, syn.comp_fac_id)
    g = capi.Problem(gctx, syn)

    def run(opts, comps=csr, prob=g, pp=syn, iters=12, trace=4096):
        prob.set_x(pp.x0)
        plan = capi.Plan(prob, *comps)
        for k, v in opts.items():
            plan.set_option(k, v)
        plan.set_option("trace_records", trace)
        plan.set_option("dump_iters", iters)
        plan.set_start(None)
        plan.solve(iters, 3e-8)
        r = plan.fetch()
        nc = len(comps[0]) - 1
        tr = [(plan.get_trace(c, trace)[0], plan.get_vectors(c, iters)) for c in range(nc)]
        info = {k: plan.info(k) for k in ("components_point_major", "point_major_group", "components_lds", "components_plain")}
        out = (r, prob.get_x(), plan.last_kernel_ms(), tr, info)
        plan.close()
        return out

    def first_line(tr):   # value the first line minimisation ended on
        t = tr[0]
        k = np.nonzero(t[:, 0] == 5)[0][0]
        return t[k, 2]

    base = {"ptm_stream": 2, "coop_group_min_factors": 0, "coop_min_factors": 0}   # (no cooperative groups: the batch list)
    ref = None
    for threads, K in ((768, 1), (256, 1), (768, 2), (512, 3), (256, 4), (256, 8)):
        r, x, (ms, nl), tr, info = run({**base, "ptm_threads": threads, "ptm_group": K})
        assert info["components_point_major"] == 5 and info["components_lds"] == 0 and info["components_plain"] == 0, info
        assert info["point_major_group"] == K and nl == 1, (info, nl)
        assert np.all(r.delta < 0) and np.all((r.status & 0xFF) != 7)
        for c in (0, 4):
            fv, fc = syn.component(c)
            sub = type("R", (), {"status": r.status[c:c + 1], "iters": r.iters[c:c + 1], "fret": r.fret[c:c + 1]})
            check_replay(syn, tr[c], sub, 12, free_vid=fv, fac_id=fc, x=syn.x0[fv], iter_tol=1e-10)
        # the variables are left assigned to what was returned
        assert np.array_equal(x[csr[1]], r.x)
        fl = np.array([first_line(tr[c]) for c in range(5)])
        if ref is None: ref = (r, fl)
        else:
            assert np.all(np.abs(fl - ref[1]) <= 1e-9 * np.abs(ref[1])), (threads, K, fl, ref[1])
            assert abs(r.fret.sum() - ref[0].fret.sum()) <= 0.25 * ref[0].fret.sum()
    # run to run: the same bits (fixed orders of summation in and across the workgroups)
    r1 = run({**base, "ptm_threads": 256, "ptm_group": 4})[0]
    r2 = run({**base, "ptm_threads": 256, "ptm_group": 4})[0]
    assert np.array_equal(r1.fret, r2.fret) and np.array_equal(r1.x, r2.x) and np.array_equal(r1.nfeval, r2.nfeval)

    # a lone workgroup, run to run: the same bits too (the camera partials of a gradient are summed round by round in LDS,
    # rounds in order, within a round by wave and lane)
    ra = run({**base, "ptm_threads": 768, "ptm_group": 1})[0]
    assert np.array_equal(ra.fret, ref[0].fret) and np.array_equal(ra.x, ref[0].x) and np.array_equal(ra.nfeval, ref[0].nfeval)

    # constants among the slots: ladybug's camera components (points fixed) and point components (cameras
    # fixed: rotation records only read, a single point block -- most workgroups of a group own nothing)
    lb = P.load_bal(ncams=49, npts=500)
    cams, pts = P.ba_alternation_plans(lb)
    gl = capi.Problem(gctx, lb)
    def first(comps, k):   # the first k components of a decomposition
        fp, fv, cp, ci = comps
        return (fp[:k + 1], fv[:fp[k]], cp[:k + 1], ci[:cp[k]])
    # (a group needs all its workgroups resident: sixty of the point components)
    for comps, opts in ((cams, {"coop_group_min_factors": 0, "coop_min_factors": 0}), (first(pts, 60), {"row_min_components": 1 << 30, "quad_min_components": 1 << 30})):
        nc = len(comps[0]) - 1
        for K in (1, 2):
            r, x, _, tr, info = run({**base, **opts, "ptm_group": K, "ptm_threads": 256}, comps, gl, lb, 10, trace=2048)
            assert info["components_point_major"] == nc and info["point_major_group"] == K, info
            assert np.all(r.delta <= 0)
            for c in (0, nc // 2, nc - 1):
                fv = comps[1][comps[0][c]:comps[0][c + 1]]
                fc = comps[3][comps[2][c]:comps[2][c + 1]]
                sub = type("R", (), {"status": r.status[c:c + 1], "iters": r.iters[c:c + 1], "fret": r.fret[c:c + 1]})
                check_replay(lb, tr[c], sub, 10, free_vid=fv, fac_id=fc, x=lb.x0[fv], iter_tol=1e-8)


def test_point_major_solver_at_the_bench_shape(gctx):
    """BASELINE config 5, size L -- the shape bench.py's strong-scaling block and its synthetic-L workload time
    (SURVEY 8d): components of 49 cameras x 7776 points x 4 observations = 31104 factors, 23769 variables, through the
    kernels those launches use: cgd_ptm_kernel<768, .> with a workgroup per component (what one GPU runs on 1000 of
    them) and cgd_ptmg_kernel with two workgroups of 512 lanes per component (what a rank of eight runs on its 125).
    Per configuration: one component replayed by the oracle (bit-identical decisions over all 25 iterations, values and
    slopes to rounding), every component's returned value against the oracle's objective at the returned point, the
    variables left assigned, and the same bits from a second run.
    Reference: src/bundleadjust/BundleAdjustmentFactor.cpp:160-185, 351-554; src/optimizers/CGDSubspaceOptimizer.cpp:19-98."""
    ncomp = 4
    syn = P.make_synthetic_ba(ncomp, 49, 7776, obs_per_pt=4)
    csr = (syn.comp_free_ptr, syn.comp_free_vid, syn.comp_fac_ptr, syn.comp_fac_id)
    assert syn.nfac == ncomp * 31104 and syn.nvars == ncomp * 23769
    g = capi.Problem(gctx, syn)
    orc = O.OracleProblem(syn, emulate_stale_cache=False)

    def run(opts, trace=0):
        g.set_x(syn.x0)
        plan = capi.Plan(g, *csr)
        for k, v in opts.items():
            plan.set_option(k, v)
        if trace:
            plan.set_option("trace_records", trace)
            plan.set_option("dump_iters", 25)
        plan.set_start(None)
        plan.solve(25, 3e-8)
        r = plan.fetch()
        tr = (plan.get_trace(0, trace)[0], plan.get_vectors(0, 25)) if trace else None
        info = {k: plan.info(k) for k in ("components_point_major", "point_major_group", "components_cooperative", "components_lds", "components_plain")}
        x = g.get_x()
        plan.close()
        return r, tr, info, x

    base = {"coop_group_min_factors": 0, "coop_min_factors": 0}   # (four components alone would be packed into a cooperative launch)
    for opts, K in (({"ptm_group": 1}, 1), ({"ptm_group": 2, "ptm_threads": 512}, 2)):
        r, tr, info, x = run({**base, **opts}, trace=4096)
        assert info["components_point_major"] == ncomp and info["point_major_group"] == K, info
        assert info["components_cooperative"] == 0 and info["components_lds"] == 0 and info["components_plain"] == 0, info
        assert np.all(r.delta < 0) and np.all((r.status & 0xFF) != 7), (r.status, r.delta)
        assert np.array_equal(x[csr[1]], r.x)
        fv, fc = syn.component(0)
        sub = type("R", (), {"status": r.status[:1], "iters": r.iters[:1], "fret": r.fret[:1]})
        # (far-out bracketing steps: 31104 factors offer more projections next to their pole than the small cases do --
        # observed 2e-6 at one such step, 1.3e-14 / 6e-15 at every ordinary point, no decision differs)
        check_replay(syn, tr, sub, 25, free_vid=fv, fac_id=fc, x=syn.x0[fv], iter_tol=1e-10, far_tol=1e-5)
        for c in range(ncomp):   # value parity at the end of the device's own trajectory, every component
            fv, fc = syn.component(c)
            orc.assign(fv, r.x[csr[0][c]:csr[0][c + 1]])
            fo = orc.eval(fc)
            assert abs(fo - r.fret[c]) <= 1e-12 * abs(fo), (K, c, fo, r.fret[c])
        r2 = run({**base, **opts})[0]
        assert np.array_equal(r.fret, r2.fret) and np.array_equal(r.x, r2.x) and np.array_equal(r.nfeval, r2.nfeval)
    g.close()


@pytest.mark.parametrize("solver", ["lds", "point-major", "point-major x2", "plain"])
def test_batch_solvers_with_active_bounds_partial_blocks_and_rollback(gctx, solver):
    """The edges of CGDSubspaceOptimizer::optimize on every batch solver a bundle-adjustment component can reach --
    the LDS-resident one (default), the point-major streaming one alone and with two workgroups per component, the
    plain one: domains tight enough that the clamp is active during the line searches (the objective is evaluated at
    clamp(p + a xi) while CG keeps the unclamped iterate, .cpp:165-168), camera and point blocks only partly free,
    constants untouched, results inside their domains and some on a bound; and the roll-back (.cpp:66-80): a component
    whose start is already better than anything 2 iterations from a hopeless start reach is returned restored."""
    opts = {"lds": {}, "point-major": {"ptm_stream": 2, "ptm_group": 1}, "point-major x2": {"ptm_stream": 2, "ptm_group": 2, "ptm_threads": 256},
            "plain": {"lds_resident": 0, "ptm_stream": 0}}[solver]
    opts = {"coop_group_min_factors": 0, "coop_min_factors": 0, **opts}
    rng = np.random.default_rng(31)
    pp = P.make_synthetic_ba(6, 4, 300, obs_per_pt=3)
    nv = pp.nvars // 6
    w = np.where(np.arange(pp.nvars) % nv < 36, 0.02, 0.01)      # cameras a little more room than points
    pp.lo = np.maximum(pp.lo, pp.x0 - w * rng.uniform(0.2, 1.0, pp.nvars) * np.maximum(np.abs(pp.x0), 1e-3))
    pp.hi = np.minimum(pp.hi, pp.x0 + w * rng.uniform(0.2, 1.0, pp.nvars) * np.maximum(np.abs(pp.x0), 1e-3))
    # partly free blocks: a fifth of the variables held constant, spread over cameras and points
    const = rng.random(pp.nvars) < 0.2
    fp, fv, cp, ci = [0], [], [0], []
    for c in range(6):
        v, f = pp.component(c)
        v = v[~const[v]]
        fv.extend(v.tolist()); fp.append(len(fv)); ci.extend(f.tolist()); cp.append(len(ci))
    comps = tuple(np.array(a, dtype=np.int64) for a in (fp, fv, cp, ci))
    g = capi.Problem(gctx, pp)
    plan = capi.Plan(g, *comps)
    for k, v in opts.items():
        plan.set_option(k, v)
    plan.set_option("trace_records", 4096)
    plan.set_option("dump_iters", 12)
    plan.set_start(None)
    plan.solve(12, 3e-8)
    r = plan.fetch()
    info = {k: plan.info(k) for k in ("components_lds", "components_point_major", "components_plain", "point_major_group")}
    want = {"lds": "components_lds", "plain": "components_plain"}.get(solver, "components_point_major")
    assert info[want] == 6, info
    if solver == "point-major x2":
        assert info["point_major_group"] == 2, info
    fva = comps[1]
    assert np.all(r.x >= pp.lo[fva]) and np.all(r.x <= pp.hi[fva]) and np.all(r.delta <= 0)
    assert np.any((r.x == pp.lo[fva]) | (r.x == pp.hi[fva]))                        # the clamp was active
    after = g.get_x()
    assert np.array_equal(after[const], pp.x0[const]) and np.array_equal(after[fva], r.x)
    for c in (0, 3, 5):
        v, f = fva[comps[0][c]:comps[0][c + 1]], comps[3][comps[2][c]:comps[2][c + 1]]
        sub = type("R", (), {"status": r.status[c:c + 1], "iters": r.iters[c:c + 1], "fret": r.fret[c:c + 1]})
        check_replay(pp, (plan.get_trace(c, 4096)[0], plan.get_vectors(c, 12)), sub, 12, free_vid=v, fac_id=f, x=pp.x0[v], iter_tol=1e-9)
    plan.close()
    # a NaN objective (an assert in the reference, CGDSubspaceOptimizer.cpp:175): reported as such with the start
    # restored (.cpp:66-80's path), and the launch's other components are none the wiser.  Component 1's first point
    # sits at the origin and so does the camera that sees it: the projection divides 0 by 0.
    q = P.make_synthetic_ba(3, 4, 300, obs_per_pt=3)
    v1, f1 = q.component(1)
    cam, pt = int(q.cam_vid0[f1[0]]), int(q.pt_vid0[f1[0]])
    q.x0[pt:pt + 3] = 0.0
    q.x0[cam + 3:cam + 6] = 0.0
    q.lo[pt:pt + 3] = np.minimum(q.lo[pt:pt + 3], -1.0); q.hi[pt:pt + 3] = np.maximum(q.hi[pt:pt + 3], 1.0)
    q.lo[cam + 3:cam + 6] = np.minimum(q.lo[cam + 3:cam + 6], -1.0); q.hi[cam + 3:cam + 6] = np.maximum(q.hi[cam + 3:cam + 6], 1.0)
    gq = capi.Problem(gctx, q)
    pq = capi.Plan(gq)
    for k, v in opts.items():
        pq.set_option(k, v)
    pq.set_start(q.x0)
    pq.solve(3, 3e-8)
    rq = pq.fetch()
    assert (rq.status[1] & 0xFF) == 5 and (rq.status[1] & capi.STATUS_ROLLED_BACK)
    nv1 = len(v1)
    assert np.array_equal(rq.x[nv1:2 * nv1], q.x0[v1]) and np.array_equal(gq.get_x()[v1], q.x0[v1])
    for c in (0, 2):
        v, f = q.component(c)
        ro = O.OracleProblem(q, emulate_stale_cache=False).cgd(free_vid=v, fac=f, x=q.x0[v], maxiters=3)
        assert (rq.status[c] & 0xFF) == 3 and rq.delta[c] < 0
        assert abs((rq.fret[c] - rq.delta[c]) - ro.finit) <= 1e-12 * ro.finit and abs(rq.fret[c] - ro.fret) <= 0.2 * ro.fret
    pq.close()


@pytest.mark.gpu
def test_fetch_into_the_callers_own_arrays(gctx):
    """a caller that solves a plan again keeps its result arrays (the reference's caller keeps its xval): fetch(out=...) writes the
    same values into them as a fresh fetch returns, and refuses another plan's arrays"""
    pp = P.make_synthetic_ba(40, 3, 40)
    g = capi.Problem(gctx, pp)
    plan = capi.Plan(g)
    plan.set_start(pp.x0)
    plan.solve(25, 3e-8)
    fresh = plan.fetch()
    mine = plan.fetch()
    mine.x[:] = 0.0; mine.fret[:] = 0.0; mine.nfeval[:] = 0
    plan.set_start(pp.x0)
    plan.solve(25, 3e-8)
    again = plan.fetch(out=mine)
    assert again is mine
    assert np.array_equal(mine.x, fresh.x) and np.array_equal(mine.fret, fresh.fret) and np.array_equal(mine.nfeval, fresh.nfeval)
    other = capi.Plan(capi.Problem(gctx, P.make_synthetic_ba(7, 3, 40)))
    with pytest.raises(ValueError):
        other.fetch(out=mine)
