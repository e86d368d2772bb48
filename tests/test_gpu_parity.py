"""End-to-end parity WITHOUT re-synchronisation: the device's parity option == a CPU run, bit for bit.

25 unconverged CG iterations are a chaotic map of the start (tests/test_oracle.py::test_cgd_is_chaotic), so "the same
result as the reference" can only be shown end to end by bit-identical arithmetic.  The replay tests
(tests/test_gpu_solver.py) re-synchronise at every line search; these do not.  Plan option factor_rounding = 1 makes
the device round every product before it is added and add EVERY sum in the reference's order; what then still separates it
from the reference is a closed set of three last-place differences inside one factor's arithmetic, each of which the CPU
oracle has as a named run-time switch (oracle/rdis_oracle.h):

    RO_ARITH_SINCOS_ANGLE        the device's own sine / cosine of the rotation angle (below 1 ulp, like the C library's)
    RO_ARITH_RECIPROCAL          x * (1 / y) for x / y in the unit axis and the perspective divide
    RO_BA_DERIV_ADJOINT_DEVICE   the adjoint sweep for the reference's forward chain (BundleAdjustmentFactor.cpp:351-554)

(A fourth switch, RO_SUM_TOPOLOGY_COOPERATIVE -- the device's sum trees restated entry for entry -- makes the oracle return what
the DEFAULT cooperative path returns: the tests at the end of this file.)

With the three on (OracleProblem.device_parity) the oracle is the reference's algorithm -- CGDSubspaceOptimizer.cpp:19-98,
minimize_nrc.h:410-447, State.h:157-210, the stale factor cache of Variable.cpp:66-76 -- in the device's factor arithmetic,
and the tests below assert fret, x, iterations and the f / df call counts with == after 25 iterations from x0 on BASELINE
config 3 (LDS-resident solver) and config 4 (cooperative solver, plain layout).  With the three off it is the
reference-pinned oracle (83227.604227756252 ...): tests/test_oracle.py::test_device_arithmetic_switches_are_the_only_difference.
"""
import os

import numpy as np
import pytest

from oracle import oracle as O
from rdis_amd import capi, problems as P

pytestmark = pytest.mark.gpu


def _first_difference(dev_trace, orc_trace):
    n = min(len(dev_trace), len(orc_trace))
    for i in range(n):
        if dev_trace[i].tobytes() != orc_trace[i].tobytes():
            return i, dev_trace[i], orc_trace[i]
    return (n, None, None) if len(dev_trace) != len(orc_trace) else None


def _device(gctx, pp, maxiters, stale, trace=1 << 15, opts=None, x=None):
    g = capi.Problem(gctx, pp)
    plan = capi.Plan(g)
    plan.set_option("factor_rounding", 1)
    plan.set_option("emulate_stale_cache", int(stale))
    plan.set_option("trace_records", trace)
    for k, v in (opts or {}).items():
        plan.set_option(k, v)
    plan.set_start(pp.x0 if x is None else x)
    plan.solve(maxiters, 3e-8)
    r = plan.fetch()
    tr, n = plan.get_trace(0, trace)
    return plan, r, tr[:n]


def _assert_equal_runs(gctx, pp, maxiters, stale, solver_key, opts=None, x=None):
    x = pp.x0 if x is None else x
    orc = O.OracleProblem.device_parity(pp, emulate_stale_cache=stale)
    want = orc.cgd(x=x, maxiters=maxiters)
    plan, r, tr = _device(gctx, pp, maxiters, stale, opts=opts, x=x)
    assert plan.info(solver_key) == 1
    diff = None
    if not (r.fret[0] == want.fret):
        # where the two runs part: the oracle's own trace of the same solve (same record format; without the stale cache the
        # device skips evaluations whose value it knows bit for bit, so only runs with the cache line up record by record)
        ot, _ = O.OracleProblem.device_parity(pp, emulate_stale_cache=stale).record(x=x, maxiters=maxiters)
        diff = _first_difference(tr, ot) if stale else "n/a"
    assert r.fret[0] == want.fret, (r.fret[0], want.fret, diff)
    assert r.delta[0] == want.delta
    assert int(r.iters[0]) == want.iters and int(r.status[0]) == want.status
    assert int(r.nfeval[0]) == want.nfeval and int(r.ngeval[0]) == want.ngeval
    assert r.x.tobytes() == want.x.tobytes()
    return r, want


def test_factor_arithmetic_of_the_parity_option_is_the_oracles_bit_for_bit(gctx):
    """every factor of full ladybug, value and twelve partials, at x0 and at three perturbed points: the device's
    reference-rounding instantiation == the oracle with its three switches on"""
    pp = P.load_bal().single_component()
    g = capi.Problem(gctx, pp)
    g.set_factor_rounding(1)
    orc = O.OracleProblem.device_parity(pp)
    rng = np.random.default_rng(5)
    for k in range(4):
        x = pp.x0 if k == 0 else pp.x0 * (1.0 + 1e-3 * rng.standard_normal(pp.nvars))
        g.set_x(x)
        orc.assign(None, x)
        fv, fo = g.eval_each(), orc.eval_each()
        gv, go = g.grad_each_ba(), orc.grad_each_ba()
        assert fv.tobytes() == fo.tobytes(), (k, int(np.sum(fv != fo)), float(np.max(np.abs(fv - fo) / np.abs(fo))))
        assert gv.tobytes() == go.tobytes(), (k, int(np.sum(gv != go)))
    # ... and with the switches off the oracle is the reference's arithmetic: close, not equal
    ref = O.OracleProblem(pp)
    ref.assign(None, x)
    fr = ref.eval_each()
    assert np.any(fr != fv) and np.max(np.abs(fr - fv) / np.abs(fr)) < 1e-9


@pytest.mark.parametrize("stale", [True, False])
def test_config3_end_to_end_equals_the_oracle_with_the_named_switches(gctx, stale):
    """ladybug 5 cameras / 30 points, the LDS-resident solver, 25 iterations from x0, with the reference's factor cache
    (as the reference runs) and without: fret, delta, x, iterations, status, f / df calls =="""
    pp = P.load_bal(ncams=5, npts=30).single_component()
    r, want = _assert_equal_runs(gctx, pp, 25, stale, "components_lds")
    assert want.iters == 24 and want.status == 3      # 25 iterations, "too many iterations in frprmn"


@pytest.mark.parametrize("stale", [True, False])
def test_config4_end_to_end_equals_the_oracle_with_the_named_switches(gctx, stale):
    """full ladybug as one component, the cooperative solver's plain layout (256 workgroups), 25 iterations from x0"""
    pp = P.load_bal().single_component()
    r, want = _assert_equal_runs(gctx, pp, 25, stale, "components_cooperative")
    assert want.iters == 24


@pytest.mark.parametrize("solver", ["lds", "cooperative"])
def test_parity_option_over_perturbed_starts_with_and_without_the_factor_cache(gctx, solver):
    """On the trajectory from x0 the reference's factor cache (Variable.cpp:66-76: a factor keeps its value while its variables
    have moved by less than 1e-12) happens not to change a bit in the device's arithmetic; from starts moved by 1e-12 relative it
    does on about every third (the oracle's two runs differ).  Eight such starts of ladybug 5 / 30, on the LDS-resident solver and
    -- the same component given a cooperative group -- on the cooperative solver's plain layout: every run == the oracle's with
    the same setting of the cache."""
    base = P.load_bal(ncams=5, npts=30).single_component()
    opts = {"coop_group_min_factors": 64} if solver == "cooperative" else None
    key = "components_cooperative" if solver == "cooperative" else "components_lds"
    moved = 0
    for seed in range(8):
        x = base.x0 * (1 + 1e-12 * np.random.default_rng(seed).standard_normal(base.nvars))
        ends = []
        for stale in (True, False):
            pp = P.load_bal(ncams=5, npts=30).single_component()
            r, want = _assert_equal_runs(gctx, pp, 25, stale, key, opts=opts, x=x)
            ends.append(want.fret)
        moved += ends[0] != ends[1]
    assert moved >= 2      # (seeds 0, 1, 5 and 7 when this was written)


def test_parity_option_on_a_cooperative_group_of_several_workgroups(gctx):
    """ladybug's 49 cameras and first 500 points (3100 factors: a group of 13 workgroups), from x0 and from two moved starts"""
    for seed in (None, 1, 2):
        pp = P.load_bal(ncams=49, npts=500).single_component()
        x = pp.x0 if seed is None else pp.x0 * (1 + 1e-12 * np.random.default_rng(seed).standard_normal(pp.nvars))
        _assert_equal_runs(gctx, pp, 25, True, "components_cooperative", x=x)


def test_parity_option_on_a_batch_of_components(gctx):
    """the synthetic decomposition's shape (config 5-S: 3 cameras x 40 points per component), 64 components in one launch of the
    LDS-resident solver with the parity option and the factor cache: every component == its own oracle run"""
    pp = P.make_synthetic_ba(ncomp=64, ncams=3, npts=40)
    g = capi.Problem(gctx, pp)
    plan = capi.Plan(g)
    plan.set_option("factor_rounding", 1)
    plan.set_option("emulate_stale_cache", 1)
    plan.set_start(pp.x0)
    plan.solve(25, 3e-8)
    r = plan.fetch()
    assert plan.info("components_lds") == 64
    fp, fv, cp, ci = pp.comp_free_ptr, pp.comp_free_vid, pp.comp_fac_ptr, pp.comp_fac_id
    for c in range(0, 64, 7):
        orc = O.OracleProblem.device_parity(pp)
        want = orc.cgd(free_vid=fv[fp[c]:fp[c + 1]], fac=ci[cp[c]:cp[c + 1]], x=pp.x0[fv[fp[c]:fp[c + 1]]], maxiters=25)
        assert r.fret[c] == want.fret and int(r.nfeval[c]) == want.nfeval and int(r.iters[c]) == want.iters, c
        assert r.x[fp[c]:fp[c + 1]].tobytes() == want.x.tobytes(), c


# ---- the DEFAULT path: the benchmarked cooperative solvers == the oracle with a fourth switch, the device's sum trees ----------

def _default_path(gctx, pp, maxiters, opts=None, x=None):
    g = capi.Problem(gctx, pp)
    plan = capi.Plan(g)
    for k, v in (opts or {}).items():
        plan.set_option(k, v)
    plan.set_start(pp.x0 if x is None else x)
    plan.solve(maxiters, 3e-8)
    return plan, plan.fetch()


@pytest.mark.parametrize("layout", ["pipelined (the default)", "plain"])
def test_default_cooperative_path_equals_the_oracle_with_the_sum_trees(gctx, layout):
    """BASELINE config 4 as bench.py runs it -- no option set: the pipelined cooperative solver, the reference's rounding, sums as
    trees over lanes, waves and workgroups -- against the oracle with FOUR named switches: the three of the factor arithmetic and
    RO_SUM_TOPOLOGY_COOPERATIVE, the device's trees restated entry for entry (oracle/rdis_oracle.c: a wave of 64 as a balanced
    tree, the waves' sums taken l, l + 64, ... by lane l, the slope factor by factor, gg / dgg by owner lane, a camera variable's
    900 partials strided over a wave).  25 iterations from x0, no re-synchronisation: fret, delta, x, iterations, status and the
    f / df call counts ==.  So the number the driver's line carries (89607.17144518998 after 798 evaluations) is a CPU run's
    number, bit for bit, and what separates it from the reference's 83227.604227756252 is those four switches and nothing else."""
    pp = P.load_bal().single_component()
    want = O.OracleProblem.device_default(pp, lanes_per_workgroup=256 if layout == "plain" else 128).cgd(x=pp.x0, maxiters=25)
    plan, r = _default_path(gctx, pp, 25, {"coop_pipeline": 0} if layout == "plain" else None)
    assert plan.info("components_cooperative") == 1 and plan.info("pipelined") == (0 if layout == "plain" else 1)
    assert r.fret[0] == want.fret and r.delta[0] == want.delta, (r.fret[0], want.fret)
    assert int(r.iters[0]) == want.iters and int(r.status[0]) == want.status
    assert int(r.nfeval[0]) == want.nfeval and int(r.ngeval[0]) == want.ngeval
    assert r.x.tobytes() == want.x.tobytes()
    import json, os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "parity_end_values.json")) as fh:
        w = json.load(fh)["ladybug_full_default_path"]
    assert r.fret[0] == w["fret"] and int(r.nfeval[0]) == w["nfeval"]      # (the committed CPU fixture: what bench.py compares with)


def test_default_cooperative_path_on_other_components_and_starts(gctx):
    """the same on ladybug's 49 cameras with 500 and with 2000 points from x0 and from two starts moved by 1e-12, and with
    speculation off (which must not change a bit).  With 500 points the group has 64 waves for 441 camera variables of some 80
    partials each: the 64 longest runs are strided over a wave, the others added in factor-list order by a lane -- the oracle
    follows the same rule (OracleProblem.set_cooperative_topology); with 2000 points every camera variable has a wave."""
    for npts, seeds in ((500, (None, 1, 2)), (2000, (None,))):
        for seed in seeds:
            pp = P.load_bal(ncams=49, npts=npts).single_component()
            x = pp.x0 if seed is None else pp.x0 * (1 + 1e-12 * np.random.default_rng(seed).standard_normal(pp.nvars))
            for opts in (None, {"coop_speculate": 0}, {"coop_pipeline": 0}):
                plain = bool(opts) and opts.get("coop_pipeline") == 0
                want = O.OracleProblem.device_default(pp, lanes_per_workgroup=256 if plain else 128).cgd(x=x, maxiters=25)
                plan, r = _default_path(gctx, pp, 25, opts, x=x)
                assert plan.info("components_cooperative") == 1 and plan.info("pipelined") == (0 if plain else 1)
                assert r.fret[0] == want.fret and r.x.tobytes() == want.x.tobytes(), (npts, seed, opts, r.fret[0], want.fret)
                assert (int(r.iters[0]), int(r.nfeval[0]), int(r.ngeval[0])) == (want.iters, want.nfeval, want.ngeval)


# ---- the DEFAULT LDS-resident path (configs 3 and 5-S): fused multiply-adds and all ------------------------------------------------

def test_fused_factor_arithmetic_is_reproduced_by_the_host_compile(gctx):
    """The batch solvers' default arithmetic contracts a * b + c where the source has it in one expression.  factors.hpp compiled
    for the HOST by the same front end (tests/cpp/factors_host.hip, -mfma -ffp-contract=on) gives the device's bits: every factor
    of ladybug, value and twelve partials, at three points."""
    import ctypes as C
    L = O.factors_host()[0]
    pp = P.load_bal().single_component()
    g = capi.Problem(gctx, pp)
    rng = np.random.default_rng(1)
    for k in range(3):
        x = pp.x0 if k == 0 else pp.x0 * (1 + 1e-3 * rng.standard_normal(pp.nvars))
        g.set_x(x)
        fd, gd = g.eval_each(), g.grad_each_ba()
        x12 = np.ascontiguousarray(np.concatenate([x[pp.cam_vid0[:, None] + np.arange(9)], x[pp.pt_vid0[:, None] + np.arange(3)]], axis=1))
        f, g12 = np.empty(pp.nfac), np.empty((pp.nfac, 12))
        v = lambda a: a.ctypes.data_as(C.c_void_p)
        L.fh_eval_grad_each(C.c_longlong(pp.nfac), v(x12), v(np.ascontiguousarray(pp.obs)), v(f), v(g12))
        assert f.tobytes() == fd.tobytes() and g12.tobytes() == gd.tobytes(), k


def test_default_lds_path_equals_the_oracle_on_config_3(gctx):
    """BASELINE config 3 as bench.py and smoke() run it -- no option set: the LDS-resident solver, fused multiply-adds, forward-mode
    slope, sums as trees -- against the oracle with the device's own factor arithmetic plugged in (ro_set_factor_arithmetic:
    factors.hpp compiled for the host) and RO_SUM_TOPOLOGY_LDS (that solver's trees restated entry for entry): fret, delta, x,
    iterations, status, call counts == after 25 iterations from x0 and from four moved starts.  25.168503286225235 after 540
    evaluations -- the number smoke() and the bench line's configs block print -- is a CPU run's number."""
    base = P.load_bal(ncams=5, npts=30).single_component()
    for seed in (None, 0, 1, 2, 3):
        pp = P.load_bal(ncams=5, npts=30).single_component()
        x = pp.x0 if seed is None else pp.x0 * (1 + 1e-12 * np.random.default_rng(seed).standard_normal(pp.nvars))
        want = O.OracleProblem.device_lds_default(pp).cgd(x=x, maxiters=25)
        plan, r = _default_path(gctx, pp, 25, x=x)
        assert plan.info("components_lds") == 1
        assert r.fret[0] == want.fret and r.delta[0] == want.delta and r.x.tobytes() == want.x.tobytes(), (seed, r.fret[0], want.fret)
        assert (int(r.iters[0]), int(r.status[0]), int(r.nfeval[0]), int(r.ngeval[0])) == (want.iters, want.status, want.nfeval, want.ngeval)
    assert base.nfac == 121


def test_default_lds_path_equals_the_oracle_on_config_5s(gctx):
    """... and BASELINE config 5-S: 1000 components x (3 cameras, 40 points, 120 observations) in one launch, every 50th component
    against its own oracle run"""
    pp = P.make_synthetic_ba(1000, 3, 40)
    g = capi.Problem(gctx, pp)
    plan = capi.Plan(g)
    plan.set_start(pp.x0)
    plan.solve(25, 3e-8)
    r = plan.fetch()
    assert plan.info("components_lds") == 1000
    fp, fv, cp, ci = pp.comp_free_ptr, pp.comp_free_vid, pp.comp_fac_ptr, pp.comp_fac_id
    for c in range(0, 1000, 50):
        o = O.OracleProblem.device_lds_default(pp, free_vid=fv[fp[c]:fp[c + 1]], fac=ci[cp[c]:cp[c + 1]])
        want = o.cgd(free_vid=fv[fp[c]:fp[c + 1]], fac=ci[cp[c]:cp[c + 1]], x=pp.x0[fv[fp[c]:fp[c + 1]]], maxiters=25)
        assert r.fret[c] == want.fret and r.x[fp[c]:fp[c + 1]].tobytes() == want.x.tobytes(), (c, r.fret[c], want.fret)
        assert (int(r.iters[c]), int(r.status[c]), int(r.nfeval[c]), int(r.ngeval[c])) == (want.iters, want.status, want.nfeval, want.ngeval), c


def test_default_cooperative_groups_side_by_side_equal_the_oracle(gctx):
    """ladybug with the points held constant: 49 camera components (361 .. 906 factors, nine free variables each) in one plan --
    by default each gets a cooperative group (five to eight workgroups, every one of its nine variables wave-owned) and the groups
    run side by side in one launch.  Every seventh component == its own oracle run with the four switches (constants enter the
    factors; their direction entries are zero)."""
    pp = P.load_bal()
    cams, _ = P.ba_alternation_plans(pp)
    g = capi.Problem(gctx, pp)
    plan = capi.Plan(g, *cams)
    plan.set_start(pp.x0[cams[1]])
    plan.solve(25, 3e-8)
    r = plan.fetch()
    assert plan.info("components_cooperative") == 49
    fp, fv, cp, ci = cams
    lanes = 128 if plan.info("pipelined") else 256
    for c in range(0, 49, 7):
        v, f = fv[fp[c]:fp[c + 1]], ci[cp[c]:cp[c + 1]]
        want = O.OracleProblem.device_default(pp, free_vid=v, fac=f, lanes_per_workgroup=lanes).cgd(free_vid=v, fac=f, x=pp.x0[v], maxiters=25)
        assert r.fret[c] == want.fret and r.x[fp[c]:fp[c + 1]].tobytes() == want.x.tobytes(), (c, r.fret[c], want.fret)
        assert (int(r.iters[c]), int(r.status[c]), int(r.nfeval[c]), int(r.ngeval[c])) == (want.iters, want.status, want.nfeval, want.ngeval), c


def _each_component_equals(pp, r, comps, make):
    fp, fv, cp, ci = pp.comp_free_ptr, pp.comp_free_vid, pp.comp_fac_ptr, pp.comp_fac_id
    for c in comps:
        v, f = fv[fp[c]:fp[c + 1]], ci[cp[c]:cp[c + 1]]
        want = make(f).cgd(free_vid=v, fac=f, x=pp.x0[v], maxiters=25)
        assert r.fret[c] == want.fret and r.delta[c] == want.delta and r.x[fp[c]:fp[c + 1]].tobytes() == want.x.tobytes(), (c, r.fret[c], want.fret)
        assert (int(r.iters[c]), int(r.status[c]), int(r.nfeval[c]), int(r.ngeval[c])) == (want.iters, want.status, want.nfeval, want.ngeval), c


def test_default_point_major_path_equals_the_oracle_on_config_5l(gctx):
    """BASELINE config 5-L as bench.py's strong-scaling block runs it on one device -- no option set: more components (49 cameras x
    7776 points x 4 observations = 31104 factors each) than compute units, so each gets ONE workgroup of 768 lanes of the point-major
    streaming solver (solver_ptm.hpp: cameras in LDS, points streamed, trials in matrix form, the gradient's camera sums in rounds)
    -- against the oracle with factors.hpp compiled for the host plugged in (vector form for the gradient, matrix form for the
    trials) and RO_SUM_TOPOLOGY_PTM, that solver's layout and sums restated entry for entry: every 50th of 300 components ==."""
    pp = P.make_synthetic_ba(300, 49, 7776, obs_per_pt=4)
    g = capi.Problem(gctx, pp)
    plan = capi.Plan(g)
    plan.set_start(pp.x0)
    plan.solve(25, 3e-8)
    r = plan.fetch()
    assert plan.info("components_point_major") == 300 and plan.info("point_major_group") == 1
    _each_component_equals(pp, r, range(0, 300, 50), lambda f: O.OracleProblem.device_ptm_default(pp, fac=f))


@pytest.mark.parametrize("group, threads", [(2, 512), (4, 512), (3, 256)])
def test_default_point_major_groups_equal_the_oracle(gctx, group, threads):
    """... and as a rank of eight runs its 125 components: K workgroups share a component (cgd_ptmg_kernel: chunk c is workgroup c
    mod K's, every wave of the group an entry of the exchange, the partial camera gradients added in rank order).  Three components
    in groups of 2 and 4 x 512 lanes and 3 x 256 == the oracle's run with the same group."""
    pp = P.make_synthetic_ba(3, 49, 7776, obs_per_pt=4)
    g = capi.Problem(gctx, pp)
    plan = capi.Plan(g)
    for k, v in {"coop_min_factors": 0, "coop_group_min_factors": 0, "ptm_group": group, "ptm_threads": threads}.items():
        plan.set_option(k, v)
    plan.set_start(pp.x0)
    plan.solve(25, 3e-8)
    r = plan.fetch()
    assert plan.info("components_point_major") == 3 and plan.info("point_major_group") == group and not plan.info("point_major_wide")
    _each_component_equals(pp, r, range(3), lambda f: O.OracleProblem.device_ptm_default(pp, fac=f, threads=threads, group=group))


@pytest.mark.parametrize("case", ["ladybug", "ladybug 16 / 3000", "small and ragged"])
def test_point_major_path_on_real_data_equals_the_oracle(gctx, case):
    """the same solver on the BAL file (points seen by 2 .. 40 cameras: chunks of unequal slot counts, a last chunk of fewer than 64
    blocks, blocks dealt over sixteen runs of the camera-sorted order) and on a component small enough for the LDS sent there by
    option: one workgroup of 768 lanes == the oracle"""
    pp = {"ladybug": lambda: P.load_bal().single_component(), "ladybug 16 / 3000": lambda: P.load_bal(ncams=16, npts=3000).single_component(),
          "small and ragged": lambda: P.load_bal(ncams=7, npts=200).single_component()}[case]()
    g = capi.Problem(gctx, pp)
    plan = capi.Plan(g)
    for k, v in {"coop_min_factors": 0, "coop_group_min_factors": 0, "ptm_stream": 2, "ptm_group": 1}.items():
        plan.set_option(k, v)
    plan.set_start(pp.x0)
    plan.solve(25, 3e-8)
    r = plan.fetch()
    assert plan.info("components_point_major") == 1 and plan.info("point_major_group") == 1
    want = O.OracleProblem.device_ptm_default(pp).cgd(x=pp.x0, maxiters=25)
    assert r.fret[0] == want.fret and r.delta[0] == want.delta and r.x.tobytes() == want.x.tobytes(), (r.fret[0], want.fret)
    assert (int(r.iters[0]), int(r.status[0]), int(r.nfeval[0]), int(r.ngeval[0])) == (want.iters, want.status, want.nfeval, want.ngeval)


@pytest.mark.parametrize("case", ["ladybug as 64 workgroups", "24 cameras x 30000 points"])
def test_wide_point_major_group_equals_the_oracle(gctx, case):
    """ONE component on a large share of the device (solver_ptm.hpp: a wide group -- its chunks dealt to the runs by the hash of
    their position, a workgroup's waves added first and the workgroup one entry of the exchange, the partial camera gradients added
    by shares in rank order): 25 iterations == the oracle's run with that group (RO_SUM_TOPOLOGY_PTM, K < 0)."""
    if case == "ladybug as 64 workgroups":
        pp, opts = P.load_bal().single_component(), {"force_stream": 1, "ptm_group": 64}
    else:
        pp, opts = P.make_synthetic_ba(1, 24, 30000, obs_per_pt=4).single_component(), {}
    g = capi.Problem(gctx, pp)
    plan = capi.Plan(g)
    for k, v in opts.items():
        plan.set_option(k, v)
    plan.set_start(pp.x0)
    plan.solve(25, 3e-8)
    r = plan.fetch()
    assert plan.info("components_point_major") == 1 and plan.info("point_major_wide") == 1 and not plan.info("point_major_local_cameras")
    K, nt = plan.info("point_major_group"), plan.info("point_major_threads")
    assert K > 16 and nt == 512
    want = O.OracleProblem.device_ptm_default(pp, threads=nt, group=K, wide=True).cgd(x=pp.x0, maxiters=25)
    assert r.fret[0] == want.fret and r.delta[0] == want.delta and r.x.tobytes() == want.x.tobytes(), (K, r.fret[0], want.fret)
    assert (int(r.iters[0]), int(r.status[0]), int(r.nfeval[0]), int(r.ngeval[0])) == (want.iters, want.status, want.nfeval, want.ngeval)


def _sinusoid_from_the_committed_start():
    import json
    pp = P.make_high_dim_sinusoid()
    with open(os.path.join(os.path.dirname(__file__), "golden", "sinusoid_start.json")) as fh:
        pp.x0 = np.array(json.load(fh)["x0"])
    return pp.single_component()


@pytest.mark.parametrize("case", ["config 1: testpoly", "config 2: sinusoid, the bench's start", "sinusoid from its own x0", "sinusoid, moved starts"])
def test_default_path_of_the_nonlinear_product_configs_equals_the_oracle(gctx, case):
    """BASELINE configs 1 and 2 as bench.py runs them -- no option set: the plain one-workgroup solver (solver_wg.hpp) on
    nonlinear-product factors -- against the oracle with the device's sine / cosine plugged in (ro_set_trig: factors.hpp's nlp_sin /
    nlp_cos compiled for the host), its third and fourth power (RO_ARITH_POW_SMALL_INT) and that solver's sums
    (RO_SUM_TOPOLOGY_WG): fret, delta, x, iterations, status and call counts ==."""
    if case.startswith("config 1"):
        starts = [P.load_poly().single_component()]
    elif case.startswith("config 2"):
        starts = [_sinusoid_from_the_committed_start()]
    elif case == "sinusoid from its own x0":
        starts = [P.make_high_dim_sinusoid().single_component()]
    else:
        starts = []
        for seed in range(4):
            pp = _sinusoid_from_the_committed_start()
            pp.x0 = pp.x0 * (1 + 1e-9 * np.random.default_rng(seed).standard_normal(pp.nvars))
            starts.append(pp)
    for pp in starts:
        plan, r = _default_path(gctx, pp, 25, x=pp.x0)
        assert plan.info("components_plain") == 1
        want = O.OracleProblem.device_wg_default(pp).cgd(x=pp.x0, maxiters=25)
        assert r.fret[0] == want.fret and r.delta[0] == want.delta and r.x.tobytes() == want.x.tobytes(), (case, r.fret[0], want.fret)
        assert (int(r.iters[0]), int(r.status[0]), int(r.nfeval[0]), int(r.ngeval[0])) == (want.iters, want.status, want.nfeval, want.ngeval)


@pytest.mark.parametrize("case", ["forced on 24 cameras x 30000 points", "300 cameras x 60000 points"])
def test_wide_group_with_local_camera_numbering_equals_the_oracle(gctx, case):
    """... and with LOCAL camera numbering (a component with more cameras than a compute unit's LDS holds: every workgroup owns a
    contiguous range of the chunk order and keeps only the cameras it meets; a camera's partial gradient sums come from the workgroups
    that hold it, in rank order, its terms of gg / dgg from the first of them): == the oracle's run with the same ranges
    (ro_set_ptm_local; the group size and the ranges restated in oracle.py from the data and the device's 256 compute units)."""
    if case.startswith("forced"):
        pp, opts, mit = P.make_synthetic_ba(1, 24, 30000, obs_per_pt=4).single_component(), {"ptm_local_cameras": 1}, 25
    else:
        pp, opts, mit = P.make_synthetic_ba(1, 300, 60000, obs_per_pt=4).single_component(), {}, 10
    g = capi.Problem(gctx, pp)
    plan = capi.Plan(g)
    for k, v in opts.items():
        plan.set_option(k, v)
    plan.set_start(pp.x0)
    plan.solve(mit, 3e-8)
    r = plan.fetch()
    assert plan.info("components_point_major") == 1 and plan.info("point_major_wide") == 1 and plan.info("point_major_local_cameras") > 0
    o = O.OracleProblem.device_ptm_default(pp, local_cus=256)   # (MI355X: 256 compute units)
    assert len(o._wg_chunk0) - 1 == plan.info("point_major_group")
    want = o.cgd(x=pp.x0, maxiters=mit)
    assert r.fret[0] == want.fret and r.delta[0] == want.delta and r.x.tobytes() == want.x.tobytes(), (r.fret[0], want.fret)
    assert (int(r.iters[0]), int(r.status[0]), int(r.nfeval[0]), int(r.ngeval[0])) == (want.iters, want.status, want.nfeval, want.ngeval)


@pytest.mark.parametrize("case", ["ladybug's 7776 points, sixteen lanes each", "20000 points, four lanes each"])
def test_tiny_component_solver_equals_the_oracle(gctx, case):
    """every point against constant cameras is a component of its own (three free variables, a few factors: by count what RDIS asks
    the subspace solver for most) -- thousands of them a launch on solver_quad.hpp, 16 or 4 lanes a component.  Samples of the launch
    == the oracle's run with the device's factor arithmetic and that solver's sums (RO_SUM_TOPOLOGY_GROUP)."""
    if case.startswith("ladybug"):
        pp, lanes = P.load_bal(), 16
    else:
        pp, lanes = P.make_synthetic_ba(1, 8, 20000, obs_per_pt=3), 4
    _, pts = P.ba_alternation_plans(pp)
    fp, fv, cp, ci = pts
    g = capi.Problem(gctx, pp)
    plan = capi.Plan(g, *pts)
    plan.set_start(pp.x0[fv])
    plan.solve(25, 3e-8)
    r = plan.fetch()
    ncomp = len(fp) - 1
    assert plan.info("components_tiny") == ncomp
    o = O.OracleProblem.device_group_default(pp, lanes=lanes)
    for c in range(0, ncomp, max(1, ncomp // 40)):
        v, f = fv[fp[c]:fp[c + 1]], ci[cp[c]:cp[c + 1]]
        want = o.cgd(free_vid=v, fac=f, x=pp.x0[v], maxiters=25)
        o.assign(v, pp.x0[v])   # (the oracle leaves the component assigned at its end point: back to the start, as on the device's other components)
        assert r.fret[c] == want.fret and r.delta[c] == want.delta and r.x[fp[c]:fp[c + 1]].tobytes() == want.x.tobytes(), (c, r.fret[c], want.fret)
        assert (int(r.iters[c]), int(r.status[c]), int(r.nfeval[c]), int(r.ngeval[c])) == (want.iters, want.status, want.nfeval, want.ngeval), c


@pytest.mark.parametrize("ncams, npts, which", [(5, 30, "cameras"), (5, 30, "points"), (12, 300, "points"), (49, 1000, "points")])
def test_lds_path_with_constants_equals_the_oracle(gctx, ncams, npts, which):
    """the alternation plans of small problems -- cameras against fixed points, points against fixed cameras: components with
    constants, a launch of the LDS-resident solver whose cameras' rotations are records (points) or follow the trial point (cameras)
    -- == the oracle's LDS topology with the constants' slots in place (they enter the factors, their terms are skipped)."""
    pp = P.load_bal(ncams=ncams, npts=npts)
    dec = P.ba_alternation_plans(pp)[0 if which == "cameras" else 1]
    fp, fv, cp, ci = dec
    g = capi.Problem(gctx, pp)
    plan = capi.Plan(g, *dec)
    plan.set_start(pp.x0[fv])
    plan.solve(25, 3e-8)
    r = plan.fetch()
    ncomp = len(fp) - 1
    assert plan.info("components_lds") == ncomp
    mf = int(np.diff(cp).max())
    threads = 64 if mf <= 64 else 128 if mf <= 128 else 256
    for c in range(0, ncomp, max(1, ncomp // 12)):
        v, f = fv[fp[c]:fp[c + 1]], ci[cp[c]:cp[c + 1]]
        want = O.OracleProblem.device_lds_default(pp, free_vid=v, fac=f, threads=threads).cgd(free_vid=v, fac=f, x=pp.x0[v], maxiters=25)
        assert r.fret[c] == want.fret and r.delta[c] == want.delta and r.x[fp[c]:fp[c + 1]].tobytes() == want.x.tobytes(), (c, r.fret[c], want.fret)
        assert (int(r.iters[c]), int(r.status[c]), int(r.nfeval[c]), int(r.ngeval[c])) == (want.iters, want.status, want.nfeval, want.ngeval), c


@pytest.mark.parametrize("ncams, npts", [(5, 30), (12, 300)])
def test_bundle_adjustment_on_the_plain_solver_equals_the_oracle(gctx, ncams, npts):
    """the fallback of components no other solver takes (solver_wg.hpp: state in global memory, one workgroup a component), reached
    here by switching the LDS-resident, streaming and cooperative solvers off: == the oracle's RO_SUM_TOPOLOGY_WG run with the
    batch solvers' factor arithmetic"""
    pp = P.load_bal(ncams=ncams, npts=npts).single_component()
    g = capi.Problem(gctx, pp)
    plan = capi.Plan(g)
    for k, v in {"lds_resident": 0, "ptm_stream": 0, "coop_min_factors": 0, "coop_group_min_factors": 0}.items():
        plan.set_option(k, v)
    plan.set_start(pp.x0)
    plan.solve(25, 3e-8)
    r = plan.fetch()
    assert plan.info("components_plain") == 1
    want = O.OracleProblem.device_wg_default(pp).cgd(x=pp.x0, maxiters=25)
    assert r.fret[0] == want.fret and r.delta[0] == want.delta and r.x.tobytes() == want.x.tobytes(), (r.fret[0], want.fret)
    assert (int(r.iters[0]), int(r.status[0]), int(r.nfeval[0]), int(r.ngeval[0])) == (want.iters, want.status, want.nfeval, want.ngeval)


@pytest.mark.parametrize("case", ["ladybug 49 / 500", "ladybug", "the sinusoid"])
def test_grid_solver_equals_the_oracle(gctx, case):
    """the grid solver (solver_stream.hpp: several workgroups of 512 lanes on ONE component, state in global memory -- what takes a
    component the cooperative and the point-major solvers do not) == the oracle's RO_SUM_TOPOLOGY_WG run over the grid's lanes
    (ro_set_stream_topology: every wave an entry of the exchange), bundle adjustment and nonlinear products"""
    if case == "ladybug 49 / 500":
        pp, opts = P.load_bal(ncams=49, npts=500).single_component(), {"coop_min_factors": 1000, "force_stream": 1, "ptm_stream": 0}
    elif case == "ladybug":
        pp, opts = P.load_bal().single_component(), {"force_stream": 1, "ptm_stream": 0}
    else:
        pp, opts = P.make_high_dim_sinusoid().single_component(), {"coop_min_factors": 100}
        pp.x0 = np.random.default_rng(4).uniform(-6, 6, 121)
    g = capi.Problem(gctx, pp)
    plan = capi.Plan(g)
    for k, v in opts.items():
        plan.set_option(k, v)
    plan.set_start(pp.x0)
    plan.solve(25, 3e-8)
    r = plan.fetch()
    assert plan.info("components_grid_stream") == 1
    nwg = plan.info("grid_stream_workgroups")
    assert nwg >= 1
    want = O.OracleProblem.device_wg_default(pp, grid_workgroups=nwg).cgd(x=pp.x0, maxiters=25)
    assert r.fret[0] == want.fret and r.delta[0] == want.delta and r.x.tobytes() == want.x.tobytes(), (nwg, r.fret[0], want.fret)
    assert (int(r.iters[0]), int(r.status[0]), int(r.nfeval[0]), int(r.ngeval[0])) == (want.iters, want.status, want.nfeval, want.ngeval)


@pytest.mark.parametrize("case", ["ladybug", "a scattered sub-list", "one chunk", "2.2e6 factors: tiles of two chunks"])
def test_public_evaluation_entry_points_equal_the_oracle(gctx, case):
    """rdis_hip_eval and rdis_hip_eval_grad on bundle adjustment -- the batched factor / gradient evaluation of the boundary -- return,
    bit for bit, what the oracle returns with the device's factor arithmetic and those kernels' sums restated (ro_eval_device_ba,
    ro_eval_grad_device_ba: chunks of 512 entries, waves as trees; a point block's partials chunk by chunk, a camera's tile by tile,
    each one after the other in list order)."""
    fac = None
    if case == "ladybug":
        pp = P.load_bal()
    elif case == "a scattered sub-list":
        pp = P.load_bal()
        fac = np.random.default_rng(3).permutation(pp.nfac)[:5000].astype(np.int64)
    elif case == "one chunk":
        pp = P.load_bal()
        fac = np.random.default_rng(4).permutation(pp.nfac)[:300].astype(np.int64)
    else:
        pp = P.make_synthetic_ba(70, 49, 7776, obs_per_pt=4)
    g = capi.Problem(gctx, pp)
    x = pp.x0 * (1 + 1e-3 * np.random.default_rng(5).standard_normal(pp.nvars))
    x = np.minimum(np.maximum(x, pp.lo), pp.hi)
    g.set_x(x)
    o = O.OracleProblem.device_eval(pp)
    o.assign(None, x)
    fd = g.eval(fac)
    fg, gd = g.eval_grad(fac)
    fo, go = o.eval_grad_device(fac)
    assert fd == fg == fo == o.eval_device(fac), (fd, fg, fo)
    assert gd.tobytes() == go.tobytes() or np.array_equal(gd, go), float(np.max(np.abs(gd - go)))


@pytest.mark.parametrize("which", ["testpoly", "the sinusoid"])
def test_public_evaluation_entry_points_on_nonlinear_products_equal_the_oracle(gctx, which):
    """... and on the nonlinear-product functions (configs 1 and 2): value and gradient == the oracle with the device's sine / cosine
    and small powers -- the value by ro_eval_device_grid (a grid of 256-lane workgroups striding over the list), the gradient every
    variable's partials in list order, which is the reference's order"""
    pp = P.load_poly() if which == "testpoly" else P.make_high_dim_sinusoid()
    g = capi.Problem(gctx, pp)
    x = np.minimum(np.maximum(pp.x0 + 0.37 * np.random.default_rng(6).standard_normal(pp.nvars), pp.lo), pp.hi)
    g.set_x(x)
    o = O.OracleProblem.device_wg_default(pp.single_component())
    o.assign(None, x)
    fd = g.eval()
    fg, gd = g.eval_grad()
    assert fd == fg == o.eval_device_grid(), (fd, fg, o.eval_device_grid())
    assert np.array_equal(gd, o.gradient()), float(np.max(np.abs(gd - o.gradient())))
