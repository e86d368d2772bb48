"""The dispatcher under random component mixes: ONE plan per seed holds components of 2 ... 70 000 factors -- whole
blocks, blocks with constants (cameras fixed: a point each; points fixed: a camera each; partly free blocks), tight
bounds, empty factor lists -- and rdis_hip_plan_solve's thresholds (default options) send each to one of its solvers.
Whatever it picks, the contract of CGDSubspaceOptimizer::optimize (reference src/optimizers/CGDSubspaceOptimizer.cpp:19-98,
SURVEY.md 8b) holds per component: the returned value is the oracle's objective of the component's factors at the
returned point, deltaFval <= 0, the values are clamped and the variables are left assigned to them, constants are
untouched, a second solve gives the same bits, and sampled components of every solver kind replay against the oracle
with bit-identical decisions.  Over the seeds every solver kind is reached (rdis_hip_plan_get_info)."""
import numpy as np
import pytest

from oracle import oracle as O
from rdis_amd import capi, problems as P
from test_gpu_solver import check_replay

pytestmark = pytest.mark.gpu

KINDS = ("components_cooperative", "components_grid_stream", "components_tiny", "components_lds", "components_point_major", "components_plain")
SEEN = {k: 0 for k in KINDS}
TOTAL = {"components": 0, "seeds": 0}


def _concat(blocks):
    """independent bundle-adjustment blocks side by side in one function (ids offset block by block)"""
    voff = np.cumsum([0] + [b.nvars for b in blocks])
    pp = P.PackedProblem(kind=P.KIND_BA, x0=np.concatenate([b.x0 for b in blocks]), lo=np.concatenate([b.lo for b in blocks]),
                         hi=np.concatenate([b.hi for b in blocks]),
                         cam_vid0=np.concatenate([b.cam_vid0 + voff[i] for i, b in enumerate(blocks)]),
                         pt_vid0=np.concatenate([b.pt_vid0 + voff[i] for i, b in enumerate(blocks)]),
                         obs=np.concatenate([b.obs for b in blocks]))
    return pp, voff, np.cumsum([0] + [b.nfac for b in blocks])


def _mix(seed):
    """blocks (cameras, points, observations per point, how its components are cut) of one seed"""
    rng = np.random.default_rng(seed)
    spec = []
    for _ in range(14):                                                   # small and medium blocks, solved whole
        spec.append((int(rng.integers(2, 7)), int(rng.integers(1, 220)), int(rng.integers(2, 5)), "whole"))
    for _ in range(4):
        spec.append((int(rng.integers(3, 9)), int(rng.integers(40, 500)), int(rng.integers(2, 5)), "partial"))
    spec.append((int(rng.integers(4, 9)), int(rng.integers(100, 400)), 3, "cameras"))                     # a camera each (points constant)
    spec.append((int(rng.integers(3, 7)), int(rng.integers(8, 60)), 2, "points"))                          # a point each: a few tiny ones
    spec.append((int(rng.integers(8, 13)), int(rng.integers(1150, 1300)), 3, "whole"))                     # too large for the LDS, below the cooperative threshold
    spec.append((int(rng.integers(8, 13)), int(rng.integers(1150, 1300)), 3, "partial"))
    spec.append((int(rng.integers(12, 30)), int(rng.integers(1500, 9000)), int(rng.integers(3, 5)), "whole"))   # 4500 ... 36 000 factors
    if seed % 5 == 0:
        # 70 000 factors: beyond the register-resident cooperative solver -- with 30 cameras (they fit the LDS) it streams through the
        # point-major solver since round 6; with 150 it still takes the grid solver (solver_stream.hpp)
        spec.append((30 if seed == 0 else 150, 17500, 4, "whole"))
    if seed % 5 in (1, 3):
        spec.append((6, 4400 + int(rng.integers(0, 300)), 3, "points"))   # thousands of tiny components: the group solver
    spec.append((2, 3, 2, "empty"))                                      # variables nobody lists a factor for
    return rng, spec


def _build(seed):
    rng, spec = _mix(seed)
    blocks = [P.make_synthetic_ba(1, C, Pn, obs_per_pt=K, first_comp=1000 * seed + i) for i, (C, Pn, K, _) in enumerate(spec)]
    pp, voff, foff = _concat(blocks)
    free_ptr, free_vid, fac_ptr, fac_id, expect = [0], [], [0], [], []

    def add(fv, fc):
        free_vid.append(np.asarray(fv, dtype=np.int64)); fac_id.append(np.asarray(fc, dtype=np.int64))
        free_ptr.append(free_ptr[-1] + len(fv)); fac_ptr.append(fac_ptr[-1] + len(fc))
    for i, (b, (C, Pn, K, mode)) in enumerate(zip(blocks, spec)):
        v0, f0 = voff[i], foff[i]
        allv, allf = v0 + np.arange(b.nvars), f0 + np.arange(b.nfac)
        if mode == "whole":
            add(allv, allf)
        elif mode == "partial":
            # two of a camera's nine held constant for some cameras, one coordinate of some points, some points altogether
            keep = np.ones(b.nvars, bool)
            for c in range(C):
                if rng.random() < 0.5:
                    keep[9 * c + rng.choice(9, 2, replace=False)] = False
            pts = 9 * C + 3 * np.arange(Pn)
            keep[pts[rng.random(Pn) < 0.2] + rng.integers(0, 3)] = False
            whole = pts[rng.random(Pn) < 0.1]
            for k in range(3):
                keep[whole + k] = False
            add(allv[keep], allf)
        elif mode == "cameras":
            for c in range(C):
                add(v0 + 9 * c + np.arange(9), f0 + np.nonzero(b.cam_vid0 == 9 * c)[0])
        elif mode == "points":
            order = np.argsort(b.pt_vid0, kind="stable")
            cnt = np.bincount((b.pt_vid0 - 9 * C) // 3, minlength=Pn)
            ptr = np.concatenate([[0], np.cumsum(cnt)])
            for q in range(Pn):
                add(v0 + 9 * C + 3 * q + np.arange(3), f0 + np.sort(order[ptr[q]:ptr[q + 1]]))
        else:
            add(v0 + 9 * C + np.arange(3), np.zeros(0, np.int64))        # a point's variables, no factor: returns 0, touches nothing
            add(v0 + np.arange(4), np.zeros(0, np.int64))
    fv = np.concatenate(free_vid)
    # tight bounds around the start on a twentieth of the free variables: the clamp becomes active
    tight = fv[rng.random(len(fv)) < 0.05]
    w = 1e-4 * (np.abs(pp.x0[tight]) + 1e-3)
    pp.lo[tight], pp.hi[tight] = pp.x0[tight] - w * rng.random(len(tight)), pp.x0[tight] + w * rng.random(len(tight))
    csr = (np.array(free_ptr, np.int64), fv, np.array(fac_ptr, np.int64), np.concatenate(fac_id))
    return pp, csr


@pytest.mark.parametrize("seed", [0, 1, 2, 3, 4, 5])
def test_random_component_mixes_in_one_plan(seed, gctx):
    pp, csr = _build(seed)
    fp, fv, cp, ci = csr
    ncomp = len(fp) - 1
    g = capi.Problem(gctx, pp)
    plan = capi.Plan(g, *csr)
    plan.set_option("trace_records", 2048)
    plan.set_option("dump_iters", 25)
    info = {k: plan.info(k) for k in KINDS}
    assert sum(info.values()) == ncomp, info
    for k in KINDS:
        SEEN[k] += info[k]
    TOTAL["components"] += ncomp; TOTAL["seeds"] += 1
    runs = []
    for _ in range(2):
        g.set_x(pp.x0)
        plan.set_start(pp.x0[fv])
        plan.solve(25, 3e-8)
        runs.append((plan.fetch(), g.get_x()))
    (r, after), (r2, after2) = runs
    nfac = np.diff(cp)
    print("seed %d: %d components (%s), factors per component %d ... %d; exits %s" % (
        seed, ncomp, ", ".join("%s %d" % (k[11:], v) for k, v in info.items() if v), nfac.min(), nfac.max(),
        {capi.EXIT_NAMES[int(k)]: int(v) for k, v in zip(*np.unique(r.status & 0xFF, return_counts=True))}))
    # the same bits twice
    for a, b in ((r.fret, r2.fret), (r.x, r2.x), (r.delta, r2.delta), (r.iters, r2.iters), (r.status, r2.status), (r.nfeval, r2.nfeval), (after, after2)):
        assert np.array_equal(a, b)
    # no exchange gave up, no NaN; an empty factor list returns 0 and touches nothing (CGDSubspaceOptimizer.cpp:26-29)
    code = r.status & 0xFF
    assert not np.any(code == 7) and not np.any(code == 5)
    empty = nfac == 0
    assert np.all(code[empty] == 6) and np.all(r.fret[empty] == 0.0) and np.all(r.delta[empty] == 0.0) and np.all(code[~empty] != 6)
    # deltaFval <= 0 (a restored start: exactly the start's value again)
    assert np.all(r.delta <= 0.0)
    # clamped, variables left assigned to the returned values, constants untouched
    assert np.all(r.x >= pp.lo[fv]) and np.all(r.x <= pp.hi[fv])
    const = np.setdiff1d(np.arange(pp.nvars), fv)
    assert np.array_equal(after[fv], r.x) and np.array_equal(after[const], pp.x0[const])
    assert np.any((r.x == pp.lo[fv]) | (r.x == pp.hi[fv]))                 # some tight bound is active
    # every component's value is the oracle's objective of its factors at the returned point, and its delta the descent from the start
    o = O.OracleProblem(pp, emulate_stale_cache=False)
    e0 = o.eval_each()
    o.assign(np.arange(pp.nvars, dtype=np.int64), after)
    e1 = o.eval_each()
    seg = cp[:-1][~empty]
    f1 = np.zeros(ncomp); f0 = np.zeros(ncomp); scale = np.zeros(ncomp)
    f1[~empty] = np.add.reduceat(e1[ci], seg); f0[~empty] = np.add.reduceat(e0[ci], seg); scale[~empty] = np.add.reduceat(np.abs(e1[ci]) + np.abs(e0[ci]), seg)
    assert np.all(np.abs(r.fret - f1) <= 1e-12 * scale), np.max(np.abs(r.fret - f1) / np.maximum(scale, 1e-300))
    assert np.all(np.abs(r.delta - (f1 - f0)) <= 1e-12 * scale)
    # sampled components replayed: the smallest, the largest, and one at every decile of the sizes in between
    order = np.argsort(nfac, kind="stable")
    order = order[nfac[order] > 0]
    picks = sorted(set(int(order[int(q * (len(order) - 1))]) for q in (0.0, 0.3, 0.6, 0.8, 0.9, 0.95, 0.98, 1.0)))
    for c in picks:
        v, f = fv[fp[c]:fp[c + 1]], ci[cp[c]:cp[c + 1]]
        tr = plan.get_trace(c, 2048)[0]
        if len(tr) >= 2048:
            continue                                                       # (a trace that filled its buffer cannot be replayed to the end)
        sub = type("R", (), {"status": r.status[c:c + 1], "iters": r.iters[c:c + 1], "fret": r.fret[c:c + 1]})
        # (components of two or three well-fitted factors: a value is what is left of pixel-scale numbers cancelling, so
        # "1e-12 of the sum of the factor values" is a few units of the rounding bound -- observed 1.06e-12 at 5.4 units)
        # (a component that leaves by the gradient test has a gradient that all but vanishes: its sums gg, dgg are rounding
        # noise to a relative 1e-4 -- seed 5 has one -- while the decisions they lead to are still the oracle's, bit for bit)
        # (... and so is its direction: xi is that gradient)
        # (drift of p / xi over one iteration between re-syncs: Brent's tolerance is 3e-8 of the step; observed 1.1e-8)
        itol, vtol = (1e-3, 1e-3) if (r.status[c] & 0xFF) in (1, 2) else (1e-8, 1e-7)
        check_replay(pp, (tr, plan.get_vectors(c, 25)), sub, 25, free_vid=v, fac_id=f, x=pp.x0[v], iter_tol=itol, far_tol=1e-2, near_scale=10.0, vec_tol=vtol)
        # (far_tol: a far-out bracketing step that lands next to a projection's pole -- P_z what is left of O(1) terms
        # cancelling -- is ill-conditioned in the factor arithmetic itself; seed 5's largest component has one where two
        # correct fp64 evaluations part by 2.2e-4, 11 units of the rounding bound; the decisions stay bit-identical)
    plan.close()
    g.close()


def test_every_solver_kind_was_reached():
    """(after the seeds above) the mixes reached every solver the dispatcher has"""
    if TOTAL["seeds"] < 6:
        pytest.skip("the seeds did not all run")
    assert TOTAL["components"] >= 200
    assert all(v > 0 for v in SEEN.values()), SEEN
