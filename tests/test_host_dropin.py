"""The C++ plugin surface (rdis_amd/host): rdis::HipCGDSubspaceOptimizer used the way
the reference's callers use CGDSubspaceOptimizer, through tests/cpp/harness.cpp.

CPU part: the loaders build exactly the packed function the Python builders build
(ids, domains, factor order) -- index work is bit-exact.  GPU part: optimize() over
all variables (BCD one-block shape), over one camera+point block with constants
(RDIS getValueFromDomain shape), on the polynomial file, and sibling components in
one launch."""
import ctypes as C
import gzip
import os
import shutil
import subprocess

import numpy as np
import pytest

from oracle import oracle as O
from rdis_amd import capi, problems as P

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_ip = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")


@pytest.fixture(scope="module")
def harness():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "rdis_amd", "host")], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "cpp")], stdout=subprocess.DEVNULL)
    return C.CDLL(os.path.join(ROOT, "tests", "cpp", "libharness.so"))


@pytest.fixture(scope="module")
def bal_path(tmp_path_factory):
    dst = tmp_path_factory.mktemp("bal") / "ladybug.txt"
    with gzip.open(P.LADYBUG_PATH, "rb") as src, open(dst, "wb") as out:
        shutil.copyfileobj(src, out)
    return str(dst).encode()


def _pack_bal(h, path, nc, npnt):
    sizes = np.zeros(3, dtype=np.int64)
    assert h.harness_pack_bal(path, nc, npnt, sizes.ctypes.data_as(C.c_void_p), None, None, None, None, None, None) == 0
    n, f = int(sizes[0]), int(sizes[1])
    x, lo, hi = np.empty(n), np.empty(n), np.empty(n)
    cam, pt, obs = np.empty(f, np.int64), np.empty(f, np.int64), np.empty(2 * f)
    v = lambda a: a.ctypes.data_as(C.c_void_p)
    assert h.harness_pack_bal(path, nc, npnt, v(sizes), v(x), v(lo), v(hi), v(cam), v(pt), v(obs)) == 0
    return sizes, x, lo, hi, cam, pt, obs


@pytest.mark.parametrize("nc,npnt", [(5, 30), (49, 500), (0, 0)])
def test_bal_loader_matches_python_builder(harness, bal_path, nc, npnt):
    sizes, x, lo, hi, cam, pt, obs = _pack_bal(harness, bal_path, nc, npnt)
    pp = P.load_bal(ncams=nc, npts=npnt)
    assert (sizes[0], sizes[1], sizes[2]) == (pp.nvars, pp.nfac, 0)
    assert np.array_equal(cam, pp.cam_vid0) and np.array_equal(pt, pp.pt_vid0)     # indexing: bit-exact
    assert np.array_equal(x, pp.x0) and np.array_equal(obs, pp.obs.reshape(-1))
    assert np.array_equal(lo, pp.lo) and np.array_equal(hi, pp.hi)


def test_bal_save_round_trip(harness, bal_path, tmp_path):
    assert harness.harness_bal_round_trip(bal_path, 5, 30, str(tmp_path / "out.txt").encode()) == 0
    q = P.load_bal(str(tmp_path / "out.txt"))           # and the Python loader reads what the C++ saver wrote
    pp = P.load_bal(ncams=5, npts=30)
    assert np.array_equal(q.x0, pp.x0) and np.array_equal(q.obs, pp.obs) and np.array_equal(q.cam_vid0, pp.cam_vid0)


@pytest.mark.parametrize("which", [0, 1])
def test_nlp_builders_match_python(harness, which):
    pp = P.load_poly() if which == 0 else P.make_high_dim_sinusoid()
    sizes = np.zeros(3, dtype=np.int64)
    v = lambda a: a.ctypes.data_as(C.c_void_p)
    path = P.TESTPOLY_PATH.encode()
    assert harness.harness_pack_nlp(which, path, v(sizes), None, None, None, None, None, None, None, None) == 0
    n, f, nnz = (int(s) for s in sizes)
    assert (n, f, nnz) == (pp.nvars, pp.nfac, len(pp.vid))
    lo, hi, coeff = np.empty(n), np.empty(n), np.empty(f)
    rowptr, vid = np.empty(f + 1, np.int64), np.empty(nnz, np.int64)
    expo, cons, sine = np.empty(nnz), np.empty(nnz), np.empty(nnz, np.uint8)
    assert harness.harness_pack_nlp(which, path, v(sizes), v(lo), v(hi), v(coeff), v(rowptr), v(vid), v(expo), v(cons), v(sine)) == 0
    assert np.array_equal(lo, pp.lo) and np.array_equal(hi, pp.hi) and np.array_equal(coeff, pp.coeff)
    assert np.array_equal(rowptr, pp.rowptr) and np.array_equal(vid, pp.vid)
    assert np.array_equal(expo, pp.expo) and np.array_equal(cons, pp.cons) and np.array_equal(sine, pp.sine)


def _ba_cgd(h, path, nc, npnt, mode, maxit):
    out = np.zeros(9)
    nfree = C.c_longlong()
    x = np.zeros(9 * 49 + 3 * 7776)
    rc = h.harness_ba_cgd(path, nc, npnt, mode, maxit, C.c_double(3e-8), out.ctypes.data_as(C.c_void_p),
                          x.ctypes.data_as(C.c_void_p), C.byref(nfree))
    assert rc == 0
    return out, x[:nfree.value]


@pytest.mark.gpu
def test_optimize_all_variables_like_bcd(harness, bal_path, gctx):
    from rdis_amd import capi
    out, x = _ba_cgd(harness, bal_path, 5, 30, 0, 25)
    fret, delta, before, after, iters, status, nfe, nge, ok = out
    pp = P.load_bal(ncams=5, npts=30).single_component()
    o = O.OracleProblem(pp)
    assert ok == 1.0 and abs(before - o.eval()) <= 1e-12 * before           # f(x_init)
    assert abs((fret - delta) - before) <= 1e-12 * before                  # deltaFval = f_end - f_init
    assert abs(after - fret) <= 1e-12 * fret                               # vars left assigned to the result
    # identical to driving the C ABI directly (same kernels, same inputs): bit-exact
    plan = capi.Plan(capi.Problem(gctx, pp))
    plan.set_start(pp.x0)
    plan.solve(25, 3e-8)
    r = plan.fetch()
    assert fret == r.fret[0] and np.array_equal(x, r.x) and iters == r.iters[0] and nfe == r.nfeval[0]
    o.assign(None, x)
    assert abs(o.eval() - fret) <= 1e-12 * fret                            # oracle's objective at the returned point


@pytest.mark.gpu
def test_optimize_block_with_constants_like_rdis(harness, bal_path):
    out, x = _ba_cgd(harness, bal_path, 5, 30, 1, 25)
    fret, delta, before, after, iters, status, nfe, nge, ok = out
    assert ok == 1.0 and len(x) == 12 and delta < 0
    pp = P.load_bal(ncams=5, npts=30)
    free = np.concatenate([np.arange(9), np.arange(45, 48)]).astype(np.int64)
    fac = np.where((pp.cam_vid0 == 0) | (pp.pt_vid0 == 45))[0].astype(np.int64)
    o = O.OracleProblem(pp)
    assert abs(before - o.eval(fac)) <= 1e-12 * before
    o.assign(free, x)
    assert abs(o.eval(fac) - fret) <= 1e-12 * fret and abs(after - fret) <= 1e-12 * fret
    r1 = O.OracleProblem(pp).cgd(free_vid=free, fac=fac, x=pp.x0[free], maxiters=1)
    out1, _ = _ba_cgd(harness, bal_path, 5, 30, 1, 1)
    assert abs(out1[0] - r1.fret) <= 1e-6 * r1.fret                        # one line minimisation: Brent's tolerance


@pytest.mark.gpu
def test_optimize_full_ladybug_through_plugin(harness, bal_path, golden, gctx):
    out, x = _ba_cgd(harness, bal_path, 0, 0, 0, 25)
    assert out[8] == 1.0 and int(out[5]) & 0xFF == 3 and out[4] == 24
    # the plug-in adds nothing of its own: same bits as the C ABI called directly (whose end value is
    # checked against the oracle's distribution in test_gpu_solver.py)
    pp = P.load_bal()
    fv, fc = np.arange(pp.nvars, dtype=np.int64), np.arange(pp.nfac, dtype=np.int64)
    r = capi.Problem(gctx, pp).cgd_batch(np.array([0, len(fv)]), fv, np.array([0, len(fc)]), fc, pp.x0, 25, 3e-8)
    assert out[0] == r.fret[0] and np.array_equal(x, r.x)
    assert abs(out[2] - 850912.46068083902) <= 1e-12 * out[2]
    assert abs(out[3] - out[0]) <= 1e-12 * out[0]


@pytest.mark.gpu
def test_polynomial_through_plugin(harness, golden):
    t = golden["testpoly"]
    for case in t["cgd"]:
        out = np.zeros(6)
        assert harness.harness_poly_cgd(P.TESTPOLY_PATH.encode(), C.c_double(case["start"][0]), C.c_double(case["start"][1]),
                                        50, out.ctypes.data_as(C.c_void_p)) == 0
        assert abs(out[0] - case["fret"]) <= 1e-10 * abs(case["fret"])
        if "x" in case:
            assert abs(out[2] - case["x"][0]) <= 1e-5 and abs(out[3] - case["x"][1]) <= 1e-5


@pytest.mark.gpu
def test_sibling_components_in_one_launch(harness, bal_path):
    # cameras fixed => every point is an independent component
    out = np.zeros(6)
    assert harness.harness_ba_points_batch(bal_path, 49, 500, 25, out.ctypes.data_as(C.c_void_p)) == 0
    total, sum_delta, ncomp, its, before, after = out
    assert ncomp == 500 and its >= 500 and sum_delta < 0
    assert abs(total - after) <= 1e-12 * after                              # sum of component values = objective
    assert abs((before + sum_delta) - after) <= 1e-10 * before
    # the same components one at a time on the oracle: converged 3-variable problems agree
    pp = P.load_bal(ncams=49, npts=500)
    o = O.OracleProblem(pp)
    tot = 0.0
    for p in range(500):
        fv = np.arange(441 + 3 * p, 441 + 3 * p + 3, dtype=np.int64)
        fc = np.where(pp.pt_vid0 == 441 + 3 * p)[0].astype(np.int64)
        tot += o.cgd(free_vid=fv, fac=fc, x=pp.x0[fv], maxiters=25).fret
    assert abs(tot - total) <= 1e-6 * tot


@pytest.mark.gpu
def test_children_found_on_the_device_then_solved(harness, bal_path):
    # cameras assigned, points unassigned: createChildren() labels the 500 point components on the
    # device and optimizeBatch solves them -- identical to the hand-built batch above
    out, ref = np.zeros(8), np.zeros(6)
    assert harness.harness_ba_children_batch(bal_path, 49, 500, 25, out.ctypes.data_as(C.c_void_p)) == 0
    assert harness.harness_ba_points_batch(bal_path, 49, 500, 25, ref.ctypes.data_as(C.c_void_p)) == 0
    assert out[2] == 500 and out[6] == 3 and out[7] == 3
    assert np.array_equal(out[:6], ref)                                     # same components, same launch: bit-identical


@pytest.mark.gpu
@pytest.mark.parametrize("which", [0, 1])
def test_batch_shared_out_over_several_devices(harness, bal_path, which):
    """OptimizableFunction::setDevices + HipCGDSubspaceOptimizer::optimizeBatch: the sibling components of a level
    (src/Component.cpp:508-549; the reference visits them one after the other, RDISOptimizer.cpp:292-314) shared out
    over several devices from one process -- a context and a replica of the function per device, heaviest component
    first onto the least loaded device by factor count (the rule of rdis_amd/dist.py), every device's launch issued
    before any result is fetched, the batch's value summed in device order on the host.  Two and three contexts on the
    one GPU of the test box: every component's result is bit-identical to the single-context batch, the variables are
    left assigned to the same values, and a second round -- the other kind of component, starting from what the first
    round left on the OTHER contexts -- is bit-identical too (values do cross between the replicas)."""
    nc, npnt = 49, 500
    nvars = 9 * nc + 3 * npnt
    res = {}
    for ndev in (1, 2, 3):
        out, fret = np.zeros(5), np.zeros(600)
        iters, nfe, x = np.zeros(600, dtype=np.int64), np.zeros(600, dtype=np.int64), np.zeros(nvars)
        v = lambda a: a.ctypes.data_as(C.c_void_p)
        assert harness.harness_ba_children_batch_devices(bal_path, C.c_longlong(nc), C.c_longlong(npnt), 25, which, ndev, 2,
                                                         v(out), v(fret), v(iters), v(nfe), v(x)) == 0
        res[ndev] = (out.copy(), fret.copy(), iters.copy(), nfe.copy(), x.copy())
    o1, f1, i1, n1, x1 = res[1]
    ncomp = int(o1[1])
    assert ncomp == (nc if which == 0 else npnt)        # (the second round's components: the other kind)
    assert o1[3] < o1[2]
    for ndev in (2, 3):
        o, f, i, n, x = res[ndev]
        assert np.array_equal(f[:ncomp], f1[:ncomp]) and np.array_equal(i[:ncomp], i1[:ncomp]) and np.array_equal(n[:ncomp], n1[:ncomp])
        assert np.array_equal(x, x1)
        assert o[1] == o1[1] and o[3] == o1[3]                               # the function's value afterwards
        assert abs(o[0] - o1[0]) <= 1e-12 * abs(o1[0])                       # the batch's value: partial sums per device, then added
        assert o[4] >= ndev                                                  # a resident plan per device and kind


@pytest.mark.gpu
def test_batch_shared_out_over_distinct_gpus(harness, bal_path):
    """The same on DISTINCT GPUs (a node with at least two): one thread drives a context per GPU, so every C-ABI entry
    must make its context's device current before it allocates, records an event or launches -- with
    several contexts on one GPU that cannot fail, here it would.  Bit-identical to the one-device batch."""
    ngpu = capi.load_library().rdis_hip_device_count()
    if ngpu < 2:
        pytest.skip("needs at least two GPUs (%d visible)" % ngpu)
    nc, npnt = 49, 500
    res = {}
    for ndev in (1, -min(ngpu, 4)):
        out, fret = np.zeros(5), np.zeros(600)
        iters, nfe, x = np.zeros(600, dtype=np.int64), np.zeros(600, dtype=np.int64), np.zeros(9 * nc + 3 * npnt)
        v = lambda a: a.ctypes.data_as(C.c_void_p)
        assert harness.harness_ba_children_batch_devices(bal_path, C.c_longlong(nc), C.c_longlong(npnt), 25, 0, ndev, 2,
                                                         v(out), v(fret), v(iters), v(nfe), v(x)) == 0
        res[ndev] = (out.copy(), fret.copy(), iters.copy(), x.copy())
    a, b = res[1], res[-min(ngpu, 4)]
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3]) and a[0][3] == b[0][3]


@pytest.mark.gpu
def test_several_device_paths_under_virtual_devices():
    """The distinct-GPU test above needs two GPUs; a test box has one.  RDIS_HIP_VIRTUAL_DEVICES=3 makes the library offer three
    devices that all stand on GPU 0, remember which of them the calling thread made current last, and refuse every HIP call issued
    for a context while another context's device is current (rdis_hip.hip: HIPCHK) -- so an entry point that forgets to make its
    device current fails here as it would on a real node.  The several-device tests of this file (the distinct-GPU one among them:
    it no longer skips), the level driver over devices and the collective's several-context path run under it in a process of
    their own (the switch is read once)."""
    import subprocess
    import sys
    env = dict(os.environ, RDIS_HIP_VIRTUAL_DEVICES="3")
    out = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider",
                          os.path.abspath(__file__) + "::test_batch_shared_out_over_distinct_gpus",
                          os.path.abspath(__file__) + "::test_batch_shared_out_over_several_devices",
                          os.path.abspath(__file__) + "::test_level_driver_over_several_devices",
                          os.path.join(os.path.dirname(os.path.abspath(__file__)), "test_gpu_collective.py") + "::test_objective_of_several_contexts_of_one_process"],
                         env=env, capture_output=True, text=True, timeout=1200)
    assert out.returncode == 0 and " skipped" not in out.stdout.splitlines()[-1], out.stdout[-3000:] + out.stderr[-2000:]


@pytest.mark.gpu
def test_level_driver_over_several_devices(harness, bal_path):
    """HipRDISLevelOptimizer on a function replicated over two contexts: every level's components go through
    optimizeBatch's sharing-out; the sweeps, the final value and every variable are those of the one-device run."""
    nvars = 9 * 5 + 3 * 30
    res = {}
    for ndev in (1, 2):
        out, x = np.zeros(4), np.zeros(nvars)
        assert harness.harness_level_driver_devices(bal_path, C.c_longlong(5), C.c_longlong(30), 25, 20, C.c_double(0.2), ndev,
                                                    out.ctypes.data_as(C.c_void_p), x.ctypes.data_as(C.c_void_p)) == 0
        res[ndev] = (out.copy(), x.copy())
    assert res[1][0][0] < res[1][0][1]
    assert np.array_equal(res[1][1], res[2][1]) and np.array_equal(res[1][0], res[2][0])


@pytest.mark.gpu
def test_lm_optimizer_through_plugin(harness, bal_path, gctx):
    from rdis_amd import capi
    out, x = np.zeros(7), np.zeros(135)
    assert harness.harness_ba_lm(bal_path, 5, 30, 25, out.ctypes.data_as(C.c_void_p), x.ctypes.data_as(C.c_void_p)) == 0
    fret, delta, before, after, iters, stop, nsolve = out
    pp = P.load_bal(ncams=5, npts=30)
    r = capi.Problem(gctx, pp).lm_optimize(maxiters=25)                     # the C ABI directly: same kernels, same inputs
    assert fret == r.fret and np.array_equal(x, r.x) and (iters, stop, nsolve) == (r.iters, r.stop, r.nsolve)
    assert abs(before - 2111.5030158718296) <= 1e-12 * before and abs((fret - delta) - before) <= 1e-12 * before
    assert abs(after - fret) <= 1e-12 * fret and delta < 0                  # variables left assigned to the result


# ---- the caller side: separator heuristic (N4) and level driver (N1), rdis_amd/host/rdis_levels.* -------------
def _separator(h, path, nc, npnt, pct=0.2):
    sizes, *_ = _pack_bal(h, path, nc, npnt)
    sep = np.zeros(int(sizes[0]), dtype=np.int64)
    h.harness_separator.restype = C.c_longlong
    n = h.harness_separator(path, C.c_longlong(nc), C.c_longlong(npnt), C.c_double(pct), sep.ctypes.data_as(C.c_void_p))
    assert n >= 0
    return sep[:n], int(sizes[0])


def test_separator_reproduces_the_shape_of_the_reference_cut(harness, bal_path):
    """the reference's PaToH call (RDISOptimizer.cpp:779-865) cut ladybug-49-7776 at the 414 variables of
    46 of the 49 cameras, and ensureFactorWillBeAssigned (:412-458) added one point: 417 variables
    (SURVEY.md 3.2b).  The degree heuristic that stands in for the binary-only library finds the same
    shape; on the 5-camera / 30-point subset all 5 cameras and one point (the reference's 48-variable
    calls)."""
    sep, n = _separator(harness, bal_path, 0, 0)
    assert n == 23769 and len(sep) == 417
    cams = np.unique(sep[sep < 441] // 9)
    assert len(cams) == 46 and np.array_equal(np.sort(sep[sep < 441]), np.concatenate([9 * c + np.arange(9) for c in cams]))   # whole blocks
    assert np.array_equal(sep[sep >= 441], [441, 442, 443])                      # the point of the first factor that reads a cut camera
    # what is left decomposes: the 3 free cameras with the points they see, every other point on its own
    pp = P.load_bal()
    left = sorted(set(range(49)) - set(cams.tolist()))
    deg = np.bincount(pp.cam_vid0 // 9, minlength=49)
    assert sorted(np.argsort(deg, kind="stable")[:3].tolist()) == left           # the three cameras of lowest degree stay
    assigned = np.ones(pp.nvars, np.uint8)
    assigned[np.setdiff1d(np.arange(pp.nvars), sep)] = 0
    fp, fv, cp, ci = O.OracleProblem(pp).components(assigned)
    sizes = np.diff(fp)
    assert sizes.max() <= round(0.2 * 23769) and sizes.max() > 3 and np.sum(sizes == 3) == len(sizes) - 1
    sep, n = _separator(harness, bal_path, 5, 30)
    assert n == 135 and len(sep) == 48 and np.array_equal(sep, np.arange(48))    # 5 cameras + point 0


def _level_driver(h, path, nc, npnt, maxit=25, sweeps=20, pct=0.2, batch=1, nvars=0):
    out, tr, x = np.zeros(12), np.zeros((4096, 8)), np.zeros(max(nvars, 1))
    v = lambda a: a.ctypes.data_as(C.c_void_p)
    rc = h.harness_level_driver(path, C.c_longlong(nc), C.c_longlong(npnt), maxit, sweeps, C.c_double(pct), batch, v(out), v(tr),
                                C.c_longlong(4096), v(x) if nvars else None)
    assert rc == 0
    return out, tr[:int(out[9])], x


@pytest.mark.gpu
def test_level_driver_ladybug_5_30(harness, bal_path):
    """BASELINE config 3 through the level driver: beats the reference's complete RDIS run (18.443091288282886
    after 731 solver calls); monotone launch by launch; the running sum of the launches'
    deltas is the function value; batched == the same calls made one at a time, bit for bit"""
    out, tr, x = _level_driver(harness, bal_path, 5, 30, nvars=135)
    print("ladybug 5/30: %.6f -> %.12f in %d sweeps, %d launches, %.1f ms (decomposition %.1f ms); reference RDIS: 18.443091288282886" % (
        out[1], out[0], out[2], len(tr), out[6], out[7]))
    assert out[1] == 2111.5030158718296 and out[10] == 1.0 and out[11] <= 1e-9 * out[0]
    assert out[3] == 30 and out[5] == 1 and out[8] == 48                         # one split node (5 cameras + 1 point), 29 point leaves
    assert out[0] <= 18.443091288282886
    assert np.all(np.diff(tr[:, 6]) <= 0)
    pp = P.load_bal(ncams=5, npts=30)
    o = O.OracleProblem(pp, emulate_stale_cache=False)
    o.assign(None, x)
    assert abs(o.eval() - out[0]) <= 1e-12 * out[0]                              # the value IS the objective at the returned point
    assert np.all(x >= pp.lo) and np.all(x <= pp.hi)
    out1, tr1, x1 = _level_driver(harness, bal_path, 5, 30, batch=0, nvars=135)
    assert out1[0] == out[0] and np.array_equal(x1, x) and np.array_equal(tr1[:, 6], tr[:, 6])
    print("   the same calls one at a time: %.1f ms" % out1[6])


@pytest.mark.gpu
def test_level_driver_full_ladybug(harness, bal_path):
    """BASELINE config 4 through the level driver: the reference's RDIS run stood at 102978.259 when its
    240 s budget ran out (13 850 solver calls)"""
    out, tr, x = _level_driver(harness, bal_path, 0, 0, nvars=23769)
    print("ladybug full: %.3f -> %.6f in %d sweeps, %d launches, %.1f ms (decomposition %.1f ms): %d components, largest separator %d; "
          "reference RDIS: 102978.259 after 240 s" % (out[1], out[0], out[2], len(tr), out[6], out[7], out[3], out[8]))
    for r in tr[:4]:
        print("   sweep %d depth %d %s: %d components, %d variables, %d factors -> %.3f (%.2f ms)" % (
            r[0], r[1], "separators" if r[2] == 0 else "leaves", r[3], r[4], r[5], r[6], r[7]))
    assert out[10] == 1.0 and out[11] <= 1e-9 * out[0] and out[8] == 417
    assert out[0] < 102978.259
    pp = P.load_bal()
    o = O.OracleProblem(pp, emulate_stale_cache=False)
    o.assign(None, x)
    assert abs(o.eval() - out[0]) <= 1e-12 * out[0]


@pytest.mark.gpu
def test_level_driver_sweeps_on_full_ladybug_are_a_cpu_run_bit_for_bit(harness, bal_path):
    """BASELINE config 4 through the level driver, two sweeps: the separator's launch (one component, 417 free variables, 30 313
    factors: the pipelined cooperative solver) and the leaves' launch (7 775 points on the tiny-component solver, sixteen lanes each,
    beside three cameras with a cooperative group each) -- every component through the oracle that stands for ITS solver
    (device_default / device_group_default), from the values the launch before left: the device's point after the sweep, bit for bit."""
    from oracle import levels as LV
    cpp_nodes, cpp_plans = _cpp_tree(harness, 0, bal_path, 0, 0, 0.2, 0.0)
    lb = P.load_bal()
    nodes, plans, orc = _compare_tree(lb, cpp_nodes, cpp_plans, 0.2, 0.0)

    def oracle_for(plan):
        fp = plan[3]
        tiny = O.OracleProblem.device_group_default(lb, lanes=16) if int(np.sum(np.diff(fp) <= 4)) >= 4096 else None

        def make(v, fc):
            if len(v) <= 4 and tiny is not None:
                return tiny
            return O.OracleProblem.device_default(lb, free_vid=v, fac=fc)
        return make
    x = lb.x0
    for _ in range(2):   # (two sweeps: the second starts every component from what the first left)
        obj, x = LV.sweep(lb, O.OracleProblem(lb, emulate_stale_cache=False), plans, x, maxiters=25, oracle_for=oracle_for)
    out, tr, xd = _level_driver(harness, bal_path, 0, 0, sweeps=2, nvars=23769)
    assert len(tr) == 2 * len(plans)
    differ = np.nonzero(xd != x)[0]
    assert len(differ) == 0, (len(differ), differ[:10], float(np.max(np.abs(xd - x))))


@pytest.mark.gpu
def test_unchanged_caller_one_call_at_a_time(harness, bal_path):
    """What an unmodified RDISOptimizer would do with the drop-in: 80 calls over (5 cameras + 1 point) and 651
    calls over single points of ladybug 5/30 (the reference's own run), one optimize() at a
    time.  The plan cache makes the sequence faster and changes no bit.  Measured on the GPU box (host: EPYC
    9575F): 99-107 ms with the cache, 120-127 ms without, 87-108 ms for the CPU oracle on one core making the same calls --
    a 3-variable call is 0.07 ms on that CPU, less than one kernel launch plus the latency chain of its ~70
    dependent evaluations, so call-at-a-time cannot win on the tiny calls; the 80 large calls do (0.5 ms against
    1 ms each).  The same work as sibling batches (test_level_driver_ladybug_5_30) takes 20 ms."""
    import time
    v = lambda a: a.ctypes.data_as(C.c_void_p)
    res = {}
    for cache in (0, 256):
        out, x = np.zeros(6), np.zeros(135)
        assert harness.harness_call_shapes(bal_path, 80, 651, 25, cache, v(out), v(x)) == 0
        res[cache] = (out.copy(), x.copy())
    (o0, x0), (o1, x1) = res[0], res[256]
    assert o0[1] == o1[1] == 731 and o1[4] == 731 - 30 and o1[5] == 30 and o0[4] == 0
    assert o0[2] == o1[2] and o0[3] == o1[3] and np.array_equal(x0, x1)          # cached == uncached, bit for bit
    # the same sequence of calls on the CPU oracle (its own trajectory), one core
    pp = P.load_bal(ncams=5, npts=30)
    orc = O.OracleProblem(pp)
    sepv, sepf = np.arange(48, dtype=np.int64), np.arange(pp.nfac, dtype=np.int64)
    pf = [np.where(pp.pt_vid0 == 45 + 3 * p)[0].astype(np.int64) for p in range(30)]
    t0 = time.perf_counter()
    nxt, done = 1, 0
    for r in range(80):
        orc.cgd(free_vid=sepv, fac=sepf, maxiters=25)
        quota = 651 * (r + 1) // 80 - done
        for _ in range(quota):
            orc.cgd(free_vid=np.arange(45 + 3 * nxt, 48 + 3 * nxt, dtype=np.int64), fac=pf[nxt], maxiters=25)
            nxt = 1 if nxt == 29 else nxt + 1
            done += 1
    cpu_ms = (time.perf_counter() - t0) * 1e3
    print("731 optimize() calls one at a time: %.1f ms with the plan cache, %.1f ms without; CPU oracle on one core %.1f ms; "
          "function value %.6f (oracle's own trajectory: %.6f)" % (o1[0], o0[0], cpu_ms, o1[2], orc.eval()))
    # (timing is reported, not asserted beyond a generous bound: a loaded host must not fail a correctness suite.
    # Measured: on a par with one CPU core box by box, 0.9 ... 1.25 x; the plan cache's effect is asserted by its counters above)
    if not (o1[0] < o0[0] and o1[0] < 1.5 * cpu_ms):
        import warnings
        warnings.warn("731 cached optimize() calls took %.1f ms (uncached %.1f ms, one CPU core %.1f ms)" % (o1[0], o0[0], cpu_ms))
    assert o1[0] < 10.0 * max(cpu_ms, 50.0)


@pytest.mark.gpu
def test_level_driver_multi_level_tree_on_the_sinusoid(harness):
    """config 2 through the level driver from the committed full-domain start: every variable is its own
    block; with leaves of at most 12 variables and separators that cut a node roughly in half
    (sepPiecePct 0.5) the 121-variable tree is split over several depths: the nodes partition the variables, the function is monotone launch by launch, the
    batched run equals the one-call-at-a-time run bit for bit, and the result is the oracle's objective
    at the returned point"""
    import json
    with open(os.path.join(ROOT, "tests", "golden", "sinusoid_start.json")) as fh:
        x0 = np.array(json.load(fh)["x0"])
    v = lambda a: a.ctypes.data_as(C.c_void_p)
    res = {}
    for batch in (1, 0):
        out, x = np.zeros(10), np.zeros(121)
        assert harness.harness_level_driver_sinusoid(v(x0), 25, 10, C.c_double(0.1), C.c_double(0.5), batch, v(out), v(x)) == 0
        res[batch] = (out.copy(), x.copy())
    (ob, xb), (os_, xs) = res[1], res[0]
    print("sinusoid: %.3f -> %.6f in %d sweeps; %d nodes (%d split, %d leaves), depth %d; %.1f ms batched, %.1f ms one call at a time" % (
        ob[1], ob[0], ob[2], ob[3], ob[5], ob[4], ob[6], ob[9], os_[9]))
    assert abs(ob[1] - 17126.136253546265) < 1e-6 and ob[7] == 1.0 and ob[8] <= 1e-9 * max(abs(ob[0]), 1.0)
    assert ob[6] >= 2 and ob[5] >= 2                                   # really a multi-level decomposition
    assert ob[0] < 0.2 * ob[1]
    assert ob[0] == os_[0] and np.array_equal(xb, xs)                  # batched == sequential, bit for bit
    pp = P.make_high_dim_sinusoid()
    o = O.OracleProblem(pp, emulate_stale_cache=False)
    o.assign(None, xb)
    assert abs(o.eval() - ob[0]) <= 1e-12 * max(abs(ob[0]), 1.0)
    assert np.all(xb >= pp.lo) and np.all(xb <= pp.hi)


@pytest.mark.gpu
def test_level_driver_sweeps_on_the_sinusoid_are_a_cpu_run_bit_for_bit(harness):
    """... and config 2's multi-level tree: two sweeps of the level driver == the oracle's sweeps with every launch's components on
    the plain workgroup solver's restatement (OracleProblem.device_wg_default: the device's sine / cosine, its sums for the launch's
    workgroup size)"""
    import json
    from oracle import levels as LV
    with open(os.path.join(ROOT, "tests", "golden", "sinusoid_start.json")) as fh:
        x0 = np.array(json.load(fh)["x0"])
    cpp_nodes, cpp_plans = _cpp_tree(harness, 1, None, 0, 0, 0.1, 0.5)
    pp = P.make_high_dim_sinusoid()
    nodes, plans, _ = _compare_tree(pp, cpp_nodes, cpp_plans, 0.1, 0.5)

    def oracle_for(plan):
        fp, cp = plan[3], plan[5]
        mf = int(max(np.diff(cp).max(), np.diff(fp).max() // 4))
        threads = 64 if mf <= 64 else 128 if mf <= 128 else 256 if mf <= 256 else 512 if mf <= 512 else 768
        return lambda v, fc: O.OracleProblem.device_wg_default(pp, free_vid=v, fac=fc, threads=threads)
    x = x0
    for _ in range(2):
        obj, x = LV.sweep(pp, O.OracleProblem(pp, emulate_stale_cache=False), plans, x, maxiters=25, oracle_for=oracle_for)
    v = lambda a: a.ctypes.data_as(C.c_void_p)
    out, xd = np.zeros(10), np.zeros(121)
    assert harness.harness_level_driver_sinusoid(v(x0), 25, 2, C.c_double(0.1), C.c_double(0.5), 1, v(out), v(xd)) == 0
    assert np.array_equal(xd, x), float(np.max(np.abs(xd - x)))


@pytest.mark.gpu
def test_level_driver_on_a_larger_bal_file(harness, tmp_path):
    """the whole caller side at a size beyond ladybug: a BAL file of 64 cameras x 20000 points (80000
    observations, 60576 variables) written by the saver, read by the C++ loader, decomposed (separator:
    52 of the 64 cameras + 1 point; one launch of 12706 children) and optimised by sweeps -- the 65505-factor
    separator solve fills the cooperative solver to its capacity (256 workgroups of 256 lanes)"""
    pp = P.make_synthetic_ba(1, 64, 20000, obs_per_pt=4)
    path = str(tmp_path / "synthetic_64_20000.txt")
    P.save_bal(pp, path)
    out, tr, x = _level_driver(harness, path.encode(), 0, 0, sweeps=6, nvars=pp.nvars)
    print("64 x 20000: %.6g -> %.6g in %d sweeps, %d launches, %.1f ms (decomposition %.1f ms); %d components, largest separator %d" % (
        out[1], out[0], out[2], len(tr), out[6], out[7], out[3], out[8]))
    assert out[10] == 1.0 and out[11] <= 1e-9 * out[0] and out[0] < 0.05 * out[1]
    assert out[5] == 1 and out[8] % 9 == 3 and out[3] > 10000
    o = O.OracleProblem(pp, emulate_stale_cache=False)
    assert abs(o.eval() - out[1]) <= 1e-12 * out[1]                     # the file round trip kept the start
    o.assign(None, x)
    assert abs(o.eval() - out[0]) <= 1e-12 * out[0]
    assert np.all(x >= pp.lo) and np.all(x <= pp.hi)


# ---- the decomposition against its independent restatement (oracle/levels.py) -----------------------------------
def _cpp_tree(h, which, path, nc, npnt, pct, seppct):
    h.harness_level_tree.restype = C.c_longlong
    cap = 1 << 22
    buf = np.zeros(cap, dtype=np.int64)
    n = h.harness_level_tree(which, path, C.c_longlong(nc), C.c_longlong(npnt), C.c_double(pct), C.c_double(seppct),
                             buf.ctypes.data_as(C.c_void_p), C.c_longlong(cap))
    assert n > 0, n
    pos = [0]

    def take(k=None):
        if k is None:
            v = int(buf[pos[0]]); pos[0] += 1
            return v
        v = buf[pos[0]:pos[0] + k].copy(); pos[0] += k
        return v
    nodes = []
    for _ in range(take()):
        depth, parent, leaf, nv, nf, ns, nsf = (take() for _ in range(7))
        nodes.append(dict(depth=depth, parent=parent, leaf=bool(leaf), vars=take(nv), factors=take(nf), separator=take(ns), sep_factors=take(nsf)))
    plans = []
    for _ in range(take()):
        depth, kind, nc_ = take(), take(), take()
        fp = take(nc_ + 1); fv = take(take()); cp = take(nc_ + 1); ci = take(take())
        plans.append((depth, kind, fp, fv, cp, ci))
    assert pos[0] == n
    return nodes, plans


def _compare_tree(pp, cpp_nodes, cpp_plans, pct, seppct):
    from oracle import levels as LV
    orc = O.OracleProblem(pp)
    nodes = LV.build_tree(pp, orc, blkpct=pct, seppct=seppct)
    assert len(nodes) == len(cpp_nodes)
    for a, b in zip(nodes, cpp_nodes):
        assert (a.depth, a.parent, a.leaf) == (b["depth"], b["parent"], b["leaf"])
        assert np.array_equal(a.vars, b["vars"]) and np.array_equal(a.factors, b["factors"])
        assert np.array_equal(a.separator, b["separator"]) and np.array_equal(a.sep_factors, b["sep_factors"])
    plans = LV.level_plans(nodes)
    assert len(plans) == len(cpp_plans)
    for (d, k, _idx, fp, fv, cp, ci), (d2, k2, fp2, fv2, cp2, ci2) in zip(plans, cpp_plans):
        assert (d, k) == (d2, k2)
        assert np.array_equal(fp, fp2) and np.array_equal(fv, fv2) and np.array_equal(cp, cp2) and np.array_equal(ci, ci2)
    # a partition: every variable in exactly one leaf or one separator
    seen = np.zeros(pp.nvars, np.int64)
    for nd in nodes:
        np.add.at(seen, nd.vars if nd.leaf else nd.separator, 1)
    assert np.all(seen == 1)
    return nodes, plans, orc


@pytest.mark.gpu
@pytest.mark.parametrize("nc,npnt,pct,seppct", [(5, 30, 0.2, 0.0), (0, 0, 0.2, 0.0), (7, 50, 0.2, 0.0), (20, 300, 0.1, 0.0),
                                                (49, 1000, 0.05, 0.5), (12, 200, 0.1, 0.4)])
def test_level_tree_is_the_stated_rule_bit_for_bit(harness, bal_path, nc, npnt, pct, seppct):
    """The level driver's decomposition -- nodes, separators, children, the CSR lists of every launch -- equals
    the independent Python restatement of the stated rule (oracle/levels.py: sets and a fresh search per step
    instead of the C++'s incremental union-find; components from the CPU oracle's union-find instead of the
    device labelling) entry for entry: ladybug 5/30, full, and four subsets with other block fractions, two of
    them with sepPiecePct > 0 (several levels)."""
    cpp_nodes, cpp_plans = _cpp_tree(harness, 0, bal_path, nc, npnt, pct, seppct)
    pp = P.load_bal(ncams=nc, npts=npnt)
    nodes, plans, _ = _compare_tree(pp, cpp_nodes, cpp_plans, pct, seppct)
    if (nc, npnt) == (0, 0):
        assert len(nodes[0].separator) == 417 and sum(1 for nd in nodes if nd.depth == 1) == 6561   # 46 cameras + a point; 3 cameras' piece + 6560 points
    if seppct > 0:
        assert max(nd.depth for nd in nodes) >= 2


@pytest.mark.gpu
def test_level_tree_of_the_sinusoid_and_first_sweep_against_the_oracle(harness, bal_path):
    """config 2's tree (every variable its own block, sepPiecePct 0.5: depth 4) bit for bit, and the first sweep
    of ladybug 5/30 launch by launch: the oracle, solving the same components from the same values one after
    the other, follows the device's objective within the spread two roundings of this chaotic descent show
    after 25 iterations (the first launch: one 48-variable solve, a percent or two)."""
    from oracle import levels as LV
    cpp_nodes, cpp_plans = _cpp_tree(harness, 1, None, 0, 0, 0.1, 0.5)
    pp = P.make_high_dim_sinusoid()
    nodes, plans, _ = _compare_tree(pp, cpp_nodes, cpp_plans, 0.1, 0.5)
    assert max(nd.depth for nd in nodes) >= 3 and len(nodes) > 40
    # first sweep, ladybug 5/30
    cpp_nodes, cpp_plans = _cpp_tree(harness, 0, bal_path, 5, 30, 0.2, 0.0)
    lb = P.load_bal(ncams=5, npts=30)
    nodes, plans, orc = _compare_tree(lb, cpp_nodes, cpp_plans, 0.2, 0.0)
    obj, x = LV.sweep(lb, O.OracleProblem(lb), plans, lb.x0, maxiters=25)
    out, tr, _ = _level_driver(harness, bal_path, 5, 30, sweeps=1, nvars=135)
    dev = tr[:, 6]
    assert len(dev) == len(obj) == len(plans)
    print("first sweep, ladybug 5/30: device %s, oracle %s" % (np.array2string(dev, precision=6), np.array2string(obj, precision=6)))
    f0 = out[1]
    assert np.all(np.diff(np.concatenate([[f0], obj])) <= 0)                 # monotone on the oracle too
    assert np.all(np.abs(dev - obj) <= 0.1 * np.abs(obj))                    # same descent, two roundings
    # the oracle's running sum of deltas is its function value
    o2 = O.OracleProblem(lb)
    o2.assign(np.arange(lb.nvars, dtype=np.int64), x)
    assert abs(o2.eval() - obj[-1]) <= 1e-9 * abs(obj[-1])


@pytest.mark.gpu
@pytest.mark.parametrize("sweeps", [1, 3, 20])
def test_level_driver_sweeps_on_ladybug_5_30_are_a_cpu_run_bit_for_bit(harness, bal_path, sweeps):
    """... and with the oracle standing for the solver the dispatcher gives each launch (here the LDS-resident one: the device's
    factor arithmetic through factors.hpp compiled for the host, its sum trees for the launch's workgroup size --
    OracleProblem.device_lds_default) the level driver's sweeps over the tree of BASELINE config 3 end at the CPU's point bit for bit:
    one 48-variable separator solve and 29 point leaves a sweep, every component from the values the launches before it left; sweeps = 20: the WHOLE run of test_level_driver_ladybug_5_30
    (the value that beats the reference's complete RDIS run), to its last bit."""
    from oracle import levels as LV
    cpp_nodes, cpp_plans = _cpp_tree(harness, 0, bal_path, 5, 30, 0.2, 0.0)
    lb = P.load_bal(ncams=5, npts=30)
    nodes, plans, orc = _compare_tree(lb, cpp_nodes, cpp_plans, 0.2, 0.0)

    def oracle_for(plan):
        mf = int(np.diff(plan[5]).max())
        threads = 64 if mf <= 64 else 128 if mf <= 128 else 256
        return lambda v, fc: O.OracleProblem.device_lds_default(lb, free_vid=v, fac=fc, threads=threads)
    out, tr, xd = _level_driver(harness, bal_path, 5, 30, sweeps=sweeps, nvars=135)
    done = int(out[2])      # (sweeps = 20: the whole run -- the driver stops when a sweep gains less than its tolerance)
    assert len(tr) == done * len(plans) and (done == sweeps or sweeps == 20)
    x = lb.x0
    for _ in range(done):
        obj, x = LV.sweep(lb, O.OracleProblem(lb, emulate_stale_cache=False), plans, x, maxiters=25, oracle_for=oracle_for)
    assert np.array_equal(xd, x), float(np.max(np.abs(xd - x)))
    assert abs(tr[-1, 6] - obj[-1]) <= 1e-12 * abs(obj[-1])     # (the running objective: the launches' deltas added in another order)


@pytest.mark.gpu
@pytest.mark.parametrize("nc,npnt,pct,seppct,nrr,maxna,steptol", [(5, 30, 0.2, 0.0, 2, 10, 1e-2), (12, 200, 0.1, 0.4, 2, 3, 0.5), (7, 50, 0.2, 0.0, 4, 5, 0.1)])
def test_reference_schedule_of_the_level_driver(harness, bal_path, nc, npnt, pct, seppct, nrr, maxna, steptol):
    """HipRDISLevelOptimizer::optimizeReferenceSchedule -- per-node iterative improvement and random restarts the way
    RDISOptimizer::doOptimization / getValueFromDomain / updateDomain run them (src/RDISOptimizer.cpp:253-334, 971-1147,
    1507-1577) -- against oracle/levels.py::replay_reference_schedule, a restatement written from those lines that walks
    the tree depth first like the reference and takes every decision itself from the values the device's solves
    returned: kind of every step (initial values / iterative improvement / random restart), restart counts, forced
    restarts after maxNAtoRR assignments, the restart's start vector bit for bit, updateDomain's verdicts, where each
    node's loop ends -- and nothing of the device's trace may be left over.  (steptol is the reference's absolute
    progress threshold, default 1e-4; larger here so that the loops end in seconds.)  Lock-step batching over independent nodes
    and several devices change none of it; the function value never rises above the start's and the variables are left
    at the best evaluation found.  (End-to-end parity with optBA is not pinned: PaToH, Boost's generator.)"""
    from oracle import levels as LV
    pp = P.load_bal(ncams=nc, npts=npnt)
    nodes = LV.build_tree(pp, O.OracleProblem(pp), blkpct=pct, seppct=seppct)
    slo, shi = LV.ba_sampling_intervals(pp, nc)
    res = {}
    for ndev in (1, 2):
        out, tr, x = np.zeros(4), np.zeros((300000, 10)), np.zeros(pp.nvars)
        harness.harness_level_reference.restype = C.c_longlong
        n = harness.harness_level_reference(bal_path, C.c_longlong(nc), C.c_longlong(npnt), 25, C.c_double(pct), C.c_double(seppct), nrr, maxna,
                                            C.c_double(12345.0), ndev, C.c_double(300000.0), C.c_double(steptol), out.ctypes.data_as(C.c_void_p), tr.ctypes.data_as(C.c_void_p),
                                            C.c_longlong(len(tr)), x.ctypes.data_as(C.c_void_p))
        assert 0 < n <= len(tr) and out[2] == len(nodes)
        res[ndev] = (out.copy(), tr[:n].copy(), x.copy())
    out, tr, x = res[1]
    assert np.array_equal(tr, res[2][1], equal_nan=True) and np.array_equal(x, res[2][2]) and np.array_equal(out, res[2][0])   # two contexts: the same run
    rows = [tuple(r[:8]) + (int(r[8]) << 32 | int(r[9]),) for r in tr]
    assert len(tr) < 300000                                     # (the budget of calls -- the reference's time limit -- was not what ended it)
    calls = LV.replay_reference_schedule(nodes, rows, pp, slo, shi, steptol=steptol, nrr_per_lvl=nrr, max_na_to_rr=maxna, seed=12345, max_calls=300000)
    kinds = np.bincount(tr[:, 1].astype(int), minlength=3)
    print("reference schedule, %d cameras / %d points: %d nodes, %d subspace-optimizer calls (%d initial, %d iterative improvement, %d random restarts), "
          "%.6f -> %.6f" % (nc, npnt, len(nodes), calls, kinds[0], kinds[1], kinds[2], out[1], out[0]))
    assert calls == len(tr) and kinds[1] > 0 and kinds[2] > 0
    assert out[0] <= out[1]
    o = O.OracleProblem(pp)
    o.assign(np.arange(pp.nvars, dtype=np.int64), x)
    assert abs(o.eval() - out[0]) <= 1e-10 * abs(out[0])       # the variables are left at what was returned


@pytest.mark.gpu
def test_optba_entry_runs_the_reference_schedule(harness, bal_path):
    """include/rdis_optba.h (what bench.py's all_components block calls): the same run as the harness's, through the C
    entry of librdis_host.so -- same end value and variables bit for bit, the calls and CG iterations it reports are the
    trace's, and the schedule-0 run (level sweeps) is monotone and ends lower than it started."""
    lib = C.CDLL(os.path.join(ROOT, "rdis_amd", "lib", "librdis_host.so"))
    lib.rdis_optba_run.argtypes = [C.c_char_p, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.POINTER(C.c_char_p), C.POINTER(C.c_double),
                                   C.c_int32, C.c_void_p, C.c_void_p]
    v = lambda a: a.ctypes.data_as(C.c_void_p)
    names = [b"SSmaxit", b"AVblkpct", b"sepPiecePct", b"nRRperLvl", b"maxNAtoRR", b"restartSeed", b"maxCalls", b"steptol"]
    vals = [25.0, 0.2, 0.0, 2.0, 10.0, 12345.0, 300000.0, 1e-2]
    out, x = np.zeros(10), np.zeros(135)
    assert lib.rdis_optba_run(bal_path, 5, 30, 1, len(names), (C.c_char_p * len(names))(*names), (C.c_double * len(vals))(*vals), 0, v(out), v(x)) == 0
    ho, tr, hx = np.zeros(4), np.zeros((300000, 10)), np.zeros(135)
    harness.harness_level_reference.restype = C.c_longlong
    n = harness.harness_level_reference(bal_path, C.c_longlong(5), C.c_longlong(30), 25, C.c_double(0.2), C.c_double(0.0), 2, 10, C.c_double(12345.0), 1,
                                        C.c_double(300000.0), C.c_double(1e-2), v(ho), v(tr), C.c_longlong(len(tr)), v(hx))
    assert n > 0 and out[0] == ho[0] and out[1] == ho[1] and np.array_equal(x, hx)
    assert out[2] == n == out[9] and out[3] >= out[2] and out[4] < out[2] and out[5] > 0 and out[7] == ho[2]
    pp = P.load_bal(ncams=5, npts=30)
    o = O.OracleProblem(pp)
    assert abs(out[1] - o.eval()) <= 1e-12 * out[1]
    o.assign(np.arange(pp.nvars, dtype=np.int64), x)
    assert abs(o.eval() - out[0]) <= 1e-10 * abs(out[0]) and out[0] < out[1]
    out0 = np.zeros(10)
    assert lib.rdis_optba_run(bal_path, 5, 30, 0, 1, (C.c_char_p * 1)(b"SSmaxit"), (C.c_double * 1)(25.0), 0, v(out0), None) == 0
    assert out0[0] < out0[1] == out[1] and out0[2] > 0 and out0[3] >= out0[2]
    bad = (C.c_char_p * 1)(b"noSuchOption")
    assert lib.rdis_optba_run(bal_path, 5, 30, 0, 1, bad, (C.c_double * 1)(1.0), 0, v(out0), None) == -3


@pytest.mark.gpu
def test_plan_cache_is_bounded_and_never_changes_a_bit(harness, bal_path):
    """The host-side cache of resident plans: bounded by entries AND by device bytes, least recently used
    first; a call whose plan does not fit the budget at all is served by the transient path.  Whatever the bounds,
    the 243 calls (30 shapes) give the bits of the uncached run.  And an optimizer may outlive its function: the
    function takes the cached plans of its device problem with it."""
    v = lambda a: a.ctypes.data_as(C.c_void_p)

    def run(cache, nbytes):
        out, x = np.zeros(9), np.zeros(135)
        assert harness.harness_call_shapes_budget(bal_path, 27, 216, 10, cache, C.c_double(nbytes), v(out), v(x)) == 0
        return out, x
    base, xb = run(0, 0)                                   # no cache at all
    assert base[4] == 0 and base[1] == 243
    full, xf = run(256, 4 << 30)
    assert full[4] == 243 - 30 and full[5] == 30 and full[6] == 0 and full[7] == 30 and full[8] > 0
    small, xs = run(4, 4 << 30)                            # four entries for thirty shapes: constant eviction
    assert small[7] == 4 and small[5] > 30 and small[6] == 0
    per_plan = full[8] / 30
    tight, xt = run(256, 3.5 * per_plan)                   # room for a few plans (the separator's is larger than a point's)
    assert 1 <= tight[7] < 30 and 0 < tight[8] <= 3.5 * per_plan and tight[5] > 30
    none, xn = run(256, 16)                                # nothing fits: every call goes the transient way
    assert none[7] == 0 and none[8] == 0 and none[6] == none[5] == 243 and none[4] == 0
    for o, x in ((full, xf), (small, xs), (tight, xt), (none, xn)):
        assert o[2] == base[2] and o[3] == base[3] and np.array_equal(x, xb)
    assert harness.harness_optimizer_outlives_function(bal_path) == 4


@pytest.mark.gpu
def test_exponential_factor_through_the_mirrored_classes(harness):
    """NonlinearProductFactor(id, coefficient, useExponential) as the reference constructs it
    (src/NonlinearProductFactor.h:61-63): values include exp(-product) (.cpp:140), a gradient over the flagged factor
    is refused (.cpp:110 asserts), over the other factors it is the plain one."""
    x = np.array([1.3, -0.7, 0.9])
    out = np.zeros(6)
    assert harness.harness_nlp_exponential(x.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p)) == 0
    u = (x[0] - 0.5) ** 3
    f0, f1, f2 = 2.0 * np.sin(u) * x[1], 1.5 * np.exp(-(x[1] ** 2 * x[2])), -0.75 * x[2]
    assert out[0] == pytest.approx(f0 + f1 + f2, rel=1e-14) and out[1] == pytest.approx(f1, rel=1e-15)
    assert out[2] == 1
    want = [2.0 * np.cos(u) * 3 * (x[0] - 0.5) ** 2 * x[1], 2.0 * np.sin(u), -0.75]
    assert np.allclose(out[3:6], want, rtol=1e-14, atol=0)
