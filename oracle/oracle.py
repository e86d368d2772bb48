"""ctypes binding of the CPU oracle (oracle/rdis_oracle.c) and of oracle/_ref.

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  Nothing under rdis_amd/ imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle.so")
_REF = os.path.join(_HERE, "_ref", "libref_nrc.so")

_f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_i64p = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")

FUNC_CB = C.CFUNCTYPE(C.c_double, C.c_void_p, C.POINTER(C.c_double))
GRAD_CB = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double))


class ReplayReport(C.Structure):
    _fields_ = [("consumed", C.c_int64), ("first_mismatch", C.c_int64), ("step_mismatches", C.c_int64),
                ("tag_mismatches", C.c_int64), ("underrun", C.c_int32), ("reason", C.c_int32),
                ("iters", C.c_int32), ("fret", C.c_double), ("finit", C.c_double),
                ("max_step_rel", C.c_double), ("max_f_rel", C.c_double), ("max_slope_rel", C.c_double),
                ("max_iter_rel", C.c_double), ("max_vec_rel", C.c_double),
                ("max_f_rel_near", C.c_double), ("max_slope_rel_near", C.c_double), ("last_near", C.c_int32),
                ("synced_iters", C.c_int64),
                ("pending_slope", C.c_double), ("max_f_far_ulps", C.c_double),
                ("max_f_bound", C.c_double), ("max_slope_bound", C.c_double)]

    def __repr__(self):
        return "ReplayReport(" + ", ".join(f"{n}={getattr(self, n)}" for n, _ in self._fields_ if n != "pending_slope") + ")"


class _Result(C.Structure):
    _fields_ = [("fret", C.c_double), ("delta", C.c_double), ("finit", C.c_double),
                ("iters", C.c_int32), ("status", C.c_int32),
                ("nfeval", C.c_int64), ("ngeval", C.c_int64)]


@dataclass
class CGDResult:
    fret: float
    delta: float
    finit: float
    iters: int
    status: int
    nfeval: int
    ngeval: int
    x: np.ndarray


def build(force: bool = False) -> None:
    if force or not os.path.exists(_LIB) or \
            os.path.getmtime(_LIB) < os.path.getmtime(os.path.join(_HERE, "rdis_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    if os.path.isdir("/root/reference") and (force or not os.path.exists(_REF)):
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)


_lib = None
_FH = os.path.join(os.path.dirname(_HERE), "tests", "cpp", "libfactors_host.so")
_fh = None


class _FactorArith(C.Structure):
    _fields_ = [("value", C.c_void_p), ("eval_grad", C.c_void_p), ("value_slope", C.c_void_p)]


class _PtmArith(C.Structure):
    _fields_ = [("camera_trial", C.c_void_p), ("camera_trial_dir", C.c_void_p), ("trial", C.c_void_p)]


def factors_host():
    """tests/cpp/factors_host.hip -- rdis_amd/csrc/factors.hpp compiled for the HOST with the device's default contraction (the same
    front end fuses a * b + c within an expression for either target) -- as an ro_factor_arith for ro_set_factor_arithmetic.
    Built on demand (hipcc compiles host code without a GPU)."""
    global _fh
    if _fh is None:
        src = os.path.join(os.path.dirname(_FH), "factors_host.hip")
        hdr = os.path.join(os.path.dirname(_HERE), "rdis_amd", "csrc", "factors.hpp")
        if not os.path.exists(_FH) or os.path.getmtime(_FH) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
            subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17", "-mfma", "-ffp-contract=on", "-fPIC", "-shared",
                                   "-o", _FH, src], stderr=subprocess.DEVNULL)
        L = C.CDLL(_FH)
        arith = _FactorArith(C.cast(L.fh_value, C.c_void_p), C.cast(L.fh_eval_grad, C.c_void_p), C.cast(L.fh_value_slope, C.c_void_p))
        ptm = _PtmArith(C.cast(L.fh_camera_trial, C.c_void_p), C.cast(L.fh_camera_trial_dir, C.c_void_p), C.cast(L.fh_trial, C.c_void_p))
        _fh = (L, arith, ptm)
    return _fh


def ptm_point_order(cam_vid0, pt_vid0, spread: int = 16, wide: bool = False, local_cus: int = 0):
    """the point blocks of one component in the point-major streaming solver's order (rdis_hip.hip: prepare_partition), from the
    component's listed factors' camera / point blocks: by number of listed factors descending, among equals by their cameras (the
    cameras' ranks, in listed order) lexicographically, ties by id; the whole wave-chunks of 64 blocks whose first blocks have
    equally many factors are then dealt out over `spread` equal runs of their sorted order (ptm_api.hpp: PTM_SPREAD) -- round robin,
    or, for a component that gets a WIDE group, position c takes the next chunk of run h(c), h a weighted sum of c's hexadecimal
    digits mod 16 (the first run with chunks left from there on); or, for a wide group with LOCAL camera numbering (local_cus: the
    device's compute units; returns the workgroups' chunk ranges as a third value), workgroup r of K = min(compute units, 512, chunks /
    24) takes a contiguous slice of every run, its positions consecutive, wave w of eight the w-th, w + 8-th, ... of them"""
    cams, cam_rank = np.unique(cam_vid0, return_inverse=True)
    pts, pt_rank = np.unique(pt_vid0, return_inverse=True)
    npb = len(pts)
    order_f = np.argsort(pt_rank, kind="stable")            # listed factors by point block, listed order within
    deg = np.bincount(pt_rank, minlength=npb)
    ptr = np.concatenate([[0], np.cumsum(deg)])
    crk = cam_rank[order_f]
    lists = [crk[ptr[b]:ptr[b + 1]].tolist() for b in range(npb)]
    order = sorted(range(npb), key=lambda b: (-len(lists[b]), lists[b], b))
    nfull = npb // 64
    if local_cus:
        npc_all = -(-npb // 64)
        Kl = min(local_cus, 512, max(1, npc_all // 24))
        wl = [[] for _ in range(Kl)]
        a0 = 0
        while a0 < nfull:
            T = len(lists[order[64 * a0]])
            a1 = a0
            while a1 < nfull and len(lists[order[64 * a1]]) == T:
                a1 += 1
            mm = a1 - a0
            for rk in range(Kl):
                wl[rk].extend(range(a0 + rk * mm // Kl, a0 + (rk + 1) * mm // Kl))
            a0 = a1
        chunk_of = [0] * nfull
        wg_chunk0 = [0] * (Kl + 1)
        pos, nwv = 0, 8
        for rk in range(Kl):
            li, taken = wl[rk], 0
            wg_chunk0[rk] = pos
            for w in range(nwv):
                cnt = (len(li) - w + nwv - 1) // nwv if len(li) > w else 0
                for j in range(cnt):
                    chunk_of[pos + w + nwv * j] = li[taken + j]
                taken += cnt
            pos += len(li)
        wg_chunk0[Kl] = npc_all
        out = list(order)
        for a in range(nfull):
            out[64 * a:64 * a + 64] = order[64 * chunk_of[a]:64 * chunk_of[a] + 64]
        return cams.astype(np.int64), pts[np.asarray(out, dtype=np.int64)].astype(np.int64), np.asarray(wg_chunk0, dtype=np.int64)
    chunk_of = []
    a0 = 0
    while a0 < nfull:
        T = len(lists[order[64 * a0]])
        a1 = a0
        while a1 < nfull and len(lists[order[64 * a1]]) == T:
            a1 += 1
        mm = a1 - a0
        q = -(-mm // spread)
        if not wide:
            for rr in range(q):
                for gg in range(spread):
                    idx = gg * q + rr
                    if idx < mm:
                        chunk_of.append(a0 + idx)
        else:
            used = [0] * spread
            for _ in range(mm):
                pos = len(chunk_of)
                gg = ((pos & 15) + 15 * ((pos >> 4) & 15) + ((pos >> 8) & 15) + 15 * ((pos >> 12) & 15) + ((pos >> 16) & 15) + ((pos >> 20) & 15)) % spread
                for _t in range(spread):
                    if gg * q + used[gg] < min(mm, (gg + 1) * q):
                        break
                    gg = (gg + 1) % spread
                chunk_of.append(a0 + gg * q + used[gg])
                used[gg] += 1
        a0 = a1
    out = list(order)
    for a in range(nfull):
        out[64 * a:64 * a + 64] = order[64 * chunk_of[a]:64 * chunk_of[a] + 64]
    return cams.astype(np.int64), pts[np.asarray(out, dtype=np.int64)].astype(np.int64)


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB)
        L.ro_ba_factor_eval.restype = C.c_double
        L.ro_ba_factor_eval.argtypes = [_f64p, C.c_double, C.c_double]
        L.ro_ba_factor_grad.restype = C.c_double
        L.ro_ba_factor_grad.argtypes = [_f64p, C.c_double, C.c_double, _f64p]
        L.ro_ba_factor_grad_ref.restype = C.c_double
        L.ro_ba_factor_grad_ref.argtypes = [_f64p, C.c_double, C.c_double, _f64p]
        L.ro_set_ba_derivative.argtypes = [C.c_void_p, C.c_int]
        L.ro_set_arithmetic.argtypes = [C.c_void_p, C.c_int]
        L.ro_set_sum_topology.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_void_p]
        L.ro_set_factor_arithmetic.argtypes = [C.c_void_p, C.c_void_p]
        L.ro_set_lds_topology.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_void_p]
        L.ro_set_wg_topology.argtypes = [C.c_void_p, C.c_int]
        L.ro_set_group_topology.argtypes = [C.c_void_p, C.c_int]
        L.ro_eval_device_grid.restype = C.c_double
        L.ro_eval_device_grid.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int]
        L.ro_eval_device_ba.restype = C.c_double
        L.ro_eval_device_ba.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int]
        L.ro_eval_grad_device_ba.restype = C.c_double
        L.ro_eval_grad_device_ba.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.ro_set_stream_topology.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.ro_set_trig.argtypes = [C.c_void_p, C.c_void_p]
        L.ro_set_ptm_round_slots.argtypes = [C.c_void_p, C.c_int]
        L.ro_set_ptm_local.argtypes = [C.c_void_p, C.c_void_p]
        L.ro_set_ptm_topology.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
        L.ro_ba_factor_grad_device.restype = C.c_double
        L.ro_ba_factor_grad_device.argtypes = [_f64p, C.c_double, C.c_double, _f64p]
        L.ro_sincos_angle.argtypes = [C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.ro_create_ba.restype = C.c_void_p
        L.ro_create_ba.argtypes = [C.c_int64, _f64p, _f64p, _f64p, C.c_int64, _i64p, _i64p, _f64p]
        L.ro_create_nlp.restype = C.c_void_p
        L.ro_create_nlp.argtypes = [C.c_int64, _f64p, _f64p, _f64p, C.c_int64, _f64p, _i64p,
                                    _i64p, _f64p, _f64p, _u8p]
        L.ro_nlp_set_exponential.restype = C.c_int
        L.ro_nlp_set_exponential.argtypes = [C.c_void_p, C.c_void_p]
        L.ro_destroy.argtypes = [C.c_void_p]
        L.ro_set_emulate_stale_cache.argtypes = [C.c_void_p, C.c_int]
        L.ro_set_sum_order.argtypes = [C.c_void_p, C.c_int]
        L.ro_assign.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, _f64p]
        L.ro_get_x.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, _f64p]
        L.ro_eval_factors.restype = C.c_double
        L.ro_eval_factors.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
        L.ro_compute_gradient.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, _f64p, C.c_int]
        L.ro_eval_each.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, _f64p]
        L.ro_grad_each_ba.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, _f64p]
        L.ro_cgd_optimize.argtypes = [C.c_void_p, C.c_int64, _i64p, C.c_int64, C.c_void_p, _f64p,
                                      C.c_int32, C.c_double, C.c_int, C.POINTER(_Result)]
        L.ro_cgd_replay.argtypes = [C.c_void_p, C.c_int64, _i64p, C.c_int64, C.c_void_p, _f64p, C.c_int32,
                                    C.c_double, _f64p, C.c_int64, C.c_void_p, C.c_int32, C.c_void_p,
                                    C.POINTER(ReplayReport)]
        L.ro_cgd_record.restype = C.c_int64
        L.ro_cgd_record.argtypes = [C.c_void_p, C.c_int64, _i64p, C.c_int64, C.c_void_p, _f64p, C.c_int32,
                                    C.c_double, _f64p, C.c_int64]
        L.ro_frprmn.restype = C.c_int
        L.ro_frprmn.argtypes = [C.c_int, _f64p, FUNC_CB, GRAD_CB, C.c_void_p, C.c_int, C.c_double,
                                C.POINTER(C.c_double), C.POINTER(C.c_int)]
        L.ro_resjac_each_ba.restype = None
        L.ro_resjac_each_ba.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
        L.ro_components.restype = C.c_int64
        L.ro_components.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


def _opt_i64(a):
    if a is None:
        return None, None
    a = np.ascontiguousarray(a, dtype=np.int64)
    return a, a.ctypes.data_as(C.c_void_p)


class OracleProblem:
    """CPU oracle instance for one rdis_amd.problems.PackedProblem."""

    def __init__(self, pp, emulate_stale_cache: bool = True, derivative: str = "refchain", sum_order: str = "list",
                 arithmetic: str = "reference"):
        """derivative: "refchain" = the reference's forward chain operation by operation
        (BundleAdjustmentFactor.cpp:351-554; reproduces the reference's recorded runs bit for bit),
        "adjoint" = the independent reverse sweep (the derivation the device kernels use),
        "adjoint_device" = that sweep in the device's association (RO_BA_DERIV_ADJOINT_DEVICE).
        arithmetic: "reference" (every pin is made with it) or "device" = the named switches of rdis_oracle.h
        (RO_ARITH_RECIPROCAL | RO_ARITH_SINCOS_ANGLE) -- with derivative "adjoint_device" the factor arithmetic of the
        device's parity option (plan option factor_rounding = 1), bit for bit."""
        L = lib()
        self.pp = pp
        x0 = np.ascontiguousarray(pp.x0, dtype=np.float64)
        lo = np.ascontiguousarray(pp.lo, dtype=np.float64)
        hi = np.ascontiguousarray(pp.hi, dtype=np.float64)
        if pp.kind == 0:
            self.h = L.ro_create_ba(pp.nvars, x0, lo, hi, pp.nfac,
                                    np.ascontiguousarray(pp.cam_vid0), np.ascontiguousarray(pp.pt_vid0),
                                    np.ascontiguousarray(pp.obs.reshape(-1)))
        else:
            self.h = L.ro_create_nlp(pp.nvars, x0, lo, hi, pp.nfac, pp.coeff, pp.rowptr, pp.vid,
                                     pp.expo, pp.cons, pp.sine)
        L.ro_set_emulate_stale_cache(self.h, int(emulate_stale_cache))
        L.ro_set_sum_order(self.h, {"list": 0, "pairwise": 1}[sum_order])   # (an experiment's switch; "list" is the reference's)
        self.set_derivative(derivative)
        self.set_arithmetic(arithmetic)

    @classmethod
    def device_parity(cls, pp, emulate_stale_cache: bool = True):
        """the CPU side of the end-to-end == tests: the reference's algorithm, sums and stale factor cache with the three
        named last-place differences of the device's factor arithmetic switched on"""
        return cls(pp, emulate_stale_cache=emulate_stale_cache, derivative="adjoint_device", arithmetic="device")

    @classmethod
    def device_default(cls, pp, free_vid=None, fac=None, lanes_per_workgroup: int = 128):
        """the CPU side of the == test of the DEFAULT cooperative path (full ladybug's headline solve): the three switches of the
        device's factor arithmetic, no stale cache, and the cooperative solvers' sum trees (RO_SUM_TOPOLOGY_COOPERATIVE) for the
        component (free_vid, fac) -- None: all variables, all factors"""
        o = cls(pp, emulate_stale_cache=False, derivative="adjoint_device", arithmetic="device")
        o.set_cooperative_topology(free_vid, fac, lanes_per_workgroup)
        return o

    @classmethod
    def device_lds_default(cls, pp, free_vid=None, fac=None, threads: int = 0):
        """the CPU side of the == test of the DEFAULT LDS-resident path (BASELINE configs 3 and 5-S): the device's own factor
        arithmetic with its fused multiply-adds (factors.hpp compiled for the host: factors_host()), no stale cache, and that
        solver's sum trees for a workgroup of `threads` lanes (0: the dispatcher's rule for a launch whose largest component has
        this many factors -- 64 up to 64 factors, 128 up to 128, 256 up to 256)"""
        o = cls(pp, emulate_stale_cache=False)
        fv = np.arange(pp.nvars, dtype=np.int64) if free_vid is None else np.asarray(free_vid, dtype=np.int64)
        fc = np.arange(pp.nfac, dtype=np.int64) if fac is None else np.asarray(fac, dtype=np.int64)
        if threads == 0:
            threads = 64 if len(fc) <= 64 else 128 if len(fc) <= 128 else 256
        cams = np.unique(pp.cam_vid0[fc])
        pts = np.unique(pp.pt_vid0[fc])
        slots = np.concatenate([(cams[:, None] + np.arange(9)[None, :]).ravel(), (pts[:, None] + np.arange(3)[None, :]).ravel()]).astype(np.int64)
        o._fh = factors_host()
        lib().ro_set_factor_arithmetic(o.h, C.addressof(o._fh[1]))
        o._slots = np.ascontiguousarray(slots)
        lib().ro_set_lds_topology(o.h, int(threads), o._slots.shape[0], o._slots.ctypes.data_as(C.c_void_p))
        return o

    @classmethod
    def device_ptm_default(cls, pp, fac=None, threads: int = 768, group: int = 1, slots_per_block: int = 2, wide: bool = False,
                           round_slots: int = 0, lds_limit: int = 160 * 1024 - 4096, local_cus: int = 0):
        """the CPU side of the == test of the DEFAULT point-major streaming path (BASELINE config 5-L: a component too large for the
        LDS-resident solver, one workgroup of `threads` lanes, or a group of `group` of them; wide: one large component on a large share of the device, groups of up
        to 512 workgroups of 512 lanes; local_cus = the device's compute units: such a group with LOCAL camera numbering, the group size
        and the workgroups' chunk ranges by rdis_hip.hip's rule): the device's own factor arithmetic (factors_host(): the vector form
        for the gradient, the matrix form for the trials), no stale cache, that solver's layout and sum trees"""
        o = cls(pp, emulate_stale_cache=False)
        fc = np.arange(pp.nfac, dtype=np.int64) if fac is None else np.asarray(fac, dtype=np.int64)
        wg_chunk0 = None
        if local_cus:   # a wide group with LOCAL camera numbering: K, the order and the workgroups' chunk ranges follow from the data
            cams, pts, wg_chunk0 = ptm_point_order(pp.cam_vid0[fc], pp.pt_vid0[fc], local_cus=local_cus)
            wide, group, threads = True, len(wg_chunk0) - 1, 512
        else:
            cams, pts = ptm_point_order(pp.cam_vid0[fc], pp.pt_vid0[fc], wide=wide)
        o._fh = factors_host()
        lib().ro_set_factor_arithmetic(o.h, C.addressof(o._fh[1]))
        o._cams, o._pts = np.ascontiguousarray(cams), np.ascontiguousarray(pts)
        lib().ro_set_ptm_topology(o.h, int(threads), int(slots_per_block), -int(group) if wide else int(group), len(cams), o._cams.ctypes.data_as(C.c_void_p),
                                  len(pts), o._pts.ctypes.data_as(C.c_void_p), C.addressof(o._fh[2]))
        if wg_chunk0 is not None:
            o._wg_chunk0 = np.ascontiguousarray(wg_chunk0)
            lib().ro_set_ptm_local(o.h, o._wg_chunk0.ctypes.data_as(C.c_void_p))
        if round_slots == 0 and wg_chunk0 is not None:
            round_slots = 2   # (a workgroup holds a few cameras only: the staging rows always fit)
        if round_slots == 0:
            # rdis_hip.hip ptm_round_slots_for: a gradient round stages two slots where the LDS holds 2 x threads rows of nine doubles
            # beside the cameras' vectors and records (ptm_api.hpp: ptm_bytes_for) -- workgroups of up to 512 lanes only; those of 256 stand two to a compute unit
            ncb = len(cams)
            lds = ncb * 10 * (7 * 8 + 4) + ncb * (10 + 2 * 18) * 8 + (2 * threads + 1) * 72 + ((2 * ((ncb + 2) & ~1) * 2 + 7) & ~7) + 64
            round_slots = 2 if threads <= 512 and (2 if threads <= 256 else 1) * lds <= lds_limit else 1
        lib().ro_set_ptm_round_slots(o.h, int(round_slots))
        return o

    @classmethod
    def device_group_default(cls, pp, lanes: int = 16):
        """the CPU side of the == test of the solver of TINY components (solver_quad.hpp: a point against constant cameras and the
        like, at most four free variables; `lanes` = 16 a component from 4096 of them in a launch, 4 from 16384): the device's factor
        arithmetic (factors_host()), no stale cache, that solver's sums"""
        o = cls(pp, emulate_stale_cache=False)
        o._fh = factors_host()
        lib().ro_set_factor_arithmetic(o.h, C.addressof(o._fh[1]))
        lib().ro_set_group_topology(o.h, int(lanes))
        return o

    @classmethod
    def device_wg_default(cls, pp, free_vid=None, fac=None, threads: int = 0, grid_workgroups: int = 0):
        """the CPU side of the == test of the DEFAULT path of BASELINE configs 1 and 2 (nonlinear-product functions: the plain
        one-workgroup solver, solver_wg.hpp): the device's sine / cosine (factors.hpp for the host: fh_sincos) and its third and
        fourth power, no stale cache, that solver's sums for a workgroup of `threads` lanes (0: the dispatcher's rule -- by
        max(factors, variables / 4): 64 up to 64, 128, 256, 512 up to 512, else 768)"""
        o = cls(pp, emulate_stale_cache=False)
        fv = np.arange(pp.nvars, dtype=np.int64) if free_vid is None else np.asarray(free_vid, dtype=np.int64)
        fc = np.arange(pp.nfac, dtype=np.int64) if fac is None else np.asarray(fac, dtype=np.int64)
        if threads == 0:
            mf = max(len(fc), len(fv) // 4)
            threads = 64 if mf <= 64 else 128 if mf <= 128 else 256 if mf <= 256 else 512 if mf <= 512 else 768
        o._fh = factors_host()
        if pp.kind == 0:   # (KIND_BA: bundle adjustment on the fallback solver: the batch solvers' factor arithmetic)
            lib().ro_set_factor_arithmetic(o.h, C.addressof(o._fh[1]))
        else:
            lib().ro_set_trig(o.h, C.cast(o._fh[0].fh_sincos, C.c_void_p))
            lib().ro_set_arithmetic(o.h, 4)
        if grid_workgroups > 0:   # the GRID solver (solver_stream.hpp): that many workgroups of 512 lanes on the one component
            lib().ro_set_stream_topology(o.h, 512, int(grid_workgroups))
        else:
            lib().ro_set_wg_topology(o.h, int(threads))
        return o

    def set_cooperative_topology(self, free_vid=None, fac=None, lanes_per_workgroup: int = 128) -> None:
        """wave-owned variables as rdis_hip.hip's prepare_partition picks them: fed by more than 48 listed partials (bundle
        adjustment: one per listed factor that reads the variable), the longest runs first, ties in list order -- as many as the
        component's cooperative group has waves: workgroups = max(ceil(max(factors, variables) / lanes), and for groups of fewer
        than 16 workgroups min(ceil(long runs / waves per workgroup), 2 x that + 2)), lanes = 128 factor lanes per workgroup in
        the pipelined layout (the default), 256 in the plain one"""
        pp = self.pp
        fv = np.arange(pp.nvars, dtype=np.int64) if free_vid is None else np.asarray(free_vid, dtype=np.int64)
        fc = np.arange(pp.nfac, dtype=np.int64) if fac is None else np.asarray(fac, dtype=np.int64)
        cnt = np.zeros(pp.nvars, dtype=np.int64)
        for k in range(9):
            np.add.at(cnt, pp.cam_vid0[fc] + k, 1)
        for k in range(3):
            np.add.at(cnt, pp.pt_vid0[fc] + k, 1)
        runs = cnt[fv]
        longv = np.nonzero(runs > 48)[0]
        longv = longv[np.argsort(-runs[longv], kind="stable")]
        wpw = lanes_per_workgroup // 64
        need = -(-max(len(fc), len(fv)) // lanes_per_workgroup)
        for_long = min(-(-len(longv) // wpw), 2 * need + 2) if need < 16 else 0
        waves = max(1, need, for_long) * wpw
        wave = np.ascontiguousarray(fv[longv[:waves]], dtype=np.int64)
        lib().ro_set_sum_topology(self.h, 1, wave.shape[0], wave.ctypes.data_as(C.c_void_p))

    def set_arithmetic(self, arithmetic) -> None:
        flags = {"reference": 0, "device": 3, "reciprocal": 1, "sincos_angle": 2}.get(arithmetic, arithmetic)
        lib().ro_set_arithmetic(self.h, int(flags))

    def set_exponential(self, use_exp) -> None:
        """useExponential per factor (NonlinearProductFactor.cpp:140); values only, None clears"""
        u = None if use_exp is None else np.ascontiguousarray(use_exp, dtype=np.uint8)
        if lib().ro_nlp_set_exponential(self.h, None if u is None else u.ctypes.data_as(C.c_void_p)):
            raise ValueError("not a nonlinear-product problem")

    def set_derivative(self, derivative: str) -> None:
        lib().ro_set_ba_derivative(self.h, {"refchain": 0, "adjoint": 1, "adjoint_device": 2}[derivative])

    def __del__(self):
        if getattr(self, "h", None):
            lib().ro_destroy(self.h)
            self.h = None

    def assign(self, vid, val):
        v, vp = _opt_i64(vid)
        val = np.ascontiguousarray(val, dtype=np.float64)
        lib().ro_assign(self.h, val.shape[0], vp, val)

    def get_x(self, vid=None):
        v, vp = _opt_i64(vid)
        n = self.pp.nvars if v is None else v.shape[0]
        out = np.empty(n)
        lib().ro_get_x(self.h, n, vp, out)
        return out

    def resjac_each_ba(self, fac=None):
        """(res [nf, 2], J [nf, 2, 12]): pixel residuals and their Jacobian rows per listed factor"""
        f, fp = _opt_i64(fac)
        n = self.pp.nfac if f is None else f.shape[0]
        res, J = np.empty((n, 2)), np.empty((n, 2, 12))
        lib().ro_resjac_each_ba(self.h, n, fp, res.ctypes.data_as(C.c_void_p), J.ctypes.data_as(C.c_void_p))
        return res, J

    def components(self, assigned):
        """(free_ptr, free_vid, fac_ptr, fac_id) of the connected components left when the variables
        with assigned[v] != 0 are fixed"""
        a = np.ascontiguousarray(assigned, dtype=np.uint8)
        n, f = self.pp.nvars, self.pp.nfac
        fp, fv = np.zeros(n + 1, np.int64), np.zeros(max(n, 1), np.int64)
        cp, ci = np.zeros(n + 1, np.int64), np.zeros(max(f, 1), np.int64)
        v = lambda x: x.ctypes.data_as(C.c_void_p)
        nc = int(lib().ro_components(self.h, v(a), v(fp), v(fv), v(cp), v(ci)))
        return fp[:nc + 1].copy(), fv[:fp[nc]].copy(), cp[:nc + 1].copy(), ci[:cp[nc]].copy()

    def eval(self, fac=None) -> float:
        f, fp = _opt_i64(fac)
        return lib().ro_eval_factors(self.h, self.pp.nfac if f is None else f.shape[0], fp)

    @classmethod
    def device_eval(cls, pp):
        """the CPU side of the == test of the PUBLIC evaluation entry points on bundle adjustment (rdis_hip_eval, rdis_hip_eval_grad):
        the device's factor arithmetic (factors_host()); eval_device / eval_grad_device add as those kernels add"""
        o = cls(pp, emulate_stale_cache=False)
        o._fh = factors_host()
        lib().ro_set_factor_arithmetic(o.h, C.addressof(o._fh[1]))
        return o

    def eval_device(self, fac=None, lanes: int = 512) -> float:
        f, fp = _opt_i64(fac)
        return lib().ro_eval_device_ba(self.h, self.pp.nfac if f is None else f.shape[0], None if f is None else f.ctypes.data_as(C.c_void_p), lanes)

    def eval_device_grid(self, fac=None, compute_units: int = 256) -> float:
        """rdis_hip_eval's value on a nonlinear-product problem (the two-pass form's grid of 256-lane workgroups)"""
        f, fp = _opt_i64(fac)
        nf = self.pp.nfac if f is None else f.shape[0]
        return lib().ro_eval_device_grid(self.h, nf, None if f is None else f.ctypes.data_as(C.c_void_p), max(1, min(-(-nf // 256), 8 * compute_units)))

    def eval_grad_device(self, fac=None, lanes: int = 512, compute_units: int = 256):
        """(value, gradient) as rdis_hip_eval_grad forms them: chunks of 512 listed factors, tiles of max(1, min(8, chunks / (8 x
        compute units))) chunks (rdis_hip.hip: grad plan)"""
        f, fp = _opt_i64(fac)
        nf = self.pp.nfac if f is None else f.shape[0]
        nchunks = -(-nf // lanes)
        g = np.zeros(self.pp.nvars)
        val = lib().ro_eval_grad_device_ba(self.h, nf, None if f is None else f.ctypes.data_as(C.c_void_p), g.ctypes.data_as(C.c_void_p), lanes,
                                           max(1, min(8, nchunks // (8 * compute_units))))
        return val, g

    def gradient(self, fac=None, merge: bool = False) -> np.ndarray:
        f, fp = _opt_i64(fac)
        g = np.empty(self.pp.nvars)
        lib().ro_compute_gradient(self.h, self.pp.nfac if f is None else f.shape[0], fp, g, int(merge))
        return g

    def eval_each(self, fac=None) -> np.ndarray:
        f, fp = _opt_i64(fac)
        n = self.pp.nfac if f is None else f.shape[0]
        out = np.empty(n)
        lib().ro_eval_each(self.h, n, fp, out)
        return out

    def grad_each_ba(self, fac=None) -> np.ndarray:
        f, fp = _opt_i64(fac)
        n = self.pp.nfac if f is None else f.shape[0]
        out = np.empty(n * 12)
        lib().ro_grad_each_ba(self.h, n, fp, out)
        return out.reshape(n, 12)

    def cgd(self, free_vid=None, fac=None, x=None, maxiters: int = 50, ftol: float = 3e-8,
            merge: bool = False) -> CGDResult:
        fv = np.arange(self.pp.nvars, dtype=np.int64) if free_vid is None else \
            np.ascontiguousarray(free_vid, dtype=np.int64)
        f, fp = _opt_i64(fac)
        xv = self.get_x(fv) if x is None else np.array(x, dtype=np.float64)
        res = _Result()
        lib().ro_cgd_optimize(self.h, fv.shape[0], fv, self.pp.nfac if f is None else f.shape[0], fp,
                              xv, maxiters, ftol, int(merge), C.byref(res))
        return CGDResult(res.fret, res.delta, res.finit, res.iters, res.status, res.nfeval,
                         res.ngeval, xv)


def _replay_args(self, free_vid, fac, x):
    fv = np.arange(self.pp.nvars, dtype=np.int64) if free_vid is None else \
        np.ascontiguousarray(free_vid, dtype=np.int64)
    f, fp = _opt_i64(fac)
    xv = self.get_x(fv) if x is None else np.ascontiguousarray(x, dtype=np.float64)
    return fv, f, fp, xv


def _replay(self, trace, free_vid=None, fac=None, x=None, maxiters=50, ftol=3e-8, vdump=None):
    """feed a device trace ([n,4] records) through the oracle's solver -> ReplayReport;
    vdump [iters,2,nfree]: the device's p / xi at the start of each line search"""
    fv, f, fp, xv = _replay_args(self, free_vid, fac, x)
    tr = np.ascontiguousarray(trace, dtype=np.float64).reshape(-1)
    rep = ReplayReport()
    xe = np.empty(fv.shape[0])
    vd = None if vdump is None else np.ascontiguousarray(vdump, dtype=np.float64)
    lib().ro_cgd_replay(self.h, fv.shape[0], fv, self.pp.nfac if f is None else f.shape[0], fp, xv,
                        maxiters, ftol, tr, tr.shape[0] // 4,
                        None if vd is None else vd.ctypes.data_as(C.c_void_p), 0 if vd is None else vd.shape[0],
                        xe.ctypes.data_as(C.c_void_p), C.byref(rep))
    rep.x_end = xe
    return rep


def _record(self, free_vid=None, fac=None, x=None, maxiters=50, ftol=3e-8, cap=1 << 16):
    fv, f, fp, xv = _replay_args(self, free_vid, fac, x)
    tr = np.zeros(cap * 4)
    n = lib().ro_cgd_record(self.h, fv.shape[0], fv, self.pp.nfac if f is None else f.shape[0], fp, xv,
                            maxiters, ftol, tr, cap)
    return tr.reshape(cap, 4)[:min(n, cap)], n


OracleProblem.replay = _replay
OracleProblem.record = _record


def ba_factor_eval(vals, ox, oy) -> float:
    return lib().ro_ba_factor_eval(np.ascontiguousarray(vals, dtype=np.float64), ox, oy)


def ba_factor_grad(vals, ox, oy):
    g = np.empty(12)
    e = lib().ro_ba_factor_grad(np.ascontiguousarray(vals, dtype=np.float64), ox, oy, g)
    return e, g


def ba_factor_grad_device(vals, ox, oy):
    """value + partials in the device's parity-option arithmetic (all three named switches on)"""
    g = np.empty(12)
    e = lib().ro_ba_factor_grad_device(np.ascontiguousarray(vals, dtype=np.float64), ox, oy, g)
    return e, g


def ba_factor_grad_ref(vals, ox, oy):
    """the reference's forward-chain derivative (BundleAdjustmentFactor.cpp:351-554)"""
    g = np.empty(12)
    e = lib().ro_ba_factor_grad_ref(np.ascontiguousarray(vals, dtype=np.float64), ox, oy, g)
    return e, g


def _wrap_callbacks(f, df, n):
    def cf(_ctx, xp):
        return float(f(np.ctypeslib.as_array(xp, shape=(n,)).copy()))

    def cg(_ctx, xp, gp):
        g = np.asarray(df(np.ctypeslib.as_array(xp, shape=(n,)).copy()), dtype=np.float64)
        np.ctypeslib.as_array(gp, shape=(n,))[:] = g
    return FUNC_CB(cf), GRAD_CB(cg)


def frprmn(f, df, x0, maxiters=50, ftol=3e-8):
    """restated minimiser on python callbacks -> (reason, x, fret, iter)"""
    x = np.array(x0, dtype=np.float64)
    cf, cg = _wrap_callbacks(f, df, x.shape[0])
    fret, it = C.c_double(), C.c_int()
    rc = lib().ro_frprmn(x.shape[0], x, cf, cg, None, maxiters, ftol, C.byref(fret), C.byref(it))
    return rc, x, fret.value, it.value


_ref = None


def ref_available() -> bool:
    build()
    return os.path.exists(_REF)


def ref_frprmn(f, df, x0, maxiters=50, ftol=3e-8):
    """the REFERENCE's nrc::Frprmn (oracle/_ref) on python callbacks"""
    global _ref
    if _ref is None:
        _ref = C.CDLL(_REF)
        _ref.ref_frprmn.restype = C.c_int
        _ref.ref_frprmn.argtypes = [C.c_int, _f64p, FUNC_CB, GRAD_CB, C.c_void_p, C.c_int,
                                    C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_int)]
    x = np.array(x0, dtype=np.float64)
    cf, cg = _wrap_callbacks(f, df, x.shape[0])
    fret, it = C.c_double(), C.c_int()
    rc = _ref.ref_frprmn(x.shape[0], x, cf, cg, None, maxiters, ftol, C.byref(fret), C.byref(it))
    return rc, x, fret.value, it.value
