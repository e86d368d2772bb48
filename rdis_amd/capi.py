"""ctypes binding of the C ABI in include/rdis_hip.h (rdis_amd/lib/librdis_hip.so).

There is no fallback of any kind: if the HIP library is missing or a call fails
this module raises.  The oracle (oracle/) is never imported from here.
"""
from __future__ import annotations

import ctypes as C
import os
import sys
from dataclasses import dataclass
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "librdis_hip.so")

@dataclass
class LmResult:
    x: np.ndarray
    fret: float
    delta: float
    iters: int
    stop: int
    nfev: int
    njev: int
    nsolve: int
    mu: float
    camera_blocks: int
    point_blocks: int
    history: np.ndarray   # [nsolve, 4]: mu, |Dp|^2, f(trial), accepted


EXIT_NAMES = {0: "ftol", 1: "gtol", 2: "gg==0", 3: "itmax", 4: "dbrent-itmax", 5: "nan",
              6: "empty", 7: "sync-timeout"}
STATUS_ROLLED_BACK = 0x100

_ERRORS = {-1: "EINVAL", -2: "ENOMEM", -3: "EDEVICE", -4: "EOVERLAP", -5: "ERANGE"}


class RdisHipError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"rdis_hip error {_ERRORS.get(code, code)}: {msg}")
        self.code = code


_f64 = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_vp = C.c_void_p
_i64 = C.c_int64

# every symbol include/rdis_hip.h declares: (restype, argtypes)
SYMBOLS = {
    "rdis_hip_abi_version": (C.c_int, []),
    "rdis_hip_device_count": (C.c_int, []),
    "rdis_hip_create": (C.c_int, [C.c_int, C.POINTER(_vp)]),
    "rdis_hip_destroy": (None, [_vp]),
    "rdis_hip_last_error": (C.c_char_p, [_vp]),
    "rdis_hip_set_stream": (C.c_int, [_vp, _vp]),
    "rdis_hip_synchronize": (C.c_int, [_vp]),
    "rdis_hip_copy_to_host": (C.c_int, [_vp, _vp, _vp, _i64]),
    "rdis_hip_upload_ba": (C.c_int, [_vp, _i64, _vp, _vp, _vp, _i64, _vp, _vp, _vp, C.POINTER(_vp)]),
    "rdis_hip_upload_nlp": (C.c_int, [_vp, _i64, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, C.POINTER(_vp)]),
    "rdis_hip_nlp_set_exponential": (C.c_int, [_vp, _vp]),
    "rdis_hip_free_problem": (None, [_vp]),
    "rdis_hip_set_x": (C.c_int, [_vp, _i64, _vp, _vp]),
    "rdis_hip_get_x": (C.c_int, [_vp, _i64, _vp, _vp]),
    "rdis_hip_eval": (C.c_int, [_vp, _i64, _vp, C.POINTER(C.c_double)]),
    "rdis_hip_eval_grad": (C.c_int, [_vp, _i64, _vp, C.POINTER(C.c_double), _vp]),
    "rdis_hip_eval_grad_device": (C.c_int, [_vp, _i64, _vp, C.POINTER(_vp), C.POINTER(_vp)]),
    "rdis_hip_eval_each": (C.c_int, [_vp, _i64, _vp, _vp]),
    "rdis_hip_grad_each_ba": (C.c_int, [_vp, _i64, _vp, _vp]),
    "rdis_hip_set_factor_rounding": (C.c_int, [_vp, C.c_int32]),
    "rdis_hip_cgd_batch": (C.c_int, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, C.c_int32, C.c_double,
                                      _vp, _vp, _vp, _vp, _vp, _vp]),
    "rdis_hip_plan_create": (C.c_int, [_vp, _i64, _vp, _vp, _vp, _vp, C.POINTER(_vp)]),
    "rdis_hip_plan_destroy": (None, [_vp]),
    "rdis_hip_plan_set_start": (C.c_int, [_vp, _vp]),
    "rdis_hip_plan_solve": (C.c_int, [_vp, C.c_int32, C.c_double]),
    "rdis_hip_plan_fetch": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "rdis_hip_plan_objective_device": (C.c_int, [_vp, C.POINTER(_vp)]),
    "rdis_hip_comm_unique_id": (C.c_int, [_vp]),
    "rdis_hip_comm_create": (C.c_int, [_vp, C.c_int32, C.c_int32, _vp, C.POINTER(_vp)]),
    "rdis_hip_comm_create_all": (C.c_int, [C.c_int32, _vp, _vp]),
    "rdis_hip_comm_destroy": (None, [_vp]),
    "rdis_hip_allreduce_objective": (C.c_int, [_vp, _vp, C.POINTER(C.c_double)]),
    "rdis_hip_allreduce_objective_all": (C.c_int, [C.c_int32, _vp, _vp, C.POINTER(C.c_double)]),
    "rdis_hip_comm_allreduce_f64": (C.c_int, [_vp, _vp, C.c_int32, C.c_int32]),
    "rdis_hip_plan_set_option": (C.c_int, [_vp, C.c_char_p, _i64]),
    "rdis_hip_plan_last_kernel_ms": (C.c_int, [_vp, C.POINTER(C.c_double), C.POINTER(C.c_int32)]),
    "rdis_hip_plan_device_bytes": (C.c_int, [_vp, C.POINTER(C.c_int64)]),
    "rdis_hip_plan_get_info": (C.c_int, [_vp, C.c_char_p, C.POINTER(C.c_int64)]),
    "rdis_hip_plan_get_trace": (C.c_int, [_vp, _i64, _vp, _i64, C.POINTER(_i64)]),
    "rdis_hip_plan_get_vectors": (C.c_int, [_vp, _i64, _vp, _i64]),
    "rdis_hip_plan_debug_counters": (C.c_int, [_vp, _vp]),
    "rdis_hip_lm_optimize": (C.c_int, [_vp, C.c_int64, _vp, C.c_int64, _vp, _vp, C.c_int32, C.c_double, C.c_int32, _vp, _vp, _vp, _vp,
                                       C.c_int64, _vp]),
    "rdis_hip_components": (C.c_int, [_vp, _vp, _vp, _vp, _vp]),
    "rdis_hip_components_fetch": (C.c_int, [_vp, _vp, _vp, _vp, _vp]),
}

_lib: Optional[C.CDLL] = None
_live = []  # weak references to open handles, closed in order (plans, problems, contexts) at exit


def _register(obj):
    import weakref
    _live.append(weakref.ref(obj))


class _Cell:
    """A native handle and the handles that depend on it (context <- problems <- plans).  Closing a
    cell closes its dependents first, so the order in which Python finalises the wrapper objects
    (arbitrary inside a garbage cycle, e.g. the frames of a failed test) never leaves the library
    with a dangling pointer."""
    __slots__ = ("h", "kids", "destroy")

    def __init__(self, h, destroy, owner=None):
        self.h, self.kids, self.destroy = h, [], destroy
        if owner is not None:
            owner.kids = [k for k in owner.kids if k.h] + [self]

    def close(self):
        if self.h:
            for k in self.kids:
                k.close()
            self.kids = []
            self.destroy(self.h)
            self.h = None


def _close_all():
    # the HIP runtime must still be loaded when device memory and streams are released
    objs = [r() for r in _live]
    for cls in ("Comm", "Plan", "Problem", "Context"):
        for o in objs:
            if o is not None and type(o).__name__ == cls:
                try:
                    o.close()
                except Exception:
                    pass


import atexit  # noqa: E402

atexit.register(_close_all)


def load_library() -> C.CDLL:
    """dlopen the in-tree HIP library and bind every declared symbol."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RdisHipError(-3, f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               f"(make -C rdis_amd/csrc). There is no CPU fallback.")
    # If torch is (going to be) used in this process its bundled HIP runtime must be
    # the one this library binds to: both carry SONAME libamdhip64.so.7, so whichever
    # is loaded first serves both.  Import torch first when it is already requested.
    if "torch" in sys.modules or os.environ.get("RDIS_HIP_WITH_TORCH") == "1":
        import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    if lib.rdis_hip_abi_version() != 1:
        raise RdisHipError(-1, "ABI version mismatch")
    _lib = lib
    return lib


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(_vp)


def _f(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float64)


def _i(a) -> Optional[np.ndarray]:
    return None if a is None else np.ascontiguousarray(a, dtype=np.int64)


class Context:
    """rdis_hip_ctx: one per GPU."""

    def __init__(self, device: int = 0):
        self.lib = load_library()
        h = _vp()
        rc = self.lib.rdis_hip_create(device, C.byref(h))
        if rc:
            raise RdisHipError(rc, f"rdis_hip_create(device={device}) failed "
                                   f"({self.lib.rdis_hip_device_count()} HIP devices visible)")
        self._cell = _Cell(h, self.lib.rdis_hip_destroy)
        self.device = device
        _register(self)

    def check(self, rc: int):
        if rc:
            raise RdisHipError(rc, self.lib.rdis_hip_last_error(self.h).decode())

    def set_stream(self, hip_stream: Optional[int]):
        self.check(self.lib.rdis_hip_set_stream(self.h, _vp(hip_stream) if hip_stream else None))

    def synchronize(self):
        self.check(self.lib.rdis_hip_synchronize(self.h))

    def copy_to_host(self, dev_ptr: int, nbytes: int) -> bytes:
        buf = C.create_string_buffer(nbytes)
        self.check(self.lib.rdis_hip_copy_to_host(self.h, buf, _vp(dev_ptr), nbytes))
        return buf.raw

    @property
    def h(self):
        return self._cell.h

    def close(self):
        if getattr(self, "_cell", None):
            self._cell.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


@dataclass
class BatchResult:
    x: np.ndarray
    fret: np.ndarray
    delta: np.ndarray
    iters: np.ndarray
    status: np.ndarray
    nfeval: np.ndarray
    ngeval: np.ndarray

    @property
    def exit_reason(self):
        return self.status & 0xFF

    @property
    def rolled_back(self):
        return (self.status & STATUS_ROLLED_BACK) != 0


class Problem:
    """rdis_hip_problem built from an rdis_amd.problems.PackedProblem."""

    def __init__(self, ctx: Context, pp):
        self.ctx, self.pp = ctx, pp
        lib = ctx.lib
        h = _vp()
        x0, lo, hi = _f(pp.x0), _f(pp.lo), _f(pp.hi)
        if pp.kind == 0:
            cam, pt, obs = _i(pp.cam_vid0), _i(pp.pt_vid0), _f(pp.obs).reshape(-1)
            rc = lib.rdis_hip_upload_ba(ctx.h, pp.nvars, _ptr(x0), _ptr(lo), _ptr(hi), pp.nfac,
                                        _ptr(cam), _ptr(pt), _ptr(obs), C.byref(h))
        else:
            rc = lib.rdis_hip_upload_nlp(ctx.h, pp.nvars, _ptr(x0), _ptr(lo), _ptr(hi), pp.nfac,
                                         _ptr(_f(pp.coeff)), _ptr(_i(pp.rowptr)), _ptr(_i(pp.vid)),
                                         _ptr(_f(pp.expo)), _ptr(_f(pp.cons)),
                                         _ptr(np.ascontiguousarray(pp.sine, dtype=np.uint8)), C.byref(h))
        ctx.check(rc)
        self._cell = _Cell(h, lib.rdis_hip_free_problem, ctx._cell)
        self.nvars, self.nfac = pp.nvars, pp.nfac
        _register(self)

    @property
    def h(self):
        return self._cell.h

    def close(self):
        if getattr(self, "_cell", None):
            self._cell.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_exponential(self, use_exp):
        """NonlinearProductFactor's useExponential per factor (values only; None clears)"""
        u = None if use_exp is None else np.ascontiguousarray(use_exp, dtype=np.uint8)
        if u is not None and u.shape[0] != self.nfac:
            raise ValueError("one flag per factor")
        self.ctx.check(self.ctx.lib.rdis_hip_nlp_set_exponential(self.h, _ptr(u)))

    def set_factor_rounding(self, mode: int):
        """how eval_each / grad_each_ba round: 0 = fused multiply-adds, 1 = like the reference's build (the parity option's)"""
        self.ctx.check(self.ctx.lib.rdis_hip_set_factor_rounding(self.h, int(mode)))

    def set_x(self, val, vid=None):
        val, vid = _f(val), _i(vid)
        self.ctx.check(self.ctx.lib.rdis_hip_set_x(self.h, val.shape[0], _ptr(vid), _ptr(val)))

    def lm_optimize(self, free_vid=None, fac_id=None, x=None, maxiters=25, ftol=3e-8, history=256, model=1, schur=0):
        """Levenberg-Marquardt over the listed free variables / factors (bundle adjustment; the
        least-squares problem of LMSubspaceOptimizer).  Returns an LmResult; the variables are left
        assigned to the result.  schur: 0 = Schur product by fill, 1 = dense on the matrix cores, 2 = block-sparse."""
        fv = np.arange(self.nvars, dtype=np.int64) if free_vid is None else _i(free_vid)
        fc = np.arange(self.nfac, dtype=np.int64) if fac_id is None else _i(fac_id)
        xin = None if x is None else _f(x).copy()
        out = np.zeros(2)
        info = np.zeros(8)
        hist = np.zeros((max(history, 1), 4))
        nh = np.zeros(1, dtype=np.int64)
        self.ctx.check(self.ctx.lib.rdis_hip_lm_optimize(self.h, fv.shape[0], _ptr(fv), fc.shape[0], _ptr(fc), _ptr(xin), maxiters, ftol, model | (schur << 4),
                                                         C.c_void_p(out.ctypes.data), C.c_void_p(out.ctypes.data + 8), _ptr(info),
                                                         _ptr(hist), hist.shape[0], _ptr(nh)))
        return LmResult(x=self.get_x(fv), fret=float(out[0]), delta=float(out[1]), iters=int(info[0]), stop=int(info[1]),
                        nfev=int(info[2]), njev=int(info[3]), nsolve=int(info[4]), mu=float(info[5]),
                        camera_blocks=int(info[6]), point_blocks=int(info[7]), history=hist[:min(int(nh[0]), hist.shape[0])].copy())

    def components(self, assigned):
        """(free_ptr, free_vid, fac_ptr, fac_id) of the connected components left when the variables
        with assigned[v] != 0 are fixed (Component::createChildren); device union-find"""
        a = np.ascontiguousarray(assigned, dtype=np.uint8)
        if a.shape[0] != self.nvars:
            raise ValueError("assigned must have one entry per variable")
        sizes = np.zeros(3, dtype=np.int64)
        sp = sizes.ctypes.data
        self.ctx.check(self.ctx.lib.rdis_hip_components(self.h, _ptr(a), C.c_void_p(sp), C.c_void_p(sp + 8), C.c_void_p(sp + 16)))
        nc, nfree, nfac = (int(v) for v in sizes)
        free_ptr, fac_ptr = np.empty(nc + 1, np.int64), np.empty(nc + 1, np.int64)
        free_vid, fac_id = np.empty(nfree, np.int64), np.empty(nfac, np.int64)
        self.ctx.check(self.ctx.lib.rdis_hip_components_fetch(self.h, _ptr(free_ptr), _ptr(free_vid), _ptr(fac_ptr), _ptr(fac_id)))
        return free_ptr, free_vid, fac_ptr, fac_id

    def get_x(self, vid=None) -> np.ndarray:
        vid = _i(vid)
        n = self.nvars if vid is None else vid.shape[0]
        out = np.empty(n)
        self.ctx.check(self.ctx.lib.rdis_hip_get_x(self.h, n, _ptr(vid), _ptr(out)))
        return out

    def _nf(self, fac):
        fac = _i(fac)
        return fac, (self.nfac if fac is None else fac.shape[0])

    def eval(self, fac=None) -> float:
        fac, nf = self._nf(fac)
        f = C.c_double()
        self.ctx.check(self.ctx.lib.rdis_hip_eval(self.h, nf, _ptr(fac), C.byref(f)))
        return f.value

    def eval_grad(self, fac=None, out=None):
        """(value, gradient); out: a float64 array of nvars to receive the gradient (a caller that evaluates repeatedly keeps
        one: a fresh 49 MB array costs more in page faults than the copy from the device)"""
        fac, nf = self._nf(fac)
        f = C.c_double()
        g = np.empty(self.nvars) if out is None else out
        if g.dtype != np.float64 or g.shape != (self.nvars,) or not g.flags.c_contiguous:
            raise ValueError("out must be a contiguous float64 array of nvars")
        self.ctx.check(self.ctx.lib.rdis_hip_eval_grad(self.h, nf, _ptr(fac), C.byref(f), _ptr(g)))
        return f.value, g

    def eval_grad_device(self, fac=None):
        """value and gradient left on the device: (pointer to one double, pointer to nvars doubles), valid until the
        next evaluation call on this problem; asynchronous on the context's stream"""
        fac, nf = self._nf(fac)
        fd, gd = _vp(), _vp()
        self.ctx.check(self.ctx.lib.rdis_hip_eval_grad_device(self.h, nf, _ptr(fac), C.byref(fd), C.byref(gd)))
        return fd.value, gd.value

    def eval_each(self, fac=None) -> np.ndarray:
        fac, nf = self._nf(fac)
        out = np.empty(nf)
        self.ctx.check(self.ctx.lib.rdis_hip_eval_each(self.h, nf, _ptr(fac), _ptr(out)))
        return out

    def grad_each_ba(self, fac=None) -> np.ndarray:
        fac, nf = self._nf(fac)
        out = np.empty(nf * 12)
        self.ctx.check(self.ctx.lib.rdis_hip_grad_each_ba(self.h, nf, _ptr(fac), _ptr(out)))
        return out.reshape(nf, 12)

    def cgd_batch(self, free_ptr, free_vid, fac_ptr, fac_id, x, maxiters=50, ftol=3e-8) -> BatchResult:
        free_ptr, free_vid, fac_ptr, fac_id = _i(free_ptr), _i(free_vid), _i(fac_ptr), _i(fac_id)
        nc = free_ptr.shape[0] - 1
        x = np.array(x, dtype=np.float64)
        r = BatchResult(x, np.empty(nc), np.empty(nc), np.empty(nc, np.int32), np.empty(nc, np.int32),
                        np.empty(nc, np.int64), np.empty(nc, np.int64))
        self.ctx.check(self.ctx.lib.rdis_hip_cgd_batch(
            self.h, nc, _ptr(free_ptr), _ptr(free_vid), _ptr(fac_ptr), _ptr(fac_id), _ptr(x), maxiters, ftol,
            _ptr(r.fret), _ptr(r.delta), _ptr(r.iters), _ptr(r.status), _ptr(r.nfeval), _ptr(r.ngeval)))
        return r


class Comm:
    """rdis_hip_comm: the communicator of the path's one collective (the objective's all-reduce over RCCL / xGMI).
    One rank per process and GPU; `unique_id()` on rank 0, its 128 bytes to every rank, `Comm(ctx, world, rank, id)` on all."""

    @staticmethod
    def unique_id() -> bytes:
        buf = (C.c_ubyte * 128)()
        rc = load_library().rdis_hip_comm_unique_id(buf)
        if rc:
            raise RdisHipError(rc, "rdis_hip_comm_unique_id (is librccl.so there?)")
        return bytes(buf)

    def __init__(self, ctx: Context, world: int, rank: int, uid: bytes):
        assert len(uid) == 128
        self.ctx, self.world, self.rank = ctx, world, rank
        h = _vp()
        buf = (C.c_ubyte * 128).from_buffer_copy(uid)
        ctx.check(ctx.lib.rdis_hip_comm_create(ctx.h, world, rank, buf, C.byref(h)))
        self._cell = _Cell(h, ctx.lib.rdis_hip_comm_destroy, owner=ctx._cell)   # (closed before its context)
        _register(self)

    @property
    def h(self):
        return self._cell.h

    def close(self):
        self._cell.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def allreduce(self, values, op: str = "sum") -> np.ndarray:
        """a few host doubles summed / maximised over the ranks (counters; barrier + maximum of a timed region)"""
        v = np.ascontiguousarray(np.atleast_1d(values), dtype=np.float64).copy()
        self.ctx.check(self.ctx.lib.rdis_hip_comm_allreduce_f64(self.h, _ptr(v), v.shape[0], {"sum": 0, "max": 1}[op]))
        return v

    def barrier(self):
        self.allreduce([0.0])


class Plan:
    """rdis_hip_plan: a decomposition resident on the device, solvable many times."""

    def __init__(self, prob: Problem, free_ptr=None, free_vid=None, fac_ptr=None, fac_id=None):
        pp = prob.pp
        if free_ptr is None:
            if pp.comp_free_ptr is None:
                pp.single_component()
            free_ptr, free_vid, fac_ptr, fac_id = pp.comp_free_ptr, pp.comp_free_vid, pp.comp_fac_ptr, pp.comp_fac_id
        self.prob, self.ctx = prob, prob.ctx
        self.free_ptr, self.free_vid = _i(free_ptr), _i(free_vid)
        self.fac_ptr, self.fac_id = _i(fac_ptr), _i(fac_id)
        self.ncomp = self.free_ptr.shape[0] - 1
        self.nfree = int(self.free_ptr[-1])
        h = _vp()
        self.ctx.check(self.ctx.lib.rdis_hip_plan_create(prob.h, self.ncomp, _ptr(self.free_ptr), _ptr(self.free_vid),
                                                         _ptr(self.fac_ptr), _ptr(self.fac_id), C.byref(h)))
        self._cell = _Cell(h, self.ctx.lib.rdis_hip_plan_destroy, prob._cell)
        _register(self)

    @property
    def h(self):
        return self._cell.h

    def close(self):
        if getattr(self, "_cell", None):
            self._cell.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, name: str, value: int):
        self.ctx.check(self.ctx.lib.rdis_hip_plan_set_option(self.h, name.encode(), int(value)))

    def set_start(self, x=None):
        x = None if x is None else _f(x)
        self.ctx.check(self.ctx.lib.rdis_hip_plan_set_start(self.h, _ptr(x)))

    def solve(self, maxiters=50, ftol=3e-8):
        """asynchronous launch on the context's stream"""
        self.ctx.check(self.ctx.lib.rdis_hip_plan_solve(self.h, maxiters, ftol))

    def fetch(self, want_x=True, out: Optional[BatchResult] = None) -> BatchResult:
        """the last solve's results; out: a BatchResult of an earlier fetch of this plan to write into -- a caller that solves a plan
        again keeps its arrays, as the reference's caller keeps its xval (fresh arrays cost a page fault per 4 KiB inside the copy: a
        24 MB solution 1.2 ms instead of 0.4)"""
        nc = self.ncomp
        r = out if out is not None else BatchResult(np.empty(self.nfree) if want_x else None, np.empty(nc), np.empty(nc),
                                                    np.empty(nc, np.int32), np.empty(nc, np.int32), np.empty(nc, np.int64),
                                                    np.empty(nc, np.int64))
        if out is not None and (r.fret.shape[0] != nc or (r.x is not None and r.x.shape[0] != self.nfree)):
            raise ValueError("fetch(out=...): the arrays are another plan's")
        self.ctx.check(self.ctx.lib.rdis_hip_plan_fetch(self.h, _ptr(r.x), _ptr(r.fret), _ptr(r.delta), _ptr(r.iters),
                                                        _ptr(r.status), _ptr(r.nfeval), _ptr(r.ngeval)))
        return r

    def objective(self) -> float:
        """sum of fret over the plan's components, as computed on the device"""
        return float(np.frombuffer(self.ctx.copy_to_host(self.objective_device_ptr(), 8), dtype=np.float64)[0])

    def allreduce_objective(self, comm: Optional["Comm"], fetch: bool = True):
        """the objective summed over the ranks of `comm` (None: a world of one), in place on the device, on the context's
        stream; fetch: copied out (waits)"""
        out = C.c_double()
        self.ctx.check(self.ctx.lib.rdis_hip_allreduce_objective(self.h, comm.h if comm is not None else None, C.byref(out) if fetch else None))
        return out.value if fetch else None

    def objective_device_ptr(self) -> int:
        p = _vp()
        self.ctx.check(self.ctx.lib.rdis_hip_plan_objective_device(self.h, C.byref(p)))
        return p.value

    def last_kernel_ms(self):
        ms, n = C.c_double(), C.c_int32()
        self.ctx.check(self.ctx.lib.rdis_hip_plan_last_kernel_ms(self.h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def info(self, name: str) -> int:
        v = C.c_int64()
        self.ctx.check(self.ctx.lib.rdis_hip_plan_get_info(self.h, name.encode(), C.byref(v)))
        return v.value

    def device_bytes(self) -> int:
        b = C.c_int64()
        self.ctx.check(self.ctx.lib.rdis_hip_plan_device_bytes(self.h, C.byref(b)))
        return b.value

    def debug_counters(self) -> np.ndarray:
        out = np.zeros(32, dtype=np.int64)
        self.ctx.check(self.ctx.lib.rdis_hip_plan_debug_counters(self.h, _ptr(out)))
        return out

    def get_vectors(self, comp: int, dump_iters: int) -> np.ndarray:
        """[dump_iters, 2, nfree_c]: p and xi at the start of each line minimisation"""
        n = int(self.free_ptr[comp + 1] - self.free_ptr[comp])
        out = np.zeros((dump_iters, 2, n))
        self.ctx.check(self.ctx.lib.rdis_hip_plan_get_vectors(self.h, comp, _ptr(out), out.size))
        return out

    def get_trace(self, comp: int = 0, cap: int = 1 << 16) -> np.ndarray:
        rec = np.zeros((cap, 4))
        n = _i64()
        self.ctx.check(self.ctx.lib.rdis_hip_plan_get_trace(self.h, comp, _ptr(rec), cap, C.byref(n)))
        return rec[:min(n.value, cap)], n.value
