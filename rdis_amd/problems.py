"""Input builders for the subspace-solver hot path: packed (structure-of-arrays)
problems in the layout the C-ABI uploads (include/rdis_hip.h).

They restate the reference's *input formats and id layout* only -- no factor
arithmetic lives here:

* BAL text format, id layout and per-type domains:
  src/bundleadjust/BundleAdjustmentFunction.cpp:50-250 (load), :402-477
  (setDomain), BundleAdjustmentFunction.h:88-96 (getCamVID / getPointVID).
* polynomial text format: src/PolynomialFunction.cpp:60-215.
* high-dimensional sinusoid: src/OptimizableFunctionGenerator.cpp:660-760.
* the synthetic decomposable bundle adjustment (BASELINE.json config 5) has no
  counterpart in the reference; it is specified here.

Variable ids are dense 0..N-1 in creation order, factor ids are the index in
the factor list (src/OptimizableFunction.cpp:57-76,
BundleAdjustmentFunction.cpp:164-166); both are int64 like the reference's
VariableID / FactorID (src/common.h:30-32).
"""
from __future__ import annotations

import gzip
import math
import os
from dataclasses import dataclass, field
from typing import Optional

import numpy as np

KIND_BA = 0
KIND_NLP = 1

_GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                       "tests", "golden")
LADYBUG_PATH = os.path.join(_GOLDEN, "ladybug-problem-49-7776-pre.txt.gz")
TESTPOLY_PATH = os.path.join(_GOLDEN, "testpoly.txt")


@dataclass
class PackedProblem:
    """One OptimizableFunction in packed form (all factors of one kind)."""
    kind: int
    x0: np.ndarray            # [N] f64 start / currently assigned values
    lo: np.ndarray            # [N] f64 domain lower bound (single interval)
    hi: np.ndarray            # [N] f64 domain upper bound
    # bundle adjustment: factor i reads cam_vid0[i]..+8 and pt_vid0[i]..+2
    cam_vid0: Optional[np.ndarray] = None   # [F] i64
    pt_vid0: Optional[np.ndarray] = None    # [F] i64
    obs: Optional[np.ndarray] = None        # [F,2] f64
    # nonlinear product factors, CSR over (vid, exponent, constant, sine)
    coeff: Optional[np.ndarray] = None      # [F] f64
    rowptr: Optional[np.ndarray] = None     # [F+1] i64
    vid: Optional[np.ndarray] = None        # [nnz] i64
    expo: Optional[np.ndarray] = None       # [nnz] f64
    cons: Optional[np.ndarray] = None       # [nnz] f64
    sine: Optional[np.ndarray] = None       # [nnz] u8
    # decomposition into independent components (CSR), if the builder knows one
    comp_free_ptr: Optional[np.ndarray] = None   # [ncomp+1] i64
    comp_free_vid: Optional[np.ndarray] = None   # i64, ascending within a component
    comp_fac_ptr: Optional[np.ndarray] = None    # [ncomp+1] i64
    comp_fac_id: Optional[np.ndarray] = None     # i64, ascending within a component
    meta: dict = field(default_factory=dict)

    @property
    def nvars(self) -> int:
        return int(self.x0.shape[0])

    @property
    def nfac(self) -> int:
        return int(self.cam_vid0.shape[0] if self.kind == KIND_BA else self.coeff.shape[0])

    @property
    def ncomp(self) -> int:
        return 0 if self.comp_free_ptr is None else int(self.comp_free_ptr.shape[0] - 1)

    def component(self, c: int):
        """(free_vid, fac_id) of component c of the stored decomposition"""
        return (self.comp_free_vid[int(self.comp_free_ptr[c]):int(self.comp_free_ptr[c + 1])],
                self.comp_fac_id[int(self.comp_fac_ptr[c]):int(self.comp_fac_ptr[c + 1])])

    def single_component(self) -> "PackedProblem":
        """Decomposition with one component = all variables, all factors (the
        BCD-with-one-block harness shape, src/optimizers/BCDOptimizer.cpp:149)."""
        self.comp_free_ptr = np.array([0, self.nvars], dtype=np.int64)
        self.comp_free_vid = np.arange(self.nvars, dtype=np.int64)
        self.comp_fac_ptr = np.array([0, self.nfac], dtype=np.int64)
        self.comp_fac_id = np.arange(self.nfac, dtype=np.int64)
        return self


# ---------------------------------------------------------------------------
# bundle adjustment: BAL loader
# ---------------------------------------------------------------------------

def ba_domains(x0: np.ndarray, ncams: int) -> tuple[np.ndarray, np.ndarray]:
    """Per-variable [lo, hi] as BundleAdjustmentFunction::setDomain builds them
    (src/bundleadjust/BundleAdjustmentFunction.cpp:402-477): the hull of the
    1000x-scaled interval and the sampling interval around the initial value."""
    n = x0.shape[0]
    ncv = 9 * ncams
    typ = np.empty(n, dtype=np.int64)
    typ[:ncv] = np.arange(ncv) % 9
    typ[ncv:] = 9 + (np.arange(n - ncv) % 3)
    return _domains_by_type(x0, typ)


def _open_text(path: str):
    if path.endswith(".gz"):
        return gzip.open(path, "rt")
    return open(path, "r")


def load_bal(path: str = LADYBUG_PATH, ncams: int = 0, npts: int = 0) -> PackedProblem:
    """BAL file -> packed problem.  Header `ncams npts nobs`, then nobs lines
    `cam pt x y`, then 9 lines per camera and 3 per point; `ncams`/`npts` > 0
    keep the first k cameras / points and drop the other observations
    (BundleAdjustmentFunction.cpp:120-156)."""
    with _open_text(path) as fh:
        toks = []
        for line in fh:
            if not line.strip() or line[0] == "#":
                continue
            toks.extend(line.replace(",", " ").replace(";", " ").split())
    fc, fp, fo = int(toks[0]), int(toks[1]), int(toks[2])
    body = toks[3:]
    obs_t = np.array(body[:4 * fo], dtype=object).reshape(fo, 4)
    cam = obs_t[:, 0].astype(np.int64)
    pt = obs_t[:, 1].astype(np.int64)
    oxy = obs_t[:, 2:].astype(np.float64)
    params = np.array(body[4 * fo:4 * fo + 9 * fc + 3 * fp], dtype=np.float64)
    nc = fc if ncams <= 0 else ncams
    npnt = fp if npts <= 0 else npts
    assert nc <= fc and npnt <= fp
    keep = (cam < nc) & (pt < npnt)
    cam, pt, oxy = cam[keep], pt[keep], oxy[keep]
    x0 = np.concatenate([params[:9 * nc], params[9 * fc:9 * fc + 3 * npnt]])
    lo, hi = ba_domains(x0, nc)
    return PackedProblem(
        kind=KIND_BA, x0=x0, lo=lo, hi=hi,
        cam_vid0=(cam * 9).astype(np.int64),
        pt_vid0=(nc * 9 + pt * 3).astype(np.int64),
        obs=np.ascontiguousarray(oxy),
        meta={"ncams": nc, "npts": npnt, "source": os.path.basename(path)},
    )


def save_bal(pp: PackedProblem, path: str, x: Optional[np.ndarray] = None) -> None:
    """Write a packed BA problem (with state x, default x0) in BAL format -- what
    BundleAdjustmentFunction::save does (src/bundleadjust/BundleAdjustmentFunction.cpp:253-320)."""
    assert pp.kind == KIND_BA
    x = pp.x0 if x is None else np.asarray(x, dtype=np.float64)
    nc = int(pp.meta.get("ncams", int(pp.pt_vid0.min()) // 9 if pp.nfac else 0))
    npnt = (pp.nvars - 9 * nc) // 3
    with open(path, "w") as fh:
        fh.write(f"{nc} {npnt} {pp.nfac}\n")
        for c, q, (ox, oy) in zip(pp.cam_vid0 // 9, (pp.pt_vid0 - 9 * nc) // 3, pp.obs):
            fh.write(f"{int(c)} {int(q)}     {ox:.17e} {oy:.17e}\n")
        for v in x:
            fh.write(f"{v:.17e}\n")


def ba_alternation_plans(pp: PackedProblem):
    """The decomposition RDIS reaches on a BAL problem once a separator is assigned (SURVEY.md
    3.2b): with the points fixed every camera is an independent 9-variable component (its factors
    = its observations), with the cameras fixed every point is an independent 3-variable
    component.  Returns two CSR decompositions (free_ptr, free_vid, fac_ptr, fac_id): cameras, points.
    Factor lists are in ascending factor id, like Component's (src/Component.cpp:78-79,103,116)."""
    assert pp.kind == KIND_BA
    nc = int(pp.meta["ncams"])
    npnt = (pp.nvars - 9 * nc) // 3
    fid = np.arange(pp.nfac, dtype=np.int64)
    cam_of, pt_of = pp.cam_vid0 // 9, (pp.pt_vid0 - 9 * nc) // 3
    oc = np.argsort(cam_of, kind="stable")
    op = np.argsort(pt_of, kind="stable")
    cams = (np.arange(nc + 1, dtype=np.int64) * 9, np.arange(9 * nc, dtype=np.int64),
            np.concatenate([[0], np.cumsum(np.bincount(cam_of, minlength=nc))]).astype(np.int64), fid[oc])
    pts = (np.arange(npnt + 1, dtype=np.int64) * 3, 9 * nc + np.arange(3 * npnt, dtype=np.int64),
           np.concatenate([[0], np.cumsum(np.bincount(pt_of, minlength=npnt))]).astype(np.int64), fid[op])
    return cams, pts


# ---------------------------------------------------------------------------
# nonlinear product factors: polynomial file and the sinusoid generator
# ---------------------------------------------------------------------------

def _is_number(s: str) -> bool:
    try:
        float(s)
        return True
    except ValueError:
        return False


def load_poly(path: str = TESTPOLY_PATH) -> PackedProblem:
    """`name = lo:hi` lines declare variables (`default` sets the default
    domain); every other non-comment line is one term
    `[coeff][, var^exp ...]` (src/PolynomialFunction.cpp:60-215)."""
    names: list[str] = []
    doms: list[Optional[tuple[float, float]]] = []
    default = (0.0, 0.0)
    terms = []

    def var_id(name: str, dom=None) -> int:
        if name in names:
            return names.index(name)
        names.append(name)
        doms.append(dom if dom is not None else default)  # default domain as of creation
        return len(names) - 1

    with _open_text(path) as fh:
        for raw in fh:
            line = raw.rstrip("\n")
            if not line or line[0] == "#":
                continue
            if "=" in line:
                name, dom = line.split("=", 1)
                a, b = _parse_domain(dom)
                if name.strip().lower() == "default":
                    default = (a, b)
                else:
                    var_id(name.strip(), (a, b))
                continue
            coeff = 1.0
            ents = []
            for part in line.split(","):
                sv = part.split("^")
                name = sv[0].strip().lower()
                if len(sv) == 1:
                    if _is_number(name):
                        coeff = float(name)
                        continue
                    v, e = var_id(name), 1.0
                else:
                    v, e = var_id(name), float(sv[1].strip())
                # addVariable: exponent 0 is dropped, a repeated variable is ignored
                # (src/NonlinearProductFactor.cpp:27-52)
                if e != 0.0 and v not in [t[0] for t in ents]:
                    ents.append((v, e, 0.0, 0))
            terms.append((coeff, ents))
    lo = np.array([d[0] for d in doms])
    hi = np.array([d[1] for d in doms])
    return _pack_nlp(terms, np.zeros(len(names)), lo, hi, {"source": os.path.basename(path)})


def _parse_domain(s: str) -> tuple[float, float]:
    # VariableDomain::parse, single interval form (src/VariableDomain.cpp:63-72)
    toks = [t for t in s.replace("~", " ").replace(":", " ").replace(",", " ").split() if t]
    return float(toks[0]), float(toks[-1])


def _pack_nlp(terms, x0, lo, hi, meta) -> PackedProblem:
    rowptr = np.zeros(len(terms) + 1, dtype=np.int64)
    for i, (_, ents) in enumerate(terms):
        rowptr[i + 1] = rowptr[i] + len(ents)
    flat = [e for _, ents in terms for e in ents]
    return PackedProblem(
        kind=KIND_NLP, x0=np.asarray(x0, dtype=np.float64), lo=lo, hi=hi,
        coeff=np.array([c for c, _ in terms], dtype=np.float64), rowptr=rowptr,
        vid=np.array([e[0] for e in flat], dtype=np.int64),
        expo=np.array([e[1] for e in flat], dtype=np.float64),
        cons=np.array([e[2] for e in flat], dtype=np.float64),
        sine=np.array([e[3] for e in flat], dtype=np.uint8), meta=meta)


def make_high_dim_sinusoid(height: int = 4, branches: int = 3, max_arity: int = 3,
                           odd_arity: bool = False) -> PackedProblem:
    """makeHighDimSinusoid (src/OptimizableFunctionGenerator.cpp:660-760): a
    complete k-ary tree of variables; for each allowed arity one factor per
    variable deep enough, over the variable and its ancestors (listed from the
    highest ancestor down), coefficient 12 with sines (0.6, no sine, for arity
    1); plus 0.1*x^2 per variable.  Defaults = optSinusoid's
    (src/optimize_sinusoid.cpp:291-298): 121 variables, 362 factors."""
    h, k = height, branches
    twopi = 2.000001 * 3.141592653
    nvars = h + 1 if k == 1 else (round(float(k) ** (h + 1)) - 1) // (k - 1)
    max_arity = min(max_arity, h + 1)
    # the default domain goes through a "%g"-style 6-digit format (boost::format)
    bound = float("%g" % (10 * twopi))
    terms = []
    for ar in range(1, max_arity + 1):
        if ar > 1 and (ar & 1) and not odd_arity:
            continue
        lasth = h
        for v in range(nvars - 1, -1, -1):
            last_at_next = (lasth - 1) if k == 1 else int((round(float(k) ** lasth) - 1.0) / (k - 1.0) - 1.0)
            if v > last_at_next:
                varheight = lasth
            else:
                lasth -= 1
                varheight = lasth
            if varheight + 1 < ar:
                continue
            chain, cur = [], v
            for _ in range(ar):
                chain.append(cur)
                cur = math.floor((cur - 1.0) / k)
            ents = [(c, 1.0, 0.0, 1 if ar > 1 else 0) for c in reversed(chain)]
            terms.append((12.0 if ar > 1 else 0.6, ents))
    for v in range(nvars):
        terms.append((0.1, [(v, 2.0, 0.0, 0)]))
    lo = np.full(nvars, -bound)
    hi = np.full(nvars, bound)
    return _pack_nlp(terms, np.zeros(nvars), lo, hi,
                     {"generator": "sinusoid", "h": h, "k": k, "sample": (-twopi, twopi)})


# ---------------------------------------------------------------------------
# synthetic decomposable bundle adjustment (BASELINE.json config 5)
# ---------------------------------------------------------------------------

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)
_GAMMA = np.uint64(0x9E3779B97F4A7C15)


def _mix64(z: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def splitmix64(seed: np.ndarray, counter: np.ndarray) -> np.ndarray:
    """counter-based splitmix64: the (counter+1)-th output of the generator
    seeded with `seed` (both uint64 arrays, broadcast)."""
    with np.errstate(over="ignore"):
        return _mix64(seed + (counter + np.uint64(1)) * _GAMMA)


class _Stream:
    """Per-component random streams: component c draws from splitmix64 seeded
    with mix(0x5D15 + c); `uniform(tag, n)` returns the draws [tag, tag+n)."""

    def __init__(self, comp_ids: np.ndarray):
        with np.errstate(over="ignore"):
            self.seed = _mix64((np.uint64(0x5D15) + comp_ids.astype(np.uint64)) + _GAMMA)

    def uniform(self, offset: int, shape: tuple[int, ...]) -> np.ndarray:
        """uniform [0,1) of shape (ncomp,)+shape; draw index = offset + flat idx"""
        n = int(np.prod(shape)) if shape else 1
        ctr = np.uint64(offset) + np.arange(n, dtype=np.uint64)
        z = splitmix64(self.seed[:, None], ctr[None, :])
        u = (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
        return u.reshape((self.seed.shape[0],) + shape)

    def normal(self, offset: int, shape: tuple[int, ...]) -> np.ndarray:
        n = int(np.prod(shape)) if shape else 1
        u = self.uniform(offset, (2 * n,))
        r = np.sqrt(-2.0 * np.log(1.0 - u[:, :n]))
        return (r * np.cos(2.0 * math.pi * u[:, n:])).reshape((self.seed.shape[0],) + shape)


def _project(cam: np.ndarray, q: np.ndarray) -> np.ndarray:
    """numpy Snavely projection used ONLY to synthesise observations
    (generator-side; the solver's arithmetic lives in the HIP kernels)."""
    r, t, f, k1, k2 = cam[..., 0:3], cam[..., 3:6], cam[..., 6], cam[..., 7], cam[..., 8]
    th = np.linalg.norm(r, axis=-1, keepdims=True)
    v = r / th
    c, s = np.cos(th), np.sin(th)
    P = q * c + np.cross(v, q) * s + v * (1 - c) * np.sum(v * q, axis=-1, keepdims=True) + t
    pp = -P[..., :2] / P[..., 2:3]
    r2 = np.sum(pp * pp, axis=-1)
    d = 1 + r2 * (k1 + k2 * r2)
    return (f * d)[..., None] * pp


def make_synthetic_ba(ncomp: int = 1000, ncams: int = 3, npts: int = 40,
                      obs_per_pt: Optional[int] = None, first_comp: int = 0,
                      noise_px: float = 0.5) -> PackedProblem:
    """`ncomp` independent bundle-adjustment components (ids first_comp ..),
    each `ncams` cameras x `npts` points laid out like a BAL problem (cameras
    first, 9 each, then points, 3 each); component c occupies the contiguous
    variable range [c*(9C+3P), (c+1)*(9C+3P)) and factor range, observations
    sorted by (point, camera).  Every point is seen by `obs_per_pt` cameras
    (default: all if C <= 4, else 4; always >= 2), a cyclic window starting at a
    random camera.  Ground truth: |r| in [0.05, 1], depth 3..5, f ~ 400,
    k1 ~ -3e-7, k2 ~ 5e-13 (ladybug-like ranges); observations =
    projection + N(0, noise_px); start = truth + perturbation."""
    C, P = ncams, npts
    K = obs_per_pt if obs_per_pt is not None else (C if C <= 4 else 4)
    K = max(2, min(K, C))
    ids = first_comp + np.arange(ncomp, dtype=np.int64)
    rs = _Stream(ids)
    off = 0

    def U(shape):
        nonlocal off
        out = rs.uniform(off, shape)
        off += int(np.prod(shape))
        return out

    def N(shape):
        nonlocal off
        out = rs.normal(off, shape)
        off += 2 * int(np.prod(shape))
        return out

    axis = N((C, 3))
    axis /= np.linalg.norm(axis, axis=-1, keepdims=True)
    theta = 0.05 + 0.95 * U((C, 1))
    cam = np.empty((ncomp, C, 9))
    cam[..., 0:3] = axis * theta
    cam[..., 3:5] = 0.2 * (U((C, 2)) - 0.5)
    cam[..., 5] = -(3.0 + 2.0 * U((C,)))
    cam[..., 6] = 400.0 + 50.0 * (U((C,)) - 0.5)
    cam[..., 7] = -3e-7 * (0.5 + U((C,)))
    cam[..., 8] = 5e-13 * (0.5 + U((C,)))
    pts = 2.0 * (U((P, 3)) - 0.5)

    first = np.floor(U((P,)) * C).astype(np.int64) % C           # [ncomp,P]
    win = np.sort((first[..., None] + np.arange(K)) % C, axis=-1)  # [ncomp,P,K] ascending cams
    camsel = cam[np.arange(ncomp)[:, None], win.reshape(ncomp, P * K)]          # [ncomp,P*K,9]
    qsel = np.repeat(pts, K, axis=1)                                       # [ncomp,P*K,3]
    obs = _project(camsel, qsel) + noise_px * N((P * K, 2))

    x_true = np.concatenate([cam.reshape(ncomp, 9 * C), pts.reshape(ncomp, 3 * P)], axis=1)
    pert = np.concatenate([
        np.tile(np.array([0.01, 0.01, 0.01, 0.02, 0.02, 0.02, 2.0, 2e-8, 2e-14]), C),
        np.full(3 * P, 0.02)])
    x0 = x_true + pert * N((9 * C + 3 * P,))

    nv, nf = 9 * C + 3 * P, P * K
    voff = (np.arange(ncomp, dtype=np.int64) * nv)[:, None]
    cam_vid0 = (voff + 9 * win.reshape(ncomp, nf)).reshape(-1)
    pt_vid0 = (voff + 9 * C + 3 * np.repeat(np.arange(P, dtype=np.int64), K)[None, :]).reshape(-1)
    x0f = x0.reshape(-1)
    # per-type domains, exactly as the BAL loader would assign them per component
    typ = np.concatenate([np.arange(9 * C) % 9, 9 + np.arange(3 * P) % 3])
    lo, hi = _domains_by_type(x0f, np.tile(typ, ncomp))
    prob = PackedProblem(
        kind=KIND_BA, x0=x0f, lo=lo, hi=hi, cam_vid0=cam_vid0.astype(np.int64),
        pt_vid0=pt_vid0.astype(np.int64), obs=np.ascontiguousarray(obs.reshape(-1, 2)),
        comp_free_ptr=np.arange(ncomp + 1, dtype=np.int64) * nv,
        comp_free_vid=np.arange(ncomp * nv, dtype=np.int64),
        comp_fac_ptr=np.arange(ncomp + 1, dtype=np.int64) * nf,
        comp_fac_id=np.arange(ncomp * nf, dtype=np.int64),
        meta={"generator": "synthetic_ba", "ncomp": ncomp, "ncams": C, "npts": P,
              "obs_per_pt": K, "first_comp": first_comp, "x_true": x_true.reshape(-1)})
    return prob


def _domains_by_type(x0: np.ndarray, typ: np.ndarray) -> tuple[np.ndarray, np.ndarray]:
    """ba_domains for an arbitrary per-variable type vector (0..11)."""
    lo = np.empty_like(x0)
    hi = np.empty_like(x0)
    dsf = 1000.0
    rot = typ < 3
    lo[rot], hi[rot] = -math.pi * dsf, math.pi * dsf
    pos = ((typ >= 3) & (typ <= 5)) | (typ >= 9)
    slo, shi = x0[pos] + -100.0, x0[pos] + 100.0
    lo[pos], hi[pos] = np.minimum(slo * dsf, slo), np.maximum(shi * dsf, shi)
    foc = typ == 6
    slo, shi = x0[foc] + -100.0, x0[foc] + 100.0
    lo[foc] = np.minimum(np.maximum(np.minimum(slo, slo * dsf), 0.0), slo)
    hi[foc] = np.maximum(shi * dsf, shi)
    k1 = typ == 7
    lo[k1], hi[k1] = np.minimum(-1e-1, x0[k1] + -1e-4), np.maximum(1e-1, x0[k1] + 1e-4)
    k2 = typ == 8
    lo[k2], hi[k2] = np.minimum(-1e-3, x0[k2] + -1e-6), np.maximum(1e-3, x0[k2] + 1e-6)
    return lo, hi


def shard_components(ncomp: int, weights: np.ndarray, world: int) -> list[np.ndarray]:
    """Static LPT partition of components over `world` ranks by factor count
    (SURVEY.md 8e): heaviest first, always onto the lightest rank; ties broken
    by component id / rank id so that every rank computes the same partition."""
    order = np.lexsort((np.arange(ncomp), -np.asarray(weights)))
    load = np.zeros(world, dtype=np.int64)
    out: list[list[int]] = [[] for _ in range(world)]
    for c in order:
        r = int(np.argmin(load))
        out[r].append(int(c))
        load[r] += int(weights[c])
    return [np.array(sorted(o), dtype=np.int64) for o in out]
