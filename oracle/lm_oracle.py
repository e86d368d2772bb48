"""CPU oracle of the Levenberg-Marquardt subspace solver (test infrastructure, not product code).

PARITY UNPINNED.  The reference's LMSubspaceOptimizer (src/optimizers/LMSubspaceOptimizer.cpp:28-147)
delegates to levmar's `dlevmar_der`, which is neither vendored nor pinned (README.md:38-40,
CMakeLists.txt:189-208) and which no reference test exercises.  What is restated
here is (a) the reference's own part -- the least-squares problem it hands over: one residual per
factor, e_j = sqrt(2 E_j) (LMSubspaceOptimizer.cpp:176-205), Jacobian row grad E_j / e_j
(:208-278, without its lookup bug at :269-274 noted in SURVEY.md 8f), unconstrained, options
mu-scale 1e-3, eps1 = eps2 = 1e-15, eps3 = SSftol, itmax = SSmaxit (:84-101), the result clamped
into the domains afterwards (:104-108) -- and (b) the published algorithm levmar 2.6 implements:
Levenberg-Marquardt with Nielsen's damping update (Madsen, Nielsen, Tingleff, "Methods for
non-linear least squares problems", 2004, Alg. 3.16).  Dense linear algebra in numpy; the device
solver (Schur complement over camera / point blocks) is checked against it step by step.

Stop codes (levmar's numbering): 1 small gradient, 2 small step, 3 itmax, 4 singular, 5 no further
reduction possible, 6 small error, 7 invalid values."""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

EPSILON = 1e-12          # levmar's LM_CNST(EPSILON)
ONE_THIRD = 0.3333333334


@dataclass
class LMResult:
    x: np.ndarray
    fret: float
    finit: float
    iters: int
    stop: int
    nfev: int
    njev: int
    nsolve: int
    mu: float
    history: list = field(default_factory=list)   # per linear solve: (mu, |Dp|^2, f_trial, accepted)


def residuals_and_jacobian(o, free_vid, fac, want_jac=True, model=1):
    """model 1: e_j = sqrt(2 E_j) per listed factor and the dense Jacobian over the free variables
    (rows grad E_j / e_j; a zero row where E_j == 0) -- the reference's formulation.
    model 2: the two pixel residuals per factor and their Jacobian rows (not in the reference;
    lm_optimize then also projects every trial point into the variables' domains, where the
    reference's formulation clamps once at the end)."""
    pp = o.pp
    if model == 2:
        res, Jr = o.resjac_each_ba(fac)
        e = res.reshape(-1)
        if not want_jac:
            return e, None
        col = -np.ones(pp.nvars, dtype=np.int64)
        col[free_vid] = np.arange(len(free_vid))
        vids = np.concatenate([pp.cam_vid0[fac][:, None] + np.arange(9), pp.pt_vid0[fac][:, None] + np.arange(3)], axis=1)
        J = np.zeros((2 * len(fac), len(free_vid)))
        for r in range(2):
            for k in range(12):
                c = col[vids[:, k]]
                m = c >= 0
                J[2 * np.where(m)[0] + r, c[m]] += Jr[m, r, k]
        return e, J
    E = o.eval_each(fac)
    e = np.sqrt(2.0 * E)
    if not want_jac:
        return e, None
    col = -np.ones(pp.nvars, dtype=np.int64)
    col[free_vid] = np.arange(len(free_vid))
    J = np.zeros((len(fac), len(free_vid)))
    if pp.kind == 0:
        g = o.grad_each_ba(fac)
        vids = np.concatenate([pp.cam_vid0[fac][:, None] + np.arange(9), pp.pt_vid0[fac][:, None] + np.arange(3)], axis=1)
        with np.errstate(divide="ignore", invalid="ignore"):
            rows = np.where(e[:, None] > 0, g / e[:, None], 0.0)
        for k in range(12):
            c = col[vids[:, k]]
            m = c >= 0
            J[np.where(m)[0], c[m]] += rows[m, k]
    else:
        raise NotImplementedError("dense LM oracle: bundle adjustment only")
    return e, J


def lm_optimize(o, free_vid=None, fac=None, x=None, maxiters=25, ftol=3e-8, tau=1e-3, eps1=1e-15, eps2=1e-15,
                clamp=True, model=1) -> LMResult:
    """o: oracle.OracleProblem.  Leaves the free variables assigned to the result."""
    pp = o.pp
    free_vid = np.arange(pp.nvars, dtype=np.int64) if free_vid is None else np.asarray(free_vid, dtype=np.int64)
    fac = np.arange(pp.nfac, dtype=np.int64) if fac is None else np.asarray(fac, dtype=np.int64)
    p = np.array(pp.x0[free_vid] if x is None else x, dtype=np.float64)
    m = len(p)
    o.assign(free_vid, p)
    e, _ = residuals_and_jacobian(o, free_vid, fac, want_jac=False, model=model)
    p_eL2 = float(e @ e)
    finit = 0.5 * p_eL2
    eps3 = ftol
    mu, nu, stop, nfev, njev, nsolve = 0.0, 2, 0, 1, 0, 0
    hist = []
    k = 0
    while k < maxiters and not stop:
        if p_eL2 <= eps3:
            stop = 6
            break
        o.assign(free_vid, p)
        e, J = residuals_and_jacobian(o, free_vid, fac, model=model)
        njev += 1
        JtJ = J.T @ J
        Jte = -(J.T @ e)                      # levmar: e = x - hx with x = 0
        p_L2 = float(p @ p)
        if np.max(np.abs(Jte)) <= eps1:
            stop = 1
            break
        dg = np.diag(JtJ)
        # model 2 damps with Marquardt's scaling, mu * diag(J^T J) (floored at 1e-9 of the largest entry)
        damp = np.maximum(dg, 1e-9 * float(np.max(dg))) if model == 2 else np.ones(m)
        if k == 0:
            mu = tau if model == 2 else tau * float(np.max(dg))
        while True:
            A = JtJ + mu * np.diag(damp)
            solved = True
            try:
                L = np.linalg.cholesky(A)
                Dp = np.linalg.solve(L.T, np.linalg.solve(L, Jte))
            except np.linalg.LinAlgError:
                solved = False
            nsolve += 1
            if solved:
                Dp_L2 = float(Dp @ Dp)
                if Dp_L2 <= eps2 * eps2 * p_L2:
                    stop = 2
                    break
                if Dp_L2 >= (p_L2 + eps2) / (EPSILON * EPSILON):
                    stop = 4
                    break
                pDp = p + Dp
                if model == 2:                  # trial points stay inside the domains (projected step)
                    pDp = np.minimum(np.maximum(pDp, pp.lo[free_vid]), pp.hi[free_vid])
                o.assign(free_vid, pDp)
                e_new, _ = residuals_and_jacobian(o, free_vid, fac, want_jac=False, model=model)
                nfev += 1
                pDp_eL2 = float(e_new @ e_new)
                if not np.isfinite(pDp_eL2):
                    stop = 7
                    break
                dL = float(Dp @ (mu * damp * Dp + Jte))
                dF = p_eL2 - pDp_eL2
                ok = dL > 0.0 and dF > 0.0
                hist.append((mu, Dp_L2, 0.5 * pDp_eL2, ok))
                if ok:
                    tmp = 2.0 * dF / dL - 1.0
                    tmp = 1.0 - tmp * tmp * tmp
                    mu = mu * (tmp if tmp >= ONE_THIRD else ONE_THIRD)
                    nu = 2
                    p, p_eL2 = pDp, pDp_eL2
                    break
            mu *= nu
            nu2 = nu << 1
            if nu2 >= (1 << 31):               # levmar: nu wrapped around
                stop = 5
                break
            nu = nu2
        k += 1
    if not stop:
        stop = 3
    if clamp:
        p = np.minimum(np.maximum(p, pp.lo[free_vid]), pp.hi[free_vid])
    o.assign(free_vid, p)
    fret = float(o.eval(fac))
    return LMResult(x=p, fret=fret, finit=finit, iters=k, stop=stop, nfev=nfev, njev=njev, nsolve=nsolve, mu=mu, history=hist)
