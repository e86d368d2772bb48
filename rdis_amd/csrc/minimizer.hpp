// minimizer.hpp -- the subspace solver's control logic as device code.
//
// Polak-Ribiere conjugate gradients with bracketing + Brent-with-derivatives
// line minimisation, i.e. what CGDSubspaceOptimizer::optimize drives through
// nrc::Frprmn (reference src/optimizers/CGDSubspaceOptimizer.cpp:19-98,
// external/include/minimize_nrc.h:80-151, 284-404, 492-513, 585-692), with the
// reference's constants, evaluation order and exits.
//
// Every lane of the cooperating group (one workgroup, or a whole grid) runs this
// scalar logic redundantly on group-uniform values: `Env` turns a step length
// into a group-wide reduction, so all lanes take the same branches and nothing
// has to be broadcast.  FP contraction is off in here so that, fed the same
// function values, the device takes bit-identical decisions to the CPU oracle
// (tests/test_replay.py checks exactly that).
//
// Env concept:
//   double line_f(double a)                       f(clamp(p + a*xi))               [operator()]
//   void   line_fd(double a, double&f, double&s)  f and slope d/da at the same a   [operator() then df]
//   void   count_slope()                          the reference would have called df here
//   double first_eval()        assign clamp(x0), return f         (CGD .cpp:34-37)
//   double start_point()       Frprmn's fp = func(p)              (nrc :628)
//   void   gradient_to_xi()    xi <- grad f(clamp(p))             (nrc :629, :654)
//   void   cg_start()          g = -xi; xi = h = g                (nrc :631-635)
//   void   line_begin()/line_end(double amin)                     (nrc :505-511)
//   void   cg_reduce(double fp, double& test, double& gg, double& dgg)   (nrc :655-672)
//   void   cg_update(double gam)                                  (nrc :679-683)
//   void   trace(int tag, double a, double b, double c)
#pragma once
#include <hip/hip_runtime.h>
#include <cfloat>

namespace rdis_hip {

// exit reasons: low byte of status (mirrors oracle RO_EXIT_*; include/rdis_hip.h)
enum : int {
    EXIT_FTOL = 0, EXIT_GTOL = 1, EXIT_GGZERO = 2, EXIT_ITMAX = 3, EXIT_DBRENT_ITMAX = 4,
    EXIT_NAN = 5, EXIT_EMPTY = 6, EXIT_SYNC_TIMEOUT = 7
};
constexpr int STATUS_ROLLED_BACK = 0x100;

enum : int { TR_F = 1, TR_FD = 2, TR_ITER = 3, TR_START = 4, TR_LINMIN = 5 };

struct Bracket { double ax, bx, cx, fa, fb, fc; };

// Bracketmethod::bracket (nrc :80-151)
template <class Env>
__device__ void bracket_min(Env& E, Bracket& B, double a, double b) {
#pragma clang fp contract(off)
    const double GOLD = 1.618034, GLIMIT = 100.0, TINY = 1.0e-20;
    double ax = a, bx = b, cx, fa, fb, fc, fu, tmp;
    fa = E.line_f(ax);
    fb = E.line_f(bx);
    if (fb > fa) {
        tmp = ax; ax = bx; bx = tmp;
        tmp = fa; fa = fb; fb = tmp;
    }
    cx = bx + GOLD * (bx - ax);
    fc = E.line_f(cx);
    while (fb > fc) {
        const double r = (bx - ax) * (fb - fc);
        const double q = (bx - cx) * (fb - fa);
        const double qr = q - r;
        double u = bx - ((bx - cx) * q - (bx - ax) * r) / (2.0 * copysign(fmax(fabs(qr), TINY), qr));
        const double ulim = bx + GLIMIT * (cx - bx);
        if ((bx - u) * (u - cx) > 0.0) {
            fu = E.line_f(u);
            if (fu < fc) { ax = bx; bx = u; fa = fb; fb = fu; break; }
            if (fu > fb) { cx = u; fc = fu; break; }
            u = cx + GOLD * (cx - bx);
            fu = E.line_f(u);
        } else if ((cx - u) * (u - ulim) > 0.0) {
            fu = E.line_f(u);
            if (fu < fc) {
                const double unew = u + GOLD * (u - cx);
                bx = cx; cx = u; u = unew;
                fb = fc; fc = fu; fu = E.line_f(u);
            }
        } else if ((u - ulim) * (ulim - cx) >= 0.0) {
            u = ulim;
            fu = E.line_f(u);
        } else {
            u = cx + GOLD * (cx - bx);
            fu = E.line_f(u);
        }
        ax = bx; bx = cx; cx = u;
        fa = fb; fb = fc; fc = fu;
    }
    B.ax = ax; B.bx = bx; B.cx = cx; B.fa = fa; B.fb = fb; B.fc = fc;
}

// Dbrent::minimize (nrc :284-404); false = 100 iterations without convergence
template <class Env>
__device__ bool dbrent_min(Env& E, const Bracket& B, double& xmin, double& fmin) {
#pragma clang fp contract(off)
    const int ITMAX = 100;
    const double tol = 3.0e-8;  // Dbrent's own default, not the solver's ftol (nrc :288,:499)
    const double ZEPS = DBL_EPSILON * 1.0e-3;
    double a = (B.ax < B.cx ? B.ax : B.cx);
    double b = (B.ax > B.cx ? B.ax : B.cx);
    double x, w, v, fx, fw, fv, dx, dw, dv, u, fu, du;
    double d = 0.0, e = 0.0;
    x = w = v = B.bx;
    E.line_fd(x, fx, dx);
    E.count_slope();
    fw = fv = fx;
    dw = dv = dx;
    for (int it = 0; it < ITMAX; ++it) {
        const double xm = 0.5 * (a + b);
        const double tol1 = tol * fabs(x) + ZEPS;
        const double tol2 = 2.0 * tol1;
        if (fabs(x - xm) <= (tol2 - 0.5 * (b - a))) { fmin = fx; xmin = x; return true; }
        bool bisect = true;
        if (fabs(e) > tol1) {
            double d1 = 2.0 * (b - a), d2 = d1;
            if (dw != dx) d1 = (w - x) * dx / (dx - dw);
            if (dv != dx) d2 = (v - x) * dx / (dx - dv);
            const double u1 = x + d1, u2 = x + d2;
            const bool ok1 = (a - u1) * (u1 - b) > 0.0 && dx * d1 <= 0.0;
            const bool ok2 = (a - u2) * (u2 - b) > 0.0 && dx * d2 <= 0.0;
            const double olde = e;
            e = d;
            if (ok1 || ok2) {
                if (ok1 && ok2) d = (fabs(d1) < fabs(d2) ? d1 : d2);
                else if (ok1) d = d1;
                else d = d2;
                if (fabs(d) <= fabs(0.5 * olde)) {
                    u = x + d;
                    if (u - a < tol2 || b - u < tol2) d = copysign(tol1, xm - x);
                    bisect = false;
                }
            }
        }
        if (bisect) { e = (dx >= 0.0 ? a - x : b - x); d = 0.5 * e; }
        if (fabs(d) >= tol1) {
            u = x + d;
            E.line_fd(u, fu, du);
        } else {
            u = x + copysign(tol1, d);
            E.line_fd(u, fu, du);
            if (fu > fx) { fmin = fx; xmin = x; return true; }
        }
        E.count_slope();
        if (fu <= fx) {
            if (u >= x) a = x; else b = x;
            v = w; fv = fw; dv = dw;
            w = x; fw = fx; dw = dx;
            x = u; fx = fu; dx = du;
        } else {
            if (u < x) a = u; else b = u;
            if (fu <= fw || w == x) {
                v = w; fv = fw; dv = dw;
                w = u; fw = fu; dw = du;
            } else if (fu < fv || v == x || v == w) {
                v = u; fv = fu; dv = du;
            }
        }
    }
    return false;
}

struct SolveOut { double fret; int iter; int reason; };

// Frprmn::minimize (nrc :619-691) around Dlinemethod::linmin (nrc :492-513)
template <class Env>
__device__ SolveOut frprmn(Env& E, int maxiters, double ftol) {
#pragma clang fp contract(off)
    const double EPS = 1.0e-18, GTOL = 1.0e-8;
    SolveOut out{DBL_MAX, 0, EXIT_ITMAX};
    double fp = E.start_point();
    E.gradient_to_xi();
    E.cg_start();
    E.trace(TR_START, fp, 0.0, 0.0);
    for (int its = 0; its < maxiters; ++its) {
        out.iter = its;
        Bracket B;
        double amin, fmin;
        E.line_begin();
        bracket_min(E, B, 0.0, 1.0);
        if (!dbrent_min(E, B, amin, fmin)) { out.reason = EXIT_DBRENT_ITMAX; return out; }
        E.line_end(amin);
        E.trace(TR_LINMIN, amin, fmin, 0.0);
        out.fret = fmin;
        if (2.0 * fabs(out.fret - fp) <= ftol * (fabs(out.fret) + fabs(fp) + EPS)) {
            out.reason = EXIT_FTOL; return out;
        }
        fp = out.fret;
        E.gradient_to_xi();
        double test, gg, dgg;
        E.cg_reduce(fp, test, gg, dgg);
        E.trace(TR_ITER, test, gg, dgg);
        if (test < GTOL) { out.reason = EXIT_GTOL; return out; }
        if (gg == 0.0) { out.reason = EXIT_GGZERO; return out; }
        E.cg_update(dgg / gg);
    }
    return out;
}

}  // namespace rdis_hip
