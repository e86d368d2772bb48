// minimizer.hpp -- the subspace solver's control logic as a resumable scalar
// state machine.
//
// What it computes: CGDSubspaceOptimizer::optimize (reference
// src/optimizers/CGDSubspaceOptimizer.cpp:19-98) driving nrc::Frprmn -- Polak-
// Ribiere conjugate gradients (external/include/minimize_nrc.h:619-691), line
// minimisation (:492-513) by golden-section / parabolic bracketing (:80-151) and
// Brent's method with derivatives (:284-404) -- with the reference's constants,
// evaluation order, exits and f / df call counts.
//
// How: the reference nests four routines that call the objective from a dozen
// places.  Inlined on a GPU that puts a dozen copies of the factor arithmetic
// into one kernel (110 KB of code, more than the instruction cache: measured 69 %
// of the solve stalled on instruction fetch).  Here the control logic is turned
// inside out: `next()` consumes the reply to the previous request and returns
// the next request -- an evaluation ("value (and slope) at step a") or a full
// gradient, each with the vector updates that surround it -- and the kernel's
// driver loop holds the ONE copy of each heavy operation.  The machine's state
// lives in LDS; one wave per workgroup steps it and hands the request to the
// others (run_machine).  FP contraction is off in here: fed the same replies the
// device takes bit-identical decisions to the CPU oracle (tests/test_gpu_solver.py
// replays the device's trace through the oracle to check exactly that).
#pragma once
#include <hip/hip_runtime.h>
#include <cfloat>
#include <type_traits>
#include <utility>

namespace rdis_hip {

// exit reasons: low byte of status (include/rdis_hip.h RDIS_HIP_EXIT_*)
enum : int {
    EXIT_FTOL = 0, EXIT_GTOL = 1, EXIT_GGZERO = 2, EXIT_ITMAX = 3, EXIT_DBRENT_ITMAX = 4,
    EXIT_NAN = 5, EXIT_EMPTY = 6, EXIT_SYNC_TIMEOUT = 7
};
constexpr int STATUS_ROLLED_BACK = 0x100;

// trace record tags (the replay check in tests/ reads the same tags)
enum : int { TR_NONE = 0, TR_F = 1, TR_FD = 2, TR_ITER = 3, TR_START = 4, TR_LINMIN = 5 };

// What the machine asks the solver's lanes to do next.  A request is one evaluation or one
// full gradient, optionally wrapped in the vector updates that surround it in the algorithm and
// need no decision in between (every request costs a step of the machine and a hand-over):
//   REQ_EVAL      [PRE_START | PRE_UPDATE(b = gamma)] [PRE_BEGIN]  value (+ slope) at step a
//   REQ_GRAD      [PRE_LINE_END(a = step)]  xi <- grad f(clamp(p))  [POST_REDUCE(b = fp)]
//   REQ_LINE_END  xi *= a; p += xi                                  (nrc :508-511)
enum : int { REQ_EVAL = 0, REQ_GRAD, REQ_LINE_END, REQ_DONE };
enum : int {
    RF_SLOPE = 1,          // also the slope d/da; reply r0 = f, r1 = slope  (else r0 = f)
    RF_RESTORE = 2,        // evaluate at clamp(x_init) instead (the rollback, CGD .cpp:71)
    RF_LINE = 4,           // an evaluation on the current line (recorded in the trace)
    RF_PRE_START = 8,      // first: g = -xi; xi = h = g                   (nrc :631-635)
    RF_PRE_UPDATE = 16,    // first: g = -xi; xi = h = g + b*h             (nrc :679-683)
    RF_PRE_BEGIN = 32,     // then: a new line p + a*xi starts             (nrc :496-498)
    RF_PRE_LINE_END = 64,  // REQ_GRAD: first xi *= a; p += xi             (nrc :508-511)
    RF_POST_REDUCE = 128,  // REQ_GRAD: then reply r0 = test, r1 = gg, r2 = dgg with fp = b (nrc :655-672)
    RF_TR_FIRST = 256      // emit the tr_ trace record before the pre_ one
};

// Speculation.  An evaluation is a latency chain (arithmetic, reduction, exchange between
// workgroups, a step of this machine), and in a line search every trial step depends on the reply to
// the previous one.  But most steps of Brent's method on these problems are bisections towards the
// best point (the bracket is [-1.6, 1] and the minimum lies at 1e-7..1e-3 of it), and the step a
// bisection takes does not depend on the VALUE the pending trial returns, only on its being worse
// than the best point.  So a solver environment may evaluate, together with the trial step a
// request asks for, up to SPEC_MAX - 1 guesses at the following steps (Predictor, below).  When the
// machine, stepped with the reply to the real trial, asks for exactly a step that was guessed (same
// bits), the reply is already there and the machine is stepped again at once; otherwise the guess
// is dropped.  The machine itself never sees a guess: its decisions, its trace and its f / df call
// counts are those of the unspeculated run (the replay check holds with speculation on or off).
constexpr int SPEC_MAX = 4;

struct Request {
    int kind;
    int flags;
    int ncand;                   // guesses at the following trial steps (evaluated with this one), 0 = none
    double cand[SPEC_MAX - 1];
    int pre_tag;         // optional trace record of an evaluation whose value was already known
    int tr_tag;          // optional trace record emitted before the request is served
    double a, b;
    double pre_a, pre_b, pre_c;  // (emitted first; meaningful only when pre_tag is set)
    double tr_a, tr_b, tr_c;     // (meaningful only when tr_tag is set)
};

// keeps a value in a vector register, hides its origin from the optimiser
__device__ __forceinline__ void opaque(double& v) { asm("" : "+v"(v)); }

// Guesses at the trial steps the line search will ask for after the request the machine has just
// issued, from the machine's state alone: the pending trial (and every guess after it) is assumed
// to come back WORSE than the best point, a new line to go the way the previous one went.  Same
// operations as the machine's own (contraction off), so a guess that holds is the machine's next
// request bit for bit; one that does not is only wasted arithmetic.
// Round 2: a second assumption.  A trial that lands between the best point x and the second best w
// is assumed to become the new w (Brent then keeps bisecting: the case above); one that lands
// elsewhere -- beyond w, or on the other side of x -- is assumed worse than x, w AND v, which leaves
// all of Brent's state but the bracket as it is: the next step, secant or bisection, then follows
// exactly from what is known (same operations as CgdMachine::hot).  That works only while w, v and
// their slopes are known (`known`: lost with the first trial assumed to become w).  On ladybug the
// first rule alone predicts 80 % of Brent's steps, both together 84 %, and the second never spoils a
// guess the first would have got right.
struct Predictor {
    enum : int { P_STOP = 0, P_BR_FC, P_DB };
    int ph;
    bool need_first;   // Brent's first evaluation (at bx, whose slope is unknown) comes before its first trial
    bool known;        // w, v, dw, dv, d, e are those of the machine (given the assumptions so far)
    double a, b, x, dx;        // bracket (the pending trial already is one of its ends), best point, its slope
    double w, v, dw, dv, d, e; // the rest of Brent's state
    double uu;                 // the pending trial
    double ax, bx, cx;

    // swapped: the previous bracketing found f(1) > f(0) and went to the other side of 0
    __device__ void start(const struct CgdMachine& M, bool swapped);
    __device__ void begin_brent(double ax_, double bx_, double cx_, bool slope_known, double slope) {
        a = (ax_ < cx_ ? ax_ : cx_);
        b = (ax_ > cx_ ? ax_ : cx_);
        x = bx_;
        need_first = !slope_known;
        dx = slope_known ? slope : -1.0;   // the origin of a descent line: downhill to the right
        ph = P_DB;
        // Brent starts with w = v = x: its first trial becomes w whatever it returns
        known = false; w = v = x; dw = dv = dx; d = e = 0.0; uu = x;
    }
    // the next guess; false = none
    __device__ bool next(double& c) {
#pragma clang fp contract(off)
        const double GOLD = 1.618034, TOL = 3.0e-8, ZEPS = DBL_EPSILON * 1.0e-3;
        if (ph == P_BR_FC) {
            cx = bx + GOLD * (bx - ax);
            c = cx;
            // what follows a bracket that stands at once is known only when the search went back
            // across the origin (bx = 0: Brent starts there); otherwise parabolic steps follow
            if (bx == 0.0) begin_brent(ax, bx, cx, false, 0.0); else ph = P_STOP;
            return true;
        }
        if (ph != P_DB) return false;
        if (need_first) { need_first = false; c = x; return true; }
        const double xm = 0.5 * (a + b);
        const double tol1 = TOL * fabs(x) + ZEPS;
        const double tol2 = 2.0 * tol1;
        if (fabs(x - xm) <= (tol2 - 0.5 * (b - a))) { ph = P_STOP; return false; }
        const double ebis = (dx >= 0.0 ? a - x : b - x);
        // does the pending trial lie between x and w?  then it is taken to become w, and Brent to bisect
        const bool near = (uu - x) * (w - x) > 0.0 && fabs(uu - x) < fabs(w - x);
        const bool exact = known && w != x && v != x && v != w && !near;
        double dn = 0.5 * ebis, en = ebis;
        if (exact) {
            // worse than x, w and v: nrc :319-376 on the state as it stands (the operations of hot())
            const bool big = fabs(e) > tol1;
            const double dflt = 2.0 * (b - a);
            const double q1 = (w - x) * dx / (dx - dw);
            const double q2 = (v - x) * dx / (dx - dv);
            const double d1 = (dw != dx) ? q1 : dflt;
            const double d2 = (dv != dx) ? q2 : dflt;
            const double u1 = x + d1, u2 = x + d2;
            const bool ok1 = (a - u1) * (u1 - b) > 0.0 && dx * d1 <= 0.0;
            const bool ok2 = (a - u2) * (u2 - b) > 0.0 && dx * d2 <= 0.0;
            const double dsel = (ok1 && ok2) ? (fabs(d1) < fabs(d2) ? d1 : d2) : (ok1 ? d1 : d2);
            const bool accept = big && (ok1 || ok2) && (fabs(dsel) <= fabs(0.5 * e));
            const double ut = x + dsel;
            const double dacc = (ut - a < tol2 || b - ut < tol2) ? copysign(tol1, xm - x) : dsel;
            en = accept ? d : ebis;
            dn = accept ? dacc : 0.5 * ebis;
        } else {
            known = false;
        }
        const bool tn = !(fabs(dn) >= tol1);
        const double u = tn ? x + copysign(tol1, dn) : x + dn;
        c = u;
        d = dn; e = en; uu = u;
        if (tn) ph = P_STOP;
        else if (u < x) a = u; else b = u;
        return true;
    }
};

struct CgdMachine {
    enum : int {
        S_BEGIN, S_FIRST, S_GRAD0, S_BR_FA, S_BR_FB, S_BR_FC, S_BR_HEAD,
        S_BR_CASE1, S_BR_CASE2, S_BR_SHIFT, S_DB_START, S_DB_FIRST, S_DB_HEAD, S_DB_EVAL,
        S_AFTER_LINMIN, S_REDUCED, S_FINISH, S_ROLLED, S_DONE
    };
    int maxiters;
    double ftol;
    int st, its, iter, reason;
    double finit, fp, fret;
    bool saw_nan, rolled_back;
    long long nfeval, ngeval;
    // bracket (nrc :80-151); every trial also yields the slope along the line (it costs no extra
    // exchange), remembered so that Dbrent's first evaluation at bx need not be repeated
    double ax, bx, cx, fa, fb, fc, u;
    double sa, sb, sc;
    bool va, vb, vc;      // slope known at ax / bx / cx
    int pp_tag;           // pending "value already known" trace record
    double pp_a, pp_b, pp_c;
    // Brent with derivatives (nrc :284-404)
    double a, b, x, w, v, fx, fw, fv, dx, dw, dv, d, e, uu;
    int it;
    bool tiny;
    // noskip: evaluate even where the value is known bit for bit -- f at step 0 of a new line, f and slope at the
    // bracket's middle point.  Set by environments that emulate the reference's stale factor cache
    // (Variable.cpp:66-76, Factor.h:228-234): there the reference's repeated evaluation of the same point can
    // return another value (factors keep what they computed before their variables moved by less than 1e-12).
    bool noskip;

    __device__ void init(int maxiters_, double ftol_, bool noskip_ = false) {
        maxiters = maxiters_; ftol = ftol_; noskip = noskip_;
        st = S_BEGIN; its = 0; iter = 0; reason = EXIT_ITMAX;
        finit = 0.0; fp = 0.0; fret = DBL_MAX;
        saw_nan = false; rolled_back = false; nfeval = 0; ngeval = 0;
        ax = bx = cx = fa = fb = fc = u = 0.0;
        sa = sb = sc = 0.0; va = vb = vc = false;
        pp_tag = TR_NONE; pp_a = pp_b = pp_c = 0.0;
        a = b = x = w = v = fx = fw = fv = dx = dw = dv = d = e = uu = 0.0;
        it = 0; tiny = false;
    }
    __device__ int status() const { return reason | (rolled_back ? STATUS_ROLLED_BACK : 0); }

    __device__ static Request req(int kind, double a = 0.0, int flags = 0) {
        Request r; r.kind = kind; r.a = a; r.b = 0.0; r.flags = flags;
        r.ncand = 0;
        for (int k = 0; k < SPEC_MAX - 1; ++k) r.cand[k] = 0.0;
        r.pre_tag = TR_NONE; r.pre_a = r.pre_b = r.pre_c = 0.0;
        r.tr_tag = TR_NONE; r.tr_a = r.tr_b = r.tr_c = 0.0;
        return r;
    }
    __device__ Request with_pending(Request r) {
        r.pre_tag = pp_tag; r.pre_a = pp_a; r.pre_b = pp_b; r.pre_c = pp_c;
        pp_tag = TR_NONE;
        return r;
    }
    __device__ static Request traced(Request r, int tag, double ta, double tb, double tc) {
        r.tr_tag = tag; r.tr_a = ta; r.tr_b = tb; r.tr_c = tc;
        return r;
    }
    __device__ Request want_f(double at, bool line = true) { ++nfeval; return req(REQ_EVAL, at, line ? RF_LINE : 0); }
    __device__ Request want_fd(double at, int extra = 0) { ++nfeval; return req(REQ_EVAL, at, RF_SLOPE | RF_LINE | extra); }
    // the bracketing of a new line starts from (0, 1); f at a = 0 is the value the previous line
    // search ended on (or f(x0)): the same point, hence the same bits -- counted and traced like
    // the reference's call (nrc :87) but not evaluated again.  Its slope is not known.
    __device__ Request start_line(int pre_ops, double gamma) {
        iter = its;
        ax = 0.0; bx = 1.0;
        fa = fp; sa = 0.0; va = false;
        if (noskip) {   // the reference's fa = func(ax) (nrc :87), evaluated
            st = S_BR_FA;
            Request r = want_f(ax);
            r.flags |= pre_ops | RF_PRE_BEGIN;
            r.b = gamma;
            return r;
        }
        ++nfeval;
        pp_tag = TR_F; pp_a = 0.0; pp_b = fp; pp_c = 0.0;
        st = S_BR_FB;
        Request r = with_pending(want_fd(bx, pre_ops | RF_PRE_BEGIN));
        r.b = gamma;
        return r;
    }
    // a line minimisation ended at step x with value fx (nrc :646-654): unless converged, move
    // there, take the gradient and reduce it for the next direction in one request
    __device__ Request line_done() {
        if (2.0 * fabs(fx - fp) <= ftol * (fabs(fx) + fabs(fp) + 1.0e-18)) {
            st = S_AFTER_LINMIN;  // finds the same and finishes
            return traced(req(REQ_LINE_END, x), TR_LINMIN, x, fx, 0.0);
        }
        fret = fx;
        fp = fret;
        ++ngeval;
        st = S_REDUCED;
        Request r = req(REQ_GRAD, x, RF_PRE_LINE_END | RF_POST_REDUCE);
        r.b = fp;
        return traced(r, TR_LINMIN, x, fx, 0.0);
    }

    // Hot path: the reply to a Brent trial (nrc :380-401 housekeeping, then :319-376 for the next
    // trial point) -- ~85 % of all steps -- as a routine of its own: straight-line code with selects
    // instead of branches (a taken branch costs this single in-order wave more than the arithmetic
    // it skips) and exactly the operations of the branchy form in next() (S_DB_EVAL / S_DB_HEAD).
    // Everything it reads -- the Brent state, the two call counters, the pending trace record -- is
    // loaded up front in one batch (one LDS round trip), everything it changes is stored at the end.
    // Returns false, having changed nothing, when the generic code must take the step (not in a
    // Brent iteration, the closing states of a line search, the iteration limit); otherwise `un` is
    // the next trial step (request: value + slope on the line), pre_* the pending trace record to
    // emit first, and G the state the Predictor continues from.
    __device__ __forceinline__ bool hot(double r0, double r1, double& un, int& pre_tag, double& pre_a, double& pre_b,
                                        double& pre_c, Predictor& G) {
#pragma clang fp contract(off)
        const double TOL = 3.0e-8, ZEPS = DBL_EPSILON * 1.0e-3;
        const int DB_ITMAX = 100;
        double a_ = a, b_ = b, x_ = x, w_ = w, v_ = v, fx_ = fx, fw_ = fw, fv_ = fv;
        double dx_ = dx, dw_ = dw, dv_ = dv, d_ = d, e_ = e, uu_ = uu;
        long long nfe_ = nfeval, nge_ = ngeval;
        int ppt_ = pp_tag;
        double ppa_ = pp_a, ppb_ = pp_b, ppc_ = pp_c;
        const bool tiny_ = tiny;
        const int it_ = it;
        if (st != S_DB_EVAL) return false;
        opaque(a_); opaque(b_); opaque(x_); opaque(w_); opaque(v_); opaque(fx_); opaque(fw_); opaque(fv_);
        opaque(dx_); opaque(dw_); opaque(dv_); opaque(d_); opaque(e_); opaque(uu_);
        opaque(ppa_); opaque(ppb_); opaque(ppc_);
        asm("" : "+v"(nfe_), "+v"(nge_), "+v"(ppt_));
        if ((tiny_ && r0 > fx_) || !(it_ + 1 < DB_ITMAX)) return false;
        const double fu = r0, du = r1;
        const bool le = fu <= fx_;
        const bool right = uu_ >= x_, left = uu_ < x_;
        const double a1 = le ? (right ? x_ : a_) : (left ? uu_ : a_);
        const double b1 = le ? (right ? b_ : x_) : (left ? b_ : uu_);
        const bool c1 = !le && (fu <= fw_ || w_ == x_);
        const bool c2 = !le && !c1 && (fu < fv_ || v_ == x_ || v_ == w_);
        const bool vw = le || c1;           // v <- w
        const double v1 = vw ? w_ : (c2 ? uu_ : v_), fv1 = vw ? fw_ : (c2 ? fu : fv_), dv1 = vw ? dw_ : (c2 ? du : dv_);
        const double w1 = le ? x_ : (c1 ? uu_ : w_), fw1 = le ? fx_ : (c1 ? fu : fw_), dw1 = le ? dx_ : (c1 ? du : dw_);
        const double x1 = le ? uu_ : x_, fx1 = le ? fu : fx_, dx1 = le ? du : dx_;
        // S_DB_HEAD
        const double xm = 0.5 * (a1 + b1);
        const double tol1 = TOL * fabs(x1) + ZEPS;
        const double tol2 = 2.0 * tol1;
        if (fabs(x1 - xm) <= (tol2 - 0.5 * (b1 - a1))) return false;   // converged: next() redoes the step and ends the line
        const bool big = fabs(e_) > tol1;
        const double dflt = 2.0 * (b1 - a1);
        const double q1 = (w1 - x1) * dx1 / (dx1 - dw1);
        const double q2 = (v1 - x1) * dx1 / (dx1 - dv1);
        const double d1 = (dw1 != dx1) ? q1 : dflt;
        const double d2 = (dv1 != dx1) ? q2 : dflt;
        const double u1 = x1 + d1, u2 = x1 + d2;
        const bool ok1 = (a1 - u1) * (u1 - b1) > 0.0 && dx1 * d1 <= 0.0;
        const bool ok2 = (a1 - u2) * (u2 - b1) > 0.0 && dx1 * d2 <= 0.0;
        const double dsel = (ok1 && ok2) ? (fabs(d1) < fabs(d2) ? d1 : d2) : (ok1 ? d1 : d2);
        const bool accept = big && (ok1 || ok2) && (fabs(dsel) <= fabs(0.5 * e_));
        const double ut = x1 + dsel;
        const double dacc = (ut - a1 < tol2 || b1 - ut < tol2) ? copysign(tol1, xm - x1) : dsel;
        const double ebis = (dx1 >= 0.0 ? a1 - x1 : b1 - x1);
        const double enew = accept ? d_ : ebis;
        const double dnew = accept ? dacc : 0.5 * ebis;
        const bool tn = !(fabs(dnew) >= tol1);
        un = tn ? x1 + copysign(tol1, dnew) : x1 + dnew;
        if (fu != fu) saw_nan = true;
        a = a1; b = b1; v = v1; fv = fv1; dv = dv1; w = w1; fw = fw1; dw = dw1; x = x1; fx = fx1; dx = dx1;
        e = enew; d = dnew; tiny = tn; uu = un;
        it = it_ + 1;
        ngeval = nge_ + 1;   // the reply's df call (nrc :378); the new trial's operator() call (:366/:369)
        nfeval = nfe_ + 1;
        pp_tag = TR_NONE;
        pre_tag = ppt_; pre_a = ppa_; pre_b = ppb_; pre_c = ppc_;
        // the Predictor continues from here: the new trial is assumed to come back worse
        G.need_first = false;
        G.a = (!tn && un < x1) ? un : a1;
        G.b = (!tn && !(un < x1)) ? un : b1;
        G.x = x1; G.dx = dx1;
        G.known = true; G.w = w1; G.v = v1; G.dw = dw1; G.dv = dv1; G.d = dnew; G.e = enew; G.uu = un;
        G.ph = tn ? Predictor::P_STOP : Predictor::P_DB;
        return true;
    }

    // r0, r1, r2: reply to the previous request
    __device__ Request next(double r0, double r1, double r2) {
#pragma clang fp contract(off)
        const double GOLD = 1.618034, GLIMIT = 100.0, TINY = 1.0e-20;      // nrc :82
        const double GTOL = 1.0e-8;                                         // nrc :622 (EPS = 1e-18, :621, is in line_done)
        const double TOL = 3.0e-8;  // Dbrent's own default, not the solver's ftol (nrc :288, :499)
        const double ZEPS = DBL_EPSILON * 1.0e-3;
        const int DB_ITMAX = 100;
        for (;;) {
            switch (st) {
            case S_BEGIN:  // CGD .cpp:34-37: assign clamp(x0), initialFval = sfd(xval)
                st = S_FIRST;
                return want_f(0.0, false);
            case S_FIRST:
                finit = r0;
                if (r0 != r0) saw_nan = true;
                ++nfeval;            // Frprmn's own fp = func(p): same point, same value (nrc :628)
                fp = finit;
                ++ngeval;            // func.df(p, xi) (nrc :629)
                st = S_GRAD0;
                return traced(req(REQ_GRAD), TR_START, fp, 0.0, 0.0);
            case S_GRAD0:
                its = 0;
                return start_line(RF_PRE_START, 0.0);

            // ---- bracket from (0, 1) (start_line) -----------------------------------------
            case S_BR_FA:   // (noskip only)
                fa = r0; if (r0 != r0) saw_nan = true;
                st = S_BR_FB;
                return want_fd(bx);
            case S_BR_FB:
                fb = r0; sb = r1; vb = true; if (r0 != r0) saw_nan = true;
                if (fb > fa) {
                    double t = ax; ax = bx; bx = t; t = fa; fa = fb; fb = t;
                    t = sa; sa = sb; sb = t; const bool tv = va; va = vb; vb = tv;
                }
                cx = bx + GOLD * (bx - ax);
                st = S_BR_FC;
                return want_fd(cx);
            case S_BR_FC:
                fc = r0; sc = r1; vc = true; if (r0 != r0) saw_nan = true;
                st = S_BR_HEAD;
                break;
            case S_BR_HEAD: {
                if (!(fb > fc)) { st = S_DB_START; break; }
                const double r = (bx - ax) * (fb - fc);
                const double q = (bx - cx) * (fb - fa);
                const double qr = q - r;
                u = bx - ((bx - cx) * q - (bx - ax) * r) / (2.0 * copysign(fmax(fabs(qr), TINY), qr));
                const double ulim = bx + GLIMIT * (cx - bx);
                if ((bx - u) * (u - cx) > 0.0) { st = S_BR_CASE1; return want_fd(u); }
                if ((cx - u) * (u - ulim) > 0.0) { st = S_BR_CASE2; return want_fd(u); }
                if ((u - ulim) * (ulim - cx) >= 0.0) u = ulim;
                else u = cx + GOLD * (cx - bx);
                st = S_BR_SHIFT;
                return want_fd(u);
            }
            case S_BR_CASE1: {  // parabolic u between b and c
                const double fu = r0, su = r1; if (r0 != r0) saw_nan = true;
                if (fu < fc) { ax = bx; bx = u; fa = fb; fb = fu; sa = sb; va = vb; sb = su; vb = true; st = S_DB_START; break; }
                if (fu > fb) { cx = u; fc = fu; sc = su; vc = true; st = S_DB_START; break; }
                u = cx + GOLD * (cx - bx);
                st = S_BR_SHIFT;
                return want_fd(u);
            }
            case S_BR_CASE2: {  // parabolic u between c and its limit
                const double fu = r0, su = r1; if (r0 != r0) saw_nan = true;
                if (fu < fc) {
                    const double unew = u + GOLD * (u - cx);
                    bx = cx; cx = u; u = unew;
                    fb = fc; fc = fu;
                    sb = sc; vb = vc; sc = su; vc = true;
                    st = S_BR_SHIFT;
                    return want_fd(u);
                }
                ax = bx; bx = cx; cx = u;
                fa = fb; fb = fc; fc = fu;
                sa = sb; va = vb; sb = sc; vb = vc; sc = su; vc = true;
                st = S_BR_HEAD;
                break;
            }
            case S_BR_SHIFT: {
                const double fu = r0, su = r1; if (r0 != r0) saw_nan = true;
                ax = bx; bx = cx; cx = u;
                fa = fb; fb = fc; fc = fu;
                sa = sb; va = vb; sb = sc; vb = vc; sc = su; vc = true;
                st = S_BR_HEAD;
                break;
            }

            // ---- Brent with derivatives on the bracket ------------------------------
            case S_DB_START:
                a = (ax < cx ? ax : cx);
                b = (ax > cx ? ax : cx);
                x = w = v = bx;
                d = 0.0; e = 0.0; it = 0;
                if (vb && !noskip) {  // value and slope at bx are known from the bracketing: same point, same bits
                    ++nfeval; ++ngeval;  // the reference's funcd(x), funcd.df(x) (nrc :314-315)
                    fx = fb; dx = sb;
                    fw = fv = fx; dw = dv = dx;
                    pp_tag = TR_FD; pp_a = x; pp_b = fx; pp_c = dx;
                    st = S_DB_HEAD;
                    break;
                }
                st = S_DB_FIRST;
                return want_fd(x);
            case S_DB_FIRST:
                fx = r0; dx = r1; if (r0 != r0) saw_nan = true;
                ++ngeval;
                fw = fv = fx;
                dw = dv = dx;
                st = S_DB_HEAD;
                break;
            case S_DB_HEAD: {
                if (it >= DB_ITMAX) { reason = EXIT_DBRENT_ITMAX; st = S_FINISH; break; }  // nrc :403 throws
                const double xm = 0.5 * (a + b);
                const double tol1 = TOL * fabs(x) + ZEPS;
                const double tol2 = 2.0 * tol1;
                if (fabs(x - xm) <= (tol2 - 0.5 * (b - a))) return with_pending(line_done());
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
                            const double ut = x + d;
                            if (ut - a < tol2 || b - ut < tol2) d = copysign(tol1, xm - x);
                            bisect = false;
                        }
                    }
                }
                if (bisect) { e = (dx >= 0.0 ? a - x : b - x); d = 0.5 * e; }
                if (fabs(d) >= tol1) { uu = x + d; tiny = false; }
                else { uu = x + copysign(tol1, d); tiny = true; }
                st = S_DB_EVAL;
                return with_pending(want_fd(uu));
            }
            case S_DB_EVAL: {
                const double fu = r0, du = r1; if (r0 != r0) saw_nan = true;
                if (tiny && fu > fx) return line_done();  // the minimal downhill step goes uphill: done (no df call, nrc :369-376)
                ++ngeval;
                if (fu <= fx) {
                    if (uu >= x) a = x; else b = x;
                    v = w; fv = fw; dv = dw;
                    w = x; fw = fx; dw = dx;
                    x = uu; fx = fu; dx = du;
                } else {
                    if (uu < x) a = uu; else b = uu;
                    if (fu <= fw || w == x) {
                        v = w; fv = fw; dv = dw;
                        w = uu; fw = fu; dw = du;
                    } else if (fu < fv || v == x || v == w) {
                        v = uu; fv = fu; dv = du;
                    }
                }
                ++it;
                st = S_DB_HEAD;
                break;
            }

            // ---- Frprmn's loop body after linmin --------------------------------------
            case S_AFTER_LINMIN:  // only reached when line_done() found the ftol test satisfied
                fret = fx;
                reason = EXIT_FTOL; st = S_FINISH;
                break;
            case S_REDUCED: {
                const double test = r0, gg = r1, dgg = r2;
                Request nx;
                if (test < GTOL) { reason = EXIT_GTOL; st = S_FINISH; nx = finish_request(); }
                else if (gg == 0.0) { reason = EXIT_GGZERO; st = S_FINISH; nx = finish_request(); }
                else if (its + 1 >= maxiters) { reason = EXIT_ITMAX; st = S_FINISH; nx = finish_request(); }  // the final
                    // direction update of the reference has no observable effect: skipped
                else { ++its; nx = start_line(RF_PRE_UPDATE, dgg / gg); nx.flags |= RF_TR_FIRST; }
                return traced(nx, TR_ITER, test, gg, dgg);
            }

            // ---- CGD wrapper: rollback on negative progress (.cpp:61-80) -----------------
            case S_FINISH: {
                Request nx = finish_request();
                return nx;
            }
            case S_ROLLED:
                fret = r0;
                st = S_DONE;
                return req(REQ_DONE);
            default:
                return req(REQ_DONE);
            }
        }
    }

    __device__ Request finish_request() {
        if (saw_nan) { reason = EXIT_NAN; fret = DBL_MAX; }
        if (fret > finit || saw_nan) {
            rolled_back = true;
            st = S_ROLLED;
            Request r = want_f(0.0, false);
            r.flags |= RF_RESTORE;
            return r;
        }
        st = S_DONE;
        return req(REQ_DONE);
    }
};

__device__ inline void Predictor::start(const CgdMachine& M, bool swapped) {
#pragma clang fp contract(off)
    ph = P_STOP; need_first = false; known = false;
    a = b = x = dx = 0.0; ax = bx = cx = 0.0;
    w = v = dw = dv = d = e = uu = 0.0;
    switch (M.st) {
    case CgdMachine::S_BR_FB:   // pending: f at bx (= 1) of a new line from ax (= 0)
        if (swapped) { ax = M.bx; bx = M.ax; ph = P_BR_FC; }
        else { ax = M.ax; bx = M.bx; ph = P_BR_FC; }
        break;
    case CgdMachine::S_BR_FC:   // pending: f at cx; if it is no better than f(bx) the bracket stands
        begin_brent(M.ax, M.bx, M.cx, M.vb, M.sb);
        break;
    case CgdMachine::S_DB_FIRST:  // pending: value and slope at x = bx (only ever the line's origin)
        a = M.a; b = M.b; x = M.x; dx = -1.0; ph = P_DB;
        w = v = x; dw = dv = dx; uu = x;
        break;
    case CgdMachine::S_DB_EVAL:   // pending: a Brent trial at uu
        if (M.tiny) break;        // worse after a minimal step ends the line search
        a = M.a; b = M.b; x = M.x; dx = M.dx; ph = P_DB;
        if (M.uu < x) a = M.uu; else b = M.uu;
        known = true; w = M.w; v = M.v; dw = M.dw; dv = M.dv; d = M.d; e = M.e; uu = M.uu;
        break;
    default: break;
    }
}

// The driver: the single place where each heavy operation is instantiated.
// Env provides eval_value / eval_value_slope / gradient_to_xi / cg_start /
// line_begin / line_end / cg_reduce / cg_update / trace / aborted.  Replies (values, slopes,
// reductions) and aborted() need only be valid in wave 0, the one wave that consumes them.
//
// The machine's ~40 doubles of state live in LDS and only wave 0 of a workgroup
// steps it (the other waves would compute the same thing); the request is handed to
// the other waves through a double-buffered LDS slot.  Nothing of the control logic
// is therefore live in registers across the factor arithmetic.
// one step of the machine (wave 0 only); the request goes to the other waves through LDS
__device__ __forceinline__ void step_machine(CgdMachine* __restrict__ M, Request* __restrict__ out, double r0, double r1, double r2,
                                             bool writer) {
    double un, pa, pb, pc;
    int ptag;
    Predictor G;
    if (M->hot(r0, r1, un, ptag, pa, pb, pc, G)) {
        if (writer) {
            out->kind = REQ_EVAL; out->flags = RF_SLOPE | RF_LINE; out->ncand = 0;
            out->pre_tag = ptag; out->tr_tag = TR_NONE; out->a = un;
            if (ptag != TR_NONE) { out->pre_a = pa; out->pre_b = pb; out->pre_c = pc; }
        }
        return;
    }
    const Request nq = M->next(r0, r1, r2);
    if (writer) {
        out->kind = nq.kind; out->flags = nq.flags; out->ncand = 0;
        out->pre_tag = nq.pre_tag; out->tr_tag = nq.tr_tag; out->a = nq.a; out->b = nq.b;
        if (nq.pre_tag != TR_NONE) { out->pre_a = nq.pre_a; out->pre_b = nq.pre_b; out->pre_c = nq.pre_c; }
        if (nq.tr_tag != TR_NONE) { out->tr_a = nq.tr_a; out->tr_b = nq.tr_b; out->tr_c = nq.tr_c; }
    }
}

__device__ __forceinline__ double uniform(double v) {  // a wave-uniform value into scalar registers
    const long long b = __double_as_longlong(v);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)((unsigned long long)b >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// The cooperating group is normally a workgroup whose wave 0 steps the machine (Env::UNIFORM: the
// request is the same for every lane of a wave and is moved to scalar registers).  The quad solver
// (solver_quad.hpp) runs sixteen independent machines per wave instead: requests differ between
// lanes, the stepper is the first lane of each group of four.  Env supplies stepper(), writer()
// (the one lane that stores the request) and sync() (what orders that store before the reads).
template <class Env>
__device__ __forceinline__ double uni(double v) {
    if constexpr (Env::UNIFORM) return uniform(v); else return v;
}
template <class Env>
__device__ __forceinline__ int uni(int v) {
    if constexpr (Env::UNIFORM) return __builtin_amdgcn_readfirstlane(v); else return v;
}

// entry j of a small register array (j wave-uniform; a dynamic index would go through scratch)
template <int N>
__device__ __forceinline__ double pick(const double (&v)[N], int j) {
    double r = v[0];
#pragma unroll
    for (int k = 1; k < N; ++k) r = (j == k) ? v[k] : r;
    return r;
}
__device__ __forceinline__ bool same_bits(double a, double b) { return __double_as_longlong(a) == __double_as_longlong(b); }

// Env::NOSKIP (optional, default false): CgdMachine::noskip
template <class E, class = void> struct EnvNoSkip { static constexpr bool value = false; };
template <class E> struct EnvNoSkip<E, std::void_t<decltype(E::NOSKIP)>> { static constexpr bool value = E::NOSKIP; };

// Env::noskip_rt() (optional): the same decided at run time (solver_coop.hpp: the parity option with the stale-cache emulation)
template <class E, class = void> struct EnvNoSkipRt { static constexpr bool value = false; };
template <class E> struct EnvNoSkipRt<E, std::void_t<decltype(std::declval<const E&>().noskip_rt())>> { static constexpr bool value = true; };
template <class Env>
__device__ __forceinline__ bool env_noskip(const Env& E) {
    if constexpr (EnvNoSkipRt<Env>::value) return EnvNoSkip<Env>::value || E.noskip_rt();
    else return EnvNoSkip<Env>::value;
}

// Env::FUSED_GRADIENT (optional, default false): the environment offers gradient_fused(line_end, amin, reduce, fp, test, gg, dgg)
template <class E, class = void> struct EnvFusedGradient { static constexpr bool value = false; };
template <class E> struct EnvFusedGradient<E, std::void_t<decltype(E::FUSED_GRADIENT)>> { static constexpr bool value = E::FUSED_GRADIENT; };

// Env::SPEC: trial steps evaluated per value+slope request (1 = no speculation).  With SPEC > 1 Env
// provides eval_value_slope_spec(steps, f, s) for SPEC steps at once and spec_hint() / spec_note().
template <class Env>
__device__ __forceinline__ void run_machine(Env& E, CgdMachine& M /* LDS */, Request (&Q)[2] /* LDS */,
                                            int maxiters, double ftol) {
    constexpr int K = Env::SPEC;
    constexpr int KS = K > 1 ? K - 1 : 1;
    static_assert(K >= 1 && K <= SPEC_MAX, "speculation depth");
    double r0 = 0.0, r1 = 0.0, r2 = 0.0;
    double sc[KS], sf[KS], ss[KS];   // guessed steps evaluated with the last request, their values and slopes
    int ns = 0;                      // ... how many of them
    bool swapped = true;             // did the last bracketing go to the other side of the origin?
#pragma unroll
    for (int k = 0; k < KS; ++k) sc[k] = sf[k] = ss[k] = 0.0;
    // ONE lane steps the machine: its state is wave-uniform, and sixty-four lanes storing the same
    // words to the same LDS addresses is a 64-way bank conflict on every store (measured: 1700 of the
    // 2300 cycles of a step); the request reaches everybody, this lane's own wave included, through LDS
    const bool stepper = E.stepper() && E.writer();
    if (stepper) M.init(maxiters, ftol, env_noskip(E));
    for (int round = 0;; ++round) {
        const long long ts0 = E.clock();
        if (stepper) {
            if constexpr (K > 1) {
                Request* out = &Q[round & 1];
                const bool writer = E.writer();
                for (int j = 0;; ++j) {
                    Predictor G;
                    Request nq;
                    double un, pa, pb, pc;
                    int ptag;
                    const bool was_hot = M.hot(r0, r1, un, ptag, pa, pb, pc, G);
                    if (was_hot) {
                        if (j < ns && same_bits(un, pick(sc, j))) {
                            // the step asked for was guessed and has been evaluated: reply at once
                            if (ptag != TR_NONE) E.trace(ptag, pa, pb, pc);
                            r0 = pick(sf, j); r1 = pick(ss, j);
                            E.trace(TR_FD, un, r0, r1);
                            E.tick(18, 1);
                            continue;
                        }
                    } else {
                        nq = M.next(r0, r1, r2);
                        if (M.st == CgdMachine::S_BR_FC) swapped = M.ax == 1.0;
                        if (j < ns && nq.kind == REQ_EVAL && nq.flags == (RF_SLOPE | RF_LINE) && same_bits(nq.a, pick(sc, j))) {
                            if (nq.pre_tag != TR_NONE) E.trace(nq.pre_tag, nq.pre_a, nq.pre_b, nq.pre_c);
                            if (nq.tr_tag != TR_NONE) E.trace(nq.tr_tag, nq.tr_a, nq.tr_b, nq.tr_c);
                            r0 = pick(sf, j); r1 = pick(ss, j);
                            E.trace(TR_FD, nq.a, r0, r1);
                            E.tick(18, 1);
                            continue;
                        }
                        G.start(M, swapped);
                    }
                    // hand the request over, with guesses at what follows a value+slope trial
                    const bool slope_req = was_hot || (nq.kind == REQ_EVAL && (nq.flags & RF_SLOPE));
                    int nc = 0;
                    double cg[KS];
#pragma unroll
                    for (int k = 0; k < KS; ++k) cg[k] = 0.0;
                    if (slope_req && E.spec_on()) {
#pragma unroll
                        for (int k = 0; k < KS; ++k) {
                            double c;
                            if (nc == k && G.next(c)) { cg[k] = c; nc = k + 1; }
                        }
                    }
                    if (writer) {
                        out->ncand = nc;
#pragma unroll
                        for (int k = 0; k < KS; ++k) out->cand[k] = cg[k];
                        if (was_hot) {
                            out->kind = REQ_EVAL; out->flags = RF_SLOPE | RF_LINE;
                            out->pre_tag = ptag; out->tr_tag = TR_NONE; out->a = un;
                            if (ptag != TR_NONE) { out->pre_a = pa; out->pre_b = pb; out->pre_c = pc; }
                        } else {
                            out->kind = nq.kind; out->flags = nq.flags;
                            out->pre_tag = nq.pre_tag; out->tr_tag = nq.tr_tag; out->a = nq.a; out->b = nq.b;
                            if (nq.pre_tag != TR_NONE) { out->pre_a = nq.pre_a; out->pre_b = nq.pre_b; out->pre_c = nq.pre_c; }
                            if (nq.tr_tag != TR_NONE) { out->tr_a = nq.tr_a; out->tr_b = nq.tr_b; out->tr_c = nq.tr_c; }
                        }
                    }
                    break;
                }
            } else {
                step_machine(&M, &Q[round & 1], r0, r1, r2, E.writer());
            }
            if (E.aborted() && E.writer()) {  // only this wave is certain to know
                Q[round & 1].kind = REQ_DONE;
                // the restored start is what is returned: its value, no progress (CGD .cpp:66-80)
                M.reason = EXIT_SYNC_TIMEOUT; M.rolled_back = true; M.fret = M.finit;
            }
        }
        ns = 0;
        const long long ts1 = E.clock();
        E.sync();
        const Request& q = Q[round & 1];
        const int kind = uni<Env>(q.kind);
        const int flags = uni<Env>(q.flags);
        if (E.tracing()) {
            if ((flags & RF_TR_FIRST) && q.tr_tag != TR_NONE) E.trace(q.tr_tag, q.tr_a, q.tr_b, q.tr_c);
            if (q.pre_tag != TR_NONE) E.trace(q.pre_tag, q.pre_a, q.pre_b, q.pre_c);
            if (!(flags & RF_TR_FIRST) && q.tr_tag != TR_NONE) E.trace(q.tr_tag, q.tr_a, q.tr_b, q.tr_c);
        }
        const double qa = uni<Env>(q.a);
        E.tick(8, ts1 - ts0); E.tick(9, E.clock() - ts1);
        if (kind == REQ_DONE) {
            // a group of the quad solver writes its results and carries on with the next component
            if constexpr (Env::UNIFORM) break;
            else {
                if (!E.next_problem(M)) break;
                if (stepper) M.init(maxiters, ftol, env_noskip(E));
                r0 = r1 = r2 = 0.0;
                continue;
            }
        }
        const long long th0 = E.clock();
#define RDIS_TICK_KIND(K_) E.tick(12 + K_, E.clock() - th0); E.tick(22 + K_, 1)
        switch (kind) {
        case REQ_EVAL:
            if (flags & (RF_PRE_START | RF_PRE_UPDATE | RF_PRE_BEGIN)) {
                if (flags & RF_PRE_START) E.cg_start();
                if (flags & RF_PRE_UPDATE) E.cg_update(uni<Env>(q.b));
                if (flags & RF_PRE_BEGIN) E.line_begin();
            }
            if (flags & RF_SLOPE) {
                if constexpr (K > 1) {
                    // the step asked for and the guesses at the following ones, evaluated together
                    // (a missing guess repeats the step: one code path, the value is not used)
                    const int nc = uni<Env>(q.ncand);
                    double ca[K], cf[K], cs[K];
                    ca[0] = qa;
#pragma unroll
                    for (int k = 1; k < K; ++k) ca[k] = (k <= nc) ? uni<Env>(q.cand[k - 1]) : qa;
                    E.eval_value_slope_spec(ca, cf, cs);
                    r0 = uni<Env>(cf[0]); r1 = uni<Env>(cs[0]);
#pragma unroll
                    for (int k = 1; k < K; ++k) { sc[k - 1] = ca[k]; sf[k - 1] = uni<Env>(cf[k]); ss[k - 1] = uni<Env>(cs[k]); }
                    ns = nc;
                    E.tick(19, nc);
                } else {
                    E.eval_value_slope(qa, r0, r1);
                    r0 = uni<Env>(r0); r1 = uni<Env>(r1);
                }
                E.trace(TR_FD, qa, r0, r1);
                RDIS_TICK_KIND(1);
            } else {
                r0 = uni<Env>(E.eval_value(qa, (flags & RF_RESTORE) != 0));
                if (flags & RF_LINE) E.trace(TR_F, qa, r0, 0.0);
                RDIS_TICK_KIND(0);
            }
            break;
        case REQ_GRAD:
            if constexpr (EnvFusedGradient<Env>::value) {
                // one pass of the environment over its variables for the three (solver_ptm.hpp: they stream from HBM)
                const bool red = (flags & RF_POST_REDUCE) != 0;
                E.gradient_fused((flags & RF_PRE_LINE_END) != 0, qa, red, uni<Env>(q.b), r0, r1, r2);
                if (red) { r0 = uni<Env>(r0); r1 = uni<Env>(r1); r2 = uni<Env>(r2); }
            } else {
                if (flags & RF_PRE_LINE_END) E.line_end(qa);
                E.gradient_to_xi();
                if (flags & RF_POST_REDUCE) {
                    E.cg_reduce(uni<Env>(q.b), r0, r1, r2);
                    r0 = uni<Env>(r0); r1 = uni<Env>(r1); r2 = uni<Env>(r2);
                }
            }
            RDIS_TICK_KIND(2);
            break;
        case REQ_LINE_END: E.line_end(qa); RDIS_TICK_KIND(5); break;
        default: break;
        }
#undef RDIS_TICK_KIND
    }
    E.sync();  // M is final and visible to every lane of the group
}

}  // namespace rdis_hip
