// solver_coop.hpp -- a group of workgroups solves ONE large bundle-adjustment component; a launch
// runs one group or several side by side (CoopGroup, below).
//
// ladybug-49-7776 as a single component is 31843 factors over 23769 free
// variables: ~1 MB of state and ~800 dependent objective evaluations per solve.
// That is a latency problem, not a bandwidth problem, so the design removes
// memory traffic and launches from the evaluation loop altogether:
//
//  * one persistent launch runs the entire CGDSubspaceOptimizer::optimize call
//    (reference src/optimizers/CGDSubspaceOptimizer.cpp:19-98); wave 0 of every
//    workgroup steps the scalar control logic (minimizer.hpp) on the same reduced
//    values, so every workgroup takes the same branches and nothing is broadcast
//    between workgroups;
//  * lane j owns factor j: for the duration of a line minimisation the point and
//    direction of its 12 variables, their bounds and its observation live in
//    registers; a trial step a costs clamp(base + a*dir), ~310 fp64 instructions,
//    and zero memory traffic;
//  * the CG recurrence (p, xi, g, h) of free variable i lives in the registers of
//    its owner: lane i for variables fed by few factors (points), a whole wave for
//    variables fed by many (cameras: up to 906 partials) so that the per-variable
//    sum of partials is one coalesced sweep + wave reduction instead of a serial
//    chain.  Only xi is published (once per CG iteration) for the factor lanes;
//  * the per-factor partials of a full gradient are scattered straight into
//    variable-major order (PlanView::slot_pos), so every owner reads one
//    contiguous run; partials and xi cross workgroups through agent-scope coherent
//    stores / loads, the barrier in between only drains the stores;
//  * workgroups exchange partial sums through 8-byte granules in HBM, the data
//    being the flag (grid_sync.hpp); a line-search value is delivered to the
//    stepping wave only.
//
// Launched with hipLaunchCooperativeKernel so that an oversized grid is rejected
// instead of deadlocking; every workgroup of a group must be resident, so the host packs groups
// into launches of at most the resident capacity.  Bundle adjustment only (fixed arity 12).
#pragma once
#include "grid_sync.hpp"

namespace rdis_hip {

struct CoopArgs {
    long long* timing;     // [8] cycle accumulators written by lane 0 (profiling aid)
    CoopState* st;
    const int* slot_li;    // [12 * m]: local free index of each factor slot or -1 (constant)
    const int* lane_var;   // [nwg * threads]: free variable owned by this lane or -1
    const int* wave_var;   // [nwg * threads / 64]: free variable owned by this wave or -1
    double* xi_glob;       // [n] published search direction
    int comp;
    int poll_delay;        // x64 cycles between publishing and the first sweep (a store needs about that long to land)
    int speculate;         // evaluate guesses at the following trial steps with every line-search trial (minimizer.hpp)
    int reference_slope;   // plan option factor_rounding = 1, the PARITY option: every sum a trial or a CG iteration forms is added in the
                           // reference's order -- the objective over the factors in list order, every variable's partials in factor-list
                           // order, gradient times direction and the Polak-Ribiere sums over the variables in list order (ordered_sums, below;
                           // looked at by the reference-rounding instantiation only, refround_kernels.hip)
    int stale;             // ... with it, plan option emulate_stale_cache: the reference's factor cache (Variable.cpp:66-76, Factor.h:228-234)
};

// the CG recurrence of one free variable
struct VarState {
    int li;  // local free index, -1 = none
    double p, xi, g, h, xinit, lo, hi;
};

struct CoopEnv {
    const ProblemView& P;
    const PlanView& L;
    const CoopArgs& A;
    int n, m, f0, c0;
    int gt, tid;              // global lane, lane in workgroup
    GridSync X;
    double* tr;
    int trn, lm_count;
    // factor lane state
    bool has_fac;
    int fid;
    double base[12], dirv[12], lov[12], hiv[12];
    double ox, oy;
    VarState lv;   // lane-owned variable
    VarState wv;   // wave-owned variable (identical in all 64 lanes)
#ifdef RDIS_REFERENCE_SLOPE
    // The reference's factor cache as this lane's factor sees it (A.stale): the values its twelve variables were last assigned,
    // whether one of them has moved by 1e-12 or more since the factor's value was last computed, and that value.  Every lane sees
    // every assignment (each trial, each gradient), so a variable's history is the same in all the lanes that read it.
    double prevv[12], fcache;
    bool fdirty;
    __device__ __forceinline__ void note_assign(const double (&v)[12]) {
        if (!A.stale) return;
#pragma unroll
        for (int k = 0; k < 12; ++k) {
            if (!(fabs(v[k] - prevv[k]) < 1e-12)) fdirty = true;   // Variable::assign (src/Variable.cpp:70-76)
            prevv[k] = v[k];
        }
    }
    __device__ __forceinline__ double cached_value(double fj) {   // Factor::eval -> evalFactorCached (src/Factor.h:228-234)
        if (!A.stale) return fj;
        if (fdirty) { fcache = fj; fdirty = false; return fj; }
        return fcache;
    }
    __device__ bool noskip_rt() const { return A.reference_slope != 0 && A.stale != 0; }
#endif

    __device__ void trace(int tag, double a, double b, double c) {
        if (tr != nullptr && gt == 0) {
            if (trn < L.trace_cap) { double* r = tr + 4ll * trn; r[0] = (double)tag; r[1] = a; r[2] = b; r[3] = c; }
            ++trn;
        }
    }
    static constexpr bool UNIFORM = true;
    static constexpr int SPEC = COOP_SPEC;   // trial steps per exchange (minimizer.hpp: speculation)
#ifdef RDIS_REFERENCE_SLOPE
    __device__ bool spec_on() const { return A.speculate != 0 && A.reference_slope == 0; }   // (the reference's slope is a gradient pass of the whole group: one trial at a time)
#else
    __device__ bool spec_on() const { return A.speculate != 0; }
#endif
    __device__ bool stepper() const { return threadIdx.x < 64; }
    __device__ bool writer() const { return (threadIdx.x & 63) == 0; }
    __device__ void sync() const { __syncthreads(); }
    __device__ bool tracing() const { return tr != nullptr; }
    __device__ bool aborted() const { return X.dead; }
    __device__ void tick(int slot, long long dt) {
#ifdef RDIS_COOP_TIMING
        X.tm[slot] += dt;
#endif
    }
    __device__ long long clock() const { return coop_clock(); }

    // ---- evaluation at clamp(base + a*dir), straight from registers ---------------
    template <bool SLOPE>
    __device__ void eval_line(double a, double& f, double& s) {
        double fj = 0.0, sj = 0.0, dummy = 0.0;
        const long long tc0 = coop_clock();
        if (has_fac) {
            double v[12];
            {
#pragma clang fp contract(off)
#pragma unroll
                for (int k = 0; k < 12; ++k) {
                    const double t = a * dirv[k];
                    v[k] = clampd(base[k] + t, lov[k], hiv[k]);
                }
            }
#ifdef RDIS_REFERENCE_SLOPE
            if (A.reference_slope) note_assign(v);
#endif
            if constexpr (SLOPE) {
                double g[12];
                fj = ba_eval_grad(v, ox, oy, g);
#ifdef RDIS_REFERENCE_SLOPE
                if (A.reference_slope) scatter_partials(g);
                else
#endif
                {
                    double acc = 0.0;
#pragma unroll
                    for (int k = 0; k < 12; ++k) acc = __builtin_fma(g[k], dirv[k], acc);   // (fused on purpose, not by the compiler's leave: the tests' CPU restatement mirrors exactly this)
                    sj = acc;
                }
            } else {
                fj = ba_eval(v, ox, oy);
            }
#ifdef RDIS_REFERENCE_SLOPE
            if (A.reference_slope) {
                fj = cached_value(fj);
                store_f64<true>(L.seq_val + c0 + gt, fj);
            }
#endif
        }
#ifdef RDIS_REFERENCE_SLOPE
        if (A.reference_slope) ordered_sums<SLOPE>(fj, sj);
#endif
        X.tm[0] += coop_clock() - tc0;
        X.to_wave0<SLOPE ? 2 : 1>(fj, sj, dummy, SYNC_NONE);  // only the stepping wave consumes a line-search value
        X.finish_wave0(SYNC_NONE);
        f = fj; s = sj;
    }
#ifdef RDIS_REFERENCE_SLOPE
    // The slope of a trial the way the reference forms it (Df1dim::df, minimize_nrc.h:439-447, over SubfunctionFD::df,
    // CGDSubspaceOptimizer.cpp:135-157; solver_lds.hpp's slope_reference has the history): the GRADIENT at the trial point --
    // every variable's partials added in factor-list order (src/State.h:157-210), a camera's 900 too, by ONE lane -- and then
    // gradient times direction over the n variables in list order, every product rounded before it is added, by ONE lane of
    // the group: a sequential sum of 23 769 terms on full ladybug.  Nothing a reduction over 256 compute units can reproduce
    // (on this problem the population of end values follows the ORDER of that sum), so the parity option
    // (plan option factor_rounding = 1 -> CoopArgs::reference_slope, in the instantiation of refround_kernels.hip) pays for it:
    // about 0.2 ms a trial.  The products travel through
    // xi_glob, which nothing reads between line_begin and the next publish_xi.
    __device__ void scatter_partials(const double (&g)[12]) {
        const int* sp = L.slot_pos + L.slot_base[c0 + gt];
        int t[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) t[k] = sp[k];
#pragma unroll
        for (int k = 0; k < 12; ++k)
            if (t[k] >= 0) store_f64<true>(L.gfac + t[k], g[k]);
    }
    // g[b .. e) added in order by a whole wave: 64 entries per (coalesced, coherent) load, the next 64 in flight; the wave parks
    // them in LDS and its first lane reads them back sixteen at a time and adds them -- one dependent chain, the bits of the
    // plain loop (entries past the end are zeros: adding them changes nothing); the same value in all lanes
    static __device__ __forceinline__ double wave_sum_in_order(const double* g, int b, int e, int tid) {
#pragma clang fp contract(off)
        __shared__ __attribute__((aligned(16))) double park[16][64];
        const int lane = tid & 63;
        double* mine = park[(tid >> 6) & 15];
        double acc = 0.0;
        double cur = b + lane < e ? load_f64<true>(g + b + lane) : 0.0;
        for (int j0 = b; j0 < e; j0 += 64) {
            const double nxt = j0 + 64 + lane < e ? load_f64<true>(g + j0 + 64 + lane) : 0.0;
            mine[lane] = cur;
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) {
                const double2* q = reinterpret_cast<const double2*>(mine);
#pragma unroll
                for (int j = 0; j < 32; j += 8) {
                    double2 t[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) t[k] = q[j + k];
#pragma unroll
                    for (int k = 0; k < 8; ++k) { acc = acc + t[k].x; acc = acc + t[k].y; }
                }
            }
            __builtin_amdgcn_wave_barrier();
            cur = nxt;
        }
        return __shfl(acc, 0);
    }
    // The objective is added the same way (OptimizableFunction::evalFactors, src/OptimizableFunction.cpp:95-135: the listed factors'
    // values one after the other): every factor lane has left its value -- the cached one under A.stale -- in seq_val, and the
    // group's first wave adds the m of them in list order.  On return the group's first lane holds the sums, every other lane zeros.
    template <bool SLOPE>
    __device__ void ordered_sums(double& fsum, double& ssum) {
#pragma clang fp contract(off)
        X.barrier(SYNC_DRAIN);
        if constexpr (SLOPE) {
            const int* vp = L.v2s_ptr + f0;
            if (lv.li >= 0) {
                const double t = run_sum_ordered<true>(L.gfac, vp[lv.li], vp[lv.li + 1]) * lv.xi;
                store_f64<true>(A.xi_glob + lv.li, t);
            }
            if (wv.li >= 0) {
                const double t = wave_sum_in_order(L.gfac, vp[wv.li], vp[wv.li + 1], tid) * wv.xi;
                if ((tid & 63) == 0) store_f64<true>(A.xi_glob + wv.li, t);
            }
            X.barrier(SYNC_DRAIN);
        }
        double af = 0.0, as = 0.0;
        if (gt < 64) {   // (the group's first wave)
            af = wave_sum_in_order(L.seq_val + c0, 0, m, tid);
            if constexpr (SLOPE) as = wave_sum_in_order(A.xi_glob, 0, n, tid);
        }
        fsum = gt == 0 ? af : 0.0;   // (the exchange's tree adds zeros to them)
        ssum = gt == 0 ? as : 0.0;
    }
#endif
    // SPEC trial steps at once: the evaluations are independent chains in one instruction stream
    // (this wave is alone on its SIMD; a single evaluation is bound by instruction latency), and
    // they share one exchange
    __device__ void eval_value_slope_spec(const double (&ca)[SPEC], double (&cf)[SPEC], double (&cs)[SPEC]) {
        double r[2 * SPEC];
#pragma unroll
        for (int k = 0; k < 2 * SPEC; ++k) r[k] = 0.0;
        const long long tc0 = coop_clock();
        if (has_fac) {
#pragma unroll
            for (int c = 0; c < SPEC; ++c) {
                double v[12], g[12];
                {
#pragma clang fp contract(off)
#pragma unroll
                    for (int k = 0; k < 12; ++k) {
                        const double t = ca[c] * dirv[k];
                        v[k] = clampd(base[k] + t, lov[k], hiv[k]);
                    }
                }
                r[2 * c] = ba_eval_grad(v, ox, oy, g);
                double acc = 0.0;
#pragma unroll
                for (int k = 0; k < 12; ++k) acc = __builtin_fma(g[k], dirv[k], acc);   // (fused on purpose, not by the compiler's leave: the tests' CPU restatement mirrors exactly this)
                r[2 * c + 1] = acc;
            }
        }
        X.tm[0] += coop_clock() - tc0;
        X.to_wave0_n<2 * SPEC, 0>(r, SYNC_NONE);
        X.finish_wave0(SYNC_NONE);
#pragma unroll
        for (int c = 0; c < SPEC; ++c) { cf[c] = r[2 * c]; cs[c] = r[2 * c + 1]; }
    }
    __device__ double eval_value(double a, bool restore) {
#ifdef RDIS_REFERENCE_SLOPE
        if (restore && A.reference_slope && has_fac) {   // the end point is assigned before the rollback is considered (CGD .cpp:61)
            double v[12];
#pragma unroll
            for (int k = 0; k < 12; ++k) v[k] = clampd(base[k], lov[k], hiv[k]);
            note_assign(v);
        }
#endif
        if (restore) load_base(L.xstart + f0);  // objective at clamp(x_init) for the rollback
        double f, s;
        eval_line<false>(restore ? 0.0 : a, f, s);
        return f;
    }
    __device__ void eval_value_slope(double a, double& f, double& s) { eval_line<true>(a, f, s); }

    // load the factor lane's 12 (point, bounds) from a free-variable-ordered vector
    __device__ void load_base(const double* vec) {
        if (has_fac) {
            const int c = P.cam[fid], q = P.pt[fid];
            const int* sl = A.slot_li + 12ll * gt;
#pragma unroll
            for (int k = 0; k < 12; ++k) {
                const int v = k < 9 ? c + k : q + (k - 9);
                const int li = sl[k];
                if (li >= 0) { base[k] = vec[li]; lov[k] = P.lo[v]; hiv[k] = P.hi[v]; }
                else { base[k] = P.x[v]; lov[k] = -__builtin_inf(); hiv[k] = __builtin_inf(); }
                dirv[k] = 0.0;
            }
        }
    }

    // ---- the CG vectors, one variable per owner ----------------------------------------
    __device__ void var_init(VarState& V, int li) {
        V.li = li;
        V.p = V.xi = V.g = V.h = V.xinit = 0.0; V.lo = V.hi = 0.0;
        if (li >= 0) {
            const int vid = L.free_vid[f0 + li];
            V.p = L.xstart[f0 + li]; V.xinit = V.p;
            V.lo = P.lo[vid]; V.hi = P.hi[vid];
        }
    }
    __device__ void init_vectors() {
        var_init(lv, A.lane_var[gt]);
        var_init(wv, A.wave_var[gt >> 6]);
        load_base(L.xstart + f0);
#ifdef RDIS_REFERENCE_SLOPE
#pragma unroll
        for (int k = 0; k < 12; ++k) prevv[k] = __builtin_nan("");
        fcache = 0.0; fdirty = true;   // (a factor's value is computed at its first evaluation)
#endif
    }

    __device__ void gradient_to_xi() {
        const long long tg0 = coop_clock();
        if (has_fac) {
            double v[12], g[12];
#pragma unroll
            for (int k = 0; k < 12; ++k) v[k] = clampd(base[k], lov[k], hiv[k]);
#ifdef RDIS_REFERENCE_SLOPE
            if (A.reference_slope) note_assign(v);   // SubfunctionFD::df assigns its point too (.cpp:135-157)
#endif
            ba_eval_grad(v, ox, oy, g);
            const int* sp = L.slot_pos + L.slot_base[c0 + gt];
            int t[12];  // all twelve destinations first: the loads overlap (the coherent stores
                        // below are atomics, across which the compiler does not move a load)
#pragma unroll
            for (int k = 0; k < 12; ++k) t[k] = sp[k];
#pragma unroll
            for (int k = 0; k < 12; ++k)
                if (t[k] >= 0) store_f64<true>(L.gfac + t[k], g[k]);
        }
        const long long tg1 = coop_clock();
        X.barrier(SYNC_DRAIN);   // the partials are written and read with coherent accesses
        const long long tg2 = coop_clock();
        const int* vp = L.v2s_ptr + f0;
        if (lv.li >= 0) {  // few partials: serial, in factor-list order (src/State.h:157-210)
            lv.xi = run_sum_ordered<true>(L.gfac, vp[lv.li], vp[lv.li + 1]);
        }
#ifdef RDIS_REFERENCE_SLOPE
        if (wv.li >= 0 && A.reference_slope) {  // (the parity option: in factor-list order too)
            wv.xi = wave_sum_in_order(L.gfac, vp[wv.li], vp[wv.li + 1], tid);
        } else
#endif
        if (wv.li >= 0) {  // many partials: the wave strides over the run, then a butterfly (fixed order)
            wv.xi = wave_sum(run_sum_strided<true>(L.gfac, vp[wv.li], vp[wv.li + 1], tid & 63));
        }
        tick(20, tg1 - tg0); tick(21, tg2 - tg1); tick(30, coop_clock() - tg2);
    }
    __device__ void publish_xi() {
        if (lv.li >= 0) store_f64<true>(A.xi_glob + lv.li, lv.xi);
        if (wv.li >= 0 && (tid & 63) == 0) store_f64<true>(A.xi_glob + wv.li, wv.xi);
        X.barrier(SYNC_DRAIN);   // line_begin, which follows in the same request, reads it
    }
    __device__ void cg_start() {
        { const double t = -lv.xi; lv.g = t; lv.h = t; lv.xi = t; }
        { const double t = -wv.xi; wv.g = t; wv.h = t; wv.xi = t; }
        publish_xi();
    }
    __device__ void line_begin() {
        if (has_fac) {
            const int* sl = A.slot_li + 12ll * gt;
            int li[12];
#pragma unroll
            for (int k = 0; k < 12; ++k) li[k] = sl[k];
            double d[12];   // unconditional, so that the twelve loads are in flight together
#pragma unroll
            for (int k = 0; k < 12; ++k) d[k] = load_f64<true>(A.xi_glob + (li[k] >= 0 ? li[k] : 0));
#pragma unroll
            for (int k = 0; k < 12; ++k) dirv[k] = li[k] >= 0 ? d[k] : 0.0;
        }
        if (L.vdump != nullptr && lm_count < L.dump_iters) {
            double* d = L.vdump + 2ll * L.dump_iters * f0 + 2ll * lm_count * n;
            if (lv.li >= 0) { d[lv.li] = lv.p; d[n + lv.li] = lv.xi; }
            if (wv.li >= 0 && (tid & 63) == 0) { d[wv.li] = wv.p; d[n + wv.li] = wv.xi; }
        }
        ++lm_count;
    }
    __device__ void line_end(double amin) {
#pragma clang fp contract(off)
        { const double t = lv.xi * amin; lv.xi = t; lv.p = lv.p + t; }
        { const double t = wv.xi * amin; wv.xi = t; wv.p = wv.p + t; }
        if (has_fac) {
#pragma unroll
            for (int k = 0; k < 12; ++k) { const double t = dirv[k] * amin; base[k] = base[k] + t; }
        }
    }
    __device__ void cg_reduce(double fp, double& test, double& gg, double& dgg) {
#pragma clang fp contract(off)
        const double den = fmax(fabs(fp), 1.0);
        double a = 0.0, b = 0.0, t = 0.0;
        if (lv.li >= 0) {
            t = fabs(lv.xi) * fmax(fabs(lv.p), 1.0) / den;
            a = lv.g * lv.g;
            b = (lv.xi + lv.g) * lv.xi;
        }
        if (wv.li >= 0 && (tid & 63) == 0) {
            t = fmax(t, fabs(wv.xi) * fmax(fabs(wv.p), 1.0) / den);
            a = a + wv.g * wv.g;
            b = b + (wv.xi + wv.g) * wv.xi;
        }
#ifdef RDIS_REFERENCE_SLOPE
        if (A.reference_slope) {   // gg and dgg over the variables in list order (minimize_nrc.h:665-672); the maximum has no order
            double* sa = L.seq_ab + f0;
            double* sb = sa + L.seq_n;
            if (lv.li >= 0) { store_f64<true>(sa + lv.li, lv.g * lv.g); store_f64<true>(sb + lv.li, (lv.xi + lv.g) * lv.xi); }
            if (wv.li >= 0 && (tid & 63) == 0) { store_f64<true>(sa + wv.li, wv.g * wv.g); store_f64<true>(sb + wv.li, (wv.xi + wv.g) * wv.xi); }
            X.barrier(SYNC_DRAIN);
            double ga = 0.0, gb = 0.0;
            if (gt < 64) { ga = wave_sum_in_order(sa, 0, n, tid); gb = wave_sum_in_order(sb, 0, n, tid); }
            a = gt == 0 ? ga : 0.0;
            b = gt == 0 ? gb : 0.0;
        }
#endif
        X.to_wave0<3>(a, b, t, SYNC_NONE);
        X.finish_wave0(SYNC_NONE);
        gg = a; dgg = b; test = t;
    }
    __device__ void cg_update(double gam) {
        {
#pragma clang fp contract(off)
            { const double gn = -lv.xi; const double hn = gn + gam * lv.h; lv.g = gn; lv.h = hn; lv.xi = hn; }
            { const double gn = -wv.xi; const double hn = gn + gam * wv.h; wv.g = gn; wv.h = hn; wv.xi = hn; }
        }
        publish_xi();
    }
    __device__ void write_back(const VarState& V, bool restore, bool writer) {
        if (V.li >= 0 && writer) {
            const double xf = clampd(restore ? V.xinit : V.p, V.lo, V.hi);
            P.x[L.free_vid[f0 + V.li]] = xf;   // variables are left assigned (.cpp:61, :84-86)
            L.xout[f0 + V.li] = xf;
        }
    }
};

// One launch solves several components side by side: workgroups [wg0, wg0 + nwg) of the grid form
// the group of component a.comp, with its own exchange state; groups never talk to each other.
constexpr int COOP_MAX_GROUPS = 256;
struct CoopGroup {
    CoopArgs a;
    int wg0, nwg;
};

// body shared by the two entry points below
template <int THREADS>
__device__ __forceinline__ void coop_solve(const ProblemView& P, const PlanView& L, const CoopArgs& A, int nwg, int wg,
                                           int maxiters, double ftol) {
    __shared__ double bcast[8];
    const long long tk0 = coop_clock();
    const int comp = A.comp;
    const int f0 = L.free_ptr[comp], c0 = L.fac_ptr[comp];
    const int n = L.free_ptr[comp + 1] - f0, m = L.fac_ptr[comp + 1] - c0;
    const int gt = wg * blockDim.x + threadIdx.x;

    CoopEnv E{P, L, A, n, m, f0, c0, gt, (int)threadIdx.x,
              GridSync{A.st, (int)threadIdx.x, nwg, wg, bcast, A.poll_delay, 0, 0u, false, 0u, {}},
              L.trace ? L.trace + 4ll * L.trace_cap * comp : nullptr, 0, 0,
              gt < m, 0, {}, {}, {}, {}, 0.0, 0.0, {}, {}};
    if (E.has_fac) {
        E.fid = L.fac_id[c0 + gt];
        const double2 o = P.obs[E.fid];
        E.ox = o.x; E.oy = o.y;
    }

    __shared__ CgdMachine M;
    __shared__ Request Q[2];
    E.init_vectors();
    run_machine(E, M, Q, maxiters, ftol);
    const int status = M.status();   // a failed exchange is recorded in M by the stepping wave
    const bool restore = M.rolled_back;
    E.write_back(E.lv, restore, true);
    E.write_back(E.wv, restore, (threadIdx.x & 63) == 0);
    if (gt == 0) {
        L.fret[comp] = M.fret; L.delta[comp] = M.fret - M.finit; L.iters[comp] = M.iter;
        L.status[comp] = status; L.nfeval[comp] = M.nfeval; L.ngeval[comp] = M.ngeval;
        if (L.trace_n) L.trace_n[comp] = E.trn;
        E.X.tm[7] = coop_clock() - tk0;
        if (A.timing) for (int i = 0; i < COOP_TM; ++i) A.timing[i] = E.X.tm[i];
    }
}

// several groups side by side: every workgroup looks up its group in the launch's table
template <int THREADS>
__global__ void __launch_bounds__(THREADS)
cgd_coop_kernel(ProblemView P, PlanView L, const CoopGroup* __restrict__ groups, const int* __restrict__ wg_group,
                int maxiters, double ftol) {
    const CoopGroup G = groups[wg_group[blockIdx.x]];
    coop_solve<THREADS>(P, L, G.a, G.nwg, (int)blockIdx.x - G.wg0, maxiters, ftol);
}
// one group: its arguments come as kernel arguments (the headline case; 2 % faster than through the table)
template <int THREADS>
__global__ void __launch_bounds__(THREADS)
cgd_coop_single_kernel(ProblemView P, PlanView L, CoopArgs A, int maxiters, double ftol) {
    coop_solve<THREADS>(P, L, A, (int)gridDim.x, (int)blockIdx.x, maxiters, ftol);
}

// arms the granules the groups of a launch will use (one launch instead of two memsets per group)
__global__ void __launch_bounds__(256) coop_arm_kernel(const CoopGroup* __restrict__ groups, int waves_per_wg) {
    const CoopGroup G = groups[blockIdx.x];
    CoopState* st = G.a.st;
    const int entries = G.nwg * waves_per_wg;
    for (int t = threadIdx.x; t < COOP_NBUF * COOP_KP * entries; t += blockDim.x) {
        const int k = t % COOP_KP, e = (t / COOP_KP) % entries, b = t / (COOP_KP * entries);
        st->granule[b][e][k] = ~0ull;
    }
    if (threadIdx.x == 0) st->abort_flag = 0u;
}

// host side: returns hipSuccess (0) or a hipError_t.  groups / wg_group: device arrays (ngroups
// entries / one entry per workgroup of the launch); first: host copy of groups[0]
inline int launch_coop(hipStream_t stream, int kind, const ProblemView& P, const PlanView& V, const CoopGroup& first,
                       const CoopGroup* groups, const int* wg_group, int ngroups, int total_wg, int threads, int maxiters, double ftol) {
    if (kind != KIND_BA) return (int)hipErrorNotSupported;
    coop_arm_kernel<<<ngroups, 256, 0, stream>>>(groups, threads / 64);
    hipError_t e0 = hipGetLastError();
    if (e0 != hipSuccess) return (int)e0;
    ProblemView p = P;
    PlanView v = V;
    int mi = maxiters;
    double ft = ftol;
    if (ngroups == 1) {
        CoopArgs a = first.a;
        void* args[] = {&p, &v, &a, &mi, &ft};
        const void* fn = threads == 512 ? (const void*)cgd_coop_single_kernel<512>
                       : threads == 128 ? (const void*)cgd_coop_single_kernel<128>
                                        : (const void*)cgd_coop_single_kernel<256>;
        return (int)hipLaunchCooperativeKernel(fn, dim3(total_wg), dim3(threads), args, 0, stream);
    }
    const CoopGroup* gp = groups;
    const int* wp = wg_group;
    void* args[] = {&p, &v, &gp, &wp, &mi, &ft};
    const void* fn = threads == 512 ? (const void*)cgd_coop_kernel<512>
                   : threads == 128 ? (const void*)cgd_coop_kernel<128>
                                    : (const void*)cgd_coop_kernel<256>;
    return (int)hipLaunchCooperativeKernel(fn, dim3(total_wg), dim3(threads), args, 0, stream);
}

inline int coop_max_workgroups(int threads, int num_cus) {
    int per_cu = 0;
    const void* fn = threads == 512 ? (const void*)cgd_coop_kernel<512>
                   : threads == 128 ? (const void*)cgd_coop_kernel<128>
                                    : (const void*)cgd_coop_kernel<256>;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, threads, 0) != hipSuccess) return 0;
    // stay one block per CU below what the API reports when it reports more than one
    // (MI355X_MICROARCH.md: the query can be one high for SGPR-heavy kernels)
    if (per_cu > 1) per_cu -= 1;
    const long long cap = (long long)per_cu * num_cus;
    return (int)(cap > COOP_MAX_WG ? COOP_MAX_WG : cap);
}

}  // namespace rdis_hip
