// solver_lds.hpp -- one workgroup solves one bundle-adjustment component whose variables fit
// the compute unit's LDS: the whole CGDSubspaceOptimizer::optimize call (reference
// src/optimizers/CGDSubspaceOptimizer.cpp:19-98) with the CG iterate resident in LDS, a batch of
// independent components per launch (grid = components, heaviest first).
//
// solver_wg.hpp forms every trial point in global memory: phase A writes x[vid] = clamp(p + a*xi)
// for the component's free variables, a barrier, phase B gathers 12 values of x and 12 of the
// direction per factor through L2 -- per trial point and component that is a round trip of p, xi,
// lo, hi, x and dir (measured on the 256 x ladybug-size batch: 2.8 x the algorithmic bytes in HBM
// traffic, and with every compute unit busy a component ran three times slower than alone).
// Here a component's variables get SLOTS in LDS, camera blocks first (9 slots each), then point
// blocks (3 each); a slot holds a free variable or a constant the component's factors read:
//
//   Pv[s]  the CG iterate p (unclamped; a constant's assigned value)      LO[s], HI[s]  the domain
//   XI[s]  the search direction / gradient (0 for a constant)            X[s]   clamp(p + a*xi)
//   ROT[c] rotation record of camera block c (factors.hpp) when the launch uses records
//
// Per line-search trial (SubfunctionFD::operator() / df, .cpp:124-184):
//   phase A  every free slot: X = clamp(Pv + a XI)                     LDS -> LDS, no global access
//   phase B  every factor: 12 values of X and XI from two slot bases (one packed 32-bit word per
//            listed factor), its observation; value and forward-mode slope as in solver_wg.hpp
//   reduce   wave (DPP) + LDS, fixed order: the same bits run to run, and -- the factor arithmetic
//            and the order of the sums being those of solver_wg.hpp -- the same bits as that solver
//            whenever the slots are the free variables in their listed order (every block free,
//            cameras before points, ascending: what RDIS and the generators produce) and no camera
//            variable is free (a camera's gradient entries are grouped differently, below); tested.
// The full gradient, once per CG iteration: camera partials are summed across waves whose factors
// share a camera, point partials go through the variable-major gfac[] like solver_wg.hpp's (per-variable
// sums in factor-list order, src/State.h:157-210) -- see gradient_to_xi; g and h of the Polak-Ribiere
// recurrence stay in the plan's workspace (touched twice per iteration).
#pragma once
#include "solver_wg.hpp"

namespace rdis_hip {

constexpr int LDS_MAX_BYTES = 160 * 1024 - 4096;   // dynamic LDS a launch may ask for (static: machine, requests, reduction slots)
constexpr int LDS_DOUBLES_PER_SLOT = 5;            // Pv, XI, LO, HI, X
// Trials in matrix form (round 5; PlanView::ls_matrix): per camera block two records of factors.hpp's CAM_TRIAL = 16 doubles at a
// stride of 18 (an odd number of 16-byte units: lanes that read different cameras stand on different banks), behind the other arrays
constexpr int LDS_TS = 18;
__host__ __device__ inline size_t lds_matrix_offset(size_t base_bytes) { return (base_bytes + 15) & ~(size_t)15; }
__host__ __device__ inline size_t lds_matrix_bytes(int ncb) { return (size_t)ncb * 2 * LDS_TS * sizeof(double); }
__host__ __device__ inline size_t lds_bytes_for(int ns, int ncb, int nchunk) {
    // (per slot two ints: the local free index, and the assignment at which the slot last changed by 1e-12 or more --
    // the latter used by the stale-cache emulation only)
    return (size_t)ns * (LDS_DOUBLES_PER_SLOT * sizeof(double) + 2 * sizeof(int)) + (size_t)ncb * 7 * sizeof(double) +
           (size_t)nchunk * (9 * sizeof(double) + sizeof(int)) + 64;
}

// STALE: emulate the reference's factor cache (plan option emulate_stale_cache).  Variable::assign tells a variable's
// factors to recompute only when the new value differs from the variable's previous one by 1e-12 or more
// (src/Variable.cpp:66-76); Factor::eval returns the cached value otherwise (src/Factor.h:228-234).  On the plateaus
// of these descents the last trial points of a line search lie closer together than that, and the reference's sums
// then mix values of neighbouring points -- 1e-12-level differences that decide how often the 3e-8 stopping test fires
// (of 1000 synthetic components 859 leave by a tolerance with the cache, 939 without).  Emulated with two counters:
// every assignment of a trial point has a number (epoch); a slot remembers the last assignment that moved it by
// >= 1e-12 (CHE), a factor the assignment of its last VALUE evaluation (fev) and that value (fvv); the factor is
// recomputed iff one of its twelve slots moved since.  Slopes and gradients are always fresh, like the reference's
// computeGradient.  The control logic then evaluates every point the reference evaluates (CgdMachine::noskip).
template <int ROT, bool PREFETCH, bool STALE = false>
struct LdsEnv {
    static constexpr bool NOSKIP = STALE;
    const ProblemView& P;
    const PlanView& L;
    int comp, n, m, f0, c0, tid, nt, nwaves;
    int ns, ncb;              // slots, camera blocks of this component
    const double2* fobs;      // its listed factors' observations (plan-local copy in listed order)
    const unsigned* fidx;     // its listed factors' slot word: camera block | point block << 12
    const int* gperm;         // its listed factors (local index) grouped by camera block, groups padded to whole waves with -1
    int nchunk;               // ... in wave-chunks of 64
    double* CG;               // LDS [nchunk][9]: a chunk's camera partial sums
    int* CGC;                 // LDS [nchunk]: its camera block
    const int* vptr;          // v2s_ptr + free offset
    const int* svid;          // variable id of a slot
    double *Pv, *XI, *LO, *HI, *X, *ROTR;   // LDS
    double *CTR, *CDR;        // LDS [ncb][LDS_TS] each (ls_matrix): the cameras' trial records at the trial point at hand, and their derivative along the direction
    int* SF;                  // LDS: local free index of a slot, -1 = constant
    int* CHE;                 // LDS (STALE): the assignment that last moved the slot by 1e-12 or more
    int* fev;                 // (STALE) per listed factor: the assignment of its last value evaluation, -1 = never
    double* fvv;              //         ... and the value
    int epoch;                //         assignments so far
    double *g, *h;            // plan workspace, by free index
    double (*red)[3][MAX_WAVES];
    int parity;
    double* tr;
    int trn, lm_count;

    template <int K>
    __device__ void sumk(double& a, double& b, double& mx) {
        a = wave_sum(a);
        if constexpr (K >= 2) b = wave_sum(b);
        if constexpr (K >= 3) mx = wave_max(mx);
        if (nwaves > 1) {
            const int w = tid >> 6;
            if ((tid & 63) == 0) {
                red[parity][0][w] = a;
                if constexpr (K >= 2) red[parity][1][w] = b;
                if constexpr (K >= 3) red[parity][2][w] = mx;
            }
            __syncthreads();
            combine_waves<K>(red[parity], nwaves, a, b, mx);
            parity ^= 1;
        }
    }
    __device__ void trace(int tag, double a, double b, double c) {
        if (tr != nullptr && tid == 0) {
            if (trn < L.trace_cap) { double* r = tr + 4ll * trn; r[0] = (double)tag; r[1] = a; r[2] = b; r[3] = c; }
            ++trn;
        }
    }

    // rotation records of the camera blocks with a free rotation variable (ROT_RECORDS), formed
    // from the trial point the way phase A forms it -- no need to wait for X; taken by the last
    // lanes of the workgroup (the first wave steps the control logic)
    template <class At>
    __device__ void refresh_records(At at) {
        if constexpr (ROT == ROT_RECORDS) {
            for (int c = nt - 1 - tid; c < ncb; c += nt) {
                const int s = 9 * c;
                if (SF[s] < 0 && SF[s + 1] < 0 && SF[s + 2] < 0) continue;   // a constant camera: its record stays
                double r[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) r[k] = SF[s + k] >= 0 ? at(s + k) : Pv[s + k];
                store_rotation(r[0], r[1], r[2], ROTR + 7 * c);
            }
        }
    }
    // SubfunctionFD::quickAssignVals at p + a*xi (reference .cpp:160-184; the trial point is
    // formed unfused like minimize_nrc.h:434)
    // Variable::assign of an assigned variable (src/Variable.cpp:66-88)
    __device__ __forceinline__ void set_x(int s, double xn) {
        if constexpr (STALE) {
            if (!(fabs(xn - X[s]) < 1e-12)) CHE[s] = epoch;
        }
        X[s] = xn;
    }
    // ... and in matrix form (ls_matrix): a camera block with a free variable gets its trial records at the trial point -- rotation
    // matrix, translation, f, k1, k2, and with SLOPE their derivative along the direction (factors.hpp) --, one lane per camera,
    // the workgroup's last lanes; a constant camera keeps what init_vectors gave it
    template <bool SLOPE, class At>
    __device__ void refresh_trial_records(At at) {
        if constexpr (ROT == ROT_CAMFIX) return;
        for (int c = nt - 1 - tid; c < ncb; c += nt) {
            const int s0 = 9 * c;
            bool any = false;
#pragma unroll
            for (int q = 0; q < 9; ++q) any = any || SF[s0 + q] >= 0;
            if (!any) continue;
            double xc[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) xc[k] = SF[s0 + k] >= 0 ? at(s0 + k) : Pv[s0 + k];
            BaFwd rot;
            ba_rotation(xc[0], xc[1], xc[2], rot);
            ba_camera_trial(rot, xc, CTR + LDS_TS * c);
            if constexpr (SLOPE) {
                double dc[9];
#pragma unroll
                for (int k = 0; k < 9; ++k) dc[k] = XI[s0 + k];   // (a constant's entry is zero)
                ba_camera_trial_dir(rot, dc, CDR + LDS_TS * c);
            }
        }
    }
    // RECS: 0 = rotation records where the launch uses them (the vector form), 1 / 2 = trial records for a value / a value + slope trial
    template <int RECS = 0>
    __device__ void assign_line(double a) {
#pragma clang fp contract(off)
        if constexpr (STALE) ++epoch;
        auto at = [&](int s) {
#pragma clang fp contract(off)
            const double t = a * XI[s];
            return clampd(Pv[s] + t, LO[s], HI[s]);
        };
        if constexpr (RECS == 0) {
            for (int s = tid; s < ns; s += nt) {
                if (SF[s] < 0) continue;
                const double t = a * XI[s];
                set_x(s, clampd(Pv[s] + t, LO[s], HI[s]));
            }
            refresh_records(at);
        } else {   // (the cameras' own slots of X are not read by a trial in matrix form: point slots only)
            for (int s = 9 * ncb + tid; s < ns; s += nt) {
                if (SF[s] < 0) continue;
                const double t = a * XI[s];
                X[s] = clampd(Pv[s] + t, LO[s], HI[s]);
            }
            refresh_trial_records<RECS == 2>(at);
        }
        __syncthreads();
    }
    __device__ void assign_p() {
        if constexpr (STALE) ++epoch;
        for (int s = tid; s < ns; s += nt)
            if (SF[s] >= 0) set_x(s, clampd(Pv[s], LO[s], HI[s]));
        refresh_records([&](int s) { return clampd(Pv[s], LO[s], HI[s]); });
        __syncthreads();
    }
    __device__ void assign_start() {   // clamp(x_init): the rollback (CGD .cpp:71)
        const double* xs = L.xstart + f0;
        if constexpr (STALE) ++epoch;
        for (int s = tid; s < ns; s += nt)
            if (SF[s] >= 0) set_x(s, clampd(xs[SF[s]], LO[s], HI[s]));
        refresh_records([&](int s) { return clampd(xs[SF[s]], LO[s], HI[s]); });
        __syncthreads();
    }

    // the forward state of a listed factor (slot word w, observation o) at X: v = its 12 inputs (v[0..2] not loaded with records)
    __device__ __forceinline__ double forward(unsigned w, double2 o, double (&v)[12], BaFwd& t, int& cb, int& pb) {
        const int c = (int)(w & 0xFFFu);
        cb = 9 * c;
        pb = 9 * ncb + 3 * (int)(w >> 12);
#pragma unroll
        for (int k = 3; k < 9; ++k) v[k] = X[cb + k];
#pragma unroll
        for (int k = 0; k < 3; ++k) v[9 + k] = X[pb + k];
        if constexpr (ROT == ROT_PER_FACTOR) {
#pragma unroll
            for (int k = 0; k < 3; ++k) v[k] = X[cb + k];
            return ba_forward(v, o.x, o.y, t);
        } else {
            v[0] = v[1] = v[2] = 0.0;
            ba_load_rotation(ROTR + 7 * c, t);
            return ba_project(v, o.x, o.y, t);
        }
    }

    // This lane's share of the sums.  A factor's slot word and observation (20 bytes, plan-local arrays in
    // listed order: coalesced, no indirection) are fetched one round ahead: with two or three waves per
    // SIMD nothing else hides the ~600 cycles of an L2 round trip in front of ~1000 cycles of arithmetic
    // (counters before: 60 % of all wave-cycles waiting, VALU busy 39 %).  Not at three waves per SIMD: the five
    // registers it holds are spilled there (768 lanes: 3.76 against 3.47 ms on 125 x 2048 factors).
    template <bool SLOPE>
    __device__ __forceinline__ void eval_partial(double& af, double& as) {
        int j = tid;
        unsigned wn = 0u;
        double2 on = make_double2(0.0, 0.0);
        if (PREFETCH && j < m) { wn = fidx[j]; on = fobs[j]; }
        // (a workgroup with several passes over its lanes: issue priority from the count of passes done, as in solver_ptm.hpp's trial
        // loop -- the SIMD's arbiter does not serve its waves alike, and the sums wait for the last; ladybug's 49 camera components
        // 23.7 -> 22.5 ms.  Not for one or two passes: with many small workgroups on a compute unit it only costs, +1 .. 4 %)
        const bool prio = m > 2 * nt;
        int nit = 0;
        for (; j < m; j += nt) {
            if (prio) {
                const int lv = (0 - nit) & 3;
                if (lv == 3) __builtin_amdgcn_s_setprio(3); else if (lv == 2) __builtin_amdgcn_s_setprio(2);
                else if (lv == 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
                ++nit;
            }
            unsigned w;
            double2 o;
            if constexpr (PREFETCH) {
                w = wn; o = on;
                if (j + nt < m) { wn = fidx[j + nt]; on = fobs[j + nt]; }
            } else {
                w = fidx[j]; o = fobs[j];
            }
            double v[12];
            BaFwd t;
            int cb, pb;
            double fj = forward(w, o, v, t, cb, pb);
            if constexpr (STALE) {   // Factor::eval -> evalFactorCached
                int moved = CHE[pb];
#pragma unroll
                for (int k = 1; k < 3; ++k) moved = max(moved, CHE[pb + k]);
#pragma unroll
                for (int k = 0; k < 9; ++k) moved = max(moved, CHE[cb + k]);
                const int ev = fev[j];
                if (ev < 0 || moved > ev) fvv[j] = fj; else fj = fvv[j];
                fev[j] = epoch;
            }
#ifdef RDIS_REFERENCE_SLOPE
            L.seq_val[c0 + j] = fj;   // (the parity option adds the listed factors' values in list order: sum_in_order)
#endif
            af += fj;
            if constexpr (SLOPE) {
                double d[12];
#pragma unroll
                for (int k = 0; k < 9; ++k) d[k] = (ROT == ROT_CAMFIX) ? 0.0 : XI[cb + k];
#pragma unroll
                for (int k = 0; k < 3; ++k) d[9 + k] = XI[pb + k];
#ifdef RDIS_BISECT_ADJOINT_SLOPE   // (measurement build: the slope as the factor's twelve partials times the direction, like Df1dim::df)
                double gq[12], acc = 0.0;
                ba_adjoint(t, v, t.res0, t.res1, gq);
#pragma unroll
                for (int k = (ROT == ROT_CAMFIX ? 9 : 0); k < 12; ++k) acc += gq[k] * d[k];
                as += acc;
#else
                as += ba_slope_dir<ROT == ROT_CAMFIX>(t, v, d);
#endif
            }
        }
        if (prio) __builtin_amdgcn_s_setprio(0);
    }
    // ... in matrix form: the factor against its camera's trial records (88 fp64 operations for value and slope where the vector form
    // takes 164; same model, another association of the same sums: values agree to 1e-13, slopes to 1e-11 of their terms,
    // tests/cpp/factors_forms_test.hip; the replay against the oracle holds as for the vector form)
    template <bool SLOPE>
    __device__ __forceinline__ void eval_partial_matrix(double& af, double& as) {
        int j = tid;
        unsigned wn = 0u;
        double2 on = make_double2(0.0, 0.0);
        if (PREFETCH && j < m) { wn = fidx[j]; on = fobs[j]; }
        for (; j < m; j += nt) {
            unsigned w;
            double2 o;
            if constexpr (PREFETCH) {
                w = wn; o = on;
                if (j + nt < m) { wn = fidx[j + nt]; on = fobs[j + nt]; }
            } else {
                w = fidx[j]; o = fobs[j];
            }
            const int off = __mul24((int)(w & 0xFFFu), LDS_TS), pb = 9 * ncb + 3 * (int)(w >> 12);
            double TR[CAM_TRIAL], DR[CAM_DIR];
            const double2* tc = reinterpret_cast<const double2*>(CTR + off);
#pragma unroll
            for (int k = 0; k < CAM_TRIAL / 2; ++k) { const double2 v = tc[k]; TR[2 * k] = v.x; TR[2 * k + 1] = v.y; }
            const double x[3] = {X[pb], X[pb + 1], X[pb + 2]};
            BaTrial t;
            af += ba_trial_value(TR, x, o.x, o.y, t);
            if constexpr (SLOPE) {
                const double e[3] = {XI[pb], XI[pb + 1], XI[pb + 2]};
                if constexpr (ROT != ROT_CAMFIX) {
                    const double2* dc = reinterpret_cast<const double2*>(CDR + off);
#pragma unroll
                    for (int k = 0; k < CAM_DIR / 2; ++k) { const double2 v = dc[k]; DR[2 * k] = v.x; DR[2 * k + 1] = v.y; }
                } else {
#pragma unroll
                    for (int k = 0; k < CAM_DIR; ++k) DR[k] = 0.0;
                }
                as += ba_trial_slope<ROT == ROT_CAMFIX>(t, TR, DR, x, e);
            }
        }
    }
    template <bool SLOPE>
    __device__ void eval_sum(double& f, double& s) {
        double af = 0.0, as = 0.0, dummy = 0.0;
        eval_partial<SLOPE>(af, as);
        sumk<SLOPE ? 2 : 1>(af, as, dummy);
        f = af; s = as;
    }

    static constexpr bool UNIFORM = true;
    static constexpr int SPEC = 1;
    __device__ bool stepper() const { return threadIdx.x < 64; }
    __device__ bool writer() const { return (threadIdx.x & 63) == 0; }
    __device__ void sync() const { __syncthreads(); }
    __device__ bool tracing() const { return tr != nullptr; }
    __device__ bool aborted() const { return false; }
    // cycle stamps of workgroup 0's first lane (build with -DRDIS_COOP_TIMING; rdis_hip_plan_debug_counters):
    // 0 phase A, 1 phase B, 2 reduction of a value+slope trial, 3 their number; 8 / 9 control step / hand-over,
    // 12.. cycles per request kind (12 value, 13 value+slope, 14 gradient + reduction, 17 line end), 22.. their counts
#ifdef RDIS_COOP_TIMING
    long long tm[32];
    __device__ void tick(int slot, long long dt) { tm[slot] += dt; }
    __device__ long long clock() const { return clock64(); }
#else
    __device__ void tick(int, long long) {}
    __device__ long long clock() const { return 0; }
#endif
#ifdef RDIS_REFERENCE_SLOPE
    __device__ bool matrix() const { return false; }
    // The parity option (this instantiation: plan option factor_rounding = 1) adds every sum in the reference's order.
    // arr[0 .. cnt) added one after the other from 0.0 by the workgroup's first lane (sixteen loads in flight, the additions one
    // dependent chain: the plain loop's bits); every lane returns the sum.
    __device__ double sum_in_order(const double* arr, int cnt) {
#pragma clang fp contract(off)
        __syncthreads();   // (the entries are written)
        if (tid == 0) {
            double acc = 0.0;
            for (int k0 = 0; k0 < cnt; k0 += 16) {
                double t[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) t[j] = (k0 + j < cnt) ? arr[k0 + j] : 0.0;
#pragma unroll
                for (int j = 0; j < 16; ++j)
                    if (k0 + j < cnt) acc = acc + t[j];
            }
            red[parity][0][0] = acc;
        }
        __syncthreads();
        const double r = red[parity][0][0];
        parity ^= 1;
        return r;
    }
#else
    __device__ bool matrix() const { return !STALE && L.ls_matrix != 0; }
#endif
    __device__ double eval_value(double a, bool restore) {
        if (!restore && matrix()) {
            assign_line<1>(a);
            double af = 0.0, as = 0.0, dummy = 0.0;
            eval_partial_matrix<false>(af, as);
            sumk<1>(af, as, dummy);
            return af;
        }
#ifdef RDIS_REFERENCE_SLOPE
        if (restore) {
            if constexpr (STALE) assign_p();   // the end point is assigned before the rollback is considered (CGD .cpp:61)
            assign_start();
        } else {
            assign_line(a);
        }
        double af = 0.0, as = 0.0;
        eval_partial<false>(af, as);
        return sum_in_order(L.seq_val + c0, m);   // OptimizableFunction::evalFactors: in list order (.cpp:95-135)
#else
        if (restore) assign_start(); else assign_line(a);
        double f, s;
        eval_sum<false>(f, s);
        return f;
#endif
    }
#ifdef RDIS_REFERENCE_SLOPE
    // The slope of a trial the way the reference forms it (Df1dim::df, minimize_nrc.h:439-447, over SubfunctionFD::df,
    // CGDSubspaceOptimizer.cpp:135-157): the GRADIENT at the trial point -- every variable's partials added in factor-list order
    // (src/State.h:157-210) -- and then gradient times direction, variable by variable in list order, every product rounded
    // before it is added.  The solvers' own form adds factor by factor (sum_f sum_k partial_fk xi_k: no scatter, no gradient
    // per trial).  The two differ in the last place only -- and that decides a population: on ladybug 5 / 30 Dbrent takes secant
    // steps between trial points 1e-17 apart, where the difference of two slopes cancels ten digits, and the association of the
    // sum moves every such step the same way (found with the oracle's switch ro_set_experiment(2)).
    // Compiled into the reference-rounding instantiation only (refround_kernels.hip); costs a full gradient per trial.
    __device__ double slope_reference() {
#pragma clang fp contract(off)
        for (int j = tid; j < m; j += nt) {
            double v[12], gq[12];
            BaFwd t;
            int cb, pb;
            forward(fidx[j], fobs[j], v, t, cb, pb);
            ba_adjoint(t, v, t.res0, t.res1, gq);
            const int* sp = L.slot_pos + L.slot_base[c0 + j];
#pragma unroll
            for (int k = 0; k < 12; ++k) { const int u = sp[k]; if (u >= 0) L.gfac[u] = gq[k]; }
        }
        __syncthreads();
        double* GT = L.ws + 5ll * f0;   // [n] by free index (the workspace's first vector: this solver keeps p in LDS)
        for (int s = tid; s < ns; s += nt) {
            const int fi = SF[s];
            if (fi < 0) continue;
            const int b = vptr[fi], e = vptr[fi + 1];
            const double gv = b < e ? run_sum_ordered(L.gfac, b, e) : 0.0;
            GT[fi] = gv * XI[s];
        }
        return sum_in_order(GT, n);   // by free index: the order of `vars` (Df1dim::df, minimize_nrc.h:443-445)
    }
    // SubfunctionFD::df at clamp(p) the same way: every partial through gfac[], every variable's -- a camera's too -- added in
    // factor-list order by one lane (src/State.h:157-210)
    __device__ void gradient_ordered() {
        assign_p();
        for (int j = tid; j < m; j += nt) {
            double v[12], gq[12];
            BaFwd t;
            int cb, pb;
            forward(fidx[j], fobs[j], v, t, cb, pb);
            ba_adjoint(t, v, t.res0, t.res1, gq);
            const int* sp = L.slot_pos + L.slot_base[c0 + j];
#pragma unroll
            for (int k = 0; k < 12; ++k) { const int u = sp[k]; if (u >= 0) L.gfac[u] = gq[k]; }
        }
        __syncthreads();
        for (int s = tid; s < ns; s += nt) {
            const int fi = SF[s];
            if (fi < 0) continue;
            const int b = vptr[fi], e = vptr[fi + 1];
            XI[s] = b < e ? run_sum_ordered(L.gfac, b, e) : 0.0;
        }
        __syncthreads();
    }
#endif
    __device__ void eval_value_slope(double a, double& f, double& s) {
        const long long t0 = clock();
        double af = 0.0, as = 0.0, dummy = 0.0;
#ifndef RDIS_REFERENCE_SLOPE
        if (matrix()) {
            assign_line<2>(a);
            const long long t1m = clock();
            eval_partial_matrix<true>(af, as);
            const long long t2m = clock();
            sumk<2>(af, as, dummy);
            f = af; s = as;
            tick(0, t1m - t0); tick(1, t2m - t1m); tick(2, clock() - t2m); tick(3, 1);
            return;
        }
#endif
        assign_line(a);
        const long long t1 = clock();
#ifdef RDIS_REFERENCE_SLOPE
        {   // (with the stale-cache emulation too: a factor's VALUE may be the cached one, gradients are always fresh)
            eval_partial<false>(af, as);
            f = sum_in_order(L.seq_val + c0, m);
            s = slope_reference();
            (void)dummy; (void)t1;
            return;
        }
#endif
        eval_partial<true>(af, as);
        const long long t2 = clock();
        sumk<2>(af, as, dummy);
        f = af; s = as;
        tick(0, t1 - t0); tick(1, t2 - t1); tick(2, clock() - t2); tick(3, 1);
    }

    __device__ void init_vectors() {   // CGD .cpp:34-39: p = x0 (unclamped); the slots of constants hold their assigned value
        const double* xs = L.xstart + f0;
        for (int s = tid; s < ns; s += nt) {
            const int fi = SF[s], v = svid[s];
            if (fi >= 0) {
                const double lo = P.lo[v], hi = P.hi[v], x0 = xs[fi];
                Pv[s] = x0; LO[s] = lo; HI[s] = hi; X[s] = clampd(x0, lo, hi);
            } else {
                const double xc = P.x[v];
                Pv[s] = xc; X[s] = xc; LO[s] = -__builtin_inf(); HI[s] = __builtin_inf();
            }
            XI[s] = 0.0;
            if constexpr (STALE) CHE[s] = 0;
        }
        if constexpr (STALE) {
            for (int j = tid; j < m; j += nt) fev[j] = -1;
            epoch = 0;
        }
        __syncthreads();
        if constexpr (ROT != ROT_PER_FACTOR) {   // every camera block's record at the start (those of constant cameras stay)
            for (int c = tid; c < ncb; c += nt) store_rotation(X[9 * c], X[9 * c + 1], X[9 * c + 2], ROTR + 7 * c);
            __syncthreads();
        }
        if (matrix()) {   // ... and its trial records (a constant camera's direction record is zero)
            for (int c = tid; c < ncb; c += nt) {
                double xc[9];
#pragma unroll
                for (int k = 0; k < 9; ++k) xc[k] = X[9 * c + k];
                BaFwd rot;
                ba_rotation(xc[0], xc[1], xc[2], rot);
                ba_camera_trial(rot, xc, CTR + LDS_TS * c);
#pragma unroll
                for (int k = 0; k < CAM_TRIAL; ++k) CDR[LDS_TS * c + k] = 0.0;
            }
            __syncthreads();
        }
    }

    // SubfunctionFD::df(p, xi) (reference .cpp:135-157): full gradient at clamp(p).
    // The factors are taken camera by camera (gperm: the listed factors grouped by camera block, every group
    // padded to whole waves, -1 = no factor), so a wave's 64 factors share their camera:
    //   camera partials  summed across the wave (DPP), one 9-vector per wave-chunk in LDS, then per camera
    //                    variable the chunks in order -- no memory traffic at all (through the variable-major
    //                    gfac[] like solver_wg.hpp the scatter alone was 42 000 of a gradient's 70 000 cycles:
    //                    in listed order every one of a wave's 64 x 12 stores went to a cache line of its own,
    //                    and the per-camera sums over 256 partials another 16 000);
    //   point partials   to gfac[] (three per factor), then every point variable sums its run in factor-list
    //                    order (src/State.h:157-210) -- the same bits as solver_wg.hpp.
    // A camera variable's sum is therefore grouped differently from solver_wg.hpp's (which strides a wave over the
    // run): with free cameras the two solvers agree to rounding, not to the bit.
    __device__ void gradient_to_xi() {
#ifdef RDIS_REFERENCE_SLOPE
        gradient_ordered();
        return;
#endif
        if (L.ls_cam_gfac) { gradient_via_gfac(); return; }
        const long long tg0 = clock();
        assign_p();
        const int lane = tid & 63;
        for (int ch = tid >> 6; ch < nchunk; ch += nwaves) {
            const int j = gperm[64 * ch + lane];
            double gq[12];
#pragma unroll
            for (int k = 0; k < 12; ++k) gq[k] = 0.0;
            int u9 = -1, u10 = -1, u11 = -1;
            unsigned w = 0u;
            if (j >= 0) {
                double v[12];
                BaFwd t;
                int cb, pb;
                const int* sp = L.slot_pos + L.slot_base[c0 + j];
                u9 = sp[9]; u10 = sp[10]; u11 = sp[11];
                w = fidx[j];
                forward(w, fobs[j], v, t, cb, pb);
                ba_adjoint(t, v, t.res0, t.res1, gq);
                if (u9 >= 0) L.gfac[u9] = gq[9];
                if (u10 >= 0) L.gfac[u10] = gq[10];
                if (u11 >= 0) L.gfac[u11] = gq[11];
            }
            if constexpr (ROT != ROT_CAMFIX) {
                // (a chunk's first lane always holds a factor: the padding is at the end of a camera's group)
                const int c = __builtin_amdgcn_readfirstlane((int)(w & 0xFFFu));
                double cs[9];
#pragma unroll
                for (int k = 0; k < 9; ++k) cs[k] = wave_sum(gq[k]);
                if (lane < 9) CG[9 * ch + lane] = pick(cs, lane);
                if (lane == 0) CGC[ch] = c;
            }
        }
        __syncthreads();
        const long long tg1 = clock();
        if constexpr (ROT != ROT_CAMFIX) {
            for (int s = tid; s < 9 * ncb; s += nt) {
                if (SF[s] < 0) continue;
                const int c = s / 9, k = s - 9 * c;
                double sm = 0.0;
                bool first = true;
                for (int ch = 0; ch < nchunk; ++ch)
                    if (CGC[ch] == c) { sm = first ? CG[9 * ch + k] : sm + CG[9 * ch + k]; first = false; }
                XI[s] = sm;
            }
        }
        for (int s = 9 * ncb + tid; s < ns; s += nt) {
            const int fi = SF[s];
            if (fi < 0) continue;
            const int b = vptr[fi], e = vptr[fi + 1];
            XI[s] = b < e ? run_sum_ordered(L.gfac, b, e) : 0.0;
        }
        __syncthreads();
        tick(4, tg1 - tg0); tick(5, clock() - tg1); tick(10, 1);
    }

    // option lds_camera_sums = 0 (tests): every partial through gfac[] and every variable's sum formed as
    // solver_wg.hpp forms it -- short runs in factor-list order by a lane, long ones strided over a wave -- so
    // that the two solvers can be compared bit for bit with free cameras too
    __device__ void gradient_via_gfac() {
        assign_p();
        for (int j = tid; j < m; j += nt) {
            double v[12], gq[12];
            BaFwd t;
            int cb, pb;
            forward(fidx[j], fobs[j], v, t, cb, pb);
            ba_adjoint(t, v, t.res0, t.res1, gq);
            const int* sp = L.slot_pos + L.slot_base[c0 + j];
#pragma unroll
            for (int k = (ROT == ROT_CAMFIX ? 9 : 0); k < 12; ++k) { const int u = sp[k]; if (u >= 0) L.gfac[u] = gq[k]; }
        }
        __syncthreads();
        for (int s = tid; s < ns; s += nt) {
            const int fi = SF[s];
            if (fi < 0) continue;
            const int b = vptr[fi], e = vptr[fi + 1];
            if (e - b > WG_LONG_LIST) continue;
            XI[s] = b < e ? run_sum_ordered(L.gfac, b, e) : 0.0;
        }
        for (int s = tid >> 6; s < ns; s += nwaves) {   // (wave-uniform: the long runs, a wave each)
            const int fi = SF[s];
            if (fi < 0 || vptr[fi + 1] - vptr[fi] <= WG_LONG_LIST) continue;
            const double sm = wave_sum(run_sum_strided(L.gfac, vptr[fi], vptr[fi + 1], tid & 63));
            if ((tid & 63) == 0) XI[s] = sm;
        }
        __syncthreads();
    }

    __device__ void cg_start() {
        for (int s = tid; s < ns; s += nt) {
            const int fi = SF[s];
            if (fi >= 0) { const double t = -XI[s]; g[fi] = t; h[fi] = t; XI[s] = t; }
        }
        __syncthreads();
    }
    __device__ void line_begin() {
        if (L.vdump != nullptr && lm_count < L.dump_iters) {
            double* d = L.vdump + 2ll * L.dump_iters * f0 + 2ll * lm_count * n;
            for (int s = tid; s < ns; s += nt) {
                const int fi = SF[s];
                if (fi >= 0) { d[fi] = Pv[s]; d[n + fi] = XI[s]; }
            }
        }
        ++lm_count;
    }
    __device__ void line_end(double amin) {
#pragma clang fp contract(off)
        for (int s = tid; s < ns; s += nt) {
            if (SF[s] < 0) continue;
            const double t = XI[s] * amin;
            XI[s] = t;
            Pv[s] = Pv[s] + t;
        }
        __syncthreads();
    }
    __device__ void cg_reduce(double fp, double& test, double& gg, double& dgg) {
#pragma clang fp contract(off)
        const double den = fmax(fabs(fp), 1.0);
        double a = 0.0, b = 0.0, t = 0.0;
        for (int s = tid; s < ns; s += nt) {
            const int fi = SF[s];
            if (fi < 0) continue;
            const double x = XI[s], gi = g[fi];
            t = fmax(t, fabs(x) * fmax(fabs(Pv[s]), 1.0) / den);
            a = a + gi * gi;
            b = b + (x + gi) * x;
#ifdef RDIS_REFERENCE_SLOPE
            // (the parity option: the terms by free index, added in that order below -- minimize_nrc.h:665-672)
            double* GT = L.ws + 5ll * f0;
            GT[fi] = gi * gi;
            GT[n + fi] = (x + gi) * x;
#endif
        }
        sumk<3>(a, b, t);
#ifdef RDIS_REFERENCE_SLOPE
        a = sum_in_order(L.ws + 5ll * f0, n);
        b = sum_in_order(L.ws + 5ll * f0 + n, n);
#endif
        gg = a; dgg = b; test = t;
    }
    __device__ void cg_update(double gam) {
#pragma clang fp contract(off)
        for (int s = tid; s < ns; s += nt) {
            const int fi = SF[s];
            if (fi < 0) continue;
            const double gn = -XI[s];
            const double hn = gn + gam * h[fi];
            g[fi] = gn; h[fi] = hn; XI[s] = hn;
        }
        __syncthreads();
    }
};

template <int THREADS, int ROT, bool STALE = false>
__global__ void __launch_bounds__(THREADS, (THREADS <= 256 ? 2 : 1))
cgd_lds_kernel(ProblemView P, PlanView L, int maxiters, double ftol, int ns_cap, int ncb_cap, int chunk_cap) {
    extern __shared__ double lds_dyn[];
    __shared__ double red[2][3][MAX_WAVES];
    const int comp = L.order[blockIdx.x];
    const int f0 = L.free_ptr[comp], f1 = L.free_ptr[comp + 1];
    const int c0 = L.fac_ptr[comp], c1 = L.fac_ptr[comp + 1];
    const int n = f1 - f0, m = c1 - c0;

    if (m == 0) {  // nothing to optimise: return 0, leave x as it was (.cpp:26-29)
        for (int i = threadIdx.x; i < n; i += blockDim.x) L.xout[f0 + i] = L.xstart[f0 + i];
        if (threadIdx.x == 0) {
            L.fret[comp] = 0.0; L.delta[comp] = 0.0; L.iters[comp] = 0;
            L.status[comp] = EXIT_EMPTY; L.nfeval[comp] = 0; L.ngeval[comp] = 0;
            if (L.trace_n) L.trace_n[comp] = 0;
        }
        return;
    }
    const int s0 = L.ls_ptr[comp], ns = L.ls_ptr[comp + 1] - s0, ncb = L.ls_ncb[comp];
    double* base = lds_dyn;
    double* CG = base + LDS_DOUBLES_PER_SLOT * ns_cap + 7 * ncb_cap;
    int* CGC = (int*)(CG + 9 * chunk_cap);
    int* SF = CGC + chunk_cap;
    int* CHE = SF + ns_cap;
    double* CTR = reinterpret_cast<double*>(reinterpret_cast<char*>(lds_dyn) + lds_matrix_offset(lds_bytes_for(ns_cap, ncb_cap, chunk_cap)));
    double* CDR = CTR + LDS_TS * ncb_cap;   // (behind the launch's other arrays; allocated only when ls_matrix is set)
    for (int s = threadIdx.x; s < ns; s += blockDim.x) SF[s] = L.ls_free[s0 + s];
    __syncthreads();
    double* ws = L.ws + 5ll * f0;
    LdsEnv<ROT, (THREADS <= 512), STALE> E{P, L, comp, n, m, f0, c0, (int)threadIdx.x, (int)blockDim.x, (int)(blockDim.x >> 6),
                  ns, ncb, L.ls_obs + c0, L.ls_fidx + c0, L.ls_gperm + 64ll * L.ls_gptr[comp], L.ls_gptr[comp + 1] - L.ls_gptr[comp], CG, CGC, L.v2s_ptr + f0, L.ls_vid + s0,
                  base, base + ns_cap, base + 2 * ns_cap, base + 3 * ns_cap, base + 4 * ns_cap, base + LDS_DOUBLES_PER_SLOT * ns_cap,
                  CTR, CDR, SF, CHE, STALE ? L.st_ev + c0 : nullptr, STALE ? L.st_val + c0 : nullptr, 0, ws + 2ll * n, ws + 3ll * n,
                  red, 0,
                  L.trace ? L.trace + 4ll * L.trace_cap * comp : nullptr, 0, 0
#ifdef RDIS_COOP_TIMING
                  , {}
#endif
    };
    [[maybe_unused]] const long long tk0 = E.clock();

    __shared__ CgdMachine M;
    __shared__ Request Q[2];
    E.init_vectors();
    run_machine(E, M, Q, maxiters, ftol);
    // assign gdmin.p with sanitisation (.cpp:61); after a rollback X already holds clamp(x_init)
    if (!M.rolled_back) E.assign_p();
    for (int s = E.tid; s < ns; s += E.nt) {
        const int fi = SF[s];
        if (fi >= 0) { const double xv = E.X[s]; P.x[E.svid[s]] = xv; L.xout[f0 + fi] = xv; }
    }
    if (E.tid == 0) {
        L.fret[comp] = M.fret; L.delta[comp] = M.fret - M.finit; L.iters[comp] = M.iter;
        L.status[comp] = M.status(); L.nfeval[comp] = M.nfeval; L.ngeval[comp] = M.ngeval;
        if (L.trace_n) L.trace_n[comp] = E.trn;
#ifdef RDIS_COOP_TIMING
        if (blockIdx.x == 0 && L.timing) {
            E.tm[7] = E.clock() - tk0;
            for (int i = 0; i < 32; ++i) L.timing[i] = E.tm[i];
        }
#endif
    }
}

}  // namespace rdis_hip
