// solver_stream.hpp -- a whole grid solves ONE component that is too large for the
// register-resident cooperative solver (more factors than resident lanes), or whose factors
// are not bundle-adjustment factors.
//
// Same persistent-launch structure as solver_coop.hpp (every workgroup steps the same state
// machine on the same reduced values, granule exchange of grid_sync.hpp), but the state
// streams through L2 / HBM like in the single-workgroup solver:
//   per trial point   phase A  lanes stride over the free variables: x[vid] = clamp(p + a*xi)
//                     -- ordered grid barrier --
//                     phase B  lanes stride over the factors: gather x (and dir), evaluate
//                     -- exchange of the partial sums --
//   per CG iteration  per-factor partials scattered variable-major, ordered barrier, per-variable
//                     sums (a lane for short runs, a whole wave for long ones), ordered barrier.
// This is the regime where the HBM roofline of SURVEY.md 8(d) applies: 24 B per factor and
// 8-16 B per variable per evaluation.
#pragma once
#include "grid_sync.hpp"

namespace rdis_hip {

constexpr int STREAM_LONG_LIST = 64;

struct StreamArgs {
    long long* timing;
    CoopState* st;
    const int* long_vars;  // local indices of the variables fed by more than STREAM_LONG_LIST partials
    int nlong;
    int comp;
    int poll_delay;
};

template <int KIND>
struct StreamEnv {
    const ProblemView& P;
    const PlanView& L;
    const StreamArgs& A;
    int n, m, f0, c0;
    int gt, gsz, tid;     // global lane, lanes in the grid, lane in workgroup
    GridSync X;
    const int* fv;        // free variable ids
    const int* fl;        // factor ids
    const int* vptr;      // v2s_ptr + f0
    double *p, *xi, *g, *h, *xinit;
    double* tr;
    int trn, lm_count;

    __device__ void trace(int tag, double a, double b, double c) {
        if (tr != nullptr && gt == 0) {
            if (trn < L.trace_cap) { double* r = tr + 4ll * trn; r[0] = (double)tag; r[1] = a; r[2] = b; r[3] = c; }
            ++trn;
        }
    }
    static constexpr bool UNIFORM = true;
    static constexpr int SPEC = 1;   // no speculative trial steps (minimizer.hpp)
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

    // SubfunctionFD::quickAssignVals (reference CGDSubspaceOptimizer.cpp:160-184) across the grid
    __device__ void assign_line(double a) {
        {
#pragma clang fp contract(off)
            for (int i = gt; i < n; i += gsz) {
                const int v = fv[i];
                const double t = a * xi[i];
                P.x[v] = clampd(p[i] + t, P.lo[v], P.hi[v]);
            }
        }
        X.barrier_ordered();
    }
    __device__ void assign_vec(const double* src) {
        for (int i = gt; i < n; i += gsz) {
            const int v = fv[i];
            P.x[v] = clampd(src[i], P.lo[v], P.hi[v]);
        }
        X.barrier_ordered();
    }
    template <bool SLOPE>
    __device__ void eval_sum(double& f, double& s) {
        const long long tc0 = coop_clock();
        double af = 0.0, as = 0.0, dummy = 0.0;
        for (int j = gt; j < m; j += gsz) {
            double fj, sj;
            factor_value<KIND, SLOPE>(P, L.dir, fl[j], fj, sj);
            af += fj;
            if constexpr (SLOPE) as += sj;
        }
        X.tm[0] += coop_clock() - tc0;
        X.exchange(af, as, dummy, SYNC_NONE);
        f = af; s = as;
    }
    __device__ double eval_value(double a, bool restore) {
        if (restore) assign_vec(xinit); else assign_line(a);
        double f, s;
        eval_sum<false>(f, s);
        return f;
    }
    __device__ void eval_value_slope(double a, double& f, double& s) {
        assign_line(a);
        eval_sum<true>(f, s);
    }
    __device__ void init_vectors() {
        const double* xs = L.xstart + f0;
        for (int i = gt; i < n; i += gsz) { p[i] = xs[i]; xinit[i] = xs[i]; xi[i] = 0.0; }
    }

    __device__ void gradient_to_xi() {
        assign_vec(p);
        for (int j = gt; j < m; j += gsz) factor_partials<KIND>(P, L.gfac, L.slot_pos + L.slot_base[c0 + j], fl[j]);
        X.barrier_ordered();
        for (int i = gt; i < n; i += gsz) {  // short runs: serial, factor-list order
            const int b = vptr[i], e = vptr[i + 1];
            if (e - b > STREAM_LONG_LIST) continue;
            double s = 0.0;
            if (b < e) {
                s = run_sum_ordered(L.gfac, b, e);
            }
            xi[i] = s;
        }
        const int lane = tid & 63, gw = gt >> 6, nw = gsz >> 6;
        for (int q = gw; q < A.nlong; q += nw) {  // long runs: a wave strides over the run, then a butterfly
            const int i = A.long_vars[q];
            const int b = vptr[i], e = vptr[i + 1];
            const double s = wave_sum(run_sum_strided(L.gfac, b, e, lane));
            if (lane == 0) xi[i] = s;
        }
        X.barrier_ordered();  // the long runs were summed by other lanes than their owners
    }
    __device__ void cg_start() {
        for (int i = gt; i < n; i += gsz) { const double t = -xi[i]; g[i] = t; h[i] = t; xi[i] = t; }
    }
    __device__ void line_begin() {
        // dir is read by the factor lanes of other workgroups: the ordered barrier of the first
        // trial point (assign_line) publishes it together with x
        for (int i = gt; i < n; i += gsz) L.dir[fv[i]] = xi[i];
        if (L.vdump != nullptr && lm_count < L.dump_iters) {
            double* d = L.vdump + 2ll * L.dump_iters * f0 + 2ll * lm_count * n;
            for (int i = gt; i < n; i += gsz) { d[i] = p[i]; d[n + i] = xi[i]; }
        }
        ++lm_count;
    }
    __device__ void line_end(double amin) {
#pragma clang fp contract(off)
        for (int i = gt; i < n; i += gsz) {
            const double t = xi[i] * amin;
            xi[i] = t;
            p[i] = p[i] + t;
        }
    }
    __device__ void cg_reduce(double fp, double& test, double& gg, double& dgg) {
        double a = 0.0, b = 0.0, t = 0.0;
        {
#pragma clang fp contract(off)
            const double den = fmax(fabs(fp), 1.0);
            for (int i = gt; i < n; i += gsz) {
                const double x = xi[i], gi = g[i];
                t = fmax(t, fabs(x) * fmax(fabs(p[i]), 1.0) / den);
                a = a + gi * gi;
                b = b + (x + gi) * x;
            }
        }
        X.exchange(a, b, t, SYNC_NONE);
        gg = a; dgg = b; test = t;
    }
    __device__ void cg_update(double gam) {
#pragma clang fp contract(off)
        for (int i = gt; i < n; i += gsz) {
            const double gn = -xi[i];
            const double hn = gn + gam * h[i];
            g[i] = gn; h[i] = hn; xi[i] = hn;
        }
    }
};

template <int KIND, int THREADS>
__global__ void __launch_bounds__(THREADS)
cgd_stream_kernel(ProblemView P, PlanView L, StreamArgs A, int maxiters, double ftol) {
    __shared__ double bcast[8];
    __shared__ CgdMachine M;
    __shared__ Request Q[2];
    const long long tk0 = coop_clock();
    const int comp = A.comp;
    const int f0 = L.free_ptr[comp], c0 = L.fac_ptr[comp];
    const int n = L.free_ptr[comp + 1] - f0, m = L.fac_ptr[comp + 1] - c0;
    const int gt = blockIdx.x * blockDim.x + threadIdx.x, gsz = gridDim.x * blockDim.x;
    double* ws = L.ws + 5ll * f0;

    StreamEnv<KIND> E{P, L, A, n, m, f0, c0, gt, gsz, (int)threadIdx.x,
                      GridSync{A.st, (int)threadIdx.x, (int)gridDim.x, (int)blockIdx.x, bcast, A.poll_delay, 0, 0u, false, 0u, {}},
                      L.free_vid + f0, L.fac_id + c0, L.v2s_ptr + f0,
                      ws, ws + n, ws + 2ll * n, ws + 3ll * n, ws + 4ll * n,
                      L.trace ? L.trace + 4ll * L.trace_cap * comp : nullptr, 0, 0};
    E.init_vectors();
    run_machine(E, M, Q, maxiters, ftol);
    int status = M.status();
    const bool restore = M.rolled_back;
    for (int i = gt; i < n; i += gsz) {
        const int v = E.fv[i];
        const double xf = clampd(restore ? E.xinit[i] : E.p[i], P.lo[v], P.hi[v]);
        P.x[v] = xf;            // variables are left assigned (reference .cpp:61, :84-86)
        L.xout[f0 + i] = xf;
        L.dir[v] = 0.0;         // dir is shared by all plans of the problem: leave it zero
    }
    if (gt == 0) {
        L.fret[comp] = M.fret; L.delta[comp] = M.fret - M.finit; L.iters[comp] = M.iter;
        L.status[comp] = status; L.nfeval[comp] = M.nfeval; L.ngeval[comp] = M.ngeval;
        if (L.trace_n) L.trace_n[comp] = E.trn;
        E.X.tm[7] = coop_clock() - tk0;
        if (A.timing) for (int i = 0; i < COOP_TM; ++i) A.timing[i] = E.X.tm[i];
    }
}

constexpr int STREAM_THREADS = 512;

inline const void* stream_kernel_ptr(int kind) {
    return kind == KIND_BA ? (const void*)cgd_stream_kernel<KIND_BA, STREAM_THREADS>
                           : (const void*)cgd_stream_kernel<KIND_NLP, STREAM_THREADS>;
}

inline int launch_stream(hipStream_t stream, int kind, const ProblemView& P, const PlanView& V, const StreamArgs& a_in,
                         int nwg, int maxiters, double ftol) {
    hipError_t e = hipMemsetAsync(a_in.st, 0xFF, sizeof(CoopState) - 64, stream);  // arm every granule
    if (e != hipSuccess) return (int)e;
    e = hipMemsetAsync((char*)a_in.st + offsetof(CoopState, abort_flag), 0, 64, stream);
    if (e != hipSuccess) return (int)e;
    ProblemView p = P;
    PlanView v = V;
    StreamArgs a = a_in;
    int mi = maxiters;
    double ft = ftol;
    void* args[] = {&p, &v, &a, &mi, &ft};
    e = hipLaunchCooperativeKernel(stream_kernel_ptr(kind), dim3(nwg), dim3(STREAM_THREADS), args, 0, stream);
    return (int)e;
}

inline int stream_max_workgroups(int kind, int num_cus) {
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, stream_kernel_ptr(kind), STREAM_THREADS, 0) != hipSuccess) return 0;
    if (per_cu > 1) per_cu -= 1;
    const long long cap = (long long)per_cu * num_cus;
    return (int)(cap > COOP_MAX_WG ? COOP_MAX_WG : cap);
}

}  // namespace rdis_hip
