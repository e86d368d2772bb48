// solver_quad.hpp -- a few lanes solve one tiny component out of registers: four lanes each,
// sixteen components per wave (very many components), or sixteen lanes each, four per wave (fewer).
//
// With the cameras assigned, every point of a bundle-adjustment problem is a component of its
// own: 3 free variables, 2..29 factors (SURVEY.md 3.2b -- by count this is what RDIS asks the
// subspace solver for most: 13 837 of 13 850 calls of its ladybug run).  A workgroup per
// component (solver_wg.hpp) leaves 60 of its 64 lanes idle and the device holds only 2048 of
// them at a time.  Here a component gets a group of four lanes:
//   * every lane of the group keeps the component's (at most QUAD_MAX_VARS) CG vectors p, xi, g,
//     h, x_init and bounds in registers -- vector updates are redundant scalar work, no exchange;
//   * the factors are dealt round-robin to the four lanes; the two sums of a trial point and
//     the per-variable sums of a gradient are reduced over the quad with DPP moves (bit-identical
//     in the four lanes);
//   * each group has its own control machine and request slot in LDS; the group's first lane
//     steps it.  Groups of one wave are at different points of their solves: the wave runs the
//     union of their paths under execution masks (run_machine with Env::UNIFORM = false).
// Same CgdMachine, same requests, same trace records as the other solvers.  The groups are
// persistent: when a machine is done its group writes the results and takes the next component
// of the launch's list (run_machine: next_problem), until the list is empty.
//
// GroupEnv<16> gives a component a DPP row of sixteen lanes (four machines per wave): between a
// workgroup each and the quad solver when there are some thousands of components -- ladybug's
// 7776 points: 2.05 ms against 2.46 (workgroups) and 2.9 (quads).  A whole wave per component out
// of registers (G = 64, with each lane keeping its factor's constants for the whole solve) was
// built and measured too: no better than a 64-lane workgroup (2.36 against 2.46 ms) -- a trial
// point costs about 6000 cycles either way, most of it the control step and the reductions, and
// with two waves per SIMD the vector unit is saturated by lanes that are mostly idle.
#pragma once
#include "solver_wg.hpp"

namespace rdis_hip {

constexpr int QUAD_MAX_VARS = 4;
constexpr int QUAD_THREADS = 256;   // 64 groups per workgroup

template <int G>
__device__ __forceinline__ double group_sum(double v) {
    if constexpr (G == 4) {
        v += dpp_move<DPP_XOR1>(v);
        v += dpp_move<DPP_XOR2>(v);
        return v;
    } else {   // 16: one DPP row
        v += dpp_move<DPP_XOR1>(v);
        v += dpp_move<DPP_XOR2>(v);
        v += dpp_move<DPP_HALF_MIRROR>(v);
        v += dpp_move<DPP_MIRROR>(v);
        return v;
    }
}

template <int G>   // lanes per component: 4 or 16
struct GroupEnv {
    static constexpr bool UNIFORM = false;
    static constexpr int SPEC = 1;   // no speculative trial steps (minimizer.hpp)
    const ProblemView& P;
    const PlanView& L;
    int comp, n, m, f0, c0, sub;
    bool active;
    double p[QUAD_MAX_VARS], xi[QUAD_MAX_VARS], g[QUAD_MAX_VARS], h[QUAD_MAX_VARS], xinit[QUAD_MAX_VARS];
    double lo[QUAD_MAX_VARS], hi[QUAD_MAX_VARS], xt[QUAD_MAX_VARS];
    int fv[QUAD_MAX_VARS];
    double* tr;
    int trn, lm_count;
    // persistent groups: a group that finishes its component takes the next one off the launch's
    // list (heaviest first), so a wave is busy until the list is empty, not until its slowest
    // first component is done
    const int* list;
    int ncomp;
    int* queue;        // components handed out beyond the first round
    int first_round;   // = groups in the grid

    // component number ci of the list (or the next non-empty one) into the registers of the group;
    // false when the list is exhausted
    __device__ bool load_next(int ci) {
        for (;;) {
            if (ci >= ncomp) return false;
            comp = list[ci];
            f0 = L.free_ptr[comp]; c0 = L.fac_ptr[comp];
            n = L.free_ptr[comp + 1] - f0; m = L.fac_ptr[comp + 1] - c0;
            if (m > 0) break;
            if (sub == 0) {   // nothing to optimise: return 0, leave x as it was (.cpp:26-29)
                for (int t = 0; t < n; ++t) L.xout[f0 + t] = L.xstart[f0 + t];
                L.fret[comp] = 0.0; L.delta[comp] = 0.0; L.iters[comp] = 0;
                L.status[comp] = EXIT_EMPTY; L.nfeval[comp] = 0; L.ngeval[comp] = 0;
                if (L.trace_n) L.trace_n[comp] = 0;
            }
            ci = take();
        }
#pragma unroll
        for (int t = 0; t < QUAD_MAX_VARS; ++t) {
            const bool in = t < n;
            const int v = in ? L.free_vid[f0 + t] : 0;
            fv[t] = in ? v : -1;
            const double x0 = in ? L.xstart[f0 + t] : 0.0;
            p[t] = x0; xinit[t] = x0; xi[t] = 0.0; g[t] = 0.0; h[t] = 0.0; xt[t] = 0.0;
            lo[t] = in ? P.lo[v] : 0.0; hi[t] = in ? P.hi[v] : 0.0;
        }
        tr = L.trace ? L.trace + 4ll * L.trace_cap * comp : nullptr;
        trn = 0; lm_count = 0;
        return true;
    }
    __device__ int take() {   // the group's first lane draws, the others get its number
        int ci = 0;
        if (sub == 0) ci = first_round + atomicAdd(queue, 1);
        return __shfl(ci, 0, G);
    }
    // run_machine calls this when the machine is done: results out, next component in
    __device__ bool next_problem(const CgdMachine& M) {
        if (sub == 0) {
#pragma unroll
            for (int t = 0; t < QUAD_MAX_VARS; ++t) {
                if (t < n) {
                    const double xf = clampd(M.rolled_back ? xinit[t] : p[t], lo[t], hi[t]);
                    P.x[fv[t]] = xf;   // variables are left assigned (.cpp:61, :84-86)
                    L.xout[f0 + t] = xf;
                }
            }
            L.fret[comp] = M.fret; L.delta[comp] = M.fret - M.finit; L.iters[comp] = M.iter;
            L.status[comp] = M.status(); L.nfeval[comp] = M.nfeval; L.ngeval[comp] = M.ngeval;
            if (L.trace_n) L.trace_n[comp] = trn;
        }
        return load_next(take());
    }

    __device__ bool stepper() const { return active && sub == 0; }
    __device__ bool writer() const { return active && sub == 0; }
    // the request was stored by a lane of this wave: LDS operations of one wave complete in order
    __device__ void sync() const { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
    __device__ bool tracing() const { return tr != nullptr; }
    __device__ bool aborted() const { return false; }
    __device__ void tick(int, long long) {}
    __device__ long long clock() const { return 0; }
    __device__ void trace(int tag, double a, double b, double c) {
        if (tr != nullptr && sub == 0 && active) {
            if (trn < L.trace_cap) { double* r = tr + 4ll * trn; r[0] = (double)tag; r[1] = a; r[2] = b; r[3] = c; }
            ++trn;
        }
    }

    // value (and slope along xi) at the trial point xt, over this lane's share of the factors
    template <bool SLOPE>
    __device__ void eval_at(double& f, double& s) {
        double af = 0.0, as = 0.0;
        if (P.rot_mode == ROT_CAMFIX) {   // cameras constant in this launch: rotation records, point partials only
            for (int j = sub; j < m; j += G) {
                const int fid = L.fac_id[c0 + j];
                const int c = P.cam[fid], q = P.pt[fid];
                const double2 o = P.obs[fid];
                double v[12], d[3];
                BaFwd t;
                ba_load_rotation(P.xrot + c, t);
                v[0] = v[1] = v[2] = 0.0;
#pragma unroll
                for (int k = 3; k < 9; ++k) v[k] = P.x[c + k];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const int vid = q + k;
                    double val = 0.0, dv = 0.0;
                    bool fr = false;
#pragma unroll
                    for (int u = 0; u < QUAD_MAX_VARS; ++u)
                        if (u < n && vid == fv[u]) { val = xt[u]; dv = xi[u]; fr = true; }
                    v[9 + k] = fr ? val : P.x[vid];
                    d[k] = dv;
                }
                af += ba_project(v, o.x, o.y, t);
                if constexpr (SLOPE) {
                    double gg[12];
                    ba_adjoint(t, v, t.res0, t.res1, gg);
                    double acc = 0.0;
#pragma unroll
                    for (int k = 0; k < 3; ++k) acc = __builtin_fma(gg[9 + k], d[k], acc);   // (what the compiler made of acc += g d, said out loud)
                    as += acc;
                }
            }
            f = group_sum<G>(af);
            s = SLOPE ? group_sum<G>(as) : 0.0;
            return;
        }
        for (int j = sub; j < m; j += G) {
            const int fid = L.fac_id[c0 + j];
            const int c = P.cam[fid], q = P.pt[fid];
            const double2 o = P.obs[fid];
            double v[12], d[12];
#pragma unroll
            for (int k = 0; k < 12; ++k) {
                const int vid = k < 9 ? c + k : q + (k - 9);
                double val = 0.0, dv = 0.0;
                bool fr = false;
#pragma unroll
                for (int t = 0; t < QUAD_MAX_VARS; ++t)
                    if (t < n && vid == fv[t]) { val = xt[t]; dv = xi[t]; fr = true; }
                v[k] = fr ? val : P.x[vid];
                d[k] = dv;
            }
            if constexpr (SLOPE) {
                double gg[12];
                af += ba_eval_grad(v, o.x, o.y, gg);
                double acc = 0.0;
#pragma unroll
                for (int k = 0; k < 12; ++k) acc = __builtin_fma(gg[k], d[k], acc);
                as += acc;
            } else {
                af += ba_eval(v, o.x, o.y);
            }
        }
        f = group_sum<G>(af);
        s = SLOPE ? group_sum<G>(as) : 0.0;
    }
    __device__ void assign_line(double a) {
#pragma clang fp contract(off)
#pragma unroll
        for (int t = 0; t < QUAD_MAX_VARS; ++t) {
            const double u = a * xi[t];
            xt[t] = clampd(p[t] + u, lo[t], hi[t]);
        }
    }
    __device__ double eval_value(double a, bool restore) {
        if (restore) {
#pragma unroll
            for (int t = 0; t < QUAD_MAX_VARS; ++t) xt[t] = clampd(xinit[t], lo[t], hi[t]);
        } else {
            assign_line(a);
        }
        double f, s;
        eval_at<false>(f, s);
        return f;
    }
    __device__ void eval_value_slope(double a, double& f, double& s) {
        assign_line(a);
        eval_at<true>(f, s);
    }
    // full gradient at clamp(p): per-variable sums over this lane's factors, then over the quad
    __device__ void gradient_to_xi() {
        double acc[QUAD_MAX_VARS];
#pragma unroll
        for (int t = 0; t < QUAD_MAX_VARS; ++t) { xt[t] = clampd(p[t], lo[t], hi[t]); acc[t] = 0.0; }
        if (P.rot_mode == ROT_CAMFIX) {
            for (int j = sub; j < m; j += G) {
                const int fid = L.fac_id[c0 + j];
                const int c = P.cam[fid], q = P.pt[fid];
                const double2 o = P.obs[fid];
                double v[12], gg[12];
                int li[3];
                BaFwd t;
                ba_load_rotation(P.xrot + c, t);
                v[0] = v[1] = v[2] = 0.0;
#pragma unroll
                for (int k = 3; k < 9; ++k) v[k] = P.x[c + k];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const int vid = q + k;
                    double val = 0.0;
                    int l = -1;
#pragma unroll
                    for (int u = 0; u < QUAD_MAX_VARS; ++u)
                        if (u < n && vid == fv[u]) { val = xt[u]; l = u; }
                    v[9 + k] = l >= 0 ? val : P.x[vid];
                    li[k] = l;
                }
                ba_project(v, o.x, o.y, t);
                ba_adjoint(t, v, t.res0, t.res1, gg);
#pragma unroll
                for (int k = 0; k < 3; ++k)
#pragma unroll
                    for (int u = 0; u < QUAD_MAX_VARS; ++u)
                        if (li[k] == u) acc[u] += gg[9 + k];
            }
#pragma unroll
            for (int t = 0; t < QUAD_MAX_VARS; ++t) xi[t] = group_sum<G>(acc[t]);
            return;
        }
        for (int j = sub; j < m; j += G) {
            const int fid = L.fac_id[c0 + j];
            const int c = P.cam[fid], q = P.pt[fid];
            const double2 o = P.obs[fid];
            double v[12], gg[12];
            int li[12];
#pragma unroll
            for (int k = 0; k < 12; ++k) {
                const int vid = k < 9 ? c + k : q + (k - 9);
                double val = 0.0;
                int l = -1;
#pragma unroll
                for (int t = 0; t < QUAD_MAX_VARS; ++t)
                    if (t < n && vid == fv[t]) { val = xt[t]; l = t; }
                v[k] = l >= 0 ? val : P.x[vid];
                li[k] = l;
            }
            ba_eval_grad(v, o.x, o.y, gg);
#pragma unroll
            for (int k = 0; k < 12; ++k)
#pragma unroll
                for (int t = 0; t < QUAD_MAX_VARS; ++t)
                    if (li[k] == t) acc[t] += gg[k];
        }
#pragma unroll
        for (int t = 0; t < QUAD_MAX_VARS; ++t) xi[t] = group_sum<G>(acc[t]);
    }
    __device__ void cg_start() {
#pragma unroll
        for (int t = 0; t < QUAD_MAX_VARS; ++t) { const double u = -xi[t]; g[t] = u; h[t] = u; xi[t] = u; }
    }
    __device__ void line_begin() {
        if (L.vdump != nullptr && lm_count < L.dump_iters && sub == 0 && active) {
            double* d = L.vdump + 2ll * L.dump_iters * f0 + 2ll * lm_count * n;
#pragma unroll
            for (int t = 0; t < QUAD_MAX_VARS; ++t)   // (static indices: the vectors stay in registers)
                if (t < n) { d[t] = p[t]; d[n + t] = xi[t]; }
        }
        ++lm_count;
    }
    __device__ void line_end(double amin) {
#pragma clang fp contract(off)
#pragma unroll
        for (int t = 0; t < QUAD_MAX_VARS; ++t) { const double u = xi[t] * amin; xi[t] = u; p[t] = p[t] + u; }
    }
    __device__ void cg_reduce(double fp, double& test, double& gg, double& dgg) {
#pragma clang fp contract(off)
        const double den = fmax(fabs(fp), 1.0);
        double a = 0.0, b = 0.0, tt = 0.0;
#pragma unroll
        for (int t = 0; t < QUAD_MAX_VARS; ++t) {
            if (t < n) {
                tt = fmax(tt, fabs(xi[t]) * fmax(fabs(p[t]), 1.0) / den);
                a = a + g[t] * g[t];
                b = b + (xi[t] + g[t]) * xi[t];
            }
        }
        gg = a; dgg = b; test = tt;
    }
    __device__ void cg_update(double gam) {
#pragma clang fp contract(off)
#pragma unroll
        for (int t = 0; t < QUAD_MAX_VARS; ++t) {
            const double gn = -xi[t];
            const double hn = gn + gam * h[t];
            g[t] = gn; h[t] = hn; xi[t] = hn;
        }
    }
};

// list[0 .. ncomp): the components of this launch (each with at most QUAD_MAX_VARS free variables)
template <int G, int THREADS>
__global__ void __launch_bounds__(THREADS, G == 4 ? 1 : 2)
cgd_group_kernel(ProblemView P, PlanView L, const int* __restrict__ list, int ncomp, int* __restrict__ queue,
                 int maxiters, double ftol) {
    __shared__ CgdMachine Ms[THREADS / G];
    __shared__ Request Qs[THREADS / G][2];
    const int grp = threadIdx.x / G;
    GroupEnv<G> E{P, L, 0, 0, 0, 0, 0, (int)(threadIdx.x % G), false,
                  {}, {}, {}, {}, {}, {}, {}, {}, {},
                  nullptr, 0, 0, list, ncomp, queue, (int)(gridDim.x * (THREADS / G))};
    E.active = E.load_next((int)blockIdx.x * (THREADS / G) + grp);   // the first component: by position
    if (E.active) run_machine(E, Ms[grp], Qs[grp], maxiters, ftol);
}

}  // namespace rdis_hip
