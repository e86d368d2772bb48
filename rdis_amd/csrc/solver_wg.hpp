// solver_wg.hpp -- one workgroup solves one component: the whole
// CGDSubspaceOptimizer::optimize call (reference
// src/optimizers/CGDSubspaceOptimizer.cpp:19-98) stays on the device, and a batch
// of independent components is one launch (grid = components, heaviest first).
//
// Per line-search trial (SubfunctionFD::operator() / df, .cpp:124-184):
//   phase A  every free variable: x[vid] = clamp(p + a*xi)      (quickAssignVals)
//   phase B  every factor: gather 12 values of x, evaluate; for Brent also the 12
//            partials dotted with the search direction -> the slope along the line
//            needs no scatter at all
//   reduce   wave reduction (DPP) + LDS, fixed order => bit-reproducible run to run
// The full gradient is needed once per CG iteration only: per-factor partials go
// to gfac[], then each free variable sums its slots in factor-list order (the same
// order the reference's PartialGradient merge produces, src/State.h:157-210).
#pragma once
#include "device_views.hpp"
#include "factors.hpp"
#include "minimizer.hpp"

namespace rdis_hip {

constexpr int MAX_WAVES = 16;
constexpr int WG_LONG_LIST = 64;   // batch solver: variables fed by more partials than this are summed by a whole wave
constexpr int WG_LONG_QUEUE = 512; // ... up to this many per component (LDS queue)
// line-search slopes of bundle-adjustment factors in the batched / streaming solvers: forward mode along
// the search direction (factors.hpp ba_slope_dir) instead of the 12 partials dotted with it
constexpr bool FWD_SLOPE = true;

// Wave-wide reductions; every lane of a full wave returns the same bits.  Within a row of 16
// lanes the partner comes through DPP (quad_perm / row_half_mirror / row_mirror: an ALU-speed
// move, where a ds_bpermute shuffle costs an LDS round trip per step); the four row sums are
// then read as scalars and combined in a fixed order.  Must be called with all 64 lanes active.
template <int CTRL>
__device__ __forceinline__ double dpp_move(double v) {
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double read_lane(double v, int lane) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane),
                            __builtin_amdgcn_readlane(__double2loint(v), lane));
}
constexpr int DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E;            // quad_perm [1,0,3,2], [2,3,0,1]
constexpr int DPP_HALF_MIRROR = 0x141, DPP_MIRROR = 0x140;  // lane i <-> 7-i, i <-> 15-i in its row
__device__ __forceinline__ double wave_sum(double v) {
    v += dpp_move<DPP_XOR1>(v);
    v += dpp_move<DPP_XOR2>(v);
    v += dpp_move<DPP_HALF_MIRROR>(v);
    v += dpp_move<DPP_MIRROR>(v);
    return (read_lane(v, 0) + read_lane(v, 16)) + (read_lane(v, 32) + read_lane(v, 48));
}
__device__ __forceinline__ double wave_max(double v) {
    v = fmax(v, dpp_move<DPP_XOR1>(v));
    v = fmax(v, dpp_move<DPP_XOR2>(v));
    v = fmax(v, dpp_move<DPP_HALF_MIRROR>(v));
    v = fmax(v, dpp_move<DPP_MIRROR>(v));
    return fmax(fmax(read_lane(v, 0), read_lane(v, 16)), fmax(read_lane(v, 32), read_lane(v, 48)));
}

// Combines the per-wave partials a workgroup left in LDS (one entry per wave, red[k][wave]):
// lane i fetches entry i -- one LDS round trip instead of one per wave -- and, as there are at
// most 16 waves, a reduction over the first row of lanes finishes the job.  Every wave that
// calls it obtains the same bits.  K: 1 = a, 2 = a and b, 3 = a, b and the maximum mx.
template <int K>
__device__ __forceinline__ void combine_waves(const double (*red)[16], int nwaves, double& a, double& b, double& mx) {
    const int lane = threadIdx.x & 63;
    const bool in = lane < nwaves;
    double va = in ? red[0][lane] : 0.0, vb = 0.0, vm = 0.0;
    if constexpr (K >= 2) vb = in ? red[1][lane] : 0.0;
    if constexpr (K >= 3) vm = in ? red[2][lane] : 0.0;
    va += dpp_move<DPP_XOR1>(va); if constexpr (K >= 2) vb += dpp_move<DPP_XOR1>(vb); if constexpr (K >= 3) vm = fmax(vm, dpp_move<DPP_XOR1>(vm));
    va += dpp_move<DPP_XOR2>(va); if constexpr (K >= 2) vb += dpp_move<DPP_XOR2>(vb); if constexpr (K >= 3) vm = fmax(vm, dpp_move<DPP_XOR2>(vm));
    if (nwaves > 4) {
        va += dpp_move<DPP_HALF_MIRROR>(va); if constexpr (K >= 2) vb += dpp_move<DPP_HALF_MIRROR>(vb); if constexpr (K >= 3) vm = fmax(vm, dpp_move<DPP_HALF_MIRROR>(vm));
        va += dpp_move<DPP_MIRROR>(va); if constexpr (K >= 2) vb += dpp_move<DPP_MIRROR>(vb); if constexpr (K >= 3) vm = fmax(vm, dpp_move<DPP_MIRROR>(vm));
    }
    a = read_lane(va, 0);
    if constexpr (K >= 2) b = read_lane(vb, 0);
    if constexpr (K >= 3) mx = read_lane(vm, 0);
}

// Loads / stores that are coherent across the whole device without cache maintenance: relaxed
// agent-scope atomics (write-through stores, loads that do not hit a stale line of this XCD's L2).
template <bool COHERENT>
__device__ __forceinline__ double load_f64(const double* p) {
    if constexpr (COHERENT)
        return __longlong_as_double(__hip_atomic_load((const long long*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    else
        return *p;
}
template <bool COHERENT>
__device__ __forceinline__ void store_f64(double* p, double v) {
    if constexpr (COHERENT)
        __hip_atomic_store((long long*)p, __double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else
        *p = v;
}

// Sum of the contiguous run g[b..e) in index order (factor-list order, src/State.h:157-210).
// BATCH loads are in flight at a time -- after an exchange every one of them comes from memory,
// and one round trip per element is what a plain loop pays -- while the additions stay strictly
// sequential, so the result has the bits of the plain loop.
template <bool COHERENT = false, int BATCH = 8>
__device__ __forceinline__ double run_sum_ordered(const double* __restrict__ g, int b, int e) {
    double s = 0.0;
    for (int k0 = b; k0 < e; k0 += BATCH) {
        double t[BATCH];
#pragma unroll
        for (int j = 0; j < BATCH; ++j) t[j] = (k0 + j < e) ? load_f64<COHERENT>(g + k0 + j) : 0.0;
#pragma unroll
        for (int j = 0; j < BATCH; ++j)
            if (k0 + j < e) s = (k0 + j == b) ? t[j] : s + t[j];
    }
    return s;
}
// This lane's share of a long run: elements b + lane, b + lane + 64, ... (eight loads in flight);
// element k lives at g[k * pitch]
template <bool COHERENT = false>
__device__ __forceinline__ double run_sum_strided(const double* __restrict__ g, int b, int e, int lane, int pitch = 1) {
    double s = 0.0;
    for (int k0 = b + lane; k0 < e; k0 += 512) {
        double t[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] = (k0 + 64 * j < e) ? load_f64<COHERENT>(g + (long long)(k0 + 64 * j) * pitch) : 0.0;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (k0 + 64 * j < e) s += t[j];
    }
    return s;
}

__device__ __forceinline__ double clampd(double v, double lo, double hi) {
    // VariableDomain::closestVal, single interval (reference src/VariableDomain.cpp:158-163)
    if (lo <= v && v <= hi) return v;
    return v < lo ? lo : hi;
}

// ---- factor evaluation shared by all solver variants ---------------------------
// ROT (device_views.hpp): with ROT_RECORDS / ROT_CAMFIX the factor reads its camera's rotation
// record from P.xrot instead of forming angle, axis, sine and cosine itself (same arithmetic, done
// once per camera and trial point instead of once per factor); ROT_CAMFIX also forms only the
// point's three partials.
template <int KIND, bool SLOPE, int ROT = ROT_PER_FACTOR>
__device__ __forceinline__ void factor_value(const ProblemView& P, const double* __restrict__ dir,
                                             int fid, double& f, double& s) {
    if constexpr (KIND == KIND_BA && ROT != ROT_PER_FACTOR) {
        const int c = P.cam[fid], q = P.pt[fid];
        const double2 o = P.obs[fid];
        double v[12];
        BaFwd t;
        ba_load_rotation(P.xrot + c, t);
        v[0] = v[1] = v[2] = 0.0;
#pragma unroll
        for (int k = 3; k < 9; ++k) v[k] = P.x[c + k];
#pragma unroll
        for (int k = 0; k < 3; ++k) v[9 + k] = P.x[q + k];
        f = ba_project(v, o.x, o.y, t);
        s = 0.0;
        if constexpr (SLOPE && FWD_SLOPE) {
            double d[12];
#pragma unroll
            for (int k = 0; k < 9; ++k) d[k] = (ROT == ROT_RECORDS) ? dir[c + k] : 0.0;
#pragma unroll
            for (int k = 0; k < 3; ++k) d[9 + k] = dir[q + k];
            s = ba_slope_dir<ROT == ROT_CAMFIX>(t, v, d);
        } else if constexpr (SLOPE) {
            double g[12];
            ba_adjoint(t, v, t.res0, t.res1, g);
            double acc = 0.0;
            if constexpr (ROT == ROT_RECORDS) {
#pragma unroll
                for (int k = 0; k < 9; ++k) acc += g[k] * dir[c + k];
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) acc += g[9 + k] * dir[q + k];
            s = acc;
        }
    } else if constexpr (KIND == KIND_BA) {
        const int c = P.cam[fid], q = P.pt[fid];
        const double2 o = P.obs[fid];
        double v[12];
#pragma unroll
        for (int k = 0; k < 9; ++k) v[k] = P.x[c + k];
#pragma unroll
        for (int k = 0; k < 3; ++k) v[9 + k] = P.x[q + k];
        if constexpr (SLOPE && FWD_SLOPE) {
            BaFwd t;
            f = ba_forward(v, o.x, o.y, t);
            double d[12];
#pragma unroll
            for (int k = 0; k < 9; ++k) d[k] = dir[c + k];
#pragma unroll
            for (int k = 0; k < 3; ++k) d[9 + k] = dir[q + k];
            s = ba_slope_dir<false>(t, v, d);
        } else if constexpr (SLOPE) {
            double g[12];
            f = ba_eval_grad(v, o.x, o.y, g);
            double acc = 0.0;
#pragma unroll
            for (int k = 0; k < 9; ++k) acc += g[k] * dir[c + k];
#pragma unroll
            for (int k = 0; k < 3; ++k) acc += g[9 + k] * dir[q + k];
            s = acc;
        } else {
            f = ba_eval(v, o.x, o.y);
            s = 0.0;
        }
    } else {
        const int b = P.rowptr[fid], e = P.rowptr[fid + 1];
        double prod = 1.0;
        for (int k = b; k < e; ++k) prod *= nlp_term(P.x[P.vid[k]], P.expo[k], P.cons[k], P.sine[k] != 0);
        // useExponential (NonlinearProductFactor.cpp:140, 204): values only -- the library refuses a gradient or a
        // solve over such a factor (the reference's computeGradient asserts it off, :110)
        if (!SLOPE && P.useexp && P.useexp[fid]) prod = exp(-prod);
        f = prod * P.coeff[fid];
        s = 0.0;
        if constexpr (SLOPE) {
            for (int k = b; k < e; ++k) {
                const double dk = dir[P.vid[k]];
                double d = 1.0;
                for (int j = b; j < e; ++j) {
                    const double xv = P.x[P.vid[j]];
                    if (j == k) {
                        if (P.expo[j] == 1.0 && !P.sine[j]) continue;
                        d *= nlp_dterm(xv, P.expo[j], P.cons[j], P.sine[j] != 0);
                    } else {
                        d *= nlp_term(xv, P.expo[j], P.cons[j], P.sine[j] != 0);
                    }
                }
                s = __builtin_fma(d * P.coeff[fid], dk, s);   // (one fused step a variable: what the compiler made of s += d c dk, said out loud)
            }
        }
    }
}

// per-factor partials into their gradient slots.  Slot id: BA 12*fid + k, NLP the CSR
// position.  With sp (this factor's row of PlanView::slot_pos) partial k goes to gfac[sp[k]]
// (variable-major, skipped when negative); without it to gfac[global slot id].
template <int KIND, int ROT = ROT_PER_FACTOR>
__device__ __forceinline__ void factor_partials(const ProblemView& P, double* __restrict__ gfac,
                                                const int* __restrict__ sp, int fid) {
    if constexpr (KIND == KIND_BA && ROT != ROT_PER_FACTOR) {   // (sp given: the batch solvers)
        const int c = P.cam[fid], q = P.pt[fid];
        const double2 o = P.obs[fid];
        double v[12], g[12];
        BaFwd t;
        ba_load_rotation(P.xrot + c, t);
        v[0] = v[1] = v[2] = 0.0;
#pragma unroll
        for (int k = 3; k < 9; ++k) v[k] = P.x[c + k];
#pragma unroll
        for (int k = 0; k < 3; ++k) v[9 + k] = P.x[q + k];
        ba_project(v, o.x, o.y, t);
        ba_adjoint(t, v, t.res0, t.res1, g);
#pragma unroll
        for (int k = (ROT == ROT_CAMFIX ? 9 : 0); k < 12; ++k) { const int u = sp[k]; if (u >= 0) gfac[u] = g[k]; }
    } else if constexpr (KIND == KIND_BA) {
        const int c = P.cam[fid], q = P.pt[fid];
        const double2 o = P.obs[fid];
        double v[12], g[12];
#pragma unroll
        for (int k = 0; k < 9; ++k) v[k] = P.x[c + k];
#pragma unroll
        for (int k = 0; k < 3; ++k) v[9 + k] = P.x[q + k];
        ba_eval_grad(v, o.x, o.y, g);
        if (sp) {
#pragma unroll
            for (int k = 0; k < 12; ++k) { const int t = sp[k]; if (t >= 0) gfac[t] = g[k]; }
        } else {
            double* dst = gfac + 12ll * fid;
#pragma unroll
            for (int k = 0; k < 12; ++k) dst[k] = g[k];
        }
    } else {
        const int b = P.rowptr[fid], e = P.rowptr[fid + 1];
        for (int k = b; k < e; ++k) {
            double d = 1.0;
            for (int j = b; j < e; ++j) {
                const double xv = P.x[P.vid[j]];
                if (j == k) {
                    if (P.expo[j] == 1.0 && !P.sine[j]) continue;
                    d *= nlp_dterm(xv, P.expo[j], P.cons[j], P.sine[j] != 0);
                } else {
                    d *= nlp_term(xv, P.expo[j], P.cons[j], P.sine[j] != 0);
                }
            }
            const int t = sp ? sp[k - b] : k;
            if (t >= 0) gfac[t] = d * P.coeff[fid];
        }
    }
}

// With ROT_RECORDS a component rewrites the rotation records of its free cameras whenever it
// assigns a point: lanes first, first + stride, ... take a camera each and form its three rotation
// variables the way the assignment does (at(i): the clamped value given to free variable i; a
// constant where the index is negative) -- no need to wait for x, the records become visible with it.
// (at() reads vector entries other lanes wrote: the caller's vector updates end with a barrier.  The
// streaming grid solver has none there -- and measured only 2.5 % with records -- so it forms its
// rotations per factor.)
template <class At>
__device__ __forceinline__ void refresh_rotation_records(const ProblemView& P, const PlanView& L, int comp,
                                                         int first, int stride, At at) {
    if (P.rot_mode != ROT_RECORDS) return;
    const int b = L.cb_ptr[comp], e = L.cb_ptr[comp + 1];
    for (int i = b + first; i < e; i += stride) {
        const int c = L.cb[i];
        const int* li = L.cb_li + 3 * i;
        double r[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) r[k] = li[k] >= 0 ? at(li[k]) : P.x[c + k];
        store_rotation(r[0], r[1], r[2], P.xrot + c);
    }
}

// ---- the single-workgroup environment ---------------------------------------------
template <int KIND>
struct WgEnv {
    const ProblemView& P;
    const PlanView& L;
    int comp, n, m, fac0, tid, nt, nwaves;
    const int* fv;    // free variable ids of this component
    const int* fl;    // factor ids of this component
    const int* vptr;  // v2s_ptr + free offset
    double *p, *xi, *g, *h, *xinit;
    double (*red)[3][MAX_WAVES];  // LDS [2][3][MAX_WAVES]
    int parity;
    int *long_q, *long_n;         // LDS queue of the variables with long runs (gradient_to_xi)
    double* tr;
    int trn;
    int lm_count;  // line minimisations started

    // workgroup-wide sums of the first K of (a, b, max mx), the same bits in every lane
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
            if (trn < L.trace_cap) {
                double* r = tr + 4ll * trn;
                r[0] = (double)tag; r[1] = a; r[2] = b; r[3] = c;
            }
            ++trn;
        }
    }

    // SubfunctionFD::quickAssignVals at p + a*xi (reference .cpp:160-184; the
    // trial point is formed unfused like minimize_nrc.h:434)
    __device__ void assign_line(double a) {
#pragma clang fp contract(off)
        for (int i = tid; i < n; i += nt) {
            const int v = fv[i];
            const double t = a * xi[i];
            P.x[v] = clampd(p[i] + t, P.lo[v], P.hi[v]);
        }
        if constexpr (KIND == KIND_BA)
            refresh_rotation_records(P, L, comp, tid, nt, [&](int i) {
#pragma clang fp contract(off)
                const int v = fv[i];
                const double t = a * xi[i];
                return clampd(p[i] + t, P.lo[v], P.hi[v]);
            });
        __syncthreads();
    }
    __device__ void assign_vec(const double* src) {
        for (int i = tid; i < n; i += nt) {
            const int v = fv[i];
            P.x[v] = clampd(src[i], P.lo[v], P.hi[v]);
        }
        if constexpr (KIND == KIND_BA)
            refresh_rotation_records(P, L, comp, tid, nt, [&](int i) { const int v = fv[i]; return clampd(src[i], P.lo[v], P.hi[v]); });
        __syncthreads();
    }

    template <bool SLOPE>
    __device__ void eval_sum(double& f, double& s) {
        double af = 0.0, as = 0.0, dummy = 0.0;
        for (int j = tid; j < m; j += nt) {
            double fj, sj;
            if (KIND == KIND_BA && P.rot_mode == ROT_CAMFIX) factor_value<KIND, SLOPE, ROT_CAMFIX>(P, L.dir, fl[j], fj, sj);
            else if (KIND == KIND_BA && P.rot_mode == ROT_RECORDS) factor_value<KIND, SLOPE, ROT_RECORDS>(P, L.dir, fl[j], fj, sj);
            else factor_value<KIND, SLOPE>(P, L.dir, fl[j], fj, sj);
            af += fj;
            if constexpr (SLOPE) as += sj;
        }
        sumk<SLOPE ? 2 : 1>(af, as, dummy);
        f = af; s = as;
    }

    static constexpr bool UNIFORM = true;
    static constexpr int SPEC = 1;   // no speculative trial steps (minimizer.hpp)
    __device__ bool stepper() const { return threadIdx.x < 64; }
    __device__ bool writer() const { return (threadIdx.x & 63) == 0; }
    __device__ void sync() const { __syncthreads(); }
    __device__ bool tracing() const { return tr != nullptr; }
    __device__ bool aborted() const { return false; }
    __device__ void tick(int, long long) {}
    __device__ long long clock() const { return 0; }
    // value at clamp(p + a*xi), or at clamp(x_init) for the rollback (CGD .cpp:71)
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
    __device__ void init_vectors() {  // CGD .cpp:34-39: p = x0 (unclamped), keep x0 for the rollback
        const double* xs = L.xstart + (fv - L.free_vid);
        for (int i = tid; i < n; i += nt) { p[i] = xs[i]; xinit[i] = xs[i]; xi[i] = 0.0; }
        __syncthreads();
    }

    // SubfunctionFD::df(p, xi) (reference .cpp:135-157): full gradient at clamp(p)
    __device__ void gradient_to_xi() {
        assign_vec(p);
        for (int j = tid; j < m; j += nt) {
            const int fid = fl[j];
            if (KIND == KIND_BA && P.rot_mode == ROT_CAMFIX) factor_partials<KIND, ROT_CAMFIX>(P, L.gfac, L.slot_pos + L.slot_base[fac0 + j], fid);
            else if (KIND == KIND_BA && P.rot_mode == ROT_RECORDS) factor_partials<KIND, ROT_RECORDS>(P, L.gfac, L.slot_pos + L.slot_base[fac0 + j], fid);
            else factor_partials<KIND>(P, L.gfac, L.slot_pos + L.slot_base[fac0 + j], fid);
        }
        __syncthreads();
        // variables fed by few partials: one lane each, in factor-list order (src/State.h:157-210);
        // those fed by many (a camera against fixed points: hundreds) are queued for a whole wave each
        if (tid == 0) *long_n = 0;
        __syncthreads();
        for (int i = tid; i < n; i += nt) {
            const int b = vptr[i], e = vptr[i + 1];
            if (e - b > WG_LONG_LIST) {
                const int k = atomicAdd(long_n, 1);
                if (k < WG_LONG_QUEUE) { long_q[k] = i; continue; }   // (queue full: summed right here)
            }
            double s = 0.0;
            if (b < e) {
                s = run_sum_ordered(L.gfac, b, e);
            }
            xi[i] = s;
        }
        __syncthreads();
        const int nq = min(*long_n, WG_LONG_QUEUE);
        if (nq > 0) {
            for (int k = tid >> 6; k < nq; k += nwaves) {   // strided over the run + wave reduction
                const int i = long_q[k];
                const double s = wave_sum(run_sum_strided(L.gfac, vptr[i], vptr[i + 1], tid & 63));
                if ((tid & 63) == 0) xi[i] = s;
            }
            __syncthreads();
        }
    }

    __device__ void cg_start() {
        for (int i = tid; i < n; i += nt) { const double t = -xi[i]; g[i] = t; h[i] = t; xi[i] = t; }
        __syncthreads();
    }
    __device__ void line_begin() {
        for (int i = tid; i < n; i += nt) L.dir[fv[i]] = xi[i];
        if (L.vdump != nullptr && lm_count < L.dump_iters) {  // debug aid for the replay check
            double* d = L.vdump + 2ll * L.dump_iters * (fv - L.free_vid) + 2ll * lm_count * n;
            for (int i = tid; i < n; i += nt) { d[i] = p[i]; d[n + i] = xi[i]; }
        }
        ++lm_count;
        __syncthreads();
    }
    __device__ void line_end(double amin) {
#pragma clang fp contract(off)
        for (int i = tid; i < n; i += nt) {
            const double t = xi[i] * amin;
            xi[i] = t;
            p[i] = p[i] + t;
        }
        __syncthreads();
    }
    __device__ void cg_reduce(double fp, double& test, double& gg, double& dgg) {
#pragma clang fp contract(off)
        const double den = fmax(fabs(fp), 1.0);
        double a = 0.0, b = 0.0, t = 0.0;
        for (int i = tid; i < n; i += nt) {
            const double x = xi[i], gi = g[i];
            t = fmax(t, fabs(x) * fmax(fabs(p[i]), 1.0) / den);
            a = a + gi * gi;
            b = b + (x + gi) * x;
        }
        sumk<3>(a, b, t);
        gg = a; dgg = b; test = t;
    }
    __device__ void cg_update(double gam) {
#pragma clang fp contract(off)
        for (int i = tid; i < n; i += nt) {
            const double gn = -xi[i];
            const double hn = gn + gam * h[i];
            g[i] = gn; h[i] = hn; xi[i] = hn;
        }
        __syncthreads();
    }
};

// (second launch-bounds argument, HIP: minimum waves per SIMD.  Two waves of small workgroups must
// fit: without it the allocator takes a handful of registers more than 256 and halves the occupancy)
template <int KIND, int THREADS>
__global__ void __launch_bounds__(THREADS, (THREADS <= 256 ? 2 : 1))
cgd_wg_kernel(ProblemView P, PlanView L, int maxiters, double ftol) {
    __shared__ double red[2][3][MAX_WAVES];
    __shared__ int long_q[WG_LONG_QUEUE];
    __shared__ int long_n;
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

    double* ws = L.ws + 5ll * f0;
    WgEnv<KIND> E{P, L, comp, n, m, c0, (int)threadIdx.x, (int)blockDim.x, (int)(blockDim.x >> 6),
                  L.free_vid + f0, L.fac_id + c0, L.v2s_ptr + f0,
                  ws, ws + n, ws + 2ll * n, ws + 3ll * n, ws + 4ll * n,
                  red, 0, long_q, &long_n,
                  L.trace ? L.trace + 4ll * L.trace_cap * comp : nullptr, 0, 0};

    __shared__ CgdMachine M;
    __shared__ Request Q[2];
    E.init_vectors();
    run_machine(E, M, Q, maxiters, ftol);
    // assign gdmin.p with sanitisation (.cpp:61); after a rollback x already holds clamp(x_init)
    if (!M.rolled_back) E.assign_vec(E.p);
    for (int i = E.tid; i < n; i += E.nt) {
        L.xout[f0 + i] = P.x[E.fv[i]];
        L.dir[E.fv[i]] = 0.0;  // dir is shared by all plans of the problem: leave it zero
    }
    if (E.tid == 0) {
        L.fret[comp] = M.fret; L.delta[comp] = M.fret - M.finit; L.iters[comp] = M.iter;
        L.status[comp] = M.status(); L.nfeval[comp] = M.nfeval; L.ngeval[comp] = M.ngeval;
        if (L.trace_n) L.trace_n[comp] = E.trn;
    }
}

}  // namespace rdis_hip
