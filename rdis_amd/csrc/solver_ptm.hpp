// solver_ptm.hpp -- one workgroup solves one bundle-adjustment component that is too large for the
// LDS-resident solver (solver_lds.hpp): its CAMERA blocks keep their slots in LDS, its POINT blocks
// stream from HBM once per trial point, and the trial loop runs point by point.
//
// What bounds a batch of many large components is instruction issue (fp64 VALU) -- as long as the
// memory system is not in the way.  solver_wg.hpp forms every trial point in global memory (p, xi,
// lo, hi read and x written per variable, then x and the direction gathered per FACTOR: measured
// 2.8 x the algorithmic bytes in HBM traffic on 256 components of ladybug's size).  Here
//
//   cameras   slots in LDS exactly as in solver_lds.hpp (Pv, XI, LO, HI, X, rotation records);
//   points    one 96-byte record per point block in HBM: p[3], xi[3], lo[3], hi[3].  A lane takes a
//             point: loads the record ONCE, forms clamp(p + a xi) in registers and evaluates the
//             point's factors one after the other (per-point factor lists: camera block + observation,
//             20 bytes per factor, in point order: streamed, coalesced) against the cameras in LDS.
//             The trial point of a point variable is never stored.
//
// HBM bytes per value+slope trial: 32 per point variable + 20 per factor (SURVEY 8d counts 16 + 24:
// x and g once per variable, observation + two indices per factor) -- 1.3 x the algorithmic bytes on
// ladybug-shaped components, no write traffic at all.  Points are ordered by their number of factors
// (descending) so that the lanes of a wave run loops of equal length.
//
// The full gradient, once per CG iteration, runs camera by camera like solver_lds.hpp's (camera
// partials summed across a wave, point partials through gfac[] and summed per variable in
// factor-list order, src/State.h:157-210); the vector updates stream over the point records.
#pragma once
#include "solver_lds.hpp"

namespace rdis_hip {

constexpr int PT_REC = 12;   // doubles per point record: p, xi, lo, hi of the block's three variables
__host__ __device__ inline size_t ptm_bytes_for(int ncb, int nchunk) {
    return (size_t)ncb * 9 * (LDS_DOUBLES_PER_SLOT * sizeof(double) + sizeof(int)) + (size_t)ncb * 7 * sizeof(double) +
           (size_t)nchunk * (9 * sizeof(double) + sizeof(int)) + 64;
}

template <int ROT>
struct PtmEnv {
    const ProblemView& P;
    const PlanView& L;
    int comp, n, m, f0, c0, tid, nt, nwaves;
    int ncb, npb;             // camera blocks (9 LDS slots each), point blocks (a record each)
    const double2* fobs;      // listed factors' observations
    const unsigned* fidx;     // listed factors' slot word: camera block | point block << 12
    const int* gperm;         // listed factors grouped by camera block, groups padded to whole waves with -1
    int nchunk;
    const int* vptr;          // v2s_ptr + free offset
    const int* svid;          // variable id of a slot (cameras, then points)
    const int* sfree;         // local free index of a slot, -1 = constant (global copy; the cameras' also in SF)
    double *Pv, *XI, *LO, *HI, *X, *ROTR, *CG;   // LDS, cameras
    int *CGC, *SF;
    double* PT;               // [npb][12] point records
    const int* pptr;          // [npb + 1] a point's factors ...
    const int* pcam;          // ... their camera block
    const double2* pobs;      // ... their observation
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

    // ---- the cameras' trial point (LDS), as in solver_lds.hpp --------------------------------------
    template <class At>
    __device__ void refresh_records(At at) {
        if constexpr (ROT == ROT_RECORDS) {
            for (int c = nt - 1 - tid; c < ncb; c += nt) {
                const int s = 9 * c;
                if (SF[s] < 0 && SF[s + 1] < 0 && SF[s + 2] < 0) continue;
                double r[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) r[k] = SF[s + k] >= 0 ? at(s + k) : Pv[s + k];
                store_rotation(r[0], r[1], r[2], ROTR + 7 * c);
            }
        }
    }
    enum : int { AT_LINE = 0, AT_START = 1 };
    template <int MODE>
    __device__ void assign_cameras(double a) {
#pragma clang fp contract(off)
        const double* xs = L.xstart + f0;
        auto at = [&](int s) {
#pragma clang fp contract(off)
            if constexpr (MODE == AT_START) return clampd(xs[SF[s]], LO[s], HI[s]);
            else { const double t = a * XI[s]; return clampd(Pv[s] + t, LO[s], HI[s]); }
        };
        for (int s = tid; s < 9 * ncb; s += nt)
            if (SF[s] >= 0) X[s] = at(s);
        refresh_records(at);
        __syncthreads();
    }

    // ---- a point's trial values: clamp(p + a xi) from its record (registers only) -------------------
    template <int MODE>
    __device__ __forceinline__ void point_at(int ps, double a, double (&x)[3], double (&d)[3]) {
#pragma clang fp contract(off)
        const double2* r = reinterpret_cast<const double2*>(PT + (long long)PT_REC * ps);
        const double2 r0 = r[0], r1 = r[1], r2 = r[2], r3 = r[3], r4 = r[4], r5 = r[5];
        const double p[3] = {r0.x, r0.y, r1.x}, xi[3] = {r1.y, r2.x, r2.y}, lo[3] = {r3.x, r3.y, r4.x}, hi[3] = {r4.y, r5.x, r5.y};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            d[k] = xi[k];
            if constexpr (MODE == AT_START) {
                const int fi = sfree[9 * ncb + 3 * ps + k];
                x[k] = fi >= 0 ? clampd(L.xstart[f0 + fi], lo[k], hi[k]) : p[k];
            } else {
                const double t = a * xi[k];
                x[k] = clampd(p[k] + t, lo[k], hi[k]);
            }
        }
    }
    __device__ __forceinline__ double forward(int c, double2 o, const double (&x)[3], double (&v)[12], BaFwd& t) {
        const int cb = 9 * c;
#pragma unroll
        for (int k = 3; k < 9; ++k) v[k] = X[cb + k];
#pragma unroll
        for (int k = 0; k < 3; ++k) v[9 + k] = x[k];
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
    // this lane's share of the sums: its points, every point's factors
    template <bool SLOPE, int MODE>
    __device__ __forceinline__ void eval_partial(double a, double& af, double& as) {
        for (int ps = tid; ps < npb; ps += nt) {
            double x[3], dp[3];
            point_at<MODE>(ps, a, x, dp);
            const int b = pptr[ps], e = pptr[ps + 1];
            for (int q = b; q < e; ++q) {
                const int c = pcam[q];
                const double2 o = pobs[q];
                double v[12];
                BaFwd t;
                af += forward(c, o, x, v, t);
                if constexpr (SLOPE) {
                    double d[12];
#pragma unroll
                    for (int k = 0; k < 9; ++k) d[k] = (ROT == ROT_CAMFIX) ? 0.0 : XI[9 * c + k];
#pragma unroll
                    for (int k = 0; k < 3; ++k) d[9 + k] = dp[k];
                    as += ba_slope_dir<ROT == ROT_CAMFIX>(t, v, d);
                }
            }
        }
    }

    static constexpr bool UNIFORM = true;
    static constexpr int SPEC = 1;
    __device__ bool stepper() const { return threadIdx.x < 64; }
    __device__ bool writer() const { return (threadIdx.x & 63) == 0; }
    __device__ void sync() const { __syncthreads(); }
    __device__ bool tracing() const { return tr != nullptr; }
    __device__ bool aborted() const { return false; }
    __device__ void tick(int, long long) {}
    __device__ long long clock() const { return 0; }
    __device__ double eval_value(double a, bool restore) {
        double af = 0.0, as = 0.0, dummy = 0.0;
        if (restore) { assign_cameras<AT_START>(0.0); eval_partial<false, AT_START>(0.0, af, as); }
        else { assign_cameras<AT_LINE>(a); eval_partial<false, AT_LINE>(a, af, as); }
        sumk<1>(af, as, dummy);
        return af;
    }
    __device__ void eval_value_slope(double a, double& f, double& s) {
        double af = 0.0, as = 0.0, dummy = 0.0;
        assign_cameras<AT_LINE>(a);
        eval_partial<true, AT_LINE>(a, af, as);
        sumk<2>(af, as, dummy);
        f = af; s = as;
    }

    __device__ void init_vectors() {   // CGD .cpp:34-39: p = x0 (unclamped); constants hold their assigned value
        const double* xs = L.xstart + f0;
        for (int s = tid; s < 9 * ncb; s += nt) {
            const int fi = SF[s], v = svid[s];
            if (fi >= 0) {
                const double lo = P.lo[v], hi = P.hi[v], x0 = xs[fi];
                Pv[s] = x0; LO[s] = lo; HI[s] = hi; X[s] = clampd(x0, lo, hi);
            } else {
                const double xc = P.x[v];
                Pv[s] = xc; X[s] = xc; LO[s] = -__builtin_inf(); HI[s] = __builtin_inf();
            }
            XI[s] = 0.0;
        }
        for (int q = tid; q < 3 * npb; q += nt) {
            const int s = 9 * ncb + q, fi = sfree[s], v = svid[s];
            double* r = PT + (long long)PT_REC * (q / 3) + (q % 3);
            if (fi >= 0) { r[0] = xs[fi]; r[6] = P.lo[v]; r[9] = P.hi[v]; }
            else { r[0] = P.x[v]; r[6] = -__builtin_inf(); r[9] = __builtin_inf(); }
            r[3] = 0.0;
        }
        __syncthreads();
        if constexpr (ROT != ROT_PER_FACTOR) {
            for (int c = tid; c < ncb; c += nt) store_rotation(X[9 * c], X[9 * c + 1], X[9 * c + 2], ROTR + 7 * c);
            __syncthreads();
        }
    }

    // SubfunctionFD::df(p, xi) (reference .cpp:135-157): full gradient at clamp(p); see solver_lds.hpp
    __device__ void gradient_to_xi() {
        assign_cameras<AT_LINE>(0.0);
        const int lane = tid & 63;
        for (int ch = tid >> 6; ch < nchunk; ch += nwaves) {
            const int j = gperm[64 * ch + lane];
            double gq[12];
#pragma unroll
            for (int k = 0; k < 12; ++k) gq[k] = 0.0;
            unsigned w = 0u;
            if (j >= 0) {
                const int* sp = L.slot_pos + L.slot_base[c0 + j];
                const int u9 = sp[9], u10 = sp[10], u11 = sp[11];
                w = fidx[j];
                double x[3], dp[3], v[12];
                BaFwd t;
                point_at<AT_LINE>((int)(w >> 12), 0.0, x, dp);
                forward((int)(w & 0xFFFu), fobs[j], x, v, t);
                ba_adjoint(t, v, t.res0, t.res1, gq);
                if (u9 >= 0) L.gfac[u9] = gq[9];
                if (u10 >= 0) L.gfac[u10] = gq[10];
                if (u11 >= 0) L.gfac[u11] = gq[11];
            }
            if constexpr (ROT != ROT_CAMFIX) {
                const int c = __builtin_amdgcn_readfirstlane((int)(w & 0xFFFu));
                double cs[9];
#pragma unroll
                for (int k = 0; k < 9; ++k) cs[k] = wave_sum(gq[k]);
                if (lane < 9) CG[9 * ch + lane] = pick(cs, lane);
                if (lane == 0) CGC[ch] = c;
            }
        }
        __syncthreads();
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
        for (int q = tid; q < 3 * npb; q += nt) {
            const int fi = sfree[9 * ncb + q];
            if (fi < 0) continue;
            const int b = vptr[fi], e = vptr[fi + 1];
            PT[(long long)PT_REC * (q / 3) + 3 + (q % 3)] = b < e ? run_sum_ordered(L.gfac, b, e) : 0.0;
        }
        __syncthreads();
    }

    // ---- the CG recurrence: cameras in LDS, points in their records ---------------------------------
    template <class Fn>
    __device__ __forceinline__ void for_free(Fn fn) {   // fn(free index, &p, &xi)
        for (int s = tid; s < 9 * ncb; s += nt)
            if (SF[s] >= 0) fn(SF[s], Pv[s], XI[s]);
        for (int q = tid; q < 3 * npb; q += nt) {
            const int fi = sfree[9 * ncb + q];
            if (fi < 0) continue;
            double* r = PT + (long long)PT_REC * (q / 3) + (q % 3);
            fn(fi, r[0], r[3]);
        }
    }
    __device__ void cg_start() {
        for_free([&](int fi, double&, double& xi) { const double t = -xi; g[fi] = t; h[fi] = t; xi = t; });
        __syncthreads();
    }
    __device__ void line_begin() {
        if (L.vdump != nullptr && lm_count < L.dump_iters) {
            double* d = L.vdump + 2ll * L.dump_iters * f0 + 2ll * lm_count * n;
            for_free([&](int fi, double& p, double& xi) { d[fi] = p; d[n + fi] = xi; });
        }
        ++lm_count;
    }
    __device__ void line_end(double amin) {
        for_free([&](int, double& p, double& xi) {
#pragma clang fp contract(off)
            const double t = xi * amin;
            xi = t;
            p = p + t;
        });
        __syncthreads();
    }
    __device__ void cg_reduce(double fp, double& test, double& gg, double& dgg) {
        const double den = fmax(fabs(fp), 1.0);
        double a = 0.0, b = 0.0, t = 0.0;
        for_free([&](int fi, double& p, double& xi) {
#pragma clang fp contract(off)
            const double x = xi, gi = g[fi];
            t = fmax(t, fabs(x) * fmax(fabs(p), 1.0) / den);
            a = a + gi * gi;
            b = b + (x + gi) * x;
        });
        sumk<3>(a, b, t);
        gg = a; dgg = b; test = t;
    }
    __device__ void cg_update(double gam) {
        for_free([&](int fi, double&, double& xi) {
#pragma clang fp contract(off)
            const double gn = -xi;
            const double hn = gn + gam * h[fi];
            g[fi] = gn; h[fi] = hn; xi = hn;
        });
        __syncthreads();
    }
    // leave the variables assigned (.cpp:61, :84-86): clamp(p), or clamp(x_init) after the rollback
    __device__ void write_back(bool restore) {
        const double* xs = L.xstart + f0;
        for_free([&](int fi, double& p, double&) {
            // (lo, hi: from the problem -- the two storage classes need not be told apart here)
            const int v = L.free_vid[f0 + fi];
            const double xv = clampd(restore ? xs[fi] : p, P.lo[v], P.hi[v]);
            P.x[v] = xv;
            L.xout[f0 + fi] = xv;
        });
    }
};

template <int THREADS, int ROT>
__global__ void __launch_bounds__(THREADS, (THREADS <= 256 ? 2 : 1))
cgd_ptm_kernel(ProblemView P, PlanView L, int maxiters, double ftol, int ncb_cap, int chunk_cap) {
    extern __shared__ double lds_dyn[];
    __shared__ double red[2][3][MAX_WAVES];
    const int comp = L.order[blockIdx.x];
    const int f0 = L.free_ptr[comp], f1 = L.free_ptr[comp + 1];
    const int c0 = L.fac_ptr[comp], c1 = L.fac_ptr[comp + 1];
    const int n = f1 - f0, m = c1 - c0;
    // (a component without factors never gets here: it has no slot table and stays with solver_wg.hpp)
    const int s0 = L.ls_ptr[comp], ns = L.ls_ptr[comp + 1] - s0, ncb = L.ls_ncb[comp], npb = (ns - 9 * ncb) / 3;
    const int sc = 9 * ncb_cap;
    double* base = lds_dyn;
    double* CG = base + LDS_DOUBLES_PER_SLOT * sc + 7 * ncb_cap;
    int* CGC = (int*)(CG + 9 * chunk_cap);
    int* SF = CGC + chunk_cap;
    for (int s = threadIdx.x; s < 9 * ncb; s += blockDim.x) SF[s] = L.ls_free[s0 + s];
    __syncthreads();
    double* ws = L.ws + 5ll * f0;
    const int pb0 = L.pm_pt0[comp];
    PtmEnv<ROT> E{P, L, comp, n, m, f0, c0, (int)threadIdx.x, (int)blockDim.x, (int)(blockDim.x >> 6),
                  ncb, npb, L.ls_obs + c0, L.ls_fidx + c0, L.ls_gperm + 64ll * L.ls_gptr[comp], L.ls_gptr[comp + 1] - L.ls_gptr[comp],
                  L.v2s_ptr + f0, L.ls_vid + s0, L.ls_free + s0,
                  base, base + sc, base + 2 * sc, base + 3 * sc, base + 4 * sc, base + LDS_DOUBLES_PER_SLOT * sc, CG, CGC, SF,
                  L.pm_rec + (long long)PT_REC * pb0, L.pm_pptr + pb0 + L.pm_rank[comp], L.pm_cam, L.pm_obs,
                  ws + 2ll * n, ws + 3ll * n, red, 0,
                  L.trace ? L.trace + 4ll * L.trace_cap * comp : nullptr, 0, 0};

    __shared__ CgdMachine M;
    __shared__ Request Q[2];
    E.init_vectors();
    run_machine(E, M, Q, maxiters, ftol);
    E.write_back(M.rolled_back);
    if (E.tid == 0) {
        L.fret[comp] = M.fret; L.delta[comp] = M.fret - M.finit; L.iters[comp] = M.iter;
        L.status[comp] = M.status(); L.nfeval[comp] = M.nfeval; L.ngeval[comp] = M.ngeval;
        if (L.trace_n) L.trace_n[comp] = E.trn;
    }
}

// a plan's point-major factor arrays from its listed-order ones: entry e of the point-major order is listed factor jg[e]
__global__ void __launch_bounds__(256)
ptm_gather_kernel(int n, const int* __restrict__ jg, const unsigned* __restrict__ fidx, const double2* __restrict__ fobs,
                  int* __restrict__ pcam, double2* __restrict__ pobs) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int j = jg[i];
        pcam[i] = (int)(fidx[j] & 0xFFFu);
        pobs[i] = fobs[j];
    }
}

}  // namespace rdis_hip
