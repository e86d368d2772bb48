// lm_solver.hip -- Levenberg-Marquardt subspace solver for bundle adjustment on the device.
//
// What it solves is what the reference's LMSubspaceOptimizer hands to levmar
// (src/optimizers/LMSubspaceOptimizer.cpp:28-147, 176-278): one residual per factor,
// e_j = sqrt(2 E_j), Jacobian row grad E_j / e_j over the free variables, unconstrained, damping
// scale 1e-3, eps1 = eps2 = 1e-15, eps3 = SSftol, itmax = SSmaxit; the result clamped into the
// domains afterwards.  levmar itself is not vendored by the reference (parity unpinned, SURVEY.md
// 8c): the iteration is the published Levenberg-Marquardt with Nielsen's damping update that
// levmar's dlevmar_der implements, restated in oracle/lm_oracle.py, which checks this file step
// by step.
//
// How: the reference builds a dense n x m Jacobian.  Here the normal equations keep the
// camera / point block structure of bundle adjustment:
//     [ U + mu I    W        ] [dc]   [bc]       U_c = sum_j Jc_j Jc_j^T   (9x9 per camera)
//     [ W^T         V + mu I ] [dp] = [bp]       V_p = sum_j Jp_j Jp_j^T   (3x3 per point)
// and are reduced to the cameras:  (U + mu I - Z Z^T) dc = bc - Z y,  with L_p L_p^T = V_p + mu I,
// Z's block (c, p) = Jc_j (L_p^-1 Jp_j)^T for the factor j that joins camera c and point p,
// y_p = L_p^-1 bp_p;  then dp_p = L_p^-T (y_p - Z_p^T dc).
// The two contractions run on the matrix cores (v_mfma_f64_16x16x4_f64): U_c = A_c^T A_c over a
// camera's few hundred Jacobian rows, and the rank-3P update Z Z^T (9C x 9C x 3P; dense, Z stored
// k-major so that operand loads are contiguous).  Everything is summed in a fixed order.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <new>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "factors.hpp"
#include "lm_solver.hpp"

namespace rdis_hip {
namespace {

typedef double d4 __attribute__((ext_vector_type(4)));

struct Dev {            // everything the kernels need, by value
    LmProblem P;
    int nf, nca, npa, M, Mp, Kp, SK, R;       // R: residual rows per factor (1 or 2)
    const int *lf, *fci, *fpi;               // listed factor id, local camera / point block (-1: not active)
    const unsigned char* freem;              // [N]
    const int *cam_ptr, *cam_list, *pt_ptr, *pt_list;
    const int *cam_id, *pt_id;               // first variable id of each active block
    // sparse Schur product: the camera pairs (a >= b) that share a point, and per pair its (factor of a, factor of b) entries
    const int *pair_ab, *pair_ptr, *pair_ja, *pair_jb;
    int npair, sparse;
    double *Jc, *Jp, *e;                     // [nf*R*9], [nf*R*3], [nf*R]  (row jr = j*R + r)
    double *U, *bc;                          // [nca*81], [Mp]
    double *V, *bp, *Lp, *yp, *T;            // [npa*6], [npa*3], [npa*6], [npa*3], [nf*R*3]
    double *Zt;                              // [Kp][Mp]   k-major (dense Schur product)
    double *Zc;                              // [nf][3][9] a factor's Z block, column by column (block-sparse Schur product)
    double *Spart, *S, *rhs, *dc, *dp;       // [SK][Mp*Mp], [Mp*Mp], [Mp], [Mp], [npa*3]
    double *Ld;                              // [Mp/32][32][32] the factored diagonal blocks (S keeps the unfactored ones: every
                                             // workgroup of a k_panel launch reads its panel's, one writes the factor -- not over it)
    double *psave;                           // accepted point: [9*nca + 3*npa] by block slot
    double *part, *sc;                       // block partials [4096*4], scalars [16]
};

__device__ __forceinline__ double wsum(double v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ double wmax(double v) {
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
    return v;
}

// ---- 1. linearisation: residual and Jacobian row of every listed factor --------------------
__global__ void __launch_bounds__(256) k_lin(Dev D) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= D.nf) return;
    const int f = D.lf[j];
    const int c = D.P.cam[f], q = D.P.pt[f];
    double v[12], g[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) v[k] = D.P.x[k < 9 ? c + k : q + (k - 9)];
    const double2 o = D.P.obs[f];
    if (D.R == 1) {   // the reference's formulation: one residual sqrt(2 E), row grad E / e
        const double E = ba_eval_grad(v, o.x, o.y, g);
        const double e = sqrt(2.0 * E);
        const double inv = e > 0.0 ? 1.0 / e : 0.0;
        D.e[j] = e;
#pragma unroll
        for (int k = 0; k < 9; ++k) D.Jc[9ll * j + k] = D.freem[c + k] ? g[k] * inv : 0.0;
#pragma unroll
        for (int k = 0; k < 3; ++k) D.Jp[3ll * j + k] = D.freem[q + k] ? g[9 + k] * inv : 0.0;
    } else {          // the two pixel residuals and their Jacobian rows
        double res[2], g2[12];
        ba_residual_jacobian(v, o.x, o.y, res, g, g2);
        D.e[2ll * j] = res[0]; D.e[2ll * j + 1] = res[1];
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const bool fr = D.freem[c + k];
            D.Jc[18ll * j + k] = fr ? g[k] : 0.0; D.Jc[18ll * j + 9 + k] = fr ? g2[k] : 0.0;
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const bool fr = D.freem[q + k];
            D.Jp[6ll * j + k] = fr ? g[9 + k] : 0.0; D.Jp[6ll * j + 3 + k] = fr ? g2[9 + k] : 0.0;
        }
    }
}

// objective of the listed factors at the current x: block partials of sum e^2 / 2
__global__ void __launch_bounds__(256) k_obj(Dev D) {
    __shared__ double red[4];
    double acc = 0.0;
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < D.nf; j += gridDim.x * blockDim.x) {
        const int f = D.lf[j];
        const int c = D.P.cam[f], q = D.P.pt[f];
        double v[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) v[k] = D.P.x[k < 9 ? c + k : q + (k - 9)];
        const double2 o = D.P.obs[f];
        acc += ba_eval(v, o.x, o.y);
    }
    acc = wsum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) D.part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ void __launch_bounds__(256) k_obj_final(Dev D, int nblocks, int slot) {
    __shared__ double red[4];
    double acc = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += 256) acc += D.part[i];
    acc = wsum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) D.sc[slot] = (red[0] + red[1]) + (red[2] + red[3]);
}

// ---- 2. camera blocks on the matrix cores: U_c = A_c^T A_c, bc = -A_c^T e ---------------------
// one workgroup of sixteen waves per active camera: wave w takes the Jacobian rows 4w .. 4w+3 of
// every group of 64 (a camera of ladybug has 700-1800 rows: one wave alone is a chain of that many
// dependent load -> MFMA steps), the sixteen partial blocks are added in wave order.
// v_mfma_f64_16x16x4_f64: lane l supplies A[i = l & 15][k = l >> 4] and
// B[k = l >> 4][n = l & 15]; with A = B^T = (rows of Jc)^T both are the same number.  Result
// register r of lane l is D[row = (l >> 4) + 4 r][col = l & 15].
constexpr int CAM_WAVES = 16;
__global__ void __launch_bounds__(64 * CAM_WAVES) k_cam(Dev D) {
    __shared__ double part[CAM_WAVES][5][64];
    const int c = blockIdx.x, l = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int b0 = D.cam_ptr[c], cnt = (D.cam_ptr[c + 1] - b0) * D.R;   // Jacobian rows of this camera
    const int i = l & 15, kk = l >> 4;
    d4 acc = {0.0, 0.0, 0.0, 0.0};
    double bsum = 0.0;
    for (int j0 = 4 * w; j0 < cnt; j0 += 4 * CAM_WAVES) {
        double a = 0.0, ev = 0.0;
        if (j0 + kk < cnt && i < 9) {
            const int rr = j0 + kk;
            const long long jr = (long long)D.cam_list[b0 + rr / D.R] * D.R + rr % D.R;
            a = D.Jc[9 * jr + i];
            ev = D.e[jr];
        }
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, acc, 0, 0, 0);
        bsum += a * ev;
    }
    // bc: lanes i, i+16, i+32, i+48 hold the four k-phases of row i
    bsum += __shfl_xor(bsum, 16);
    bsum += __shfl_xor(bsum, 32);
#pragma unroll
    for (int r = 0; r < 4; ++r) part[w][r][l] = acc[r];
    part[w][4][l] = bsum;
    __syncthreads();
    if (w != 0) return;
    double t[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    for (int q = 0; q < CAM_WAVES; ++q)
#pragma unroll
        for (int r = 0; r < 5; ++r) t[r] += part[q][r][l];
    if (l < 9) D.bc[9 * c + l] = -t[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = kk + 4 * r, col = i;
        if (row < 9 && col < 9) D.U[81ll * c + 9 * row + col] = t[r];
    }
}

// ---- 3. point blocks: V_p (lower triangle, 6 numbers), bp ------------------------------------
// (sixteen lanes per point, like k_back)
__global__ void __launch_bounds__(256) k_pt(Dev D) {
    const int p = (blockIdx.x * blockDim.x + threadIdx.x) >> 4, sub = threadIdx.x & 15;
    if (p >= D.npa) return;
    double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};   // v00 v10 v11 v20 v21 v22 b0 b1 b2
    const int t0 = D.pt_ptr[p], nrows = (D.pt_ptr[p + 1] - t0) * D.R;
    for (int rr = sub; rr < nrows; rr += 16) {
        const long long jr = (long long)D.pt_list[t0 + rr / D.R] * D.R + rr % D.R;
        const double a = D.Jp[3 * jr], b = D.Jp[3 * jr + 1], c = D.Jp[3 * jr + 2], ev = D.e[jr];
        acc[0] += a * a; acc[1] += b * a; acc[2] += b * b; acc[3] += c * a; acc[4] += c * b; acc[5] += c * c;
        acc[6] += a * ev; acc[7] += b * ev; acc[8] += c * ev;
    }
#pragma unroll
    for (int k = 0; k < 9; ++k)
        for (int o = 8; o > 0; o >>= 1) acc[k] += __shfl_xor(acc[k], o);
    if (sub != 0) return;
    double* V = D.V + 6ll * p;
#pragma unroll
    for (int k = 0; k < 6; ++k) V[k] = acc[k];
    D.bp[3ll * p] = -acc[6]; D.bp[3ll * p + 1] = -acc[7]; D.bp[3ll * p + 2] = -acc[8];
}

// ---- 4. scalars: max diag(J^T J), |J^T e|_inf, |p|^2 (one block) --------------------------------
// several blocks; the block that finishes last combines the blocks' partials in block order
constexpr int SCALARS_MAX_BLOCKS = 256, SC_TICKET2 = 13;
__global__ void __launch_bounds__(1024) k_scalars(Dev D) {
    __shared__ double r0[16], r1[16], r2[16];
    double md = 0.0, mb = 0.0, pl = 0.0;
    const int g0 = blockIdx.x * 1024 + threadIdx.x, gs = gridDim.x * 1024;
    for (int t = g0; t < 9 * D.nca; t += gs) {
        const int c = t / 9, k = t - 9 * c;
        md = fmax(md, D.U[81ll * c + 10 * k]);
        mb = fmax(mb, fabs(D.bc[t]));
        const int v = D.cam_id[c] + k;
        if (D.freem[v]) { const double x = D.P.x[v]; pl += x * x; }
    }
    for (int t = g0; t < 3 * D.npa; t += gs) {
        const int p = t / 3, k = t - 3 * p;
        md = fmax(md, D.V[6ll * p + (k == 0 ? 0 : k == 1 ? 2 : 5)]);
        mb = fmax(mb, fabs(D.bp[t]));
        const int v = D.pt_id[p] + k;
        if (D.freem[v]) { const double x = D.P.x[v]; pl += x * x; }
    }
    md = wmax(md); mb = wmax(mb); pl = wsum(pl);
    if ((threadIdx.x & 63) == 0) { r0[threadIdx.x >> 6] = md; r1[threadIdx.x >> 6] = mb; r2[threadIdx.x >> 6] = pl; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0.0, b = 0.0, s = 0.0;
        for (int w = 0; w < 16; ++w) { a = fmax(a, r0[w]); b = fmax(b, r1[w]); s += r2[w]; }
        double* part = D.part + 3072;   // (k_obj: [0, 2048), k_apply: [2048, 3072))
        unsigned* ticket = reinterpret_cast<unsigned*>(D.sc + SC_TICKET2);
        part[3 * blockIdx.x] = a; part[3 * blockIdx.x + 1] = b; part[3 * blockIdx.x + 2] = s;
        __threadfence();
        if (atomicAdd(ticket, 1u) == gridDim.x - 1) {
            __threadfence();
            a = 0.0; b = 0.0; s = 0.0;
            for (unsigned q = 0; q < gridDim.x; ++q) {
                a = fmax(a, __hip_atomic_load(part + 3 * q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                b = fmax(b, __hip_atomic_load(part + 3 * q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                s += __hip_atomic_load(part + 3 * q + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            D.sc[0] = a; D.sc[1] = b; D.sc[2] = s;
            *ticket = 0u;
        }
    }
}

// ---- 5. per damping value: point Cholesky, T_j = L_p^-1 Jp_j, Z --------------------------------
// damping added to a diagonal entry d of J^T J: mu (Levenberg, what levmar does: model 1) or
// mu * max(d, floor) (Marquardt's scaling by the diagonal: model 2)
// the damping of the current attempt: written by the host into the device scalars before every solve
constexpr int SC_MU = 10, SC_FLOOR = 11;
__device__ __forceinline__ double damp(const Dev& D, double mu, double floor_, double d) {
    return D.R == 2 ? mu * fmax(d, floor_) : mu;
}
__global__ void __launch_bounds__(256) k_ptchol(Dev D) {
    const double mu = D.sc[SC_MU], floor_ = D.sc[SC_FLOOR];
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= D.npa) return;
    const double* V = D.V + 6ll * p;
    const double l00 = sqrt(V[0] + damp(D, mu, floor_, V[0]));
    const double l10 = V[1] / l00, l20 = V[3] / l00;
    const double l11 = sqrt(V[2] + damp(D, mu, floor_, V[2]) - l10 * l10);
    const double l21 = (V[4] - l20 * l10) / l11;
    const double l22 = sqrt(V[5] + damp(D, mu, floor_, V[5]) - l20 * l20 - l21 * l21);
    double* L = D.Lp + 6ll * p;
    L[0] = l00; L[1] = l10; L[2] = l11; L[3] = l20; L[4] = l21; L[5] = l22;
    const double y0 = D.bp[3ll * p] / l00;
    const double y1 = (D.bp[3ll * p + 1] - l10 * y0) / l11;
    const double y2 = (D.bp[3ll * p + 2] - l20 * y0 - l21 * y1) / l22;
    D.yp[3ll * p] = y0; D.yp[3ll * p + 1] = y1; D.yp[3ll * p + 2] = y2;
}
__global__ void __launch_bounds__(256) k_z(Dev D) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= D.nf) return;
    const int ci = D.fci[j], pi = D.fpi[j];
    if (pi < 0) return;
    const double* L = D.Lp + 6ll * pi;
    double t0[2] = {0, 0}, t1[2] = {0, 0}, t2[2] = {0, 0};
    for (int r = 0; r < D.R; ++r) {
        const long long jr = (long long)j * D.R + r;
        t0[r] = D.Jp[3 * jr] / L[0];
        t1[r] = (D.Jp[3 * jr + 1] - L[1] * t0[r]) / L[2];
        t2[r] = (D.Jp[3 * jr + 2] - L[3] * t0[r] - L[4] * t1[r]) / L[5];
        D.T[3 * jr] = t0[r]; D.T[3 * jr + 1] = t1[r]; D.T[3 * jr + 2] = t2[r];
    }
    if (ci < 0) return;
    if (D.sparse) {   // the block itself, 27 numbers: what k_schur's operands are
        double* z = D.Zc + 27ll * j;
#pragma unroll
        for (int a = 0; a < 9; ++a) {
            const double jc0 = D.Jc[9ll * j * D.R + a], jc1 = D.R == 2 ? D.Jc[9ll * (2 * j + 1) + a] : 0.0;
            z[a] = jc0 * t0[0] + jc1 * t0[1]; z[9 + a] = jc0 * t1[0] + jc1 * t1[1]; z[18 + a] = jc0 * t2[0] + jc1 * t2[1];
        }
        return;
    }
#pragma unroll
    for (int a = 0; a < 9; ++a) {
        const double jc0 = D.Jc[9ll * j * D.R + a], jc1 = D.R == 2 ? D.Jc[9ll * (2 * j + 1) + a] : 0.0;
        double* z = D.Zt + (long long)(3 * pi) * D.Mp + 9 * ci + a;
        z[0] = jc0 * t0[0] + jc1 * t0[1]; z[D.Mp] = jc0 * t1[0] + jc1 * t1[1]; z[2ll * D.Mp] = jc0 * t2[0] + jc1 * t2[1];
    }
}

// ---- 6. Z Z^T on the matrix cores ----------------------------------------------------------------
// grid (tile pairs, K slices); one wave per 64 x 64 tile of the product (sixteen accumulators:
// eight operand loads feed sixteen MFMAs) and slice of K; partial
// products per slice, summed in slice order by k_sfinish.
__global__ void __launch_bounds__(64) k_syrk(Dev D) {
    const int nt = D.Mp / 64;
    const int ti = blockIdx.x / nt, tj = blockIdx.x % nt;
    if (tj > ti) return;                       // lower triangle of tiles; mirrored by k_sfinish
    const int l = threadIdx.x, i = l & 15, kk = l >> 4;
    const int kslice = (D.Kp / 4 + D.SK - 1) / D.SK * 4;
    const int k0 = blockIdx.y * kslice, k1 = min(D.Kp, k0 + kslice);
    d4 c[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) c[a][b] = d4{0, 0, 0, 0};
    const double* za = D.Zt + 64 * ti + i;
    const double* zb = D.Zt + 64 * tj + i;
    for (int k = k0; k < k1; k += 4) {
        const long long off = (long long)(k + kk) * D.Mp;
        double av[4], bv[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) { av[a] = za[off + 16 * a]; bv[a] = zb[off + 16 * a]; }
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) c[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[a], bv[b], c[a][b], 0, 0, 0);
    }
    double* out = D.Spart + (long long)blockIdx.y * D.Mp * D.Mp;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                out[(long long)(64 * ti + 16 * a + kk + 4 * r) * D.Mp + 64 * tj + 16 * b + i] = c[a][b][r];
}
// S = U + mu I - sum over slices (lower triangle; padding rows get a unit diagonal); rhs = bc - Z y
__global__ void __launch_bounds__(256) k_sfinish(Dev D) {
    const double mu = D.sc[SC_MU], floor_ = D.sc[SC_FLOOR];
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)D.Mp * D.Mp) return;
    const int row = (int)(t / D.Mp), col = (int)(t % D.Mp);
    if (col > row) return;
    double s = 0.0;
    for (int sl = 0; sl < D.SK; ++sl) s += D.Spart[(long long)sl * D.Mp * D.Mp + t];
    double v = -s;
    if (row < D.M) {
        if (row / 9 == col / 9) v += D.U[81ll * (row / 9) + 9 * (row % 9) + (col % 9)];
        if (row == col) v += damp(D, mu, floor_, D.U[81ll * (row / 9) + 10 * (row % 9)]);
    } else {
        v = row == col ? 1.0 : 0.0;
    }
    D.S[t] = v;
}
// ---- 6b. the same product where Z is what it is: block-sparse ------------------------------------------------------
// Z's block (c, p) is non-zero only where camera c sees point p, so block (a, b) of Z Z^T is a sum over the points
// both cameras see: sum_p Z_(a,p) Z_(b,p)^T with Z_(c,p) = Jc_j^T T_j (9 x R times R x 3) for the factor j that joins c
// and p.  On a BAL problem a point is seen by a handful of the cameras (ladybug: 4.1 of 49): the dense rank-3P update
// above multiplies zeros 98 % of the time (4.5 GFLOP against 0.06), and writing, zeroing and re-reading the dense Z (82 MB)
// is most of what a damped solve costs besides the Cholesky chain.  Here: the host lists, once per call, the camera pairs
// (a >= b) that share a point with their (factor of a, factor of b) entries in point order; one workgroup per pair owns the
// 9 x 9 block and subtracts the sum from S, which k_sinit has set to U + damping (block diagonal) and zero elsewhere.
// No atomics, a fixed order of summation.  (First form: a wave per pair, a lane per entry of the block, looping over the
// pair's factors -- a chain of dependent gathers, 906 links long for a camera with itself: slower than the dense product.)  The dense path stays for
// problems where most cameras see most points (chosen by fill; `model` bits 4-5 of rdis_hip_lm_optimize force one).
__global__ void __launch_bounds__(256) k_sinit(Dev D) {
    const double mu = D.sc[SC_MU], floor_ = D.sc[SC_FLOOR];
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)D.Mp * D.Mp) return;
    const int row = (int)(t / D.Mp), col = (int)(t % D.Mp);
    if (col > row) return;
    double v = 0.0;
    if (row < D.M) {
        if (row / 9 == col / 9) v = D.U[81ll * (row / 9) + 9 * (row % 9) + (col % 9)];
        if (row == col) v += damp(D, mu, floor_, D.U[81ll * (row / 9) + 10 * (row % 9)]);
    } else {
        v = row == col ? 1.0 : 0.0;
    }
    D.S[t] = v;
}
// One workgroup of sixteen waves per camera pair; the block is a small product on the matrix cores (v_mfma_f64_16x16x4_f64
// as in k_cam, with two different operands): D[row][col] = sum_k A[row][k] B[k][col], k running over (entry, t) -- the three
// columns of every shared point's Z blocks --, lane l supplying A[i = l & 15][k = l >> 4] = Z_(a,p)[i][t] and
// B[k][n = l & 15] = Z_(b,p)[n][t], read from the factors' 27-number Z blocks (k_z; 7 MB for ladybug where the dense Z is 82).
// Wave w takes the steps w, w + 16, ... (four k each), four at a time so that their gathers are in flight together;
// the sixteen partial blocks are added in wave order.
constexpr int SCH_WAVES = 16;
__global__ void __launch_bounds__(64 * SCH_WAVES) k_schur(Dev D) {
    __shared__ double part[SCH_WAVES][4][64];
    const int q = blockIdx.x, l = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int a = D.pair_ab[2 * q], b = D.pair_ab[2 * q + 1];
    const int e0 = D.pair_ptr[q], nk = 3 * (D.pair_ptr[q + 1] - e0);
    const int i = l & 15, kk = l >> 4;
    auto operands = [&](int step, double& av, double& bv) {
        av = 0.0; bv = 0.0;
        const int k = 4 * step + kk;
        if (k < nk && i < 9) {
            const int e = e0 + k / 3, t = k % 3;
            av = D.Zc[27ll * D.pair_ja[e] + 9 * t + i];
            bv = D.Zc[27ll * D.pair_jb[e] + 9 * t + i];
        }
    };
    d4 acc = {0.0, 0.0, 0.0, 0.0};
    const int nsteps = (nk + 3) / 4;
    for (int s = w; s < nsteps; s += 4 * SCH_WAVES) {   // four steps' loads in flight (beyond the end: zeros)
        double av[4], bv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) operands(s + u * SCH_WAVES, av[u], bv[u]);
#pragma unroll
        for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[u], acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) part[w][r][l] = acc[r];
    __syncthreads();
    if (w != 0) return;
    double t[4] = {0.0, 0.0, 0.0, 0.0};
    for (int v = 0; v < SCH_WAVES; ++v)
#pragma unroll
        for (int r = 0; r < 4; ++r) t[r] += part[v][r][l];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = kk + 4 * r, col = i;
        if (row < 9 && col < 9) {
            const int gr = 9 * a + row, gc = 9 * b + col;
            if (gc <= gr) D.S[(long long)gr * D.Mp + gc] -= t[r];   // lower triangle (a >= b; within a diagonal block col <= row)
        }
    }
}

// rhs = bc - Z y, one wave per active camera over its factor list: Z's block of factor j times y_p
// is Jc_j (T_j . y_p)
constexpr int RHS_WAVES = 8;   // (a camera of ladybug has 361-906 factors: eight waves share them)
__global__ void __launch_bounds__(64 * RHS_WAVES) k_rhs(Dev D) {
    __shared__ double part[RHS_WAVES][9];
    const int c = blockIdx.x, l = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (c == 0) for (int t = D.M + (int)threadIdx.x; t < D.Mp; t += 64 * RHS_WAVES) D.dc[t] = 0.0;   // the padding, whatever a failed solve left there
    double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int t = D.cam_ptr[c] + (int)threadIdx.x; t < D.cam_ptr[c + 1]; t += 64 * RHS_WAVES) {
        const int j = D.cam_list[t], pi = D.fpi[j];
        if (pi < 0) continue;
        for (int r = 0; r < D.R; ++r) {
            const long long jr = (long long)j * D.R + r;
            const double wv = D.T[3 * jr] * D.yp[3ll * pi] + D.T[3 * jr + 1] * D.yp[3ll * pi + 1] + D.T[3 * jr + 2] * D.yp[3ll * pi + 2];
#pragma unroll
            for (int a = 0; a < 9; ++a) acc[a] += D.Jc[9 * jr + a] * wv;
        }
    }
#pragma unroll
    for (int a = 0; a < 9; ++a) {
        const double s = wsum(acc[a]);
        if (l == 0) part[w][a] = s;
    }
    __syncthreads();
    if (threadIdx.x < 9) {
        double s = 0.0;
        for (int q = 0; q < RHS_WAVES; ++q) s += part[q][threadIdx.x];
        const double v = D.bc[9 * c + threadIdx.x] - s;
        D.rhs[9 * c + threadIdx.x] = v; D.dc[9 * c + threadIdx.x] = v;   // dc: the row the factorisation carries along
    }
}

// ---- 7. dense Cholesky of the reduced system, blocked (panel width 32), and the solves ----------
// The padded matrix (order Mp, a multiple of 32, unit diagonal in the padding) is factored in
// place, right-looking, two launches per panel:
//   k_panel  every workgroup factors the 32 x 32 diagonal block inside one wave (a row per lane,
//            in registers, columns broadcast with v_readlane: no LDS round trip and no barrier in
//            the 32 dependent steps), then its rows below take X = A L^-T, one lane per row (measured and dropped:
//            the rows moved row-wise through LDS, 32 lanes a row, and loaded while the block is factored --
//            17.9 against 15.1 us a launch: the in-wave factorisation, not the rows' traffic, is what a launch costs);
//            the factored block goes to Ld, not over S's: the other workgroups of the launch still read that;
//   k_trail  trailing matrix -= X X^T on the matrix cores.
// The right-hand side rides along as one more row (kept in dc): when the last panel is done it
// holds z = L^-1 rhs, so only the backward substitution L^T dc = z is left for k_trsv.
__device__ __forceinline__ double bcast_lane(double v, int lane) {   // lane: a compile-time constant after unrolling
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
// lane i (< 32) holds row i of the block's lower triangle in r[0..i]; returns false on a pivot <= 0
__device__ __forceinline__ bool potf2_wave(double (&r)[32], int i) {
    bool ok = true;
#pragma unroll
    for (int k = 0; k < 32; ++k) {
        const double dkk = bcast_lane(r[k], k);
        ok = ok && dkk > 0.0;
        // 1/sqrt(d) from the hardware estimate and two Newton steps (the square root and the division
        // it replaces are a third of this loop's dependent chain); l_kk = d * rsqrt(d)
        double y = __builtin_amdgcn_rsq(dkk);
        y = y * (1.5 - 0.5 * dkk * y * y);
        y = y * (1.5 - 0.5 * dkk * y * y);
        r[k] = i == k ? dkk * y : r[k] * y;        // (lanes above the diagonal carry values nobody reads)
#pragma unroll
        for (int c = k + 1; c < 32; ++c) r[c] -= r[k] * bcast_lane(r[k], c);
    }
    return ok;
}
__global__ void __launch_bounds__(256) k_panel(Dev D, int kb) {
    __shared__ double L[32][33];
    __shared__ double rd[32];
    const int ld = D.Mp;
    if (threadIdx.x < 64) {
        const int i = threadIdx.x & 31;
        double r[32];
        const double* src = D.S + (long long)(kb + i) * ld + kb;
#pragma unroll
        for (int c = 0; c < 32; ++c) r[c] = c <= i ? src[c] : 0.0;
        const bool ok = potf2_wave(r, i);
        if (threadIdx.x < 32) {
            double dii = 1.0;
#pragma unroll
            for (int c = 0; c < 32; ++c) {
                L[i][c] = c <= i ? r[c] : 0.0;
                if (c == i) dii = r[c];
            }
            rd[i] = 1.0 / dii;
            if (blockIdx.x == 0) {
                double* dst = D.Ld + 32ll * kb + 32 * i;   // (block kb / 32, row i)
#pragma unroll
                for (int c = 0; c < 32; ++c) if (c <= i) dst[c] = r[c];
                if (!ok) D.sc[8] = 1.0;
            }
        }
    }
    __syncthreads();
    const int row = kb + 32 + blockIdx.x * 256 + threadIdx.x;   // row Mp: the right-hand side
    if (row > D.Mp) return;
    double* a = row < D.Mp ? D.S + (long long)row * ld + kb : D.dc + kb;
    double x[32];
#pragma unroll
    for (int c = 0; c < 32; ++c) x[c] = a[c];
#pragma unroll
    for (int c = 0; c < 32; ++c) {
        double v = x[c];
#pragma unroll
        for (int t = 0; t < c; ++t) v -= x[t] * L[c][t];
        x[c] = v * rd[c];
    }
#pragma unroll
    for (int c = 0; c < 32; ++c) a[c] = x[c];
}
__global__ void __launch_bounds__(64) k_trail(Dev D, int kb) {
    const int r0 = kb + 32, nt = (D.Mp - r0) / 32;
    const int l = threadIdx.x, ld = D.Mp;
    if ((int)blockIdx.x >= nt * nt) {   // the right-hand-side row: z[r0 ..] -= X z[kb .. kb + 32)
        if (l < 32) {
            const int row = r0 + 32 * ((int)blockIdx.x - nt * nt) + l;
            const double* xr = D.S + (long long)row * ld + kb;
            double acc = 0.0;
#pragma unroll
            for (int c = 0; c < 32; ++c) acc += xr[c] * D.dc[kb + c];
            D.dc[row] -= acc;
        }
        return;
    }
    const int ti = blockIdx.x / nt, tj = blockIdx.x % nt;
    if (tj > ti) return;
    const int i = l & 15, kk = l >> 4;
    d4 c00 = {0, 0, 0, 0}, c01 = {0, 0, 0, 0}, c10 = {0, 0, 0, 0}, c11 = {0, 0, 0, 0};
    const double* xa = D.S + (long long)(r0 + 32 * ti + i) * ld + kb + kk;
    const double* xb = D.S + (long long)(r0 + 32 * tj + i) * ld + kb + kk;
#pragma unroll
    for (int k = 0; k < 32; k += 4) {
        const double a0 = xa[k], a1 = xa[16ll * ld + k], b0 = xb[k], b1 = xb[16ll * ld + k];
        c00 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, c00, 0, 0, 0);
        c01 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, c01, 0, 0, 0);
        c10 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, c10, 0, 0, 0);
        c11 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, c11, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = kk + 4 * r, col = i;
        double* o = D.S + (long long)(r0 + 32 * ti + row) * ld + r0 + 32 * tj + col;
        o[0] -= c00[r];
        o[16] -= c01[r];
        o[16ll * ld] -= c10[r];
        o[16ll * ld + 16] -= c11[r];
    }
}
// L^T dc = z (z = L^-1 rhs is already in dc, see above), one workgroup, blocks of 32 from the bottom.  Right-looking:
// z lives in LDS; per block one wave solves the 32 x 32 triangle (values broadcast with v_readlane) and everybody
// then takes the block's contribution off the entries above it -- thread j holds rows kb .. kb + 31 of column j,
// loaded BEFORE the triangle is solved (they do not depend on it), so a block costs the triangle, two barriers and
// 32 multiply-adds instead of a chain of dependent round trips to the L2; the diagonal element is selected first and
// inverted once (the select-after-divide form cost a division per column).  ladybug's 448 unknowns: 82 -> 51 us.
// (Measured and dropped: a pass over the whole factor first, to bring it into this unit's caches: +5 us, no gain.)
constexpr int TRSV_MAX = 4096;    // unknowns this form keeps in LDS
constexpr int TRSV_THREADS = 512;
__global__ void __launch_bounds__(TRSV_THREADS) k_trsv(Dev D) {
    __shared__ double zs[TRSV_MAX];
    __shared__ double xb[32];
    const int n = D.Mp, ld = D.Mp, tid = threadIdx.x, i = tid & 31;
    const double* S = D.S;
    if (D.sc[8] != 0.0) return;
    for (int r = tid; r < n; r += TRSV_THREADS) zs[r] = D.dc[r];
    __syncthreads();
    double col[32];   // wave 0: this lane's column of the diagonal block -- the NEXT block's is loaded while this one is solved
    if (tid < 64) {
#pragma unroll
        for (int c = 0; c < 32; ++c) col[c] = c >= i ? D.Ld[32ll * (n - 32) + 32 * c + i] : 0.0;
    }
    for (int kb = n - 32; kb >= 0; kb -= 32) {
        double a[32], coln[32];
        if (tid < kb) {
#pragma unroll
            for (int c = 0; c < 32; ++c) a[c] = S[(long long)(kb + c) * ld + tid];
        }
        if (tid < 64) {
            if (kb >= 32) {
#pragma unroll
                for (int c = 0; c < 32; ++c) coln[c] = c >= i ? D.Ld[32ll * (kb - 32) + 32 * c + i] : 0.0;
            }
            double zi = zs[kb + i], dii = 1.0;
#pragma unroll
            for (int c = 0; c < 32; ++c) if (c == i) dii = col[c];   // (select, then ONE division: not one per column)
            const double dinv = 1.0 / dii;
#pragma unroll
            for (int c = 31; c >= 0; --c) {
                const double xc = bcast_lane(zi * dinv, c);
                if (i == c) zi = xc;
                else if (i < c) zi -= col[c] * xc;
            }
            if (tid < 32) { zs[kb + i] = zi; xb[i] = zi; }
            if (kb >= 32) {
#pragma unroll
                for (int c = 0; c < 32; ++c) col[c] = coln[c];
            }
        }
        __syncthreads();
        if (tid < kb) {
            double acc = 0.0;
#pragma unroll
            for (int c = 0; c < 32; ++c) acc += a[c] * xb[c];
            zs[tid] -= acc;
        }
        for (int j = tid + TRSV_THREADS; j < kb; j += TRSV_THREADS) {   // (more unknowns than threads)
            double acc = 0.0;
#pragma unroll 8
            for (int c = 0; c < 32; ++c) acc += S[(long long)(kb + c) * ld + j] * xb[c];
            zs[j] -= acc;
        }
        __syncthreads();
    }
    for (int r = tid; r < n; r += TRSV_THREADS) D.dc[r] = zs[r];
}
// The same solve for systems beyond TRSV_MAX unknowns, in place in global memory: one workgroup, left-looking by blocks of 32
// from the bottom: the 32 x 32 lane grid first sums what the rows already solved contribute to the
// block (lane column i, row group g; S is read once, coalesced), then one wave finishes the sums
// and solves the 32 x 32 triangle (values broadcast with v_readlane).
__global__ void __launch_bounds__(1024) k_trsv_left(Dev D) {
    __shared__ double red[32][33];
    const int n = D.Mp, ld = D.Mp, tid = threadIdx.x, i = tid & 31, g = tid >> 5;
    const double* S = D.S;
    double* z = D.dc;
    if (D.sc[8] != 0.0) return;
    for (int kb = n - 32; kb >= 0; kb -= 32) {
        double col[32];   // wave 0: this lane's column of the diagonal block, in flight during the sums
        if (tid < 64) {
#pragma unroll
            for (int c = 0; c < 32; ++c) col[c] = c >= i ? D.Ld[32ll * kb + 32 * c + i] : 0.0;
        }
        double acc = 0.0;
        for (int r = kb + 32 + g; r < n; r += 32) acc += S[(long long)r * ld + kb + i] * z[r];
        red[g][i] = acc;
        __syncthreads();
        if (tid < 64) {
            double sum = 0.0;
#pragma unroll
            for (int q = 0; q < 32; ++q) sum += red[q][i];
            double zi = z[kb + i] - sum, dii = 1.0;
#pragma unroll
            for (int c = 0; c < 32; ++c) if (c == i) dii = col[c];
            const double dinv = 1.0 / dii;
#pragma unroll
            for (int c = 31; c >= 0; --c) {
                const double xc = bcast_lane(zi * dinv, c);
                if (i == c) zi = xc;
                else if (i < c) zi -= col[c] * xc;
            }
            if (tid < 32) z[kb + i] = zi;
        }
        __syncthreads();
    }
}

// ---- 8. back-substitution for the points ------------------------------------------------------
// sixteen lanes per point: a point of ladybug has up to 29 factors (58 Jacobian rows), one lane each
__global__ void __launch_bounds__(256) k_back(Dev D) {
    const int p = (blockIdx.x * blockDim.x + threadIdx.x) >> 4, sub = threadIdx.x & 15;
    if (p >= D.npa) return;   // (whole groups of sixteen leave together)
    double s0 = 0, s1 = 0, s2 = 0;
    const int t0 = D.pt_ptr[p], nrows = (D.pt_ptr[p + 1] - t0) * D.R;
    for (int rr = sub; rr < nrows; rr += 16) {
        const int j = D.pt_list[t0 + rr / D.R];
        const int ci = D.fci[j];
        if (ci < 0) continue;
        const long long jr = (long long)j * D.R + rr % D.R;
        double w = 0.0;
#pragma unroll
        for (int a = 0; a < 9; ++a) w += D.Jc[9 * jr + a] * D.dc[9 * ci + a];
        s0 += D.T[3 * jr] * w; s1 += D.T[3 * jr + 1] * w; s2 += D.T[3 * jr + 2] * w;
    }
    for (int o = 8; o > 0; o >>= 1) { s0 += __shfl_xor(s0, o); s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
    if (sub != 0) return;
    const double* L = D.Lp + 6ll * p;
    const double r0 = D.yp[3ll * p] - s0, r1 = D.yp[3ll * p + 1] - s1, r2 = D.yp[3ll * p + 2] - s2;
    const double d2 = r2 / L[5];
    const double d1 = (r1 - L[4] * d2) / L[2];
    const double d0 = (r0 - L[1] * d1 - L[3] * d2) / L[0];
    D.dp[3ll * p] = d0; D.dp[3ll * p + 1] = d1; D.dp[3ll * p + 2] = d2;
}

// ---- 9. trial point, |Dp|^2, dL = Dp . (mu Dp + J^T e) -------------------------------------------
// mode 0: x = psave + Dp (model 2: clamped into the domain);  mode 1: psave = x (accept);  mode 2: x = psave (restore);
// mode 3: x = clamp(psave) (final, LMSubspaceOptimizer.cpp:104-108).  Several blocks; for mode 0 the
// block that finishes last adds the blocks' partial sums in block order.
constexpr int APPLY_MAX_BLOCKS = 512, SC_TICKET = 12;
__global__ void __launch_bounds__(1024) k_apply(Dev D, int mode) {
    const double mu = D.sc[SC_MU], floor_ = D.sc[SC_FLOOR];
    __shared__ double r0[16], r1[16];
    __shared__ bool last;
    double dl2 = 0.0, dL = 0.0;
    const int nc9 = 9 * D.nca, ntot = nc9 + 3 * D.npa;
    for (int t = blockIdx.x * 1024 + threadIdx.x; t < ntot; t += gridDim.x * 1024) {
        int v; double dpv, b, dg;
        if (t < nc9) { v = D.cam_id[t / 9] + t % 9; dpv = D.nca ? D.dc[t] : 0.0; b = D.bc[t]; dg = D.U[81ll * (t / 9) + 10 * (t % 9)]; }
        else { const int u = t - nc9, k = u % 3; v = D.pt_id[u / 3] + k; dpv = D.dp[u]; b = D.bp[u]; dg = D.V[6ll * (u / 3) + (k == 0 ? 0 : k == 1 ? 2 : 5)]; }
        if (!D.freem[v]) continue;
        if (mode == 0) {
            double xn = D.psave[t] + dpv;
            if (D.R == 2) {   // the pixel-residual model keeps every trial point inside the domains
                const double lo = D.P.lo[v], hi = D.P.hi[v];
                xn = (lo <= xn && xn <= hi) ? xn : (xn < lo ? lo : hi);
            }
            D.P.x[v] = xn;
            dl2 += dpv * dpv;
            dL += dpv * (damp(D, mu, floor_, dg) * dpv + b);
        } else if (mode == 1) {
            D.psave[t] = D.P.x[v];
        } else if (mode == 2) {
            D.P.x[v] = D.psave[t];
        } else {
            const double x = D.psave[t], lo = D.P.lo[v], hi = D.P.hi[v];
            D.P.x[v] = (lo <= x && x <= hi) ? x : (x < lo ? lo : hi);
        }
    }
    if (mode != 0) return;
    dl2 = wsum(dl2); dL = wsum(dL);
    if ((threadIdx.x & 63) == 0) { r0[threadIdx.x >> 6] = dl2; r1[threadIdx.x >> 6] = dL; }
    __syncthreads();
    double* part = D.part + 2048;   // (k_obj's partials occupy the first half)
    unsigned* ticket = reinterpret_cast<unsigned*>(D.sc + SC_TICKET);
    if (threadIdx.x == 0) {
        double a = 0.0, b = 0.0;
        for (int w = 0; w < 16; ++w) { a += r0[w]; b += r1[w]; }
        part[2 * blockIdx.x] = a; part[2 * blockIdx.x + 1] = b;
        __threadfence();
        last = atomicAdd(ticket, 1u) == gridDim.x - 1;
        if (last) {
            __threadfence();
            double sa = 0.0, sb = 0.0;
            for (unsigned q = 0; q < gridDim.x; ++q) {
                sa += __hip_atomic_load(part + 2 * q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                sb += __hip_atomic_load(part + 2 * q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            D.sc[3] = sa; D.sc[4] = sb;
            *ticket = 0u;
        }
    }
}

struct Buf {   // a slice of the caller's workspace
    void* p = nullptr;
    template <class T> T* as() { return static_cast<T*>(p); }
};

#define LM_CHK(expr)                                   \
    do {                                               \
        const hipError_t e_ = (expr);                  \
        if (e_ != hipSuccess) return (int)e_;          \
    } while (0)

template <class T>
int up(Buf& b, const std::vector<T>& h, hipStream_t s) {
    if (!h.empty()) LM_CHK(hipMemcpyAsync(b.p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, s));
    return 0;
}

}  // namespace

LmWorkspace::~LmWorkspace() {
    if (dev) (void)hipFree(dev);
    if (pinned) (void)hipHostFree(pinned);
}

int device_lm_ba(hipStream_t stream, const LmProblem& P, int64_t nfree, const int64_t* free_vid, int64_t nf,
                 const int64_t* fac, const LmOptions& opt, LmWorkspace* ws, LmResult* out, std::string* err) {
    *out = LmResult();
    auto fail = [&](const std::string& m) { if (err) *err = m; return -1; };
    if (nfree <= 0 || nf <= 0) return fail("lm: empty variable or factor list");
    // ---- host: which camera / point blocks are active, factor lists per block --------------------
    std::vector<unsigned char> freem((size_t)P.N, 0);
    for (int64_t i = 0; i < nfree; ++i) {
        if (free_vid[i] < 0 || free_vid[i] >= P.N) return fail("lm: variable id out of range");
        freem[(size_t)free_vid[i]] = 1;
    }
    const int *h_cam = P.h_cam, *h_pt = P.h_pt;
    const int npts_all = (P.N - 9 * P.ncams) / 3;
    std::vector<int> cam_local((size_t)std::max(P.ncams, 1), -1), pt_local((size_t)std::max(npts_all, 1), -1);
    auto block_free = [&](int v0, int len) { for (int k = 0; k < len; ++k) if (freem[(size_t)(v0 + k)]) return true; return false; };
    std::vector<int> lf((size_t)nf), fci((size_t)nf, -1), fpi((size_t)nf, -1), cam_id, pt_id;
    for (int64_t j = 0; j < nf; ++j) {
        if (fac[j] < 0 || fac[j] >= P.F) return fail("lm: factor id out of range");
        lf[(size_t)j] = (int)fac[j];
    }
    // active blocks (those with a free variable) in ascending id order; a free variable that no
    // listed factor reads has a zero column: its step is 0, like in the dense formulation
    for (int c = 0; c < P.ncams; ++c) if (block_free(9 * c, 9)) { cam_local[(size_t)c] = (int)cam_id.size(); cam_id.push_back(9 * c); }
    for (int p = 0; p < npts_all; ++p) if (block_free(9 * P.ncams + 3 * p, 3)) { pt_local[(size_t)p] = (int)pt_id.size(); pt_id.push_back(9 * P.ncams + 3 * p); }
    const int nca = (int)cam_id.size(), npa = (int)pt_id.size();
    std::vector<int> cam_ptr((size_t)nca + 1, 0), pt_ptr((size_t)npa + 1, 0);
    for (int64_t j = 0; j < nf; ++j) {
        const int f = lf[(size_t)j];
        fci[(size_t)j] = cam_local[(size_t)(h_cam[(size_t)f] / 9)];
        fpi[(size_t)j] = pt_local[(size_t)((h_pt[(size_t)f] - 9 * P.ncams) / 3)];
        if (fci[(size_t)j] >= 0) ++cam_ptr[(size_t)fci[(size_t)j] + 1];
        if (fpi[(size_t)j] >= 0) ++pt_ptr[(size_t)fpi[(size_t)j] + 1];
    }
    for (int c = 0; c < nca; ++c) cam_ptr[(size_t)c + 1] += cam_ptr[(size_t)c];
    for (int p = 0; p < npa; ++p) pt_ptr[(size_t)p + 1] += pt_ptr[(size_t)p];
    std::vector<int> cam_list((size_t)cam_ptr[(size_t)nca]), pt_list((size_t)pt_ptr[(size_t)npa]);
    {
        std::vector<int> cc(cam_ptr.begin(), cam_ptr.end() - 1), pc(pt_ptr.begin(), pt_ptr.end() - 1);
        for (int64_t j = 0; j < nf; ++j) {   // listed order = summation order
            if (fci[(size_t)j] >= 0) cam_list[(size_t)cc[(size_t)fci[(size_t)j]]++] = (int)j;
            if (fpi[(size_t)j] >= 0) pt_list[(size_t)pc[(size_t)fpi[(size_t)j]]++] = (int)j;
        }
    }
    // two factors joining the same camera and point would need their Z blocks added: not the BAL model
    {
        std::vector<int> seen((size_t)std::max(nca, 1), -1);
        for (int p = 0; p < npa; ++p)
            for (int t = pt_ptr[(size_t)p]; t < pt_ptr[(size_t)p + 1]; ++t) {
                const int ci = fci[(size_t)pt_list[(size_t)t]];
                if (ci < 0) continue;
                if (seen[(size_t)ci] == p) return fail("lm: two listed factors join the same camera and point");
                seen[(size_t)ci] = p;
            }
    }
    if (opt.model != 1 && opt.model != 2) return fail("lm: residual model must be 1 or 2");
    if (opt.schur < 0 || opt.schur > 2) return fail("lm: schur must be 0 (auto), 1 (dense) or 2 (sparse)");
    {   // (before anything sized by the reduced system is built)
        const double Mp0 = (double)std::max(64, (9 * nca + 63) / 64 * 64);
        if (Mp0 * Mp0 * 8.0 > 16e9) return fail("lm: reduced system too large");
    }
    // the camera pairs (a >= b) that share a point, with their (factor of a, factor of b) entries in point order --
    // not needed when the dense Schur path is forced; found by sorting (pair, entry) records: memory by the pairs that
    // occur (the sum of squared point degrees), not by the square of the number of cameras
    std::vector<int> pair_ab, pair_ptr(1, 0), pair_ja, pair_jb;
    int64_t pair_entries = 0;
    if (opt.schur != 1) {
        struct PairEntry { int64_t key; int ja, jb; };
        std::vector<PairEntry> ents;
        try {
            for (int p = 0; p < npa; ++p) {
                const int t0 = pt_ptr[(size_t)p], t1 = pt_ptr[(size_t)p + 1];
                for (int t = t0; t < t1; ++t) {
                    const int ja = pt_list[(size_t)t], a = fci[(size_t)ja];
                    if (a < 0) continue;
                    for (int u = t0; u < t1; ++u) {
                        const int jb = pt_list[(size_t)u], b = fci[(size_t)jb];
                        if (b < 0 || b > a) continue;
                        ents.push_back({(int64_t)a * (int64_t)(nca + 1) + b, ja, jb});
                    }
                }
            }
            // (stable: within a pair the entries keep their point order -- the order of the sums)
            std::stable_sort(ents.begin(), ents.end(), [](const PairEntry& x, const PairEntry& y) { return x.key < y.key; });
            pair_ja.reserve(ents.size()); pair_jb.reserve(ents.size());
            for (size_t i = 0; i < ents.size(); ++i) {
                if (i == 0 || ents[i].key != ents[i - 1].key) {
                    if (i > 0) pair_ptr.push_back((int)i);
                    pair_ab.push_back((int)(ents[i].key / (nca + 1))); pair_ab.push_back((int)(ents[i].key % (nca + 1)));
                }
                pair_ja.push_back(ents[i].ja); pair_jb.push_back(ents[i].jb);
            }
            if (!ents.empty()) pair_ptr.push_back((int)ents.size());
        } catch (const std::bad_alloc&) {
            return fail("lm: out of host memory for the camera-pair list (schur = 1 selects the dense path)");
        }
        pair_entries = (int64_t)ents.size();
    }
    Dev D{};
    D.P = P; D.nf = (int)nf; D.nca = nca; D.npa = npa; D.R = opt.model;
    D.M = 9 * nca; D.Mp = std::max(64, (D.M + 63) / 64 * 64); D.Kp = std::max(4, (3 * npa + 3) / 4 * 4);
    const int ntile = D.Mp / 64;
    D.SK = std::max(1, std::min(64, 2048 / std::max(1, ntile * (ntile + 1) / 2)));
    D.SK = std::min(D.SK, std::max(1, D.Kp / 64));
    // sparse or dense Schur product: the flops of the pairs' 9 x 9 blocks against the dense rank-3P update on the matrix
    // cores (which runs some three times faster per flop than the vector units and streams instead of gathering)
    D.npair = (int)(pair_ptr.size() - 1);
    {
        const double sparse_flops = (double)pair_entries * 81.0 * (2.0 * opt.model * opt.model + 1.0) * 2.0;
        const double dense_flops = (double)D.Mp * D.Mp * D.Kp;   // lower triangle, 2 flops per multiply-add
        D.sparse = opt.schur == 2 || (opt.schur == 0 && sparse_flops * 8.0 < dense_flops) ? 1 : 0;   // (schur = 1: no pair list, dense)
    }
    if (!D.sparse && (double)D.Mp * D.Kp * 8.0 > 16e9) return fail("lm: reduced system too large for the dense Schur path");
    if ((double)D.Mp * D.Mp * 8.0 > 16e9) return fail("lm: reduced system too large");

    // Every device buffer of the solve is a slice of one workspace the caller keeps between solves
    // (some thirty allocations and releases per call otherwise: 2 ms on ladybug).
    Buf b_lf, b_fci, b_fpi, b_free, b_cptr, b_clist, b_pptr, b_plist, b_cid, b_pid, b_pab, b_pptr2, b_pja, b_pjb;
    Buf Jc, Jp, e, U, bc, V, bp, Lp, yp, T, Zt, Zc, Spart, S, Ld, rhs, dc, dp, psave, part, sc;
    const size_t MM = (size_t)D.Mp * D.Mp;
    {
        const size_t n1c = (size_t)std::max(nca, 1), n1p = (size_t)std::max(npa, 1);
        const std::pair<Buf*, size_t> want[] = {
            {&b_lf, lf.size() * 4}, {&b_fci, fci.size() * 4}, {&b_fpi, fpi.size() * 4}, {&b_free, freem.size()},
            {&b_cptr, cam_ptr.size() * 4}, {&b_clist, cam_list.size() * 4}, {&b_pptr, pt_ptr.size() * 4},
            {&b_plist, pt_list.size() * 4}, {&b_cid, cam_id.size() * 4}, {&b_pid, pt_id.size() * 4},
            {&b_pab, pair_ab.size() * 4}, {&b_pptr2, pair_ptr.size() * 4}, {&b_pja, pair_ja.size() * 4}, {&b_pjb, pair_jb.size() * 4},
            {&Jc, (size_t)nf * 72 * D.R}, {&Jp, (size_t)nf * 24 * D.R}, {&e, (size_t)nf * 8 * D.R}, {&U, n1c * 81 * 8},
            {&bc, (size_t)D.Mp * 8}, {&V, n1p * 48}, {&bp, n1p * 24}, {&Lp, n1p * 48}, {&yp, (size_t)D.Kp * 8},
            {&T, (size_t)nf * 24 * D.R}, {&Zt, D.sparse ? 8 : (size_t)D.Kp * D.Mp * 8}, {&Zc, D.sparse ? (size_t)nf * 27 * 8 : 8}, {&Spart, D.sparse ? 8 : MM * 8 * (size_t)D.SK}, {&S, MM * 8}, {&Ld, (size_t)D.Mp * 32 * 8},
            {&rhs, (size_t)D.Mp * 8}, {&dc, (size_t)D.Mp * 8}, {&dp, (size_t)D.Kp * 8},
            {&psave, (size_t)(9 * nca + 3 * npa + 1) * 8}, {&part, 4096 * 8}, {&sc, 16 * 8}};
        size_t total = 0;
        for (const auto& w : want) total += (std::max<size_t>(w.second, 8) + 255) / 256 * 256;
        if (ws->dev_bytes < total) {
            LM_CHK(hipStreamSynchronize(stream));
            if (ws->dev) { LM_CHK(hipFree(ws->dev)); ws->dev = nullptr; ws->dev_bytes = 0; }
            LM_CHK(hipMalloc(&ws->dev, total + total / 4));
            ws->dev_bytes = total + total / 4;
        }
        char* base = static_cast<char*>(ws->dev);
        for (const auto& w : want) { w.first->p = base; base += (std::max<size_t>(w.second, 8) + 255) / 256 * 256; }
        if (!ws->pinned) LM_CHK(hipHostMalloc((void**)&ws->pinned, 24 * sizeof(double), hipHostMallocDefault));
    }
    int rc;
    if ((rc = up(b_lf, lf, stream)) || (rc = up(b_fci, fci, stream)) || (rc = up(b_fpi, fpi, stream)) || (rc = up(b_free, freem, stream)) ||
        (rc = up(b_cptr, cam_ptr, stream)) || (rc = up(b_clist, cam_list, stream)) || (rc = up(b_pptr, pt_ptr, stream)) ||
        (rc = up(b_plist, pt_list, stream)) || (rc = up(b_cid, cam_id, stream)) || (rc = up(b_pid, pt_id, stream)) ||
        (rc = up(b_pab, pair_ab, stream)) || (rc = up(b_pptr2, pair_ptr, stream)) || (rc = up(b_pja, pair_ja, stream)) || (rc = up(b_pjb, pair_jb, stream))) return rc;
    D.lf = b_lf.as<int>(); D.fci = b_fci.as<int>(); D.fpi = b_fpi.as<int>(); D.freem = b_free.as<unsigned char>();
    D.cam_ptr = b_cptr.as<int>(); D.cam_list = b_clist.as<int>(); D.pt_ptr = b_pptr.as<int>(); D.pt_list = b_plist.as<int>();
    D.cam_id = b_cid.as<int>(); D.pt_id = b_pid.as<int>();
    D.pair_ab = b_pab.as<int>(); D.pair_ptr = b_pptr2.as<int>(); D.pair_ja = b_pja.as<int>(); D.pair_jb = b_pjb.as<int>();
    D.Jc = Jc.as<double>(); D.Jp = Jp.as<double>(); D.e = e.as<double>(); D.U = U.as<double>(); D.bc = bc.as<double>();
    D.V = V.as<double>(); D.bp = bp.as<double>(); D.Lp = Lp.as<double>(); D.yp = yp.as<double>(); D.T = T.as<double>();
    D.Zt = Zt.as<double>(); D.Zc = Zc.as<double>(); D.Spart = Spart.as<double>(); D.S = S.as<double>(); D.Ld = Ld.as<double>(); D.rhs = rhs.as<double>();
    D.dc = dc.as<double>(); D.dp = dp.as<double>(); D.psave = psave.as<double>(); D.part = part.as<double>(); D.sc = sc.as<double>();
    LM_CHK(hipMemsetAsync(D.sc, 0, 16 * 8, stream));                      // (the workspace is reused: k_apply's ticket starts at zero)
    if (!D.sparse) LM_CHK(hipMemsetAsync(D.Zt, 0, (size_t)D.Kp * D.Mp * 8, stream));   // the block pattern of Z is fixed: zero once
    LM_CHK(hipMemsetAsync(D.bc, 0, (size_t)D.Mp * 8, stream));
    LM_CHK(hipMemsetAsync(D.yp, 0, (size_t)D.Kp * 8, stream));
    LM_CHK(hipMemsetAsync(D.dc, 0, (size_t)D.Mp * 8, stream));
    LM_CHK(hipMemsetAsync(D.dp, 0, (size_t)D.Kp * 8, stream));
    LM_CHK(hipMemsetAsync(D.rhs, 0, (size_t)D.Mp * 8, stream));

    const int gf = (int)((nf + 255) / 256), gp = (npa + 255) / 256, gobj = (int)std::min<int64_t>(gf, 2048);
    const int gapply = std::max(1, std::min(APPLY_MAX_BLOCKS, (9 * nca + 3 * npa + 1023) / 1024));
    // Host <-> device scalars go through one pinned buffer: h[0..15] the device's D.sc, h[16..17] the
    // damping (mu, floor) of the next attempt.
    double* h = ws->pinned;
    std::memset(h, 0, 24 * sizeof(double));
    auto fetch_scalars = [&]() -> int { LM_CHK(hipMemcpyAsync(h, D.sc, 16 * sizeof(double), hipMemcpyDeviceToHost, stream)); return 0; };
    auto objective = [&](int slot) -> int {
        k_obj<<<gobj, 256, 0, stream>>>(D);
        k_obj_final<<<1, 256, 0, stream>>>(D, gobj, slot);
        LM_CHK(hipGetLastError());
        return 0;
    };
    // The two launch sequences of an iteration: linearise (4 kernels), and one damped solve with its
    // trial point (about 45 kernels, most of them a few microseconds: the Cholesky panels).
    auto enqueue_linearise = [&]() -> int {
        k_lin<<<gf, 256, 0, stream>>>(D);
        if (nca) k_cam<<<nca, 64 * CAM_WAVES, 0, stream>>>(D);
        if (npa) k_pt<<<(npa + 15) / 16, 256, 0, stream>>>(D);
        k_scalars<<<std::max(1, std::min(SCALARS_MAX_BLOCKS, (3 * npa + 9 * nca + 1023) / 1024)), 1024, 0, stream>>>(D);
        LM_CHK(hipGetLastError());
        return fetch_scalars();
    };
    auto enqueue_attempt = [&]() -> int {
        LM_CHK(hipMemcpyAsync(D.sc + SC_MU, h + 16, 2 * sizeof(double), hipMemcpyHostToDevice, stream));
        if (npa) {
            k_ptchol<<<gp, 256, 0, stream>>>(D);
            k_z<<<gf, 256, 0, stream>>>(D);
        }
        if (nca) {
            if (D.sparse) {
                k_sinit<<<(unsigned)((MM + 255) / 256), 256, 0, stream>>>(D);
                if (D.npair) k_schur<<<D.npair, 64 * SCH_WAVES, 0, stream>>>(D);
            } else {
                const int nt = D.Mp / 64;
                k_syrk<<<dim3(nt * nt, D.SK), 64, 0, stream>>>(D);
                k_sfinish<<<(unsigned)((MM + 255) / 256), 256, 0, stream>>>(D);
            }
            k_rhs<<<nca, 64 * RHS_WAVES, 0, stream>>>(D);
            LM_CHK(hipMemsetAsync(D.sc + 8, 0, 8, stream));
            for (int kb = 0; kb < D.Mp; kb += 32) {
                const int rem = D.Mp - kb - 32;
                k_panel<<<(rem + 1 + 255) / 256, 256, 0, stream>>>(D, kb);
                if (rem > 0) {
                    const int ntr = rem / 32;
                    k_trail<<<ntr * ntr + ntr, 64, 0, stream>>>(D, kb);
                }
            }
            if (D.Mp <= TRSV_MAX) k_trsv<<<1, TRSV_THREADS, 0, stream>>>(D);
            else k_trsv_left<<<1, 1024, 0, stream>>>(D);
        }
        if (npa) k_back<<<(npa + 15) / 16, 256, 0, stream>>>(D);
        k_apply<<<gapply, 1024, 0, stream>>>(D, 0);
        LM_CHK(hipGetLastError());
        int rc_ = objective(5);
        return rc_ ? rc_ : fetch_scalars();
    };
    // (Replaying the two sequences as hipGraphs was measured: 24.2 against 24.3 ms for 25 iterations of
    // full ladybug -- the time is the kernels' own serial chain, not the launches.)
    auto run = [&](auto& enqueue) -> int {
        if (int rc_ = enqueue()) return rc_;
        LM_CHK(hipStreamSynchronize(stream));
        return 0;
    };

    k_apply<<<gapply, 1024, 0, stream>>>(D, 1);   // psave = x
    if ((rc = objective(5)) || (rc = fetch_scalars())) return rc;
    LM_CHK(hipStreamSynchronize(stream));
    double p_eL2 = 2.0 * h[5];
    out->finit = h[5];
    out->ncam_blocks = nca; out->npt_blocks = npa; out->sparse_schur = D.sparse;
    out->nfev = 1;
    double mu = 0.0;
    long long nu = 2;
    int stop = 0, k = 0;
    const double EPSILON = 1e-12, ONE_THIRD = 0.3333333334;
    while (k < opt.maxiters && !stop) {
        if (p_eL2 <= opt.eps3) { stop = 6; break; }
        if ((rc = run(enqueue_linearise))) return rc;
        ++out->njev;
        const double maxdiag = h[0], jte_inf = h[1], p_L2 = h[2];
        if (jte_inf <= opt.eps1) { stop = 1; break; }
        // Marquardt scaling (model 2): mu is relative to the diagonal; entries below 1e-9 of the largest
        // (a free variable nothing reads) are damped as if they were that large
        const double floor_ = 1e-9 * maxdiag;
        if (k == 0) mu = D.R == 2 ? opt.tau : opt.tau * maxdiag;
        for (;;) {
            h[16] = mu; h[17] = floor_;
            if ((rc = run(enqueue_attempt))) return rc;
            ++out->nsolve;
            const bool solved = !(nca && h[8] != 0.0);
            if (solved) {
                const double Dp_L2 = h[3], dL = h[4];
                if (Dp_L2 <= opt.eps2 * opt.eps2 * p_L2) { stop = 2; break; }
                if (Dp_L2 >= (p_L2 + opt.eps2) / (EPSILON * EPSILON)) { stop = 4; break; }
                ++out->nfev;
                const double pDp_eL2 = 2.0 * h[5];
                if (!std::isfinite(pDp_eL2)) { stop = 7; break; }
                const double dF = p_eL2 - pDp_eL2;
                const bool ok = dL > 0.0 && dF > 0.0;
                out->history.push_back(LmStep{mu, Dp_L2, h[5], ok ? 1 : 0});
                if (ok) {
                    double tmp = 2.0 * dF / dL - 1.0;
                    tmp = 1.0 - tmp * tmp * tmp;
                    mu = mu * (tmp >= ONE_THIRD ? tmp : ONE_THIRD);
                    nu = 2;
                    p_eL2 = pDp_eL2;
                    k_apply<<<gapply, 1024, 0, stream>>>(D, 1);   // accept: psave = x
                    break;
                }
            }
            mu *= (double)nu;
            const long long nu2 = nu << 1;
            if (nu2 >= (1ll << 31)) { stop = 5; break; }
            nu = nu2;
        }
        ++k;
    }
    if (!stop) stop = 3;
    k_apply<<<gapply, 1024, 0, stream>>>(D, 3);   // x = clamp(accepted point)
    if ((rc = objective(5)) || (rc = fetch_scalars())) return rc;
    LM_CHK(hipStreamSynchronize(stream));
    out->fret = h[5];
    out->mu = mu; out->iters = k; out->stop = stop;
    return 0;
}

}  // namespace rdis_hip
