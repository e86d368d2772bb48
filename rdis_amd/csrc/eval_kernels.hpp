// eval_kernels.hpp -- batched factor evaluation over a factor list at the
// currently assigned x (K1 / K2 of SURVEY.md 2.1): the device side of
// OptimizableFunction::evalFactors and computeGradient(facs, pg)
// (reference src/OptimizableFunction.cpp:95-135, 234-262).
//
// HBM-bound streaming kernels: one lane per factor, grid-stride, 16-byte obs load
// + two int32 indices per factor, variable blocks gathered through L1/L2.
// Reductions are fixed-order (wave butterfly -> LDS -> per-block partial ->
// single-block final pass), so results are bit-reproducible.
#pragma once
#include "solver_wg.hpp"

namespace rdis_hip {

template <int KIND>
__global__ void __launch_bounds__(256)
eval_each_kernel(ProblemView P, int nf, const int* __restrict__ fac, double* __restrict__ out) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nf; i += gridDim.x * blockDim.x) {
        double f, s;
        factor_value<KIND, false>(P, nullptr, fac ? fac[i] : i, f, s);
        out[i] = f;
    }
}

// per-factor partials; BA only for the public grad_each entry point
template <int KIND>
__global__ void __launch_bounds__(256)
partials_kernel(ProblemView P, int nf, const int* __restrict__ fac, double* __restrict__ gfac) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nf; i += gridDim.x * blockDim.x)
        factor_partials<KIND>(P, gfac, nullptr, fac ? fac[i] : i);
}

// rotation record of every camera block at the assigned x (one lane per camera): what the factors of
// a launch without free camera variables read instead of redoing ba_rotation per trial point
__global__ void __launch_bounds__(256)
camera_rotations_kernel(const double* __restrict__ x, const int* __restrict__ cam_blocks, int nblocks,
                        double* __restrict__ xrot) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nblocks) return;
    const int c = cam_blocks[i];
    store_rotation(x[c], x[c + 1], x[c + 2], xrot + c);
}

// a plan's copy of its listed factors' observations, in listed order (solver_lds.hpp)
__global__ void __launch_bounds__(256)
gather_obs_kernel(int n, const int* __restrict__ fac, const double2* __restrict__ obs, double2* __restrict__ out) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) out[i] = obs[fac[i]];
}

__global__ void __launch_bounds__(256)
gather_rows12_kernel(int nf, const int* __restrict__ fac, const double* __restrict__ gfac,
                     double* __restrict__ out) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nf * 12; i += gridDim.x * blockDim.x) {
        const int r = i / 12, k = i - 12 * r;
        out[i] = gfac[12ll * (fac ? fac[r] : r) + k];
    }
}

__device__ __forceinline__ double block_sum(double v, double* red /* [MAX_WAVES] */) {
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double r = 0.0;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) r += red[i];
    __syncthreads();
    return r;
}

// f partial per block (+ optionally the per-factor partials for the gradient)
template <int KIND, bool GRAD>
__global__ void __launch_bounds__(256)
eval_sum_kernel(ProblemView P, int nf, const int* __restrict__ fac, double* __restrict__ gfac,
                double* __restrict__ block_partial) {
    __shared__ double red[MAX_WAVES];
    double acc = 0.0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nf; i += gridDim.x * blockDim.x) {
        const int fid = fac ? fac[i] : i;
        double f, s;
        factor_value<KIND, false>(P, nullptr, fid, f, s);
        acc += f;
        if constexpr (GRAD) factor_partials<KIND>(P, gfac, nullptr, fid);
    }
    acc = block_sum(acc, red);
    if (threadIdx.x == 0) block_partial[blockIdx.x] = acc;
}

__global__ void __launch_bounds__(256)
final_sum_kernel(int n, const double* __restrict__ partial, double* __restrict__ out) {
    __shared__ double red[MAX_WAVES];
    double acc = 0.0;
    // (eight loads in flight, added in index order: the bits of the plain loop without its round trip per element)
    for (int i0 = threadIdx.x; i0 < n; i0 += 8 * blockDim.x) {
        double t[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] = (i0 + j * (int)blockDim.x < n) ? partial[i0 + j * (int)blockDim.x] : 0.0;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (i0 + j * (int)blockDim.x < n) acc += t[j];
    }
    acc = block_sum(acc, red);
    if (threadIdx.x == 0) out[0] = acc;
}

// g[v] = sum of the slots that feed v, in factor-list order (src/State.h:157-210)
__global__ void __launch_bounds__(256)
gather_grad_kernel(int nvars, const int* __restrict__ v2s_ptr, const int* __restrict__ v2s_idx,
                   const double* __restrict__ gfac, double* __restrict__ g) {
    for (int v = blockIdx.x * blockDim.x + threadIdx.x; v < nvars; v += gridDim.x * blockDim.x) {
        const int b = v2s_ptr[v], e = v2s_ptr[v + 1];
        // sixteen slots in flight (their positions first, then the values), the additions strictly in
        // order: the bits of the plain loop without its one memory round trip per slot -- a camera's
        // run is hundreds of slots long
        double s = 0.0;
        for (int k0 = b; k0 < e; k0 += 16) {
            int id[16];
            double t[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) id[j] = (k0 + j < e) ? v2s_idx[k0 + j] : 0;
#pragma unroll
            for (int j = 0; j < 16; ++j) t[j] = (k0 + j < e) ? gfac[id[j]] : 0.0;
#pragma unroll
            for (int j = 0; j < 16; ++j)
                if (k0 + j < e) s = (k0 + j == b) ? t[j] : s + t[j];
        }
        g[v] = s;
    }
}

__global__ void scatter_x_kernel(int n, const int* __restrict__ vid, const double* __restrict__ val,
                                 double* __restrict__ x) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        x[vid ? vid[i] : i] = val[i];
}
__global__ void gather_x_kernel(int n, const int* __restrict__ vid, const double* __restrict__ x,
                                double* __restrict__ out) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        out[i] = x[vid ? vid[i] : i];
}

__global__ void __launch_bounds__(256)
objective_sum_kernel(int ncomp, const double* __restrict__ fret, double* __restrict__ out) {
    // top-level objective = sum over components (reference src/RDISOptimizer.cpp:1491-1494)
    __shared__ double red[MAX_WAVES];
    double acc = 0.0;
    for (int i = threadIdx.x; i < ncomp; i += blockDim.x) acc += fret[i];
    acc = block_sum(acc, red);
    if (threadIdx.x == 0) out[0] = acc;
}

}  // namespace rdis_hip
