// grad_fused.hip -- kernels of the fused value + gradient pass (grad_fused.hpp) and their launches.
#include "grad_fused.hpp"
#include "solver_wg.hpp"

namespace rdis_hip {

#ifdef RDIS_GRAD_STAMPS
// (measurement builds only) cycles of the first workgroup's first wave, summed over its chunks: 0 the factors, 1 stage 2 + wave sum +
// the wait at the first barrier, 2 the segment sums, 3 the wait at the second barrier, 4 chunks, 5 the tile's prologue + epilogue
__device__ long long grad_stamps[8];
#define GRAD_STAMP(i, expr) do { const long long t_ = clock64(); if (blockIdx.x == 0 && threadIdx.x == 0) grad_stamps[i] += t_ - tlast; tlast = t_; (void)(expr); } while (0)
#else
#define GRAD_STAMP(i, expr) do { } while (0)
#endif

// one lane per camera block: its record for this call
__global__ void __launch_bounds__(256)
grad_camera_records_kernel(const double* __restrict__ x, const int* __restrict__ cam_blocks, int nblocks, double* __restrict__ camrec) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nblocks) return;
    const int c = cam_blocks[i];
    double* r = camrec + (size_t)i * GRAD_REC;
    store_rotation(x[c], x[c + 1], x[c + 2], r);   // [v0 v1 v2 theta 1/theta sin cos]
#pragma unroll
    for (int k = 0; k < 6; ++k) r[7 + k] = x[c + 3 + k];
    r[13] = 0.0;
}

// the sum of a workgroup's values, waves in order (the order eval_chunks_kernel and grad_fused_kernel share)
__device__ __forceinline__ double chunk_sum_of(const double* red) {
    double r = 0.0;
#pragma unroll
    for (int i = 0; i < GRAD_LANES / 64; ++i) r += red[i];
    return r;
}

// ROT = ROT_CAMFIX: the cameras' rotation records (P.xrot, camera_rotations_kernel) are current for the assigned x -- the factor
// reads angle, axis, sine and cosine instead of forming them (the same arithmetic once per camera: the same bits)
template <int KIND, int ROT = ROT_PER_FACTOR>
__global__ void __launch_bounds__(GRAD_LANES)
eval_chunks_kernel(ProblemView P, int nf, const int* __restrict__ fac, double* __restrict__ partial) {
    __shared__ double red[GRAD_LANES / 64];
    const int nchunks = (nf + GRAD_LANES - 1) / GRAD_LANES;
    for (int ch = blockIdx.x; ch < nchunks; ch += gridDim.x) {
        const int j = ch * GRAD_LANES + (int)threadIdx.x;
        double f = 0.0, s;
        if (j < nf) factor_value<KIND, false, ROT>(P, nullptr, fac ? fac[j] : j, f, s);
        f = wave_sum(f);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = f;
        __syncthreads();
        if (threadIdx.x == 0) partial[ch] = chunk_sum_of(red);
        __syncthreads();
    }
}

// rows [r0, r1) of a staging area (row stride STRIDE doubles, entry k), added to `s` in row order, eight loads in flight
template <int STRIDE>
__device__ __forceinline__ double add_rows(double s, const double* __restrict__ rows, int r0, int r1, int k) {
    for (int b = r0; b < r1; b += 8) {
        double t[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] = (b + j < r1) ? rows[(b + j) * STRIDE + k] : 0.0;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (b + j < r1) s += t[j];
    }
    return s;
}

// ... the same sum from a staging area in LDS, leaner: the loads are unconditional (eight at immediate offsets from one address;
// up to seven rows behind the segment are read and not used -- GRAD_ROW_SLACK doubles behind the area keep that inside the
// workgroup's LDS), a row beyond the segment is left out by selecting the sum before it
template <int STRIDE>
__device__ __forceinline__ double add_rows_lds(double s, const double* __restrict__ rows, int r0, int r1, int k) {
    const double* p = rows + r0 * STRIDE + k;
    for (int n = r1 - r0; n > 0; n -= 8, p += 8 * STRIDE) {
        double t[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] = p[j * STRIDE];
#pragma unroll
        for (int j = 0; j < 8; ++j) { const double u = s + t[j]; s = j < n ? u : s; }
    }
    return s;
}

// What a lane needs of a chunk, asked for a chunk ahead.  A workgroup meets at two barriers per chunk, so its waves wait
// for memory together and only the other workgroup of the compute unit fills the gap: every load of a chunk is issued
// while the chunk before it is evaluated -- first the list's tables (stage 1), then what their entries point to (stage
// 2: the observation of an explicit list's factor, the point's values), half a chunk later.
constexpr int GRAD_CSEG_LANES = GRAD_LANES / 9 * 9, GRAD_PSEG_LANES = GRAD_LANES / 3 * 3;   // lanes that sum segments: whole groups of 9 / 3
struct GradEntry {
    unsigned cl;          // camera (tile numbering), GRAD_NO_ENTRY = no factor
    int q, fid;           // point block, factor id
    unsigned rc, rp;      // rows
    int2 cs, ps;          // this lane's camera / point segment of the chunk: {camera | destination, first row} ...
    int cs_end, ps_end;   // ... and the row behind it
    int extra;            // (uniform) the chunk has more segments than lanes / 9 (/ 3): a list in no order; the further ones come from the tables
    double2 o;
    double x0, x1, x2;
};
__device__ __forceinline__ void grad_stage1(const GradTables& T, int ch, int tid, GradEntry& e) {
    const int j = ch * GRAD_LANES + tid;
    e.cl = T.cl[j]; e.rc = T.rc[j]; e.rp = T.rp[j];
    e.q = T.ptv[j];
    e.fid = T.fac ? T.fac[j < T.nf ? j : T.nf - 1] : j;
    // the segment this lane sums after the chunk's barrier: lane = 9 segment + k (cameras), 3 segment + k (point blocks);
    // a chunk with more segments than lanes / 9 (lanes / 3) gives the lane segment + lanes / 9, ... as well (from the tables)
    const int cs0 = T.chunk_cseg0[ch], cns = T.chunk_cseg0[ch + 1] - cs0 - 1;
    const int ps0 = T.chunk_pseg0[ch], pns = T.chunk_pseg0[ch + 1] - ps0 - 1;
    const int sc = tid / 9, sp = tid / 3;
    const bool hc = sc < cns && tid < GRAD_CSEG_LANES, hp = sp < pns && tid < GRAD_PSEG_LANES;
    e.cs = T.cseg[cs0 + (hc ? sc : 0)]; e.cs_end = hc ? T.cseg[cs0 + sc + 1].y : 0;
    e.ps = T.pseg[ps0 + (hp ? sp : 0)]; e.ps_end = hp ? T.pseg[ps0 + sp + 1].y : 0;
    if (!hc) e.cs.y = 0;
    if (!hp) e.ps.y = 0;
    e.extra = (cns * 9 > GRAD_CSEG_LANES || pns * 3 > GRAD_PSEG_LANES) ? 1 : 0;
}
__device__ __forceinline__ void grad_stage2(const double* __restrict__ x, const double2* __restrict__ obs, GradEntry& e) {
    if (e.cl != GRAD_NO_ENTRY) {
        e.o = obs[e.fid];
        e.x0 = x[e.q]; e.x1 = x[e.q + 1]; e.x2 = x[e.q + 2];
    }
}

__global__ void __launch_bounds__(GRAD_LANES)
grad_fused_kernel(GradTables T, const double* __restrict__ x, const double2* __restrict__ obs, const double* __restrict__ camrec,
                  double* __restrict__ cstage, double* __restrict__ pstage, double* __restrict__ partial, double* __restrict__ g) {
    extern __shared__ __align__(16) double smem[];
    double* const rec = smem;                                   // [ncam_cap][GRAD_REC]
    double* const acc = rec + (size_t)T.ncam_cap * GRAD_REC;    // [ncam_cap][9]
    // the rows of a chunk: GRAD_ROW_BUFFERS staging areas taken in turn, each [GRAD_LANES][9] camera rows, [GRAD_LANES][3] point rows, [waves] values
    double* const rows0 = acc + (size_t)T.ncam_cap * 9;
    constexpr int ROWS = GRAD_LANES * 12 + 16 + GRAD_ROW_SLACK;
    const int tid = (int)threadIdx.x;
    const int kc = tid % 9, kp = tid % 3;
    for (int tile = blockIdx.x; tile < T.ntiles; tile += gridDim.x) {
        const int c0 = T.tile_cam0[tile], nc = T.tile_cam0[tile + 1] - c0;
        const int ch_begin = T.tile_chunk0[tile], ch_end = T.tile_chunk0[tile + 1];
        GradEntry cur, nxt;
#ifdef RDIS_GRAD_STAMPS
        long long tlast = clock64();
#endif
        grad_stage1(T, ch_begin, tid, cur);
        // the tile's cameras: records in, accumulators cleared
        for (int i = tid; i < nc * GRAD_REC; i += GRAD_LANES) {
            const int c = i / GRAD_REC, k = i - c * GRAD_REC;
            rec[i] = camrec[(size_t)T.tile_cam[c0 + c].x * GRAD_REC + k];
        }
        for (int i = tid; i < nc * 9; i += GRAD_LANES) acc[i] = 0.0;
        grad_stage2(x, obs, cur);
        __syncthreads();
        GRAD_STAMP(5, 0);
        for (int ch = ch_begin; ch < ch_end; ++ch) {
            const bool more = ch + 1 < ch_end;   // (uniform)
            // With two staging areas a chunk needs ONE barrier: the lanes that still add chunk ch's rows have not passed the barrier
            // of chunk ch + 1, so nobody is yet writing the rows of chunk ch + 2 into the area they read.
            double* const crow = rows0 + (size_t)((ch - ch_begin) % GRAD_ROW_BUFFERS) * ROWS;
            double* const prow = crow + GRAD_LANES * 9;
            double* const red = prow + GRAD_LANES * 3;
            if (more) grad_stage1(T, ch + 1, tid, nxt);
            double E = 0.0;
            if (cur.cl != GRAD_NO_ENTRY) {
                const double2* R = reinterpret_cast<const double2*>(rec + cur.cl * GRAD_REC);
                const double2 a0 = R[0], a1 = R[1], a2 = R[2], a3 = R[3], a4 = R[4], a5 = R[5], a6 = R[6];
                BaFwd t;
                t.v0 = a0.x; t.v1 = a0.y; t.v2 = a1.x; t.theta = a1.y; t.itheta = a2.x; t.s = a2.y; t.c = a3.x;
                double v[12], gr[12];
                v[0] = v[1] = v[2] = 0.0;   // (not read: the rotation comes from the record)
                v[3] = a3.y; v[4] = a4.x; v[5] = a4.y; v[6] = a5.x; v[7] = a5.y; v[8] = a6.x;
                v[9] = cur.x0; v[10] = cur.x1; v[11] = cur.x2;
#if defined(RDIS_GRAD_ABLATE) && (RDIS_GRAD_ABLATE == 1 || RDIS_GRAD_ABLATE == 4)
                // (measurement builds only -- wrong results: no factor arithmetic)
                E = v[9] * cur.o.x + t.c;
#pragma unroll
                for (int k = 0; k < 12; ++k) gr[k] = v[k] * cur.o.y;
#else
                E = ba_project(v, cur.o.x, cur.o.y, t);
                ba_adjoint(t, v, t.res0, t.res1, gr);
#endif
                double* cr = crow + (int)cur.rc * 9;
                double* pr = prow + (int)cur.rp * 3;
#pragma unroll
                for (int k = 0; k < 9; ++k) cr[k] = gr[k];
#pragma unroll
                for (int k = 0; k < 3; ++k) pr[k] = gr[9 + k];
            }
            GRAD_STAMP(0, 0);
            if (more) grad_stage2(x, obs, nxt);
            E = wave_sum(E);
            if ((tid & 63) == 0) red[tid >> 6] = E;
            __syncthreads();
            GRAD_STAMP(1, 0);
            if (tid == 0) partial[ch] = chunk_sum_of(red);
#if defined(RDIS_GRAD_ABLATE) && (RDIS_GRAD_ABLATE == 2 || RDIS_GRAD_ABLATE == 4)
            // (measurement builds only -- wrong results: no segment sums)
            if (false) {
#else
            {
#endif
            // cameras: lane (segment, k) adds the segment's rows to the camera's accumulator -- its first segment from the
            // registers asked for a chunk ago, further ones (a chunk with more than GRAD_LANES / 9 cameras) from the tables
            if (cur.cs_end > cur.cs.y) {
                double* a = acc + cur.cs.x * 9 + kc;
                *a = add_rows_lds<9>(*a, crow, cur.cs.y, cur.cs_end, kc);
            }
            if (cur.extra) {   // (else nothing here waits for global memory: the list bounds are not even looked at)
                const int s0 = T.chunk_cseg0[ch], ns = T.chunk_cseg0[ch + 1] - s0 - 1;
                for (int w = tid + GRAD_CSEG_LANES; w < ns * 9 && tid < GRAD_CSEG_LANES; w += GRAD_CSEG_LANES) {
                    const int sidx = w / 9;
                    const int2 sg = T.cseg[s0 + sidx];
                    const int r1 = T.cseg[s0 + sidx + 1].y;
                    double* a = acc + sg.x * 9 + kc;
                    *a = add_rows_lds<9>(*a, crow, sg.y, r1, kc);
                }
            }
            // point blocks: lane (segment, k) forms the block's sum of this chunk
            if (cur.ps_end > cur.ps.y) {
                const double sum = add_rows_lds<3>(0.0, prow, cur.ps.y, cur.ps_end, kp);
                if (cur.ps.x >= 0) g[cur.ps.x + kp] = sum; else pstage[(size_t)(~cur.ps.x) * 3 + kp] = sum;
            }
            if (cur.extra) {
                const int s0 = T.chunk_pseg0[ch], ns = T.chunk_pseg0[ch + 1] - s0 - 1;
                for (int w = tid + GRAD_PSEG_LANES; w < ns * 3 && tid < GRAD_PSEG_LANES; w += GRAD_PSEG_LANES) {
                    const int sidx = w / 3;
                    const int2 sg = T.pseg[s0 + sidx];
                    const int r1 = T.pseg[s0 + sidx + 1].y;
                    const double sum = add_rows_lds<3>(0.0, prow, sg.y, r1, kp);
                    if (sg.x >= 0) g[sg.x + kp] = sum; else pstage[(size_t)(~sg.x) * 3 + kp] = sum;
                }
            }
            }
            GRAD_STAMP(2, 0);
            if constexpr (GRAD_ROW_BUFFERS < 2) __syncthreads();
            GRAD_STAMP(3, 0);
#ifdef RDIS_GRAD_STAMPS
            if (blockIdx.x == 0 && threadIdx.x == 0) grad_stamps[4] += 1;
#endif
            if (more) cur = nxt;
        }
        if constexpr (GRAD_ROW_BUFFERS >= 2) __syncthreads();   // (the last chunk's sums are in the accumulators)
        for (int i = tid; i < nc * 9; i += GRAD_LANES) {
            const int c = i / 9, k = i - 9 * c;
            const int dest = T.tile_cam[c0 + c].y;
            if (dest >= 0) g[dest + k] = acc[i]; else cstage[(size_t)(~dest) * 9 + k] = acc[i];
        }
        __syncthreads();
        GRAD_STAMP(5, 0);
    }
}

// what more than one chunk / tile contributed to, in chunk / tile order; zero where no listed factor reads
__global__ void __launch_bounds__(256)
grad_combine_kernel(GradTables T, const double* __restrict__ cstage, const double* __restrict__ pstage, double* __restrict__ g) {
    const long long nc9 = 9ll * T.ncs, np3 = 3ll * T.nps, total = nc9 + np3 + T.nz;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        if (i < nc9) {
            const int b = (int)(i / 9), k = (int)(i - 9ll * b);
            g[T.cs_var[b] + k] = add_rows<9>(0.0, cstage, T.cs_ptr[b], T.cs_ptr[b + 1], k);
        } else if (i < nc9 + np3) {
            const long long u = i - nc9;
            const int b = (int)(u / 3), k = (int)(u - 3ll * b);
            g[T.ps_var[b] + k] = add_rows<3>(0.0, pstage, T.ps_ptr[b], T.ps_ptr[b + 1], k);
        } else {
            g[T.zvar[i - nc9 - np3]] = 0.0;
        }
    }
}

#ifdef RDIS_GRAD_STAMPS
}  // namespace rdis_hip
extern "C" int rdis_hip_debug_grad_stamps(long long* out, int clear) {
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(rdis_hip::grad_stamps), sizeof(long long) * 8);
    if (e == hipSuccess && clear) { long long z[8] = {}; e = hipMemcpyToSymbol(HIP_SYMBOL(rdis_hip::grad_stamps), z, sizeof z); }
    return (int)e;
}
namespace rdis_hip {
#endif
hipError_t grad_camera_records_launch(hipStream_t s, int grid, const double* x, const int* cam_blocks, int nblocks, double* camrec) {
    grad_camera_records_kernel<<<grid, 256, 0, s>>>(x, cam_blocks, nblocks, camrec);
    return hipGetLastError();
}
hipError_t grad_fused_launch(hipStream_t s, int grid, size_t dyn, const GradTables& T, const double* x, const double2* obs,
                             const double* camrec, double* cstage, double* pstage, double* partial, double* g) {
    if (dyn > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)grad_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
        if (e != hipSuccess) return e;
    }
    grad_fused_kernel<<<grid, GRAD_LANES, dyn, s>>>(T, x, obs, camrec, cstage, pstage, partial, g);
    return hipGetLastError();
}
hipError_t grad_combine_launch(hipStream_t s, int grid, const GradTables& T, const double* cstage, const double* pstage, double* g) {
    grad_combine_kernel<<<grid, 256, 0, s>>>(T, cstage, pstage, g);
    return hipGetLastError();
}
hipError_t eval_chunks_launch(hipStream_t s, int grid, const ProblemView& P, int nf, const int* fac, double* partial) {
    if (P.kind == KIND_BA && P.rot_mode == ROT_CAMFIX && P.xrot != nullptr) eval_chunks_kernel<KIND_BA, ROT_CAMFIX><<<grid, GRAD_LANES, 0, s>>>(P, nf, fac, partial);
    else if (P.kind == KIND_BA) eval_chunks_kernel<KIND_BA><<<grid, GRAD_LANES, 0, s>>>(P, nf, fac, partial);
    else eval_chunks_kernel<KIND_NLP><<<grid, GRAD_LANES, 0, s>>>(P, nf, fac, partial);
    return hipGetLastError();
}
}  // namespace rdis_hip
