// grad_fused.hpp -- K2 of SURVEY.md 2.1 behind the public gradient entry point: value and gradient of a factor list in ONE
// pass over the factors, the device side of OptimizableFunction::computeGradient(facs, pg) + productGradient's merge
// (reference src/OptimizableFunction.cpp:234-262, src/State.h:157-210, src/bundleadjust/BundleAdjustmentFactor.cpp:351-554)
// for bundle-adjustment functions.  What the host side (rdis_hip.hip) needs: the layout of a list's tables and the
// launches; the kernels are compiled in grad_fused.hip.
//
// The two-kernel form this replaces wrote a factor's twelve partials to HBM (96 bytes per factor) and read them back through
// per-variable slot lists: 5.3 x the algorithmic bytes of SURVEY 8d.  Here no partial leaves the compute unit:
//   * the list is cut into CHUNKS of GRAD_LANES consecutive entries (a lane per entry) and TILES of a few chunks; a workgroup
//     takes a tile;
//   * a tile's distinct cameras have their records -- rotation (angle, axis, sine, cosine: factors.hpp ba_rotation) and the
//     six other camera values, formed once per camera and call by grad_camera_records_kernel -- and their nine gradient
//     accumulators in LDS for the whole tile;
//   * per chunk every lane evaluates its factor (forward pass + adjoint sweep, factors.hpp) and leaves its nine camera
//     partials and three point partials in LDS rows whose order the list's tables fix: rows sorted by camera / by point
//     block, list order within a block.  Then lane (segment, k) adds a block's rows in that order -- cameras into the tile's
//     accumulators, points into their final sum;
//   * a point block whose listed factors all stand in one chunk (every point of a list in the loaders' point-major order
//     but those that straddle a chunk boundary) and a camera whose listed factors all stand in one tile go straight to g;
//     the others leave one partial sum per (block, chunk / tile) in a staging array and grad_combine_kernel adds those in
//     chunk / tile order.  Every sum has a fixed order: the same bits run to run.
// Traffic per factor: observation 16 B + point id 4 B + three 16-bit table entries 6 B (+ 4 B of factor id for an explicit
// list) against SURVEY's 24 B; the point's values through L1 / L2; per chunk and tile a few hundred bytes of segment tables.
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>
#include "device_views.hpp"

namespace rdis_hip {

#ifndef RDIS_GRAD_LANES
#define RDIS_GRAD_LANES 512
#endif
constexpr int GRAD_ROW_SLACK = 32;             // doubles behind a staging area that its segment sums may read and not use
constexpr int GRAD_LANES = RDIS_GRAD_LANES;   // entries of a chunk = lanes of a workgroup (eight waves: two per SIMD, two workgroups per compute unit)
constexpr int GRAD_REC = 14;           // doubles of a camera's record: [v0 v1 v2 theta 1/theta sin cos | t0 t1 t2 f k1 k2 | pad] -- seven
                                       // 16-byte units, an odd number: lanes that read different cameras stand on different LDS banks
constexpr int GRAD_MAX_TILE_CHUNKS = 8;
constexpr unsigned short GRAD_NO_ENTRY = 0xFFFF;

// a factor list's tables on the device (all built on the host once per list, rdis_hip.hip: build_grad_plan)
struct GradTables {
    int nf, nchunks, ntiles, ncam_cap;        // list entries; chunks of GRAD_LANES; tiles; the largest number of distinct cameras in a tile
    const int* fac;                           // [nf] factor ids, or null = 0 .. nf-1
    // per entry (padded to whole chunks)
    const unsigned short* cl;                 // the factor's camera, numbered within its tile (GRAD_NO_ENTRY: padding)
    const unsigned short* rc;                 // its row among the chunk's entries ordered by camera
    const unsigned short* rp;                 // ... ordered by point block
    const int* ptv;                           // first variable id of its point block
    // per tile
    const int* tile_chunk0;                   // [ntiles + 1] first chunk
    const int* tile_cam0;                     // [ntiles + 1] first entry of tile_cam
    const int2* tile_cam;                     // per distinct camera of a tile: {ordinal of the camera block, destination}: >= 0 the camera's
                                              // first variable id (complete in this tile: straight to g), < 0: ~slot in the camera staging array
    // per chunk: segments of equal camera / point block in its sorted rows; a chunk's list ends with one entry {-, rows of the chunk}
    const int* chunk_cseg0;                   // [nchunks + 1]
    const int2* cseg;                         // {camera (tile numbering), first row}
    const int* chunk_pseg0;                   // [nchunks + 1]
    const int2* pseg;                         // {destination (>= 0: first variable id, < 0: ~slot in the point staging array), first row}
    // the staged blocks (grad_combine_kernel): a block's partial sums are consecutive slots, in chunk / tile order
    int ncs, nps, nz;
    const int* cs_var; const int* cs_ptr;     // [ncs], [ncs + 1]: camera block's first variable id, its slots
    const int* ps_var; const int* ps_ptr;     // [nps], [nps + 1]
    const int* zvar;                          // [nz] variables no listed factor reads: their gradient entry is 0
};

#ifndef RDIS_GRAD_ROW_BUFFERS
#define RDIS_GRAD_ROW_BUFFERS 1
#endif
// 2 (a measurement build, round 6): a chunk's rows alternate between two staging areas and a chunk needs ONE workgroup barrier --
// but 98 KB of rows leave room for one workgroup per compute unit instead of two: 154 -> 256 us at 8e6 factors.  1 is the product.
constexpr int GRAD_ROW_BUFFERS = RDIS_GRAD_ROW_BUFFERS;
__host__ __device__ inline size_t grad_lds_bytes(int ncam_cap) {
    return ((size_t)ncam_cap * (GRAD_REC + 9) + (size_t)GRAD_ROW_BUFFERS * ((size_t)GRAD_LANES * 12 + 16 + GRAD_ROW_SLACK)) * sizeof(double);
}

// launches (grad_fused.hip).  camrec: [camera blocks][GRAD_REC]; cstage / pstage: staging arrays of 9 / 3 doubles per slot;
// partial: [nchunks] the chunks' value sums; g: [N]
hipError_t grad_camera_records_launch(hipStream_t s, int grid, const double* x, const int* cam_blocks, int nblocks, double* camrec);
hipError_t grad_fused_launch(hipStream_t s, int grid, size_t dyn, const GradTables& T, const double* x, const double2* obs,
                             const double* camrec, double* cstage, double* pstage, double* partial, double* g);
hipError_t grad_combine_launch(hipStream_t s, int grid, const GradTables& T, const double* cstage, const double* pstage, double* g);
// the value alone, chunk by chunk: the same per-chunk sums as grad_fused_launch leaves in `partial` (same bits)
hipError_t eval_chunks_launch(hipStream_t s, int grid, const ProblemView& P, int nf, const int* fac, double* partial);

}  // namespace rdis_hip
