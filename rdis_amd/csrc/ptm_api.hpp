// ptm_api.hpp -- what the host side (rdis_hip.hip) needs of the point-major streaming solver (solver_ptm.hpp):
// layout constants, the LDS a launch asks for, and the launches.  The kernels themselves are compiled in a
// translation unit of their own (ptm_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>
#include "device_views.hpp"

namespace rdis_hip {

struct SmallCoopState;

constexpr int PT_REC = 6;    // doubles per point record: p, xi of the block's three variables
constexpr int PT_BND = 6;    // ... its bounds: lo[3], hi[3] -- floats rounded inward (PB), exact doubles (PE)
#ifndef RDIS_PTM_BLK
#define RDIS_PTM_BLK 2
#endif
constexpr int PTM_BLK = RDIS_PTM_BLK;  // slots of a point block asked for (and evaluated) together; pm_cam / pm_obs are padded by 64 (PTM_BLK - 1) entries
constexpr int PTM_CAM_VECTORS = 7;     // LDS vectors over the camera slots: Pv, XI, LO, HI, X and g, h of the Polak-Ribiere recurrence
constexpr int PTM_PAIR_MAX_THREADS = 512;   // the largest workgroup whose gradient rounds may stage two slots (solver_ptm.hpp: PAIR)
constexpr int PTM_SPREAD = 16;         // a component's wave-chunks of equal slot count are dealt out over this many runs of the camera-sorted order
constexpr int PTM_MAX_GROUP = 16;      // workgroups per component of a launch of many groups (SMALL_COOP_ENTRIES / 12 waves, rounded down to a power of two)
constexpr int PTM_MAX_CAMERAS = 4095;  // camera blocks per component (twelve bits of a factor's slot word; two bytes in the trial stream)

// A camera block's ten LDS slots: [tx ty tz f k1 k2 | rx ry rz | pad] -- the six values every factor reads first, from a
// 16-byte boundary (three 16-byte reads; ds_read2_b64 pairs move half as many bytes per LDS cycle, MI355X_MICROARCH.md LDS);
// variable k of the block (reference order r, t, f, k1, k2: BundleAdjustmentCommon.h:36-59) stands at slot ptm_slot_of(k).
// Rotation records: seven doubles from a 16-byte boundary, stride ten; a camera's trial records (factors.hpp: CAM_TRIAL = 16
// doubles) at stride eighteen.  Every stride an ODD number of 16-byte units: a 16-byte LDS read serves sixteen lanes a cycle
// when their units differ mod 16, and lanes that read DIFFERENT cameras at a stride of 8 units (16 doubles) all stand on one.
constexpr int PTM_CS = 10, PTM_ROT0 = 6, PTM_RS = 10, PTM_TS = 18;
__host__ __device__ inline int ptm_slot_of(int k) { return k < 3 ? PTM_ROT0 + k : k - 3; }
__host__ __device__ inline int ptm_var_of(int slot) { return slot < 6 ? slot + 3 : slot < 9 ? slot - PTM_ROT0 : -1; }   // -1: the pad

// A round table's length in 16-bit words: the first staging row of each of the component's ncb cameras and the end of
// the last (ncb + 1 values), rounded up to whole 32-bit words.
__host__ __device__ inline int ptm_round_stride(int ncb) { return (ncb + 2) & ~1; }

// LDS of a workgroup: [7 vectors of 10 ncb_cap camera slots][10 ncb_cap rotation records][2 x 18 ncb_cap trial records (factors.hpp)][(threads + 1) x 9 staged camera
// partials][two round tables][10 ncb_cap free indices (int)]
// (round_slots: the slots a gradient round stages, 1 or 2 -- solver_ptm.hpp: rs)
__host__ __device__ inline size_t ptm_bytes_for(int ncb, int threads, int round_slots = 1) {
    return (size_t)ncb * PTM_CS * (PTM_CAM_VECTORS * sizeof(double) + sizeof(int)) + (size_t)ncb * (PTM_RS + 2 * PTM_TS) * sizeof(double) +
           (size_t)(round_slots * threads + 1) * 9 * sizeof(double) + (((size_t)2 * ptm_round_stride(ncb) * sizeof(unsigned short) + 7) & ~(size_t)7) + 64;
}

struct PtmGroupArgs {
    void* st;             // one exchange state per group of the launch: SmallCoopState, or CoopState for a wide group (grid_sync.hpp)
    double* xch;          // [groups][2 K + 2][10 ncb_cap] partial camera sums of a group's workgroups (two gradients' worth) and their totals
    int K, ngroups;       // workgroups per component, components of the launch
    int poll_delay;
    // a wide group with local camera numbering (solver_ptm.hpp: LOCAL; one component a launch): per workgroup a table in `lc` at
    // lc_off[rank] -- {cameras, first chunk, end chunk, 0, local -> component camera [cameras], speaks-for flags [cameras]} --, per
    // component camera the (rank << 8 | local number) pairs that hold it, and the buffer of the cameras' summed gradient entries
    const long long* lc_off;
    const int* lc;
    const int* cr_ptr;
    const int* cr;
    double* tot;
};

// launches (ptm_kernels.hip); rot = ROT_PER_FACTOR / ROT_RECORDS / ROT_CAMFIX, threads = 256 / 512 / 768
hipError_t ptm_launch(int rot, int threads, int grid, size_t dyn, hipStream_t stream, const ProblemView& P, const PlanView& V,
                      int maxiters, double ftol, int ncb_cap);
const void* ptmg_kernel_fn(int rot, int threads, bool wide = false, bool local = false);   // cgd_ptmg_kernel<threads, rot, wide, local> (a cooperative launch by the caller; wide: 512 lanes)
constexpr int PTM_WIDE_THREADS = 512;  // workgroup of a wide group (one per compute unit)
constexpr int PTM_WIDE_MAX_GROUP = 512;  // ... and their number (COOP_MAX_WG)
hipError_t ptm_gather_launch(int grid, hipStream_t stream, int n, const int* jg, const unsigned* fidx, const double2* fobs,
                             short* pcam, double2* pobs);

}  // namespace rdis_hip
