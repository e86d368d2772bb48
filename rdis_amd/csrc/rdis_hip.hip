// rdis_hip.hip -- implementation of the C ABI in include/rdis_hip.h: handle
// management, uploads, plan construction (validation + gather lists) and kernel
// launches.  gfx950 only; no CPU evaluation path exists in this library.
#include "../../include/rdis_hip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <memory>
#include <new>
#include <numeric>
#include <string>
#include <vector>

#include "host_blocks.hpp"
#include "components.hpp"
#include "lm_solver.hpp"
#include "eval_kernels.hpp"
#include "grad_fused.hpp"
#include "refround_api.hpp"
#include "solver_coop.hpp"
#include "solver_lds.hpp"
#include "ptm_api.hpp"
#include "solver_pipe.hpp"
#include "solver_quad.hpp"
#include "solver_stream.hpp"

using namespace rdis_hip;

struct rdis_hip_ctx {
    int device = 0;                   // the device as the caller names it
    int phys = 0;                     // ... and the GPU behind it (the same, except under RDIS_HIP_VIRTUAL_DEVICES: tests)
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int num_cus = 0;
    std::string err;
    // second stream for the batched launch of a plan that also has cooperative launches (they overlap)
    hipStream_t aux = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    // resident-workgroup caps of the cooperative layouts on THIS device (occupancy queries, asked once per context)
    int cap_pipe = -1, cap_coop[3] = {-1, -1, -1};
    int cap_pipe_rr = -1, cap_coop_rr[3] = {-1, -1, -1};   // ... of the reference-rounding instantiations (refround_api.hpp)
    // dynamic LDS a launch may ask for on THIS device: what a compute unit has, less the solvers' static share (at most
    // solver_lds.hpp's LDS_MAX_BYTES, which is gfx950's); components that do not fit go to the solvers that need none
    size_t lds_limit = LDS_MAX_BYTES;
};

namespace {

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    bool owned = true;  // false: a slice of the problem's arena
    bool host = false;  // pinned host memory (hipHostMalloc), mapped into the device's address space
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    DevBuf(DevBuf&& o) noexcept : p(o.p), bytes(o.bytes), owned(o.owned), host(o.host) { o.p = nullptr; o.bytes = 0; }
    ~DevBuf() { release(); }
    void release() { if (p && owned) (void)(host ? hipHostFree(p) : hipFree(p)); p = nullptr; bytes = 0; owned = true; host = false; }
    template <class T> T* as() const { return static_cast<T*>(p); }
};

int fail(rdis_hip_ctx* c, int code, const std::string& msg) {
    if (c) c->err = msg;
    return code;
}
// Several contexts from one thread (OptimizableFunction::setDevices): every entry point must make its context's device current
// before it allocates or launches.  A box with ONE GPU cannot show a forgotten one -- so RDIS_HIP_VIRTUAL_DEVICES=n (tests) makes
// the library offer n devices that all stand on GPU 0, remember which of them the calling thread made current last
// (make_current), and refuse every HIP call issued for a context while another context's device is the current one.
inline int virtual_devices() {
    static const int n = [] { const char* e = std::getenv("RDIS_HIP_VIRTUAL_DEVICES"); return e ? std::max(0, std::atoi(e)) : 0; }();
    return n;
}
inline int& current_logical_device() { static thread_local int d = -1; return d; }
inline hipError_t make_current(const rdis_hip_ctx* c) { current_logical_device() = c->device; return hipSetDevice(c->phys); }
#define HIPCHK(ctx, expr)                                                                    \
    do {                                                                                     \
        if (virtual_devices() > 0 && (ctx) != nullptr && current_logical_device() != (ctx)->device)  \
            return fail((ctx), RDIS_HIP_EDEVICE, std::string(#expr) + ": issued while device " + std::to_string(current_logical_device()) + \
                        " is current, for a context of device " + std::to_string((ctx)->device) + " (an entry point that did not make its device current)"); \
        hipError_t e__ = (expr);                                                             \
        if (e__ != hipSuccess)                                                               \
            return fail((ctx), e__ == hipErrorOutOfMemory ? RDIS_HIP_ENOMEM : RDIS_HIP_EDEVICE, \
                        std::string(#expr) + ": " + hipGetErrorString(e__));                 \
    } while (0)

// Every entry point makes its context's device the calling thread's current one first: a host that drives several
// contexts from one thread (rdis::OptimizableFunction::setDevices) would otherwise allocate, create events and launch
// on whichever device its previous call left current.
#define USE_DEVICE(ctx)                                        \
    do {                                                       \
        hipError_t d__ = make_current(ctx);                    \
        HIPCHK((ctx), d__);                                    \
    } while (0)

int dalloc(rdis_hip_ctx* c, DevBuf& b, size_t bytes) {
    b.release();
    if (bytes == 0) bytes = 8;
    HIPCHK(c, hipMalloc(&b.p, bytes));
    b.bytes = bytes;
    b.owned = true;
    return 0;
}
// pinned host memory, mapped into the device's address space (the same pointer on both sides)
int halloc(rdis_hip_ctx* c, DevBuf& b, size_t bytes) {
    b.release();
    if (bytes == 0) bytes = 8;
    HIPCHK(c, hipHostMalloc(&b.p, bytes, hipHostMallocDefault));
    b.bytes = bytes;
    b.owned = true;
    b.host = true;
    return 0;
}
template <class T>
int upload(rdis_hip_ctx* c, DevBuf& b, const T* src, size_t n) {
    int rc = dalloc(c, b, n * sizeof(T));
    if (rc) return rc;
    if (n) HIPCHK(c, hipMemcpyAsync(b.p, src, n * sizeof(T), hipMemcpyHostToDevice, c->stream));
    return 0;
}
template <class T>
int upload(rdis_hip_ctx* c, DevBuf& b, const std::vector<T>& v) { return upload(c, b, v.data(), v.size()); }
inline int upload(rdis_hip_ctx* c, DevBuf& b, const ivec& v) { return upload(c, b, v.data(), v.size()); }

int grid_for(const rdis_hip_ctx* c, long long work, int threads) {
    long long blocks = (work + threads - 1) / threads;
    const long long cap = (long long)std::max(1, c->num_cus) * 8;  // grid-stride beyond 8 blocks/CU
    return (int)std::max(1ll, std::min(blocks, cap));
}

}  // namespace

struct VarMark { int stamp, owner, gidx, pad; };

// the fused gradient pass's tables for one factor list (grad_fused.hpp), kept by the problem
struct GradPlan {
    unsigned long long key0 = 0, key1 = 0, used = 0;   // two hashes of the id list (0, 0: all factors); last use
    int64_t nf = 0;
    DevBuf ints, shorts, fac;      // one int32 block, one 16-bit block, the list as int32 (explicit lists)
    DevBuf cstage, pstage, partial;
    std::vector<int64_t> ids;      // the list itself (explicit lists): compared on a hash hit -- two lists with the same hashes must not share tables
    size_t device_bytes() const { return ints.bytes + shorts.bytes + fac.bytes + cstage.bytes + pstage.bytes + partial.bytes; }
    rdis_hip::GradTables T{};
};

struct rdis_hip_problem {
    rdis_hip_ctx* ctx = nullptr;
    int kind = KIND_BA;
    int64_t N = 0, F = 0, nnz = 0;
    int each_rounding = 0;             // rdis_hip_set_factor_rounding: how eval_each / grad_each_ba round (1: like the reference's build)
    DevBuf x, lo, hi, cam, pt, obs, coeff, rowptr, vid, expo, cons, sine;
    DevBuf useexp;                     // per factor: value = coeff * exp(-product) (rdis_hip_nlp_set_exponential)
    std::vector<uint8_t> h_useexp;     // empty: no factor has it
    DevBuf cam_blocks, xrot;                        // distinct camera blocks; their rotation records (shadow of x)
    int64_t ncam_blocks = 0;
    ivec h_block_of;                    // [N] first variable id of the camera block a variable belongs to, -1 = none
    ivec h_ptblock_of;                  // [N] ... of the point block, -1 = none (valid when ncam_blocks > 0)
    ivec h_blk_stamp, h_blk_idx;        // [N] scratch of the slot tables (solver_lds.hpp), valid by stamp
    ivec h_cam, h_pt, h_rowptr, h_vid;  // host copies for plan building
    // scratch for the eval entry points
    DevBuf gfac, partial, scalar, tmp_idx, tmp_val, tmp_out, g_all;
    DevBuf all_v2s_ptr, all_v2s_idx;  // gather lists for "all factors"
    bool have_all_v2s = false;
    // fused value + gradient (bundle adjustment with disjoint blocks): per-call camera records, the lists' tables
    DevBuf camrec;
    ivec h_cam_ord;                   // [N] ordinal of the camera block that starts at a variable id, -1 = none
    std::vector<std::unique_ptr<GradPlan>> grad_plans;
    unsigned long long grad_tick = 0;
    // shared by the plans of this problem (solves on a context are serialised)
    DevBuf dir, coop_state, coop_timing;   // search direction by variable id (kept zero between solves), ...
    int coop_state_gen = 0;                // bumped whenever coop_state moves: plans re-derive the pointers they baked in
    DevBuf arena;                          // memory of the transient plan of rdis_hip_cgd_batch
    DevBuf stage_x;                        // pinned staging of a resident plan's start point (rdis_hip_plan_set_start)
    hipEvent_t stage_ev = nullptr;         // ... recorded behind its copy: the next use waits for it
    bool stage_busy = false;
    size_t arena_used = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    struct rdis_hip_plan* last_timed_plan = nullptr;
    ivec h_local, h_owner_stamp, h_fac_stamp;  // validity by stamp: no O(N) clears per call
    std::vector<VarMark> h_mark;           // [N] plan_create's view of a variable: free in which component, where in the free list
    LmWorkspace lm_ws;                     // scratch of rdis_hip_lm_optimize, kept between calls
    CcWorkspace cc_ws;                     // ... and of rdis_hip_components
    ComponentLists comps;                  // result of the last rdis_hip_components call
    DevBuf assigned;
    int stamp = 0;
    ~rdis_hip_problem() { if (ev0) (void)hipEventDestroy(ev0); if (ev1) (void)hipEventDestroy(ev1); if (stage_ev) (void)hipEventDestroy(stage_ev); }

    ProblemView view() const {
        ProblemView v{};
        v.kind = kind; v.N = (int)N; v.F = (int)F;
        v.x = x.as<double>(); v.lo = lo.as<double>(); v.hi = hi.as<double>();
        v.cam = cam.as<int>(); v.pt = pt.as<int>(); v.obs = obs.as<double2>();
        v.xrot = nullptr; v.rot_mode = ROT_PER_FACTOR;
        v.coeff = coeff.as<double>(); v.rowptr = rowptr.as<int>(); v.vid = vid.as<int>();
        v.expo = expo.as<double>(); v.cons = cons.as<double>(); v.sine = sine.as<uint8_t>();
        v.useexp = h_useexp.empty() ? nullptr : useexp.as<uint8_t>();
        return v;
    }
    int64_t nslots() const { return kind == KIND_BA ? 12 * F : nnz; }
    int arity(int f) const { return kind == KIND_BA ? 12 : h_rowptr[f + 1] - h_rowptr[f]; }
    int var_of(int f, int k) const {
        if (kind == KIND_BA) return k < 9 ? h_cam[f] + k : h_pt[f] + (k - 9);
        return h_vid[h_rowptr[f] + k];
    }
    int slot_of(int f, int k) const { return kind == KIND_BA ? 12 * f + k : h_rowptr[f] + k; }
};


struct CoopItem {
    int comp = 0, nwg = 0;
    size_t xi_off = 0;  // where its published search direction starts in the plan's xi_glob
    // offsets into the plan's coop_ints block (one upload for all cooperative components)
    size_t slot_li = 0;   // [12 * m] local free index of each factor slot, -1 = constant
    size_t lane_var = 0;  // [nwg * threads] free variable owned by a lane, -1 = none
    size_t wave_var = 0;  // [nwg * threads / 64] free variable owned by a whole wave, -1 = none
};

struct CoopLaunch {      // cooperative groups that run side by side in one launch
    int first = 0, count = 0, total_wg = 0;   // items [first, first + count) of the plan's list
    DevBuf groups, wg_group;
};

struct StreamItem {
    int comp = 0, nwg = 0, nlong = 0;
    DevBuf long_vars;  // local indices of the variables fed by many partials
};

struct rdis_hip_plan {
    rdis_hip_problem* prob = nullptr;
    bool transient = false;  // lives in the problem's arena (rdis_hip_cgd_batch): one at a time
    size_t dev_bytes = 0;    // device memory of its own (plan_alloc; rdis_hip_plan_device_bytes)
    int64_t ncomp = 0, nfree = 0, nfac = 0, nslots = 0, ngfac = 0;
    // one device block of int32 (order | free_ptr | free_vid | fac_ptr | fac_id | v2s_ptr |
    // slot_base | slot_pos) and one of results (see out_layout)
    DevBuf ints, ws, gfac, xstart, outbuf, objective, trace, vdump;
    size_t off_order = 0, off_free_ptr = 0, off_free_vid = 0, off_fac_ptr = 0, off_fac_id = 0, off_v2s_ptr = 0,
           off_slot_base = 0, off_slot_pos = 0;
    // results block: xout[nfree] fret[nc] delta[nc] (f64) | nfeval[nc] ngeval[nc] (i64) | iters[nc] status[nc] trace_n[nc] (i32)
    size_t out_bytes = 0;
    cvec h_out;
    ivec h_blk;   // host image of `ints` (kept alive: its upload is asynchronous)
    ivec h_order, h_fac_ptr, h_free_ptr, h_free_vid, h_fac_id, h_v2s_ptr;
    ivec h_slot_li;           // [nslots] per listed factor slot: index of its variable within its component, -1 = constant (may be dropped: plan_create)
    bool have_start = false;
    // which components go where (rebuilt when an option changes)
    bool partition_dirty = true;
    int coop_state_gen = -1;          // generation of the problem's exchange-state buffer this partition points into
    std::vector<CoopItem> coop;
    std::vector<CoopLaunch> coop_launches;
    ivec h_coop_ints;   // host image of coop_ints (kept: the upload is asynchronous)
    DevBuf coop_ints;
    std::vector<std::vector<CoopGroup>> h_coop_groups;
    std::vector<ivec> h_coop_wg;
    std::vector<StreamItem> stream;
    ivec h_rest;
    DevBuf rest_order, xi_glob, queue;
    int group_blocks4 = 0, group_blocks16 = 0;   // resident blocks of the tiny-component kernels
    int tiny_max_blocks = 0;                     // option (tests): cap on their grid, 0 = what is resident
    // options
    int block_threads = 0;
    int64_t coop_min_factors = 4096;  // cooperative solver from this many factors ...
    int coop_max_components = 48;     // ... for at most this many components per plan (packed two or more to a launch); or ...
    int64_t coop_group_min_factors = 256;  // ... for every component of at least this many factors when all their groups fit the device at once
    int rest_tiny = 0;                // the first rest_tiny entries of the batch list run on the quad / wave solver
    int rest_lds = 0;                 // the LAST rest_lds entries run on the LDS-resident solver (solver_lds.hpp)
    int rest_ptm = 0;                 // the rest_ptm entries before them on the point-major streaming solver (solver_ptm.hpp)
    int ptm_stream = 1;               // option "ptm_stream": 0 = never, 1 = components too large for the LDS, 2 = every component
    int ptm_threads = 0;              // option "ptm_threads": its workgroup size, 0 = auto
    int ptm_ncb_cap = 0, ptm_rot_mode = ROT_PER_FACTOR;
    int64_t pm_blocks = 0, pm_entries = 0;
    ivec h_pm_jg;
    DevBuf pm_rec, pm_gh, pm_cbox, pm_bex, pm_cam, pm_obs;
    int64_t pm_cptr_len = 0;          // entries of pm_cptr (a component's wave-chunks + 1)
    // the gradient's round lists (solver_ptm.hpp), built for one workgroup size and group size at a time (ptm_build_rounds)
    DevBuf pm_rounds, pm_rd_off, pm_rd_n, pm_grow, pm_segs, pm_sg_off;
    int rounds_threads = 0, rounds_K = 0, rounds_slots = 1;   // (slots a round stages: ptm_round_slots_for)
    int ptm_round_slots = 0;          // option "ptm_round_slots": 0 = two where the LDS holds their staging rows, 1, 2
    // ... shared by several workgroups each (cgd_ptmg_kernel) when a launch has fewer components than compute units
    int ptm_group = 0;                // option "ptm_group": 0 = auto, 1 = never, k = k workgroups per component
    int ptm_last_group = 1;           // what the last solve used (rdis_hip_plan_debug_counters has no slot for it: get_option)
    int ptm_last_threads = 0;        // ... and their lanes
    int64_t ls_total_chunks = 0;      // gradient chunks of all slot tables
    int64_t ptm_min_points = 0;       // the smallest streaming component's point blocks
    DevBuf ptm_xch, ptm_state;
    size_t off_pm_pt0 = 0, off_pm_ch0 = 0, off_pm_cptr = 0, off_pm_jg = 0;
    int lds_resident = 1;             // option "lds_resident": 0 = never
    int lds_rot = -1;                 // option "lds_rot": rotation records in that solver, -1 = auto, 0 = per factor, 1 = records
    int lds_threads = 0;              // option "lds_threads": its workgroup size, 0 = auto
    int lds_camera_sums = 1;          // option "lds_camera_sums": 0 = camera partials through gfac[] like the plain batch solver
    int lds_matrix = 0;               // option "lds_matrix": 1 = line-search trials in matrix form where the cameras' trial records fit the LDS too (solver_lds.hpp);
                                      // measured slower there (a camera's records are a lone lane's chain in front of every trial: 1000 x 2048 factors 16.6 against 10.7 ms)
    // the dynamic LDS of the LDS-resident launch, and whether its trials run in matrix form (the records must fit behind the other arrays)
    bool lds_matrix_on(const rdis_hip_ctx* c) const {
        return lds_matrix != 0 && !emulate_stale && factor_rounding != 1 &&
               lds_matrix_offset(lds_bytes_for(lds_ns_cap, lds_ncb_cap, lds_chunk_cap)) + lds_matrix_bytes(lds_ncb_cap) <= c->lds_limit;
    }
    size_t lds_dyn_bytes(const rdis_hip_ctx* c) const {
        const size_t base = lds_bytes_for(lds_ns_cap, lds_ncb_cap, lds_chunk_cap);
        return lds_matrix_on(c) ? lds_matrix_offset(base) + lds_matrix_bytes(lds_ncb_cap) : base;
    }
    int emulate_stale = 0;            // option "emulate_stale_cache": the reference's factor cache, emulated (solver_lds.hpp; that solver only)
    int factor_rounding = -1;         // option "factor_rounding": -1 = auto (the cooperative solvers round like the reference's build -- it costs them 4 % --, the batch
                                      // solvers use fused multiply-adds), 0 = fused multiply-adds everywhere, 1 = the reference's rounding (and, in the LDS-resident
                                      // solver, its association of a trial's slope) everywhere: refround_api.hpp
    bool coop_reference_rounding() const { return factor_rounding != 0; }
    bool batch_reference_rounding() const { return factor_rounding == 1; }
    DevBuf st_ev, st_val;
    bool ptm_wide_wanted = false;     // prepare_partition: a large component joined the batch list for a wide point-major group
    bool ptm_wide_last = false;       // the last solve ran its point-major launch as wide groups
    // a wide group with LOCAL camera numbering (solver_ptm.hpp: a component with more cameras than the LDS holds; one such component a plan)
    int ptm_local_cameras = -1;       // option "ptm_local_cameras": -1 = where the cameras do not fit, 0 = never, 1 = for every wide component (tests)
    bool ptm_local = false, ptm_local_off = false;   // the plan has such a component; it was tried and did not fit (then never again)
    int ptm_local_K = 0, ptm_local_comp = -1;
    ivec h_lc, h_cr_ptr, h_cr, h_wg_chunk0;          // PtmGroupArgs' tables; the workgroups' chunk ranges
    std::vector<long long> h_lc_off;
    std::vector<short> h_pm_lcam;                    // entry of the point-major order -> its camera's number in the owning workgroup
    DevBuf lc_dev, lc_off_dev, cr_ptr_dev, cr_dev, ptm_tot;
    DevBuf seq_val, seq_ab;           // the parity option's buffers (PlanView::seq_val, seq_ab), allocated at the first solve that needs them
    int lds_ns_cap = 0, lds_ncb_cap = 0, lds_chunk_cap = 0, lds_rot_mode = ROT_PER_FACTOR;
    int64_t lds_max_factors = 0;
    DevBuf lds_ints, lds_obs;
    ivec h_lds_ints;
    size_t off_ls_ptr = 0, off_ls_vid = 0, off_ls_free = 0, off_ls_ncb = 0, off_ls_fidx = 0, off_ls_gperm = 0, off_ls_gptr = 0;
    int tiny_group = 4;               // ... with this many lanes per component (4 or 16)
    int64_t row_min_components = 4096; // option: sixteen lanes each from this many tiny components (below: a workgroup each)
    int quad_max_vars = QUAD_MAX_VARS; // option "quad_max_vars": 0 = never use the quad solver
    int64_t quad_min_components = 16384;  // ... and only for at least this many tiny components
    int rest_rot_mode = ROT_PER_FACTOR; // how the factors of the batch list get their camera rotations (device_views.hpp)
    int overlap_batch = 1;            // option: run the batched launch concurrently with cooperative launches
    int camera_records = 1;           // option "camera_records": 0 = every factor forms its rotation itself, 1 = auto, 2 = records wherever possible
    size_t off_cb_ptr = 0, off_cb = 0, off_cb_li = 0;
    int coop_workgroups = 0, coop_threads = 256, coop_poll_delay = 16;
    int coop_speculate = 1;           // option: guesses at the following trial steps ride along with every line-search trial
    int coop_pipeline = 1;            // option: cooperative groups with a control wave of their own (solver_pipe.hpp); 0 = solver_coop.hpp
    bool use_pipe = false;            // decided by prepare_partition: the plan's groups fit the pipelined layout (fewer factor lanes per workgroup)
    bool pipelined() const { return use_pipe; }
    int coop_lanes() const { return pipelined() ? PIPE_LANES : coop_threads; }   // factor lanes per workgroup of a cooperative group
    bool force_stream = false;        // send large components to the streaming grid solver even if they fit the register-resident one
    int trace_records = 0;
    int dump_iters = 0;
    int last_launches = 0;
    bool timed = false;

    const int* ip(size_t off) const { return ints.as<int>() + off; }
    double* out_f64(size_t idx) const { return outbuf.as<double>() + idx; }
    PlanView view() const {
        PlanView v{};
        const size_t nc = (size_t)ncomp;
        v.ncomp = (int)ncomp;
        v.order = rest_order.as<int>();
        v.free_ptr = ip(off_free_ptr); v.free_vid = ip(off_free_vid);
        v.fac_ptr = ip(off_fac_ptr); v.fac_id = ip(off_fac_id);
        v.v2s_ptr = ip(off_v2s_ptr); v.slot_base = ip(off_slot_base); v.slot_pos = ip(off_slot_pos);
        v.cb_ptr = ip(off_cb_ptr); v.cb = ip(off_cb); v.cb_li = ip(off_cb_li);
        const int* li = lds_ints.as<int>();
        v.ls_ptr = li + off_ls_ptr; v.ls_vid = li + off_ls_vid; v.ls_free = li + off_ls_free; v.ls_ncb = li + off_ls_ncb;
        v.ls_fidx = reinterpret_cast<const unsigned*>(li + off_ls_fidx);
        v.ls_obs = lds_obs.as<double2>();
        v.ls_gperm = li + off_ls_gperm; v.ls_gptr = li + off_ls_gptr;
        v.ls_cam_gfac = lds_camera_sums ? 0 : 1;
        v.ls_matrix = lds_matrix_on(prob->ctx) ? 1 : 0;
        v.st_ev = emulate_stale ? st_ev.as<int>() : nullptr; v.st_val = emulate_stale ? st_val.as<double>() : nullptr;
        v.pm_pt0 = li + off_pm_pt0; v.pm_ch0 = li + off_pm_ch0; v.pm_cptr = li + off_pm_cptr;
        v.pm_rec = pm_rec.as<double>(); v.pm_gh = pm_gh.as<double>(); v.pm_cbox = pm_cbox.as<float>(); v.pm_bex = pm_bex.as<double>(); v.pm_cam = pm_cam.as<short>(); v.pm_obs = pm_obs.as<double2>();
        v.pm_grow = pm_grow.as<unsigned short>(); v.pm_rounds = pm_rounds.as<unsigned short>(); v.pm_rd_off = pm_rd_off.as<long long>(); v.pm_rd_n = pm_rd_n.as<int>(); v.pm_round_slots = rounds_slots;
        v.pm_segs = pm_segs.as<int>(); v.pm_sg_off = pm_sg_off.as<long long>();
        v.seq_val = seq_val.as<double>(); v.seq_ab = seq_ab.as<double>(); v.seq_n = (int)nfree;
        v.timing = prob->coop_timing.as<long long>();
        v.ws = ws.as<double>(); v.dir = prob->dir.as<double>(); v.gfac = gfac.as<double>();
        v.xstart = xstart.as<double>();
        v.xout = out_f64(0);
        v.fret = out_f64((size_t)nfree); v.delta = out_f64((size_t)nfree + nc);
        long long* i64 = reinterpret_cast<long long*>(out_f64((size_t)nfree + 2 * nc));
        v.nfeval = i64; v.ngeval = i64 + nc;
        int* i32 = reinterpret_cast<int*>(i64 + 2 * nc);
        v.iters = i32; v.status = i32 + nc; v.trace_n = i32 + 2 * nc;
        v.trace = trace_records > 0 ? trace.as<double>() : nullptr;
        v.trace_cap = trace_records;
        v.vdump = dump_iters > 0 ? vdump.as<double>() : nullptr;
        v.dump_iters = dump_iters;
        return v;
    }
};


// =====================================================================================
// context
// =====================================================================================
extern "C" int rdis_hip_abi_version(void) { return RDIS_HIP_ABI_VERSION; }

extern "C" int rdis_hip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return (n > 0 && virtual_devices() > 0) ? virtual_devices() : n;
}

extern "C" int rdis_hip_create(int device, rdis_hip_ctx** out) {
    if (!out) return RDIS_HIP_EINVAL;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= (virtual_devices() > 0 ? virtual_devices() : n)) return RDIS_HIP_EDEVICE;
    rdis_hip_ctx* c = new (std::nothrow) rdis_hip_ctx;
    if (!c) return RDIS_HIP_ENOMEM;
    c->device = device;
    c->phys = virtual_devices() > 0 ? 0 : device;
    hipDeviceProp_t prop;
    if (make_current(c) != hipSuccess || hipGetDeviceProperties(&prop, c->phys) != hipSuccess ||
        hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        delete c;
        return RDIS_HIP_EDEVICE;
    }
    c->own_stream = true;
    c->num_cus = prop.multiProcessorCount;
    if (prop.maxSharedMemoryPerMultiProcessor > 4096)
        c->lds_limit = std::min<size_t>((size_t)LDS_MAX_BYTES, (size_t)prop.maxSharedMemoryPerMultiProcessor - 4096);
    *out = c;
    return 0;
}

extern "C" void rdis_hip_destroy(rdis_hip_ctx* c) {
    if (!c) return;
    (void)make_current(c);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    if (c->aux) (void)hipStreamDestroy(c->aux);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    delete c;
}

extern "C" const char* rdis_hip_last_error(const rdis_hip_ctx* c) { return c ? c->err.c_str() : "null context"; }

extern "C" int rdis_hip_set_stream(rdis_hip_ctx* c, void* s) {
    if (!c) return RDIS_HIP_EINVAL;
    USE_DEVICE(c);
    if (c->own_stream && c->stream) { (void)hipStreamSynchronize(c->stream); (void)hipStreamDestroy(c->stream); }
    if (s) { c->stream = (hipStream_t)s; c->own_stream = false; }
    else {
        HIPCHK(c, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        c->own_stream = true;
    }
    return 0;
}

extern "C" int rdis_hip_synchronize(rdis_hip_ctx* c) {
    if (!c) return RDIS_HIP_EINVAL;
    USE_DEVICE(c);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int rdis_hip_copy_to_host(rdis_hip_ctx* c, void* dst, const void* src, int64_t bytes) {
    if (!c || bytes < 0 || (bytes && (!dst || !src))) return RDIS_HIP_EINVAL;
    if (bytes == 0) return 0;
    USE_DEVICE(c);
    HIPCHK(c, hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

// =====================================================================================
// problems
// =====================================================================================
namespace {
int upload_common(rdis_hip_ctx* c, rdis_hip_problem* p, int64_t nvars, const double* x0, const double* lo,
                  const double* hi) {
    int rc;
    if ((rc = upload(c, p->x, x0, (size_t)nvars))) return rc;
    if ((rc = upload(c, p->lo, lo, (size_t)nvars))) return rc;
    if ((rc = upload(c, p->hi, hi, (size_t)nvars))) return rc;
    if ((rc = dalloc(c, p->scalar, 64))) return rc;
    return 0;
}
}  // namespace

extern "C" int rdis_hip_upload_ba(rdis_hip_ctx* c, int64_t nvars, const double* x0, const double* lo,
                                  const double* hi, int64_t nfac, const int64_t* cam_vid0,
                                  const int64_t* pt_vid0, const double* obs, rdis_hip_problem** out) {
    if (!c || !out || nvars < 0 || nfac < 0 || (nvars && (!x0 || !lo || !hi)) ||
        (nfac && (!cam_vid0 || !pt_vid0 || !obs)))
        return fail(c, RDIS_HIP_EINVAL, "upload_ba: bad argument");
    if (nvars >= (1ll << 31) - 16 || nfac >= ((1ll << 31) - 16) / 12) return fail(c, RDIS_HIP_ERANGE, "upload_ba: too large for int32 device indices");
    USE_DEVICE(c);
    rdis_hip_problem* p = new (std::nothrow) rdis_hip_problem;
    if (!p) return fail(c, RDIS_HIP_ENOMEM, "upload_ba: host allocation");
    p->ctx = c; p->kind = KIND_BA; p->N = nvars; p->F = nfac;
    p->h_cam.resize((size_t)nfac); p->h_pt.resize((size_t)nfac);
    for (int64_t i = 0; i < nfac; ++i) {
        if (cam_vid0[i] < 0 || cam_vid0[i] + 9 > nvars || pt_vid0[i] < 0 || pt_vid0[i] + 3 > nvars) {
            delete p;
            return fail(c, RDIS_HIP_EINVAL, "upload_ba: variable id out of range in factor " + std::to_string(i));
        }
        p->h_cam[(size_t)i] = (int)cam_vid0[i];
        p->h_pt[(size_t)i] = (int)pt_vid0[i];
    }
    int rc = upload_common(c, p, nvars, x0, lo, hi);
    if (!rc) rc = upload(c, p->cam, p->h_cam);
    if (!rc) rc = upload(c, p->pt, p->h_pt);
    if (!rc) rc = upload(c, p->obs, obs, (size_t)(2 * nfac));
    ivec blocks;   // (alive until the copies below have completed)
    {   // the distinct camera blocks, for the rotation records of launches that leave the cameras constant
        p->h_block_of.assign((size_t)nvars, -1);
        blocks = p->h_cam;
        std::sort(blocks.begin(), blocks.end());
        blocks.erase(std::unique(blocks.begin(), blocks.end()), blocks.end());
        // blocks that overlap without coinciding would share record slots: no records then
        bool disjoint = true;
        for (size_t i = 1; i < blocks.size(); ++i) disjoint = disjoint && blocks[i] - blocks[i - 1] >= 9;
        for (int cb : blocks)
            for (int k = 0; k < 9; ++k) p->h_block_of[(size_t)cb + k] = cb;
        for (int64_t i = 0; i < nfac && disjoint; ++i)   // a point block inside a camera block: likewise
            for (int k = 0; k < 3; ++k) disjoint = disjoint && p->h_block_of[(size_t)p->h_pt[(size_t)i] + k] < 0;
        if (disjoint) {   // the point blocks likewise: coinciding or at least 3 apart
            p->h_ptblock_of.assign((size_t)nvars, -1);
            for (int64_t i = 0; i < nfac && disjoint; ++i) {
                const int q = p->h_pt[(size_t)i];
                for (int k = 0; k < 3; ++k) {
                    int& o = p->h_ptblock_of[(size_t)q + k];
                    disjoint = disjoint && (o < 0 || o == q);
                    o = q;
                }
            }
        }
        p->ncam_blocks = disjoint ? (int64_t)blocks.size() : 0;
        if (!rc && p->ncam_blocks > 0) {
            rc = upload(c, p->cam_blocks, blocks);
            if (!rc) rc = dalloc(c, p->xrot, (size_t)nvars * sizeof(double));
        }
    }
    if (!rc && hipStreamSynchronize(c->stream) != hipSuccess) rc = fail(c, RDIS_HIP_EDEVICE, "upload_ba: sync");
    if (rc) { delete p; return rc; }
    *out = p;
    return 0;
}

extern "C" int rdis_hip_upload_nlp(rdis_hip_ctx* c, int64_t nvars, const double* x0, const double* lo,
                                   const double* hi, int64_t nfac, const double* coeff,
                                   const int64_t* rowptr, const int64_t* vid, const double* expo,
                                   const double* cons, const uint8_t* sine, rdis_hip_problem** out) {
    if (!c || !out || nvars < 0 || nfac < 0 || (nvars && (!x0 || !lo || !hi)) || !rowptr || (nfac && !coeff))
        return fail(c, RDIS_HIP_EINVAL, "upload_nlp: bad argument");
    const int64_t nnz = rowptr[nfac];
    if (nnz < 0 || (nnz && (!vid || !expo || !cons || !sine))) return fail(c, RDIS_HIP_EINVAL, "upload_nlp: bad CSR");
    if (nvars >= (1ll << 31) - 16 || nfac >= (1ll << 31) - 16 || nnz >= (1ll << 31) - 16)
        return fail(c, RDIS_HIP_ERANGE, "upload_nlp: too large for int32 device indices");
    USE_DEVICE(c);
    rdis_hip_problem* p = new (std::nothrow) rdis_hip_problem;
    if (!p) return fail(c, RDIS_HIP_ENOMEM, "upload_nlp: host allocation");
    p->ctx = c; p->kind = KIND_NLP; p->N = nvars; p->F = nfac; p->nnz = nnz;
    p->h_rowptr.resize((size_t)nfac + 1); p->h_vid.resize((size_t)nnz);
    for (int64_t i = 0; i <= nfac; ++i) {
        if (rowptr[i] < 0 || (i && rowptr[i] < rowptr[i - 1])) { delete p; return fail(c, RDIS_HIP_EINVAL, "upload_nlp: rowptr not monotone"); }
        p->h_rowptr[(size_t)i] = (int)rowptr[i];
    }
    for (int64_t k = 0; k < nnz; ++k) {
        if (vid[k] < 0 || vid[k] >= nvars) { delete p; return fail(c, RDIS_HIP_EINVAL, "upload_nlp: variable id out of range"); }
        p->h_vid[(size_t)k] = (int)vid[k];
    }
    int rc = upload_common(c, p, nvars, x0, lo, hi);
    if (!rc) rc = upload(c, p->coeff, coeff, (size_t)nfac);
    if (!rc) rc = upload(c, p->rowptr, p->h_rowptr);
    if (!rc) rc = upload(c, p->vid, p->h_vid);
    if (!rc) rc = upload(c, p->expo, expo, (size_t)nnz);
    if (!rc) rc = upload(c, p->cons, cons, (size_t)nnz);
    if (!rc) rc = upload(c, p->sine, sine, (size_t)nnz);
    if (!rc && hipStreamSynchronize(c->stream) != hipSuccess) rc = fail(c, RDIS_HIP_EDEVICE, "upload_nlp: sync");
    if (rc) { delete p; return rc; }
    *out = p;
    return 0;
}

extern "C" int rdis_hip_nlp_set_exponential(rdis_hip_problem* p, const uint8_t* use_exp) {
    if (!p) return RDIS_HIP_EINVAL;
    rdis_hip_ctx* c = p->ctx;
    if (p->kind != KIND_NLP) return fail(c, RDIS_HIP_EINVAL, "nlp_set_exponential: not a nonlinear-product problem");
    USE_DEVICE(c);
    bool any = false;
    for (int64_t i = 0; use_exp && i < p->F; ++i) any |= use_exp[i] != 0;
    if (!any) { p->h_useexp.clear(); return 0; }
    // (a kernel of an earlier call may still read the old flags: the copy is ordered after it on the stream)
    if (p->useexp.bytes < (size_t)p->F) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (int rc = dalloc(c, p->useexp, (size_t)p->F)) return rc;
    }
    p->h_useexp.assign(use_exp, use_exp + p->F);
    HIPCHK(c, hipMemcpyAsync(p->useexp.p, p->h_useexp.data(), (size_t)p->F, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" void rdis_hip_free_problem(rdis_hip_problem* p) {
    if (!p) return;
    (void)make_current(p->ctx);
    (void)hipStreamSynchronize(p->ctx->stream);
    delete p;
}

namespace {
// stage an int64 id list as int32 on the device (nullptr stays nullptr)
int stage_ids(rdis_hip_problem* p, int64_t n, const int64_t* ids, int64_t limit, const int** dev) {
    rdis_hip_ctx* c = p->ctx;
    *dev = nullptr;
    if (!ids) return 0;
    ivec tmp((size_t)n);
    for (int64_t i = 0; i < n; ++i) {
        if (ids[i] < 0 || ids[i] >= limit) return fail(c, RDIS_HIP_EINVAL, "id out of range");
        tmp[(size_t)i] = (int)ids[i];
    }
    if (p->tmp_idx.bytes < (size_t)n * sizeof(int)) { int rc = dalloc(c, p->tmp_idx, (size_t)n * sizeof(int)); if (rc) return rc; }
    HIPCHK(c, hipMemcpyAsync(p->tmp_idx.p, tmp.data(), (size_t)n * sizeof(int), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));  // tmp goes out of scope
    *dev = p->tmp_idx.as<int>();
    return 0;
}
int ensure(rdis_hip_ctx* c, DevBuf& b, size_t bytes) { return b.bytes >= bytes ? 0 : dalloc(c, b, bytes); }
}  // namespace

extern "C" int rdis_hip_set_x(rdis_hip_problem* p, int64_t n, const int64_t* vid, const double* val) {
    if (!p || n < 0 || (n && !val)) return RDIS_HIP_EINVAL;
    rdis_hip_ctx* c = p->ctx;
    if (n == 0) return 0;
    USE_DEVICE(c);
    if (!vid) {
        if (n > p->N) return fail(c, RDIS_HIP_EINVAL, "set_x: n > nvars");
        HIPCHK(c, hipMemcpyAsync(p->x.p, val, (size_t)n * sizeof(double), hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        return 0;
    }
    const int* dv;
    int rc = stage_ids(p, n, vid, p->N, &dv);
    if (rc) return rc;
    if ((rc = ensure(c, p->tmp_val, (size_t)n * sizeof(double)))) return rc;
    HIPCHK(c, hipMemcpyAsync(p->tmp_val.p, val, (size_t)n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    scatter_x_kernel<<<grid_for(c, n, 256), 256, 0, c->stream>>>((int)n, dv, p->tmp_val.as<double>(), p->x.as<double>());
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int rdis_hip_get_x(rdis_hip_problem* p, int64_t n, const int64_t* vid, double* out) {
    if (!p || n < 0 || (n && !out)) return RDIS_HIP_EINVAL;
    rdis_hip_ctx* c = p->ctx;
    if (n == 0) return 0;
    USE_DEVICE(c);
    if (!vid) {
        if (n > p->N) return fail(c, RDIS_HIP_EINVAL, "get_x: n > nvars");
        HIPCHK(c, hipMemcpyAsync(out, p->x.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        return 0;
    }
    const int* dv;
    int rc = stage_ids(p, n, vid, p->N, &dv);
    if (rc) return rc;
    if ((rc = ensure(c, p->tmp_out, (size_t)n * sizeof(double)))) return rc;
    gather_x_kernel<<<grid_for(c, n, 256), 256, 0, c->stream>>>((int)n, dv, p->x.as<double>(), p->tmp_out.as<double>());
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(out, p->tmp_out.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

// =====================================================================================
// batched evaluation
// =====================================================================================
namespace {

// for every variable the gradient slots that feed it, in factor-list order
void build_v2s(const rdis_hip_problem* p, int64_t nf, const int64_t* fac, ivec& ptr,
               ivec& idx) {
    ptr.assign((size_t)p->N + 1, 0);
    for (int64_t i = 0; i < nf; ++i) {
        const int f = (int)(fac ? fac[i] : i);
        for (int k = 0, a = p->arity(f); k < a; ++k) ++ptr[(size_t)p->var_of(f, k) + 1];
    }
    for (int64_t v = 0; v < p->N; ++v) ptr[(size_t)v + 1] += ptr[(size_t)v];
    idx.resize((size_t)ptr[(size_t)p->N]);
    ivec fill(ptr.begin(), ptr.end() - 1);
    for (int64_t i = 0; i < nf; ++i) {
        const int f = (int)(fac ? fac[i] : i);
        for (int k = 0, a = p->arity(f); k < a; ++k) idx[(size_t)fill[(size_t)p->var_of(f, k)]++] = p->slot_of(f, k);
    }
}

int check_list(rdis_hip_problem* p, int64_t nf, const int64_t* fac) {
    if (nf < 0) return fail(p->ctx, RDIS_HIP_EINVAL, "negative factor count");
    if (!fac && nf != p->F) return fail(p->ctx, RDIS_HIP_EINVAL, "fac == NULL requires nf == factor count");
    return 0;
}

// NonlinearProductFactor::computeGradient asserts its factor has no exponential (src/NonlinearProductFactor.cpp:110):
// a gradient or a solve over a list that holds such a factor is refused, not computed from the plain product
int refuse_exponential(rdis_hip_problem* p, int64_t nf, const int64_t* fac, const char* who) {
    if (p->h_useexp.empty()) return 0;
    for (int64_t i = 0; i < nf; ++i) {
        const int64_t f = fac ? fac[i] : i;
        if (f >= 0 && f < p->F && p->h_useexp[(size_t)f])
            return fail(p->ctx, RDIS_HIP_EINVAL, std::string(who) + ": factor " + std::to_string(f) +
                        " is exponential; the reference's gradient asserts it is not (NonlinearProductFactor.cpp:110)");
    }
    return 0;
}

template <bool GRAD>
int launch_eval_sum(rdis_hip_problem* p, int nf, const int* dfac, int blocks) {
    rdis_hip_ctx* c = p->ctx;
    ProblemView V = p->view();
    if (p->kind == KIND_BA)
        eval_sum_kernel<KIND_BA, GRAD><<<blocks, 256, 0, c->stream>>>(V, nf, dfac, p->gfac.as<double>(), p->partial.as<double>());
    else
        eval_sum_kernel<KIND_NLP, GRAD><<<blocks, 256, 0, c->stream>>>(V, nf, dfac, p->gfac.as<double>(), p->partial.as<double>());
    HIPCHK(c, hipGetLastError());
    final_sum_kernel<<<1, 256, 0, c->stream>>>(blocks, p->partial.as<double>(), p->scalar.as<double>());
    HIPCHK(c, hipGetLastError());
    return 0;
}
}  // namespace

namespace {

// Factor lists whose tables the problem keeps: up to GRAD_PLAN_CACHE of them within GRAD_PLAN_BYTES of device memory (least
// recently used goes first).  A caller that asks for the gradients of a level's components in turn (computeGradient(facs, pg)
// per component, as the reference does) cycles through dozens of lists: with room for four (round 5) every call rebuilt and
// reallocated its tables.  A list of at most one chunk needs none (eval_grad_short).
constexpr size_t GRAD_PLAN_CACHE = 64;
constexpr size_t GRAD_PLAN_BYTES = (size_t)1 << 30;

bool fused_path(const rdis_hip_problem* p) { return p->kind == KIND_BA && p->ncam_blocks > 0; }

// The tables of grad_fused.hpp for one factor list, built once per list.  Everything here is index work over the list:
// cut it into chunks and tiles, number a tile's cameras, rank a chunk's entries by camera and by point block (list order
// within a block: the order of the sums), decide per block whether one chunk / tile holds all its listed factors (straight to
// g) or several do (a staging slot each, a block's slots consecutive in chunk / tile order).
int build_grad_plan(rdis_hip_problem* p, int64_t nf64, const int64_t* fac, GradPlan& G) {
    rdis_hip_ctx* c = p->ctx;
    const int nf = (int)nf64, GW = GRAD_LANES;
    const int nchunks = (nf + GW - 1) / GW;
    const size_t N = (size_t)p->N, ncb = (size_t)p->ncam_blocks;
    if (p->h_cam_ord.empty()) {   // (camera blocks are the distinct values of h_cam, ascending: their ordinal = their rank)
        p->h_cam_ord.assign(N, -1);
        ivec blocks(p->h_cam);
        std::sort(blocks.begin(), blocks.end());
        blocks.erase(std::unique(blocks.begin(), blocks.end()), blocks.end());
        for (size_t i = 0; i < blocks.size(); ++i) p->h_cam_ord[(size_t)blocks[i]] = (int)i;
    }
    auto fid_of = [&](int j) { return fac ? (int)fac[j] : j; };
    // how many listed factors read each block
    ivec cnt_cam(ncb, 0), cnt_pt(N, 0);
    for (int j = 0; j < nf; ++j) { const int f = fid_of(j); ++cnt_cam[(size_t)p->h_cam_ord[(size_t)p->h_cam[(size_t)f]]]; ++cnt_pt[(size_t)p->h_pt[(size_t)f]]; }
    // tiles: up to tile_chunks chunks, fewer where their cameras would not fit (a chunk alone always does)
    const int want_tiles = 8 * std::max(1, c->num_cus);
    int tile_chunks = std::max(1, std::min(GRAD_MAX_TILE_CHUNKS, nchunks / want_tiles));
    if (const char* ev = std::getenv("RDIS_HIP_GRAD_TILE_CHUNKS")) tile_chunks = std::max(1, std::min(64, std::atoi(ev)));   // (tuning)
    const int cam_soft = 192;
    const size_t npad = (size_t)nchunks * GW;
    std::vector<unsigned short> sh(3 * npad, GRAD_NO_ENTRY);   // cl | rc | rp
    unsigned short* const cl = sh.data();
    unsigned short* const rc = cl + npad;
    unsigned short* const rp = rc + npad;
    ivec ptv(npad, 0), tile_chunk0, tile_cam0, chunk_cseg0((size_t)nchunks + 1, 0), chunk_pseg0((size_t)nchunks + 1, 0);
    std::vector<int2> tile_cam, cseg, pseg;
    ivec cam_tile(ncb, -1), cam_local(ncb, 0);           // camera ordinal -> tile it was last numbered in, its number there
    ivec tcnt;                                            // listed factors of a tile's camera inside the tile
    std::vector<std::pair<int, int>> cstaged, pstaged;    // (block, index of the table entry to patch), in tile / chunk order
    ivec pt_chunk(N, -1), pt_seg(N, 0);                   // point block -> chunk it was last seen in, its segment there
    ivec segcnt, segrow, seg_of((size_t)GW), cam_seg_chunk, cam_seg;
    int ncam_cap = 1;
    tile_chunk0.push_back(0); tile_cam0.push_back(0);
    for (int ch0 = 0; ch0 < nchunks;) {
        const int tile = (int)tile_chunk0.size() - 1;
        const size_t tc0 = tile_cam.size();
        int ch1 = ch0;
        while (ch1 < nchunks && ch1 - ch0 < tile_chunks) {
            // the cameras this chunk would add
            const size_t before = tile_cam.size();
            const int j0 = ch1 * GW, j1 = std::min(nf, j0 + GW);
            for (int j = j0; j < j1; ++j) {
                const int o = p->h_cam_ord[(size_t)p->h_cam[(size_t)fid_of(j)]];
                if (cam_tile[(size_t)o] != tile) { cam_tile[(size_t)o] = tile; cam_local[(size_t)o] = (int)(tile_cam.size() - tc0); tile_cam.push_back(int2{o, 0}); }
            }
            if (ch1 > ch0 && tile_cam.size() - tc0 > (size_t)cam_soft) {   // too many: the chunk starts the next tile
                for (size_t k = before; k < tile_cam.size(); ++k) cam_tile[(size_t)tile_cam[k].x] = -1;
                tile_cam.resize(before);
                break;
            }
            ++ch1;
        }
        const int nc = (int)(tile_cam.size() - tc0);
        ncam_cap = std::max(ncam_cap, nc);
        tcnt.assign((size_t)nc, 0);
        cam_seg_chunk.assign((size_t)nc, -1); cam_seg.assign((size_t)nc, 0);
        for (int ch = ch0; ch < ch1; ++ch) {
            const int j0 = ch * GW, j1 = std::min(nf, j0 + GW);
            // segments by camera, in order of first appearance; list order within a segment
            segcnt.clear();
            for (int j = j0; j < j1; ++j) {
                const int f = fid_of(j);
                const int l = cam_local[(size_t)p->h_cam_ord[(size_t)p->h_cam[(size_t)f]]];
                cl[j] = (unsigned short)l;
                ptv[(size_t)j] = p->h_pt[(size_t)f];
                ++tcnt[(size_t)l];
                if (cam_seg_chunk[(size_t)l] != ch) { cam_seg_chunk[(size_t)l] = ch; cam_seg[(size_t)l] = (int)segcnt.size(); segcnt.push_back(0); cseg.push_back(int2{l, 0}); }
                seg_of[(size_t)(j - j0)] = cam_seg[(size_t)l];
                ++segcnt[(size_t)cam_seg[(size_t)l]];
            }
            {
                const size_t s0 = cseg.size() - segcnt.size();
                segrow.assign(segcnt.size() + 1, 0);
                for (size_t k = 0; k < segcnt.size(); ++k) { segrow[k + 1] = segrow[k] + segcnt[k]; cseg[s0 + k].y = segrow[k]; }
                for (int j = j0; j < j1; ++j) rc[j] = (unsigned short)segrow[(size_t)seg_of[(size_t)(j - j0)]]++;
                cseg.push_back(int2{0, j1 - j0});
                chunk_cseg0[(size_t)ch + 1] = (int)cseg.size();
            }
            // ... and by point block
            segcnt.clear();
            for (int j = j0; j < j1; ++j) {
                const int q = ptv[(size_t)j];
                if (pt_chunk[(size_t)q] != ch) { pt_chunk[(size_t)q] = ch; pt_seg[(size_t)q] = (int)segcnt.size(); segcnt.push_back(0); pseg.push_back(int2{q, 0}); }
                seg_of[(size_t)(j - j0)] = pt_seg[(size_t)q];
                ++segcnt[(size_t)pt_seg[(size_t)q]];
            }
            {
                const size_t s0 = pseg.size() - segcnt.size();
                segrow.assign(segcnt.size() + 1, 0);
                for (size_t k = 0; k < segcnt.size(); ++k) {
                    segrow[k + 1] = segrow[k] + segcnt[k];
                    pseg[s0 + k].y = segrow[k];
                    // all the block's listed factors in this chunk: its sum goes straight to g (destination = its id, as set)
                    if (segcnt[k] != cnt_pt[(size_t)pseg[s0 + k].x]) pstaged.emplace_back(pseg[s0 + k].x, (int)(s0 + k));
                }
                for (int j = j0; j < j1; ++j) rp[j] = (unsigned short)segrow[(size_t)seg_of[(size_t)(j - j0)]]++;
                pseg.push_back(int2{0, j1 - j0});
                chunk_pseg0[(size_t)ch + 1] = (int)pseg.size();
            }
        }
        for (int l = 0; l < nc; ++l) {
            int2& e = tile_cam[tc0 + (size_t)l];
            const int o = e.x;
            if (tcnt[(size_t)l] == cnt_cam[(size_t)o]) e.y = -1;   // (patched below: the block's first variable id)
            else { e.y = -2; cstaged.emplace_back(o, (int)(tc0 + (size_t)l)); }
        }
        tile_chunk0.push_back(ch1);
        tile_cam0.push_back((int)tile_cam.size());
        ch0 = ch1;
    }
    // camera ordinal -> first variable id
    ivec cam_first(ncb, 0);
    for (size_t v = 0; v < N; ++v) if (p->h_cam_ord[v] >= 0) cam_first[(size_t)p->h_cam_ord[v]] = (int)v;
    for (int2& e : tile_cam) if (e.y == -1) e.y = cam_first[(size_t)e.x];
    // staging slots: a block's partial sums consecutive, in the order they were made
    ivec cs_var, cs_ptr(1, 0), ps_var, ps_ptr(1, 0);
    {
        ivec n_of(ncb, 0), base(ncb, 0);
        for (const auto& e : cstaged) ++n_of[(size_t)e.first];
        for (size_t o = 0; o < ncb; ++o)
            if (n_of[o] > 0) { base[o] = cs_ptr.back(); cs_var.push_back(cam_first[o]); cs_ptr.push_back(cs_ptr.back() + n_of[o]); }
        for (const auto& e : cstaged) tile_cam[(size_t)e.second].y = ~(base[(size_t)e.first]++);
    }
    {
        ivec& n_of = pt_seg;    // (scratch: the per-chunk segment numbers are not needed any more)
        ivec& base = pt_chunk;
        for (const auto& e : pstaged) n_of[(size_t)e.first] = 0;
        for (const auto& e : pstaged) ++n_of[(size_t)e.first];
        ivec blocks;
        for (const auto& e : pstaged) if (n_of[(size_t)e.first] > 0) { blocks.push_back(e.first); n_of[(size_t)e.first] = -n_of[(size_t)e.first]; }
        std::sort(blocks.begin(), blocks.end());
        for (int q : blocks) { base[(size_t)q] = ps_ptr.back(); ps_var.push_back(q); ps_ptr.push_back(ps_ptr.back() - n_of[(size_t)q]); }
        for (const auto& e : pstaged) pseg[(size_t)e.second].x = ~(base[(size_t)e.first]++);
    }
    // variables no listed factor reads
    ivec zvar;
    {
        cvec touched(N, 0);
        for (size_t o = 0; o < ncb; ++o) if (cnt_cam[o] > 0) for (int k = 0; k < 9; ++k) touched[(size_t)cam_first[o] + k] = 1;
        for (size_t v = 0; v < N; ++v) if (cnt_pt[v] > 0) for (int k = 0; k < 3; ++k) touched[v + k] = 1;
        for (size_t v = 0; v < N; ++v) if (!touched[v]) zvar.push_back((int)v);
    }
    // one int32 block, one 16-bit block
    ivec blk;
    auto put = [&](const int* v, size_t n) { const size_t off = blk.size(); blk.insert(blk.end(), v, v + n); if (blk.size() & 1) blk.push_back(0); return off; };
    const size_t o_ptv = put(ptv.data(), ptv.size()), o_tc0 = put(tile_chunk0.data(), tile_chunk0.size()), o_tm0 = put(tile_cam0.data(), tile_cam0.size());
    const size_t o_tcam = put(reinterpret_cast<const int*>(tile_cam.data()), 2 * tile_cam.size());
    const size_t o_cs0 = put(chunk_cseg0.data(), chunk_cseg0.size()), o_cseg = put(reinterpret_cast<const int*>(cseg.data()), 2 * cseg.size());
    const size_t o_ps0 = put(chunk_pseg0.data(), chunk_pseg0.size()), o_pseg = put(reinterpret_cast<const int*>(pseg.data()), 2 * pseg.size());
    const size_t o_csv = put(cs_var.data(), cs_var.size()), o_csp = put(cs_ptr.data(), cs_ptr.size());
    const size_t o_psv = put(ps_var.data(), ps_var.size()), o_psp = put(ps_ptr.data(), ps_ptr.size());
    const size_t o_z = put(zvar.data(), zvar.size());
    int rc_ = upload(c, G.ints, blk);
    if (!rc_) rc_ = upload(c, G.shorts, sh.data(), sh.size());
    if (!rc_ && fac) {
        ivec f32((size_t)nf);
        for (int j = 0; j < nf; ++j) f32[(size_t)j] = (int)fac[j];
        rc_ = upload(c, G.fac, f32);
        if (!rc_) HIPCHK(c, hipStreamSynchronize(c->stream));   // (f32 goes out of scope)
    }
    if (!rc_) rc_ = dalloc(c, G.cstage, (size_t)std::max(cs_ptr.back(), 1) * 9 * sizeof(double));
    if (!rc_) rc_ = dalloc(c, G.pstage, (size_t)std::max(ps_ptr.back(), 1) * 3 * sizeof(double));
    if (!rc_) rc_ = dalloc(c, G.partial, (size_t)std::max(nchunks, 1) * sizeof(double));
    if (rc_) {   // (the uploads above are asynchronous copies out of host vectors that go out of scope with this call)
        (void)hipStreamSynchronize(c->stream);
        return rc_;
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));   // (the host images go out of scope)
    const int* I = G.ints.as<int>();
    const unsigned short* S = G.shorts.as<unsigned short>();
    GradTables& T = G.T;
    T.nf = nf; T.nchunks = nchunks; T.ntiles = (int)tile_chunk0.size() - 1; T.ncam_cap = ncam_cap;
    T.fac = fac ? G.fac.as<int>() : nullptr;
    T.cl = S; T.rc = S + npad; T.rp = S + 2 * npad;
    T.ptv = I + o_ptv; T.tile_chunk0 = I + o_tc0; T.tile_cam0 = I + o_tm0; T.tile_cam = reinterpret_cast<const int2*>(I + o_tcam);
    T.chunk_cseg0 = I + o_cs0; T.cseg = reinterpret_cast<const int2*>(I + o_cseg);
    T.chunk_pseg0 = I + o_ps0; T.pseg = reinterpret_cast<const int2*>(I + o_pseg);
    T.ncs = (int)cs_var.size(); T.nps = (int)ps_var.size(); T.nz = (int)zvar.size();
    T.cs_var = I + o_csv; T.cs_ptr = I + o_csp; T.ps_var = I + o_psv; T.ps_ptr = I + o_psp; T.zvar = I + o_z;
    G.nf = nf64;
    return 0;
}

// the tables of a list: found among the problem's (by two hashes of the ids) or built
int grad_plan_for(rdis_hip_problem* p, int64_t nf, const int64_t* fac, GradPlan** out) {
    unsigned long long k0 = 0, k1 = 0;
    if (fac) {
        k0 = 1469598103934665603ull; k1 = 0x9E3779B97F4A7C15ull;
        for (int64_t i = 0; i < nf; ++i) {
            const unsigned long long v = (unsigned long long)fac[i];
            if (fac[i] < 0 || fac[i] >= p->F) return fail(p->ctx, RDIS_HIP_EINVAL, "id out of range");
            k0 = (k0 ^ v) * 1099511628211ull;
            k1 = (k1 + v + 0x632BE59BD9B4E019ull) * 0xFF51AFD7ED558CCDull; k1 ^= k1 >> 29;
        }
        if (k0 == 0 && k1 == 0) k1 = 1;
    }
    for (auto& g : p->grad_plans)
        if (g->nf == nf && g->key0 == k0 && g->key1 == k1 &&
            (fac ? g->ids.size() == (size_t)nf && std::memcmp(g->ids.data(), fac, (size_t)nf * sizeof(int64_t)) == 0 : g->ids.empty())) {
            g->used = ++p->grad_tick; *out = g.get(); return 0;
        }
    auto held = [&]() { size_t b = 0; for (auto& g : p->grad_plans) b += g->device_bytes(); return b; };
    bool synced = false;
    while (!p->grad_plans.empty() && (p->grad_plans.size() >= GRAD_PLAN_CACHE || held() > GRAD_PLAN_BYTES)) {
        size_t lru = 0;
        for (size_t i = 1; i < p->grad_plans.size(); ++i) if (p->grad_plans[i]->used < p->grad_plans[lru]->used) lru = i;
        if (!synced) { HIPCHK(p->ctx, hipStreamSynchronize(p->ctx->stream)); synced = true; }   // (a launch that reads its tables may be in flight)
        p->grad_plans.erase(p->grad_plans.begin() + (long)lru);
    }
    std::unique_ptr<GradPlan> G(new (std::nothrow) GradPlan);
    if (!G) return fail(p->ctx, RDIS_HIP_ENOMEM, "eval_grad: host allocation");
    int rc;
    try { rc = build_grad_plan(p, nf, fac, *G); } catch (const std::bad_alloc&) { return fail(p->ctx, RDIS_HIP_ENOMEM, "eval_grad: host allocation"); }
    if (rc) return rc;
    G->key0 = k0; G->key1 = k1; G->used = ++p->grad_tick;
    if (fac) G->ids.assign(fac, fac + nf);
    *out = G.get();
    p->grad_plans.push_back(std::move(G));
    return 0;
}

// value and gradient of a list, left on the device: p->scalar[0] and p->g_all[N] (asynchronous on the context's stream)
int eval_grad_fused(rdis_hip_problem* p, int64_t nf, const int64_t* fac) {
    rdis_hip_ctx* c = p->ctx;
    GradPlan* G = nullptr;
    int rc = grad_plan_for(p, nf, fac, &G);
    if (rc) return rc;
    if ((rc = ensure(c, p->g_all, (size_t)p->N * sizeof(double)))) return rc;
    if ((rc = ensure(c, p->camrec, (size_t)p->ncam_blocks * GRAD_REC * sizeof(double)))) return rc;
    const GradTables& T = G->T;
    const size_t dyn = grad_lds_bytes(T.ncam_cap);
    if (dyn > (size_t)160 * 1024) return fail(c, RDIS_HIP_EINVAL, "eval_grad: a tile's cameras do not fit the LDS");
    HIPCHK(c, grad_camera_records_launch(c->stream, (int)((p->ncam_blocks + 255) / 256), p->x.as<double>(), p->cam_blocks.as<int>(),
                                         (int)p->ncam_blocks, p->camrec.as<double>()));
    HIPCHK(c, grad_fused_launch(c->stream, T.ntiles, dyn, T, p->x.as<double>(), p->obs.as<double2>(), p->camrec.as<double>(),
                                G->cstage.as<double>(), G->pstage.as<double>(), G->partial.as<double>(), p->g_all.as<double>()));
    const long long items = 9ll * T.ncs + 3ll * T.nps + T.nz;
    if (items > 0) HIPCHK(c, grad_combine_launch(c->stream, grid_for(c, items, 256), T, G->cstage.as<double>(), G->pstage.as<double>(), p->g_all.as<double>()));
    final_sum_kernel<<<1, 256, 0, c->stream>>>(T.nchunks, G->partial.as<double>(), p->scalar.as<double>());
    HIPCHK(c, hipGetLastError());
    return 0;
}

// the two-pass form (nonlinear-product functions; bundle adjustment whose blocks overlap): per-factor partials, then one
// lane per variable adds its slots in factor-list order
int eval_grad_two_pass(rdis_hip_problem* p, int64_t nf, const int64_t* fac) {
    rdis_hip_ctx* c = p->ctx;
    int rc;
    const int* dfac;
    if ((rc = stage_ids(p, nf, fac, p->F, &dfac))) return rc;
    // gather lists: cached for the all-factors case
    const int *dptr, *didx;
    DevBuf lptr, lidx;
    if (!fac) {
        if (!p->have_all_v2s) {
            ivec ptr, idx;
            build_v2s(p, nf, nullptr, ptr, idx);
            if ((rc = upload(c, p->all_v2s_ptr, ptr))) return rc;
            if ((rc = upload(c, p->all_v2s_idx, idx))) return rc;
            HIPCHK(c, hipStreamSynchronize(c->stream));
            p->have_all_v2s = true;
        }
        dptr = p->all_v2s_ptr.as<int>(); didx = p->all_v2s_idx.as<int>();
    } else {
        ivec ptr, idx;
        build_v2s(p, nf, fac, ptr, idx);
        if ((rc = upload(c, lptr, ptr))) return rc;
        if ((rc = upload(c, lidx, idx))) return rc;
        HIPCHK(c, hipStreamSynchronize(c->stream));
        dptr = lptr.as<int>(); didx = lidx.as<int>();
    }
    const int blocks = grid_for(c, nf, 256);
    if ((rc = ensure(c, p->partial, (size_t)blocks * sizeof(double)))) return rc;
    if ((rc = ensure(c, p->gfac, (size_t)p->nslots() * sizeof(double)))) return rc;
    if ((rc = ensure(c, p->g_all, (size_t)p->N * sizeof(double)))) return rc;
    if ((rc = launch_eval_sum<true>(p, (int)nf, dfac, blocks))) return rc;
    gather_grad_kernel<<<grid_for(c, p->N, 256), 256, 0, c->stream>>>((int)p->N, dptr, didx, p->gfac.as<double>(), p->g_all.as<double>());
    HIPCHK(c, hipGetLastError());
    if (fac) HIPCHK(c, hipStreamSynchronize(c->stream));   // (lptr / lidx go out of scope)
    return 0;
}

// A list of at most one chunk (GRAD_LANES entries: a component's handful of factors, a single factor) needs no tables: the
// per-factor partials, then one lane per variable adds its slots in list order -- which for one chunk is exactly the order of the
// fused pass (one chunk, one tile: every block's rows in list order), the same bits; the value by the chunk kernel rdis_hip_eval
// uses, the same bits too.  What the fused form would pay for such a list is its table build: six arrays over all N variables,
// half a dozen allocations, three uploads and two or three synchronisations (advisor, round 5).
int eval_grad_short(rdis_hip_problem* p, int64_t nf, const int64_t* fac) {
    rdis_hip_ctx* c = p->ctx;
    int rc;
    const int* dfac;
    if ((rc = stage_ids(p, nf, fac, p->F, &dfac))) return rc;
    ivec ptr, idx;
    build_v2s(p, nf, fac, ptr, idx);
    DevBuf lptr, lidx;
    if ((rc = upload(c, lptr, ptr))) return rc;
    if ((rc = upload(c, lidx, idx))) { (void)hipStreamSynchronize(c->stream); return rc; }
    if ((rc = ensure(c, p->partial, sizeof(double)))) { (void)hipStreamSynchronize(c->stream); return rc; }
    if (!rc) rc = ensure(c, p->gfac, (size_t)p->nslots() * sizeof(double));
    if (!rc) rc = ensure(c, p->g_all, (size_t)p->N * sizeof(double));
    if (rc) { (void)hipStreamSynchronize(c->stream); return rc; }
    ProblemView V = p->view();
    partials_kernel<KIND_BA><<<grid_for(c, nf, 256), 256, 0, c->stream>>>(V, (int)nf, dfac, p->gfac.as<double>());
    HIPCHK(c, hipGetLastError());
    gather_grad_kernel<<<grid_for(c, p->N, 256), 256, 0, c->stream>>>((int)p->N, lptr.as<int>(), lidx.as<int>(), p->gfac.as<double>(), p->g_all.as<double>());
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, eval_chunks_launch(c->stream, 1, V, (int)nf, dfac, p->partial.as<double>()));
    final_sum_kernel<<<1, 256, 0, c->stream>>>(1, p->partial.as<double>(), p->scalar.as<double>());
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));   // (lptr / lidx and the host vectors go out of scope)
    return 0;
}

int eval_grad_on_device(rdis_hip_problem* p, int64_t nf, const int64_t* fac) {
    if (int rc = refuse_exponential(p, nf, fac, "eval_grad")) return rc;
    if (nf == 0) {
        rdis_hip_ctx* c = p->ctx;
        int rc = ensure(c, p->g_all, (size_t)p->N * sizeof(double));
        if (rc) return rc;
        HIPCHK(c, hipMemsetAsync(p->g_all.p, 0, (size_t)p->N * sizeof(double), c->stream));
        HIPCHK(c, hipMemsetAsync(p->scalar.p, 0, sizeof(double), c->stream));
        return 0;
    }
    if (fused_path(p) && fac && nf <= GRAD_LANES) return eval_grad_short(p, nf, fac);
    return fused_path(p) ? eval_grad_fused(p, nf, fac) : eval_grad_two_pass(p, nf, fac);
}

}  // namespace

extern "C" int rdis_hip_eval(rdis_hip_problem* p, int64_t nf, const int64_t* fac, double* f) {
    if (!p || !f) return RDIS_HIP_EINVAL;
    rdis_hip_ctx* c = p->ctx;
    int rc = check_list(p, nf, fac);
    if (rc) return rc;
    USE_DEVICE(c);
    if (nf == 0) { *f = 0.0; return 0; }
    const int* dfac;
    if ((rc = stage_ids(p, nf, fac, p->F, &dfac))) return rc;
    if (fused_path(p)) {
        // chunk by chunk, the chunks' sums added in a fixed order: the value rdis_hip_eval_grad returns, bit for bit
        const int nchunks = (int)((nf + GRAD_LANES - 1) / GRAD_LANES);
        if ((rc = ensure(c, p->partial, (size_t)nchunks * sizeof(double)))) return rc;
        ProblemView V = p->view();
        if (nf >= 4 * p->ncam_blocks) {   // a long list: the cameras' rotation records once per camera instead of once per factor (the same bits)
            camera_rotations_kernel<<<(int)((p->ncam_blocks + 255) / 256), 256, 0, c->stream>>>(p->x.as<double>(), p->cam_blocks.as<int>(), (int)p->ncam_blocks,
                                                                                                p->xrot.as<double>());
            V.xrot = p->xrot.as<double>(); V.rot_mode = ROT_CAMFIX;
        }
        HIPCHK(c, eval_chunks_launch(c->stream, std::min(nchunks, 16 * std::max(1, c->num_cus)), V, (int)nf, dfac, p->partial.as<double>()));
        final_sum_kernel<<<1, 256, 0, c->stream>>>(nchunks, p->partial.as<double>(), p->scalar.as<double>());
        HIPCHK(c, hipGetLastError());
    } else {
        const int blocks = grid_for(c, nf, 256);
        if ((rc = ensure(c, p->partial, (size_t)blocks * sizeof(double)))) return rc;
        if ((rc = launch_eval_sum<false>(p, (int)nf, dfac, blocks))) return rc;
    }
    HIPCHK(c, hipMemcpyAsync(f, p->scalar.p, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int rdis_hip_eval_grad(rdis_hip_problem* p, int64_t nf, const int64_t* fac, double* f, double* g) {
    if (!p || !f || !g) return RDIS_HIP_EINVAL;
    rdis_hip_ctx* c = p->ctx;
    int rc = check_list(p, nf, fac);
    if (rc) return rc;
    USE_DEVICE(c);
    if (nf == 0) { *f = 0.0; std::fill(g, g + p->N, 0.0); return 0; }
    if ((rc = eval_grad_on_device(p, nf, fac))) return rc;
    HIPCHK(c, hipMemcpyAsync(f, p->scalar.p, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(g, p->g_all.p, (size_t)p->N * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int rdis_hip_eval_grad_device(rdis_hip_problem* p, int64_t nf, const int64_t* fac, void** f_dev, void** g_dev) {
    if (!p || !f_dev || !g_dev) return RDIS_HIP_EINVAL;
    rdis_hip_ctx* c = p->ctx;
    int rc = check_list(p, nf, fac);
    if (rc) return rc;
    USE_DEVICE(c);
    if ((rc = eval_grad_on_device(p, nf, fac))) return rc;
    *f_dev = p->scalar.p;
    *g_dev = p->g_all.p;
    return 0;
}

extern "C" int rdis_hip_set_factor_rounding(rdis_hip_problem* p, int32_t mode) {
    if (!p) return RDIS_HIP_EINVAL;
    if (mode != 0 && mode != 1) return fail(p->ctx, RDIS_HIP_EINVAL, "set_factor_rounding: 0 (fused multiply-adds) or 1 (the reference's rounding)");
    if (mode == 1 && p->kind != KIND_BA) return fail(p->ctx, RDIS_HIP_EINVAL, "set_factor_rounding: bundle adjustment only");
    p->each_rounding = mode;
    return 0;
}

extern "C" int rdis_hip_eval_each(rdis_hip_problem* p, int64_t nf, const int64_t* fac, double* fvals) {
    if (!p || (nf && !fvals)) return RDIS_HIP_EINVAL;
    rdis_hip_ctx* c = p->ctx;
    int rc = check_list(p, nf, fac);
    if (rc) return rc;
    USE_DEVICE(c);
    if (nf == 0) return 0;
    const int* dfac;
    if ((rc = stage_ids(p, nf, fac, p->F, &dfac))) return rc;
    if ((rc = ensure(c, p->tmp_out, (size_t)nf * sizeof(double)))) return rc;
    ProblemView V = p->view();
    const int blocks = grid_for(c, nf, 256);
    if (p->kind == KIND_BA && p->each_rounding == 1) HIPCHK(c, refround_eval_each(blocks, c->stream, &V, (int)nf, dfac, p->tmp_out.as<double>()));
    else if (p->kind == KIND_BA) eval_each_kernel<KIND_BA><<<blocks, 256, 0, c->stream>>>(V, (int)nf, dfac, p->tmp_out.as<double>());
    else eval_each_kernel<KIND_NLP><<<blocks, 256, 0, c->stream>>>(V, (int)nf, dfac, p->tmp_out.as<double>());
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(fvals, p->tmp_out.p, (size_t)nf * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int rdis_hip_grad_each_ba(rdis_hip_problem* p, int64_t nf, const int64_t* fac, double* g12) {
    if (!p || (nf && !g12)) return RDIS_HIP_EINVAL;
    rdis_hip_ctx* c = p->ctx;
    if (p->kind != KIND_BA) return fail(c, RDIS_HIP_EINVAL, "grad_each_ba: not a bundle-adjustment problem");
    int rc = check_list(p, nf, fac);
    if (rc) return rc;
    USE_DEVICE(c);
    if (nf == 0) return 0;
    const int* dfac;
    if ((rc = stage_ids(p, nf, fac, p->F, &dfac))) return rc;
    if ((rc = ensure(c, p->gfac, (size_t)p->nslots() * sizeof(double)))) return rc;
    if ((rc = ensure(c, p->tmp_out, (size_t)nf * 12 * sizeof(double)))) return rc;
    ProblemView V = p->view();
    if (p->each_rounding == 1) {
        HIPCHK(c, refround_grad_each(grid_for(c, nf, 256), c->stream, &V, (int)nf, dfac, p->tmp_out.as<double>()));
    } else {
        partials_kernel<KIND_BA><<<grid_for(c, nf, 256), 256, 0, c->stream>>>(V, (int)nf, dfac, p->gfac.as<double>());
        HIPCHK(c, hipGetLastError());
        gather_rows12_kernel<<<grid_for(c, nf * 12, 256), 256, 0, c->stream>>>((int)nf, dfac, p->gfac.as<double>(), p->tmp_out.as<double>());
        HIPCHK(c, hipGetLastError());
    }
    HIPCHK(c, hipMemcpyAsync(g12, p->tmp_out.p, (size_t)nf * 12 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

// =====================================================================================
// plans
// =====================================================================================

// =====================================================================================
// plans
// =====================================================================================
namespace {

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
constexpr size_t DIRECT_X_BYTES = 4u << 20;    // a result vector this long is copied straight into the caller's array (rdis_hip_plan_fetch)
constexpr size_t STAGE_MAX_BYTES = 1u << 20;   // start points staged through pinned memory up to this size (resident plans)

// plan memory: a persistent plan owns hipMalloc'ed buffers; a transient one (cgd_batch, i.e.
// one optimize() call) carves them out of the problem's arena so that a call costs no
// hipMalloc / hipFree (each of which synchronises the device)
int plan_alloc(rdis_hip_plan* L, DevBuf& b, size_t bytes) {
    rdis_hip_problem* p = L->prob;
    if (!L->transient) {
        L->dev_bytes -= std::min(L->dev_bytes, b.owned ? b.bytes : 0);
        const int rc = dalloc(p->ctx, b, bytes);
        if (!rc) L->dev_bytes += b.bytes;
        return rc;
    }
    b.release();
    bytes = align_up(std::max<size_t>(bytes, 8), 256);
    // (the arena is sized from bounds, plan_create_impl; what exceeds them -- rare shapes -- gets memory of its own)
    if (p->arena_used + bytes > p->arena.bytes) return dalloc(p->ctx, b, bytes);
    b.p = static_cast<char*>(p->arena.p) + p->arena_used;
    b.bytes = bytes;
    b.owned = false;
    p->arena_used += bytes;
    return 0;
}

int ensure_problem_scratch(rdis_hip_problem* p) {
    rdis_hip_ctx* c = p->ctx;
    if (!p->dir.p) {
        int rc = dalloc(c, p->dir, (size_t)p->N * sizeof(double));
        if (!rc) rc = dalloc(c, p->coop_state, coop_state_bytes());
        if (!rc) rc = dalloc(c, p->coop_timing, COOP_TM * sizeof(long long));
        if (rc) return rc;
        HIPCHK(c, hipMemsetAsync(p->dir.p, 0, p->dir.bytes, c->stream));
        HIPCHK(c, hipEventCreate(&p->ev0));
        HIPCHK(c, hipEventCreate(&p->ev1));
        p->h_mark.assign((size_t)p->N, VarMark{0, -1, -1, 0});
        p->h_local.assign((size_t)p->N, -1);
        p->h_owner_stamp.assign((size_t)p->N, 0);
        p->h_fac_stamp.assign((size_t)p->F, 0);
    }
    return 0;
}

}  // namespace

static int plan_create_impl(rdis_hip_problem* p, bool transient, int64_t ncomp, const int64_t* free_ptr,
                            const int64_t* free_vid, const int64_t* fac_ptr, const int64_t* fac_id,
                            rdis_hip_plan** out) {
    if (!p || !out || ncomp < 0 || !free_ptr || !fac_ptr) return RDIS_HIP_EINVAL;
    rdis_hip_ctx* c = p->ctx;
    *out = nullptr;
    const int64_t nfree = free_ptr[ncomp], nfac = fac_ptr[ncomp];
    if (free_ptr[0] != 0 || fac_ptr[0] != 0 || nfree < 0 || nfac < 0 || (nfree && !free_vid) || (nfac && !fac_id))
        return fail(c, RDIS_HIP_EINVAL, "plan_create: bad CSR");
    if (nfree >= (1ll << 31) / 5 || nfac >= ((1ll << 31) - 16) / 12) return fail(c, RDIS_HIP_ERANGE, "plan_create: too large");
    if (int rc = refuse_exponential(p, nfac, fac_id, "plan_create")) return rc;
    USE_DEVICE(c);
    int rc = ensure_problem_scratch(p);
    if (rc) return rc;

    rdis_hip_plan* L = new (std::nothrow) rdis_hip_plan;
    if (!L) return fail(c, RDIS_HIP_ENOMEM, "plan_create: host allocation");
    struct Guard { rdis_hip_plan* l; ~Guard() { if (l) rdis_hip_plan_destroy(l); } } guard{L};
    L->prob = p; L->transient = transient; L->ncomp = ncomp; L->nfree = nfree; L->nfac = nfac;
    // (A/B runs of callers that cannot reach the plan's options, e.g. rdis_hip_cgd_batch: RDIS_HIP_COOP_PIPELINE=0 / 1)
    if (const char* ev = std::getenv("RDIS_HIP_COOP_PIPELINE")) L->coop_pipeline = std::atoi(ev) != 0;

    // --- validate independence (free sets disjoint, factors owned once, no factor of one
    // component reading a free variable of another); the marks are a persistent per-problem
    // array validated by a stamp, so a call costs O(its own size), not O(N)
    const int stamp = ++p->stamp;
    VarMark* const mark = p->h_mark.data();
    L->h_free_ptr.resize((size_t)ncomp + 1); L->h_fac_ptr.resize((size_t)ncomp + 1);
    L->h_free_vid.resize((size_t)nfree); L->h_fac_id.resize((size_t)nfac);
    for (int64_t cc = 0; cc <= ncomp; ++cc) {
        if (cc && (free_ptr[cc] < free_ptr[cc - 1] || fac_ptr[cc] < fac_ptr[cc - 1]))
            return fail(c, RDIS_HIP_EINVAL, "plan_create: ptr not monotone");
        L->h_free_ptr[(size_t)cc] = (int)free_ptr[cc];
        L->h_fac_ptr[(size_t)cc] = (int)fac_ptr[cc];
    }
    for (int64_t cc = 0; cc < ncomp; ++cc)
        for (int64_t i = free_ptr[cc]; i < free_ptr[cc + 1]; ++i) {
            const int64_t v = free_vid[i];
            if (v < 0 || v >= p->N) return fail(c, RDIS_HIP_EINVAL, "plan_create: free variable id out of range");
            VarMark& mk = mark[(size_t)v];
            if (mk.stamp == stamp) return fail(c, RDIS_HIP_EOVERLAP, "plan_create: variable " + std::to_string(v) + " is free in two components (or listed twice)");
            mk.stamp = stamp; mk.owner = (int)cc; mk.gidx = (int)i;
            L->h_free_vid[(size_t)i] = (int)v;
        }
    // One pass over the listed factors' slots leaves, per slot, the position of its variable in the
    // plan's free list (-1: a constant for this solve) and counts the slots per variable; a second,
    // sequential one turns that into the variable-major position of the slot's partial (slot_pos) and
    // the variable's index within its component (h_slot_li: what the cooperative layouts address by).
    ivec& v2s_ptr = L->h_v2s_ptr;
    v2s_ptr.assign((size_t)nfree + 1, 0);
    ivec slot_base((size_t)nfac + 1);
    slot_base[0] = 0;
    int* const fid32 = L->h_fac_id.data();
    int* const cnt = v2s_ptr.data() + 1;
    const bool ba = p->kind == KIND_BA;
    if (ba) L->nslots = 12 * nfac;
    else {
        int64_t ns = 0;
        for (int64_t j = 0; j < nfac; ++j) {
            const int64_t f = fac_id[j];
            if (f < 0 || f >= p->F) return fail(c, RDIS_HIP_EINVAL, "plan_create: factor id out of range");
            ns += p->arity((int)f);
        }
        if (ns >= (1ll << 31) - 16) return fail(c, RDIS_HIP_ERANGE, "plan_create: too large");
        L->nslots = ns;
    }
    ivec& sli = L->h_slot_li;
    sli.resize((size_t)L->nslots);
    int* const gi = sli.data();
    auto other_owner = [&](int64_t f, int64_t cc, int o) {
        return fail(c, RDIS_HIP_EOVERLAP, "plan_create: factor " + std::to_string(f) + " of component " + std::to_string(cc) + " reads a free variable of component " + std::to_string(o));
    };
    for (int64_t cc = 0; cc < ncomp; ++cc) {
        const int icc = (int)cc;
        for (int64_t j = fac_ptr[cc]; j < fac_ptr[cc + 1]; ++j) {
            const int64_t f = fac_id[j];
            if (f < 0 || f >= p->F) return fail(c, RDIS_HIP_EINVAL, "plan_create: factor id out of range");
            if (p->h_fac_stamp[(size_t)f] == stamp) return fail(c, RDIS_HIP_EOVERLAP, "plan_create: factor " + std::to_string(f) + " listed twice");
            p->h_fac_stamp[(size_t)f] = stamp;
            fid32[j] = (int)f;
            if (ba) {
                slot_base[(size_t)j + 1] = 12 * (int)(j + 1);
                int* g = gi + 12 * j;
                const VarMark* mc = mark + p->h_cam[(size_t)f];
                const VarMark* mp = mark + p->h_pt[(size_t)f];
#pragma unroll
                for (int k = 0; k < 12; ++k) {
                    const VarMark& mk = k < 9 ? mc[k] : mp[k - 9];
                    int pos = -1;
                    if (mk.stamp == stamp) {
                        if (mk.owner != icc) return other_owner(f, cc, mk.owner);
                        pos = mk.gidx;
                        ++cnt[pos];
                    }
                    g[k] = pos;
                }
            } else {
                const int a = p->arity((int)f), sb = slot_base[(size_t)j];
                slot_base[(size_t)j + 1] = sb + a;
                for (int k = 0; k < a; ++k) {
                    const VarMark& mk = mark[(size_t)p->var_of((int)f, k)];
                    int pos = -1;
                    if (mk.stamp == stamp) {
                        if (mk.owner != icc) return other_owner(f, cc, mk.owner);
                        pos = mk.gidx;
                        ++cnt[pos];
                    }
                    gi[sb + k] = pos;
                }
            }
        }
    }
    for (int64_t i = 0; i < nfree; ++i) v2s_ptr[(size_t)i + 1] += v2s_ptr[(size_t)i];
    L->ngfac = v2s_ptr[(size_t)nfree];
    // heaviest components first (longest-processing-time order for the launch)
    L->h_order.resize((size_t)ncomp);
    std::iota(L->h_order.begin(), L->h_order.end(), 0);
    std::stable_sort(L->h_order.begin(), L->h_order.end(), [&](int a, int b) {
        return (fac_ptr[a + 1] - fac_ptr[a]) > (fac_ptr[b + 1] - fac_ptr[b]);
    });

    // --- one int32 block, one H2D copy
    ivec& blk = L->h_blk;
    blk.reserve((size_t)(4 * ncomp + 3 * nfree + 2 * nfac + L->nslots) + 64);
    auto put = [&](const ivec& v) { const size_t off = blk.size(); blk.insert(blk.end(), v.begin(), v.end()); return off; };
    L->off_order = put(L->h_order);
    L->off_free_ptr = put(L->h_free_ptr); L->off_free_vid = put(L->h_free_vid);
    L->off_fac_ptr = put(L->h_fac_ptr); L->off_fac_id = put(L->h_fac_id);
    L->off_v2s_ptr = put(v2s_ptr); L->off_slot_base = put(slot_base);
    {   // gfac is variable-major: slot_pos[s] = where listed factor slot s lands in it (-1: not free here),
        // the slots of a variable in listed order; the positions above become component-local indices
        L->off_slot_pos = blk.size();
        blk.resize(blk.size() + (size_t)L->nslots);
        int* const slot_pos = blk.data() + L->off_slot_pos;
        ivec fill(v2s_ptr.begin(), v2s_ptr.end() - 1);
        int* const fl = fill.data();
        for (int64_t cc = 0; cc < ncomp; ++cc) {
            const int f0 = (int)free_ptr[cc];
            const int64_t s0 = slot_base[(size_t)fac_ptr[cc]], s1 = slot_base[(size_t)fac_ptr[cc + 1]];
            for (int64_t s = s0; s < s1; ++s) {
                const int g = gi[s];
                slot_pos[s] = g >= 0 ? fl[g]++ : -1;
                gi[s] = g >= 0 ? g - f0 : -1;
            }
        }
    }
    // (the per-slot local indices serve prepare_partition's cooperative groups, bundle adjustment only; a plan
    // too large for them to matter does not keep them: prepare_partition then forms a group's indices itself)
    if (!ba || L->nslots > (64ll << 20)) { ivec().swap(L->h_slot_li); }
    {   // per component: the camera blocks with a free rotation variable (their records follow the trial
        // point) and where in the component's vectors their three rotation variables are (-1: constant)
        ivec cb_ptr((size_t)ncomp + 1, 0), cb, cb_li;
        if (p->kind == KIND_BA && p->ncam_blocks > 0) {
            std::vector<std::array<int, 3>> ent;   // (block, which of the three, local index)
            for (int64_t cc = 0; cc < ncomp; ++cc) {
                ent.clear();
                for (int64_t i = free_ptr[cc]; i < free_ptr[cc + 1]; ++i) {
                    const int v = L->h_free_vid[(size_t)i], b = p->h_block_of[(size_t)v];
                    if (b >= 0 && v - b < 3) ent.push_back({b, v - b, (int)(i - free_ptr[cc])});
                }
                std::sort(ent.begin(), ent.end());
                for (const auto& e : ent) {
                    if (cb.size() == (size_t)cb_ptr[(size_t)cc] || cb.back() != e[0]) {
                        cb.push_back(e[0]);
                        cb_li.insert(cb_li.end(), {-1, -1, -1});
                    }
                    cb_li[cb_li.size() - 3 + (size_t)e[1]] = e[2];
                }
                cb_ptr[(size_t)cc + 1] = (int)cb.size();
            }
        }
        for (int64_t cc = 0; cc < ncomp; ++cc) cb_ptr[(size_t)cc + 1] = std::max(cb_ptr[(size_t)cc + 1], cb_ptr[(size_t)cc]);
        L->off_cb_ptr = put(cb_ptr); L->off_cb = put(cb); L->off_cb_li = put(cb_li);
    }

    const size_t nc = (size_t)ncomp;
    L->out_bytes = ((size_t)nfree + 2 * nc) * 8 + 2 * nc * 8 + 3 * nc * 4;
    if (transient) {
        // everything this call can need, including what prepare_partition adds for the cooperative solver
        const size_t lanes_max = (size_t)COOP_MAX_WG * 512;
        size_t need = 4096 + 16 * 256;
        need += align_up(blk.size() * 4, 256) + align_up((size_t)(5 * nfree) * 8, 256) + align_up((size_t)L->ngfac * 8, 256);
        need += align_up((size_t)nfree * 8, 256) + align_up(L->out_bytes, 256) + 256;
        need += align_up(nc * 4, 256) + align_up((size_t)nfree * 8, 256);                       // rest_order, xi_glob
        need += align_up((size_t)nfree * 4, 256) + 8 * 256;                                      // long_vars of streaming components
        // lane_var / wave_var of the cooperative groups: at most 3 x the lanes a component needs, rounded up to workgroups
        need += 5 * (3 * (size_t)(nfac + nfree) + 2048 * (size_t)std::min<int64_t>(ncomp, 4096)) + align_up(lanes_max * 4, 256);
        need += align_up((size_t)(12 * nfac) * 4, 256) + 8 * 256;                               // slot_li of all cooperative components
        need += align_up((size_t)(2 * (12 * nfac + 9 * nfree) + 2 * nfac + 64 * (nfac / 64 + std::min<int64_t>(nfac, nfree / 9 + 1) + ncomp) + 3 * ncomp + 3) * 4, 256) + align_up((size_t)nfac * 16, 256) + 256;   // slot tables, observations of the LDS-resident solver
        need += align_up((size_t)(2 * nfac + nfree / 3 + 3 * ncomp + 16) * 4, 256) + align_up((size_t)(nfac + 1) * 20, 256) + align_up((size_t)(24 * nfac + 6 * nfree + 16 * ncomp + 2048) * 8, 256);   // ... of the streaming solver (records, point-major arrays, camera partials)
        need += (size_t)COOP_MAX_GROUPS * (4 * 256 + sizeof(CoopGroup)) + (size_t)COOP_MAX_WG * 4 + lanes_max * 4 + 64 * 256;  // groups, their alignment slack
        if (p->arena.bytes < need) {
            HIPCHK(c, hipStreamSynchronize(c->stream));
            rc = dalloc(c, p->arena, std::max(need, 2 * p->arena.bytes));
            if (rc) return rc;
        }
        p->arena_used = 0;
    }
    rc = plan_alloc(L, L->ints, blk.size() * sizeof(int));
    if (!rc) rc = plan_alloc(L, L->ws, (size_t)(5 * nfree) * sizeof(double));
    if (!rc) rc = plan_alloc(L, L->gfac, (size_t)L->ngfac * sizeof(double));
    if (!rc) rc = plan_alloc(L, L->xstart, (size_t)nfree * sizeof(double));
    if (!rc) rc = plan_alloc(L, L->outbuf, L->out_bytes);
    if (!rc) rc = plan_alloc(L, L->objective, 64);
    if (rc) return rc;
    L->h_out.resize(L->out_bytes);
    if (!blk.empty()) HIPCHK(c, hipMemcpyAsync(L->ints.p, blk.data(), blk.size() * sizeof(int), hipMemcpyHostToDevice, c->stream));
    guard.l = nullptr;
    *out = L;
    return 0;
}

extern "C" int rdis_hip_plan_create(rdis_hip_problem* p, int64_t ncomp, const int64_t* free_ptr,
                                    const int64_t* free_vid, const int64_t* fac_ptr, const int64_t* fac_id,
                                    rdis_hip_plan** out) {
    return plan_create_impl(p, false, ncomp, free_ptr, free_vid, fac_ptr, fac_id, out);
}

extern "C" void rdis_hip_plan_destroy(rdis_hip_plan* L) {
    if (!L) return;
    if (L->prob) (void)make_current(L->prob->ctx);
    if (L->prob && !L->transient) (void)hipStreamSynchronize(L->prob->ctx->stream);
    delete L;
}

extern "C" int rdis_hip_plan_set_start(rdis_hip_plan* L, const double* xs) {
    if (!L) return RDIS_HIP_EINVAL;
    rdis_hip_ctx* c = L->prob->ctx;
    USE_DEVICE(c);
    if (L->nfree == 0) { L->have_start = true; return 0; }
    if (xs) {
        const size_t bytes = (size_t)L->nfree * sizeof(double);
        rdis_hip_problem* p = L->prob;
        if (!L->transient && bytes <= STAGE_MAX_BYTES) {
            // a resident plan's caller may reuse xs the moment this returns: the values go through a pinned staging
            // buffer of the problem (one host copy, an asynchronous upload) instead of a wait for the stream
            if (p->stage_busy) { HIPCHK(c, hipEventSynchronize(p->stage_ev)); p->stage_busy = false; }
            if (p->stage_x.bytes < bytes) { int rc = halloc(c, p->stage_x, std::max<size_t>(2 * bytes, 4096)); if (rc) return rc; }
            if (!p->stage_ev) HIPCHK(c, hipEventCreateWithFlags(&p->stage_ev, hipEventDisableTiming));
            std::memcpy(p->stage_x.p, xs, bytes);
            HIPCHK(c, hipMemcpyAsync(L->xstart.p, p->stage_x.p, bytes, hipMemcpyHostToDevice, c->stream));
            HIPCHK(c, hipEventRecord(p->stage_ev, c->stream));
            p->stage_busy = true;
        } else {
            // pageable source: a transient plan's caller keeps xs alive until the results are fetched
            HIPCHK(c, hipMemcpyAsync(L->xstart.p, xs, bytes, hipMemcpyHostToDevice, c->stream));
            if (!L->transient) HIPCHK(c, hipStreamSynchronize(c->stream));
        }
    } else {
        gather_x_kernel<<<grid_for(c, L->nfree, 256), 256, 0, c->stream>>>((int)L->nfree, L->ip(L->off_free_vid), L->prob->x.as<double>(), L->xstart.as<double>());
        HIPCHK(c, hipGetLastError());
    }
    L->have_start = true;
    return 0;
}

extern "C" int rdis_hip_plan_set_option(rdis_hip_plan* L, const char* name, int64_t value) {
    if (!L || !name) return RDIS_HIP_EINVAL;
    rdis_hip_ctx* c = L->prob->ctx;
    USE_DEVICE(c);
    const std::string n(name);
    if (n == "block_threads") {
        if (value != 0 && value != 64 && value != 128 && value != 256 && value != 512 && value != 768 && value != 1024)
            return fail(c, RDIS_HIP_EINVAL, "block_threads must be 0, 64, 128, 256, 512, 768 or 1024");
        L->block_threads = (int)value;
    } else if (n == "coop_min_factors") {
        if (value < 0) return fail(c, RDIS_HIP_EINVAL, "coop_min_factors < 0");
        L->coop_min_factors = value;
    } else if (n == "quad_min_components") {
        if (value < 0) return fail(c, RDIS_HIP_EINVAL, "quad_min_components < 0");
        L->quad_min_components = value;
    } else if (n == "coop_group_min_factors") {
        if (value < 0) return fail(c, RDIS_HIP_EINVAL, "coop_group_min_factors < 0");
        L->coop_group_min_factors = value;
    } else if (n == "tiny_max_blocks") {
        if (value < 0 || value > (1 << 20)) return fail(c, RDIS_HIP_EINVAL, "tiny_max_blocks out of range");
        L->tiny_max_blocks = (int)value;
    } else if (n == "overlap_batch") {
        L->overlap_batch = value != 0;
    } else if (n == "row_min_components") {
        if (value < 0) return fail(c, RDIS_HIP_EINVAL, "row_min_components < 0");
        L->row_min_components = value;
    } else if (n == "camera_records") {
        if (value < 0 || value > 2) return fail(c, RDIS_HIP_EINVAL, "camera_records must be 0, 1 or 2");
        L->camera_records = (int)value;
    } else if (n == "quad_max_vars") {
        if (value < 0 || value > QUAD_MAX_VARS) return fail(c, RDIS_HIP_EINVAL, "quad_max_vars out of range");
        L->quad_max_vars = (int)value;
    } else if (n == "coop_max_components") {
        if (value < 0 || value > 4096) return fail(c, RDIS_HIP_EINVAL, "coop_max_components out of range");
        L->coop_max_components = (int)value;
    } else if (n == "coop_workgroups") {
        if (value < 0 || value > COOP_MAX_WG) return fail(c, RDIS_HIP_EINVAL, "coop_workgroups out of range");
        L->coop_workgroups = (int)value;
    } else if (n == "coop_threads") {
        if (value != 128 && value != 256 && value != 512) return fail(c, RDIS_HIP_EINVAL, "coop_threads must be 128, 256 or 512");
        L->coop_threads = (int)value;
    } else if (n == "coop_speculate") {
        L->coop_speculate = value != 0;
    } else if (n == "coop_pipeline") {
        L->coop_pipeline = value != 0;
    } else if (n == "lds_resident") {
        L->lds_resident = value != 0;
    } else if (n == "ptm_stream") {
        if (value < 0 || value > 2) return fail(c, RDIS_HIP_EINVAL, "ptm_stream must be 0 (never), 1 (components too large for the LDS) or 2 (every component its tables fit)");
        L->ptm_stream = (int)value;
    } else if (n == "ptm_group") {
        if (value < 0 || value > PTM_WIDE_MAX_GROUP) return fail(c, RDIS_HIP_EINVAL, "ptm_group must be 0 (auto), 1 (never) or the number of workgroups per component (at most 16; up to 512 for the wide groups of a few large components)");
        L->ptm_group = (int)value;
        return 0;   // (no table depends on it)
    } else if (n == "ptm_round_slots") {
        if (value < 0 || value > 2) return fail(c, RDIS_HIP_EINVAL, "ptm_round_slots must be 0 (two where the LDS holds them), 1 or 2");
        L->ptm_round_slots = (int)value;
        L->rounds_threads = L->rounds_K = 0;   // (the round tables are built for one choice)
        return 0;
    } else if (n == "ptm_threads") {
        if (value != 0 && value != 256 && value != 512 && value != 768) return fail(c, RDIS_HIP_EINVAL, "ptm_threads must be 0, 256, 512 or 768");
        L->ptm_threads = (int)value;
    } else if (n == "ptm_local_cameras") {
        if (value < -1 || value > 1) return fail(c, RDIS_HIP_EINVAL, "ptm_local_cameras must be -1 (where a wide group's cameras do not fit the LDS), 0 (never) or 1 (every wide group)");
        L->ptm_local_cameras = (int)value;
        L->ptm_local_off = false;
    } else if (n == "emulate_stale_cache") {
        L->emulate_stale = value != 0;
    } else if (n == "factor_rounding") {
        if (value < -1 || value > 1) return fail(c, RDIS_HIP_EINVAL, "factor_rounding must be -1 (auto), 0 (fused multiply-adds) or 1 (the reference's rounding)");
        L->factor_rounding = (int)value;
    } else if (n == "lds_camera_sums") {
        L->lds_camera_sums = value != 0;
    } else if (n == "lds_matrix") {
        L->lds_matrix = value != 0;
        return 0;   // (no table depends on it)
    } else if (n == "lds_rot") {
        if (value < -1 || value > 1) return fail(c, RDIS_HIP_EINVAL, "lds_rot must be -1, 0 or 1");
        L->lds_rot = (int)value;
    } else if (n == "lds_threads") {
        if (value != 0 && value != 64 && value != 128 && value != 256 && value != 512 && value != 768 && value != 1024)
            return fail(c, RDIS_HIP_EINVAL, "lds_threads must be 0, 64, 128, 256, 512, 768 or 1024");
        L->lds_threads = (int)value;
    } else if (n == "force_stream") {
        L->force_stream = value != 0;
    } else if (n == "coop_poll_delay") {
        if (value < 0 || value > 1024) return fail(c, RDIS_HIP_EINVAL, "coop_poll_delay out of range");
        L->coop_poll_delay = (int)value;
    } else if (n == "trace_records") {
        if (value < 0 || value > (1 << 22)) return fail(c, RDIS_HIP_EINVAL, "trace_records out of range");
        if (L->transient && value) return fail(c, RDIS_HIP_EINVAL, "tracing needs a persistent plan");
        L->trace_records = (int)value;
        if (value > 0) {
            int rc = dalloc(c, L->trace, (size_t)L->ncomp * (size_t)value * 4 * sizeof(double));
            if (rc) return rc;
        }
    } else if (n == "dump_iters") {
        if (value < 0 || value > 4096 || (double)value * 2.0 * (double)L->nfree > 4e9) return fail(c, RDIS_HIP_EINVAL, "dump_iters out of range");
        if (L->transient && value) return fail(c, RDIS_HIP_EINVAL, "vector dumps need a persistent plan");
        L->dump_iters = (int)value;
        if (value > 0) {
            int rc = dalloc(c, L->vdump, (size_t)value * 2 * (size_t)L->nfree * sizeof(double));
            if (rc) return rc;
            HIPCHK(c, hipMemsetAsync(L->vdump.p, 0, L->vdump.bytes, c->stream));
        }
    } else {
        return fail(c, RDIS_HIP_EINVAL, "unknown option " + n);
    }
    L->partition_dirty = true;
    return 0;
}

namespace {
// decide which components the cooperative solver takes and build what it needs
int prepare_partition(rdis_hip_plan* L) {
    rdis_hip_ctx* c = L->prob->ctx;
    rdis_hip_problem* p = L->prob;
    // (the device memory of the items about to be dropped leaves the plan's account with them)
    auto drop = [&](DevBuf& b) { if (!L->transient && b.owned) L->dev_bytes -= std::min(L->dev_bytes, b.bytes); b.release(); };
    for (CoopLaunch& cl : L->coop_launches) { drop(cl.groups); drop(cl.wg_group); }
    for (StreamItem& st : L->stream) drop(st.long_vars);
    L->coop.clear();
    L->h_coop_ints.clear();
    L->coop_launches.clear();
    L->h_coop_groups.clear();
    L->h_coop_wg.clear();
    L->stream.clear();
    L->h_rest.clear();
    int cap = 0, scap = 0;
    // The grid solvers are for a few large components that would leave the device idle as single
    // workgroups, one launch each.  When there are more large components than that, the batch
    // kernel fills the device by itself (one workgroup per component) and is the better fit.
    std::vector<int64_t> nlong_of((size_t)L->ncomp, -1);   // (counted once per component: the layouts are compared below)
    auto groups_of = [&](int cc) {   // workgroups of a component's cooperative group: a lane per factor / variable, a wave per long gradient run
        const int wpw = L->coop_lanes() / 64;
        const int64_t m = L->h_fac_ptr[(size_t)cc + 1] - L->h_fac_ptr[(size_t)cc];
        const int64_t n = L->h_free_ptr[(size_t)cc + 1] - L->h_free_ptr[(size_t)cc];
        const int f0 = L->h_free_ptr[(size_t)cc];
        int64_t& nlong = nlong_of[(size_t)cc];
        if (nlong < 0) {
            nlong = 0;
            for (int64_t i = 0; i < n; ++i)
                if (L->h_v2s_ptr[(size_t)(f0 + i) + 1] - L->h_v2s_ptr[(size_t)(f0 + i)] > COOP_LONG_LIST) ++nlong;
        }
        const int64_t need = (std::max(m, n) + L->coop_lanes() - 1) / L->coop_lanes();
        // (small groups only: a large one has its waves anyway, and more workgroups lengthen every sweep)
        const int64_t for_long = need < 16 ? std::min<int64_t>((nlong + wpw - 1) / wpw, 2 * need + 2) : 0;
        return std::max<int64_t>(1, std::max(need, for_long));
    };
    int64_t nbig = 0;
    for (int64_t cc = 0; cc < L->ncomp; ++cc)
        if (L->h_fac_ptr[(size_t)cc + 1] - L->h_fac_ptr[(size_t)cc] >= L->coop_min_factors) ++nbig;
    bool any_big = L->coop_min_factors > 0 && L->coop_max_components > 0 && L->nfac >= L->coop_min_factors &&
                   nbig <= L->coop_max_components;
    const bool coop_on = L->coop_min_factors > 0 && L->coop_max_components > 0 && p->kind == KIND_BA && !L->force_stream;
    auto cap_of = [&]() {
        // (the occupancy queries behind these are not free: asked once per context -- a device -- and layout;
        // calls on a context are serialised, include/rdis_hip.h)
        const int ti = L->coop_threads == 128 ? 0 : L->coop_threads == 256 ? 1 : 2;
        const bool rr = L->coop_reference_rounding();
        int& kc = rr ? c->cap_coop_rr[ti] : c->cap_coop[ti];
        int& kp = rr ? c->cap_pipe_rr : c->cap_pipe;
        if (L->pipelined() && kp < 0) kp = rr ? refround_pipe_max_workgroups(c->num_cus) : pipe_max_workgroups(c->num_cus);
        if (!L->pipelined() && kc < 0) kc = rr ? refround_coop_max_workgroups(L->coop_threads, c->num_cus) : coop_max_workgroups(L->coop_threads, c->num_cus);
        int k = L->pipelined() ? kp : kc;
        if (L->coop_workgroups > 0) k = std::min(k, L->coop_workgroups);
        return k;
    };
    // The pipelined layout (solver_pipe.hpp) has half the factor lanes per workgroup: it is used when
    // everything that gets a cooperative group with the plain layout also gets one with it.
    L->use_pipe = false;
    // (factor_rounding = 1, the parity option: the reference's slope -- a gradient pass of the whole group per trial and one
    // sequential sum -- exists in the plain cooperative layout only, solver_coop.hpp)
    if (coop_on && L->coop_pipeline != 0 && L->factor_rounding != 1 && L->coop_threads == PIPE_THREADS) {
        auto census = [&](int64_t& group_total, int64_t& group_count, int64_t& big_unfit) {
            const int k = cap_of();
            group_total = group_count = big_unfit = 0;
            for (int cc : L->h_order) {
                const int64_t m = L->h_fac_ptr[(size_t)cc + 1] - L->h_fac_ptr[(size_t)cc];
                if (L->coop_group_min_factors > 0 && m >= L->coop_group_min_factors) { group_total += groups_of(cc); ++group_count; }
                if (m >= L->coop_min_factors && groups_of(cc) > k) ++big_unfit;
            }
            return k;
        };
        int64_t gt0, gc0, bu0, gt1, gc1, bu1;
        const int k0 = census(gt0, gc0, bu0);
        L->use_pipe = true;
        const int k1 = census(gt1, gc1, bu1);
        const bool group0 = gc0 > 0 && gt0 <= k0 && gc0 <= COOP_MAX_GROUPS, group1 = gc1 > 0 && gt1 <= k1 && gc1 <= COOP_MAX_GROUPS;
        if ((group0 && !group1) || bu1 > bu0) L->use_pipe = false;
    }
    if (coop_on) cap = cap_of();
    // Group mode: every component of some size gets a cooperative group when all the groups are
    // resident at once -- a device that the batch kernel would leave mostly idle (49 camera components
    // of ladybug: 6.6 ms as one workgroup each).  Otherwise only the few very large ones do.
    int64_t group_min = INT64_MAX;
    if (coop_on && cap > 0 && L->coop_group_min_factors > 0) {
        int64_t total = 0, count = 0;
        for (int cc : L->h_order) {
            if (L->h_fac_ptr[(size_t)cc + 1] - L->h_fac_ptr[(size_t)cc] < L->coop_group_min_factors) break;   // heaviest first
            total += groups_of(cc);
            ++count;
        }
        if (count > 0 && total <= cap && count <= COOP_MAX_GROUPS) { group_min = L->coop_group_min_factors; any_big = true; }
    }
    if (any_big) scap = stream_max_workgroups(p->kind, c->num_cus);
    if (L->coop_workgroups > 0) scap = std::min(scap, L->coop_workgroups);
    int64_t max_n = 0;
    ivec blk_all;
    L->ptm_wide_wanted = false;
    cvec wide_comp((size_t)L->ncomp, 0);
    auto wide_ptm_ok = [&](int cc) {
        if (p->kind != KIND_BA || L->ptm_stream == 0 || p->ncam_blocks <= 0 || (L->lds_resident == 0 && L->ptm_stream != 2)) return false;
        if (L->factor_rounding == 1 || L->emulate_stale) return false;   // (instantiated for the cooperative and LDS-resident solvers only)
        const int c0 = L->h_fac_ptr[(size_t)cc], c1 = L->h_fac_ptr[(size_t)cc + 1];
        if (p->h_blk_stamp.empty()) { p->h_blk_stamp.assign((size_t)p->N, 0); p->h_blk_idx.assign((size_t)p->N, 0); }
        const int stamp = ++p->stamp;
        int ncb = 0;
        for (int j = c0; j < c1; ++j) {
            const int b = p->h_cam[(size_t)L->h_fac_id[(size_t)j]];
            if (p->h_blk_stamp[(size_t)b] != stamp) { p->h_blk_stamp[(size_t)b] = stamp; ++ncb; }
        }
        for (int i = L->h_free_ptr[(size_t)cc]; i < L->h_free_ptr[(size_t)cc + 1]; ++i) {
            const int v = L->h_free_vid[(size_t)i];
            const int b = p->h_block_of[(size_t)v];
            if (b >= 0) { if (p->h_blk_stamp[(size_t)b] != stamp) { p->h_blk_stamp[(size_t)b] = stamp; ++ncb; } }
            else if (p->h_ptblock_of[(size_t)v] < 0) return false;
        }
        // (more cameras than fit: a wide group whose workgroups keep their own cameras only -- decided with the tables below)
        return ncb <= PTM_MAX_CAMERAS && (ptm_bytes_for(ncb, PTM_WIDE_THREADS) <= c->lds_limit || (L->ptm_local_cameras != 0 && !L->ptm_local_off));
    };
    // (one allocation: growing this by appending costs a transient call on ladybug 1.5 ms in page faults)
    L->h_coop_ints.reserve((size_t)(any_big ? 12 * L->nfac + 4 * (L->nfac + L->nfree) + 4096 * (int64_t)std::min<int64_t>(L->ncomp, 64) : 0));
    for (int cc : L->h_order) {  // heaviest first
        const int64_t m = L->h_fac_ptr[(size_t)cc + 1] - L->h_fac_ptr[(size_t)cc];
        const int64_t n = L->h_free_ptr[(size_t)cc + 1] - L->h_free_ptr[(size_t)cc];
        const int64_t need = groups_of(cc);
        const bool grouped = m >= group_min;
        const bool big = any_big && (grouped || (m >= L->coop_min_factors && (int)(L->coop.size() + L->stream.size()) < L->coop_max_components));
        const bool take = big && cap > 0 && need <= cap && coop_on;
        // A bundle-adjustment component too large for a cooperative group whose CAMERA blocks fit a compute unit's LDS goes to the
        // point-major streaming solver as a wide group -- a workgroup per compute unit on the one component (solver_ptm.hpp: a trial
        // streams 18 bytes a factor and 48 a point block, nothing is written; the grid solver below forms every trial point in x[]
        // and gathers 24 doubles a factor through L2).  It joins the batch list; the tables below decide (cameras, LDS).
        if (!take && big && wide_ptm_ok(cc)) { L->h_rest.push_back(cc); L->ptm_wide_wanted = true; wide_comp[(size_t)cc] = 1; continue; }
        if (!take && big && scap > 0) {
            // too large for the register-resident solver (or not bundle adjustment): the streaming
            // grid solver; about two factors per lane and trial point, at most what is resident
            L->stream.emplace_back();
            StreamItem& st = L->stream.back();
            st.comp = cc;
            st.nwg = (int)std::max<int64_t>(1, std::min<int64_t>(scap, (std::max(m, n) + 2 * STREAM_THREADS - 1) / (2 * STREAM_THREADS)));
            const int f0s = L->h_free_ptr[(size_t)cc];
            ivec longv;
            for (int64_t i = 0; i < n; ++i)
                if (L->h_v2s_ptr[(size_t)(f0s + i) + 1] - L->h_v2s_ptr[(size_t)(f0s + i)] > STREAM_LONG_LIST) longv.push_back((int)i);
            st.nlong = (int)longv.size();
            int rcs = plan_alloc(L, st.long_vars, std::max<size_t>(longv.size(), 1) * sizeof(int));
            if (rcs) return rcs;
            if (!longv.empty()) {
                HIPCHK(c, hipMemcpyAsync(st.long_vars.p, longv.data(), longv.size() * sizeof(int), hipMemcpyHostToDevice, c->stream));
                HIPCHK(c, hipStreamSynchronize(c->stream));  // local
            }
            continue;
        }
        if (!take) { L->h_rest.push_back(cc); continue; }
        L->coop.emplace_back();
        CoopItem& it = L->coop.back();
        it.comp = cc;
        it.nwg = (int)need;
        it.xi_off = (size_t)max_n;   // (max_n: running total of the cooperative components' variables)
        const int f0 = L->h_free_ptr[(size_t)cc], c0 = L->h_fac_ptr[(size_t)cc];
        // local free index of each factor slot: plan_create's table, or (a plan that did not keep it) formed
        // here from the per-problem marks, valid for one stamp
        ivec sl_own;
        const int* sl = nullptr;
        if (!L->h_slot_li.empty()) sl = L->h_slot_li.data() + 12 * (size_t)c0;
        else {
            for (int64_t i = 0; i < n; ++i) p->h_local[(size_t)L->h_free_vid[(size_t)(f0 + i)]] = (int)i;
            const int stamp = ++p->stamp;
            for (int64_t i = 0; i < n; ++i) p->h_owner_stamp[(size_t)L->h_free_vid[(size_t)(f0 + i)]] = stamp;
            sl_own.resize((size_t)(12 * m));
            for (int64_t j = 0; j < m; ++j) {
                const int f = L->h_fac_id[(size_t)(c0 + j)];
                for (int k = 0; k < 12; ++k) {
                    const int v = p->var_of(f, k);
                    sl_own[(size_t)(12 * j + k)] = p->h_owner_stamp[(size_t)v] == stamp ? p->h_local[(size_t)v] : -1;
                }
            }
            sl = sl_own.data();
        }
        // owners of the CG recurrence: a lane per variable, a whole wave for variables fed by
        // many partials (longest first), see solver_coop.hpp
        const int lanes = it.nwg * L->coop_lanes(), waves = lanes / 64;
        ivec lane_var((size_t)lanes, -1), wave_var((size_t)waves, -1), longv;
        for (int64_t i = 0; i < n; ++i)
            if (L->h_v2s_ptr[(size_t)(f0 + i) + 1] - L->h_v2s_ptr[(size_t)(f0 + i)] > COOP_LONG_LIST) longv.push_back((int)i);
        std::stable_sort(longv.begin(), longv.end(), [&](int a, int b) {
            return (L->h_v2s_ptr[(size_t)(f0 + a) + 1] - L->h_v2s_ptr[(size_t)(f0 + a)]) >
                   (L->h_v2s_ptr[(size_t)(f0 + b) + 1] - L->h_v2s_ptr[(size_t)(f0 + b)]);
        });
        cvec wave_owned((size_t)n, 0);
        for (size_t k = 0; k < longv.size() && (int)k < waves; ++k) { wave_var[k] = longv[k]; wave_owned[(size_t)longv[k]] = 1; }
        for (int64_t i = 0; i < n; ++i) if (!wave_owned[(size_t)i]) lane_var[(size_t)i] = (int)i;
        auto append = [&](const int* v, size_t count) {
            const size_t off = L->h_coop_ints.size();
            L->h_coop_ints.insert(L->h_coop_ints.end(), v, v + count);
            L->h_coop_ints.resize((L->h_coop_ints.size() + 63) / 64 * 64, -1);
            return off;
        };
        it.slot_li = append(sl, (size_t)(12 * m));
        it.lane_var = append(lane_var.data(), lane_var.size());
        it.wave_var = append(wave_var.data(), wave_var.size());
        max_n += n;
    }
    // tiny bundle-adjustment components (at most QUAD_MAX_VARS free variables: a point against fixed
    // cameras) go first in the batch list: four lanes each (solver_quad.hpp) instead of a workgroup
    {
        auto tiny = [&](int cc) {
            return p->kind == KIND_BA && L->quad_max_vars > 0 &&
                   L->h_free_ptr[(size_t)cc + 1] - L->h_free_ptr[(size_t)cc] <= std::min(L->quad_max_vars, QUAD_MAX_VARS);
        };
        // ... when there are enough of them to fill the device: below that a wave's sixteen components
        // finish at different times and the wave runs as long as its slowest (measured: ladybug's 7776
        // points 2.8 ms with a workgroup each, 4.2 ms as quads; 31104 synthetic points 4.5 vs 2.3 ms)
        int64_t ntiny = 0;
        for (int cc : L->h_rest) ntiny += tiny(cc) ? 1 : 0;
        L->rest_tiny = 0;
        L->tiny_group = ntiny >= L->quad_min_components ? 4 : ntiny >= L->row_min_components ? 16 : 0;
        if (L->tiny_group != 0) {
            auto mid = std::stable_partition(L->h_rest.begin(), L->h_rest.end(), tiny);
            L->rest_tiny = (int)(mid - L->h_rest.begin());
        }
    }
    L->rest_rot_mode = ROT_PER_FACTOR;
    if (p->kind == KIND_BA && L->camera_records != 0 && p->ncam_blocks > 0 && !L->h_rest.empty()) {
        bool camfix = true;
        for (size_t i = 0; i < L->h_rest.size() && camfix; ++i) {
            const int cc = L->h_rest[i];
            for (int k = L->h_free_ptr[(size_t)cc]; k < L->h_free_ptr[(size_t)cc + 1] && camfix; ++k)
                camfix = p->h_block_of[(size_t)L->h_free_vid[(size_t)k]] < 0;
        }
        // Free cameras: rewriting their records at every trial point puts one lane's rotation latency in
        // front of the evaluation and takes the rotation out of every factor -- a gain only where a lane
        // has many factors per camera (64 components of 31843 factors: 74 against 79 ms; no difference
        // for ladybug's 49 camera components or 1000 small components).
        int64_t mf = 0;
        for (int cc : L->h_rest) mf = std::max<int64_t>(mf, L->h_fac_ptr[(size_t)cc + 1] - L->h_fac_ptr[(size_t)cc]);
        L->rest_rot_mode = camfix ? ROT_CAMFIX : (L->camera_records == 2 || mf > 2048) ? ROT_RECORDS : ROT_PER_FACTOR;
    }
    // Slot tables.  The LDS-resident solver (solver_lds.hpp) takes the bundle-adjustment components of the
    // batch list whose variables -- free ones and the constants their factors read -- fit a compute unit's
    // LDS as slots: camera blocks (9 slots each, ascending by id), then point blocks (3 each, ascending).
    // Of the others, those whose CAMERA blocks fit go to the point-major streaming solver (solver_ptm.hpp:
    // point blocks as records in HBM, ordered by their number of factors, descending).  Both move to the
    // end of the list (streaming ones first); what fits neither stays with solver_wg.hpp.
    L->rest_lds = L->rest_ptm = 0;
    L->h_lds_ints.clear();
    L->h_pm_jg.clear();
    L->lds_ns_cap = L->lds_ncb_cap = L->lds_chunk_cap = L->ptm_ncb_cap = 0;
    L->rounds_threads = L->rounds_K = 0;
    L->lds_max_factors = 0;
    L->pm_blocks = L->pm_entries = 0;
    if (p->kind == KIND_BA && (L->lds_resident != 0 || L->ptm_stream != 0) && p->ncam_blocks > 0 && (int)L->h_rest.size() > L->rest_tiny) {
        if (p->h_blk_stamp.empty()) { p->h_blk_stamp.assign((size_t)p->N, 0); p->h_blk_idx.assign((size_t)p->N, 0); }
        const size_t nc = (size_t)L->ncomp;
        cvec kind_of(nc, 0);   // 1 = LDS-resident, 2 = point-major streaming
        // (the staging area of its gradient grows with the workgroup; a plan with a component meant for a wide group keeps to that group's 512 lanes)
        const int ptm_max_threads = L->ptm_threads ? L->ptm_threads : L->ptm_wide_wanted ? PTM_WIDE_THREADS : 768;
        ivec ls_ncb(nc, 0), ls_gcount(nc, 0), pm_pt0(nc, 0), ls_fidx((size_t)L->nfac, 0);
        ivec ls_pidx((size_t)L->nfac, 0);   // (host only) a listed factor's point block within its component: all 31 bits of it
        std::vector<ivec> vid_of(nc), free_of(nc), gp_of(nc), pptr_of(nc);
        ivec cams, pts, deg;
        bool local_failed = false;
        L->ptm_local = false; L->ptm_local_K = 0; L->ptm_local_comp = -1;
        L->h_lc.clear(); L->h_lc_off.clear(); L->h_cr_ptr.clear(); L->h_cr.clear(); L->h_wg_chunk0.clear(); L->h_pm_lcam.clear();
        std::vector<ivec> local_cams;   // per workgroup of the local group: its cameras (component numbers), ascending
        for (size_t r = (size_t)L->rest_tiny; r < L->h_rest.size(); ++r) {
            const int cc = L->h_rest[r];
            const int f0 = L->h_free_ptr[(size_t)cc], f1 = L->h_free_ptr[(size_t)cc + 1];
            const int c0 = L->h_fac_ptr[(size_t)cc], c1 = L->h_fac_ptr[(size_t)cc + 1];
            if (c1 == c0) continue;   // (an empty factor list needs no table: solver_wg.hpp returns 0 for it)
            const int stamp = ++p->stamp;
            cams.clear(); pts.clear();
            auto note = [&](int b, ivec& list) {
                if (p->h_blk_stamp[(size_t)b] != stamp) { p->h_blk_stamp[(size_t)b] = stamp; list.push_back(b); }
            };
            bool ok = true, free_cam = false;
            for (int j = c0; j < c1; ++j) { const int f = L->h_fac_id[(size_t)j]; note(p->h_cam[(size_t)f], cams); note(p->h_pt[(size_t)f], pts); }
            for (int i = f0; i < f1 && ok; ++i) {
                const int v = L->h_free_vid[(size_t)i];
                if (p->h_block_of[(size_t)v] >= 0) { note(p->h_block_of[(size_t)v], cams); free_cam = true; }
                else if (p->h_ptblock_of[(size_t)v] >= 0) note(p->h_ptblock_of[(size_t)v], pts);
                else ok = false;   // a variable no factor of the problem reads: no block to put it in
            }
            if (!ok || cams.size() > (size_t)PTM_MAX_CAMERAS) continue;
            const bool many_points = pts.size() >= (1u << 20);   // (beyond the LDS-resident solver's slot word; the streaming tables use ls_pidx)
            const int ncb = (int)cams.size(), npb = (int)pts.size(), ns = 9 * ncb + 3 * npb, m = c1 - c0;
            std::sort(cams.begin(), cams.end());
            for (int k = 0; k < ncb; ++k) p->h_blk_idx[(size_t)cams[(size_t)k]] = k;
            // the order of the gradient pass: camera by camera (listed order within a camera), whole waves per camera --
            // where a camera variable is free; otherwise nothing is summed per camera and the listed order serves
            ivec& gp = gp_of[(size_t)cc];
            if (free_cam) {
                ivec start((size_t)ncb + 1, 0);
                for (int j = c0; j < c1; ++j) ++start[(size_t)p->h_blk_idx[(size_t)p->h_cam[(size_t)L->h_fac_id[(size_t)j]]] + 1];
                for (int k = 0; k < ncb; ++k) start[(size_t)k + 1] = start[(size_t)k] + (start[(size_t)k + 1] + 63) / 64 * 64;
                gp.assign((size_t)start[(size_t)ncb], -1);
                for (int j = c0; j < c1; ++j) gp[(size_t)start[(size_t)p->h_blk_idx[(size_t)p->h_cam[(size_t)L->h_fac_id[(size_t)j]]]]++] = j - c0;
            } else {
                gp.assign((size_t)((m + 63) / 64 * 64), -1);
                std::iota(gp.begin(), gp.begin() + m, 0);
            }
            const int nchunk = (int)gp.size() / 64;
            // (the streaming solver is for what is too LARGE for the LDS: with lds_resident = 0 such components stay with
            // solver_wg.hpp -- the comparison the bit-identity tests make; ptm_stream = 2 sends everything its tables fit)
            const bool lds_size_ok = !many_points && lds_bytes_for(ns, ncb, nchunk) <= c->lds_limit;
            const bool fits_lds = L->lds_resident != 0 && lds_size_ok && L->ptm_stream != 2;
            const bool want_local = wide_comp[(size_t)cc] && !L->ptm_local_off && L->ptm_local_cameras != 0 && !local_failed &&
                                    (L->ptm_local_cameras == 1 || ptm_bytes_for(ncb, PTM_WIDE_THREADS) > c->lds_limit);
            const bool fits_ptm = !fits_lds && (L->ptm_stream == 2 || (L->ptm_stream == 1 && !lds_size_ok)) &&
                                  (want_local || ptm_bytes_for(ncb, ptm_max_threads) <= c->lds_limit);
            if (want_local && L->ptm_local) { local_failed = true; gp.clear(); continue; }   // (one such component a plan)
            if (!fits_lds && !fits_ptm) { gp.clear(); continue; }
            if (fits_lds) std::sort(pts.begin(), pts.end());
            else {
                // By number of factors, descending: the lanes of a wave run loops of equal length.  Among blocks of equal
                // count by their cameras, in listed order, lexicographically: neighbours in a wave-chunk then read the SAME
                // camera slots at the same time -- one LDS access serves them all, and the few distinct cameras of a chunk
                // are neighbours too, which keeps them on different banks (random cameras: 54 % of the LDS cycles were bank
                // conflicts, profiles/r03_a_pmc_lds_synthL.txt).  Ties: ascending id.
                std::sort(pts.begin(), pts.end());
                for (int k = 0; k < npb; ++k) p->h_blk_idx[(size_t)pts[(size_t)k]] = k;
                deg.assign((size_t)npb + 1, 0);
                for (int j = c0; j < c1; ++j) ++deg[(size_t)p->h_blk_idx[(size_t)p->h_pt[(size_t)L->h_fac_id[(size_t)j]]] + 1];
                for (int k = 0; k < npb; ++k) deg[(size_t)k + 1] += deg[(size_t)k];   // (now a CSR over the blocks in id order)
                ivec pcam((size_t)m), fill(deg.begin(), deg.end() - 1);
                for (int j = c0; j < c1; ++j) {
                    const int f = L->h_fac_id[(size_t)j];
                    // (camera numbers within the component: h_blk_idx of the camera blocks was set above and is still valid for them)
                    pcam[(size_t)fill[(size_t)p->h_blk_idx[(size_t)p->h_pt[(size_t)f]]]++] = p->h_blk_idx[(size_t)p->h_cam[(size_t)f]];
                }
                ivec ord((size_t)npb);
                std::iota(ord.begin(), ord.end(), 0);
                std::sort(ord.begin(), ord.end(), [&](int a, int b) {
                    const int da = deg[(size_t)a + 1] - deg[(size_t)a], db = deg[(size_t)b + 1] - deg[(size_t)b];
                    if (da != db) return da > db;
                    const int* pa = pcam.data() + deg[(size_t)a];
                    const int* pb = pcam.data() + deg[(size_t)b];
                    for (int t = 0; t < da; ++t) if (pa[t] != pb[t]) return pa[t] < pb[t];
                    return a < b;
                });
                // Whole wave-chunks of equal slot count are then dealt out over PTM_SPREAD runs of the sorted order: the chunks
                // that the waves of a workgroup evaluate at the same time come from different runs and meet different cameras
                // (the gradient's round sums, solver_ptm.hpp, are as long as a round's longest camera segment).
                if (want_local) {
                    // LOCAL: workgroup r owns a CONTIGUOUS slice of every run of chunks of equal slot count (the runs stand in
                    // camera order: a slice meets few cameras; a slice of every run: equal work), its positions in the chunk order
                    // are consecutive ([h_wg_chunk0[r], h_wg_chunk0[r + 1])), and inside them wave w (position - first mod 8) takes a
                    // contiguous eighth of the workgroup's chunks: the waves of a workgroup meet different cameras at a time
                    const int nfull = npb / 64, npc_all = (npb + 63) / 64;
                    const int Kl = (int)std::min<int64_t>(std::min<int64_t>(c->num_cus, PTM_WIDE_MAX_GROUP), std::max<int64_t>(1, npc_all / 24));
                    if (Kl <= PTM_MAX_GROUP) local_failed = true;
                    else {
                        std::vector<ivec> wl((size_t)Kl);
                        for (int a0 = 0; a0 < nfull;) {
                            const int T = deg[(size_t)ord[(size_t)(64 * a0)] + 1] - deg[(size_t)ord[(size_t)(64 * a0)]];
                            int a1 = a0;
                            while (a1 < nfull && deg[(size_t)ord[(size_t)(64 * a1)] + 1] - deg[(size_t)ord[(size_t)(64 * a1)]] == T) ++a1;
                            const long long mm = a1 - a0;
                            for (int rk = 0; rk < Kl; ++rk)
                                for (long long a = a0 + rk * mm / Kl; a < a0 + (rk + 1) * mm / Kl; ++a) wl[(size_t)rk].push_back((int)a);
                            a0 = a1;
                        }
                        ivec chunk_of((size_t)nfull);
                        L->h_wg_chunk0.assign((size_t)Kl + 1, 0);
                        int pos = 0;
                        for (int rk = 0; rk < Kl; ++rk) {
                            const ivec& li = wl[(size_t)rk];
                            const int nr = (int)li.size(), nwv = PTM_WIDE_THREADS / 64;
                            L->h_wg_chunk0[(size_t)rk] = pos;
                            int taken = 0;
                            for (int w = 0; w < nwv; ++w) {   // wave w's positions: first + w, first + w + 8, ... -- the next (nr - w + 7) / 8 chunks of the list
                                const int cnt = nr > w ? (nr - w + nwv - 1) / nwv : 0;
                                for (int j = 0; j < cnt; ++j) chunk_of[(size_t)(pos + w + nwv * j)] = li[(size_t)(taken + j)];
                                taken += cnt;
                            }
                            pos += nr;
                        }
                        L->h_wg_chunk0[(size_t)Kl] = npc_all;   // (the last workgroup also takes the chunk of the npb % 64 blocks left over)
                        ivec ord2(ord);
                        for (int a = 0; a < nfull; ++a)
                            for (int l = 0; l < 64; ++l) ord2[(size_t)(64 * a + l)] = ord[(size_t)(64 * chunk_of[(size_t)a] + l)];
                        ord.swap(ord2);
                        // every workgroup's cameras
                        local_cams.assign((size_t)Kl, ivec());
                        const size_t cam_cap = [&] { size_t k = 1; while (k < 255 && ptm_bytes_for((int)k + 1, PTM_WIDE_THREADS) <= c->lds_limit) ++k; return k; }();
                        ivec mark((size_t)ncb, -1);
                        size_t worst = 0;
                        for (int rk = 0; rk < Kl && !local_failed; ++rk) {
                            ivec& lc = local_cams[(size_t)rk];
                            for (int k = 64 * L->h_wg_chunk0[(size_t)rk]; k < std::min(npb, 64 * L->h_wg_chunk0[(size_t)rk + 1]); ++k) {
                                const int a = ord[(size_t)k];
                                for (int t = deg[(size_t)a]; t < deg[(size_t)a + 1]; ++t)
                                    if (mark[(size_t)pcam[(size_t)t]] != rk) { mark[(size_t)pcam[(size_t)t]] = rk; lc.push_back(pcam[(size_t)t]); }
                            }
                            std::sort(lc.begin(), lc.end());
                            if (lc.empty()) lc.push_back(0);   // (a workgroup without chunks still has its LDS laid out for one camera)
                            worst = std::max(worst, lc.size());
                            if (lc.size() > cam_cap) local_failed = true;
                        }
                        if (std::getenv("RDIS_HIP_LOCAL_STATS"))
                            std::fprintf(stderr, "local cameras: %d workgroups over %d chunks, %d cameras in the component, at most %zu in a workgroup (the LDS holds %zu)\n",
                                         Kl, npc_all, ncb, worst, cam_cap);
                        if (!local_failed) { L->ptm_local = true; L->ptm_local_K = Kl; L->ptm_local_comp = cc; }
                    }
                    if (local_failed) { gp.clear(); continue; }
                } else {
                    const int nfull = npb / 64;
                    ivec chunk_of((size_t)nfull);
                    int pos = 0;
                    for (int a0 = 0; a0 < nfull;) {
                        const int T = deg[(size_t)ord[(size_t)(64 * a0)] + 1] - deg[(size_t)ord[(size_t)(64 * a0)]];
                        int a1 = a0;
                        while (a1 < nfull && deg[(size_t)ord[(size_t)(64 * a1)] + 1] - deg[(size_t)ord[(size_t)(64 * a1)]] == T) ++a1;
                        const int mm = a1 - a0, q = (mm + PTM_SPREAD - 1) / PTM_SPREAD;
                        if (!wide_comp[(size_t)cc]) {
                            for (int rr = 0; rr < q; ++rr)
                                for (int gg = 0; gg < PTM_SPREAD; ++gg) { const int idx = gg * q + rr; if (idx < mm) chunk_of[(size_t)pos++] = a0 + idx; }
                        } else {
                            // A wide group deals chunk c to workgroup c mod K, wave (c / K) mod waves -- with K a multiple of
                            // PTM_SPREAD the round robin above would hand all the waves of a workgroup neighbours of ONE run, i.e.
                            // one camera: a gradient round's sums (one lane per camera entry, solver_ptm.hpp) 512 rows long, 22 000
                            // of a round's 26 000 cycles at 8e6 factors.  Here position c takes the next chunk of run h(c), h a
                            // weighted sum of c's hexadecimal digits mod 16: the positions c, c + K, c + 2 K, ... of a workgroup's
                            // waves meet different runs for every K that occurs (searched over K = 1 .. 16, 32 .. 512).
                            int used[PTM_SPREAD] = {};
                            for (int t = 0; t < mm; ++t) {
                                int gg = ((pos & 15) + 15 * ((pos >> 4) & 15) + ((pos >> 8) & 15) + 15 * ((pos >> 12) & 15) + ((pos >> 16) & 15) + ((pos >> 20) & 15)) % PTM_SPREAD;
                                for (int tries = 0; tries < PTM_SPREAD && gg * q + used[gg] >= std::min(mm, (gg + 1) * q); ++tries) gg = (gg + 1) % PTM_SPREAD;
                                chunk_of[(size_t)pos++] = a0 + gg * q + used[gg]++;
                            }
                        }
                        a0 = a1;
                    }
                    ivec ord2(ord);
                    for (int a = 0; a < nfull; ++a)
                        for (int l = 0; l < 64; ++l) ord2[(size_t)(64 * a + l)] = ord[(size_t)(64 * chunk_of[(size_t)a] + l)];
                    ord.swap(ord2);
                }
                ivec pts2((size_t)npb);
                for (int k = 0; k < npb; ++k) pts2[(size_t)k] = pts[(size_t)ord[(size_t)k]];
                pts.swap(pts2);
            }
            for (int k = 0; k < npb; ++k) p->h_blk_idx[(size_t)pts[(size_t)k]] = k;
            // local free index of the component's variables (the per-problem arrays are valid for one stamp)
            for (int i = f0; i < f1; ++i) { const int v = L->h_free_vid[(size_t)i]; p->h_owner_stamp[(size_t)v] = stamp; p->h_local[(size_t)v] = i - f0; }
            ivec& sv = vid_of[(size_t)cc];
            ivec& sf = free_of[(size_t)cc];
            sv.reserve((size_t)ns); sf.reserve((size_t)ns);
            auto slot = [&](int v) { sv.push_back(v); sf.push_back(p->h_owner_stamp[(size_t)v] == stamp ? p->h_local[(size_t)v] : -1); };
            if (fits_lds) { for (int b : cams) for (int k = 0; k < 9; ++k) slot(b + k); }
            else {   // the streaming solver's camera slots: ten per block, [t f k1 k2 | r | pad] (ptm_api.hpp)
                for (int b : cams)
                    for (int q = 0; q < PTM_CS; ++q) {
                        const int k = ptm_var_of(q);
                        if (k >= 0) slot(b + k); else { sv.push_back(b); sf.push_back(-1); }
                    }
            }
            for (int b : pts) for (int k = 0; k < 3; ++k) slot(b + k);
            for (int j = c0; j < c1; ++j) {
                const int f = L->h_fac_id[(size_t)j];
                ls_fidx[(size_t)j] = (int)((unsigned)p->h_blk_idx[(size_t)p->h_cam[(size_t)f]] | ((unsigned)p->h_blk_idx[(size_t)p->h_pt[(size_t)f]] << 12));
                ls_pidx[(size_t)j] = p->h_blk_idx[(size_t)p->h_pt[(size_t)f]];
            }
            ls_ncb[(size_t)cc] = ncb;
            ls_gcount[(size_t)cc] = nchunk;
            if (fits_lds) {
                kind_of[(size_t)cc] = 1;
                L->lds_ns_cap = std::max(L->lds_ns_cap, ns);
                L->lds_ncb_cap = std::max(L->lds_ncb_cap, ncb);
                L->lds_chunk_cap = std::max(L->lds_chunk_cap, nchunk);
                L->lds_max_factors = std::max<int64_t>(L->lds_max_factors, m);
            } else {
                kind_of[(size_t)cc] = 2;
                if (L->ptm_local && L->ptm_local_comp == cc) {   // (a workgroup's LDS holds its own cameras only)
                    for (const ivec& lc : local_cams) L->ptm_ncb_cap = std::max(L->ptm_ncb_cap, (int)lc.size());
                } else {
                    L->ptm_ncb_cap = std::max(L->ptm_ncb_cap, ncb);
                }
                // the point's factors, in listed order (a CSR over the point blocks in their slot order)
                ivec& pp = pptr_of[(size_t)cc];
                pp.assign((size_t)npb + 1, 0);
                for (int j = c0; j < c1; ++j) ++pp[(size_t)ls_pidx[(size_t)j] + 1];
                for (int k = 0; k < npb; ++k) pp[(size_t)k + 1] += pp[(size_t)k];
            }
        }
        // (the maxima of a launch may come from different components: its LDS must hold them together)
        if (L->lds_ns_cap > 0 && lds_bytes_for(L->lds_ns_cap, L->lds_ncb_cap, L->lds_chunk_cap) > c->lds_limit)
            for (size_t cc = 0; cc < nc; ++cc) if (kind_of[cc] == 1) kind_of[cc] = 0;
        auto first_other = std::stable_partition(L->h_rest.begin() + L->rest_tiny, L->h_rest.end(), [&](int cc) { return kind_of[(size_t)cc] == 0; });
        auto first_lds = std::stable_partition(first_other, L->h_rest.end(), [&](int cc) { return kind_of[(size_t)cc] == 2; });
        L->rest_ptm = (int)(first_lds - first_other);
        L->rest_lds = (int)(L->h_rest.end() - first_lds);
        // local camera numbering did not work out (a workgroup's cameras beyond the LDS, too few chunks, a second streaming
        // component in the plan): once more without it -- the component then takes the grid solver
        if (local_failed || (L->ptm_local && L->rest_ptm != 1)) {
            L->ptm_local = false;
            L->ptm_local_off = true;
            return prepare_partition(L);
        }
        if (L->rest_ptm + L->rest_lds > 0) {
            // one int32 block, tables in component order (a component without one has empty ranges)
            ivec sptr(nc + 1, 0), gptr(nc + 1, 0), svid, sfree, gperm;
            for (size_t cc = 0; cc < nc; ++cc) {
                const bool has = kind_of[cc] != 0;
                sptr[cc + 1] = sptr[cc] + (has ? (int)vid_of[cc].size() : 0);
                gptr[cc + 1] = gptr[cc] + (has ? ls_gcount[cc] : 0);
                if (!has) continue;
                svid.insert(svid.end(), vid_of[cc].begin(), vid_of[cc].end());
                sfree.insert(sfree.end(), free_of[cc].begin(), free_of[cc].end());
                gperm.insert(gperm.end(), gp_of[cc].begin(), gp_of[cc].end());
            }
            // Point-major factor order of the streaming components.  A component's point blocks stand in slot order
            // (by number of factors, descending) and are taken 64 at a time -- a wave-chunk, a lane per block.  The
            // factors of a chunk's blocks are laid out slot-major: entry cptr[chunk] + 64 t + lane is the t-th listed
            // factor of the lane's block (or no factor: -1), so a wave's loads of a slot are 64 neighbours and their
            // addresses depend on nothing the wave has loaded before.  entry -> listed factor (plan-wide index).
            ivec pm_ch0v(nc, 0), cptr;
            L->h_pm_jg.clear();
            for (size_t cc = 0; cc < nc; ++cc) {
                if (kind_of[cc] != 2) continue;
                const int c0 = L->h_fac_ptr[cc], c1 = L->h_fac_ptr[cc + 1], npb = (int)pptr_of[cc].size() - 1, npc = (npb + 63) / 64;
                pm_pt0[cc] = (int)L->pm_blocks;
                pm_ch0v[cc] = (int)cptr.size();

                const int e0 = (int)L->h_pm_jg.size();
                ivec cbase((size_t)npc + 1, 0);
                for (int ch = 0; ch < npc; ++ch)   // (descending: a chunk's first block has the most factors)
                    cbase[(size_t)ch + 1] = cbase[(size_t)ch] + 64 * (pptr_of[cc][(size_t)(64 * ch) + 1] - pptr_of[cc][(size_t)(64 * ch)]);
                L->h_pm_jg.resize((size_t)e0 + (size_t)cbase[(size_t)npc], -1);
                ivec fill((size_t)npb, 0);
                for (int j = c0; j < c1; ++j) {
                    const int pi = ls_pidx[(size_t)j];
                    const int e = e0 + cbase[(size_t)(pi / 64)] + 64 * fill[(size_t)pi]++ + (pi % 64);
                    L->h_pm_jg[(size_t)e] = j;
                }
                for (int k = 0; k <= npc; ++k) cptr.push_back(e0 + cbase[(size_t)k]);
                L->pm_blocks += npb;
                if (L->ptm_local && L->ptm_local_comp == (int)cc) {
                    // the tables of the local group (ptm_api.hpp: PtmGroupArgs): per workgroup its cameras and chunk range, per entry
                    // of the factor stream the camera's number in the workgroup that owns the chunk, per component camera who holds it
                    const int Kl = L->ptm_local_K, ncbg = ls_ncb[cc];
                    L->h_pm_lcam.assign((size_t)cbase[(size_t)npc] + 64 * PTM_BLK, (short)-1);
                    L->h_lc_off.assign((size_t)Kl, 0);
                    std::vector<ivec> holders((size_t)ncbg);
                    ivec g2l((size_t)ncbg, -1);
                    for (int rk = 0; rk < Kl; ++rk) {
                        const ivec& lc = local_cams[(size_t)rk];
                        for (size_t k = 0; k < lc.size(); ++k) g2l[(size_t)lc[k]] = (int)k;
                        L->h_lc_off[(size_t)rk] = (long long)L->h_lc.size();
                        L->h_lc.push_back((int)lc.size()); L->h_lc.push_back(L->h_wg_chunk0[(size_t)rk]); L->h_lc.push_back(L->h_wg_chunk0[(size_t)rk + 1]); L->h_lc.push_back(0);
                        L->h_lc.insert(L->h_lc.end(), lc.begin(), lc.end());
                        for (size_t k = 0; k < lc.size(); ++k) {   // (the first workgroup that holds a camera speaks for it)
                            L->h_lc.push_back(holders[(size_t)lc[k]].empty() ? 1 : 0);
                            holders[(size_t)lc[k]].push_back((rk << 8) | (int)k);
                        }
                        for (int e = cbase[(size_t)L->h_wg_chunk0[(size_t)rk]]; e < cbase[(size_t)L->h_wg_chunk0[(size_t)rk + 1]]; ++e) {
                            const int j = L->h_pm_jg[(size_t)(e0 + e)];
                            if (j >= 0) L->h_pm_lcam[(size_t)e] = (short)g2l[(size_t)(((unsigned)ls_fidx[(size_t)j]) & 0xFFFu)];
                        }
                    }
                    L->h_cr_ptr.assign(1, 0);
                    for (int g = 0; g < ncbg; ++g) {
                        L->h_cr.insert(L->h_cr.end(), holders[(size_t)g].begin(), holders[(size_t)g].end());
                        L->h_cr_ptr.push_back((int)L->h_cr.size());
                    }
                }
            }
            L->pm_entries = (int64_t)L->h_pm_jg.size();
            L->pm_cptr_len = (int64_t)cptr.size();
            L->ls_total_chunks = gptr[nc];
            L->ptm_min_points = INT64_MAX;
            for (size_t cc = 0; cc < nc; ++cc)
                if (kind_of[cc] == 2) L->ptm_min_points = std::min<int64_t>(L->ptm_min_points, (int64_t)pptr_of[cc].size() - 1);
            ivec& blk = L->h_lds_ints;
            auto put = [&](const ivec& v) { const size_t off = blk.size(); blk.insert(blk.end(), v.begin(), v.end()); return off; };
            L->off_ls_ptr = put(sptr); L->off_ls_vid = put(svid); L->off_ls_free = put(sfree);
            L->off_ls_ncb = put(ls_ncb); L->off_ls_fidx = put(ls_fidx); L->off_ls_gptr = put(gptr); L->off_ls_gperm = put(gperm);
            L->off_pm_pt0 = put(pm_pt0); L->off_pm_ch0 = put(pm_ch0v); L->off_pm_cptr = put(cptr); L->off_pm_jg = put(L->h_pm_jg);
            // rotations: no camera variable free among a launch's components -> records, read only; otherwise records
            // that follow the trial point when a lane has several factors per camera and trial (else each factor forms its own)
            auto camfix_of = [&](size_t r0, size_t r1) {
                for (size_t r = r0; r < r1; ++r) {
                    const int cc = L->h_rest[r];
                    for (int k = L->h_free_ptr[(size_t)cc]; k < L->h_free_ptr[(size_t)cc + 1]; ++k)
                        if (p->h_block_of[(size_t)L->h_free_vid[(size_t)k]] >= 0) return false;
                }
                return true;
            };
            const size_t r_lds = L->h_rest.size() - (size_t)L->rest_lds, r_ptm = r_lds - (size_t)L->rest_ptm;
            const bool records = L->lds_rot >= 0 ? L->lds_rot == 1 : L->lds_max_factors > 512;
            // (the stale-cache emulation is instantiated for per-factor rotations only: the same bits per factor, fewer kernels)
            L->lds_rot_mode = (L->camera_records == 0 || L->emulate_stale) ? ROT_PER_FACTOR : camfix_of(r_lds, L->h_rest.size()) ? ROT_CAMFIX : records ? ROT_RECORDS : ROT_PER_FACTOR;
            // (the streaming solver always works from per-camera records: rewritten at every trial point, or never)
            L->ptm_rot_mode = camfix_of(r_ptm, r_lds) ? ROT_CAMFIX : ROT_RECORDS;
        }
    }
    if (L->rest_tiny > 0 && L->group_blocks4 == 0) {
        int b4 = 0, b16 = 0;
        HIPCHK(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(&b4, cgd_group_kernel<4, QUAD_THREADS>, QUAD_THREADS, 0));
        HIPCHK(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(&b16, cgd_group_kernel<16, 64>, 64, 0));
        L->group_blocks4 = std::max(1, b4) * c->num_cus;
        L->group_blocks16 = std::max(1, b16) * c->num_cus;
    }
    int rc = plan_alloc(L, L->rest_order, std::max<size_t>(L->h_rest.size(), 1) * sizeof(int));
    if (!rc) rc = plan_alloc(L, L->queue, 256);
    if (!rc && max_n > 0) rc = plan_alloc(L, L->xi_glob, (size_t)max_n * sizeof(double));
    if (rc) return rc;
    if (L->rest_lds + L->rest_ptm > 0) {
        rc = plan_alloc(L, L->lds_ints, L->h_lds_ints.size() * sizeof(int));
        if (rc) return rc;
        HIPCHK(c, hipMemcpyAsync(L->lds_ints.p, L->h_lds_ints.data(), L->h_lds_ints.size() * sizeof(int), hipMemcpyHostToDevice, c->stream));
        rc = plan_alloc(L, L->lds_obs, (size_t)std::max<int64_t>(L->nfac, 1) * sizeof(double2));
        if (rc) return rc;
        gather_obs_kernel<<<grid_for(c, L->nfac, 256), 256, 0, c->stream>>>((int)L->nfac, L->ip(L->off_fac_id), p->obs.as<double2>(), L->lds_obs.as<double2>());
        HIPCHK(c, hipGetLastError());
    }
    if (L->rest_ptm > 0) {   // the streaming components' point records and point-major factor arrays
        rc = plan_alloc(L, L->pm_rec, (size_t)L->pm_blocks * PT_REC * sizeof(double));
        if (!rc) rc = plan_alloc(L, L->pm_gh, (size_t)L->pm_blocks * PT_REC * sizeof(double));
        if (!rc) rc = plan_alloc(L, L->pm_cbox, (size_t)std::max<int64_t>(L->pm_cptr_len, 1) * 8 * sizeof(float));
        if (!rc) rc = plan_alloc(L, L->pm_bex, (size_t)L->pm_blocks * PT_BND * sizeof(double));
        // (a block of PTM_BLK slots is loaded whole: the last chunk's may reach past the last entry)
        if (!rc) rc = plan_alloc(L, L->pm_cam, ((size_t)L->pm_entries + 64 * PTM_BLK) * sizeof(short));
        if (!rc) rc = plan_alloc(L, L->pm_obs, ((size_t)L->pm_entries + 64 * PTM_BLK) * sizeof(double2));
        if (rc) return rc;
        const PlanView V = L->view();
        HIPCHK(c, ptm_gather_launch(grid_for(c, L->pm_entries, 256), c->stream, (int)L->pm_entries, L->lds_ints.as<int>() + L->off_pm_jg, V.ls_fidx, V.ls_obs,
                                    L->pm_cam.as<short>(), L->pm_obs.as<double2>()));
        if (L->ptm_local) {   // the stream names a factor's camera by its number in the workgroup that owns the chunk; the group's tables
            HIPCHK(c, hipMemcpyAsync(L->pm_cam.p, L->h_pm_lcam.data(), std::min(L->h_pm_lcam.size(), (size_t)L->pm_entries + 64 * PTM_BLK) * sizeof(short),
                                     hipMemcpyHostToDevice, c->stream));
            rc = plan_alloc(L, L->lc_dev, L->h_lc.size() * sizeof(int));
            if (!rc) rc = plan_alloc(L, L->lc_off_dev, L->h_lc_off.size() * sizeof(long long));
            if (!rc) rc = plan_alloc(L, L->cr_ptr_dev, L->h_cr_ptr.size() * sizeof(int));
            if (!rc) rc = plan_alloc(L, L->cr_dev, std::max<size_t>(L->h_cr.size(), 1) * sizeof(int));
            if (!rc) rc = plan_alloc(L, L->ptm_tot, 2 * (size_t)PTM_CS * (L->h_cr_ptr.size() - 1) * sizeof(double));
            if (rc) return rc;
            HIPCHK(c, hipMemcpyAsync(L->lc_dev.p, L->h_lc.data(), L->h_lc.size() * sizeof(int), hipMemcpyHostToDevice, c->stream));
            HIPCHK(c, hipMemcpyAsync(L->lc_off_dev.p, L->h_lc_off.data(), L->h_lc_off.size() * sizeof(long long), hipMemcpyHostToDevice, c->stream));
            HIPCHK(c, hipMemcpyAsync(L->cr_ptr_dev.p, L->h_cr_ptr.data(), L->h_cr_ptr.size() * sizeof(int), hipMemcpyHostToDevice, c->stream));
            HIPCHK(c, hipMemcpyAsync(L->cr_dev.p, L->h_cr.data(), L->h_cr.size() * sizeof(int), hipMemcpyHostToDevice, c->stream));
        }
    }
    if (!L->coop.empty()) {
        rc = plan_alloc(L, L->coop_ints, L->h_coop_ints.size() * sizeof(int));
        if (rc) return rc;
        HIPCHK(c, hipMemcpyAsync(L->coop_ints.p, L->h_coop_ints.data(), L->h_coop_ints.size() * sizeof(int), hipMemcpyHostToDevice, c->stream));
    }
    // cooperative groups, packed into launches of at most `cap` workgroups
    size_t max_groups = 1;
    for (size_t i = 0; i < L->coop.size();) {
        CoopLaunch cl;
        cl.first = (int)i;
        std::vector<CoopGroup> hg;
        ivec hw;
        while (i < L->coop.size() && (cl.count == 0 || cl.total_wg + L->coop[i].nwg <= cap) && cl.count < COOP_MAX_GROUPS) {
            const CoopItem& it = L->coop[i];
            CoopGroup g{};
            g.a = CoopArgs{i == 0 ? p->coop_timing.as<long long>() : nullptr, nullptr /* set below */, L->coop_ints.as<int>() + it.slot_li,
                           L->coop_ints.as<int>() + it.lane_var, L->coop_ints.as<int>() + it.wave_var, L->xi_glob.as<double>() + it.xi_off, it.comp,
                           // (a small group's sweep is one entry per lane: polling early costs it less than waiting)
                           it.nwg * (L->coop_lanes() / 64) <= 64 ? std::min(4, L->coop_poll_delay) : L->coop_poll_delay,
                           L->coop_speculate, L->factor_rounding == 1 ? 1 : 0, (L->factor_rounding == 1 && L->emulate_stale) ? 1 : 0};
            g.wg0 = cl.total_wg; g.nwg = it.nwg;
            hg.push_back(g);
            hw.insert(hw.end(), (size_t)it.nwg, cl.count);
            cl.total_wg += it.nwg; ++cl.count; ++i;
        }
        max_groups = std::max(max_groups, hg.size());
        if (p->coop_state.bytes < max_groups * sizeof(CoopState)) {   // one exchange state per concurrent group
            // The buffer moves: every other plan of this problem has its address in its uploaded group
            // tables.  The generation tells rdis_hip_plan_solve to rebuild theirs before they run again.
            HIPCHK(c, hipStreamSynchronize(c->stream));
            if (c->aux) HIPCHK(c, hipStreamSynchronize(c->aux));
            rc = dalloc(c, p->coop_state, std::max(max_groups, 2 * (p->coop_state.bytes / sizeof(CoopState))) * sizeof(CoopState));
            if (rc) return rc;
            ++p->coop_state_gen;
        }
        rc = plan_alloc(L, cl.groups, hg.size() * sizeof(CoopGroup));
        if (!rc) rc = plan_alloc(L, cl.wg_group, hw.size() * sizeof(int));
        if (rc) return rc;
        L->coop_launches.push_back(std::move(cl));
        L->h_coop_groups.push_back(std::move(hg));
        L->h_coop_wg.push_back(std::move(hw));   // (host images are kept: the uploads are asynchronous)
        HIPCHK(c, hipMemcpyAsync(L->coop_launches.back().wg_group.p, L->h_coop_wg.back().data(),
                                 L->h_coop_wg.back().size() * sizeof(int), hipMemcpyHostToDevice, c->stream));
    }
    // (the state pointers only now: the buffer may have moved while the launches were sized)
    for (size_t l = 0; l < L->coop_launches.size(); ++l) {
        std::vector<CoopGroup>& hg = L->h_coop_groups[l];
        for (size_t g = 0; g < hg.size(); ++g) hg[g].a.st = p->coop_state.as<CoopState>() + g;
        HIPCHK(c, hipMemcpyAsync(L->coop_launches[l].groups.p, hg.data(), hg.size() * sizeof(CoopGroup), hipMemcpyHostToDevice, c->stream));
    }
    // (a transient plan lives until its results are fetched, which waits for the stream)
    if (!L->transient) HIPCHK(c, hipStreamSynchronize(c->stream));
    if (!L->h_rest.empty()) HIPCHK(c, hipMemcpyAsync(L->rest_order.p, L->h_rest.data(), L->h_rest.size() * sizeof(int), hipMemcpyHostToDevice, c->stream));
    L->partition_dirty = false;
    L->coop_state_gen = p->coop_state_gen;
    return 0;
}

template <int KIND>
int launch_wg(rdis_hip_plan* L, hipStream_t stream, int threads, int first, int grid, int maxiters, double ftol) {
    rdis_hip_ctx* c = L->prob->ctx;
    ProblemView P = L->prob->view();
    P.xrot = L->prob->xrot.as<double>(); P.rot_mode = L->rest_rot_mode;
    PlanView V = L->view();
    V.order += first;   // components [first, first + grid) of the batch list
    switch (threads) {
        case 64: cgd_wg_kernel<KIND, 64><<<grid, 64, 0, stream>>>(P, V, maxiters, ftol); break;
        case 128: cgd_wg_kernel<KIND, 128><<<grid, 128, 0, stream>>>(P, V, maxiters, ftol); break;
        case 256: cgd_wg_kernel<KIND, 256><<<grid, 256, 0, stream>>>(P, V, maxiters, ftol); break;
        case 512: cgd_wg_kernel<KIND, 512><<<grid, 512, 0, stream>>>(P, V, maxiters, ftol); break;
        case 768: cgd_wg_kernel<KIND, 768><<<grid, 768, 0, stream>>>(P, V, maxiters, ftol); break;
        default: cgd_wg_kernel<KIND, 1024><<<grid, 1024, 0, stream>>>(P, V, maxiters, ftol); break;
    }
    HIPCHK(c, hipGetLastError());
    return 0;
}
template <int ROT>
int launch_lds_rot(rdis_hip_plan* L, hipStream_t stream, int threads, int first, int grid, int maxiters, double ftol) {
    rdis_hip_ctx* c = L->prob->ctx;
    ProblemView P = L->prob->view();
    PlanView V = L->view();
    V.order += first;
    const size_t dyn = L->lds_dyn_bytes(c);
    const int nsc = L->lds_ns_cap, ncc = L->lds_ncb_cap, chc = L->lds_chunk_cap;
#define RDIS_LDS_LAUNCH(T)                                                                                              \
    do {                                                                                                                \
        if (dyn > 48 * 1024)                                                                                            \
            HIPCHK(c, hipFuncSetAttribute((const void*)cgd_lds_kernel<T, ROT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn)); \
        cgd_lds_kernel<T, ROT><<<grid, T, dyn, stream>>>(P, V, maxiters, ftol, nsc, ncc, chc);                                \
    } while (0)
    switch (threads) {
        case 64: RDIS_LDS_LAUNCH(64); break;
        case 128: RDIS_LDS_LAUNCH(128); break;
        case 256: RDIS_LDS_LAUNCH(256); break;
        case 512: RDIS_LDS_LAUNCH(512); break;
        case 768: RDIS_LDS_LAUNCH(768); break;
        default: RDIS_LDS_LAUNCH(1024); break;
    }
#undef RDIS_LDS_LAUNCH
    HIPCHK(c, hipGetLastError());
    return 0;
}
// The gradient's rounds of the point-major streaming solver (solver_ptm.hpp: gradient_to_xi) for workgroups of
// `threads` lanes, K to a component.  Workgroup (component, rank r) takes the point chunks c = r (mod K), its wave w
// those with (c / K) mod waves = w, slot by slot: that is the wave's sequence of steps, and round rr of the workgroup
// is every wave's rr-th step.  Within a round the factors are ranked by camera block (within a camera by wave and
// lane): a factor's rank is the staging row of its camera partials (pm_grow, two bytes per factor), and a round's
// table names per camera the first row of its segment [ncb + 1] -- the order of the sums.  Built on the host from the
// plan's point-major tables, once per (threads, K); nothing to build when no camera variable is free.
// A trial's work by wave (solver_ptm.hpp: eval_line), for the same workgroups: the workgroup's wave-chunks cut into blocks of
// PTM_BLK slots, the blocks in chunk order dealt out in equal CONTIGUOUS shares -- a wave's share is a run of whole chunks
// with at most a partial one at either end (a chunk's 64 point blocks times some of their slots).  By whole chunks 41 of them
// over 12 waves are four for some and three for the rest, and a trial waits for the slowest.
static int ptm_build_segments(rdis_hip_plan* L, int threads, int K) {
    rdis_hip_ctx* c = L->prob->ctx;
    const int nw = threads / 64;
    const int* li = L->h_lds_ints.data();
    const int* ls_ptr = li + L->off_ls_ptr;
    const int* ls_ncb = li + L->off_ls_ncb;
    const int* pm_ch0 = li + L->off_pm_ch0;
    const int* cptr = li + L->off_pm_cptr;
    const size_t nwg = (size_t)L->ncomp * (size_t)K;
    std::vector<long long> off(nwg, 0);
    ivec rows;   // per workgroup: its number of rows R, 0, 0, 0, then three planes of R ints
    std::vector<ivec> share((size_t)nw);
    const size_t r_lds = L->h_rest.size() - (size_t)L->rest_lds, r_ptm = r_lds - (size_t)L->rest_ptm;
    for (size_t ri = r_ptm; ri < r_lds; ++ri) {
        const int cc = L->h_rest[ri];
        const int ncb = ls_ncb[cc], ns = ls_ptr[cc + 1] - ls_ptr[cc], npb = (ns - PTM_CS * ncb) / 3, npc = (npb + 63) / 64;
        const int* cp = cptr + pm_ch0[cc];
        const bool local = L->ptm_local && L->ptm_local_comp == cc;   // (a workgroup's chunks: consecutive ones instead of every K-th)
        for (int rk = 0; rk < K; ++rk) {
            const int ch0 = local ? L->h_wg_chunk0[(size_t)rk] : rk, chend = local ? L->h_wg_chunk0[(size_t)rk + 1] : npc, chstep = local ? 1 : K;
            long long units = 0;
            for (int ch = ch0; ch < chend; ch += chstep) units += ((cp[ch + 1] - cp[ch]) / 64 + PTM_BLK - 1) / PTM_BLK;
            size_t depth = 0;
            long long u = 0;   // blocks dealt out so far
            int ch = ch0, done = 0;   // the chunk at hand and its blocks already dealt out
            for (int w = 0; w < nw; ++w) {
                ivec& sh = share[(size_t)w];
                sh.clear();
                const long long end = units * (w + 1) / nw;
                while (u < end) {
                    const int nb = ((cp[ch + 1] - cp[ch]) / 64 + PTM_BLK - 1) / PTM_BLK;
                    if (done >= nb) { ch += chstep; done = 0; continue; }
                    const int take = (int)std::min<long long>(nb - done, end - u);
                    const int e0 = cp[ch] + 64 * PTM_BLK * done, e1 = std::min(cp[ch + 1], e0 + 64 * PTM_BLK * take);
                    sh.push_back(ch); sh.push_back(e0); sh.push_back(e1); sh.push_back(0);
                    done += take; u += take;
                }
                depth = std::max(depth, sh.size() / 4);
            }
            // (two rows of nothing behind every wave's last: the loop asks for its rows two ahead)
            // (and at least one row of work: a rank with no chunks still has the three rows a wave reads up front)
            const size_t base = rows.size(), R = (std::max<size_t>(depth, 1) + 2) * (size_t)nw;
            off[(size_t)cc * K + rk] = (long long)base;
            rows.resize(base + 4 + 3 * R, 0);
            rows[base] = (int)R;
            for (int w = 0; w < nw; ++w)
                for (size_t k = 0; k < share[(size_t)w].size() / 4; ++k)
                    for (size_t q = 0; q < 3; ++q) rows[base + 4 + q * R + k * (size_t)nw + (size_t)w] = share[(size_t)w][4 * k + q];
        }
    }
    int rc = plan_alloc(L, L->pm_segs, std::max<size_t>(rows.size(), 4) * sizeof(int));
    if (!rc) rc = plan_alloc(L, L->pm_sg_off, nwg * sizeof(long long));
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(L->pm_segs.p, rows.data(), rows.size() * sizeof(int), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(L->pm_sg_off.p, off.data(), nwg * sizeof(long long), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}
// the slots a gradient round evaluates and stages (solver_ptm.hpp: rs): two -- a whole block of slots -- where the LDS holds 2 x threads
// staging rows beside the cameras, else one
static int ptm_round_slots_for(const rdis_hip_plan* L, int threads) {
    if (L->ptm_round_slots == 1 || PTM_BLK != 2 || threads > PTM_PAIR_MAX_THREADS) return 1;
    // (workgroups of 256 lanes stand two to a compute unit: both must keep their place)
    return (size_t)(threads <= 256 ? 2 : 1) * ptm_bytes_for(L->ptm_ncb_cap, threads, 2) <= (size_t)L->prob->ctx->lds_limit ? 2 : 1;
}
int ptm_build_rounds(rdis_hip_plan* L, int threads, int K) {
    rdis_hip_ctx* c = L->prob->ctx;
    const int rs = ptm_round_slots_for(L, threads);
    if (L->rounds_threads == threads && L->rounds_K == K && L->rounds_slots == rs) return 0;
    // (the key is set when everything below has succeeded: a caller that frees memory and solves again after ENOMEM
    // must find the tables either complete or absent, never "already built" with buffers missing)
    L->rounds_threads = L->rounds_K = 0;
    { const int rc = ptm_build_segments(L, threads, K); if (rc) return rc; }
    if (L->ptm_rot_mode == ROT_CAMFIX) { L->rounds_threads = threads; L->rounds_K = K; L->rounds_slots = rs; return 0; }
    const int nw = threads / 64;
    const int* li = L->h_lds_ints.data();
    const int* ls_ptr = li + L->off_ls_ptr;
    const int* ls_ncb = li + L->off_ls_ncb;
    const unsigned* fidx = reinterpret_cast<const unsigned*>(li + L->off_ls_fidx);
    const int* pm_ch0 = li + L->off_pm_ch0;
    const int* cptr = li + L->off_pm_cptr;
    const int* jg = li + L->off_pm_jg;
    const size_t nwg = (size_t)L->ncomp * (size_t)K;
    std::vector<long long> off(nwg, 0);
    ivec nr(nwg, 0);
    std::vector<unsigned short> tab, grow((size_t)L->pm_entries + 64 * PTM_BLK, 0);
    std::vector<ivec> steps((size_t)nw), step_slots((size_t)nw);
    ivec seg, pos;
    const size_t r_lds = L->h_rest.size() - (size_t)L->rest_lds, r_ptm = r_lds - (size_t)L->rest_ptm;
    for (size_t ri = r_ptm; ri < r_lds; ++ri) {
        const int cc = L->h_rest[ri];
        const int ncb_all = ls_ncb[cc], ns = ls_ptr[cc + 1] - ls_ptr[cc], npb = (ns - PTM_CS * ncb_all) / 3, npc = (npb + 63) / 64;
        const int* cp = cptr + pm_ch0[cc];
        const bool local = L->ptm_local && L->ptm_local_comp == cc;
        for (int rk = 0; rk < K; ++rk) {
            // (a local group: the workgroup's own cameras, under its own numbers; its chunks consecutive, wave w every eighth from the w-th on)
            const int ncb = local ? L->h_lc[(size_t)L->h_lc_off[(size_t)rk]] : ncb_all;
            const size_t stride = (size_t)ptm_round_stride(ncb);
            auto cam_of = [&](int e, int j) { return local ? (int)L->h_pm_lcam[(size_t)e] : (int)(fidx[j] & 0xFFFu); };   // (an entry's camera as the stream names it)
            size_t nrounds = 0;
            for (int w = 0; w < nw; ++w) {
                ivec& st = steps[(size_t)w];
                ivec& sn = step_slots[(size_t)w];
                st.clear(); sn.clear();
                const int ch0 = local ? L->h_wg_chunk0[(size_t)rk] + w : rk + K * w, chend = local ? L->h_wg_chunk0[(size_t)rk + 1] : npc, chstep = local ? nw : K * nw;
                // (a step: one slot of a chunk, or -- two slots a round -- a block of up to two of ONE chunk)
                for (int ch = ch0; ch < chend; ch += chstep)
                    for (int e = cp[ch]; e < cp[ch + 1]; e += 64 * rs) { st.push_back(e); sn.push_back(std::min(rs, (cp[ch + 1] - e) / 64)); }
                nrounds = std::max(nrounds, st.size());
            }
            const size_t base = tab.size();
            off[(size_t)cc * K + rk] = (long long)base;
            nr[(size_t)cc * K + rk] = (int)nrounds;
            tab.resize(base + nrounds * stride, 0);
            for (size_t rr = 0; rr < nrounds; ++rr) {
                unsigned short* rec = tab.data() + base + rr * stride;
                seg.assign((size_t)ncb + 1, 0);
                for (int w = 0; w < nw; ++w) {
                    if (rr >= steps[(size_t)w].size()) continue;
                    const int e = steps[(size_t)w][rr];
                    for (int l = 0; l < 64 * step_slots[(size_t)w][rr]; ++l) { const int j = jg[e + l]; if (j >= 0) ++seg[(size_t)cam_of(e + l, j) + 1]; }
                }
                for (int k = 0; k < ncb; ++k) seg[(size_t)k + 1] += seg[(size_t)k];
                for (int k = 0; k <= ncb; ++k) rec[k] = (unsigned short)seg[(size_t)k];
                pos.assign(seg.begin(), seg.end() - 1);
                // (within a camera: the round's first slots by wave and lane, then its second slots -- with every chunk an even number of
                // slots long that is the order of one slot a round)
                for (int sl = 0; sl < rs; ++sl)
                    for (int w = 0; w < nw; ++w) {
                        if (rr >= steps[(size_t)w].size() || sl >= step_slots[(size_t)w][rr]) continue;
                        const int e = steps[(size_t)w][rr] + 64 * sl;
                        for (int l = 0; l < 64; ++l) {
                            const int j = jg[e + l];
                            if (j >= 0) grow[(size_t)(e + l)] = (unsigned short)pos[(size_t)cam_of(e + l, j)]++;
                        }
                    }
            }
        }
    }
    if (std::getenv("RDIS_HIP_ROUND_STATS") && nwg > 0) {   // (tuning: how uneven are a round's camera segments?)
        const int cc = L->h_rest[r_ptm];
        const int ncb = ls_ncb[cc];
        const size_t stride = (size_t)ptm_round_stride(ncb);
        const unsigned short* t0 = tab.data() + off[(size_t)cc * K];
        long long sum_max = 0, worst = 0, rows = 0, active = 0;
        const int nrd = nr[(size_t)cc * K];
        for (int rr = 0; rr < nrd; ++rr) {
            const unsigned short* rec = t0 + (size_t)rr * stride;
            int mx = 0;
            for (int k = 0; k < ncb; ++k) { const int len = rec[k + 1] - rec[k]; mx = std::max(mx, len); active += len > 0; }
            sum_max += mx; worst = std::max<long long>(worst, mx); rows += rec[ncb];
        }
        std::fprintf(stderr, "ptm rounds (threads %d, K %d): first workgroup %d rounds, %d cameras; rows per round %.1f, active cameras per round %.1f, longest segment: mean %.1f, worst %lld\n",
                     threads, K, nrd, ncb, (double)rows / std::max(nrd, 1), (double)active / std::max(nrd, 1), (double)sum_max / std::max(nrd, 1), worst);
    }
    int rc = plan_alloc(L, L->pm_rounds, std::max<size_t>(tab.size(), 2) * sizeof(unsigned short));
    if (!rc) rc = plan_alloc(L, L->pm_grow, grow.size() * sizeof(unsigned short));
    if (!rc) rc = plan_alloc(L, L->pm_rd_off, nwg * sizeof(long long));
    if (!rc) rc = plan_alloc(L, L->pm_rd_n, nwg * sizeof(int));
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(L->pm_rounds.p, tab.data(), tab.size() * sizeof(unsigned short), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(L->pm_grow.p, grow.data(), grow.size() * sizeof(unsigned short), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(L->pm_rd_off.p, off.data(), nwg * sizeof(long long), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(L->pm_rd_n.p, nr.data(), nwg * sizeof(int), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));   // (the host tables go out of scope; a launch on another stream follows)
    L->rounds_threads = threads; L->rounds_K = K; L->rounds_slots = rs;
    return 0;
}
int launch_ptm(rdis_hip_plan* L, hipStream_t stream, int threads, int first, int grid, int maxiters, double ftol) {
    rdis_hip_ctx* c = L->prob->ctx;
    int rc = ptm_build_rounds(L, threads, 1);
    if (rc) return rc;
    ProblemView P = L->prob->view();
    PlanView V = L->view();
    V.order += first;
    HIPCHK(c, ptm_launch(L->ptm_rot_mode, threads, grid, ptm_bytes_for(L->ptm_ncb_cap, threads, L->rounds_slots), stream, P, V, maxiters, ftol, L->ptm_ncb_cap));
    return 0;
}
// K workgroups per component (cgd_ptmg_kernel): how many groups of K fit the device, and the launch
int ptmg_resident_workgroups(rdis_hip_plan* L, int threads, int* out, bool wide = false) {
    rdis_hip_ctx* c = L->prob->ctx;
    const size_t dyn = ptm_bytes_for(L->ptm_ncb_cap, threads, ptm_round_slots_for(L, threads));
    const void* fn = ptmg_kernel_fn(L->ptm_rot_mode, threads, wide);
    if (dyn > 48 * 1024) HIPCHK(c, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
    int per_cu = 0;
    HIPCHK(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, threads, dyn));
    *out = per_cu * c->num_cus;
    return 0;
}
int launch_ptm_groups(rdis_hip_plan* L, hipStream_t stream, int threads, int first, int ngroups, int K, int maxiters, double ftol, bool wide = false) {
    const bool local = wide && L->ptm_local;
    rdis_hip_ctx* c = L->prob->ctx;
    int rc = ptm_build_rounds(L, threads, K);
    if (rc) return rc;
    ProblemView P = L->prob->view();
    PlanView V = L->view();
    V.order += first;
    const size_t dyn = ptm_bytes_for(L->ptm_ncb_cap, threads, L->rounds_slots);
    int ncc = L->ptm_ncb_cap;
    // (a wide group's workgroups and waves outnumber a small state's entries: a CoopState each)
    const size_t st_bytes = wide ? sizeof(CoopState) : sizeof(SmallCoopState);
    const size_t abort_off = wide ? offsetof(CoopState, abort_flag) : offsetof(SmallCoopState, abort_flag);
    if (L->ptm_state.bytes < (size_t)ngroups * st_bytes) {
        rc = plan_alloc(L, L->ptm_state, (size_t)ngroups * st_bytes);
        if (rc) return rc;
    }
    const size_t xch_bytes = (size_t)ngroups * (2 * (size_t)K + 2) * PTM_CS * (size_t)ncc * sizeof(double);
    if (L->ptm_xch.bytes < xch_bytes) {
        rc = plan_alloc(L, L->ptm_xch, xch_bytes);
        if (rc) return rc;
    }
    // arm every granule, clear the abort words
    HIPCHK(c, hipMemsetAsync(L->ptm_state.p, 0xFF, (size_t)ngroups * st_bytes, stream));
    HIPCHK(c, hipMemset2DAsync((char*)L->ptm_state.p + abort_off, st_bytes, 0, 64, (size_t)ngroups, stream));
    PtmGroupArgs A{L->ptm_state.p, L->ptm_xch.as<double>(), K, ngroups, wide ? L->coop_poll_delay : std::min(4, L->coop_poll_delay),
                   nullptr, nullptr, nullptr, nullptr, nullptr};
    if (local) {
        A.lc_off = L->lc_off_dev.as<long long>(); A.lc = L->lc_dev.as<int>(); A.cr_ptr = L->cr_ptr_dev.as<int>(); A.cr = L->cr_dev.as<int>();
        A.tot = L->ptm_tot.as<double>();
    }
    int mi = maxiters;
    double ft = ftol;
    void* args[] = {&P, &V, &A, &mi, &ft, &ncc};
    const int grid = wide ? K * ngroups : 8 * K * ((ngroups + 7) / 8);
    const void* fn = ptmg_kernel_fn(L->ptm_rot_mode, threads, wide, local);
    if (local && dyn > 48 * 1024) HIPCHK(c, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
    HIPCHK(c, hipLaunchCooperativeKernel(fn, dim3(grid), dim3(threads), args, dyn, stream));
    return 0;
}
int launch_lds_stale(rdis_hip_plan* L, hipStream_t stream, int threads, int first, int grid, int maxiters, double ftol) {
    rdis_hip_ctx* c = L->prob->ctx;
    ProblemView P = L->prob->view();
    PlanView V = L->view();
    V.order += first;
    const size_t dyn = lds_bytes_for(L->lds_ns_cap, L->lds_ncb_cap, L->lds_chunk_cap);
    const int nsc = L->lds_ns_cap, ncc = L->lds_ncb_cap, chc = L->lds_chunk_cap;
#define RDIS_LDS_LAUNCH(T)                                                                                              \
    do {                                                                                                                \
        if (dyn > 48 * 1024)                                                                                            \
            HIPCHK(c, hipFuncSetAttribute((const void*)cgd_lds_kernel<T, ROT_PER_FACTOR, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn)); \
        cgd_lds_kernel<T, ROT_PER_FACTOR, true><<<grid, T, dyn, stream>>>(P, V, maxiters, ftol, nsc, ncc, chc);                \
    } while (0)
    switch (threads) {
        case 64: RDIS_LDS_LAUNCH(64); break;
        case 128: RDIS_LDS_LAUNCH(128); break;
        case 256: RDIS_LDS_LAUNCH(256); break;
        case 512: RDIS_LDS_LAUNCH(512); break;
        case 768: RDIS_LDS_LAUNCH(768); break;
        default: RDIS_LDS_LAUNCH(1024); break;
    }
#undef RDIS_LDS_LAUNCH
    HIPCHK(c, hipGetLastError());
    return 0;
}
int launch_lds(rdis_hip_plan* L, hipStream_t stream, int threads, int first, int grid, int maxiters, double ftol) {
    if (L->emulate_stale && !L->batch_reference_rounding()) return launch_lds_stale(L, stream, threads, first, grid, maxiters, ftol);
    if (L->batch_reference_rounding()) {
        rdis_hip_ctx* c = L->prob->ctx;
        ProblemView P = L->prob->view();
        PlanView V = L->view();
        V.order += first;
        HIPCHK(c, refround_launch_lds(L->lds_rot_mode, L->emulate_stale ? 1 : 0, threads, grid, lds_bytes_for(L->lds_ns_cap, L->lds_ncb_cap, L->lds_chunk_cap), stream, &P, &V,
                                      maxiters, ftol, L->lds_ns_cap, L->lds_ncb_cap, L->lds_chunk_cap));
        return 0;
    }
    switch (L->lds_rot_mode) {
        case ROT_CAMFIX: return launch_lds_rot<ROT_CAMFIX>(L, stream, threads, first, grid, maxiters, ftol);
        case ROT_RECORDS: return launch_lds_rot<ROT_RECORDS>(L, stream, threads, first, grid, maxiters, ftol);
        default: return launch_lds_rot<ROT_PER_FACTOR>(L, stream, threads, first, grid, maxiters, ftol);
    }
}
}  // namespace

extern "C" int rdis_hip_plan_solve(rdis_hip_plan* L, int32_t maxiters, double ftol) {
    if (!L || maxiters <= 0) return RDIS_HIP_EINVAL;
    rdis_hip_problem* p = L->prob;
    rdis_hip_ctx* c = p->ctx;
    USE_DEVICE(c);
    // (a plan made before rdis_hip_nlp_set_exponential marked one of its factors: the reference's gradient asserts the flag off,
    // src/NonlinearProductFactor.cpp:110 -- refused here as at plan_create, not solved with values and slopes of two functions)
    if (!p->h_useexp.empty())
        for (int f : L->h_fac_id)
            if (p->h_useexp[(size_t)f])
                return fail(c, RDIS_HIP_EINVAL, "plan_solve: factor " + std::to_string(f) + " is exponential (set after the plan was created); "
                                                "the reference's gradient asserts it is not (NonlinearProductFactor.cpp:110)");
    if (!L->have_start) { int rc = rdis_hip_plan_set_start(L, nullptr); if (rc) return rc; }
    L->last_launches = 0;
    L->timed = false;
    if (L->ncomp == 0) return 0;

    // a few very large components go to the cooperative multi-workgroup solver, one
    // launch each; everything else is one batched launch, one workgroup per component
    // (also when another plan of the problem has moved the exchange-state buffer this plan's
    // cooperative group tables point into)
    if (L->partition_dirty || (!L->coop.empty() && L->coop_state_gen != p->coop_state_gen)) { int rc = prepare_partition(L); if (rc) return rc; }
    if (L->emulate_stale) {
        // implemented where config 5's components run: the LDS-resident batch solver (and per-factor rotations) -- and, under the
        // parity option, in the cooperative solver's plain layout
        if ((size_t)L->rest_lds != L->h_rest.size() || (!L->coop.empty() && L->factor_rounding != 1) || !L->stream.empty() ||
            (L->rest_lds > 0 && L->lds_rot_mode != ROT_PER_FACTOR))
            return fail(c, RDIS_HIP_EINVAL, "emulate_stale_cache: every component of the plan must run on the LDS-resident batch solver "
                                            "(bundle adjustment, variables fitting a compute unit's LDS; cooperative groups only with factor_rounding = 1)");
        if (!L->st_ev.p) {
            int rc = plan_alloc(L, L->st_ev, (size_t)std::max<int64_t>(L->nfac, 1) * sizeof(int));
            if (!rc) rc = plan_alloc(L, L->st_val, (size_t)std::max<int64_t>(L->nfac, 1) * sizeof(double));
            if (rc) return rc;
        }
    }
    if (L->factor_rounding == 1) {
        // instantiated for the solvers BASELINE's configs 3 - 5 reach: the cooperative ones and the LDS-resident batch solver
        if ((size_t)L->rest_lds != L->h_rest.size() || !L->stream.empty() || p->kind != KIND_BA)
            return fail(c, RDIS_HIP_EINVAL, "factor_rounding = 1: every component of the plan must run on a cooperative solver or on the "
                                            "LDS-resident batch solver (bundle adjustment; no streaming, tiny-component or plain launches)");
        if (!L->seq_val.p) {
            int rc = plan_alloc(L, L->seq_val, (size_t)std::max<int64_t>(L->nfac, 1) * sizeof(double));
            if (!rc) rc = plan_alloc(L, L->seq_ab, 2 * (size_t)std::max<int64_t>(L->nfree, 1) * sizeof(double));
            if (rc) return rc;
        }
    }
    PlanView V = L->view();
    HIPCHK(c, hipEventRecord(p->ev0, c->stream));
    // The batched launch is independent of the cooperative ones (disjoint components): it goes to a
    // second stream and fills the CUs the few workgroups of a cooperative solve leave idle.  The
    // cooperative kernels are enqueued first, so their workgroups are resident before the batch arrives.
    // (not next to a streaming solve: that grid takes every CU and waits for all of its workgroups)
    // (and only when the cooperative launches leave half of the device free: their workgroups must all be
    // resident and take a compute unit each; were the batch to arrive first on a full device, they would
    // wait for it -- serialised, and spinning meanwhile -- instead of running beside it)
    int coop_wg_max = 0;
    for (const CoopLaunch& cl : L->coop_launches) coop_wg_max = std::max(coop_wg_max, cl.total_wg);
    const bool overlap = L->overlap_batch && !L->h_rest.empty() && !L->coop.empty() && L->stream.empty() &&
                         2 * coop_wg_max <= c->num_cus;
    hipStream_t bs = c->stream;
    if (overlap) {
        if (!c->aux) {
            HIPCHK(c, hipStreamCreateWithFlags(&c->aux, hipStreamNonBlocking));
            HIPCHK(c, hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
            HIPCHK(c, hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
        }
        bs = c->aux;
        HIPCHK(c, hipEventRecord(c->ev_fork, c->stream));
        HIPCHK(c, hipStreamWaitEvent(bs, c->ev_fork, 0));
    }
    for (size_t l = 0; l < L->coop_launches.size(); ++l) {
        const CoopLaunch& cl = L->coop_launches[l];
        const ProblemView PVc = p->view();
        int rc;
        if (L->coop_reference_rounding())
            rc = L->pipelined()
                     ? refround_launch_pipe(c->stream, p->kind, &PVc, &V, &L->h_coop_groups[l][0], cl.groups.p, cl.wg_group.as<int>(), cl.count, cl.total_wg, maxiters, ftol)
                     : refround_launch_coop(c->stream, p->kind, &PVc, &V, &L->h_coop_groups[l][0], cl.groups.p, cl.wg_group.as<int>(), cl.count, cl.total_wg,
                                            L->coop_threads, maxiters, ftol);
        else
            rc = L->pipelined()
                     ? launch_pipe(c->stream, p->kind, PVc, V, L->h_coop_groups[l][0], cl.groups.as<CoopGroup>(), cl.wg_group.as<int>(),
                                   cl.count, cl.total_wg, maxiters, ftol)
                     : launch_coop(c->stream, p->kind, PVc, V, L->h_coop_groups[l][0], cl.groups.as<CoopGroup>(), cl.wg_group.as<int>(),
                                   cl.count, cl.total_wg, L->coop_threads, maxiters, ftol);
        if (rc != 0) return fail(c, RDIS_HIP_EDEVICE, std::string("cooperative solver launch: ") + hipGetErrorString((hipError_t)rc));
        ++L->last_launches;
    }
    for (size_t i = 0; i < L->stream.size(); ++i) {
        const StreamItem& it = L->stream[i];
        StreamArgs sa{p->coop_timing.as<long long>(), p->coop_state.as<CoopState>(), it.long_vars.as<int>(), it.nlong,
                      it.comp, L->coop_poll_delay};
        int rc = launch_stream(c->stream, p->kind, p->view(), V, sa, it.nwg, maxiters, ftol);
        if (rc != 0) return fail(c, RDIS_HIP_EDEVICE, std::string("streaming grid solver launch: ") + hipGetErrorString((hipError_t)rc));
        ++L->last_launches;
    }
    const int rest = (int)L->h_rest.size() - L->rest_tiny - L->rest_lds - L->rest_ptm;   // components of the plain batch solver
    // (the LDS-resident solver forms the records it reads itself)
    if (L->rest_rot_mode != ROT_PER_FACTOR && (L->rest_tiny > 0 || rest > 0)) {   // (cameras the batch reads are not free in the launches above)
        camera_rotations_kernel<<<(int)((p->ncam_blocks + 255) / 256), 256, 0, bs>>>(
            p->x.as<double>(), p->cam_blocks.as<int>(), (int)p->ncam_blocks, p->xrot.as<double>());
        HIPCHK(c, hipGetLastError());
    }
    if (L->rest_tiny > 0) {
        ProblemView PV = p->view();
        PV.xrot = p->xrot.as<double>();
        PV.rot_mode = L->rest_rot_mode == ROT_CAMFIX ? ROT_CAMFIX : ROT_PER_FACTOR;   // (the quad solver has no refresh)
        // persistent groups: as many blocks as are resident, every group draws components until the list is empty
        HIPCHK(c, hipMemsetAsync(L->queue.p, 0, sizeof(int), bs));
        if (L->tiny_group == 4) {
            const int gpb = QUAD_THREADS / 4;
            int grid = std::min((L->rest_tiny + gpb - 1) / gpb, std::max(1, L->group_blocks4));
            if (L->tiny_max_blocks > 0) grid = std::min(grid, L->tiny_max_blocks);
            cgd_group_kernel<4, QUAD_THREADS><<<grid, QUAD_THREADS, 0, bs>>>(
                PV, V, L->rest_order.as<int>(), L->rest_tiny, L->queue.as<int>(), maxiters, ftol);
        } else {
            int grid = std::min((L->rest_tiny + 3) / 4, std::max(1, L->group_blocks16));
            if (L->tiny_max_blocks > 0) grid = std::min(grid, L->tiny_max_blocks);
            cgd_group_kernel<16, 64><<<grid, 64, 0, bs>>>(
                PV, V, L->rest_order.as<int>(), L->rest_tiny, L->queue.as<int>(), maxiters, ftol);
        }
        HIPCHK(c, hipGetLastError());
        ++L->last_launches;
    }
    if (L->rest_lds > 0) {
        const int64_t mf = L->lds_max_factors;
        int threads = L->lds_threads ? L->lds_threads : L->block_threads;
        // A lane per factor up to 512, then 768 lanes (three waves per SIMD) -- unless the launch has more
        // components than compute units: then 256 lanes, two workgroups per compute unit, so that one
        // component's arithmetic fills the unit while the other's control step (one lane) or barrier runs
        // (1000 x 2048 factors: 11.9 against 13.7 ms; 125 of them, a unit each: 3.5 against 4.7 ms)
        if (threads == 0) threads = mf <= 64 ? 64 : mf <= 128 ? 128 : mf <= 256 ? 256 : L->rest_lds > c->num_cus ? 256 : mf <= 512 ? 512 : 768;
        int rc = launch_lds(L, bs, threads, L->rest_tiny + rest + L->rest_ptm, L->rest_lds, maxiters, ftol);
        if (rc) return rc;
        ++L->last_launches;
    }
    if (L->rest_ptm > 0) {
        // 768 lanes, three waves per SIMD (1000 / 500 components of ladybug's size: 187 / 99 ms, 196 / 109 with 256 lanes and two
        // workgroups per compute unit).  Fewer components than resident workgroups: K workgroups share a component
        // (cooperative launch, every workgroup of it resident), of 256 lanes -- two per compute unit, the finer grain
        // loses less to whole wave-chunks -- as long as a workgroup keeps some twenty wave-chunks of points per trial point
        // (below that the exchange costs what the split saves: 125 components of 2048 points 9.6 ms alone, 9.7 as pairs;
        // of 7776 points 35.8 ms alone, 27.7 as groups of four)
        int threads = L->ptm_threads ? L->ptm_threads : L->ptm_wide_wanted ? PTM_WIDE_THREADS : 768;
        int K = 1;
        if (L->ptm_group != 1 && !overlap && L->coop.empty() && L->stream.empty()) {
            // groups of K workgroups: of 512 lanes (a workgroup per compute unit) or of 256 (two) -- whichever brings more
            // lanes to a component, at equal lanes the larger workgroup (125 components of ladybug's size: pairs of 512
            // lanes 16.0 ms, fours of 256 17.0, one workgroup of 768 each 24.2; round 4)
            const int slots8 = 8 * ((L->rest_ptm + 7) / 8);   // (groups are placed eight at a time, one per XCD)
            const int useful = (int)std::max<int64_t>(1, (L->ptm_min_points + 63) / 64 / 24);
            int best_lanes = 0, best_threads = 0, best_K = 1;
            const int cands[2] = {L->ptm_threads ? L->ptm_threads : 512, 256};
            for (int ci = 0; ci < (L->ptm_threads ? 1 : 2); ++ci) {
                const int gt = cands[ci];
                int cap = 0;
                int rc = ptmg_resident_workgroups(L, gt, &cap);
                if (rc) return rc;
                const int fit = std::min(std::min(PTM_MAX_GROUP, cap / slots8), SMALL_COOP_ENTRIES / (gt / 64));
                const int Kc = L->ptm_group > 1 ? std::min(L->ptm_group, fit) : std::min(fit, useful);
                if (Kc >= 2 && Kc * gt > best_lanes) { best_lanes = Kc * gt; best_threads = gt; best_K = Kc; }
            }
            // (a group must bring more lanes to a component than the one workgroup it replaces: 250 components of
            // ladybug's size 50.4 ms a workgroup of 768 lanes each, 56.0 as pairs of 256)
            if (best_K >= 2 && (L->ptm_group > 1 || best_lanes > threads)) { K = best_K; threads = best_threads; }
        }
        // A few components and a device: wide groups -- as many workgroups of 512 lanes a component as are resident and have some
        // twenty wave-chunks of points each (one component of 8e6 factors: 256 workgroups, 15 chunks a wave)
        bool wide = false;
        if (L->ptm_local) {   // (its tables are made for this group size)
            K = L->ptm_local_K; threads = PTM_WIDE_THREADS; wide = true;
            if (overlap || !L->coop.empty() || !L->stream.empty())
                return fail(c, RDIS_HIP_EINVAL, "a wide group with local camera numbering takes the device to itself: no cooperative or grid launch beside it");
        } else if (L->ptm_group != 1 && !overlap && L->coop.empty() && L->stream.empty() && L->rest_ptm <= 8 &&
            (L->ptm_threads == 0 || L->ptm_threads == PTM_WIDE_THREADS)) {
            int cap = 0;
            int rc = ptmg_resident_workgroups(L, PTM_WIDE_THREADS, &cap, true);
            if (rc) return rc;
            const int useful = (int)std::max<int64_t>(1, (L->ptm_min_points + 63) / 64 / 24);
            const int fit = std::min(std::min(PTM_WIDE_MAX_GROUP, cap / L->rest_ptm), COOP_MAX_WG * COOP_MAX_WAVES / (PTM_WIDE_THREADS / 64));
            const int Kw = L->ptm_group > 1 ? std::min(L->ptm_group, fit) : std::min(fit, useful);
            if (Kw > PTM_MAX_GROUP && Kw * PTM_WIDE_THREADS > K * threads) { K = Kw; threads = PTM_WIDE_THREADS; wide = true; }
        }
        L->ptm_last_group = std::max(K, 1);
        L->ptm_last_threads = threads;
        L->ptm_wide_last = wide;
        int rc = K >= 2 ? launch_ptm_groups(L, bs, threads, L->rest_tiny + rest, L->rest_ptm, K, maxiters, ftol, wide)
                        : launch_ptm(L, bs, threads, L->rest_tiny + rest, L->rest_ptm, maxiters, ftol);
        if (rc) return rc;
        ++L->last_launches;
    }
    if (rest > 0) {
        int64_t mf = 0;
        for (size_t i = (size_t)L->rest_tiny; i < L->h_rest.size() - (size_t)L->rest_lds - (size_t)L->rest_ptm; ++i) {
            const int cc = L->h_rest[i];
            mf = std::max<int64_t>(mf, std::max<int64_t>(L->h_fac_ptr[(size_t)cc + 1] - L->h_fac_ptr[(size_t)cc],
                                                         (L->h_free_ptr[(size_t)cc + 1] - L->h_free_ptr[(size_t)cc]) / 4));
        }
        int threads = L->block_threads;
        // More than 512 factors: 768 lanes = three waves per SIMD at 168 registers (a few spills) beat
        // two waves at 250 and four at 128 (heavy spills): +20 % / +40 % throughput on large components,
        // and 5 % on the 361..906-factor camera components of ladybug.
        if (threads == 0) threads = mf <= 64 ? 64 : mf <= 128 ? 128 : mf <= 256 ? 256 : mf <= 512 ? 512 : 768;
        int rc = p->kind == KIND_BA ? launch_wg<KIND_BA>(L, bs, threads, L->rest_tiny, rest, maxiters, ftol)
                                    : launch_wg<KIND_NLP>(L, bs, threads, L->rest_tiny, rest, maxiters, ftol);
        if (rc) return rc;
        ++L->last_launches;
    }
    if (overlap) {
        HIPCHK(c, hipEventRecord(c->ev_join, bs));
        HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_join, 0));
    }
    HIPCHK(c, hipEventRecord(p->ev1, c->stream));
    L->timed = true;
    p->last_timed_plan = L;
    objective_sum_kernel<<<1, 256, 0, c->stream>>>((int)L->ncomp, V.fret, L->objective.as<double>());
    HIPCHK(c, hipGetLastError());
    return 0;
}

extern "C" int rdis_hip_plan_fetch(rdis_hip_plan* L, double* x_out, double* fret, double* delta, int32_t* iters,
                                   int32_t* status, int64_t* nfeval, int64_t* ngeval) {
    if (!L) return RDIS_HIP_EINVAL;
    rdis_hip_ctx* c = L->prob->ctx;
    const size_t nc = (size_t)L->ncomp, nf = (size_t)L->nfree;
    USE_DEVICE(c);
    if (L->out_bytes == 0) return 0;
    // one D2H copy of the whole results block (x first: skipped when not wanted) -- or, for a long x, two: x straight
    // into the caller's array (no pass through the staging block and no host copy of it: 190 MB for the 1000-component
    // decomposition of the bench's strong-scaling block), the per-component results into the staging block
    const bool direct_x = x_out != nullptr && nf * 8 >= DIRECT_X_BYTES;
    const size_t skip = (x_out && !direct_x) ? 0 : nf * 8;
    if (direct_x) HIPCHK(c, hipMemcpyAsync(x_out, L->outbuf.p, nf * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(L->h_out.data() + skip, static_cast<char*>(L->outbuf.p) + skip, L->out_bytes - skip,
                             hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const char* h = L->h_out.data();
    if (x_out && !direct_x) std::memcpy(x_out, h, nf * 8);
    if (fret) std::memcpy(fret, h + nf * 8, nc * 8);
    if (delta) std::memcpy(delta, h + (nf + nc) * 8, nc * 8);
    const char* i64 = h + (nf + 2 * nc) * 8;
    if (nfeval) std::memcpy(nfeval, i64, nc * 8);
    if (ngeval) std::memcpy(ngeval, i64 + nc * 8, nc * 8);
    const char* i32 = i64 + 2 * nc * 8;
    if (iters) std::memcpy(iters, i32, nc * 4);
    if (status) std::memcpy(status, i32 + nc * 4, nc * 4);
    return 0;
}

extern "C" int rdis_hip_plan_objective_device(rdis_hip_plan* L, void** dev_ptr) {
    if (!L || !dev_ptr) return RDIS_HIP_EINVAL;
    *dev_ptr = L->objective.p;
    return 0;
}

// =====================================================================================
// the path's one collective: the all-reduce of the top-level objective (reference src/RDISOptimizer.cpp:1491-1494 adds the
// components' values on one host; here the components of a level are sharded over GPUs and the partial sums meet over xGMI)
// =====================================================================================
// RCCL is loaded at the first communicator (dlopen): a single-GPU user never maps it.
#include <dlfcn.h>
#include <rccl/rccl.h>

struct rdis_hip_comm {
    rdis_hip_ctx* ctx = nullptr;
    ncclComm_t comm = nullptr;
    int world = 1, rank = 0;
    DevBuf scratch;            // small host-buffer reductions (rdis_hip_comm_allreduce_f64)
};

namespace {
struct RcclApi {
    void* h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};
RcclApi& rccl() {
    static RcclApi A = [] {
        RcclApi a;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            a.h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (a.h) break;
        }
        if (!a.h) return a;
#define RDIS_RCCL_SYM(field, sym) a.field = reinterpret_cast<decltype(a.field)>(dlsym(a.h, sym))
        RDIS_RCCL_SYM(GetUniqueId, "ncclGetUniqueId"); RDIS_RCCL_SYM(CommInitRank, "ncclCommInitRank"); RDIS_RCCL_SYM(CommInitAll, "ncclCommInitAll");
        RDIS_RCCL_SYM(CommDestroy, "ncclCommDestroy"); RDIS_RCCL_SYM(AllReduce, "ncclAllReduce"); RDIS_RCCL_SYM(GroupStart, "ncclGroupStart");
        RDIS_RCCL_SYM(GroupEnd, "ncclGroupEnd"); RDIS_RCCL_SYM(GetErrorString, "ncclGetErrorString");
#undef RDIS_RCCL_SYM
        a.ok = a.GetUniqueId && a.CommInitRank && a.CommInitAll && a.CommDestroy && a.AllReduce && a.GroupStart && a.GroupEnd;
        return a;
    }();
    return A;
}
int rccl_fail(rdis_hip_ctx* c, ncclResult_t r, const char* what) {
    const char* msg = rccl().GetErrorString ? rccl().GetErrorString(r) : "?";
    if (c) return fail(c, RDIS_HIP_EDEVICE, std::string(what) + ": RCCL: " + msg);
    return RDIS_HIP_EDEVICE;
}
}  // namespace

extern "C" int rdis_hip_comm_unique_id(void* id128) {
    if (!id128) return RDIS_HIP_EINVAL;
    static_assert(sizeof(ncclUniqueId) == RDIS_HIP_COMM_ID_BYTES, "RCCL's unique id is 128 bytes");
    if (!rccl().ok) return RDIS_HIP_EDEVICE;
    ncclUniqueId id;
    if (rccl().GetUniqueId(&id) != ncclSuccess) return RDIS_HIP_EDEVICE;
    std::memcpy(id128, &id, sizeof id);
    return 0;
}

extern "C" int rdis_hip_comm_create(rdis_hip_ctx* c, int32_t world, int32_t rank, const void* id128, rdis_hip_comm** out) {
    if (!c || !out || world < 1 || rank < 0 || rank >= world || !id128) return RDIS_HIP_EINVAL;
    if (!rccl().ok) return fail(c, RDIS_HIP_EDEVICE, "comm_create: librccl.so could not be loaded");
    USE_DEVICE(c);
    std::unique_ptr<rdis_hip_comm> m(new (std::nothrow) rdis_hip_comm);
    if (!m) return fail(c, RDIS_HIP_ENOMEM, "comm_create: host allocation");
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof id);
    const ncclResult_t r = rccl().CommInitRank(&m->comm, world, id, rank);
    if (r != ncclSuccess) return rccl_fail(c, r, "comm_create");
    m->ctx = c; m->world = world; m->rank = rank;
    *out = m.release();
    return 0;
}

extern "C" int rdis_hip_comm_create_all(int32_t n, rdis_hip_ctx* const* ctxs, rdis_hip_comm** comms) {
    if (n < 1 || !ctxs || !comms) return RDIS_HIP_EINVAL;
    for (int i = 0; i < n; ++i) if (!ctxs[i]) return RDIS_HIP_EINVAL;
    rdis_hip_ctx* c0 = ctxs[0];
    if (!rccl().ok) return fail(c0, RDIS_HIP_EDEVICE, "comm_create_all: librccl.so could not be loaded");
    std::vector<int> devs((size_t)n);
    for (int i = 0; i < n; ++i) {
        devs[(size_t)i] = ctxs[i]->phys;
        for (int j = 0; j < i; ++j)
            if (devs[(size_t)j] == devs[(size_t)i]) return fail(c0, RDIS_HIP_EINVAL, "comm_create_all: a device is listed twice (one rank per GPU)");
    }
    std::vector<ncclComm_t> cc((size_t)n, nullptr);
    const ncclResult_t r = rccl().CommInitAll(cc.data(), n, devs.data());
    if (r != ncclSuccess) return rccl_fail(c0, r, "comm_create_all");
    for (int i = 0; i < n; ++i) {
        rdis_hip_comm* m = new (std::nothrow) rdis_hip_comm;
        if (!m) { for (int j = 0; j < i; ++j) { delete comms[j]; comms[j] = nullptr; } for (ncclComm_t q : cc) rccl().CommDestroy(q); return fail(c0, RDIS_HIP_ENOMEM, "comm_create_all: host allocation"); }
        m->ctx = ctxs[i]; m->comm = cc[(size_t)i]; m->world = n; m->rank = i;
        comms[i] = m;
    }
    return 0;
}

extern "C" void rdis_hip_comm_destroy(rdis_hip_comm* m) {
    if (!m) return;
    if (m->ctx) { (void)make_current(m->ctx); (void)hipStreamSynchronize(m->ctx->stream); }
    if (m->comm && rccl().ok) rccl().CommDestroy(m->comm);
    if (m->scratch.p) (void)hipFree(m->scratch.p);
    m->scratch.p = nullptr;
    delete m;
}

extern "C" int rdis_hip_allreduce_objective(rdis_hip_plan* L, rdis_hip_comm* m, double* sum_out) {
    if (!L) return RDIS_HIP_EINVAL;
    rdis_hip_ctx* c = L->prob->ctx;
    USE_DEVICE(c);
    if (m) {
        if (m->ctx != c) return fail(c, RDIS_HIP_EINVAL, "allreduce_objective: the communicator belongs to another context");
        // in place, on the stream the solve ran on: ordered behind objective_sum_kernel, ahead of whatever reads the sum
        const ncclResult_t r = rccl().AllReduce(L->objective.p, L->objective.p, 1, ncclDouble, ncclSum, m->comm, c->stream);
        if (r != ncclSuccess) return rccl_fail(c, r, "allreduce_objective");
    }
    if (sum_out) {
        HIPCHK(c, hipMemcpyAsync(sum_out, L->objective.p, sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    return 0;
}

extern "C" int rdis_hip_allreduce_objective_all(int32_t n, rdis_hip_plan* const* plans, rdis_hip_comm* const* comms, double* sum_out) {
    if (n < 1 || !plans) return RDIS_HIP_EINVAL;
    for (int i = 0; i < n; ++i) if (!plans[i]) return RDIS_HIP_EINVAL;
    rdis_hip_ctx* c0 = plans[0]->prob->ctx;
    if (comms) {
        if (!rccl().ok) return fail(c0, RDIS_HIP_EDEVICE, "allreduce_objective_all: librccl.so could not be loaded");
        for (int i = 0; i < n; ++i) if (!comms[i] || comms[i]->ctx != plans[i]->prob->ctx) return fail(c0, RDIS_HIP_EINVAL, "allreduce_objective_all: communicator i must belong to plan i's context");
        // one thread drives all the ranks: the calls are grouped, or the first would wait for the others
        ncclResult_t r = rccl().GroupStart();
        for (int i = 0; i < n && r == ncclSuccess; ++i) {
            rdis_hip_ctx* c = plans[i]->prob->ctx;
            USE_DEVICE(c);
            r = rccl().AllReduce(plans[i]->objective.p, plans[i]->objective.p, 1, ncclDouble, ncclSum, comms[i]->comm, c->stream);
        }
        const ncclResult_t e = rccl().GroupEnd();
        if (r != ncclSuccess || e != ncclSuccess) return rccl_fail(c0, r != ncclSuccess ? r : e, "allreduce_objective_all");
        if (sum_out) {
            USE_DEVICE(c0);
            HIPCHK(c0, hipMemcpyAsync(sum_out, plans[0]->objective.p, sizeof(double), hipMemcpyDeviceToHost, c0->stream));
            HIPCHK(c0, hipStreamSynchronize(c0->stream));
        }
        return 0;
    }
    // no communicators (the same device listed twice, or no RCCL): the partial sums meet on the host, in plan order
    if (sum_out) {
        double acc = 0.0;
        for (int i = 0; i < n; ++i) {
            rdis_hip_ctx* c = plans[i]->prob->ctx;
            double v = 0.0;
            USE_DEVICE(c);
            HIPCHK(c, hipMemcpyAsync(&v, plans[i]->objective.p, sizeof(double), hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            acc += v;
        }
        *sum_out = acc;
    }
    return 0;
}

extern "C" int rdis_hip_comm_allreduce_f64(rdis_hip_comm* m, double* inout, int32_t n, int32_t op) {
    if (!m || !inout || n < 1 || n > 4096 || (op != 0 && op != 1)) return RDIS_HIP_EINVAL;
    rdis_hip_ctx* c = m->ctx;
    USE_DEVICE(c);
    if (m->scratch.bytes < (size_t)n * sizeof(double)) { if (int rc = dalloc(c, m->scratch, 4096 * sizeof(double))) return rc; }
    HIPCHK(c, hipMemcpyAsync(m->scratch.p, inout, (size_t)n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    const ncclResult_t r = rccl().AllReduce(m->scratch.p, m->scratch.p, (size_t)n, ncclDouble, op == 0 ? ncclSum : ncclMax, m->comm, c->stream);
    if (r != ncclSuccess) return rccl_fail(c, r, "comm_allreduce_f64");
    HIPCHK(c, hipMemcpyAsync(inout, m->scratch.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int rdis_hip_plan_get_info(rdis_hip_plan* L, const char* name, int64_t* value) {
    if (!L || !name || !value) return RDIS_HIP_EINVAL;
    rdis_hip_ctx* c = L->prob->ctx;
    USE_DEVICE(c);
    if (L->partition_dirty) { int rc = prepare_partition(L); if (rc) return rc; }
    const std::string n(name);
    const int64_t rest = (int64_t)L->h_rest.size() - L->rest_tiny - L->rest_lds - L->rest_ptm;
    if (n == "components_cooperative") *value = (int64_t)L->coop.size();
    else if (n == "components_grid_stream") *value = (int64_t)L->stream.size();
    else if (n == "grid_stream_workgroups") *value = L->stream.empty() ? 0 : L->stream.front().nwg;   // (of the first such component)
    else if (n == "components_tiny") *value = L->rest_tiny;
    else if (n == "components_lds") *value = L->rest_lds;
    else if (n == "components_point_major") *value = L->rest_ptm;
    else if (n == "point_major_wide") *value = L->ptm_wide_last ? 1 : 0;
    else if (n == "point_major_local_cameras") *value = (L->ptm_wide_last && L->ptm_local) ? L->ptm_ncb_cap : 0;
    else if (n == "components_plain") *value = rest;
    else if (n == "pipelined") *value = L->pipelined() ? 1 : 0;
    else if (n == "point_major_group") *value = L->ptm_last_group;
    else if (n == "point_major_threads") *value = L->ptm_last_threads;
    else if (n == "point_major_round_slots") *value = L->rounds_slots;
    else return fail(c, RDIS_HIP_EINVAL, "plan_get_info: unknown name '" + n + "'");
    return 0;
}

extern "C" int rdis_hip_plan_device_bytes(rdis_hip_plan* L, int64_t* bytes) {
    if (!L || !bytes) return RDIS_HIP_EINVAL;
    USE_DEVICE(L->prob->ctx);
    // (the solvers' tables are built on demand: counted once they exist -- built here if they do not yet; what a first
    // solve adds for its launch shape, e.g. the streaming solver's round tables, shows in a query after that solve)
    if (L->partition_dirty && !L->transient) { int rc = prepare_partition(L); if (rc) return rc; }
    *bytes = (int64_t)(L->dev_bytes + L->trace.bytes + L->vdump.bytes);
    return 0;
}

extern "C" int rdis_hip_plan_last_kernel_ms(rdis_hip_plan* L, double* ms, int32_t* launches) {
    if (!L || !ms) return RDIS_HIP_EINVAL;
    rdis_hip_problem* p = L->prob;
    rdis_hip_ctx* c = p->ctx;
    *ms = 0.0;
    USE_DEVICE(c);
    if (launches) *launches = L->last_launches;
    if (!L->timed || p->last_timed_plan != L) return 0;  // the events belong to the problem's last solve
    HIPCHK(c, hipEventSynchronize(p->ev1));
    float t = 0.f;
    HIPCHK(c, hipEventElapsedTime(&t, p->ev0, p->ev1));
    *ms = t;
    return 0;
}

extern "C" int rdis_hip_plan_get_trace(rdis_hip_plan* L, int64_t comp, double* rec4, int64_t cap, int64_t* nrec) {
    if (!L || !nrec || comp < 0 || comp >= L->ncomp) return RDIS_HIP_EINVAL;
    rdis_hip_ctx* c = L->prob->ctx;
    USE_DEVICE(c);
    if (L->trace_records <= 0) { *nrec = 0; return 0; }
    int n = 0;
    HIPCHK(c, hipMemcpyAsync(&n, L->view().trace_n + comp, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    *nrec = n;
    const int64_t k = std::min<int64_t>(std::min<int64_t>(n, L->trace_records), cap);
    if (rec4 && k > 0) {
        HIPCHK(c, hipMemcpyAsync(rec4, L->trace.as<double>() + 4ll * L->trace_records * comp, (size_t)k * 4 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    return 0;
}

extern "C" int rdis_hip_lm_optimize(rdis_hip_problem* p, int64_t nfree, const int64_t* free_vid, int64_t nf, const int64_t* fac_id,
                                    double* x_inout, int32_t maxiters, double ftol, int32_t model, double* fret, double* delta, double* info8,
                                    double* hist4, int64_t hist_cap, int64_t* nhist) {
    if (!p) return RDIS_HIP_EINVAL;
    rdis_hip_ctx* c = p->ctx;
    if (p->kind != KIND_BA) return fail(c, RDIS_HIP_EINVAL, "lm_optimize: bundle adjustment problems only");
    if (nfree <= 0 || nf <= 0 || !free_vid || !fac_id || maxiters < 1) return fail(c, RDIS_HIP_EINVAL, "lm_optimize: bad arguments");
    USE_DEVICE(c);
    // cameras occupy the ids below the first point block (BundleAdjustmentFunction.h:88-96)
    int minpt = INT32_MAX;
    for (int64_t f = 0; f < p->F; ++f) minpt = std::min(minpt, p->h_pt[(size_t)f]);
    const int ncams = p->F > 0 ? minpt / 9 : 0;
    for (int64_t f = 0; f < p->F; ++f)
        if (p->h_cam[(size_t)f] % 9 != 0 || p->h_cam[(size_t)f] >= 9 * ncams || (p->h_pt[(size_t)f] - 9 * ncams) % 3 != 0)
            return fail(c, RDIS_HIP_EINVAL, "lm_optimize: variables are not laid out as camera blocks of 9 followed by point blocks of 3");
    if (x_inout) {
        std::vector<int64_t> ids(free_vid, free_vid + nfree);
        int rc = rdis_hip_set_x(p, nfree, ids.data(), x_inout);
        if (rc) return rc;
    }
    LmProblem P{(int)p->N, (int)p->F, ncams, p->x.as<double>(), p->lo.as<double>(), p->hi.as<double>(),
                p->cam.as<int>(), p->pt.as<int>(), p->h_cam.data(), p->h_pt.data(), p->obs.as<double2>()};
    LmOptions o{maxiters, 1e-3, 1e-15, 1e-15, ftol, model & 15, (model >> 4) & 3};
    LmResult r;
    std::string err;
    const int e = device_lm_ba(c->stream, P, nfree, free_vid, nf, fac_id, o, &p->lm_ws, &r, &err);
    if (e < 0) return fail(c, RDIS_HIP_EINVAL, err);
    if (e > 0) return fail(c, RDIS_HIP_EDEVICE, std::string("lm_optimize: ") + hipGetErrorString((hipError_t)e));
    if (x_inout) {
        int rc = rdis_hip_get_x(p, nfree, free_vid, x_inout);
        if (rc) return rc;
    }
    if (fret) *fret = r.fret;
    if (delta) *delta = r.fret - r.finit;
    if (info8) {
        info8[0] = r.iters; info8[1] = r.stop; info8[2] = r.nfev; info8[3] = r.njev; info8[4] = r.nsolve; info8[5] = r.mu;
        info8[6] = r.ncam_blocks; info8[7] = r.npt_blocks;
    }
    if (nhist) *nhist = (int64_t)r.history.size();
    if (hist4)
        for (size_t i = 0; i < r.history.size() && (int64_t)i < hist_cap; ++i) {
            hist4[4 * i] = r.history[i].mu; hist4[4 * i + 1] = r.history[i].dp_l2;
            hist4[4 * i + 2] = r.history[i].f_trial; hist4[4 * i + 3] = r.history[i].accepted;
        }
    return 0;
}

extern "C" int rdis_hip_components(rdis_hip_problem* p, const uint8_t* assigned, int64_t* ncomp, int64_t* nfree, int64_t* nfac) {
    if (!p) return RDIS_HIP_EINVAL;
    rdis_hip_ctx* c = p->ctx;
    if (!assigned && p->N > 0) return fail(c, RDIS_HIP_EINVAL, "components: assigned is null");
    USE_DEVICE(c);
    if (p->N >= INT32_MAX || p->F >= INT32_MAX) return fail(c, RDIS_HIP_ERANGE, "components: problem too large for 32-bit ids");
    int rc = dalloc(c, p->assigned, (size_t)std::max<int64_t>(p->N, 1));
    if (rc) return rc;
    if (p->N > 0) HIPCHK(c, hipMemcpyAsync(p->assigned.p, assigned, (size_t)p->N, hipMemcpyHostToDevice, c->stream));
    const int e = device_components(c->stream, p->kind, (int)p->N, (int)p->F, p->cam.as<int>(), p->pt.as<int>(),
                                    p->rowptr.as<int>(), p->vid.as<int>(), p->assigned.as<unsigned char>(), &p->cc_ws, &p->comps);
    if (e != 0) return fail(c, RDIS_HIP_EDEVICE, std::string("components: ") + hipGetErrorString((hipError_t)e));
    if (ncomp) *ncomp = p->comps.ncomp;
    if (nfree) *nfree = p->comps.nfree;
    if (nfac) *nfac = p->comps.nfac;
    return 0;
}

extern "C" int rdis_hip_components_fetch(rdis_hip_problem* p, int64_t* free_ptr, int64_t* free_vid, int64_t* fac_ptr, int64_t* fac_id) {
    if (!p) return RDIS_HIP_EINVAL;
    const ComponentLists& L = p->comps;
    if (L.free_ptr.empty()) return fail(p->ctx, RDIS_HIP_EINVAL, "components_fetch: call rdis_hip_components first");
    if (free_ptr) std::memcpy(free_ptr, L.free_ptr.data(), L.free_ptr.size() * sizeof(int64_t));
    if (fac_ptr) std::memcpy(fac_ptr, L.fac_ptr.data(), L.fac_ptr.size() * sizeof(int64_t));
    if (free_vid && !L.free_vid.empty()) std::memcpy(free_vid, L.free_vid.data(), L.free_vid.size() * sizeof(int64_t));
    if (fac_id && !L.fac_id.empty()) std::memcpy(fac_id, L.fac_id.data(), L.fac_id.size() * sizeof(int64_t));
    return 0;
}

extern "C" int rdis_hip_plan_debug_counters(rdis_hip_plan* L, int64_t* out32) {
    if (!L || !out32) return RDIS_HIP_EINVAL;
    rdis_hip_ctx* c = L->prob->ctx;
    USE_DEVICE(c);
    HIPCHK(c, hipMemcpyAsync(out32, L->prob->coop_timing.p, COOP_TM * sizeof(long long), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int rdis_hip_plan_get_vectors(rdis_hip_plan* L, int64_t comp, double* out, int64_t cap_doubles) {
    if (!L || !out || comp < 0 || comp >= L->ncomp) return RDIS_HIP_EINVAL;
    rdis_hip_ctx* c = L->prob->ctx;
    USE_DEVICE(c);
    if (L->dump_iters <= 0) return fail(c, RDIS_HIP_EINVAL, "get_vectors: dump_iters is 0");
    const int64_t n = L->h_free_ptr[(size_t)comp + 1] - L->h_free_ptr[(size_t)comp];
    const int64_t want = 2ll * L->dump_iters * n;
    if (cap_doubles < want) return fail(c, RDIS_HIP_EINVAL, "get_vectors: buffer too small");
    if (want == 0) return 0;
    HIPCHK(c, hipMemcpyAsync(out, L->vdump.as<double>() + 2ll * L->dump_iters * L->h_free_ptr[(size_t)comp],
                             (size_t)want * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

// one optimize() call: a transient plan in the problem's arena -- no hipMalloc / hipFree, one
// H2D copy of the decomposition, one of the start, the launch(es), one D2H copy of the results
extern "C" int rdis_hip_cgd_batch(rdis_hip_problem* p, int64_t ncomp, const int64_t* free_ptr,
                                  const int64_t* free_vid, const int64_t* fac_ptr, const int64_t* fac_id,
                                  double* x_inout, int32_t maxiters, double ftol, double* fret, double* delta,
                                  int32_t* iters, int32_t* status, int64_t* nfeval, int64_t* ngeval) {
    rdis_hip_plan* L = nullptr;
    // RDIS_HIP_TIMING=1: where a one-shot call's host time goes (stderr), for the tuning of this path
    static const bool timing = std::getenv("RDIS_HIP_TIMING") != nullptr;
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = timing ? now() : 0.0;
    int rc = plan_create_impl(p, true, ncomp, free_ptr, free_vid, fac_ptr, fac_id, &L);
    if (rc) return rc;
    const double t1 = timing ? now() : 0.0;
    rc = rdis_hip_plan_set_start(L, x_inout);
    double t2 = 0.0, t3 = 0.0;
    if (!rc && timing) { t2 = now(); rc = prepare_partition(L); t3 = now(); }
    if (!rc) rc = rdis_hip_plan_solve(L, maxiters, ftol);
    const double t4 = timing ? now() : 0.0;
    if (!rc) rc = rdis_hip_plan_fetch(L, x_inout, fret, delta, iters, status, nfeval, ngeval);
    if (timing)
        std::fprintf(stderr, "rdis_hip_cgd_batch: lists + index tables %.3f ms, start %.3f, partition %.3f, launch %.3f, wait + results %.3f\n",
                     t1 - t0, t2 - t1, t3 - t2, t4 - t3, now() - t4);
    if (rc) {   // kernels or copies that use the plan's host staging vectors / arena slices may still be queued
        (void)hipStreamSynchronize(p->ctx->stream);
        if (p->ctx->aux) (void)hipStreamSynchronize(p->ctx->aux);
    }
    rdis_hip_plan_destroy(L);
    return rc;
}
