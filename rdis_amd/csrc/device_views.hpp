// device_views.hpp -- plain structs handed to the kernels by value: where a packed
// problem and a decomposition plan live in HBM.
//
// HBM layout (structure of arrays, everything fp64 / int32, resident for the
// lifetime of the handle):
//   x[N]          currently assigned value of every variable (Variable::m_value);
//                 during a solve the free entries hold the clamped trial point
//   lo[N], hi[N]  single-interval domain (VariableDomain::interval())
//   BA  factors:  cam[F], pt[F] first variable id of the camera / point block,
//                 obs[F] (double2, one 16-byte load per factor);
//                 cam_blocks[] the distinct camera blocks, xrot[N] their rotation records
//   NLP factors:  coeff[F], rowptr[F+1], vid/expo/cons/sine[nnz]   (CSR)
//   plan:         order[ncomp] (heaviest component first), free_ptr/free_vid,
//                 fac_ptr/fac_id, v2s_ptr + slot_pos (gfac is variable-major: the
//                 partials that feed free variable i are the contiguous range
//                 gfac[v2s_ptr[i] .. v2s_ptr[i+1]) in factor-list order; slot_pos maps
//                 a factor's slot to its position there, so factor lanes scatter and
//                 variable lanes read a contiguous run),
//                 ws[5 * nfree] (p, xi, g, h, x_init per component, contiguous),
//                 dir[N] (search direction scattered by variable id; zero at
//                 non-free variables), gfac[12F | nnz] (per-factor partials of the
//                 last full-gradient evaluation), xstart[nfree], outputs.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace rdis_hip {

constexpr int KIND_BA = 0;
constexpr int KIND_NLP = 1;
constexpr int ROT_PER_FACTOR = 0, ROT_RECORDS = 1, ROT_CAMFIX = 2;

struct ProblemView {
    int kind;
    int N, F;
    double* x;
    const double* lo;
    const double* hi;
    const int* cam;
    const int* pt;
    const double2* obs;
    double* xrot;         // null, or a shadow of x holding at every camera block's first id the camera's
                          // rotation record (7 doubles, factors.hpp), current for the assigned x at launch
    int rot_mode;         // ROT_PER_FACTOR: not used; ROT_RECORDS: factors read the records and a component rewrites
                          // those of its free cameras (PlanView::cb) at every trial point; ROT_CAMFIX: no camera
                          // variable is free in the launch -- records only read, point partials only
    const double* coeff;
    const int* rowptr;
    const int* vid;
    const double* expo;
    const double* cons;
    const uint8_t* sine;
    const uint8_t* useexp;  // null, or per factor: value = coeff * exp(-product) (NonlinearProductFactor.cpp:140)
};

struct PlanView {
    int ncomp;
    const int* order;
    const int* free_ptr;
    const int* free_vid;
    const int* fac_ptr;
    const int* fac_id;
    const int* v2s_ptr;   // [nfree_total + 1]
    const int* slot_base; // [nfac_total + 1] first slot of each LISTED factor (plan-local: 12 per BA factor, arity per NLP factor)
    const int* cb_ptr;    // [ncomp + 1] ...
    const int* cb;        // ... camera blocks (first variable id) with a free rotation variable, per component
    const int* cb_li;     // [3 per block] local index of the block's three rotation variables, -1 = constant
    const int* slot_pos;  // [slot_base[nfac_total]] listed factor's slot -> position in gfac (variable-major), -1 = not a free variable
    // LDS-resident batch solver (solver_lds.hpp): a component's variables -- free ones and the constants its
    // factors read -- as slots, camera blocks first (9 each), then point blocks (3 each)
    const int* ls_ptr;    // [ncomp + 1] first slot of a component (equal to the next: the component has no table)
    const int* ls_vid;    // [slots] variable id
    const int* ls_free;   // [slots] local free index, -1 = constant
    const int* ls_ncb;    // [ncomp] camera blocks of the component
    const double2* ls_obs;    // [nfac_total] observation of every listed factor (copy in listed order: no indirection in the trial loop)
    int ls_cam_gfac;          // option lds_camera_sums = 0: camera partials through gfac[] like solver_wg.hpp (bit-for-bit comparisons)
    int ls_matrix;            // 1: line-search trials in matrix form (factors.hpp: ba_camera_trial / ba_trial_value / ba_trial_slope) -- a camera's
                              // rotation matrix and its derivative along the direction once per camera and trial, 16 + 16 doubles in LDS behind
                              // the solver's other arrays; 0: the vector form per factor (the bits of solver_wg.hpp)
    const int* ls_gptr;       // [ncomp + 1] wave-chunks (64 entries) of ls_gperm, per component
    const int* ls_gperm;      // per component: local indices of its listed factors grouped by camera block (listed order inside a
                          // group), every group padded to whole chunks with -1
    const unsigned* ls_fidx;  // [nfac_total] listed factor -> camera block | point block << 12 (block numbers within the component)
    // point-major streaming solver (solver_ptm.hpp): the same slot tables, point blocks ordered by their number of factors
    // and taken 64 at a time (a wave-chunk)
    const int* pm_pt0;    // [ncomp] a component's first point block in pm_rec
    const int* pm_ch0;    // [ncomp] its first entry in pm_cptr (which has one more entry than wave-chunks per component)
    const int* pm_cptr;   // a wave-chunk's factors: entries [pm_cptr[c], pm_cptr[c + 1]) of pm_cam / pm_obs / pm_cgp, slot-major --
                          // entry pm_cptr[c] + 64 t + lane is the t-th factor of the lane's point block
    double* pm_rec;       // [blocks][6] p, xi of a point block's three variables
    double* pm_gh;        // [blocks][6] g, h of the Polak-Ribiere recurrence for them
    float* pm_cbox;       // [entries of pm_cptr][8] per wave-chunk a box inside the domains of its blocks: lo[3], -, hi[3], - (floats, rounded inward)
    double* pm_bex;       // [blocks][6] the blocks' exact bounds lo[3], hi[3]
    const short* pm_cam;  // [entries] camera block (number within the component), -1 = no factor ...
    const double2* pm_obs;  // ... and observation of a point's factor
    // the gradient's rounds (solver_ptm.hpp: gradient_to_xi), for workgroup w = component * K + rank of the launch:
    const unsigned short* pm_grow;     // [entries] the factor's staging row in its round: its rank among the round's factors ordered by camera
    const unsigned short* pm_rounds;   // per round ptm_round_stride(ncb) 16-bit words: per camera the first row of its segment [ncb + 1]
    const long long* pm_rd_off;        // [ncomp * K] a workgroup's first word in pm_rounds ...
    const int* pm_rd_n;                // ... and its number of rounds
    int pm_round_slots;                // slots a round evaluates and stages (1 or 2): what the tables above were built for
    // a trial's work by wave (solver_ptm.hpp: eval_line): rows (wave-chunk, first entry, end entry); wave w of workgroup
    // (component, rank) takes the rows w, w + waves, w + 2 waves, ... up to the first empty one.  A workgroup's table:
    // its number of rows R and three ints of nothing, then the rows' chunks [R], first entries [R], end entries [R]
    const int* pm_segs;
    const long long* pm_sg_off;        // [ncomp * K] a workgroup's table in pm_segs
    int* st_ev;           // stale-cache emulation (solver_lds.hpp): per listed factor the assignment of its last value evaluation ...
    double* st_val;       // ... and that value; null unless the plan's option emulate_stale_cache is set
    // the parity option (plan option factor_rounding = 1; solver_coop.hpp / solver_lds.hpp in refround_kernels.hip): sums in the
    // reference's order are added from these by one lane / wave
    double* seq_val;      // [nfac_total] a listed factor's value at the trial point at hand
    double* seq_ab;       // [2][seq_n] by free index: the terms of gg and dgg (cooperative solver)
    int seq_n;            // nfree_total
    long long* timing;    // debug counters of the batch solvers (-DRDIS_COOP_TIMING builds), or null
    double* ws;           // 5 vectors per component, component c at 5*free_ptr[c]
    double* dir;          // [N]
    double* gfac;         // [v2s_ptr[nfree_total]] per-factor partials, variable-major
    const double* xstart; // [nfree_total]
    double* xout;         // [nfree_total]
    double* fret;
    double* delta;
    int* iters;
    int* status;
    long long* nfeval;
    long long* ngeval;
    double* trace;        // [ncomp][trace_cap][4] or null
    int* trace_n;         // [ncomp]
    int trace_cap;
    double* vdump;        // [2 * dump_iters * nfree_total] p, xi at the start of each line search, or null
    int dump_iters;
};

}  // namespace rdis_hip
