// solver_ptm.hpp -- one workgroup (or a few) solves one bundle-adjustment component that is too large
// for the LDS-resident solver (solver_lds.hpp): its CAMERA blocks keep their slots in LDS, its POINT
// blocks stream from HBM once per trial point, and the trial loop runs point by point.
//
// What bounds a batch of many large components is instruction issue (fp64 VALU) -- as long as the
// memory system is not in the way.  solver_wg.hpp forms every trial point in global memory (p, xi,
// lo, hi read and x written per variable, then x and the direction gathered per FACTOR: measured
// 2.8 x the algorithmic bytes in HBM traffic on 256 components of ladybug's size).  Here
//
//   cameras   slots in LDS as in solver_lds.hpp (Pv, XI, LO, HI, X, rotation records; also g and h
//             of the Polak-Ribiere recurrence);
//   points    one 48-byte record per point block in HBM: p[3], xi[3].  The blocks stand in camera-lexicographic order
//             (blocks seen by the same cameras are neighbours), dealt out PTM_SPREAD chunks apart so that a chunk's
//             blocks keep differing in their cameras, and are taken 64 at a time, a wave-chunk.  Per CHUNK a box of six
//             floats rounded INWARD that lies inside every block's domain: a trial value strictly inside it is inside
//             the block's bounds and is its own clamp; only a lane whose value is not (an active bound: rare) fetches
//             the block's exact bounds from a second array and clamps as ever -- the same bits either way, 48 bytes less
//             per block and trial than bounds in the record.  A lane takes a block, loads its record ONCE, forms
//             clamp(p + a xi) in registers and evaluates the block's factors one after the other against the cameras
//             in LDS.  The trial point of a point variable is never stored.
//   factors   slot-major per wave-chunk: entry cptr[chunk] + 64 t + lane is the t-th listed factor of the
//             lane's block (camera block 2 B + observation 16 B; camera -1 = the block has fewer
//             factors).  A wave's loads of a slot are 64 neighbours, and their addresses depend on
//             nothing the wave has loaded before.
//
// Nothing a wave needs is asked for when it is needed (round 4).  A chunk's entry range is read two chunks ahead;
// its point records and its box are loaded while the chunk BEFORE it is evaluated, the slots of block t + 1 while
// block t (PTM_BLK slots) is evaluated, and the first block of the next chunk during the last block of this one.  Before, every chunk began with five dependent
// memory round trips (range -> records -> bounds x 3) in front of four slots' arithmetic, and three waves per SIMD
// do not hide that: 62 % of all wave-cycles were parked (profiles/r03_c_pmc_sq_synthL.txt).
//
// A trial's cameras are prepared once per trial point: rotation matrix R(w) and, for the slope, dR along the direction
// (factors.hpp: ba_camera_trial, 16 doubles a camera in LDS, read as 16-byte pairs at a stride of 10 slots); a factor
// is then R p + t, the projection and the radial term -- 88 instead of 164 fp64 operations for value and slope.
//
// HBM bytes per value+slope trial: 16 per point variable + 18 per factor (SURVEY 8d counts 16 + 24:
// x and g once per variable, observation + two indices per factor), no write traffic at all.
//
// The full gradient, once per CG iteration, is ONE point-major pass like a trial (forward + adjoint; a block's
// point entries summed in registers in slot order = factor-list order, src/State.h:157-210, and written to the
// record).  The nine CAMERA partials of a factor never leave the compute unit: the pass runs in ROUNDS -- in round r
// every wave evaluates its r-th slot of factors and leaves the 64 x 9 camera partials in an LDS staging area; after a
// barrier lane (camera c, entry k) adds up the staged partials of camera c, whose staging indices the plan lists per
// round grouped by camera (two bytes a factor and gradient from HBM; the list of round r + 1 is fetched during round
// r), into the camera's gradient entry in LDS.  Sums in a fixed order (rounds in order, within a round by wave and
// lane): the same bits run to run.  (Round 3 handed the camera partials from a point-major pass to a pass in camera
// order through HBM, 160 bytes per factor and gradient -- most of the 1.6 x the algorithmic bytes the launch moved.)
//
// GROUP = true: K workgroups share a component (cgd_ptmg_kernel, a cooperative launch of several such
// groups side by side).  A launch with fewer components than compute units -- one rank's share of a
// decomposition spread over eight GPUs -- would otherwise leave the rest of the device idle while every
// busy unit works through its component's factors alone.  The camera slots are replicated in every
// workgroup's LDS (each steps the same control logic on the same sums and applies the same vector updates to
// them); the point chunks are dealt out round robin (chunk c belongs to workgroup c mod K).  Per trial point the
// workgroups exchange their partial (value, slope) through the granules of grid_sync.hpp (every wave an entry,
// summed in entry order: the same bits in every workgroup); per CG iteration ONE ordered grid barrier hands over
// the workgroups' partial camera gradients (9 ncb doubles each, summed in rank order by everybody).
// A group's workgroups sit on one XCD (blockIdx mod 8 is the same for all of them), so what they hand to
// each other through plain stores stays in that XCD's L2.
#pragma once
#include "solver_lds.hpp"
#include "grid_sync.hpp"
#include "ptm_api.hpp"

namespace rdis_hip {

// the float nearest to a bound on its inner side (never subnormal: conversions may flush those)
__device__ __forceinline__ float inner_lo32(double lo) {
    float f = (float)lo;
    if ((double)f < lo) f = nextafterf(f, __builtin_inff());
    if (fabsf(f) < 1.17549435e-38f) f = lo <= 0.0 ? 0.0f : 1.17549435e-38f;
    return f;
}
__device__ __forceinline__ float inner_hi32(double hi) {
    float f = (float)hi;
    if ((double)f > hi) f = nextafterf(f, -__builtin_inff());
    if (fabsf(f) < 1.17549435e-38f) f = hi >= 0.0 ? 0.0f : -1.17549435e-38f;
    return f;
}

// camera block (-1: no factor) and observation of PTM_BLK consecutive slots of a lane's point block
struct SlotBlock {
    int c[PTM_BLK];
    double2 o[PTM_BLK];
    int r[PTM_BLK];   // (the gradient only) the staging row of the factor's camera partials in its round
};
// a point block's record and float bounds as loaded (a chunk ahead of their use).  Twelve separate loads on purpose:
// the values are carried around the chunk loop, and the halves of a 16-byte load's register tuple are not coalesced
// with loop-carried registers -- the compiler copies them out right behind the load, i.e. waits for it at once.
struct PtRecs {
    double p[3], xi[3];
    float lo[3], hi[3];   // the chunk's box: inside every block's domain (the same in every lane)
    double g[3];          // (the gradient pass only) g of the recurrence
};

// LOCAL (wide groups only): a component with more cameras than a compute unit's LDS holds.  A workgroup then keeps only the
// cameras its OWN point chunks meet -- it owns a contiguous range of the camera-sorted chunk order, so they are few -- under local
// numbers (the factor stream names cameras by them); the component's cameras are `ncbg`, and what crosses workgroups (the partial
// camera gradients) goes through per-camera lists of (rank, local number).
struct PtmLocal {
    int ncbg;            // cameras of the component
    int cbeg, cend;      // this workgroup's chunks [cbeg, cend)
    const int* cmap;     // [ncb] local -> component camera
    const int* own;      // [ncb] 1: this workgroup speaks for the camera where a camera counts once (the sums' terms, write-back)
    const int* cr_ptr;   // [ncbg + 1] per component camera the workgroups that hold it ...
    const int* cr;       // ... as rank << 8 | local number, ascending rank
    double* tot;         // [2][PTM_CS ncbg] the cameras' summed gradient entries (two gradients' worth)
};

// PAIR: the kernel may run its gradient rounds two slots at a time (rs == 2, the plan's choice) -- compiled in for workgroups of up
// to 512 lanes only: at 768 lanes (168 registers a lane) the second slot's evaluation spills the whole pass.
template <int ROT, bool GROUP = false, class ST = SmallCoopState, bool LOCAL = false, bool PAIR = false>
struct PtmEnv {
    const ProblemView& P;
    const PlanView& L;
    int comp, n, m, f0, c0, tid, nt, nwaves;
    int ncb, npb, npc;        // camera blocks (9 LDS slots each), point blocks (a record each), wave-chunks of point blocks
    const int* svid;          // variable id of a slot (cameras, then points)
    const int* sfree;         // local free index of a slot, -1 = constant (global copy; the cameras' also in SF)
    double *Pv, *XI, *LO, *HI, *X, *GC, *HC, *ROTR;   // LDS, cameras (GC, HC: g and h of the recurrence)
    double *CTR, *CDR;        // LDS [ncb][PTM_TS] each: the cameras' trial records at the trial point at hand (factors.hpp: CAM_TRIAL)
    double* STG;              // LDS [rs nt + 1][9]: the camera partials of a round's factors, grouped by camera (a factor's row: grow); a row of zeros
    unsigned short* RL;       // LDS [2][rl_cap]: round tables (per camera the first row of its segment)
    int rl_cap;
    int* SF;                  // LDS: local free index of a camera slot
    double* PT;               // [6][npb] the point blocks' p[3], xi[3], a plane each (six loads a lane that no pass of the compiler
                              // can put together: the halves of a wider load are copied out right behind it, i.e. waited for at once)
    long long pstr;           // ... a plane's length (npb)
    float* CBX;               // [npc][8] per wave-chunk a box inside the domains of all its blocks (lo[3], hi[3] as floats rounded inward)
    double* PE;               // [npb][6] the blocks' exact bounds (lo[3], hi[3])
    const int* cptr;          // [npc + 1] a point chunk's entries ...
    const short* pcam;        // ... their camera block (-1: none; at most 4095 camera blocks: two bytes a factor and trial)
    const double2* pobs;      // ... their observation
    const unsigned short* grow;     // ... the staging row of their camera partials in the gradient's rounds
    const unsigned short* rounds;   // this workgroup's round tables (per camera the first row of its segment), rl_stride 16-bit words each ...
    int nrounds, rl_stride;         // ... and their number
    int rs;                         // slots a round evaluates and stages: 1, or 2 (a whole block of slots: half the barriers and sums' heads; the plan's choice, by the LDS)
    const int* segs;                // a trial's work by wave: rows (wave-chunk, first entry, end entry), wave w's at w, w + waves, ...;
    int seg_rows;                   // three planes of seg_rows ints (three loads no pass of the compiler can put together, see PT)
    int sg[9];                      // this wave's first three rows (the same for every trial point of the solve: scalar registers)
    double* PG;               // [6][npb] g[3] and h[3] of the recurrence for the blocks' variables, likewise
    double (*red)[3][MAX_WAVES];
    int parity;
    double* tr;
    int trn, lm_count;
    // GROUP: rank r of the K workgroups that share the component, their exchange, their partial camera gradients in HBM
    int r, K;
    GridSyncT<ST> GX;
    double* xch;              // [2][K][xch_stride] the workgroups' partial camera gradients; behind them [2][xch_stride] their sums (wide groups)
    int xch_stride, gpar;
    PtmLocal LC;              // (LOCAL)
    // the first slot of the point blocks in the component's slot tables (svid, sfree): behind ALL its cameras
    __device__ __forceinline__ int pslot0() const { return PTM_CS * (LOCAL ? LC.ncbg : ncb); }
    // the component-wide slot of this workgroup's camera slot s
    __device__ __forceinline__ int cam_gslot(int s) const { if constexpr (LOCAL) { const int c = s / PTM_CS; return PTM_CS * LC.cmap[c] + (s - PTM_CS * c); } else return s; }
    // does this workgroup speak for camera slot s where a camera counts once?
    __device__ __forceinline__ bool speaks_for(int s) const { if constexpr (LOCAL) return LC.own[s / PTM_CS] != 0; else return !GROUP || r == 0; }
#ifdef RDIS_COOP_TIMING
    long long tmv[32];
    long long wv[32];
    long long tbar;   // (-DRDIS_COOP_TIMING=2: a wave's clock behind eval_line's barrier; slots 0..15 / 16..31 then hold, per wave,
                      // the cycles from there to the end of its factors / from the trial's start to there, summed over the trials)
#endif

    // the wave-chunks of point blocks this wave takes: chunk c belongs to workgroup c mod K, wave (c / K) mod nwaves
    __device__ __forceinline__ int first_chunk() const { return LOCAL ? LC.cbeg + (tid >> 6) : GROUP ? r + K * (tid >> 6) : (tid >> 6); }
    __device__ __forceinline__ int chunk_step() const { return LOCAL ? nwaves : GROUP ? K * nwaves : nwaves; }
    __device__ __forceinline__ int chunk_end() const { return LOCAL ? LC.cend : npc; }
    // the point blocks this workgroup owns, a lane each
    template <class Fn>
    __device__ __forceinline__ void my_points(Fn fn) const {
        const int lane = tid & 63;
        for (int c = first_chunk(); c < chunk_end(); c += chunk_step()) {
            const int ps = 64 * c + lane;
            if (ps < npb) fn(ps);
        }
    }
    // ... and their variables: fn(point block, k)
    template <class Fn>
    __device__ __forceinline__ void my_point_vars(Fn fn) const {
        my_points([&](int ps) {
#pragma unroll
            for (int k = 0; k < 3; ++k) fn(ps, k);
        });
    }

    // sums of the first KK of (a, b, max mx) over the component: the workgroup's, or the group's (entry order)
    template <int KK>
    __device__ void sumk(double& a, double& b, double& mx) {
        if constexpr (GROUP) {
            if constexpr (GridSyncT<ST>::wg_entry) {   // (a wide group: the workgroup's sums first, one entry a workgroup)
                a = wave_sum(a); b = wave_sum(b); mx = wave_max(mx);
                if ((tid & 63) == 0) { red[parity][0][tid >> 6] = a; red[parity][1][tid >> 6] = b; red[parity][2][tid >> 6] = mx; }
                __syncthreads();
                combine_waves<3>(red[parity], nwaves, a, b, mx);
                parity ^= 1;
            }
            GX.exchange(a, b, mx, SYNC_NONE);
            return;
        }
        a = wave_sum(a);
        if constexpr (KK >= 2) b = wave_sum(b);
        if constexpr (KK >= 3) mx = wave_max(mx);
        if (nwaves > 1) {
            const int w = tid >> 6;
            if ((tid & 63) == 0) {
                red[parity][0][w] = a;
                if constexpr (KK >= 2) red[parity][1][w] = b;
                if constexpr (KK >= 3) red[parity][2][w] = mx;
            }
            __syncthreads();
            combine_waves<KK>(red[parity], nwaves, a, b, mx);
            parity ^= 1;
        }
    }
    __device__ void trace(int tag, double a, double b, double c) {
        if (tr != nullptr && tid == 0 && (!GROUP || r == 0)) {
            if (trn < L.trace_cap) { double* rec = tr + 4ll * trn; rec[0] = (double)tag; rec[1] = a; rec[2] = b; rec[3] = c; }
            ++trn;
        }
    }

    // ---- the cameras' trial point (LDS) -----------------------------------------------------------------
    // X = clamp(p + a xi) slot by slot as in solver_lds.hpp; then, a lane per camera with a free variable (the workgroup's
    // last lanes: its first wave steps the control logic), what the camera's factors read at this trial point:
    //   TRIAL   the trial records of factors.hpp -- rotation matrix, translation, f, k1, k2 (CTR) and, SLOPE, their derivative
    //           along the search direction (CDR); a camera without a free variable keeps the records init_vectors gave it;
    //   !TRIAL  the rotation record (axis, angle, sine, cosine: ROTR) the gradient's forward + adjoint sweep reads.
    template <bool TRIAL, bool SLOPE, class At>
    __device__ void refresh_records(At at) {
        if constexpr (ROT == ROT_CAMFIX) return;
        for (int c = nt - 1 - tid; c < ncb; c += nt) {
            const int s0 = PTM_CS * c;
            bool any = false;
#pragma unroll
            for (int q = 0; q < 9; ++q) any = any || SF[s0 + q] >= 0;
            if (!any) continue;
            double xc[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) { const int s = s0 + ptm_slot_of(k); xc[k] = SF[s] >= 0 ? at(s) : Pv[s]; }
            BaFwd rot;
            ba_rotation(xc[0], xc[1], xc[2], rot);
            if constexpr (TRIAL) {
                ba_camera_trial(rot, xc, CTR + PTM_TS * c);
                if constexpr (SLOPE) {
                    double dc[9];
#pragma unroll
                    for (int k = 0; k < 9; ++k) dc[k] = XI[s0 + ptm_slot_of(k)];   // (a constant's entry is zero)
                    ba_camera_trial_dir(rot, dc, CDR + PTM_TS * c);
                }
            } else {
                double* rr = ROTR + PTM_RS * c;
                rr[ROT_V0] = rot.v0; rr[ROT_V0 + 1] = rot.v1; rr[ROT_V0 + 2] = rot.v2;
                rr[ROT_THETA] = rot.theta; rr[ROT_ITHETA] = rot.itheta; rr[ROT_SIN] = rot.s; rr[ROT_COS] = rot.c;
            }
        }
    }
    enum : int { AT_LINE = 0, AT_START = 1 };
    // RECS: 0 = the gradient's point (rotation records), 1 = a value trial (trial records), 2 = a value + slope trial
    // SYNC = false: the caller's barrier follows (eval_line puts its first loads in front of it)
    template <int MODE, int RECS, bool SYNC = true>
    __device__ void assign_cameras(double a) {
#pragma clang fp contract(off)
        const double* xs = L.xstart + f0;
        auto at = [&](int s) {
#pragma clang fp contract(off)
            if constexpr (MODE == AT_START) return clampd(xs[SF[s]], LO[s], HI[s]);
            else { const double t = a * XI[s]; return clampd(Pv[s] + t, LO[s], HI[s]); }
        };
        if constexpr (RECS == 0) {   // (the trial records are formed from p, xi and the bounds directly: X is the gradient's)
            for (int s = tid; s < PTM_CS * ncb; s += nt)
                if (SF[s] >= 0) X[s] = at(s);
        }
        refresh_records<RECS != 0, RECS == 2>(at);
        if constexpr (SYNC) __syncthreads();
    }

    // ---- a point's trial values: clamp(p + a xi) from its record (registers only) -------------------
    // the records of chunk c (this lane's block: the chunk's last where the chunk has fewer) and the chunk's box
    template <bool GRAD = false>
    __device__ __forceinline__ void load_recs(int c, PtRecs& R) const {
        const long long ps = min(64 * c + (tid & 63), npb - 1);
        const float* bq = CBX + 8 * c;
#pragma unroll
        for (int k = 0; k < 3; ++k) { R.p[k] = PT[k * pstr + ps]; R.xi[k] = PT[(3 + k) * pstr + ps]; R.lo[k] = bq[k]; R.hi[k] = bq[4 + k]; }
        if constexpr (GRAD) {
#pragma unroll
            for (int k = 0; k < 3; ++k) R.g[k] = PG[k * pstr + ps];
        }
    }
    // exact clamp of a block's three values that are not all strictly inside the inward-rounded bounds
    __device__ __forceinline__ void clamp_exact(int ps, double (&x)[3]) const {
        const double2* be = reinterpret_cast<const double2*>(PE + (long long)PT_BND * ps);
        const double2 e0 = be[0], e1 = be[1], e2 = be[2];
        x[0] = clampd(x[0], e0.x, e1.y);
        x[1] = clampd(x[1], e0.y, e2.x);
        x[2] = clampd(x[2], e1.x, e2.y);
    }
    // (all six comparisons, no short circuit: nothing here may wait for memory)
    __device__ __forceinline__ bool inside32(const PtRecs& R, const double (&x)[3]) const {
        int in = 1;
#pragma unroll
        for (int k = 0; k < 3; ++k) in &= (int)(x[k] > (double)R.lo[k]) & (int)(x[k] < (double)R.hi[k]);
        return in != 0;
    }
    // clamp(p + a xi) and the direction, from a loaded record
    __device__ __forceinline__ void point_line(const PtRecs& R, int ps, double a, double (&x)[3], double (&d)[3]) const {
#pragma clang fp contract(off)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            d[k] = R.xi[k];
            const double t = a * d[k];
            x[k] = R.p[k] + t;
        }
        if (!inside32(R, x)) clamp_exact(ps, x);
    }
    // clamp(p): the gradient's point
    __device__ __forceinline__ void point_position(const PtRecs& R, int ps, double (&x)[3]) const {
#pragma unroll
        for (int k = 0; k < 3; ++k) x[k] = R.p[k];
        if (!inside32(R, x)) clamp_exact(ps, x);
    }
    // clamp(x_start): the rollback's point (rare: loads where it stands)
    __device__ __forceinline__ void point_start(int ps, double (&x)[3]) const {
        const double* be = PE + (long long)PT_BND * ps;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int fi = sfree[pslot0() + 3 * ps + k];
            x[k] = fi >= 0 ? clampd(L.xstart[f0 + fi], be[k], be[3 + k]) : PT[k * pstr + ps];
        }
    }
    __device__ __forceinline__ double forward(int c, double2 o, const double (&x)[3], double (&v)[12], BaFwd& t) const {
        // (16-byte reads: a camera's slots stand [t f k1 k2 | r | pad], ten doubles from a 16-byte boundary)
        const double2* xc = reinterpret_cast<const double2*>(X + PTM_CS * c);
        const double2 a0 = xc[0], a1 = xc[1], a2 = xc[2];
        v[3] = a0.x; v[4] = a0.y; v[5] = a1.x; v[6] = a1.y; v[7] = a2.x; v[8] = a2.y;
#pragma unroll
        for (int k = 0; k < 3; ++k) v[9 + k] = x[k];
        // (the rotation always comes from the camera's record: ROT_RECORDS rewrites it at the gradient's point, ROT_CAMFIX never)
        v[0] = v[1] = v[2] = 0.0;
        const double2* rr = reinterpret_cast<const double2*>(ROTR + PTM_RS * c);
        const double2 r0 = rr[0], r1 = rr[1], r2 = rr[2], r3 = rr[3];
        t.v0 = r0.x; t.v1 = r0.y; t.v2 = r1.x; t.theta = r1.y; t.itheta = r2.x; t.s = r2.y; t.c = r3.x;
        return ba_project(v, o.x, o.y, t);
    }
    // a factor's value (and slope) at the trial point, from its camera's trial records (factors.hpp: matrix form)
    // (RDIS_PTM_ABLATE, measurement builds only -- wrong results: 1 = no slope arithmetic, 2 = every factor reads camera 0
    // (no LDS traffic in the loop), 3 = every block of slots is the component's first (no HBM traffic for the factors))
    __device__ __forceinline__ void factor_trial(int cc, double2 o, const double (&x)[3], const double (&dp)[3], bool slope, double& af, double& as) const {
#if defined(RDIS_PTM_ABLATE) && RDIS_PTM_ABLATE == 2
        cc = 0;
#endif
#if defined(RDIS_PTM_ABLATE) && RDIS_PTM_ABLATE == 1
        slope = false;
#endif
        double TR[CAM_TRIAL], DR[CAM_DIR];
        const int off = __mul24(cc, PTM_TS);   // (cameras < 2^12: the 24-bit product is exact and full rate, the 32-bit one a quarter of it)
        const double2* tc = reinterpret_cast<const double2*>(CTR + off);
#pragma unroll
        for (int k = 0; k < CAM_TRIAL / 2; ++k) { const double2 v = tc[k]; TR[2 * k] = v.x; TR[2 * k + 1] = v.y; }
        BaTrial t;
        af += ba_trial_value(TR, x, o.x, o.y, t);
        if (slope) {
            if constexpr (ROT != ROT_CAMFIX) {
                const double2* dc = reinterpret_cast<const double2*>(CDR + off);
#pragma unroll
                for (int k = 0; k < CAM_DIR / 2; ++k) { const double2 v = dc[k]; DR[2 * k] = v.x; DR[2 * k + 1] = v.y; }
            } else {
#pragma unroll
                for (int k = 0; k < CAM_DIR; ++k) DR[k] = 0.0;
            }
            as += ba_trial_slope<ROT == ROT_CAMFIX>(t, TR, DR, x, dp);
        }
    }
    // the (camera, observation) entries of PTM_BLK consecutive slots of a chunk, from entry e on (this lane's: e + lane, + 64 per
    // slot).  Unconditional: slots beyond the chunk's last are loaded too (the arrays are padded) and not looked at.
    template <bool ROWS = false>
    __device__ __forceinline__ void load_block(int e, SlotBlock& B) const {
        const int lane = tid & 63;
#if defined(RDIS_PTM_ABLATE) && RDIS_PTM_ABLATE == 3
        e = cptr[0] + (e & 64);
#endif
#pragma unroll
        for (int k = 0; k < PTM_BLK; ++k) {
            B.c[k] = pcam[e + 64 * k + lane]; B.o[k] = pobs[e + 64 * k + lane];
            if constexpr (ROWS) B.r[k] = grow[e + 64 * k + lane];
        }
    }
    // holds a block of slots in registers at this point of the program (no instruction)
    static __device__ __forceinline__ void pin_block(SlotBlock& B) {
#pragma unroll
        for (int k = 0; k < PTM_BLK; ++k) asm volatile("" : "+v"(B.c[k]), "+v"(B.o[k].x), "+v"(B.o[k].y));
    }
    // This workgroup's share of the sums at clamp(p + a xi): its point chunks, a block per lane, the block's factors
    // PTM_BLK slots at a time.  Everything is asked for a block of slots before it is used: while one block of slots is
    // evaluated the next one's cameras and observations are in flight (the next chunk's first block during a chunk's last),
    // from a chunk's first block on the next chunk's point records, and the entry range of the chunk after that.
    // The waves' shares are equal runs of blocks of slots in chunk order (the plan's rows, rdis_hip.hip: ptm_build_segments): a
    // wave evaluates some whole chunks and at most part of the slots of one more at either end.
    // The workgroup's barrier behind assign_cameras<.., false> stands in here, behind the first loads: the wave that forms
    // the cameras' records is the last to arrive, and what the others asked for is on its way meanwhile.
    template <bool SLOPE>
    __device__ __forceinline__ void eval_line(double a, double& af, double& as) {
        const int lane = tid & 63, cs = nwaves;
        // this wave's share: runs of whole blocks of slots of a chunk (the plan's rows; two rows of nothing behind the last)
        const int* sp = segs + (tid >> 6) + 2 * cs;
        const int sr = seg_rows;
        int cu = sg[0], e = sg[1], e1 = sg[2];
        const bool work = e < e1;
        int cx = sg[3], ne = sg[4], ne1 = sg[5], vc = sg[6], v0 = sg[7], v1 = sg[8];
        PtRecs R = {};
        SlotBlock N = {};
        if (work) {
            load_recs(cu, R);
            load_block(e, N);
        }
        __syncthreads();   // the cameras' records of this trial point
#ifdef RDIS_COOP_TIMING
        tbar = clock64();
#endif
        if (!work) return;
        double x[3] = {0.0, 0.0, 0.0}, dp[3] = {0.0, 0.0, 0.0};
        // Where the loop waits for memory: only where a block of slots is taken over from the registers it was loaded into (B = N,
        // pinned to its place in the program), i.e. behind a block's arithmetic.  The counter of loads in flight is kept in order
        // of issue, so nothing may be asked for between a wait and the loads it is for; a load under a condition, or registers
        // that one path of a loop writes and another keeps, are waited for where the paths join (hence two loops, and
        // unconditional loads: behind the last run the first row of nothing names chunk 0 -- loaded, never looked at).
        SlotBlock B = N;
        pin_block(B);
        int nblk = 0;
        for (;;) {   // a run: its trial point from the records; the next run's records
            point_line(R, min(64 * cu + lane, npb - 1), a, x, dp);
            load_recs(cx, R);
            for (;;) {   // its blocks of slots
                // The waves of a SIMD are not served alike -- the arbiter prefers one, and at two waves a SIMD a slot took 737 .. 1039
                // cycles by wave (tools/microbench/trial_loop.hip) --, while their shares are equal: the phase waited for the
                // wave served last.  Issue priority from the count of blocks done, mod 4: a wave one block behind its
                // neighbours is one level above them.  Scheduling only, the same bits (round 5: 125 components as pairs 14.45
                // -> 14.05 ms, 1000 components 85.8 -> 84.3).
                {
                    const int lv = (0 - nblk) & 3;
                    if (lv == 3) __builtin_amdgcn_s_setprio(3); else if (lv == 2) __builtin_amdgcn_s_setprio(2);
                    else if (lv == 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
                    ++nblk;
                }
                const int bn = min(PTM_BLK, (e1 - e) >> 6);
                const int en = e + 64 * PTM_BLK;
                const bool more = en < e1;
                load_block(more ? en : ne < ne1 ? ne : e, N);
#pragma unroll
                for (int k = 0; k < PTM_BLK; ++k)
                    if (k < bn && B.c[k] >= 0) factor_trial(B.c[k], B.o[k], x, dp, SLOPE, af, as);
                B = N;
                pin_block(B);
                if (!more) break;
                e = en;
            }
            if (ne >= ne1) break;
            cu = cx; e = ne; e1 = ne1;
            cx = __builtin_amdgcn_readfirstlane(vc); ne = __builtin_amdgcn_readfirstlane(v0); ne1 = __builtin_amdgcn_readfirstlane(v1);   // (asked for a run ago)
            sp += cs;
            vc = sp[0]; v0 = sp[sr]; v1 = sp[2 * sr];
        }
        __builtin_amdgcn_s_setprio(0);
    }
    // ... at clamp(x_start) (the rollback; value only)
    __device__ void eval_start(double& af) {
        const int lane = tid & 63;
        for (int c = first_chunk(); c < chunk_end(); c += chunk_step()) {
            const int e0 = __builtin_amdgcn_readfirstlane(cptr[c]), e1 = __builtin_amdgcn_readfirstlane(cptr[c + 1]);
            const int ps = 64 * c + lane;
            double x[3] = {0.0, 0.0, 0.0}, dp[3] = {0.0, 0.0, 0.0}, as = 0.0;
            if (ps < npb) point_start(ps, x);
            for (int e = e0; e < e1; e += 64) {
                const int cc = pcam[e + lane];
                const double2 o = pobs[e + lane];
                if (cc >= 0) factor_trial(cc, o, x, dp, false, af, as);
            }
        }
    }

    static constexpr bool UNIFORM = true;
    static constexpr int SPEC = 1;
    static constexpr bool FUSED_GRADIENT = true;
    __device__ bool stepper() const { return threadIdx.x < 64; }
    __device__ bool writer() const { return (threadIdx.x & 63) == 0; }
    __device__ void sync() const { __syncthreads(); }
    __device__ bool tracing() const { return tr != nullptr; }
    __device__ bool aborted() const { if constexpr (GROUP) return GX.dead; else return false; }
    // cycle stamps of the launch's first workgroup (build with -DRDIS_COOP_TIMING; rdis_hip_plan_debug_counters):
    // 0 cameras' trial point, 1 this workgroup's factors, 2 sums (GROUP: the exchange), 3 their number (value+slope
    // trials); 4 / 5 gradient: the rounds / what follows them, 10 their number; 8 / 9 control step / hand-over, 12.. cycles
    // per request kind, 22.. their counts; GROUP: 28 exchanges, 29 publish, 30 sweep, 31 tail of the exchanges
#ifdef RDIS_COOP_TIMING
    __device__ long long* wave_cycles() const { return reinterpret_cast<long long*>(STG); }   // (the staging area is idle during trials)
    __device__ void tick(int slot, long long dt) { tmv[slot] += dt; }
    __device__ long long clock() const { return clock64(); }
#else
    __device__ void tick(int, long long) {}
    __device__ long long clock() const { return 0; }
#endif
    __device__ double eval_value(double a, bool restore) {
        double af = 0.0, as = 0.0, dummy = 0.0;
        if (restore) { assign_cameras<AT_START, 1>(0.0); eval_start(af); }
        else { assign_cameras<AT_LINE, 1, false>(a); eval_line<false>(a, af, as); }
        sumk<1>(af, as, dummy);
        return af;
    }
    __device__ void eval_value_slope(double a, double& f, double& s) {
        double af = 0.0, as = 0.0, dummy = 0.0;
        const long long t0 = clock();
        assign_cameras<AT_LINE, 2, false>(a);
        const long long t1 = clock();
        eval_line<true>(a, af, as);
        const long long t2 = clock();
#ifdef RDIS_COOP_TIMING
        if ((tid & 63) == 0) { wave_cycles()[tid >> 6] = t2 - t1; wave_cycles()[16 + (tid >> 6)] = tbar - t0; wave_cycles()[32 + (tid >> 6)] = t2 - tbar; }   // (read after the barrier of the sums)
#endif
        sumk<2>(af, as, dummy);
        f = af; s = as;
        tick(0, t1 - t0); tick(1, t2 - t1); tick(2, clock() - t2); tick(3, 1);
#ifdef RDIS_COOP_TIMING
        if (tid == 0) {   // the factor phase by wave: slowest, fastest, mean (slots 20, 6, 21)
            long long mx = 0, mn = 1ll << 62, sm = 0;
            for (int w = 0; w < nwaves; ++w) { const long long v = wave_cycles()[w]; mx = v > mx ? v : mx; mn = v < mn ? v : mn; sm += v; }
            tick(20, mx); tick(6, mn); tick(21, sm / nwaves);
#pragma unroll
            for (int w = 0; w < 16; ++w)   // (constant indices: an array of the environment indexed by a variable would move it to memory)
                if (w < nwaves) { wv[w] += wave_cycles()[32 + w]; wv[16 + w] += wave_cycles()[16 + w]; }
        }
#endif
    }

    __device__ void init_vectors() {   // CGD .cpp:34-39: p = x0 (unclamped); constants hold their assigned value
        const double* xs = L.xstart + f0;
        for (int s = tid; s < PTM_CS * ncb; s += nt) {
            const int fi = SF[s], v = svid[cam_gslot(s)];
            if (fi >= 0) {
                const double lo = P.lo[v], hi = P.hi[v], x0 = xs[fi];
                Pv[s] = x0; LO[s] = lo; HI[s] = hi; X[s] = clampd(x0, lo, hi);
            } else {
                const double xc = P.x[v];
                Pv[s] = xc; X[s] = xc; LO[s] = -__builtin_inf(); HI[s] = __builtin_inf();
            }
            XI[s] = 0.0; GC[s] = 0.0; HC[s] = 0.0;
        }
        my_point_vars([&](int ps, int k) {
            const int s = pslot0() + 3 * ps + k, fi = sfree[s], v = svid[s];
            double lo = -__builtin_inf(), hi = __builtin_inf();
            if (fi >= 0) { PT[k * pstr + ps] = xs[fi]; lo = P.lo[v]; hi = P.hi[v]; }
            else PT[k * pstr + ps] = P.x[v];
            PT[(3 + k) * pstr + ps] = 0.0;
            PG[k * pstr + ps] = 0.0; PG[(3 + k) * pstr + ps] = 0.0;
            PE[(long long)PT_BND * ps + k] = lo; PE[(long long)PT_BND * ps + 3 + k] = hi;
        });
        // a chunk's box: the largest inner lower bound and the smallest inner upper bound of its blocks (a value strictly
        // inside it is strictly inside every block's bounds); blocks' own bounds only where a trial value leaves it
        {
            const int lane = tid & 63;
            for (int c = first_chunk(); c < chunk_end(); c += chunk_step()) {
                const int ps = 64 * c + lane;
                double bl[3], bh[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    double lo = -__builtin_inf(), hi = __builtin_inf();
                    if (ps < npb) {
                        const int sl = pslot0() + 3 * ps + k;
                        if (sfree[sl] >= 0) { const int v = svid[sl]; lo = P.lo[v]; hi = P.hi[v]; }
                    }
                    bl[k] = wave_max((double)inner_lo32(lo));
                    bh[k] = -wave_max(-(double)inner_hi32(hi));
                }
                if (lane == 0) {
#pragma unroll
                    for (int k = 0; k < 3; ++k) { CBX[8 * c + k] = (float)bl[k]; CBX[8 * c + 4 + k] = (float)bh[k]; }
                }
            }
        }
        __syncthreads();
        // every camera's records at the start (a camera without a free variable keeps them; its direction record is zero)
        for (int c = tid; c < ncb; c += nt) {
            double xc[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) xc[k] = X[PTM_CS * c + ptm_slot_of(k)];
            store_rotation(xc[0], xc[1], xc[2], ROTR + PTM_RS * c);
            BaFwd rot;
            ba_load_rotation(ROTR + PTM_RS * c, rot);
            ba_camera_trial(rot, xc, CTR + PTM_TS * c);
#pragma unroll
            for (int k = 0; k < CAM_TRIAL; ++k) CDR[PTM_TS * c + k] = 0.0;
        }
        __syncthreads();
    }

    // SubfunctionFD::df(p, xi) (reference .cpp:135-157): the full gradient at clamp(p) in one point-major pass.
    // Every lane evaluates its block's factors (forward + adjoint) against the cameras in LDS.  The block's three point
    // entries are the sums of its factors' point partials in slot order, which is factor-list order
    // (src/State.h:157-210): formed in registers, written to the record.  The camera partials are summed in LDS, round by
    // round (file header): in round rr a wave evaluates the rr-th of its (chunk, slot) steps and stages its lanes' nine
    // camera partials at STG[9 lane-of-workgroup ..]; the round's list -- fetched from HBM a round ahead -- names per camera
    // the staging indices that hold a factor of it, and lane (camera, entry) adds them to the camera's gradient entry.
    // With no camera variable free (ROT_CAMFIX) there are no rounds: the pass runs like a trial.
    // Fused with the pass (the records are streamed anyway):
    //   LE   the end of the line search before it (Dlinemethod::linmin, minimize_nrc.h:508-511: xi *= amin, p += xi) -- a block's
    //        new p is formed from its record, is the gradient's point, and goes back with the block's gradient entries;
    //   RED  the sums of Frprmn's tests and of gamma behind it (minimize_nrc.h:658-674), a block's terms when its entries are known.
    __device__ void gradient_to_xi() { double a, b, c; gradient_fused(false, 0.0, false, 0.0, a, b, c); }
    __device__ void gradient_fused(bool LE, double amin, bool RED, double fp, double& test, double& gg, double& dgg) {
        constexpr bool CAMS = ROT != ROT_CAMFIX;
        const long long tg0 = clock();
        const double den = fmax(fabs(fp), 1.0);
        double ra = 0.0, rb = 0.0, rt = 0.0;
        if (LE) { line_end_cameras(amin); __syncthreads(); }
        assign_cameras<AT_LINE, 0>(0.0);
        const int lane = tid & 63, cs = chunk_step();
        const int nd = rl_stride >> 1;   // a round's table in 32-bit words
        const unsigned* rsrc = reinterpret_cast<const unsigned*>(rounds);
        unsigned lv0 = 0u, lv1 = 0u;
        if constexpr (CAMS) {
            for (int s = tid; s < PTM_CS * ncb; s += nt) XI[s] = 0.0;
            for (int k = tid; k < 9; k += nt) STG[9 * (rs * nt) + k] = 0.0;   // (the row that stands for "no factor" in the sums)
            if (nrounds > 0) {
                if (tid < nd) lv0 = rsrc[tid];
                if (tid + nt < nd) lv1 = rsrc[tid + nt];
            }
        }
        // this wave's chunks, as in eval_line: a block of slots, the next chunk's records and the range of the chunk after
        // that are asked for a block / a chunk ahead; a round evaluates ONE slot (slot ks of the block at hand)
        int cu = __builtin_amdgcn_readfirstlane(first_chunk());
        int e = 0, e1 = 0;
        const int cend = chunk_end();
        if (cu < cend) { e = __builtin_amdgcn_readfirstlane(cptr[cu]); e1 = __builtin_amdgcn_readfirstlane(cptr[cu + 1]); }
        bool have = e < e1;
        int cx = cu + cs, ne = 0, ne1 = 0, v0 = 0, v1 = 0;
        PtRecs R = {};
        SlotBlock B = {}, N = {};
        if (have) {
            if (cx < cend) { ne = __builtin_amdgcn_readfirstlane(cptr[cx]); ne1 = __builtin_amdgcn_readfirstlane(cptr[cx + 1]); }
            if (cx + cs < cend) { v0 = cptr[cx + cs]; v1 = cptr[cx + cs + 1]; }
            load_recs<true>(cu, R);
            load_block<CAMS>(e, N);
        }
        bool fresh = true;
        int ks = 0;
        double x[3] = {0.0, 0.0, 0.0}, s0 = 0.0, s1 = 0.0, s2 = 0.0, pn[3] = {0.0, 0.0, 0.0}, go[3] = {0.0, 0.0, 0.0};
        // a block's entries are known: its record (new p, gradient), its terms of the sums
        auto finish_block = [&](int ps, const double (&pnew)[3], const double (&gold)[3], double e0, double e1_, double e2) {
            const int* sf = sfree + pslot0() + 3 * ps;
            const double en[3] = {sf[0] >= 0 ? e0 : 0.0, sf[1] >= 0 ? e1_ : 0.0, sf[2] >= 0 ? e2 : 0.0};
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                if (LE) PT[k * pstr + ps] = pnew[k];
                PT[(3 + k) * pstr + ps] = en[k];
                if (RED) reduce_term(pnew[k], en[k], gold[k], den, ra, rb, rt);
            }
        };
        auto line_point = [&](const PtRecs& Rc, int psc, double (&pnew)[3], double (&xx)[3]) {
#pragma clang fp contract(off)
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const double t = Rc.xi[k] * amin;
                pnew[k] = LE ? Rc.p[k] + t : Rc.p[k];
                xx[k] = pnew[k];
            }
            if (!inside32(Rc, xx)) clamp_exact(psc, xx);
        };
        for (int rr = 0; CAMS ? rr < nrounds : have; ++rr) {
            const long long tr0 = clock();
            double gq[12];
            int cc = -1, row = 0;
            // (rs == 2, cameras only) the block's second slot: evaluated behind the first barrier, straight into its staging row
            int cb = -1, rowb = 0;
            double2 ob = make_double2(0.0, 0.0);
            bool pair_round = false, block_ends_chunk = false;
            if (CAMS && PAIR && rs == 2) {
                pair_round = true;
                if (have) {
                    const bool first = fresh;
                    B = N;
                    if (fresh) {   // a chunk's first block: its blocks' position from the records; the next chunk's records
                        line_point(R, min(64 * cu + lane, npb - 1), pn, x);
#pragma unroll
                        for (int k = 0; k < 3; ++k) go[k] = R.g[k];
                        s0 = s1 = s2 = 0.0;
                        if (ne < ne1) load_recs<true>(cx, R);
                        fresh = false;
                    }
                    if (e + 64 * PTM_BLK < e1) load_block<CAMS>(e + 64 * PTM_BLK, N);
                    else if (ne < ne1) load_block<CAMS>(ne, N);
                    cc = B.c[0]; row = B.r[0];
                    if (cc >= 0) {
                        double v[12];
                        BaFwd t;
                        forward(cc, B.o[0], x, v, t);
                        ba_adjoint(t, v, t.res0, t.res1, gq);
                        s0 = first ? gq[9] : s0 + gq[9]; s1 = first ? gq[10] : s1 + gq[10]; s2 = first ? gq[11] : s2 + gq[11];
                    }
                    if (e + 64 < e1) { cb = B.c[1]; rowb = B.r[1]; ob = B.o[1]; block_ends_chunk = e + 128 >= e1; }
                    else block_ends_chunk = true;
                }
            } else
            if (have) {
                const bool first = fresh;
                if (ks == 0) {   // a block's first slot
                    B = N;
                    if (fresh) {   // ... a chunk's: its blocks' position from the records; the next chunk's records
                        line_point(R, min(64 * cu + lane, npb - 1), pn, x);
#pragma unroll
                        for (int k = 0; k < 3; ++k) go[k] = R.g[k];
                        s0 = s1 = s2 = 0.0;
                        if (ne < ne1) load_recs<true>(cx, R);
                        fresh = false;
                    }
                    if (e + 64 * PTM_BLK < e1) load_block<CAMS>(e + 64 * PTM_BLK, N);
                    else if (ne < ne1) load_block<CAMS>(ne, N);
                }
                cc = B.c[0]; row = B.r[0];
                double2 o = B.o[0];
#pragma unroll
                for (int k = 1; k < PTM_BLK; ++k) {
                    cc = ks == k ? B.c[k] : cc; row = ks == k ? B.r[k] : row;
                    o.x = ks == k ? B.o[k].x : o.x; o.y = ks == k ? B.o[k].y : o.y;
                }
                if (cc >= 0) {
                    double v[12];
                    BaFwd t;
                    forward(cc, o, x, v, t);
                    ba_adjoint(t, v, t.res0, t.res1, gq);
                    s0 = first ? gq[9] : s0 + gq[9]; s1 = first ? gq[10] : s1 + gq[10]; s2 = first ? gq[11] : s2 + gq[11];
                }
                ++ks;
                if (e + 64 * ks >= e1) {   // the chunk's last slot: its blocks' point entries
                    const int ps = 64 * cu + lane;
                    if (ps < npb) finish_block(ps, pn, go, s0, s1, s2);
                    have = ne < ne1;
                    cu = cx; e = ne; e1 = ne1; fresh = true; ks = 0;
                    cx += cs;
                    ne = __builtin_amdgcn_readfirstlane(v0); ne1 = __builtin_amdgcn_readfirstlane(v1);
                    v0 = v1 = 0;
                    if (have && cx + cs < cend) { v0 = cptr[cx + cs]; v1 = cptr[cx + cs + 1]; }
                } else if (ks == PTM_BLK) { e += 64 * PTM_BLK; ks = 0; }
            }
            const long long tr1 = clock();
            if constexpr (CAMS) {
                __syncthreads();   // the sums of the round before have read the staging area
                const long long tr2 = clock();
                if (cc >= 0) {     // (its row: the factor's rank among the round's factors in camera order)
                    double* dst = STG + 9 * row;
#pragma unroll
                    for (int k = 0; k < 9; ++k) dst[k] = gq[k];
                }
                if (pair_round && have) {
                    if (cb >= 0) {   // the block's second slot (a lane with a factor here had one in the first slot too)
                        double v[12], g2[12];
                        BaFwd t;
                        forward(cb, ob, x, v, t);
                        ba_adjoint(t, v, t.res0, t.res1, g2);
                        s0 = s0 + g2[9]; s1 = s1 + g2[10]; s2 = s2 + g2[11];
                        double* dst = STG + 9 * rowb;
#pragma unroll
                        for (int k = 0; k < 9; ++k) dst[k] = g2[k];
                    }
                    if (block_ends_chunk) {   // the chunk's last block: its blocks' point entries
                        const int ps = 64 * cu + lane;
                        if (ps < npb) finish_block(ps, pn, go, s0, s1, s2);
                        have = ne < ne1;
                        cu = cx; e = ne; e1 = ne1; fresh = true;
                        cx += cs;
                        ne = __builtin_amdgcn_readfirstlane(v0); ne1 = __builtin_amdgcn_readfirstlane(v1);
                        v0 = v1 = 0;
                        if (have && cx + cs < cend) { v0 = cptr[cx + cs]; v1 = cptr[cx + cs + 1]; }
                    } else e += 64 * PTM_BLK;
                }
                {   // this round's table into LDS, the next one's on its way
                    unsigned* d = reinterpret_cast<unsigned*>(RL + (rr & 1) * rl_cap);
                    if (tid < nd) d[tid] = lv0;
                    if (tid + nt < nd) d[tid + nt] = lv1;
                    for (int i = tid + 2 * nt; i < nd; i += nt) d[i] = rsrc[(long long)rr * nd + i];
                    if (rr + 1 < nrounds) {
                        const unsigned* nx = rsrc + (long long)(rr + 1) * nd;
                        if (tid < nd) lv0 = nx[tid];
                        if (tid + nt < nd) lv1 = nx[tid + nt];
                    }
                }
                const long long tr3 = clock();
                __syncthreads();
                const long long tr4 = clock();
                // lane (camera c, entry k of its nine): the rows [seg[c], seg[c + 1]) in order, eight loads in flight at a time
                const unsigned short* seg = RL + (rr & 1) * rl_cap;
                for (int j = tid; j < 9 * ncb; j += nt) {
                    const int c = j / 9, k = j - 9 * c;
                    const int s = PTM_CS * c + ptm_slot_of(k);
                    if (SF[s] < 0) continue;
                    const int b = seg[c], en = seg[c + 1];
                    if (b >= en) continue;
                    double sm = XI[s];
                    for (int q0 = b; q0 < en; q0 += 8) {
                        double tv[8];
#pragma unroll
                        for (int t = 0; t < 8; ++t) tv[t] = STG[9 * (q0 + t < en ? q0 + t : rs * nt) + k];
#pragma unroll
                        for (int t = 0; t < 8; ++t) sm += tv[t];
                    }
                    XI[s] = sm;
                }
                // (stamps 15, 16, 18, 19, 11 of the first wave: a round's factor, the wait for the others, staging + table, the wait, the sums)
                tick(15, tr1 - tr0); tick(16, tr2 - tr1); tick(18, tr3 - tr2); tick(19, tr4 - tr3); tick(11, clock() - tr4);
            }
        }
        // blocks no listed factor reads (free variables only; their chunks stand last): zero entries
        for (int c = first_chunk(); c < cend; c += cs) {
            if (cptr[c] < cptr[c + 1]) continue;
            const int ps = 64 * c + lane;
            if (ps < npb) {
                PtRecs Rz;
                load_recs<true>(c, Rz);
                double pz[3], xz[3];
                line_point(Rz, ps, pz, xz);
                finish_block(ps, pz, Rz.g, 0.0, 0.0, 0.0);
            }
        }
        const long long tg1 = clock();
        __syncthreads();
        if constexpr (GROUP && CAMS) {   // the workgroups' partial camera gradients, summed in rank order
            double* buf = xch + (long long)gpar * K * xch_stride;
            double* tot = xch + 2ll * K * xch_stride + (long long)gpar * xch_stride;
            gpar ^= 1;
            for (int s = tid; s < PTM_CS * ncb; s += nt) buf[(long long)r * xch_stride + s] = XI[s];
            GX.barrier_ordered();
            if constexpr (LOCAL) {
                // every workgroup holds its own cameras under its own numbers: a share of the component's camera slots each, a
                // slot's partials from the workgroups that hold the camera (the plan's list, ascending rank: a fixed order), a
                // second barrier, everybody fetches the sums of its own cameras
                double* gt = LC.tot + (long long)(gpar ^ 1) * PTM_CS * LC.ncbg;   // (gpar has been flipped above)
                const int S = PTM_CS * LC.ncbg, per = (S + K - 1) / K;
                for (int gs = r * per + tid; gs < min(S, (r + 1) * per); gs += nt) {
                    const int g = gs / PTM_CS, q = gs - PTM_CS * g;
                    double sm = 0.0;
                    for (int e = LC.cr_ptr[g], e1 = LC.cr_ptr[g + 1], first = 1; e < e1; ++e, first = 0) {
                        const int w = LC.cr[e];
                        const double v = buf[(long long)(w >> 8) * xch_stride + PTM_CS * (w & 255) + q];
                        sm = first ? v : sm + v;
                    }
                    gt[gs] = sm;
                }
                GX.barrier_ordered();
                for (int s = tid; s < PTM_CS * ncb; s += nt)
                    if (SF[s] >= 0) XI[s] = gt[cam_gslot(s)];
            } else if (K <= PTM_MAX_GROUP) {   // a few workgroups: everybody adds everything
                for (int s = tid; s < PTM_CS * ncb; s += nt) {
                    if (SF[s] < 0) continue;
                    double sm = buf[s];
                    for (int q = 1; q < K; ++q) sm += buf[(long long)q * xch_stride + s];
                    XI[s] = sm;
                }
            } else {
                // a wide group (a whole device on one component): K x 10 ncb doubles read by each of K workgroups would be the
                // gradient's largest stream -- every workgroup adds a share of the slots (the same rank order: the same bits),
                // a second barrier, everybody fetches the sums
                const int S = PTM_CS * ncb, per = (S + K - 1) / K;
                for (int s = r * per + tid; s < min(S, (r + 1) * per); s += nt) {
                    double sm = buf[s];
                    for (int q0 = 1; q0 < K; q0 += 8) {
                        double tv[8];
#pragma unroll
                        for (int t = 0; t < 8; ++t) tv[t] = q0 + t < K ? buf[(long long)(q0 + t) * xch_stride + s] : 0.0;
#pragma unroll
                        for (int t = 0; t < 8; ++t)
                            if (q0 + t < K) sm += tv[t];
                    }
                    tot[s] = sm;
                }
                GX.barrier_ordered();
                for (int s = tid; s < S; s += nt)
                    if (SF[s] >= 0) XI[s] = tot[s];
            }
            __syncthreads();
        }
        if (RED) {
            camera_pass<true>([&](double& p, double& xi, double& gv, double&) { reduce_term(p, xi, gv, den, ra, rb, rt); });
            sumk<3>(ra, rb, rt);
            gg = ra; dgg = rb; test = rt;
        }
        tick(4, tg1 - tg0); tick(5, clock() - tg1); tick(10, 1);
    }

    // ---- the CG recurrence: cameras in LDS (every workgroup of a group keeps them), points in their records --
    // A point block's p, xi stand in its record (PT), g and h of the Polak-Ribiere recurrence in a second record (PG), both
    // by block: a pass over the point variables is coalesced loads and stores with no index in between (round 3 kept g and h
    // by free index behind a per-variable table: three dependent round trips per variable, 100 000 cycles a pass).  The
    // entries of a variable that is not free are zero throughout (xi, g, h) and take part as zeros.
    // fn(p, xi, g, h) on every camera slot with a free variable
    // ONCE: only the cameras this workgroup speaks for (a camera's terms count once in a group)
    template <bool ONCE = false, class Fn>
    __device__ __forceinline__ void camera_pass(Fn fn) {
        for (int s = tid; s < PTM_CS * ncb; s += nt)
            if (SF[s] >= 0 && (!ONCE || speaks_for(s))) fn(Pv[s], XI[s], GC[s], HC[s]);
    }
    // ... and on the three variables of every point block of this workgroup; WREC / WGH: the pass changes (p, xi) / (g, h)
    template <bool WREC, bool WGH, class Fn>
    __device__ __forceinline__ void point_pass(Fn fn) {
        my_points([&](int ps) {
            double rc[6], gh[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) { rc[k] = PT[k * pstr + ps]; gh[k] = PG[k * pstr + ps]; }
#pragma unroll
            for (int k = 0; k < 3; ++k) fn(rc[k], rc[3 + k], gh[k], gh[3 + k]);
            if constexpr (WREC) {
#pragma unroll
                for (int k = 0; k < 6; ++k) PT[k * pstr + ps] = rc[k];
            }
            if constexpr (WGH) {
#pragma unroll
                for (int k = 0; k < 6; ++k) PG[k * pstr + ps] = gh[k];
            }
        });
    }
    __device__ void cg_start() {
        auto fn = [&](double&, double& xi, double& gv, double& hv) { const double t = -xi; gv = t; hv = t; xi = t; };
        camera_pass(fn);
        point_pass<true, true>(fn);
        __syncthreads();
    }
    __device__ void line_begin() {
        if (L.vdump != nullptr && lm_count < L.dump_iters) {   // (tests: p and the direction by free index)
            double* d = L.vdump + 2ll * L.dump_iters * f0 + 2ll * lm_count * n;
            for (int s = tid; s < PTM_CS * ncb; s += nt)
                if (SF[s] >= 0 && speaks_for(s)) { d[SF[s]] = Pv[s]; d[n + SF[s]] = XI[s]; }
            my_point_vars([&](int ps, int k) {
                const int fi = sfree[pslot0() + 3 * ps + k];
                if (fi >= 0) { d[fi] = PT[k * pstr + ps]; d[n + fi] = PT[(3 + k) * pstr + ps]; }
            });
        }
        ++lm_count;
    }
    __device__ void line_end_cameras(double amin) {
        camera_pass([&](double& p, double& xi, double&, double&) {
#pragma clang fp contract(off)
            const double t = xi * amin;
            xi = t;
            p = p + t;
        });
    }
    __device__ void line_end(double amin) {
        line_end_cameras(amin);
        point_pass<true, false>([&](double& p, double& xi, double&, double&) {
#pragma clang fp contract(off)
            const double t = xi * amin;
            xi = t;
            p = p + t;
        });
        __syncthreads();
    }
    // the terms of Frprmn's tests and of gamma for one variable (minimize_nrc.h:658-674)
    static __device__ __forceinline__ void reduce_term(double p, double x, double gi, double den, double& a, double& b, double& t) {
#pragma clang fp contract(off)
        t = fmax(t, fabs(x) * fmax(fabs(p), 1.0) / den);
        a = a + gi * gi;
        b = b + (x + gi) * x;
    }
    __device__ void cg_reduce(double fp, double& test, double& gg, double& dgg) {
        const double den = fmax(fabs(fp), 1.0);
        double a = 0.0, b = 0.0, t = 0.0;
        auto fn = [&](double& p, double& xi, double& gv, double&) { reduce_term(p, xi, gv, den, a, b, t); };
        camera_pass<true>(fn);   // (a camera counts once in a group)
        point_pass<false, false>(fn);
        sumk<3>(a, b, t);
        gg = a; dgg = b; test = t;
    }
    __device__ void cg_update(double gam) {
        auto fn = [&](double&, double& xi, double& gv, double& hv) {
#pragma clang fp contract(off)
            const double gn = -xi;
            const double hn = gn + gam * hv;
            gv = gn; hv = hn; xi = hn;
        };
        camera_pass(fn);
        point_pass<true, true>(fn);
        __syncthreads();
    }
    // leave the variables assigned (.cpp:61, :84-86): clamp(p), or clamp(x_init) after the rollback
    __device__ void write_back(bool restore) {
        const double* xs = L.xstart + f0;
        auto out = [&](int fi, double p) {
            // (lo, hi: from the problem -- the two storage classes need not be told apart here)
            const int v = L.free_vid[f0 + fi];
            const double xv = clampd(restore ? xs[fi] : p, P.lo[v], P.hi[v]);
            P.x[v] = xv;
            L.xout[f0 + fi] = xv;
        };
        for (int s = tid; s < PTM_CS * ncb; s += nt)
            if (SF[s] >= 0 && speaks_for(s)) out(SF[s], Pv[s]);
        my_point_vars([&](int ps, int k) {
            const int fi = sfree[pslot0() + 3 * ps + k];
            if (fi >= 0) out(fi, PT[k * pstr + ps]);
        });
    }
};

// LDS of a workgroup (ptm_bytes_for): [7 vectors of 10 ncb_cap camera slots][8 ncb_cap rotation records][2 x 16 ncb_cap trial records][9 (blockDim + 1) staged
// camera partials][two round lists (16-bit)][10 ncb_cap free indices (int)]
template <int ROT, bool GROUP, class ST = SmallCoopState, bool LOCAL = false, bool PAIR = false>
__device__ __forceinline__ PtmEnv<ROT, GROUP, ST, LOCAL, PAIR> ptm_env(const ProblemView& P, const PlanView& L, int comp, double* lds, double (*red)[3][MAX_WAVES],
                                                                  int ncb_cap, int r, int K, ST* st, double* bcast,
                                                                  int poll_delay, double* xch, const PtmGroupArgs* GA = nullptr) {
    const int f0 = L.free_ptr[comp], c0 = L.fac_ptr[comp];
    const int n = L.free_ptr[comp + 1] - f0, m = L.fac_ptr[comp + 1] - c0;
    const int s0 = L.ls_ptr[comp], ns = L.ls_ptr[comp + 1] - s0, ncbg = L.ls_ncb[comp], npb = (ns - PTM_CS * ncbg) / 3;
    // (LOCAL: this workgroup's table -- its number of cameras, its chunk range, then local -> component camera, then the cameras it speaks for)
    PtmLocal LC{ncbg, 0, 0, nullptr, nullptr, nullptr, nullptr, nullptr};
    int ncb = ncbg;
    if constexpr (LOCAL) {
        const int* t = GA->lc + GA->lc_off[r];
        ncb = t[0]; LC.cbeg = t[1]; LC.cend = t[2];
        LC.cmap = t + 4; LC.own = t + 4 + ncb;
        LC.cr_ptr = GA->cr_ptr; LC.cr = GA->cr; LC.tot = GA->tot;
    }
    const int sc = PTM_CS * ncb_cap, nt = (int)blockDim.x;
    double* ROTR = lds + PTM_CAM_VECTORS * sc;
    double* CTR = ROTR + PTM_RS * ncb_cap;
    double* CDR = CTR + PTM_TS * ncb_cap;
    double* STG = CDR + PTM_TS * ncb_cap;
    const int rs = (PAIR && L.pm_round_slots == 2) ? 2 : 1;
    unsigned short* RL = reinterpret_cast<unsigned short*>(STG + 9 * (rs * nt + 1));
    const int rl_cap = ptm_round_stride(ncb_cap);
    int* SF = reinterpret_cast<int*>(reinterpret_cast<char*>(RL) + (((size_t)2 * rl_cap * sizeof(unsigned short) + 7) & ~(size_t)7));
    for (int s = threadIdx.x; s < PTM_CS * ncb; s += blockDim.x) {
        int gs = s;
        if constexpr (LOCAL) { const int c = s / PTM_CS; gs = PTM_CS * LC.cmap[c] + (s - PTM_CS * c); }
        SF[s] = L.ls_free[s0 + gs];
    }
    __syncthreads();
    const int pb0 = L.pm_pt0[comp];
    const int* segs = L.pm_segs + L.pm_sg_off[(long long)comp * K + r];
    const int seg_rows = segs[0];
    segs += 4;
    int sg[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) sg[q] = __builtin_amdgcn_readfirstlane(segs[(q % 3) * seg_rows + (q / 3) * (nt >> 6) + (threadIdx.x >> 6)]);
    const unsigned short* rounds = nullptr;
    int nrounds = 0;
    if constexpr (ROT != ROT_CAMFIX) {
        const long long w = (long long)comp * K + r;
        rounds = L.pm_rounds + L.pm_rd_off[w];
        nrounds = L.pm_rd_n[w];
    }
    return PtmEnv<ROT, GROUP, ST, LOCAL, PAIR>{P, L, comp, n, m, f0, c0, (int)threadIdx.x, nt, nt >> 6,
                              ncb, npb, (npb + 63) / 64,
                              L.ls_vid + s0, L.ls_free + s0,
                              lds, lds + sc, lds + 2 * sc, lds + 3 * sc, lds + 4 * sc, lds + 5 * sc, lds + 6 * sc, ROTR, CTR, CDR,
                              STG, RL, rl_cap, SF,
                              L.pm_rec + (long long)PT_REC * pb0, (long long)npb, L.pm_cbox + 8ll * L.pm_ch0[comp], L.pm_bex + (long long)PT_BND * pb0,
                              L.pm_cptr + L.pm_ch0[comp], L.pm_cam, L.pm_obs, L.pm_grow,
                              rounds, nrounds, ptm_round_stride(ncb), rs,
                              segs, seg_rows, {sg[0], sg[1], sg[2], sg[3], sg[4], sg[5], sg[6], sg[7], sg[8]},
                              L.pm_gh + (long long)PT_REC * pb0, red, 0,
                              L.trace ? L.trace + 4ll * L.trace_cap * comp : nullptr, 0, 0,
                              r, K, GridSyncT<ST>{st, (int)threadIdx.x, K, r, bcast, poll_delay, 0, 0u, false, 0u, {}},
                              xch, PTM_CS * ncb_cap, 0, LC
#ifdef RDIS_COOP_TIMING
                              , {}
#endif
    };
}

template <int THREADS, int ROT>
__global__ void __launch_bounds__(THREADS, (THREADS <= 256 ? 2 : 1))
cgd_ptm_kernel(ProblemView P, PlanView L, int maxiters, double ftol, int ncb_cap) {
    extern __shared__ __attribute__((aligned(16))) double lds_dyn[];
    __shared__ double red[2][3][MAX_WAVES];
    const int comp = L.order[blockIdx.x];
    // (a component without factors never gets here: it has no slot table and stays with solver_wg.hpp)
    constexpr bool PAIR = THREADS <= PTM_PAIR_MAX_THREADS;
    PtmEnv<ROT, false, SmallCoopState, false, PAIR> E = ptm_env<ROT, false, SmallCoopState, false, PAIR>(P, L, comp, lds_dyn, red, ncb_cap, 0, 1, (SmallCoopState*)nullptr, nullptr, 0, nullptr);
    [[maybe_unused]] const long long tk0 = E.clock();
    __shared__ CgdMachine M;
    __shared__ Request Q[2];
    E.init_vectors();
    run_machine(E, M, Q, maxiters, ftol);
    E.write_back(M.rolled_back);
    if (E.tid == 0) {
        L.fret[comp] = M.fret; L.delta[comp] = M.fret - M.finit; L.iters[comp] = M.iter;
        L.status[comp] = M.status(); L.nfeval[comp] = M.nfeval; L.ngeval[comp] = M.ngeval;
        if (L.trace_n) L.trace_n[comp] = E.trn;
#ifdef RDIS_COOP_TIMING
        if (blockIdx.x == 0 && L.timing) {
            E.tmv[7] = E.clock() - tk0;
#pragma unroll
            for (int i = 0; i < 32; ++i) L.timing[i] = RDIS_COOP_TIMING == 2 ? E.wv[i] : E.tmv[i];
        }
#endif
    }
}

// K workgroups per component, the groups of a launch side by side (cooperative launch: every workgroup resident).
// Block b -> group (b / 8K) * 8 + b mod 8, rank (b / 8) mod K: the workgroups of a group share b mod 8, i.e. the XCD
// the dispatcher is observed to give them (a matter of speed only); blocks beyond the last group leave at once.
// WIDE: a group is a large share of the device -- ONE component too large for the register-resident cooperative solver whose
// cameras fit the LDS (what solver_stream.hpp took before round 6: every trial through x[] in global memory, every gradient's
// partials through HBM).  Its K workgroups are consecutive blocks, i.e. dealt round robin over the XCDs; the exchange state is a
// CoopState (every wave an entry: K x waves of them); the partial camera gradients are added by shares (gradient_fused).
template <int THREADS, int ROT, bool WIDE = false, bool LOCAL = false>
__global__ void __launch_bounds__(THREADS, (THREADS <= 256 ? 2 : 1))
cgd_ptmg_kernel(ProblemView P, PlanView L, PtmGroupArgs A, int maxiters, double ftol, int ncb_cap) {
    static_assert(!LOCAL || WIDE, "local camera numbering is for wide groups");
    extern __shared__ __attribute__((aligned(16))) double lds_dyn[];
    __shared__ double red[2][3][MAX_WAVES];
    __shared__ double bcast[8];
    using ST = std::conditional_t<WIDE, WideCoopState, SmallCoopState>;
    const int b = blockIdx.x, K = A.K;
    const int grp = WIDE ? b / K : (b / (8 * K)) * 8 + (b & 7), r = WIDE ? b % K : (b >> 3) % K;
    if (grp >= A.ngroups) return;
    const int comp = L.order[grp];
    constexpr bool PAIR = THREADS <= PTM_PAIR_MAX_THREADS;
    PtmEnv<ROT, true, ST, LOCAL, PAIR> E = ptm_env<ROT, true, ST, LOCAL, PAIR>(P, L, comp, lds_dyn, red, ncb_cap, r, K, reinterpret_cast<ST*>(A.st) + grp, bcast, A.poll_delay,
                                              A.xch ? A.xch + (long long)grp * (2 * K + 2) * PTM_CS * ncb_cap : nullptr, &A);
    [[maybe_unused]] const long long tk0 = E.clock();
    __shared__ CgdMachine M;
    __shared__ Request Q[2];
    E.init_vectors();
    run_machine(E, M, Q, maxiters, ftol);
    E.write_back(M.rolled_back);
    if (E.tid == 0 && r == 0) {
        L.fret[comp] = M.fret; L.delta[comp] = M.fret - M.finit; L.iters[comp] = M.iter;
        L.status[comp] = M.status(); L.nfeval[comp] = M.nfeval; L.ngeval[comp] = M.ngeval;
        if (L.trace_n) L.trace_n[comp] = E.trn;
#ifdef RDIS_COOP_TIMING
        if (grp == 0 && L.timing) {
            E.tmv[7] = E.clock() - tk0;
            E.tmv[28] = E.GX.tm[5]; E.tmv[29] = E.GX.tm[2]; E.tmv[30] = E.GX.tm[3]; E.tmv[31] = E.GX.tm[4];
#pragma unroll
            for (int i = 0; i < 32; ++i) L.timing[i] = E.tmv[i];
        }
#endif
    }
}

// a plan's point-major factor arrays from its listed-order ones: entry e of the point-major order is listed factor jg[e]
// (-1: the slot is empty, the lane's block has fewer factors than its chunk's first)
__global__ void __launch_bounds__(256)
ptm_gather_kernel(int n, const int* __restrict__ jg, const unsigned* __restrict__ fidx, const double2* __restrict__ fobs,
                  short* __restrict__ pcam, double2* __restrict__ pobs) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int j = jg[i];
        pcam[i] = j >= 0 ? (short)(fidx[j] & 0xFFFu) : (short)-1;
        pobs[i] = j >= 0 ? fobs[j] : make_double2(0.0, 0.0);
    }
}

}  // namespace rdis_hip
