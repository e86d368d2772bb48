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
//   points    one 72-byte record per point block in HBM: p[3], xi[3] (doubles) and its bounds as six floats rounded
//             INWARD (lo32 >= lo, hi32 <= hi): a trial value strictly inside them is inside the bounds and is its
//             own clamp; only a lane whose value is not (an active bound: rare) fetches the block's exact bounds
//             from a third array and clamps as ever -- the same bits either way, 24 bytes less per block and trial.
//             The blocks stand
//             in order of their number of factors (descending) and are taken 64 at a time, a
//             wave-chunk: a lane takes a block, loads its record ONCE, forms clamp(p + a xi) in
//             registers and evaluates the block's factors one after the other against the cameras in
//             LDS.  The trial point of a point variable is never stored.
//   factors   slot-major per wave-chunk: entry cptr[chunk] + 64 t + lane is the t-th listed factor of the
//             lane's block (camera block 2 B + observation 16 B; camera -1 = the block has fewer
//             factors).  A wave's loads of a slot are 64 neighbours, and their addresses depend on
//             nothing the wave has loaded before: slot t + 1 is fetched while slot t is evaluated.
//             (Round 3, first form: a CSR per point -- point -> range -> camera, observation, every factor
//             a dependent chain of loads in front of ~1000 cycles of arithmetic.  Stamps: half a trial's
//             time was a fixed ~65 000 cycles that did not shrink with the work; three waves per SIMD do
//             not hide thirty exposed round trips.)
//
// HBM bytes per value+slope trial: 24 per point variable + 18 per factor (SURVEY 8d counts 16 + 24:
// x and g once per variable, observation + two indices per factor), no write traffic at all.
//
// The full gradient, once per CG iteration, is a point-major pass like a trial (forward + adjoint; a block's
// point entries summed in registers in slot order = factor-list order, src/State.h:157-210; a factor's nine camera
// partials written to pm_cgq at its position in that order) followed by a pass in camera order (ls_gperm: the factors grouped by
// camera block, whole wave-chunks per camera) that only gathers those partials and adds: see gradient_to_xi.
//
// GROUP = true: K workgroups share a component (cgd_ptmg_kernel, a cooperative launch of several such
// groups side by side).  A launch with fewer components than compute units -- one rank's share of a
// decomposition spread over eight GPUs -- would otherwise leave the rest of the device idle while every
// busy unit works through its component's factors alone.  The camera slots are replicated in every
// workgroup's LDS (each steps the same control logic on the same sums and applies the same vector updates to
// them: nothing about the cameras is ever exchanged but chunk sums of the gradient); the point chunks are
// dealt out round robin (chunk c belongs to workgroup c mod K), and so are the gradient pass's factor
// chunks.  Per trial point the workgroups exchange their partial (value, slope) through the granules of
// grid_sync.hpp (every wave an entry, summed in entry order: the same bits in every workgroup); per CG
// iteration two ordered grid barriers hand over the points' positions (read by the gradient pass of
// whichever workgroup takes a factor) and the per-factor point partials / per-chunk camera sums.
// A group's workgroups sit on one XCD (blockIdx mod 8 is the same for all of them), so what they hand to
// each other through plain stores stays in that XCD's L2.
#pragma once
#include "solver_lds.hpp"
#include "grid_sync.hpp"

namespace rdis_hip {

constexpr int PT_REC = 6;    // doubles per point record: p, xi of the block's three variables
constexpr int PT_BND = 6;    // ... its bounds: lo[3], hi[3] -- floats rounded inward (PB), exact doubles (PE)
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
constexpr int PTM_DOUBLES_PER_SLOT = LDS_DOUBLES_PER_SLOT + 2;   // Pv, XI, LO, HI, X and g, h of the Polak-Ribiere recurrence
constexpr int PTM_MAX_GROUP = 16;  // workgroups per component (SMALL_COOP_ENTRIES / 12 waves, rounded down to a power of two)
constexpr unsigned PTM_NO_FACTOR = 0xFFFFFFFFu;
constexpr int CGQ_REC = 10;  // doubles per camera-partials record (nine + one of padding: five 16-byte accesses)
__host__ __device__ inline size_t ptm_bytes_for(int ncb, int nchunk) {
    return (size_t)ncb * 9 * (PTM_DOUBLES_PER_SLOT * sizeof(double) + sizeof(int)) + (size_t)ncb * (7 * sizeof(double) + 2 * sizeof(int)) +
           (size_t)nchunk * (9 * sizeof(double) + sizeof(int)) + 64;
}

struct PtmGroupArgs {
    SmallCoopState* st;   // one exchange state per group of the launch
    double* cgg;          // [plan's gradient chunks][9] camera partial sums of a chunk (handed over between workgroups)
    int K, ngroups;       // workgroups per component, components of the launch
    int poll_delay;
};

template <int ROT, bool GROUP = false>
struct PtmEnv {
    const ProblemView& P;
    const PlanView& L;
    int comp, n, m, f0, c0, tid, nt, nwaves;
    int ncb, npb, npc;        // camera blocks (9 LDS slots each), point blocks (a record each), wave-chunks of point blocks
    int nchunk;               // wave-chunks of the gradient pass (factors grouped by camera)
    const unsigned* gqw;      // ... per position: slot word,
    const int* gqe;           //     point-major entry,
    const double2* gqobs;     //     observation
    const int* svid;          // variable id of a slot (cameras, then points)
    const int* sfree;         // local free index of a slot, -1 = constant (global copy; the cameras' also in SF)
    double *Pv, *XI, *LO, *HI, *X, *GC, *HC, *ROTR, *CG;   // LDS, cameras (GC, HC: g and h of the recurrence)
    int *CGC, *CST, *CEN, *SF;   // LDS: a gradient chunk's camera; a camera's chunks [CST, CEN); local free index of a camera slot
    double* PT;               // [npb][6] point records: p, xi
    float* PB;                // [npb][6] their bounds rounded inward (lo, hi)
    double* PE;               // [npb][6] ... and exact
    const int* cptr;          // [npc + 1] a point chunk's entries ...
    const short* pcam;        // ... their camera block (-1: none; at most 4095 camera blocks: two bytes a factor and trial)
    const double2* pobs;      // ... their observation
    double* pg;               // ... the three point partials of the last gradient pass (gradient_camera_order)
    const int* gqpos;         // ... the factor's position in the camera-grouped order
    double* cgq;              // [positions of that order][CGQ_REC] a factor's nine camera partials, handed from pass 1 to pass 2
    double *g, *h;            // plan workspace, by free index (point variables)
    double (*red)[3][MAX_WAVES];
    int parity;
    double* tr;
    int trn, lm_count;
    // GROUP: rank r of the K workgroups that share the component, their exchange, the chunk sums in HBM
    int r, K;
    GridSyncT<SmallCoopState> GX;
    double* cgg;
#ifdef RDIS_COOP_TIMING
    long long tmv[32];
#endif

    // the wave-chunks (of point blocks, or of the gradient pass) this wave takes: chunk c belongs to workgroup
    // c mod K, wave (c / K) mod nwaves
    __device__ __forceinline__ int first_chunk() const { return GROUP ? r + K * (tid >> 6) : (tid >> 6); }
    __device__ __forceinline__ int chunk_step() const { return GROUP ? K * nwaves : nwaves; }
    // the point blocks this workgroup owns, a lane each
    template <class Fn>
    __device__ __forceinline__ void my_points(Fn fn) const {
        const int lane = tid & 63;
        for (int c = first_chunk(); c < npc; c += chunk_step()) {
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

    // ---- the cameras' trial point (LDS), as in solver_lds.hpp --------------------------------------
    template <class At>
    __device__ void refresh_records(At at) {
        if constexpr (ROT == ROT_RECORDS) {
            for (int c = nt - 1 - tid; c < ncb; c += nt) {
                const int s = 9 * c;
                if (SF[s] < 0 && SF[s + 1] < 0 && SF[s + 2] < 0) continue;
                double rv[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) rv[k] = SF[s + k] >= 0 ? at(s + k) : Pv[s + k];
                store_rotation(rv[0], rv[1], rv[2], ROTR + 7 * c);
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
    // exact clamp of a block's three values that are not all strictly inside the inward-rounded bounds
    __device__ __forceinline__ void clamp_exact(int ps, double (&x)[3]) {
        const double2* be = reinterpret_cast<const double2*>(PE + (long long)PT_BND * ps);
        const double2 e0 = be[0], e1 = be[1], e2 = be[2];
        x[0] = clampd(x[0], e0.x, e1.y);
        x[1] = clampd(x[1], e0.y, e2.x);
        x[2] = clampd(x[2], e1.x, e2.y);
    }
    __device__ __forceinline__ bool inside32(int ps, const double (&x)[3]) {
        const float2* bq = reinterpret_cast<const float2*>(PB + (long long)PT_BND * ps);
        const float2 b0 = bq[0], b1 = bq[1], b2 = bq[2];
        return x[0] > (double)b0.x && x[0] < (double)b1.y && x[1] > (double)b0.y && x[1] < (double)b2.x &&
               x[2] > (double)b1.x && x[2] < (double)b2.y;
    }
    template <int MODE>
    __device__ __forceinline__ void point_at(int ps, double a, double (&x)[3], double (&d)[3]) {
#pragma clang fp contract(off)
        const double2* rec = reinterpret_cast<const double2*>(PT + (long long)PT_REC * ps);
        const double2 r0 = rec[0], r1 = rec[1], r2 = rec[2];
        const double p[3] = {r0.x, r0.y, r1.x}, xi[3] = {r1.y, r2.x, r2.y};
        if constexpr (MODE == AT_START) {
            const double* be = PE + (long long)PT_BND * ps;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                d[k] = xi[k];
                const int fi = sfree[9 * ncb + 3 * ps + k];
                x[k] = fi >= 0 ? clampd(L.xstart[f0 + fi], be[k], be[3 + k]) : p[k];
            }
        } else {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                d[k] = xi[k];
                const double t = a * xi[k];
                x[k] = p[k] + t;
            }
            if (!inside32(ps, x)) clamp_exact(ps, x);
        }
    }
    // ... at clamp(p), position only (the gradient pass: any workgroup's lanes, a factor each)
    __device__ __forceinline__ void point_position(int ps, double (&x)[3]) {
        const double2* rec = reinterpret_cast<const double2*>(PT + (long long)PT_REC * ps);
        const double2 r0 = rec[0], r1 = rec[1];
        x[0] = r0.x; x[1] = r0.y; x[2] = r1.x;
        if (!inside32(ps, x)) clamp_exact(ps, x);
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
    // This workgroup's share of the sums: its point chunks, a block per lane, the block's factors slot by slot.
    // The next slot's camera and observation are in flight while a slot is evaluated.
    template <bool SLOPE, int MODE>
    __device__ __forceinline__ void eval_partial(double a, double& af, double& as) {
        const int lane = tid & 63;
        for (int c = first_chunk(); c < npc; c += chunk_step()) {
            const int e0 = __builtin_amdgcn_readfirstlane(cptr[c]), e1 = __builtin_amdgcn_readfirstlane(cptr[c + 1]);
            int cn = -1;
            double2 on = make_double2(0.0, 0.0);
            if (e0 < e1) { cn = pcam[e0 + lane]; on = pobs[e0 + lane]; }
            const int ps = 64 * c + lane;
            double x[3] = {0.0, 0.0, 0.0}, dp[3] = {0.0, 0.0, 0.0};
            if (ps < npb) point_at<MODE>(ps, a, x, dp);
            for (int e = e0; e < e1; e += 64) {
                const int cc = cn;
                const double2 o = on;
                if (e + 64 < e1) { cn = pcam[e + 64 + lane]; on = pobs[e + 64 + lane]; }
                if (cc >= 0) {
                    double v[12];
                    BaFwd t;
                    af += forward(cc, o, x, v, t);
                    if constexpr (SLOPE) {
                        double d[12];
#pragma unroll
                        for (int k = 0; k < 9; ++k) d[k] = (ROT == ROT_CAMFIX) ? 0.0 : XI[9 * cc + k];
#pragma unroll
                        for (int k = 0; k < 3; ++k) d[9 + k] = dp[k];
                        as += ba_slope_dir<ROT == ROT_CAMFIX>(t, v, d);
                    }
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
    __device__ bool aborted() const { if constexpr (GROUP) return GX.dead; else return false; }
    // cycle stamps of the launch's first workgroup (build with -DRDIS_COOP_TIMING; rdis_hip_plan_debug_counters):
    // 0 cameras' trial point, 1 this workgroup's factors, 2 sums (GROUP: the exchange), 3 their number (value+slope
    // trials); 4 / 5 gradient before / after the hand-over, 10 their number; 8 / 9 control step / hand-over, 12.. cycles
    // per request kind, 22.. their counts; GROUP: 28 exchanges, 29 publish, 30 sweep, 31 tail of the exchanges
#ifdef RDIS_COOP_TIMING
    __device__ void tick(int slot, long long dt) { tmv[slot] += dt; }
    __device__ long long clock() const { return clock64(); }
#else
    __device__ void tick(int, long long) {}
    __device__ long long clock() const { return 0; }
#endif
    __device__ double eval_value(double a, bool restore) {
        double af = 0.0, as = 0.0, dummy = 0.0;
        if (restore) { assign_cameras<AT_START>(0.0); eval_partial<false, AT_START>(0.0, af, as); }
        else { assign_cameras<AT_LINE>(a); eval_partial<false, AT_LINE>(a, af, as); }
        sumk<1>(af, as, dummy);
        return af;
    }
    __device__ void eval_value_slope(double a, double& f, double& s) {
        double af = 0.0, as = 0.0, dummy = 0.0;
        const long long t0 = clock();
        assign_cameras<AT_LINE>(a);
        const long long t1 = clock();
        eval_partial<true, AT_LINE>(a, af, as);
        const long long t2 = clock();
        sumk<2>(af, as, dummy);
        f = af; s = as;
        tick(0, t1 - t0); tick(1, t2 - t1); tick(2, clock() - t2); tick(3, 1);
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
            XI[s] = 0.0; GC[s] = 0.0; HC[s] = 0.0;
        }
        my_point_vars([&](int ps, int k) {
            const int s = 9 * ncb + 3 * ps + k, fi = sfree[s], v = svid[s];
            double* rec = PT + (long long)PT_REC * ps + k;
            double lo = -__builtin_inf(), hi = __builtin_inf();
            if (fi >= 0) { rec[0] = xs[fi]; lo = P.lo[v]; hi = P.hi[v]; }
            else rec[0] = P.x[v];
            rec[3] = 0.0;
            PE[(long long)PT_BND * ps + k] = lo; PE[(long long)PT_BND * ps + 3 + k] = hi;
            PB[(long long)PT_BND * ps + k] = inner_lo32(lo); PB[(long long)PT_BND * ps + 3 + k] = inner_hi32(hi);
        });
        // the gradient pass's chunks by camera: chunk -> camera, and every camera's (contiguous) range of chunks
        for (int c = tid; c < ncb; c += nt) { CST[c] = 0; CEN[c] = 0; }
        for (int ch = tid; ch < nchunk; ch += nt) CGC[ch] = (int)(gqw[64 * ch] & 0xFFFu);   // (a chunk's first entry always is a factor)
        __syncthreads();
        if constexpr (ROT != ROT_CAMFIX) {
            for (int ch = tid; ch < nchunk; ch += nt) {
                const int c = CGC[ch];
                if (ch == 0 || CGC[ch - 1] != c) CST[c] = ch;
                if (ch == nchunk - 1 || CGC[ch + 1] != c) CEN[c] = ch + 1;
            }
        }
        if constexpr (ROT != ROT_PER_FACTOR)
            for (int c = tid; c < ncb; c += nt) store_rotation(X[9 * c], X[9 * c + 1], X[9 * c + 2], ROTR + 7 * c);
        __syncthreads();
    }

    // SubfunctionFD::df(p, xi) (reference .cpp:135-157): full gradient at clamp(p), in two passes.
    //   1. point-major, like a trial: every lane evaluates its block's factors (forward + adjoint) against the cameras in
    //      LDS -- coalesced, no indirection.  The block's three point entries are the sums of its factors' point partials
    //      in slot order, which is factor-list order (src/State.h:157-210): formed in registers, written to the record.
    //      A factor's nine CAMERA partials go to pm_cgq at the factor's position in the camera-grouped order (an 80-byte
    //      record; the position comes with the factor's camera and observation: a scattered store nobody waits for).
    //   2. camera by camera: the gradient pass's order (ls_gperm: factors grouped by camera block, whole wave-chunks per
    //      camera) with every wave taking a contiguous run of chunks; a lane reads the record at its position -- 64
    //      neighbours per load, no index: positions that are no factor hold zeros --, keeps adding over the chunks of one
    //      camera, and the wave reduces once per camera and run.  The run's sums stand at its last chunk of the camera, zeros at the others, so "a camera's
    //      chunks in order" is the whole sum.
    // (One workgroup per component; workgroups that share a component use gradient_camera_order below.  1000 components of
    // ladybug's size: 175.5 ms a launch with this form, 183.6 with the other.)
    __device__ void gradient_to_xi() {
        if constexpr (GROUP) gradient_camera_order();
        else if (L.pm_grad_form != 0) gradient_camera_order();
        else gradient_two_pass();
    }
    __device__ void gradient_two_pass() {
        const long long tg0 = clock();
        assign_cameras<AT_LINE>(0.0);
        const int lane = tid & 63;
        for (int c = first_chunk(); c < npc; c += chunk_step()) {
            const int e0 = __builtin_amdgcn_readfirstlane(cptr[c]), e1 = __builtin_amdgcn_readfirstlane(cptr[c + 1]);
            int cn = -1, qn = 0;
            double2 on = make_double2(0.0, 0.0);
            if (e0 < e1) { cn = pcam[e0 + lane]; on = pobs[e0 + lane]; qn = gqpos[e0 + lane]; }
            const int ps = 64 * c + lane;
            double x[3] = {0.0, 0.0, 0.0};
            if (ps < npb) point_position(ps, x);
            double s0 = 0.0, s1 = 0.0, s2 = 0.0;
            for (int e = e0; e < e1; e += 64) {
                const int cc = cn, q = qn;
                const double2 o = on;
                if (e + 64 < e1) { cn = pcam[e + 64 + lane]; on = pobs[e + 64 + lane]; qn = gqpos[e + 64 + lane]; }
                if (cc >= 0) {
                    double v[12], gq[12];
                    BaFwd t;
                    forward(cc, o, x, v, t);
                    ba_adjoint(t, v, t.res0, t.res1, gq);
                    s0 = (e == e0) ? gq[9] : s0 + gq[9]; s1 = (e == e0) ? gq[10] : s1 + gq[10]; s2 = (e == e0) ? gq[11] : s2 + gq[11];
                    if constexpr (ROT != ROT_CAMFIX) {
                        double2* dst = reinterpret_cast<double2*>(cgq + (long long)CGQ_REC * q);
                        dst[0] = make_double2(gq[0], gq[1]); dst[1] = make_double2(gq[2], gq[3]);
                        dst[2] = make_double2(gq[4], gq[5]); dst[3] = make_double2(gq[6], gq[7]);
                        dst[4] = make_double2(gq[8], 0.0);
                    }
                }
            }
            if (ps < npb) {
                const int* sf = sfree + 9 * ncb + 3 * ps;
                double* rec = PT + (long long)PT_REC * ps + 3;
                if (sf[0] >= 0) rec[0] = s0;
                if (sf[1] >= 0) rec[1] = s1;
                if (sf[2] >= 0) rec[2] = s2;
            }
        }
        const long long tg1 = clock();
        if constexpr (ROT != ROT_CAMFIX) {
            if constexpr (GROUP) GX.barrier_ordered(); else __syncthreads();   // pm_cgq
            const int gw = (GROUP ? r * nwaves : 0) + (tid >> 6), gwn = (GROUP ? K : 1) * nwaves;
            const int ch0 = (int)((long long)gw * nchunk / gwn), ch1 = (int)((long long)(gw + 1) * nchunk / gwn);
            // (positions that are no factor hold zeros: nothing but the chunk number decides an address; the next chunk's
            // records are in flight while a chunk is added)
            double2 nx[5];
#pragma unroll
            for (int i = 0; i < 5; ++i) nx[i] = make_double2(0.0, 0.0);
            if (ch0 < ch1) {
                const double2* src = reinterpret_cast<const double2*>(cgq + (long long)CGQ_REC * (64 * ch0 + lane));
#pragma unroll
                for (int i = 0; i < 5; ++i) nx[i] = src[i];
            }
            double acc[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) acc[k] = 0.0;
            for (int ch = ch0; ch < ch1; ++ch) {
                double2 cur[5];
#pragma unroll
                for (int i = 0; i < 5; ++i) cur[i] = nx[i];
                if (ch + 1 < ch1) {
                    const double2* src = reinterpret_cast<const double2*>(cgq + (long long)CGQ_REC * (64 * (ch + 1) + lane));
#pragma unroll
                    for (int i = 0; i < 5; ++i) nx[i] = src[i];
                }
                acc[0] += cur[0].x; acc[1] += cur[0].y; acc[2] += cur[1].x; acc[3] += cur[1].y; acc[4] += cur[2].x;
                acc[5] += cur[2].y; acc[6] += cur[3].x; acc[7] += cur[3].y; acc[8] += cur[4].x;
                double* dstc = GROUP ? cgg + 9 * ch : CG + 9 * ch;
                if (ch + 1 == ch1 || CGC[ch + 1] != CGC[ch]) {   // the run's last chunk of this camera (wave-uniform)
                    double cs[9];
#pragma unroll
                    for (int k = 0; k < 9; ++k) { cs[k] = wave_sum(acc[k]); acc[k] = 0.0; }
                    if (lane < 9) dstc[lane] = pick(cs, lane);
                } else {
                    if (lane < 9) dstc[lane] = 0.0;
                }
            }
            if constexpr (GROUP) {
                GX.barrier_ordered();   // cgg[] of every workgroup
                for (int q = tid; q < 9 * nchunk; q += nt) CG[q] = cgg[q];
            }
            __syncthreads();
            for (int s = tid; s < 9 * ncb; s += nt) {
                if (SF[s] < 0) continue;
                const int c = s / 9, k = s - 9 * c;
                const int b = CST[c], e = CEN[c];
                double sm = 0.0;
                for (int q = b; q < e; ++q) sm = (q == b) ? CG[9 * q + k] : sm + CG[9 * q + k];
                XI[s] = sm;
            }
        }
        __syncthreads();
        tick(4, tg1 - tg0); tick(5, clock() - tg1); tick(10, 1);
    }

    // The same gradient in ONE pass in camera order, for workgroups that share a component: there the two-pass form's
    // hand-over of 80 bytes per factor through an ordered grid barrier (a write-back of megabytes of freshly dirtied L2
    // lines) costs more than it saves -- 125 components of ladybug's size as groups of four: 27.5 ms against 25.7.  Every
    // wave takes a contiguous run of the camera-grouped chunks; a factor gathers its point's position from the block's
    // record (read after an ordered barrier: it may belong to another workgroup), its three point partials go to pm_pg at
    // its point-major entry for the owner, its nine camera partials are added lane-wise over the chunks of one camera and
    // reduced once per camera and run (sums at the run's last chunk of the camera, zeros at the others); after a second
    // barrier every workgroup adds up ALL chunks of a camera in chunk order, and every point variable its block's
    // entries slot by slot (entries that are no factor hold zeros, set once at plan creation).
    __device__ void gradient_camera_order() {
        const long long tg0 = clock();
        assign_cameras<AT_LINE>(0.0);
        if constexpr (GROUP) GX.barrier_ordered();   // the point records as line_end / init_vectors left them
        const int lane = tid & 63;
        const int gw = (GROUP ? r * nwaves : 0) + (tid >> 6), gwn = (GROUP ? K : 1) * nwaves;
        const int ch0 = (int)((long long)gw * nchunk / gwn), ch1 = (int)((long long)(gw + 1) * nchunk / gwn);
        // two chunks ahead: the slot word; one chunk ahead: entry, observation and the point's position (its address
        // comes from the slot word) -- a factor's gathers are in flight while the chunk before it is evaluated
        unsigned wn = PTM_NO_FACTOR, wnn = PTM_NO_FACTOR;
        int en = 0;
        double2 on = make_double2(0.0, 0.0);
        double xn[3] = {0.0, 0.0, 0.0};
        if (ch0 < ch1) {
            wn = gqw[64 * ch0 + lane]; en = gqe[64 * ch0 + lane]; on = gqobs[64 * ch0 + lane];
            if (ch0 + 1 < ch1) wnn = gqw[64 * (ch0 + 1) + lane];
            if (wn != PTM_NO_FACTOR) point_position((int)(wn >> 12), xn);
        }
        double acc[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) acc[k] = 0.0;
        for (int ch = ch0; ch < ch1; ++ch) {
            const unsigned w = wn;
            const int e = en;
            const double2 o = on;
            const double x[3] = {xn[0], xn[1], xn[2]};
            if (ch + 1 < ch1) {
                wn = wnn; en = gqe[64 * (ch + 1) + lane]; on = gqobs[64 * (ch + 1) + lane];
                if (ch + 2 < ch1) wnn = gqw[64 * (ch + 2) + lane];
                if (wn != PTM_NO_FACTOR) point_position((int)(wn >> 12), xn);
            }
            if (w != PTM_NO_FACTOR) {
                double v[12], gq[12];
                BaFwd t;
                forward((int)(w & 0xFFFu), o, x, v, t);
                ba_adjoint(t, v, t.res0, t.res1, gq);
                double* dst = pg + 3ll * e;
                dst[0] = gq[9]; dst[1] = gq[10]; dst[2] = gq[11];
                if constexpr (ROT != ROT_CAMFIX) {
#pragma unroll
                    for (int k = 0; k < 9; ++k) acc[k] += gq[k];
                }
            }
            if constexpr (ROT != ROT_CAMFIX) {
                double* dstc = GROUP ? cgg + 9 * ch : CG + 9 * ch;
                if (ch + 1 == ch1 || CGC[ch + 1] != CGC[ch]) {   // the run's last chunk of this camera (wave-uniform)
                    double cs[9];
#pragma unroll
                    for (int k = 0; k < 9; ++k) { cs[k] = wave_sum(acc[k]); acc[k] = 0.0; }
                    if (lane < 9) dstc[lane] = pick(cs, lane);
                } else {
                    if (lane < 9) dstc[lane] = 0.0;
                }
            }
        }
        const long long tg1 = clock();
        if constexpr (GROUP) {
            GX.barrier_ordered();   // pm_pg and cgg[] of every workgroup
            if constexpr (ROT != ROT_CAMFIX)
                for (int q = tid; q < 9 * nchunk; q += nt) CG[q] = cgg[q];
        }
        __syncthreads();
        if constexpr (ROT != ROT_CAMFIX) {
            for (int s = tid; s < 9 * ncb; s += nt) {
                if (SF[s] < 0) continue;
                const int c = s / 9, k = s - 9 * c;
                const int b = CST[c], e = CEN[c];
                double sm = 0.0;
                for (int q = b; q < e; ++q) sm = (q == b) ? CG[9 * q + k] : sm + CG[9 * q + k];
                XI[s] = sm;
            }
        }
        for (int c = first_chunk(); c < npc; c += chunk_step()) {
            const int ps = 64 * c + lane;
            const int e0 = __builtin_amdgcn_readfirstlane(cptr[c]), e1 = __builtin_amdgcn_readfirstlane(cptr[c + 1]);
            double s0 = 0.0, s1 = 0.0, s2 = 0.0;
            for (int e = e0; e < e1; e += 64) {
                const double* src = pg + 3ll * (e + lane);
                const double a0 = src[0], a1 = src[1], a2 = src[2];
                s0 = (e == e0) ? a0 : s0 + a0; s1 = (e == e0) ? a1 : s1 + a1; s2 = (e == e0) ? a2 : s2 + a2;
            }
            if (ps < npb) {
                const int* sf = sfree + 9 * ncb + 3 * ps;
                double* rec = PT + (long long)PT_REC * ps + 3;
                if (sf[0] >= 0) rec[0] = s0;
                if (sf[1] >= 0) rec[1] = s1;
                if (sf[2] >= 0) rec[2] = s2;
            }
        }
        __syncthreads();
        tick(4, tg1 - tg0); tick(5, clock() - tg1); tick(10, 1);
    }

    // ---- the CG recurrence: cameras in LDS (every workgroup of a group keeps them), points in their records --
    // fn(free index, p, xi, g, h) for the free variables this workgroup updates; cameras = false: the point
    // variables only (reductions of a group's workgroups other than the first: a camera counts once)
    template <class Fn>
    __device__ __forceinline__ void for_free(Fn fn, bool cameras = true) {
        if (cameras)
            for (int s = tid; s < 9 * ncb; s += nt)
                if (SF[s] >= 0) fn(SF[s], Pv[s], XI[s], GC[s], HC[s]);
        my_point_vars([&](int ps, int k) {
            const int fi = sfree[9 * ncb + 3 * ps + k];
            if (fi < 0) return;
            double* rec = PT + (long long)PT_REC * ps + k;
            fn(fi, rec[0], rec[3], g[fi], h[fi]);
        });
    }
    __device__ void cg_start() {
        for_free([&](int, double&, double& xi, double& gv, double& hv) { const double t = -xi; gv = t; hv = t; xi = t; });
        __syncthreads();
    }
    __device__ void line_begin() {
        if (L.vdump != nullptr && lm_count < L.dump_iters) {
            double* d = L.vdump + 2ll * L.dump_iters * f0 + 2ll * lm_count * n;
            for_free([&](int fi, double& p, double& xi, double&, double&) { d[fi] = p; d[n + fi] = xi; }, !GROUP || r == 0);
        }
        ++lm_count;
    }
    __device__ void line_end(double amin) {
        for_free([&](int, double& p, double& xi, double&, double&) {
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
        for_free([&](int, double& p, double& xi, double& gv, double&) {
#pragma clang fp contract(off)
            const double x = xi, gi = gv;
            t = fmax(t, fabs(x) * fmax(fabs(p), 1.0) / den);
            a = a + gi * gi;
            b = b + (x + gi) * x;
        }, !GROUP || r == 0);
        sumk<3>(a, b, t);
        gg = a; dgg = b; test = t;
    }
    __device__ void cg_update(double gam) {
        for_free([&](int, double&, double& xi, double& gv, double& hv) {
#pragma clang fp contract(off)
            const double gn = -xi;
            const double hn = gn + gam * hv;
            gv = gn; hv = hn; xi = hn;
        });
        __syncthreads();
    }
    // leave the variables assigned (.cpp:61, :84-86): clamp(p), or clamp(x_init) after the rollback
    __device__ void write_back(bool restore) {
        const double* xs = L.xstart + f0;
        for_free([&](int fi, double& p, double&, double&, double&) {
            // (lo, hi: from the problem -- the two storage classes need not be told apart here)
            const int v = L.free_vid[f0 + fi];
            const double xv = clampd(restore ? xs[fi] : p, P.lo[v], P.hi[v]);
            P.x[v] = xv;
            L.xout[f0 + fi] = xv;
        }, !GROUP || r == 0);
    }
};

// LDS of a workgroup: [7 vectors of 9 ncb_cap camera slots][7 ncb_cap rotation records][9 chunk_cap chunk sums]
// [chunk_cap chunk cameras][2 ncb_cap chunk ranges][9 ncb_cap free indices] (the last three int)
template <int ROT, bool GROUP>
__device__ __forceinline__ PtmEnv<ROT, GROUP> ptm_env(const ProblemView& P, const PlanView& L, int comp, double* lds, double (*red)[3][MAX_WAVES],
                                                       int ncb_cap, int chunk_cap, int r, int K, SmallCoopState* st, double* bcast,
                                                       int poll_delay, double* cgg_base) {
    const int f0 = L.free_ptr[comp], c0 = L.fac_ptr[comp];
    const int n = L.free_ptr[comp + 1] - f0, m = L.fac_ptr[comp + 1] - c0;
    const int s0 = L.ls_ptr[comp], ns = L.ls_ptr[comp + 1] - s0, ncb = L.ls_ncb[comp], npb = (ns - 9 * ncb) / 3;
    const int sc = 9 * ncb_cap;
    double* CG = lds + PTM_DOUBLES_PER_SLOT * sc + 7 * ncb_cap;
    int* CGC = (int*)(CG + 9 * chunk_cap);
    int* CST = CGC + chunk_cap;
    int* CEN = CST + ncb_cap;
    int* SF = CEN + ncb_cap;
    for (int s = threadIdx.x; s < 9 * ncb; s += blockDim.x) SF[s] = L.ls_free[s0 + s];
    __syncthreads();
    double* ws = L.ws + 5ll * f0;
    const int pb0 = L.pm_pt0[comp];
    const long long q0 = 64ll * L.ls_gptr[comp];
    return PtmEnv<ROT, GROUP>{P, L, comp, n, m, f0, c0, (int)threadIdx.x, (int)blockDim.x, (int)(blockDim.x >> 6),
                              ncb, npb, (npb + 63) / 64, L.ls_gptr[comp + 1] - L.ls_gptr[comp],
                              L.pm_gqw + q0, L.pm_gqe + q0, L.pm_gqobs + q0,
                              L.ls_vid + s0, L.ls_free + s0,
                              lds, lds + sc, lds + 2 * sc, lds + 3 * sc, lds + 4 * sc, lds + 5 * sc, lds + 6 * sc,
                              lds + PTM_DOUBLES_PER_SLOT * sc, CG, CGC, CST, CEN, SF,
                              L.pm_rec + (long long)PT_REC * pb0, L.pm_bnd + (long long)PT_BND * pb0, L.pm_bex + (long long)PT_BND * pb0, L.pm_cptr + L.pm_ch0[comp], L.pm_cam, L.pm_obs, L.pm_pg, L.pm_gqpos, L.pm_cgq + (long long)CGQ_REC * q0,
                              ws + 2ll * n, ws + 3ll * n, red, 0,
                              L.trace ? L.trace + 4ll * L.trace_cap * comp : nullptr, 0, 0,
                              r, K, GridSyncT<SmallCoopState>{st, (int)threadIdx.x, K, r, bcast, poll_delay, 0, 0u, false, 0u, {}},
                              cgg_base ? cgg_base + 9ll * L.ls_gptr[comp] : nullptr
#ifdef RDIS_COOP_TIMING
                              , {}
#endif
    };
}

template <int THREADS, int ROT>
__global__ void __launch_bounds__(THREADS, (THREADS <= 256 ? 2 : 1))
cgd_ptm_kernel(ProblemView P, PlanView L, int maxiters, double ftol, int ncb_cap, int chunk_cap) {
    extern __shared__ double lds_dyn[];
    __shared__ double red[2][3][MAX_WAVES];
    const int comp = L.order[blockIdx.x];
    // (a component without factors never gets here: it has no slot table and stays with solver_wg.hpp)
    PtmEnv<ROT, false> E = ptm_env<ROT, false>(P, L, comp, lds_dyn, red, ncb_cap, chunk_cap, 0, 1, nullptr, nullptr, 0, nullptr);
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
            for (int i = 0; i < 32; ++i) L.timing[i] = E.tmv[i];
        }
#endif
    }
}

// K workgroups per component, the groups of a launch side by side (cooperative launch: every workgroup resident).
// Block b -> group (b / 8K) * 8 + b mod 8, rank (b / 8) mod K: the workgroups of a group share b mod 8, i.e. the XCD
// the dispatcher is observed to give them (a matter of speed only); blocks beyond the last group leave at once.
template <int THREADS, int ROT>
__global__ void __launch_bounds__(THREADS, (THREADS <= 256 ? 2 : 1))
cgd_ptmg_kernel(ProblemView P, PlanView L, PtmGroupArgs A, int maxiters, double ftol, int ncb_cap, int chunk_cap) {
    extern __shared__ double lds_dyn[];
    __shared__ double red[2][3][MAX_WAVES];
    __shared__ double bcast[8];
    const int b = blockIdx.x, K = A.K;
    const int grp = (b / (8 * K)) * 8 + (b & 7), r = (b >> 3) % K;
    if (grp >= A.ngroups) return;
    const int comp = L.order[grp];
    PtmEnv<ROT, true> E = ptm_env<ROT, true>(P, L, comp, lds_dyn, red, ncb_cap, chunk_cap, r, K, A.st + grp, bcast, A.poll_delay, A.cgg);
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
// ... and the gradient pass's arrays: position q of a component's ls_gperm -> slot word, point-major entry, observation
__global__ void __launch_bounds__(256)
ptm_gather_gradient_kernel(PlanView L, const int* __restrict__ eof, unsigned* __restrict__ gqw, int* __restrict__ gqe, double2* __restrict__ gqobs,
                           int* __restrict__ gqpos) {
    for (int comp = blockIdx.x; comp < L.ncomp; comp += gridDim.x) {
        const int c0 = L.fac_ptr[comp];
        const long long q0 = 64ll * L.ls_gptr[comp], q1 = 64ll * L.ls_gptr[comp + 1];
        for (long long q = q0 + threadIdx.x; q < q1; q += blockDim.x) {
            const int jl = L.ls_gperm[q];
            const int j = jl >= 0 ? c0 + jl : -1;
            gqw[q] = j >= 0 ? L.ls_fidx[j] : PTM_NO_FACTOR;
            gqe[q] = j >= 0 ? eof[j] : 0;
            gqobs[q] = j >= 0 ? L.ls_obs[j] : make_double2(0.0, 0.0);
            if (j >= 0 && eof[j] >= 0) gqpos[eof[j]] = (int)(q - q0);   // (position within the component's order)
        }
    }
}

}  // namespace rdis_hip
