// solver_pipe.hpp -- the cooperative solver (solver_coop.hpp) with the control logic and the
// exchange on waves of their own, so that a line search runs as a pipeline instead of a chain.
//
// In solver_coop.hpp an objective evaluation is a chain of four latencies: the factor arithmetic of
// one wave (~2100 cycles), reduction + publication (~450), the exchange between workgroups (store
// lands, load returns, slowest wave: ~4200), a step of the control logic (~2000) -- and the wave
// that sweeps and steps is also a wave that evaluates, so nothing overlaps.  Here a workgroup of
// four waves is
//
//   wave 0   STEPPER: keeps the state machine of minimizer.hpp in registers (nothing of the factor
//            arithmetic competes for them), steps it on the sums it is handed and posts requests in
//            an LDS mailbox.  It touches no global memory.
//   wave 1   COLLECTOR: sweeps the exchange slot the stepper names (all entries of the group, summed
//            in index order), re-arms dead slots, hands the sums back through LDS.
//   waves 2, 3   LANES: a factor per lane, variables per lane / wave exactly as in solver_coop.hpp;
//            they serve the request in the mailbox and publish their partial sums.
//
// ladybug-49-7776 then needs 249 workgroups of 256 instead of 125 -- the device has 256 compute
// units and the solve used half of them.  The three sides are decoupled, which is what makes
// speculation free: with a value+slope request the stepper also posts a CHAIN of guesses at the
// following trial steps (Predictor, minimizer.hpp: Brent's method mostly bisects towards the best
// point, and the step a bisection takes does not depend on the value the pending trial returns).
// The lanes evaluate the request, then the guesses one after the other, each into its own exchange
// slot, looking at the mailbox in between; the collector is sent to the first guess's slot while the
// stepper still works on the reply before it.  When the machine, stepped with that reply, asks for
// exactly the step that was guessed (same bits) its sums are there or on their way, and the lanes
// never stopped working: per trial step the three sides then cost max(arithmetic, sweep, step)
// instead of their sum.  A guess that does not hold costs the arithmetic of lanes that would have
// idled; the machine never sees a guess -- decisions, trace and call counts are those of the
// unspeculated run, and the sums are formed entry by entry in the order solver_coop.hpp uses (same
// bits: entry = the group's lane wave, 64 consecutive factors).
//
// Exchange slots.  Every exchange has a number e, agreed by construction (all steppers take the
// same decisions on the same bits): a request's evaluation gets the next free number E, the
// guesses of its chain E+1 .. E+DEPTH; a hit continues at E+1, a miss jumps to E+DEPTH+1.  Slot e
// lives in buffer e mod 32 of the granule ring (data is the flag, grid_sync.hpp).  A lane wave
// publishes every slot at most once, in increasing order, and waits for its previous stores before
// each publication, so its stores land in order.  After a completed sweep of slot y a collector
// re-arms its workgroup's entries of all slots <= y - DEPTH - 1: the sweep of y completing proves
// that every lane wave has published y, hence that its stepper had posted a request numbered
// >= y - DEPTH, hence consumed every sweep below that.  The collector hands the sums of y to its
// stepper BEFORE it issues those re-arming stores (they are waited for by its next sweep), so what
// is certain when a stepper holds the result of slot q is only that its workgroup's entries are
// re-armed up to q - 2 DEPTH - 2 (the re-arming after the previous completed sweep, which was of a
// slot >= q - DEPTH - 1).  From there the steppers can go on by a miss and a full chain of hits to
// slot q + 3 DEPTH + 2 before this collector must have swept again.  A buffer comes round again
// after 32 >= 5 DEPTH + 4 slots: nobody can poll a slot whose buffer still holds an earlier use
// (with 16 buffers that held only by timing -- a 15k-cycle window against one store latency).
#pragma once
#include "solver_coop.hpp"

namespace rdis_hip {

constexpr int PIPE_NBUF = 32;     // granule buffers (ring over exchange numbers)
constexpr int PIPE_DEPTH = 3;     // guesses in flight behind a request
constexpr int PIPE_ENT = 512;     // entries (lane waves of a group) per buffer: 256 workgroups, one per compute unit
constexpr int PIPE_MAILS = 4;     // request slots in LDS (every post is a request the lanes must act on before the next one can follow)
constexpr int PIPE_THREADS = 256;
constexpr int PIPE_CTRL = 2;      // stepper + collector
constexpr int PIPE_LANES = PIPE_THREADS - 64 * PIPE_CTRL;   // factor lanes per workgroup
constexpr int PIPE_QUIT = 0x7FFFFFFF;
static_assert(PIPE_NBUF >= 5 * PIPE_DEPTH + 4 && (PIPE_NBUF & (PIPE_NBUF - 1)) == 0, "ring too short for the run-ahead");

// same memory as a CoopState (one per concurrent group), cut differently
struct PipeState {
    // granules 0, 1 of an entry (the value and slope of a line-search trial: the exchange that matters)
    // are packed 16 bytes apart, granules 2, 3 live in a second array: a sweep of pairs reads half
    // the cache lines it would read from 32-byte entries
    alignas(32) unsigned long long granule[2][PIPE_NBUF][PIPE_ENT][2];
    unsigned int abort_flag;
    unsigned int pad[15];
};
static_assert(sizeof(PipeState) <= sizeof(CoopState), "PipeState must fit the exchange state the host allocates");

// a request: its head (32 bytes) is posted first -- the lanes start on it at once -- and, with a
// value+slope request, the Predictor (minimizer.hpp) they continue from follows while they evaluate
struct alignas(16) PipeMail {   // 160 bytes: ten 16-byte LDS accesses
    int kind, flags;
    int e;                       // exchange number of the request's first collective operation
    int pad0;
    double a, b;
    // (second part, PipeShared::pred_seq)
    double g_a, g_b, g_x, g_dx, g_ax, g_bx, g_cx;
    int g_need_first, g_known;
    double g_w, g_v, g_dw, g_dv, g_d, g_e, g_uu;
    int g_ph, pad1;
};
static_assert(sizeof(PipeMail) == 160, "mail layout");
constexpr int PIPE_RES = 4;
static_assert(PIPE_RES > PIPE_DEPTH, "result ring");
constexpr int PIPE_RECS = 8;    // guess records the stepper may still want: slots verified + 1 .. verified + DEPTH + 1
struct PipeShared {
    PipeMail mail[PIPE_MAILS];
    alignas(8) int seq;          // number of the latest post (mail[seq % PIPE_MAILS])
    int verified;                // stepper -> lanes: the latest slot the machine has asked for; they stay within DEPTH of it
                                 // (next to seq: a lane wave reads both with one 8-byte access per guess)
    int pred_seq;                // number of the latest post whose predictor part is there too
    // lanes -> stepper: what the chain guesses for slot z: rec_slot[z % 8] = 2 z + 1 and the step in rec_val, or 2 z = no guess
    int rec_slot[PIPE_RECS];
    double rec_val[PIPE_RECS];
    // stepper -> collector: (slot << 2 | number of values), slots only ever increase; sleep before the first look
    int cmd, cmd_delay;
    // collector -> stepper: the sums of slot res_slot[slot % 4] (the collector follows a chain of guesses
    // on its own and can be up to DEPTH slots ahead of what the stepper has consumed)
    int res_slot[PIPE_RES];
    double res[PIPE_RES][3];
    int bar_done;                // the last grid-wide barrier slot that is complete (stepper -> lanes)
    int pub[PIPE_LANES / 64];    // lanes -> collector: the last slot each lane wave of this workgroup has published
    int dead;                    // an exchange gave up
    // the result, for the lanes' write-back
    int status, rolled_back, iter;
    long long nfeval, ngeval;
    double fret, finit;
};
// words of PipeShared that one wave writes and another polls.  The casts name the LDS address
// space: a volatile access through a generic pointer is compiled as a FLAT access at system scope
// followed by a wait for every outstanding memory operation of the wave.
typedef __attribute__((address_space(3))) int lds_i32;
typedef __attribute__((address_space(3))) double lds_f64;
__device__ __forceinline__ int lds_int(const int& x) { return *(const volatile lds_i32*)&x; }
__device__ __forceinline__ void lds_set(int& x, int v) { *(volatile lds_i32*)&x = v; }
typedef __attribute__((address_space(3))) long long lds_i64;
__device__ __forceinline__ long long lds_int_pair(const int& x) { return *(const volatile lds_i64*)&x; }   // x and the int after it (8-byte aligned)
__device__ __forceinline__ double lds_f64_get(const double& x) { return *(const volatile lds_f64*)&x; }
__device__ __forceinline__ void lds_f64_set(double& x, double v) { *(volatile lds_f64*)&x = v; }

struct PipeSync {
    PipeState* st;
    int tid, nwg, wg, nw;        // lane in workgroup, #workgroups, my workgroup, lane waves per workgroup
    int poll_delay;
    int e;                       // next exchange number of the request being served (stepper and lanes agree)
    int rearmed;                 // collector: slots <= rearmed are re-armed
    bool dead;
#ifdef RDIS_COOP_TIMING
    long long tm[COOP_TM];
#endif
    __device__ void tick(int slot, long long dt) {
#ifdef RDIS_COOP_TIMING
        tm[slot] += dt;
#endif
    }
    __device__ gu64* gran(int ex, int k, int w) const { return (gu64*)&st->granule[k >> 1][ex & (PIPE_NBUF - 1)][w][k & 1]; }
    __device__ u64x2 load_pair(int ex, int k, int w) const {
        u64x2 r;
        const gu64* p = gran(ex, k, w);
        asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(r) : "v"(p) : "memory");
        return r;
    }
    __device__ void store_pair(int ex, int k, int w, unsigned long long a, unsigned long long b) const {
        u64x2 t; t.x = a; t.y = b;
        gu64* p = gran(ex, k, w);
        asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(p), "v"(t) : "memory");
    }

    // A lane wave (all 64 lanes) publishes its partial results of exchange ex: sums, the last NMAX maxima.
    template <int N, int NMAX>
    __device__ void publish(int ex, double (&v)[N], int* pub) {
#pragma unroll
        for (int k = 0; k < N; ++k) v[k] = k < N - NMAX ? wave_sum(v[k]) : wave_max(v[k]);
        // this wave's earlier stores -- data handed over by the exchange (SYNC_DRAIN of grid_sync.hpp)
        // and its previous publication -- are complete: a wave's publications land in order
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if ((tid & 63) == 0) {
            const int ent = wg * nw + (tid >> 6) - PIPE_CTRL;
#pragma unroll
            for (int k = 0; k + 1 < N; k += 2) store_pair(ex, k, ent, GridSync::bits_of(v[k]), GridSync::bits_of(v[k + 1]));
            if constexpr (N & 1) GridSync::publish(gran(ex, N - 1, ent), v[N - 1]);
            lds_set(pub[(tid >> 6) - PIPE_CTRL], ex);   // the collector next door: the others' stores are about as far
        }
    }

    // The collector gathers exchange ex: on success v holds the sums (maxima) over all lane waves of
    // the group, entry by entry in index order -- bit-identical in every workgroup -- and what is
    // certainly dead has been re-armed.  It first waits (in LDS) for the lane waves of its own
    // workgroup to publish the slot -- the other workgroups' are about as far, whatever held them
    // up -- and, if it had to wait, sleeps `delay` (x64 cycles: a store needs about that long to land;
    // polling earlier only slows it down) before the first look at memory.
    // Returns false, v untouched, when the stepper has named another slot in the meantime (`cmd`).
    template <int N, int NMAX>
    __device__ bool sweep(int ex, double (&v)[N], int delay, const int& cmd, int cmd_mine, const int* pub) {
        const int lane = tid & 63;
        const int nent = nwg * nw;
        double acc[N];
#pragma unroll
        for (int k = 0; k < N; ++k) acc[k] = 0.0;
        unsigned spins = 0;
        bool ok = !dead;
        const int per = (nent + 63) >> 6;
        constexpr int CH = 8;
        const long long tq0 = coop_clock();
        {
            bool waited = false;
            for (;;) {
                int lo = lds_int(pub[0]);
#pragma unroll
                for (int w = 1; w < PIPE_LANES / 64; ++w) { const int t = lds_int(pub[w]); lo = t < lo ? t : lo; }
                if (lo >= ex || dead) break;
                if (lds_int(cmd) != cmd_mine) return false;
                waited = true;
                if (++spins > COOP_SPIN_LIMIT) break;
                __builtin_amdgcn_s_sleep(1);
            }
            spins = 0;
            if (waited) { tick(26, coop_clock() - tq0); tick(27, 1); }
            if (waited) for (int d = 0; d < delay; d += 8) __builtin_amdgcn_s_sleep(8);
        }
        const long long tq1 = coop_clock();
        for (int j0 = 0; j0 < per && ok; j0 += CH) {
            unsigned long long val[CH][N];
#pragma unroll
            for (int j = 0; j < CH; ++j)
#pragma unroll
                for (int k = 0; k < N; ++k) val[j][k] = COOP_SENTINEL;
            for (;;) {
                tick(6, 1);
                constexpr int NP = N / 2;
                u64x2 pr[CH][NP > 0 ? NP : 1];
#pragma unroll
                for (int j = 0; j < CH; ++j) {
                    const int ww = lane + ((j0 + j) << 6);
                    const int wc = ww < nent ? ww : 0;
#pragma unroll
                    for (int k = 0; k + 1 < N; k += 2) {
                        pr[j][k / 2].x = val[j][k]; pr[j][k / 2].y = val[j][k + 1];
                        if (val[j][k] == COOP_SENTINEL || val[j][k + 1] == COOP_SENTINEL) pr[j][k / 2] = load_pair(ex, k, wc);
                    }
                    if constexpr (N & 1) {
                        if (val[j][N - 1] == COOP_SENTINEL)
                            val[j][N - 1] = __hip_atomic_load(gran(ex, N - 1, wc), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
                if constexpr (NP > 0) {
#pragma unroll
                    for (int j = 0; j < CH; ++j)
#pragma unroll
                        for (int q = 0; q < NP; ++q) asm volatile("" : "+v"(pr[j][q]));
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                    for (int j = 0; j < CH; ++j)
#pragma unroll
                        for (int q = 0; q < NP; ++q) {
                            asm volatile("" : "+v"(pr[j][q]));
                            val[j][2 * q] = pr[j][q].x; val[j][2 * q + 1] = pr[j][q].y;
                        }
                }
                bool here = true;
#pragma unroll
                for (int j = 0; j < CH; ++j)
#pragma unroll
                    for (int k = 0; k < N; ++k) here = here && val[j][k] != COOP_SENTINEL;
                if (__all(here)) break;
                if (lds_int(cmd) != cmd_mine) return false;   // the stepper wants something else
                ++spins;
                if (spins > COOP_SPIN_LIMIT ||
                    ((spins & 255u) == 0u &&
                     __hip_atomic_load((gu32*)&st->abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
                    if (lane == 0) __hip_atomic_store((gu32*)&st->abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok = false;
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                if (lane + ((j0 + j) << 6) < nent) {
#pragma unroll
                    for (int k = 0; k < N; ++k) {
                        const double x = __longlong_as_double(val[j][k]);
                        if (k < N - NMAX) acc[k] += x; else acc[k] = fmax(acc[k], x);
                    }
                }
            }
        }
        if (!__all(ok)) dead = true;
        const long long tq2 = coop_clock();
#pragma unroll
        for (int k = 0; k < N; ++k) v[k] = k < N - NMAX ? wave_sum(acc[k]) : wave_max(acc[k]);
        tick(5, 1); tick(28, tq2 - tq1); tick(29, coop_clock() - tq2);
        return true;
    }
    // after a completed sweep of slot ex (every memory operation of this wave, the previous re-arming
    // stores included, is complete: the sweep's last wait was for vmcnt(0))
    __device__ void rearm_behind(int ex) {
        const int upto = ex - PIPE_DEPTH - 1;
        if ((tid & 63) < nw) {
            const int ent = wg * nw + (tid & 63);
            for (int x = rearmed + 1; x <= upto; ++x) {
#pragma unroll
                for (int k = 0; k < COOP_KP; k += 2) store_pair(x, k, ent, COOP_SENTINEL, COOP_SENTINEL);
            }
        }
        if (upto > rearmed) rearmed = upto;
    }

    // Grid-wide barrier as the lanes see it: their coherent stores issued before it are visible to
    // coherent loads after it.  Takes one exchange number; the stepper's side is pipe_ctrl_barrier.
    __device__ void barrier(PipeShared& S) {
        double z[1] = {0.0};
        publish<1, 0>(e, z, S.pub);
        while (lds_int(S.bar_done) < e) __builtin_amdgcn_s_sleep(1);
        ++e;
        if (lds_int(S.dead) != 0) dead = true;
    }
};

// The lanes' side: factor and variable state in registers, as CoopEnv (solver_coop.hpp) -- the
// arithmetic, the ownership of the CG recurrence and the orders of summation are the same.
struct PipeEnv {
    const ProblemView& P;
    const PlanView& L;
    const CoopArgs& A;
    PipeShared& S;
    int n, m, f0, c0;
    int gt, tid;              // factor lane index in the group (-1 on the two control waves), lane in workgroup
    PipeSync X;
    double* tr;
    int trn, lm_count;
    bool has_fac;
    int fid;
    int gpos[12];             // where this factor's partials go in gfac (variable-major), -1 = not a free variable: constant for the solve
    double base[12], dirv[12], lov[12], hiv[12];
    double ox, oy;
    VarState lv, wv;

    __device__ void trace(int tag, double a, double b, double c) {
        if (tr != nullptr && X.wg == 0 && tid == 0) {
            if (trn < L.trace_cap) { double* r = tr + 4ll * trn; r[0] = (double)tag; r[1] = a; r[2] = b; r[3] = c; }
            ++trn;
        }
    }

    template <bool SLOPE>
    __device__ void eval_line(double a, double& f, double& s) {
        double fj = 0.0, sj = 0.0;
        if (has_fac) {
            double v[12];
            {
#pragma clang fp contract(off)
#pragma unroll
                for (int k = 0; k < 12; ++k) {
                    const double t = a * dirv[k];
                    v[k] = clampd(base[k] + t, lov[k], hiv[k]);
                }
            }
            if constexpr (SLOPE) {
                double g[12];
                fj = ba_eval_grad(v, ox, oy, g);
                double acc = 0.0;
#pragma unroll
                for (int k = 0; k < 12; ++k) acc = __builtin_fma(g[k], dirv[k], acc);   // (fused on purpose, not by the compiler's leave: the tests' CPU restatement mirrors exactly this)
                sj = acc;
            } else {
                fj = ba_eval(v, ox, oy);
            }
        }
        f = fj; s = sj;
    }
    __device__ void load_base(const double* vec) {
        if (has_fac) {
            const int c = P.cam[fid], q = P.pt[fid];
            const int* sl = A.slot_li + 12ll * gt;
#pragma unroll
            for (int k = 0; k < 12; ++k) {
                const int v = k < 9 ? c + k : q + (k - 9);
                const int li = sl[k];
                if (li >= 0) { base[k] = vec[li]; lov[k] = P.lo[v]; hiv[k] = P.hi[v]; }
                else { base[k] = P.x[v]; lov[k] = -__builtin_inf(); hiv[k] = __builtin_inf(); }
                dirv[k] = 0.0;
            }
        }
    }
    __device__ void var_init(VarState& V, int li) {
        V.li = li;
        V.p = V.xi = V.g = V.h = V.xinit = 0.0; V.lo = V.hi = 0.0;
        if (li >= 0) {
            const int vid = L.free_vid[f0 + li];
            V.p = L.xstart[f0 + li]; V.xinit = V.p;
            V.lo = P.lo[vid]; V.hi = P.hi[vid];
        }
    }
    __device__ void init_vectors() {
        var_init(lv, gt >= 0 ? A.lane_var[gt] : -1);
        var_init(wv, gt >= 0 ? A.wave_var[gt >> 6] : -1);
        load_base(L.xstart + f0);
    }
    __device__ void gradient_to_xi() {
        const long long tg0 = coop_clock();
        if (has_fac) {
            double v[12], g[12];
#pragma unroll
            for (int k = 0; k < 12; ++k) v[k] = clampd(base[k], lov[k], hiv[k]);
            ba_eval_grad(v, ox, oy, g);
#pragma unroll
            for (int k = 0; k < 12; ++k)
                if (gpos[k] >= 0) store_f64<true>(L.gfac + gpos[k], g[k]);
        }
        const long long tg1 = coop_clock();
        X.barrier(S);
        const long long tg2 = coop_clock();
        const int* vp = L.v2s_ptr + f0;
        if (lv.li >= 0) lv.xi = run_sum_ordered<true, 32>(L.gfac, vp[lv.li], vp[lv.li + 1]);
        if (wv.li >= 0) wv.xi = wave_sum(run_sum_strided<true>(L.gfac, vp[wv.li], vp[wv.li + 1], tid & 63));
        X.tick(3, tg1 - tg0); X.tick(18, tg2 - tg1); X.tick(30, coop_clock() - tg2); X.tick(31, 1);
    }
    __device__ void publish_xi() {
        if (lv.li >= 0) store_f64<true>(A.xi_glob + lv.li, lv.xi);
        if (wv.li >= 0 && (tid & 63) == 0) store_f64<true>(A.xi_glob + wv.li, wv.xi);
        X.barrier(S);
    }
    __device__ void cg_start() {
        { const double t = -lv.xi; lv.g = t; lv.h = t; lv.xi = t; }
        { const double t = -wv.xi; wv.g = t; wv.h = t; wv.xi = t; }
        publish_xi();
    }
    // FROM_GH: the new direction is formed here, from the -gradient the owners published with their
    // Polak-Ribiere sums and the previous direction, with the owner's own operations (cg_update:
    // h_new = g_new + gamma h): same bits, and no exchange between the direction update and the line's
    // first trial
    template <bool FROM_GH>
    __device__ void line_begin(double gam) {
        if (has_fac) {
            const int* sl = A.slot_li + 12ll * gt;
            int li[12];
#pragma unroll
            for (int k = 0; k < 12; ++k) li[k] = sl[k];
            double d[12];
            if constexpr (FROM_GH) {
                // (the previous h of a variable is the previous direction: what dirv still holds)
                const double* gb = L.ws + 5ll * f0;
#pragma unroll
                for (int k = 0; k < 12; ++k) d[k] = load_f64<true>(gb + (li[k] >= 0 ? li[k] : 0));
                {
#pragma clang fp contract(off)
#pragma unroll
                    for (int k = 0; k < 12; ++k) d[k] = d[k] + gam * dirv[k];
                }
            } else {
#pragma unroll
                for (int k = 0; k < 12; ++k) d[k] = load_f64<true>(A.xi_glob + (li[k] >= 0 ? li[k] : 0));
            }
#pragma unroll
            for (int k = 0; k < 12; ++k) dirv[k] = li[k] >= 0 ? d[k] : 0.0;
        }
        if (L.vdump != nullptr && lm_count < L.dump_iters) {
            double* d = L.vdump + 2ll * L.dump_iters * f0 + 2ll * lm_count * n;
            if (lv.li >= 0) { d[lv.li] = lv.p; d[n + lv.li] = lv.xi; }
            if (wv.li >= 0 && (tid & 63) == 0) { d[wv.li] = wv.p; d[n + wv.li] = wv.xi; }
        }
        ++lm_count;
    }
    __device__ void line_end(double amin) {
#pragma clang fp contract(off)
        { const double t = lv.xi * amin; lv.xi = t; lv.p = lv.p + t; }
        { const double t = wv.xi * amin; wv.xi = t; wv.p = wv.p + t; }
        if (has_fac) {
#pragma unroll
            for (int k = 0; k < 12; ++k) { const double t = dirv[k] * amin; base[k] = base[k] + t; }
        }
    }
    // the lanes' share of the Polak-Ribiere sums (nrc :655-672); the collector gathers them
    __device__ void cg_reduce_publish(double fp) {
        // first what the factor lanes will form the next direction from (line_begin<true>): g_new =
        // -gradient of every variable; the publication below waits for these stores
        {
            double* gb = L.ws + 5ll * f0;
            if (lv.li >= 0) store_f64<true>(gb + lv.li, -lv.xi);
            if (wv.li >= 0 && (tid & 63) == 0) store_f64<true>(gb + wv.li, -wv.xi);
        }
        double a = 0.0, b = 0.0, t = 0.0;
        {
#pragma clang fp contract(off)
            const double den = fmax(fabs(fp), 1.0);
            if (lv.li >= 0) {
                t = fabs(lv.xi) * fmax(fabs(lv.p), 1.0) / den;
                a = lv.g * lv.g;
                b = (lv.xi + lv.g) * lv.xi;
            }
            if (wv.li >= 0 && (tid & 63) == 0) {
                t = fmax(t, fabs(wv.xi) * fmax(fabs(wv.p), 1.0) / den);
                a = a + wv.g * wv.g;
                b = b + (wv.xi + wv.g) * wv.xi;
            }
        }
        double v[3] = {a, b, t};
        X.publish<3, 1>(X.e, v, S.pub);
        ++X.e;
    }
    __device__ void cg_update(double gam) {   // (nothing to publish: line_begin<true>)
#pragma clang fp contract(off)
        { const double gn = -lv.xi; const double hn = gn + gam * lv.h; lv.g = gn; lv.h = hn; lv.xi = hn; }
        { const double gn = -wv.xi; const double hn = gn + gam * wv.h; wv.g = gn; wv.h = hn; wv.xi = hn; }
    }
    __device__ void write_back(const VarState& V, bool restore, bool writer) {
        if (V.li >= 0 && writer) {
            const double xf = clampd(restore ? V.xinit : V.p, V.lo, V.hi);
            P.x[L.free_vid[f0 + V.li]] = xf;
            L.xout[f0 + V.li] = xf;
        }
    }
};

// debug counters (-DRDIS_COOP_TIMING)
// lanes (wave 2 of workgroup 0): 0 factor arithmetic, 2 reduce + publish, 9 waiting for a request, 10 requests evaluated, 11 guesses evaluated
// collector: 1 cycles in completed sweeps, 5 their number, 6 polls, 4 sweeps given up for another slot
// stepper: 7 whole kernel, 8 step + post, 16 / 17 guessed / fresh value+slope steps and 20 / 21 the cycles spent waiting for their sums,
//          19 guesses posted, 12.. cycles serving a request after its post: 12 value, 13 value+slope, 14 gradient (+ reduction),
//          15 value+slope at the start of a line (direction update first); 22.. their counts
// collector also: 26 cycles waiting for the own lanes' publication (27 how often), 28 polling memory, 29 final reduction
// lanes also, per full gradient (31 how many): 3 partials + scatter, 18 the grid-wide barrier behind it, 30 per-variable sums
__device__ __forceinline__ int pipe_slot_owner(int i) {   // 0 stepper, 1 collector, 2 lanes
    return (i == 0 || i == 2 || i == 3 || i == 18 || (i >= 9 && i <= 11) || i == 30 || i == 31) ? 2 : (i == 1 || (i >= 4 && i <= 6) || (i >= 26 && i <= 29)) ? 1 : 0;
}

// the stepper posts a request (lane 0 writes; LDS operations of a wave execute in order, so whoever
// reads the new sequence number reads the new slot)
__device__ __forceinline__ void pipe_post(PipeShared& S, int seq, int kind, int flags, int e, double a, double b, bool writer) {
    if (writer) {
        PipeMail* m = &S.mail[seq & (PIPE_MAILS - 1)];
        m->kind = kind; m->flags = flags; m->e = e; m->pad0 = 0; m->a = a; m->b = b;
        asm volatile("" ::: "memory");
        lds_set(S.seq, seq);
    }
}
__device__ __forceinline__ void pipe_post_predictor(PipeShared& S, int seq, const Predictor& G, bool writer) {
    if (writer) {
        PipeMail* m = &S.mail[seq & (PIPE_MAILS - 1)];
        m->g_ph = G.ph; m->g_need_first = G.need_first ? 1 : 0; m->g_known = G.known ? 1 : 0; m->pad1 = 0;
        m->g_a = G.a; m->g_b = G.b; m->g_x = G.x; m->g_dx = G.dx; m->g_ax = G.ax; m->g_bx = G.bx; m->g_cx = G.cx;
        m->g_w = G.w; m->g_v = G.v; m->g_dw = G.dw; m->g_dv = G.dv; m->g_d = G.d; m->g_e = G.e; m->g_uu = G.uu;
        asm volatile("" ::: "memory");
        lds_set(S.pred_seq, seq);
    }
}
// the stepper sends the collector to slot ex (n values) ...
__device__ __forceinline__ void pipe_command(PipeShared& S, int ex, int n, int delay, bool writer) {
    if (writer) { lds_set(S.cmd_delay, delay); lds_set(S.cmd, (ex << 2) | n); }
}
// ... and waits for its sums
template <int N>
__device__ __forceinline__ void pipe_result(PipeShared& S, int ex, double (&v)[N], bool& dead) {
    while (lds_int(S.res_slot[ex & (PIPE_RES - 1)]) != ex) __builtin_amdgcn_s_sleep(1);
    asm volatile("" ::: "memory");
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] = uniform(lds_f64_get(S.res[ex & (PIPE_RES - 1)][k]));
    if (lds_int(S.dead) != 0) dead = true;
}
__device__ __forceinline__ void pipe_ctrl_barrier(PipeSync& X, PipeShared& S, bool writer) {
    double z[1];
    pipe_command(S, X.e, 1, X.poll_delay, writer);
    pipe_result(S, X.e, z, X.dead);
    if (writer) lds_set(S.bar_done, X.e);
    ++X.e;
}

// ---- wave 0: the control logic -----------------------------------------------------------
__device__ __forceinline__ void pipe_stepper(PipeEnv& E, PipeShared& S, int maxiters, double ftol) {
    CgdMachine M;
    M.init(maxiters, ftol);
    int e_last = 0;             // slot of the last value+slope evaluation
    bool chain_live = false;    // ... and the lanes may be evaluating guesses behind it
    int next_free = 0;          // first exchange number not handed out
    bool swapped = true;        // did the last bracketing go to the other side of the origin?
    double r0 = 0.0, r1 = 0.0, r2 = 0.0;
    int seq = 0;
    const bool writer = E.tid == 0;
    for (;;) {
        const long long ts0 = coop_clock();
        // What do the lanes guess for the slot after the last evaluation?  (They wrote it down before
        // they evaluated that one.)  It is probably what the machine will ask for next: the collector
        // has gone on to that slot by itself and gathers its sums while the machine is stepped.
        bool guessed = false;
        double gval = 0.0;
        int ahead_slot = -1;          // the collector's result words for that slot, read ahead: they travel
        double ahead0 = 0.0, ahead1 = 0.0;   // from LDS while the machine is stepped (slot word first: sums valid if it matches)
        if (chain_live) {
            const int z = e_last + 1;
            int rec = lds_int(S.rec_slot[z & (PIPE_RECS - 1)]);
            ahead_slot = lds_int(S.res_slot[z & (PIPE_RES - 1)]);
            ahead0 = lds_f64_get(S.res[z & (PIPE_RES - 1)][0]);
            ahead1 = lds_f64_get(S.res[z & (PIPE_RES - 1)][1]);
            while ((rec >> 1) != z) { __builtin_amdgcn_s_sleep(1); rec = lds_int(S.rec_slot[z & (PIPE_RECS - 1)]); }
            guessed = (rec & 1) != 0;
            gval = uniform(lds_f64_get(S.rec_val[z & (PIPE_RECS - 1)]));
        }
        asm volatile("; PIPE_STEP_BEGIN");
        Request nq;
        Predictor G;
        G.ph = Predictor::P_STOP; G.need_first = false; G.known = false; G.a = G.b = G.x = G.dx = 0.0; G.ax = G.bx = G.cx = 0.0;
        G.w = G.v = G.dw = G.dv = G.d = G.e = G.uu = 0.0;
        bool was_hot;
        {
            double un, pa, pb, pc;
            int ptag;
            Predictor unused;   // (what hot() would hand on is formed from the machine below, and only for a fresh request)
            was_hot = M.hot(r0, r1, un, ptag, pa, pb, pc, unused);
            if (was_hot) {
                nq = CgdMachine::req(REQ_EVAL, un, RF_SLOPE | RF_LINE);
                nq.pre_tag = ptag; nq.pre_a = pa; nq.pre_b = pb; nq.pre_c = pc;
            } else {
                nq = M.next(r0, r1, r2);
                if (M.st == CgdMachine::S_BR_FC) swapped = M.ax == 1.0;
            }
        }
        if (E.X.dead) {   // an exchange gave up: the restored start is what is returned (CGD .cpp:66-80)
            nq = CgdMachine::req(REQ_DONE);
            M.reason = EXIT_SYNC_TIMEOUT; M.rolled_back = true; M.fret = M.finit;
        }
        const bool slope = nq.kind == REQ_EVAL && (nq.flags & RF_SLOPE) != 0;
        const bool hit = slope && nq.flags == (RF_SLOPE | RF_LINE) && guessed && same_bits(nq.a, gval);
        int e;
        if (hit) {
            // the step asked for is the chain's guess: its slot is the next one, and the lanes, who are at
            // work on the chain already, only need to hear how far the machine has come
            e = e_last + 1;
            if (writer) lds_set(S.verified, e);
        } else {
            e = next_free;
            pipe_post(S, ++seq, nq.kind, nq.flags, e, nq.a, nq.b, writer);
            if (slope) {   // (the lanes are at work on the request; what they continue with follows)
                if (E.A.speculate) G.start(M, swapped);
                pipe_post_predictor(S, seq, G, writer);
            }
        }
        if (E.tr != nullptr) {
            if ((nq.flags & RF_TR_FIRST) && nq.tr_tag != TR_NONE) E.trace(nq.tr_tag, nq.tr_a, nq.tr_b, nq.tr_c);
            if (nq.pre_tag != TR_NONE) E.trace(nq.pre_tag, nq.pre_a, nq.pre_b, nq.pre_c);
            if (!(nq.flags & RF_TR_FIRST) && nq.tr_tag != TR_NONE) E.trace(nq.tr_tag, nq.tr_a, nq.tr_b, nq.tr_c);
        }
        asm volatile("; PIPE_STEP_END");
        const long long ts1 = coop_clock();
        E.X.tick(8, ts1 - ts0);
        if (nq.kind == REQ_DONE) break;
        E.X.e = e;
        chain_live = false;
        [[maybe_unused]] const int tkind = nq.kind == REQ_GRAD ? 2 : !slope ? 0 : (nq.flags & (RF_PRE_START | RF_PRE_UPDATE)) ? 3 : 1;
        switch (nq.kind) {
        case REQ_EVAL:
            if (nq.flags & RF_PRE_START) pipe_ctrl_barrier(E.X, S, writer);   // the lanes' publish_xi (a direction update needs none)
            if (slope) {
                // a fresh step takes the lanes an evaluation: the collector goes there now
                if (!hit) pipe_command(S, E.X.e, 2, E.X.poll_delay, writer);
                double v[2];
                const long long tw0 = coop_clock();
                if (hit && __builtin_amdgcn_readfirstlane(ahead_slot) == E.X.e) {
                    v[0] = uniform(ahead0); v[1] = uniform(ahead1);
                    if (lds_int(S.dead) != 0) E.X.dead = true;
                } else {
                    pipe_result(S, E.X.e, v, E.X.dead);
                }
                if (hit) { E.X.tick(20, coop_clock() - tw0); E.X.tick(16, 1); }
                else { E.X.tick(21, coop_clock() - tw0); E.X.tick(17, 1); }
                r0 = v[0]; r1 = v[1];
                e_last = E.X.e;
                chain_live = true;
                next_free = e_last + PIPE_DEPTH + 1;
                E.trace(TR_FD, nq.a, r0, r1);
            } else {
                double v[1];
                pipe_command(S, E.X.e, 1, E.X.poll_delay, writer);
                pipe_result(S, E.X.e, v, E.X.dead);
                r0 = v[0];
                next_free = E.X.e + 1;
                if (nq.flags & RF_LINE) E.trace(TR_F, nq.a, r0, 0.0);
            }
            break;
        case REQ_GRAD:
            pipe_ctrl_barrier(E.X, S, writer);   // gradient_to_xi: partials -> per-variable sums
            if (nq.flags & RF_POST_REDUCE) {
                double v[3];
                pipe_command(S, E.X.e, 3, E.X.poll_delay, writer);
                pipe_result(S, E.X.e, v, E.X.dead);
                ++E.X.e;
                r1 = v[0]; r2 = v[1]; r0 = v[2];   // (test, gg, dgg) <- (max, sum a, sum b)
            }
            next_free = E.X.e;
            break;
        default:   // REQ_LINE_END: the lanes' own business
            break;
        }
        // (static slots: a computed index would move the counters to scratch memory)
        if (tkind == 0) { E.X.tick(12, coop_clock() - ts1); E.X.tick(22, 1); }
        else if (tkind == 1) { E.X.tick(13, coop_clock() - ts1); E.X.tick(23, 1); }
        else if (tkind == 2) { E.X.tick(14, coop_clock() - ts1); E.X.tick(24, 1); }
        else { E.X.tick(15, coop_clock() - ts1); E.X.tick(25, 1); }
    }
    if (writer) {
        lds_set(S.cmd, PIPE_QUIT);
        // the result, for everybody
        S.status = M.status(); S.rolled_back = M.rolled_back ? 1 : 0; S.iter = M.iter;
        S.nfeval = M.nfeval; S.ngeval = M.ngeval; S.fret = M.fret; S.finit = M.finit;
    }
}

// ---- wave 1: the exchange ----------------------------------------------------------------
// Sweeps the slot the stepper names; after a (value, slope) slot it goes on to the next one by
// itself as long as the lanes' chain has a guess there -- the stepper finds the sums of a guessed
// step waiting -- until the stepper names another slot.
__device__ __forceinline__ void pipe_collector(PipeSync& X, PipeShared& S) {
    int mine = -1;   // the command last acted on
    int next = -1;   // the slot to go on to without being told
    const bool writer = (X.tid & 63) == 0;
    for (;;) {
        const int c = lds_int(S.cmd);
        int ex, n;
        if (c != mine) {
            if (c == PIPE_QUIT) break;
            mine = c;
            ex = c >> 2; n = c & 3;
        } else if (next >= 0) {
            ex = next; n = 2;
        } else {
            __builtin_amdgcn_s_sleep(1);
            continue;
        }
        next = -1;
        const int delay = lds_int(S.cmd_delay);
        double v[3] = {0.0, 0.0, 0.0};
        bool got;
        const long long t0 = coop_clock();
        if (n == 2) { double w[2]; got = X.sweep<2, 0>(ex, w, delay, S.cmd, mine, S.pub); v[0] = w[0]; v[1] = w[1]; }
        else if (n == 3) { got = X.sweep<3, 1>(ex, v, delay, S.cmd, mine, S.pub); }
        else { double w[1]; got = X.sweep<1, 0>(ex, w, delay, S.cmd, mine, S.pub); v[0] = w[0]; }
        if (!got) { X.tick(4, 1); continue; }
        X.tick(1, coop_clock() - t0);
        if (writer) {
            if (X.dead) lds_set(S.dead, 1);
            double* r = S.res[ex & (PIPE_RES - 1)];
            lds_f64_set(r[0], v[0]); lds_f64_set(r[1], v[1]); lds_f64_set(r[2], v[2]);
            lds_set(S.res_slot[ex & (PIPE_RES - 1)], ex);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        X.rearm_behind(ex);
        if (n == 2 && !X.dead) {
            // does the chain go on?  (the lanes write a slot's guess down before they evaluate the slot before it)
            const int z = ex + 1;
            for (;;) {
                const int rec = lds_int(S.rec_slot[z & (PIPE_RECS - 1)]);
                if ((rec >> 1) == z) { if (rec & 1) next = z; break; }
                if (lds_int(S.cmd) != mine) break;
                __builtin_amdgcn_s_sleep(1);
            }
        }
    }
}

// ---- waves 2, 3: the factors and variables -------------------------------------------------
// what the chain guesses for slot z (both lane waves know; one writes it down for the stepper)
__device__ __forceinline__ void pipe_record(PipeShared& S, int z, bool valid, double val, bool writer) {
    if (writer) {
        lds_f64_set(S.rec_val[z & (PIPE_RECS - 1)], val);
        lds_set(S.rec_slot[z & (PIPE_RECS - 1)], 2 * z + (valid ? 1 : 0));
    }
}
__device__ __forceinline__ void pipe_lanes(PipeEnv& E, PipeShared& S) {
    int seen = 0;      // last post acted on
    const bool recorder = E.tid == 64 * PIPE_CTRL;
    for (;;) {
        int s;
        const long long tm0 = coop_clock();
        while ((s = lds_int(S.seq)) == seen) __builtin_amdgcn_s_sleep(1);
        E.X.tick(9, coop_clock() - tm0);
        // Every post is acted on, in order: REQ_LINE_END is the one request the stepper does not wait
        // for, and the post that follows it (REQ_DONE after an ftol exit, or the rollback's evaluation)
        // can be there before a lane wave that was still evaluating a guess has looked.  Jumping to the
        // newest post would skip the move to the line minimum: x of that wave's variables would be the
        // point before the step.  The stepper is never more than two posts ahead (PIPE_MAILS = 4 slots).
        seen = seen + 1;
        asm volatile("" ::: "memory");
        const PipeMail* m = &S.mail[seen & (PIPE_MAILS - 1)];
        const int kind = __builtin_amdgcn_readfirstlane(m->kind);
        const int flags = __builtin_amdgcn_readfirstlane(m->flags);
        E.X.e = __builtin_amdgcn_readfirstlane(m->e);
        const double qa = uniform(m->a);
        if (kind == REQ_DONE) break;
        if (kind == REQ_EVAL) {
            if (flags & RF_PRE_START) E.cg_start();
            if (flags & RF_PRE_UPDATE) E.cg_update(uniform(m->b));
            if (flags & RF_PRE_BEGIN) { if (flags & RF_PRE_UPDATE) E.line_begin<true>(uniform(m->b)); else E.line_begin<false>(0.0); }
            if (flags & RF_SLOPE) {
                const int e0 = E.X.e;
                // the chain of guesses behind this step: the one for a slot is written down before the slot
                // before it is evaluated (for the first: while this step's sums travel), so the stepper
                // finds it when it has those sums
                {
                    double v[2];
                    const long long te0 = coop_clock();
                    E.eval_line<true>(qa, v[0], v[1]);
                    const long long te1 = coop_clock();
                    E.X.publish<2, 0>(e0, v, S.pub);
                    E.X.tick(0, te1 - te0); E.X.tick(2, coop_clock() - te1); E.X.tick(10, 1);

                }
                // the predictor has followed the request's head by now
                while (lds_int(S.pred_seq) < seen) __builtin_amdgcn_s_sleep(1);   // (posts only ever increase)
                asm volatile("" ::: "memory");
                Predictor G;
                G.ph = __builtin_amdgcn_readfirstlane(m->g_ph);
                G.need_first = __builtin_amdgcn_readfirstlane(m->g_need_first) != 0;
                G.a = uniform(m->g_a); G.b = uniform(m->g_b); G.x = uniform(m->g_x); G.dx = uniform(m->g_dx);
                G.ax = uniform(m->g_ax); G.bx = uniform(m->g_bx); G.cx = uniform(m->g_cx);
                G.known = __builtin_amdgcn_readfirstlane(m->g_known) != 0;
                G.w = uniform(m->g_w); G.v = uniform(m->g_v); G.dw = uniform(m->g_dw); G.dv = uniform(m->g_dv);
                G.d = uniform(m->g_d); G.e = uniform(m->g_e); G.uu = uniform(m->g_uu);
                // (written down long before the stepper can have this step's sums: they are still on their way)
                double cval = 0.0;
                bool cvalid = G.next(cval);
                int cz = e0 + 1;           // (cvalid, cval): the guess for slot cz, not evaluated yet
                pipe_record(S, cz, cvalid, cval, recorder);
                // guesses, at most DEPTH slots ahead of what the machine has asked for, until the stepper
                // has something new to say or the chain ends
                for (;;) {
                    const long long sv = lds_int_pair(S.seq);   // (seq, verified)
                    if (!cvalid || (int)sv != seen) break;
                    const int ver = (int)(sv >> 32);
                    if (cz > (ver > e0 ? ver : e0) + PIPE_DEPTH) { __builtin_amdgcn_s_sleep(1); continue; }
                    const double ca = cval;
                    const int z = cz;
                    cvalid = G.next(cval);
                    cz = z + 1;
                    pipe_record(S, cz, cvalid, cval, recorder);
                    double v[2];
                    const long long te0 = coop_clock();
                    E.eval_line<true>(ca, v[0], v[1]);
                    const long long te1 = coop_clock();
                    E.X.publish<2, 0>(z, v, S.pub);
                    E.X.tick(0, te1 - te0); E.X.tick(2, coop_clock() - te1); E.X.tick(11, 1);
                }
            } else {
                if (flags & RF_RESTORE) E.load_base(E.L.xstart + E.f0);
                double v[1], dummy;
                E.eval_line<false>((flags & RF_RESTORE) ? 0.0 : qa, v[0], dummy);
                E.X.publish<1, 0>(E.X.e, v, S.pub);
            }
        } else if (kind == REQ_GRAD) {
            if (flags & RF_PRE_LINE_END) E.line_end(qa);
            E.gradient_to_xi();
            if (flags & RF_POST_REDUCE) E.cg_reduce_publish(uniform(m->b));
        } else if (kind == REQ_LINE_END) {
            E.line_end(qa);
        }
    }
}

template <int THREADS>
__device__ __forceinline__ void pipe_solve(const ProblemView& P, const PlanView& L, const CoopArgs& A, int nwg, int wg,
                                           int maxiters, double ftol) {
    static_assert(THREADS == PIPE_THREADS, "stepper + collector + two lane waves");
    __shared__ PipeShared S;
    [[maybe_unused]] const long long tk0 = coop_clock();
    const int comp = A.comp;
    const int f0 = L.free_ptr[comp], c0 = L.fac_ptr[comp];
    const int n = L.free_ptr[comp + 1] - f0, m = L.fac_ptr[comp + 1] - c0;
    const int tid = (int)threadIdx.x;
    const int gt = tid < 64 * PIPE_CTRL ? -1 : wg * PIPE_LANES + tid - 64 * PIPE_CTRL;

    PipeEnv E{P, L, A, S, n, m, f0, c0, gt, tid,
              PipeSync{(PipeState*)A.st, tid, nwg, wg, PIPE_LANES / 64, A.poll_delay, 0, -1, false
#ifdef RDIS_COOP_TIMING
                       , {}
#endif
              },
              L.trace ? L.trace + 4ll * L.trace_cap * comp : nullptr, 0, 0,
              gt >= 0 && gt < m, 0, {}, {}, {}, {}, {}, 0.0, 0.0, {}, {}};
    if (E.has_fac) {
        E.fid = L.fac_id[c0 + gt];
        const double2 o = P.obs[E.fid];
        E.ox = o.x; E.oy = o.y;
        const int* sp = L.slot_pos + L.slot_base[c0 + gt];
#pragma unroll
        for (int k = 0; k < 12; ++k) E.gpos[k] = sp[k];
    }
    if (tid == 0) { S.seq = 0; S.pred_seq = 0; S.cmd = -1; S.cmd_delay = 0; S.bar_done = -1; S.dead = 0; S.verified = -1; }
    if (tid < PIPE_RECS) S.rec_slot[tid] = -1;
    if (tid < PIPE_RES) S.res_slot[tid] = -1;
    if (tid < PIPE_LANES / 64) S.pub[tid] = -1;
    E.init_vectors();
    __syncthreads();
    if (tid < 64) pipe_stepper(E, S, maxiters, ftol);
    else if (tid < 128) pipe_collector(E.X, S);
    else pipe_lanes(E, S);
    __syncthreads();
    const bool restore = S.rolled_back != 0;
    E.write_back(E.lv, restore, true);
    E.write_back(E.wv, restore, (tid & 63) == 0);
    if (wg == 0 && tid == 0) {
        L.fret[comp] = S.fret; L.delta[comp] = S.fret - S.finit; L.iters[comp] = S.iter;
        L.status[comp] = S.status; L.nfeval[comp] = S.nfeval; L.ngeval[comp] = S.ngeval;
        if (L.trace_n) L.trace_n[comp] = E.trn;
    }
#ifdef RDIS_COOP_TIMING
    if (wg == 0 && (tid & 63) == 0 && tid < 192 && A.timing) {
        if (tid == 0) E.X.tm[7] = coop_clock() - tk0;
        for (int i = 0; i < COOP_TM; ++i) if (pipe_slot_owner(i) == (tid >> 6)) A.timing[i] = E.X.tm[i];
    }
#endif
}

template <int THREADS>
__global__ void __launch_bounds__(THREADS)
cgd_pipe_kernel(ProblemView P, PlanView L, const CoopGroup* __restrict__ groups, const int* __restrict__ wg_group,
                int maxiters, double ftol) {
    const CoopGroup G = groups[wg_group[blockIdx.x]];
    pipe_solve<THREADS>(P, L, G.a, G.nwg, (int)blockIdx.x - G.wg0, maxiters, ftol);
}
template <int THREADS>
__global__ void __launch_bounds__(THREADS)
cgd_pipe_single_kernel(ProblemView P, PlanView L, CoopArgs A, int maxiters, double ftol) {
    pipe_solve<THREADS>(P, L, A, (int)gridDim.x, (int)blockIdx.x, maxiters, ftol);
}

// arms the granules the groups of a launch will use: a block per (group, buffer)
__global__ void __launch_bounds__(256) pipe_arm_kernel(const CoopGroup* __restrict__ groups) {
    const CoopGroup G = groups[blockIdx.x / PIPE_NBUF];
    const int b = blockIdx.x % PIPE_NBUF;
    PipeState* st = (PipeState*)G.a.st;
    const int entries = G.nwg * (PIPE_LANES / 64);
    for (int t = threadIdx.x; t < COOP_KP * entries; t += blockDim.x) {
        const int k = t % COOP_KP, e = t / COOP_KP;
        st->granule[k >> 1][b][e][k & 1] = ~0ull;
    }
    if (b == 0 && threadIdx.x == 0) st->abort_flag = 0u;
}

inline int launch_pipe(hipStream_t stream, int kind, const ProblemView& P, const PlanView& V, const CoopGroup& first,
                       const CoopGroup* groups, const int* wg_group, int ngroups, int total_wg, int maxiters, double ftol) {
    if (kind != KIND_BA) return (int)hipErrorNotSupported;
    pipe_arm_kernel<<<ngroups * PIPE_NBUF, 256, 0, stream>>>(groups);
    hipError_t e0 = hipGetLastError();
    if (e0 != hipSuccess) return (int)e0;
    ProblemView p = P;
    PlanView v = V;
    int mi = maxiters;
    double ft = ftol;
    if (ngroups == 1) {
        CoopArgs a = first.a;
        void* args[] = {&p, &v, &a, &mi, &ft};
        return (int)hipLaunchCooperativeKernel((const void*)cgd_pipe_single_kernel<PIPE_THREADS>, dim3(total_wg), dim3(PIPE_THREADS), args, 0, stream);
    }
    const CoopGroup* gp = groups;
    const int* wp = wg_group;
    void* args[] = {&p, &v, &gp, &wp, &mi, &ft};
    return (int)hipLaunchCooperativeKernel((const void*)cgd_pipe_kernel<PIPE_THREADS>, dim3(total_wg), dim3(PIPE_THREADS), args, 0, stream);
}

inline int pipe_max_workgroups(int num_cus) {
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)cgd_pipe_kernel<PIPE_THREADS>, PIPE_THREADS, 0) != hipSuccess) return 0;
    if (per_cu > 1) per_cu -= 1;
    const long long cap = (long long)per_cu * num_cus;
    const long long lim = PIPE_ENT / (PIPE_LANES / 64);
    return (int)(cap > lim ? lim : cap);
}

}  // namespace rdis_hip
