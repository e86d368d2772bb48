// solver_pipe.hpp -- the cooperative solver (solver_coop.hpp) with the control logic on a wave of
// its own, so that a line search runs as a pipeline instead of a chain.
//
// In solver_coop.hpp an objective evaluation is a chain of four latencies: the factor arithmetic of
// one wave (~2100 cycles), reduction + publication (~450), the exchange between workgroups (store
// lands, load returns, slowest wave: ~4200), a step of the control logic (~2000) -- and the wave
// that sweeps and steps is also a wave that evaluates, so nothing overlaps.  Here a workgroup is
//
//   wave 0        CONTROL: owns no factor and no variable.  It keeps the state machine of
//                 minimizer.hpp in REGISTERS (nothing of the factor arithmetic competes for them),
//                 sweeps the exchange, steps the machine and posts requests in an LDS mailbox.
//   waves 1..     LANES: a factor per lane, variables per lane / wave exactly as in solver_coop.hpp;
//                 they serve the request in the mailbox and publish their partial sums.
//
// ladybug-49-7776 then needs 166 workgroups of 256 instead of 125 -- the device has 256 compute
// units and the solve used half of them.  The two sides are decoupled, which is what makes
// speculation free: with a value+slope request the control wave also posts a CHAIN of guesses at the
// following trial steps (Predictor, minimizer.hpp: Brent's method mostly bisects towards the best
// point, and the step a bisection takes does not depend on the value the pending trial returns).
// The lanes evaluate the request, then the guesses one after the other, each into its own exchange
// slot, looking at the mailbox in between.  When the machine, stepped with the reply, asks for
// exactly the step that was guessed (same bits) the reply is already on its way or there: the
// control wave sweeps that slot at once and the lanes never stopped working.  A guess that does not
// hold costs nothing but the arithmetic of lanes that would have idled; the machine never sees a
// guess -- decisions, trace and call counts are those of the unspeculated run, and the sums are
// formed entry by entry in the order solver_coop.hpp uses (same bits).
//
// Exchange slots.  Every exchange has a number e, agreed by construction (all control waves take
// the same decisions on the same bits): a request's evaluation gets the next free number E, the
// guesses of its chain E+1 .. E+DEPTH; a hit continues at E+1, a miss jumps to E+DEPTH+1.  Slot e
// lives in buffer e mod 16 of the granule ring (data is the flag, grid_sync.hpp).  A lane wave
// publishes every slot at most once, in increasing order, and waits for its previous stores before
// each publication, so its stores land in order.  After its sweep of slot y a control wave re-arms
// its workgroup's entries of all slots <= y - DEPTH - 1: the sweep of y completing proves that every
// lane wave has published y, hence that its control wave had posted a request numbered >= y - DEPTH,
// hence finished every sweep below that.  Those re-arming stores are waited for by the next sweep
// (before the next request is posted); a buffer comes round again after 16 >= 4 DEPTH + 3 slots,
// which is more than the lanes can be ahead of that point.
#pragma once
#include "solver_coop.hpp"

namespace rdis_hip {

constexpr int PIPE_NBUF = 16;     // granule buffers (ring over exchange numbers)
constexpr int PIPE_DEPTH = 3;     // guesses in flight behind a request
constexpr int PIPE_ENT = 1024;    // entries (lane waves of a group) per buffer
constexpr int PIPE_MAILS = 8;     // request slots in LDS: the control wave posts at most 2 (DEPTH + 1) - 1 < 8 times before a lane wave must act
constexpr int PIPE_THREADS = 256;
constexpr int PIPE_LANES = PIPE_THREADS - 64;   // factor lanes per workgroup
static_assert(PIPE_NBUF >= 4 * PIPE_DEPTH + 3 && (PIPE_NBUF & (PIPE_NBUF - 1)) == 0, "ring too short for the run-ahead");

// same memory as a CoopState (one per concurrent group), cut differently
struct PipeState {
    alignas(32) unsigned long long granule[PIPE_NBUF][PIPE_ENT][COOP_KP];
    unsigned int abort_flag;
    unsigned int pad[15];
};
static_assert(sizeof(PipeState) <= sizeof(CoopState), "PipeState must fit the exchange state the host allocates");

struct alignas(16) PipeMail {   // 64 bytes: four 16-byte LDS accesses
    int kind, flags;
    int e;                       // exchange number of the request's first collective operation
    int ng;                      // guesses: guess[j] belongs to slot (slot of the request's evaluation) + 1 + j
    double a, b;
    double guess[PIPE_DEPTH];
    double pad;
};
constexpr int PIPE_PRE = PIPE_ENT / 64;   // 16-byte pairs per lane a control wave can fetch ahead
struct PipeShared {
    // the next slot's (value, slope) pairs fetched ahead by the control wave: [chunk][lane]
    alignas(16) unsigned long long pre[PIPE_PRE][64][2];
    PipeMail mail[PIPE_MAILS];
    int seq;                     // number of the latest post (mail[seq % PIPE_MAILS])
    int dead;                    // an exchange gave up (set by the control wave before a workgroup barrier)
    // the result, for the lanes' write-back
    int status, rolled_back, iter;
    long long nfeval, ngeval;
    double fret, finit;
};

struct PipeSync {
    PipeState* st;
    int tid, nwg, wg, nw;        // lane in workgroup, #workgroups, my workgroup, lane waves per workgroup
    int poll_delay;
    int e;                       // next exchange number of the request being served (all waves agree)
    int rearmed;                 // control wave: slots <= rearmed are re-armed
    bool dead;
#ifdef RDIS_COOP_TIMING
    long long tm[COOP_TM];
#endif
    __device__ bool ctrl() const { return tid < 64; }
    __device__ void tick(int slot, long long dt) {
#ifdef RDIS_COOP_TIMING
        tm[slot] += dt;
#endif
    }
    __device__ gu64* gran(int ex, int k, int w) const { return (gu64*)&st->granule[ex & (PIPE_NBUF - 1)][w][k]; }
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
    __device__ void publish(int ex, double (&v)[N]) {
#pragma unroll
        for (int k = 0; k < N; ++k) v[k] = k < N - NMAX ? wave_sum(v[k]) : wave_max(v[k]);
        // this wave's earlier stores -- data handed over by the exchange (SYNC_DRAIN of grid_sync.hpp)
        // and its previous publication -- are complete: a wave's publications land in order
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if ((tid & 63) == 0) {
            const int ent = wg * nw + (tid >> 6) - 1;
#pragma unroll
            for (int k = 0; k + 1 < N; k += 2) store_pair(ex, k, ent, GridSync::bits_of(v[k]), GridSync::bits_of(v[k + 1]));
            if constexpr (N & 1) GridSync::publish(gran(ex, N - 1, ent), v[N - 1]);
        }
    }

    // The control wave collects exchange ex: on return v holds the sums (maxima) over all lane waves
    // of the group, entry by entry in index order -- bit-identical in every workgroup.  `delay`
    // (x64 cycles) is slept before the first look.  Then re-arms what is certainly dead.
    // The control wave asks for the (value, slope) pairs of slot ex ahead of time: straight into LDS
    // (global_load_lds: no register is tied up while the wave steps the machine), to be handed to
    // sweep() as `pre`.  What has not arrived by then is polled for as usual.
    __device__ bool fetch_ahead(int ex, unsigned long long (*pre)[64][2]) const {
        const int nent = nwg * nw, per = (nent + 63) >> 6, lane = tid & 63;
        if (per > 8) return false;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (j < per) {
                const int ww = lane + (j << 6);
                const gu64* src = gran(ex, 0, ww < nent ? ww : 0);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)&pre[j][0][0], 16, 0, 16 /* sc1 */);
            }
        }
        return true;
    }

    template <int N, int NMAX>
    __device__ void sweep(int ex, double (&v)[N], int delay, const unsigned long long (*pre)[64][2] = nullptr) {
        const int lane = tid & 63;
        const int nent = nwg * nw;
        double acc[N];
#pragma unroll
        for (int k = 0; k < N; ++k) acc[k] = 0.0;
        unsigned spins = 0;
        bool ok = !dead;
        const int per = (nent + 63) >> 6;
        constexpr int CH = N <= 2 ? 8 : 4;
        for (int d = 0; d < delay; d += 8) __builtin_amdgcn_s_sleep(8);
        for (int j0 = 0; j0 < per && ok; j0 += CH) {
            unsigned long long val[CH][N];
#pragma unroll
            for (int j = 0; j < CH; ++j)
#pragma unroll
                for (int k = 0; k < N; ++k) val[j][k] = COOP_SENTINEL;
            if constexpr (N == 2) {
                if (pre != nullptr) {   // fetched ahead (fetch_ahead: at most 8 chunks, i.e. j0 = 0 only)
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                    for (int j = 0; j < CH; ++j) {
                        if (j0 + j < per) { val[j][0] = pre[j0 + j][lane][0]; val[j][1] = pre[j0 + j][lane][1]; }
                    }
                }
            }
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
#pragma unroll
        for (int k = 0; k < N; ++k) v[k] = k < N - NMAX ? wave_sum(acc[k]) : wave_max(acc[k]);
        tick(5, 1);
        // (every memory operation of this wave, the previous re-arming stores included, is complete:
        // the sweep's last wait was for vmcnt(0))
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int upto = ex - PIPE_DEPTH - 1;
        if (tid < nw) {
            const int ent = wg * nw + tid;
            for (int x = rearmed + 1; x <= upto; ++x) {
#pragma unroll
                for (int k = 0; k < COOP_KP; k += 2) store_pair(x, k, ent, COOP_SENTINEL, COOP_SENTINEL);
            }
        }
        if (upto > rearmed) rearmed = upto;
    }

    // Grid-wide barrier, all lanes of the workgroup call: the lanes' coherent stores issued before it
    // are visible to coherent loads after it.  Takes one exchange number.
    __device__ void barrier(PipeShared& S) {
        double z[1] = {0.0};
        if (ctrl()) {
            sweep<1, 0>(e, z, poll_delay);
            if (tid == 0 && dead) S.dead = 1;
        } else {
            publish<1, 0>(e, z);
        }
        ++e;
        __syncthreads();
        if (S.dead != 0) dead = true;
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
    int gt, tid;              // factor lane index in the group (-1 on the control wave), lane in workgroup
    PipeSync X;
    double* tr;
    int trn, lm_count;
    bool has_fac;
    int fid;
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
                for (int k = 0; k < 12; ++k) acc += g[k] * dirv[k];
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
        if (has_fac) {
            double v[12], g[12];
#pragma unroll
            for (int k = 0; k < 12; ++k) v[k] = clampd(base[k], lov[k], hiv[k]);
            ba_eval_grad(v, ox, oy, g);
            const int* sp = L.slot_pos + L.slot_base[c0 + gt];
            int t[12];
#pragma unroll
            for (int k = 0; k < 12; ++k) t[k] = sp[k];
#pragma unroll
            for (int k = 0; k < 12; ++k)
                if (t[k] >= 0) store_f64<true>(L.gfac + t[k], g[k]);
        }
        X.barrier(S);
        const int* vp = L.v2s_ptr + f0;
        if (lv.li >= 0) lv.xi = run_sum_ordered<true>(L.gfac, vp[lv.li], vp[lv.li + 1]);
        if (wv.li >= 0) wv.xi = wave_sum(run_sum_strided<true>(L.gfac, vp[wv.li], vp[wv.li + 1], tid & 63));
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
    __device__ void line_begin() {
        if (has_fac) {
            const int* sl = A.slot_li + 12ll * gt;
            int li[12];
#pragma unroll
            for (int k = 0; k < 12; ++k) li[k] = sl[k];
            double d[12];
#pragma unroll
            for (int k = 0; k < 12; ++k) d[k] = load_f64<true>(A.xi_glob + (li[k] >= 0 ? li[k] : 0));
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
    // the lanes' share of the Polak-Ribiere sums (nrc :655-672); the control wave collects them
    __device__ void cg_reduce_publish(double fp) {
#pragma clang fp contract(off)
        const double den = fmax(fabs(fp), 1.0);
        double a = 0.0, b = 0.0, t = 0.0;
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
        double v[3] = {a, b, t};
        X.publish<3, 1>(X.e, v);
        ++X.e;
    }
    __device__ void cg_update(double gam) {
        {
#pragma clang fp contract(off)
            { const double gn = -lv.xi; const double hn = gn + gam * lv.h; lv.g = gn; lv.h = hn; lv.xi = hn; }
            { const double gn = -wv.xi; const double hn = gn + gam * wv.h; wv.g = gn; wv.h = hn; wv.xi = hn; }
        }
        publish_xi();
    }
    __device__ void write_back(const VarState& V, bool restore, bool writer) {
        if (V.li >= 0 && writer) {
            const double xf = clampd(restore ? V.xinit : V.p, V.lo, V.hi);
            P.x[L.free_vid[f0 + V.li]] = xf;
            L.xout[f0 + V.li] = xf;
        }
    }
};

// debug counters (-DRDIS_COOP_TIMING): slots filled by a lane wave, the others by the control wave
// lanes: 0 factor arithmetic, 2 reduce + publish, 9 waiting for a request, 10 requests evaluated, 11 guesses evaluated
// control: 1 / 3 sweep of a guessed / fresh step (16 / 17 their counts), 5 #sweeps, 6 #polls, 7 whole kernel, 8 step + post,
//          18 guesses that held, 19 guesses posted, 12.. cycles serving a request after its post: 12 value, 13 value+slope,
//          14 gradient (+ reduction), 15 value+slope at the start of a line (direction update first); 22.. their counts
__device__ __forceinline__ bool pipe_lane_slot(int i) { return i == 0 || i == 2 || (i >= 9 && i <= 11); }
__device__ __forceinline__ int lds_seq(const PipeShared& S) { return *(const volatile int*)&S.seq; }

// the control wave posts a request (lane 0 writes; LDS operations of a wave execute in order, so
// whoever reads the new sequence number reads the new slot)
__device__ __forceinline__ void pipe_post(PipeShared& S, int seq, int kind, int flags, int e, int ng, double a, double b,
                                          const double (&guess)[PIPE_DEPTH], bool writer) {
    if (writer) {
        PipeMail t;
        t.kind = kind; t.flags = flags; t.e = e; t.ng = ng; t.a = a; t.b = b; t.pad = 0.0;
#pragma unroll
        for (int k = 0; k < PIPE_DEPTH; ++k) t.guess[k] = guess[k];
        S.mail[seq & (PIPE_MAILS - 1)] = t;
        asm volatile("" ::: "memory");
        *(volatile int*)&S.seq = seq;
    }
}

// ---- control wave ------------------------------------------------------------------------
__device__ __forceinline__ void pipe_control(PipeEnv& E, PipeShared& S, int maxiters, double ftol) {
    CgdMachine M;
    M.init(maxiters, ftol);
    Predictor G;
    G.ph = Predictor::P_STOP; G.need_first = false; G.a = G.b = G.x = G.dx = 0.0; G.ax = G.bx = G.cx = 0.0;
    double chain[PIPE_DEPTH];   // guesses posted and not yet asked for: chain[j] is slot e_last + 1 + j
#pragma unroll
    for (int k = 0; k < PIPE_DEPTH; ++k) chain[k] = 0.0;
    int nchain = 0;
    int e_last = 0;             // slot of the last value+slope evaluation
    int next_free = 0;          // first exchange number not handed out
    bool swapped = true;        // did the last bracketing go to the other side of the origin?
    double r0 = 0.0, r1 = 0.0, r2 = 0.0;
    int seq = 0;
    const bool writer = E.tid == 0;
    for (;;) {
        const long long ts0 = coop_clock();
        // the reply to the first guess of the chain is probably what the machine will ask for next:
        // its loads travel while the machine is stepped
        const bool fetched = nchain > 0 && E.X.fetch_ahead(e_last + 1, S.pre);
        Request nq;
        Predictor Gn;
        bool was_hot;
        {
            double un, pa, pb, pc;
            int ptag;
            was_hot = M.hot(r0, r1, un, ptag, pa, pb, pc, Gn);
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
        const bool hit = slope && nq.flags == (RF_SLOPE | RF_LINE) && nchain > 0 && same_bits(nq.a, chain[0]);
        int e;
        if (hit) {
            // the step asked for is the first guess of the chain: its slot is the next one, the chain moves up
            e = e_last + 1;
#pragma unroll
            for (int k = 0; k + 1 < PIPE_DEPTH; ++k) chain[k] = chain[k + 1];
            --nchain;
            E.X.tick(18, 1);
        } else {
            e = next_free;
            nchain = 0;
        }
        // (a guessed step: the lanes are at work on the chain already and hear of it once, with the chain's new end)
        if (!hit) pipe_post(S, ++seq, nq.kind, nq.flags, e, nchain, nq.a, nq.b, chain, writer);
        if (E.tr != nullptr) {
            if ((nq.flags & RF_TR_FIRST) && nq.tr_tag != TR_NONE) E.trace(nq.tr_tag, nq.tr_a, nq.tr_b, nq.tr_c);
            if (nq.pre_tag != TR_NONE) E.trace(nq.pre_tag, nq.pre_a, nq.pre_b, nq.pre_c);
            if (!(nq.flags & RF_TR_FIRST) && nq.tr_tag != TR_NONE) E.trace(nq.tr_tag, nq.tr_a, nq.tr_b, nq.tr_c);
        }
        const long long ts1 = coop_clock();
        E.X.tick(8, ts1 - ts0);
        if (nq.kind == REQ_DONE) break;
        E.X.e = e;
        [[maybe_unused]] const int tkind = nq.kind == REQ_GRAD ? 2 : !slope ? 0 : (nq.flags & (RF_PRE_START | RF_PRE_UPDATE)) ? 3 : 1;
        switch (nq.kind) {
        case REQ_EVAL:
            if (nq.flags & (RF_PRE_START | RF_PRE_UPDATE)) E.X.barrier(S);   // the lanes' publish_xi
            if (slope) {
                // lengthen the chain while the lanes work, and tell them (a second post of the same
                // request: its evaluation is slot E.X.e, which they have or are about to publish)
                if (!hit) { if (was_hot) G = Gn; else G.start(M, swapped); }
                const int had = nchain;
                if (E.A.speculate) {
#pragma unroll
                    for (int k = 0; k < PIPE_DEPTH; ++k) {
                        double c;
                        if (nchain == k && G.next(c)) { chain[k] = c; nchain = k + 1; }
                    }
                }
                if (hit || nchain != had) pipe_post(S, ++seq, REQ_EVAL, RF_SLOPE | RF_LINE, E.X.e, nchain, nq.a, nq.b, chain, writer);
                E.X.tick(19, nchain - had);
                double v[2] = {0.0, 0.0};
                const long long tw0 = coop_clock();
                // a guessed step is under way or done; a fresh one takes the lanes an evaluation
                E.X.sweep<2, 0>(E.X.e, v, hit ? 0 : E.X.poll_delay, hit && fetched ? S.pre : nullptr);
                if (hit) { E.X.tick(1, coop_clock() - tw0); E.X.tick(16, 1); }
                else { E.X.tick(3, coop_clock() - tw0); E.X.tick(17, 1); }
                r0 = v[0]; r1 = v[1];
                e_last = E.X.e;
                next_free = e_last + PIPE_DEPTH + 1;
                E.trace(TR_FD, nq.a, r0, r1);
            } else {
                double v[1] = {0.0};
                E.X.sweep<1, 0>(E.X.e, v, E.X.poll_delay);
                r0 = v[0];
                next_free = E.X.e + 1;
                if (nq.flags & RF_LINE) E.trace(TR_F, nq.a, r0, 0.0);
            }
            break;
        case REQ_GRAD:
            E.X.barrier(S);   // gradient_to_xi: partials -> per-variable sums
            if (nq.flags & RF_POST_REDUCE) {
                double v[3] = {0.0, 0.0, 0.0};
                E.X.sweep<3, 1>(E.X.e, v, E.X.poll_delay);
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
    // the result, for everybody
    if (writer) {
        S.status = M.status(); S.rolled_back = M.rolled_back ? 1 : 0; S.iter = M.iter;
        S.nfeval = M.nfeval; S.ngeval = M.ngeval; S.fret = M.fret; S.finit = M.finit;
    }
}

// ---- lane waves ----------------------------------------------------------------------------
__device__ __forceinline__ void pipe_lanes(PipeEnv& E, PipeShared& S) {
    int seen = 0;      // last post acted on
    int done = -1;     // highest slot this wave has published
    for (;;) {
        int s;
        const long long tm0 = coop_clock();
        while ((s = lds_seq(S)) == seen) __builtin_amdgcn_s_sleep(1);
        E.X.tick(9, coop_clock() - tm0);
        seen = s;
        asm volatile("" ::: "memory");
        const PipeMail mm = S.mail[s & (PIPE_MAILS - 1)];
        const PipeMail* m = &mm;
        const int kind = __builtin_amdgcn_readfirstlane(m->kind);
        const int flags = __builtin_amdgcn_readfirstlane(m->flags);
        const int ng = __builtin_amdgcn_readfirstlane(m->ng);
        E.X.e = __builtin_amdgcn_readfirstlane(m->e);
        const double qa = uniform(m->a);
        if (kind == REQ_DONE) break;
        if (kind == REQ_EVAL) {
            if (flags & RF_PRE_START) E.cg_start();
            if (flags & RF_PRE_UPDATE) E.cg_update(uniform(m->b));
            if (flags & RF_PRE_BEGIN) E.line_begin();
            if (flags & RF_SLOPE) {
                const int e0 = E.X.e;
                if (e0 > done) {
                    double v[2];
                    const long long te0 = coop_clock();
                    E.eval_line<true>(qa, v[0], v[1]);
                    const long long te1 = coop_clock();
                    E.X.publish<2, 0>(e0, v);
                    E.X.tick(0, te1 - te0); E.X.tick(2, coop_clock() - te1); E.X.tick(10, 1);
                    done = e0;
                }
                // guesses at the following steps, until the control wave has something new to say
#pragma unroll
                for (int j = 0; j < PIPE_DEPTH; ++j) {
                    if (j >= ng || lds_seq(S) != seen) break;
                    const int ge = e0 + 1 + j;
                    if (ge <= done) continue;
                    double v[2];
                    const long long te0 = coop_clock();
                    E.eval_line<true>(uniform(m->guess[j]), v[0], v[1]);
                    const long long te1 = coop_clock();
                    E.X.publish<2, 0>(ge, v);
                    E.X.tick(0, te1 - te0); E.X.tick(2, coop_clock() - te1); E.X.tick(11, 1);
                    done = ge;
                }
            } else {
                if (flags & RF_RESTORE) E.load_base(E.L.xstart + E.f0);
                double v[1], dummy;
                E.eval_line<false>((flags & RF_RESTORE) ? 0.0 : qa, v[0], dummy);
                E.X.publish<1, 0>(E.X.e, v);
                done = E.X.e;
            }
        } else if (kind == REQ_GRAD) {
            if (flags & RF_PRE_LINE_END) E.line_end(qa);
            E.gradient_to_xi();
            if (flags & RF_POST_REDUCE) { E.cg_reduce_publish(uniform(m->b)); done = E.X.e - 1; }
        } else if (kind == REQ_LINE_END) {
            E.line_end(qa);
        }
    }
}

template <int THREADS>
__device__ __forceinline__ void pipe_solve(const ProblemView& P, const PlanView& L, const CoopArgs& A, int nwg, int wg,
                                           int maxiters, double ftol) {
    static_assert(THREADS == PIPE_THREADS, "one control wave + three lane waves");
    __shared__ PipeShared S;
    [[maybe_unused]] const long long tk0 = coop_clock();
    const int comp = A.comp;
    const int f0 = L.free_ptr[comp], c0 = L.fac_ptr[comp];
    const int n = L.free_ptr[comp + 1] - f0, m = L.fac_ptr[comp + 1] - c0;
    const int tid = (int)threadIdx.x;
    const int gt = tid < 64 ? -1 : wg * PIPE_LANES + tid - 64;

    PipeEnv E{P, L, A, S, n, m, f0, c0, gt, tid,
              PipeSync{(PipeState*)A.st, tid, nwg, wg, PIPE_LANES / 64, A.poll_delay, 0, -1, false
#ifdef RDIS_COOP_TIMING
                       , {}
#endif
              },
              L.trace ? L.trace + 4ll * L.trace_cap * comp : nullptr, 0, 0,
              gt >= 0 && gt < m, 0, {}, {}, {}, {}, 0.0, 0.0, {}, {}};
    if (E.has_fac) {
        E.fid = L.fac_id[c0 + gt];
        const double2 o = P.obs[E.fid];
        E.ox = o.x; E.oy = o.y;
    }
    if (tid == 0) { S.seq = 0; S.dead = 0; }
    E.init_vectors();
    __syncthreads();
    if (tid < 64) pipe_control(E, S, maxiters, ftol);
    else pipe_lanes(E, S);
    __syncthreads();
    const bool restore = S.rolled_back != 0;
    E.write_back(E.lv, restore, true);
    E.write_back(E.wv, restore, (tid & 63) == 0);
    if (wg == 0 && tid == 0) {
        L.fret[comp] = S.fret; L.delta[comp] = S.fret - S.finit; L.iters[comp] = S.iter;
        L.status[comp] = S.status; L.nfeval[comp] = S.nfeval; L.ngeval[comp] = S.ngeval;
        if (L.trace_n) L.trace_n[comp] = E.trn;
#ifdef RDIS_COOP_TIMING
        E.X.tm[7] = coop_clock() - tk0;
        if (A.timing) for (int i = 0; i < COOP_TM; ++i) if (!pipe_lane_slot(i)) A.timing[i] = E.X.tm[i];
#endif
    }
#ifdef RDIS_COOP_TIMING
    if (wg == 0 && tid == 64 && A.timing) for (int i = 0; i < COOP_TM; ++i) if (pipe_lane_slot(i)) A.timing[i] = E.X.tm[i];
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

__global__ void __launch_bounds__(256) pipe_arm_kernel(const CoopGroup* __restrict__ groups) {
    const CoopGroup G = groups[blockIdx.x];
    PipeState* st = (PipeState*)G.a.st;
    const int entries = G.nwg * (PIPE_LANES / 64);
    for (int t = threadIdx.x; t < PIPE_NBUF * COOP_KP * entries; t += blockDim.x) {
        const int k = t % COOP_KP, e = (t / COOP_KP) % entries, b = t / (COOP_KP * entries);
        st->granule[b][e][k] = ~0ull;
    }
    if (threadIdx.x == 0) st->abort_flag = 0u;
}

inline int launch_pipe(hipStream_t stream, int kind, const ProblemView& P, const PlanView& V, const CoopGroup& first,
                       const CoopGroup* groups, const int* wg_group, int ngroups, int total_wg, int maxiters, double ftol) {
    if (kind != KIND_BA) return (int)hipErrorNotSupported;
    pipe_arm_kernel<<<ngroups, 256, 0, stream>>>(groups);
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
