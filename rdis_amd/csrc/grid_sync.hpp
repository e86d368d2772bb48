// grid_sync.hpp -- exchange of partial sums between the workgroups of one cooperative launch.
//
// 8-byte granules in HBM (relaxed agent-scope atomics, written through to memory): the data is
// the flag.  A granule holds an all-ones NaN until its owner publishes.  Every WAVE owns an entry
// (workgroup * waves + wave) and publishes its own partial sums straight after its wave reduction
// -- no LDS stage, no workgroup barrier between the arithmetic and the stores (that stage cost
// ~900 cycles per exchange).  Wave 0 of every workgroup sweeps all entries, eight per lane in
// flight, and reduces them in a fixed order, so every workgroup obtains bit-identical results and
// takes identical branches.
//
// Four buffers rotate.  After its sweep of exchange e, wave 0 re-arms the entries of its
// workgroup's waves in buffer e+3 (= e-1) and first waits for the re-arming stores it issued after
// e-1.  Safe: the sweep of e completing proves that every workgroup has published in e, i.e. has
// left its sweep of e-1; and the re-arming stores are complete before this workgroup publishes in
// e+1, which every other workgroup must see before it can look at buffer e+2 -- nobody can read a
// granule of an earlier use of a buffer.
//
// An exchange can also order memory (`sync`): SYNC_DRAIN makes every wave wait for its own
// outstanding stores before it publishes -- enough when the data handed over is itself written and
// read with coherent accesses (store_f64<true> / load_f64<true>); SYNC_FENCE additionally brackets
// the exchange with an agent-scope release / acquire (cdna_hip_programming.md Guideline 16: L2
// write-back and invalidate) so that plain stores issued before it are visible to plain loads
// after it.  Placement-independent; every spin is bounded and raises `dead`.
//
// Two forms:
//   to_wave0 + finish_wave0   the sums are delivered to wave 0 of every workgroup only -- the
//                             wave that steps the solver's state machine is the only consumer of
//                             a line-search value, so the other waves go straight to the barrier
//                             that hands them the next request;
//   exchange / barrier        delivered to every lane (used inside operations that continue
//                             with the result, e.g. partials -> per-variable sums).
#pragma once
#include "solver_wg.hpp"

namespace rdis_hip {

constexpr int COOP_MAX_WG = 512;
constexpr int COOP_SPEC = 1;  // line-search trial points evaluated per exchange (minimizer.hpp: speculation)
constexpr int COOP_K = (2 * COOP_SPEC > 3 ? 2 * COOP_SPEC : 3);  // values per exchange
constexpr int COOP_NBUF = 4;
constexpr int COOP_MAX_WAVES = 8;  // waves per workgroup of the grid solvers (512 lanes)
enum : int { SYNC_NONE = 0, SYNC_DRAIN = 1, SYNC_FENCE = 2 };
constexpr int COOP_TM = 32;  // debug counters (rdis_hip_plan_debug_counters)
constexpr int COOP_LONG_LIST = 48;  // variables fed by more partials than this are wave-owned
constexpr unsigned long long COOP_SENTINEL = 0xFFFFFFFFFFFFFFFFull;
constexpr unsigned long long COOP_CANON_NAN = 0x7FF8000000000000ull;
constexpr unsigned COOP_SPIN_LIMIT = 1u << 22;

// shader-clock stamps for the per-phase breakdown (rdis_hip_plan_debug_counters).  Each stamp
// is a scalar memory read (~150 cycles on the critical path), so they are compiled in only
// with -DRDIS_COOP_TIMING (make -C rdis_amd/csrc EXTRA=-DRDIS_COOP_TIMING).
__device__ __forceinline__ long long coop_clock() {
#ifdef RDIS_COOP_TIMING
    return clock64();
#else
    return 0;
#endif
}

typedef __attribute__((address_space(1))) unsigned long long gu64;
typedef __attribute__((address_space(1))) unsigned int gu32;
typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(1))) u64x2 gu64x2;

constexpr int COOP_KP = (COOP_K + 1) / 2 * 2;   // granules per entry, padded to 16-byte pairs
struct CoopState {
    // [COOP_NBUF buffers][entries: workgroup * waves + wave][COOP_KP] granules -- the values an entry
    // publishes in one exchange are neighbours, written and read two at a time (16 bytes: half the
    // memory operations of a sweep; each half is still checked on its own, nothing relies on the
    // pair arriving together) -- then an abort word
    alignas(32) unsigned long long granule[COOP_NBUF][COOP_MAX_WG * COOP_MAX_WAVES][COOP_KP];
    unsigned int abort_flag;
    unsigned int pad[15];
};
inline size_t coop_state_bytes() { return sizeof(CoopState); }

// the same for a small group: a component shared by a few workgroups of the point-major streaming solver
// (solver_ptm.hpp), one state per group that runs concurrently -- up to a hundred or so per launch
constexpr int SMALL_COOP_ENTRIES = 256;   // workgroups x waves of a group
struct SmallCoopState {
    alignas(32) unsigned long long granule[COOP_NBUF][SMALL_COOP_ENTRIES][COOP_KP];
    unsigned int abort_flag;
    unsigned int pad[15];
};

// ... and for a WIDE group of that solver (a large share of the device on one component): a CoopState whose entries are one per
// WORKGROUP -- its first wave publishes values the caller has already summed over the workgroup (every lane holds them) --
// 256 entries to sweep instead of 2048
struct WideCoopState : CoopState {};

template <class ST>
struct GridSyncT {
    static constexpr bool wg_entry = std::is_same<ST, WideCoopState>::value;
    ST* st;
    int tid, nwg, wg;                   // lane in workgroup, #workgroups, my workgroup
    double* bcast;                      // LDS [2][4]
    int poll_delay;                     // x64 cycles between publishing and the first sweep
    int parity;
    unsigned epoch;
    bool dead;                          // a spin gave up: unwind quickly
    unsigned kused;                     // byte b: how many granules per entry the last exchange in buffer b published
    long long tm[COOP_TM]; // cycles: 0 factor arithmetic, 1 workgroup reduce, 2 publish, 3 sweep, 4 tail,
                      // 5 #exchanges, 6 #sweeps, 7 whole kernel, 8 state-machine step, 9 request hand-over,
                      // 10 combine waves, 11 release, 12.. handler cycles: 12 value, 13 value+slope, 14 gradient (+reduce), 17 line end; 22.. their counts


    // ---- inter-workgroup exchange ------------------------------------------------
    __device__ gu64* gran(int buf, int k, int w) const { return (gu64*)&st->granule[buf][w][k]; }
    // granules k, k + 1 (k even) of an entry as one 16-byte access at agent scope (sc1: served by
    // L2 / the fabric like the 8-byte agent-scope atomics; the compiler has no 16-byte form of those,
    // and a volatile access is system scope -- measured twice as slow).  The load is asynchronous:
    // its result may be used only after the s_waitcnt that follows the batch (to_wave0_n).
    __device__ u64x2 load_pair(int buf, int k, int w) const {
        u64x2 r;
        const gu64* p = (const gu64*)&st->granule[buf][w][k];
        asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(r) : "v"(p) : "memory");
        return r;
    }
    __device__ void store_pair(int buf, int k, int w, unsigned long long a, unsigned long long b) const {
        u64x2 t; t.x = a; t.y = b;
        gu64* p = (gu64*)&st->granule[buf][w][k];
        asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(p), "v"(t) : "memory");
    }
    __device__ static unsigned long long bits_of(double v) {
        const unsigned long long u = (unsigned long long)__double_as_longlong(v);
        return u == COOP_SENTINEL ? COOP_CANON_NAN : u;   // the one NaN pattern that means "not yet"
    }

    // All lanes call.  On return wave 0 of every workgroup holds, for the N values of `v`, the sum
    // (the last NMAX of them: the maximum) over the whole grid, bit-identical in every workgroup; the
    // other waves hold garbage.  Must be paired with finish_wave0(sync) before the workgroup's next
    // barrier.  Every WAVE publishes its own partial results (entry = workgroup * waves + wave): no
    // LDS stage and no workgroup barrier between the arithmetic and the stores; wave 0 sweeps all
    // entries.
    template <int N, int NMAX>
    __device__ void to_wave0_n(double (&v)[N], int sync) {
        static_assert(N >= 1 && N <= COOP_K && NMAX >= 0 && NMAX <= N, "exchange width");
        const long long t0 = coop_clock();
        long long t2 = t0;
        const int w = tid >> 6, lane = tid & 63, nwv = wg_entry ? 1 : blockDim.x >> 6;
        if (!wg_entry) {
#pragma unroll
            for (int k = 0; k < N; ++k) v[k] = k < N - NMAX ? wave_sum(v[k]) : wave_max(v[k]);
        }
        const int buf = epoch & (COOP_NBUF - 1);
        kused = (kused & ~(0xFFu << (8 * buf))) | ((unsigned)N << (8 * buf));
        if (sync != SYNC_NONE) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's stores are out
        if (wg_entry && sync != SYNC_NONE) __syncthreads();   // ... and every wave's, before the workgroup's one entry says so
        if (lane == 0 && (!wg_entry || w == 0)) {
            // (this lane's re-arming stores to these granules were issued three exchanges
            // ago and have been waited for, see finish_wave0)
            if (sync == SYNC_FENCE) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            const int e = wg * nwv + w;
#pragma unroll
            for (int k = 0; k + 1 < N; k += 2) store_pair(buf, k, e, bits_of(v[k]), bits_of(v[k + 1]));
            if constexpr (N & 1) publish(gran(buf, N - 1, e), v[N - 1]);
        }
        t2 = coop_clock();
        if (w == 0) {
            // sweep: lane l looks after entries l, l+64, ..., CH of them in flight at a time (one memory
            // round trip per chunk; a lane's out-of-range slots read entry 0).  A chunk is re-polled until
            // all of it is there -- only the granules still missing are loaded again -- and then added
            // up: chunks in order, entries in order, so every workgroup gets the same bits whatever
            // the timing.
            const int nent = nwg * nwv;
            double acc[N];
#pragma unroll
            for (int k = 0; k < N; ++k) acc[k] = 0.0;
            unsigned spins = 0;
            bool ok = !dead;
            const int per = (nent + 63) >> 6;
            constexpr int CH = N <= 2 ? 8 : 4;
            // a store needs about this long to land; polling earlier only slows it down
            for (int d = 0; d < poll_delay; d += 8) __builtin_amdgcn_s_sleep(8);
            for (int j0 = 0; j0 < per && ok; j0 += CH) {
                unsigned long long val[CH][N];
#pragma unroll
                for (int j = 0; j < CH; ++j)
#pragma unroll
                    for (int k = 0; k < N; ++k) val[j][k] = COOP_SENTINEL;
                for (;;) {
                    ++tm[6];
                    constexpr int NP = N / 2;
                    u64x2 pr[CH][NP > 0 ? NP : 1];
#pragma unroll
                    for (int j = 0; j < CH; ++j) {
                        const int ww = lane + ((j0 + j) << 6);
                        const int wc = ww < nent ? ww : 0;
#pragma unroll
                        for (int k = 0; k + 1 < N; k += 2) {
                            pr[j][k / 2].x = val[j][k]; pr[j][k / 2].y = val[j][k + 1];
                            if (val[j][k] == COOP_SENTINEL || val[j][k + 1] == COOP_SENTINEL) pr[j][k / 2] = load_pair(buf, k, wc);
                        }
                        if constexpr (N & 1) {
                            if (val[j][N - 1] == COOP_SENTINEL)
                                val[j][N - 1] = __hip_atomic_load(gran(buf, N - 1, wc), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                    }
                    if constexpr (NP > 0) {
                        // (the 16-byte loads are invisible to the compiler's own wait counting: every
                        // pair passes through these statements before it is looked at)
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
        }
        const long long t3 = coop_clock();
        tm[2] += t2 - t0; tm[3] += t3 - t2; ++tm[5];
    }
    // the first K of (sum a, sum b, max mx)
    template <int K>
    __device__ void to_wave0(double& a, double& b, double& mx, int sync) {
        if constexpr (K == 1) { double v[1] = {a}; to_wave0_n<1, 0>(v, sync); a = v[0]; }
        else if constexpr (K == 2) { double v[2] = {a, b}; to_wave0_n<2, 0>(v, sync); a = v[0]; b = v[1]; }
        else { double v[3] = {a, b, mx}; to_wave0_n<3, 1>(v, sync); a = v[0]; b = v[1]; mx = v[2]; }
    }
    __device__ static void publish(gu64* g, double v) {
        unsigned long long u = __double_as_longlong(v);
        if (u == COOP_SENTINEL) u = COOP_CANON_NAN;   // the one NaN pattern that means "not yet"
        __hip_atomic_store(g, u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }

    // Acquire side of an ordered exchange, and re-arming of this workgroup's granules three
    // exchanges ahead (see the header).  All lanes advance epoch / parity.
    __device__ void finish_wave0(int sync) {
        // Wave 0 re-arms the entries of all waves of its workgroup (lane w: wave w's), and only here,
        // after its sweep: the sweep of exchange e completing proves that every workgroup has left
        // the sweep of e-1, whose buffer is the one re-armed.  (A wave re-arming its own entry right
        // after publishing could pull a granule from under a slower workgroup's sweep.)
        const int nwv = wg_entry ? 1 : blockDim.x >> 6;
        if (tid < nwv) {
            if (tid == 0 && sync == SYNC_FENCE) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the previous re-arming stores: an exchange old
            const int nb = (epoch + 3u) & (COOP_NBUF - 1);
            const int e = wg * nwv + tid;
            const int nk = (int)((kused >> (8 * nb)) & 0xFFu);   // what the buffer's last use published
#pragma unroll
            for (int k = 0; k < COOP_KP; k += 2)
                if (k < nk) store_pair(nb, k, e, COOP_SENTINEL, COOP_SENTINEL);
        }
        parity ^= 1;
        ++epoch;
    }

    // Sum (k = 0,1) / max (k = 2) of one value per workgroup, delivered to every lane of every
    // workgroup.
    __device__ void exchange(double& a, double& b, double& mx, int sync) {
        const long long t3 = coop_clock();
        to_wave0<3>(a, b, mx, sync);
        if (tid == 0) {
            bcast[parity * 4 + 0] = a; bcast[parity * 4 + 1] = b; bcast[parity * 4 + 2] = mx;
            bcast[parity * 4 + 3] = dead ? 0.0 : 1.0;
        }
        const int par = parity;
        finish_wave0(sync);
        __syncthreads();
        a = bcast[par * 4 + 0]; b = bcast[par * 4 + 1]; mx = bcast[par * 4 + 2];
        if (bcast[par * 4 + 3] == 0.0) dead = true;
        tm[4] += coop_clock() - t3;
    }
    // a grid-wide barrier that orders memory as `sync` says; no payload
    __device__ void barrier(int sync) {
        double a = 0.0, b = 0.0, c = 0.0;
        const long long t3 = coop_clock();
        to_wave0<1>(a, b, c, sync);
        if (tid == 0) bcast[parity * 4 + 3] = dead ? 0.0 : 1.0;
        const int par = parity;
        finish_wave0(sync);
        __syncthreads();
        if (bcast[par * 4 + 3] == 0.0) dead = true;
        tm[4] += coop_clock() - t3;
    }
    __device__ void barrier_ordered() { barrier(SYNC_FENCE); }

};
using GridSync = GridSyncT<CoopState>;

}  // namespace rdis_hip
