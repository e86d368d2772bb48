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
constexpr int COOP_K = 3;  // values per exchange
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

struct CoopState {
    // [COOP_NBUF buffers][COOP_K][entries: workgroup * waves + wave] granules, then an abort word
    unsigned long long granule[COOP_NBUF][COOP_K][COOP_MAX_WG * COOP_MAX_WAVES];
    unsigned int abort_flag;
    unsigned int pad[15];
};
inline size_t coop_state_bytes() { return sizeof(CoopState); }

struct GridSync {
    CoopState* st;
    int tid, nwg, wg;                   // lane in workgroup, #workgroups, my workgroup
    double* bcast;                      // LDS [2][4]
    int poll_delay;                     // x64 cycles between publishing and the first sweep
    int parity;
    unsigned epoch;
    bool dead;                          // a spin gave up: unwind quickly
    long long tm[COOP_TM]; // cycles: 0 factor arithmetic, 1 workgroup reduce, 2 publish, 3 sweep, 4 tail,
                      // 5 #exchanges, 6 #sweeps, 7 whole kernel, 8 state-machine step, 9 request hand-over,
                      // 10 combine waves, 11 release, 12.. handler cycles: 12 value, 13 value+slope, 14 gradient (+reduce), 17 line end; 22.. their counts

    // ---- inter-workgroup exchange ------------------------------------------------
    __device__ gu64* gran(int buf, int k, int w) const { return (gu64*)&st->granule[buf][k][w]; }

    // All lanes call.  On return wave 0 of every workgroup holds the first K of (sum a, sum b,
    // max mx) over the whole grid, bit-identical in every workgroup; the other waves hold
    // garbage.  Must be paired with finish_wave0(sync) before the workgroup's next barrier.
    // Every WAVE publishes its own partial sums (entry = workgroup * waves + wave): no LDS stage and
    // no workgroup barrier between the arithmetic and the stores; wave 0 sweeps all entries.
    template <int K>
    __device__ void to_wave0(double& a, double& b, double& mx, int sync) {
        const long long t0 = coop_clock();
        long long t2 = t0;
        const int w = tid >> 6, lane = tid & 63, nwv = blockDim.x >> 6;
        a = wave_sum(a);
        if constexpr (K >= 2) b = wave_sum(b);
        if constexpr (K >= 3) mx = wave_max(mx);
        const int buf = epoch & (COOP_NBUF - 1);
        if (sync != SYNC_NONE) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's stores are out
        if (lane == 0) {
            // (this lane's re-arming stores to these granules were issued three exchanges
            // ago and have been waited for, see finish_wave0)
            if (sync == SYNC_FENCE) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            const int e = wg * nwv + w;
            publish(gran(buf, 0, e), a);
            if constexpr (K >= 2) publish(gran(buf, 1, e), b);
            if constexpr (K >= 3) publish(gran(buf, 2, e), mx);
        }
        t2 = coop_clock();
        if (w == 0) {
            // sweep: lane l looks after entries l, l+64, ...
            const int nent = nwg * nwv;
            double sa = 0.0, sb = 0.0, sm = 0.0;
            unsigned spins = 0;
            bool ok = !dead;
            const int per = (nent + 63) >> 6;
            // a store needs about this long to land; polling earlier only slows it down
            for (int d = 0; d < poll_delay; d += 8) __builtin_amdgcn_s_sleep(8);
            while (!dead) {
                ++tm[6];
                ok = true;
                sa = 0.0; sb = 0.0; sm = 0.0;
                // eight entries (x K values) per lane in flight at a time: one memory round trip
                // per chunk (a lane's out-of-range slots read entry 0)
                for (int j0 = 0; j0 < per; j0 += 8) {
                    unsigned long long va[8], vb[8], vm[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int ww = lane + ((j0 + j) << 6);
                        const int wc = ww < nent ? ww : 0;
                        va[j] = __hip_atomic_load(gran(buf, 0, wc), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if constexpr (K >= 2) vb[j] = __hip_atomic_load(gran(buf, 1, wc), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if constexpr (K >= 3) vm[j] = __hip_atomic_load(gran(buf, 2, wc), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        ok = ok && va[j] != COOP_SENTINEL;
                        if constexpr (K >= 2) ok = ok && vb[j] != COOP_SENTINEL;
                        if constexpr (K >= 3) ok = ok && vm[j] != COOP_SENTINEL;
                        if (lane + ((j0 + j) << 6) < nent) {
                            sa += __longlong_as_double(va[j]);
                            if constexpr (K >= 2) sb += __longlong_as_double(vb[j]);
                            if constexpr (K >= 3) sm = fmax(sm, __longlong_as_double(vm[j]));
                        }
                    }
                }
                if (__all(ok)) break;
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
            if (!__all(ok)) dead = true;
            a = wave_sum(sa);
            if constexpr (K >= 2) b = wave_sum(sb);
            if constexpr (K >= 3) mx = wave_max(sm);
        }
        const long long t3 = coop_clock();
        tm[2] += t2 - t0; tm[3] += t3 - t2; ++tm[5];
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
        const int nwv = blockDim.x >> 6;
        if (tid < nwv) {
            if (tid == 0 && sync == SYNC_FENCE) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the previous re-arming stores: an exchange old
            const int nb = (epoch + 3u) & (COOP_NBUF - 1);
            const int e = wg * nwv + tid;
            __hip_atomic_store(gran(nb, 0, e), COOP_SENTINEL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(gran(nb, 1, e), COOP_SENTINEL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(gran(nb, 2, e), COOP_SENTINEL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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

}  // namespace rdis_hip
