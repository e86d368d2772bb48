// grid_sync.hpp -- exchange of partial sums between the workgroups of one cooperative launch.
//
// 8-byte granules in HBM (relaxed agent-scope atomics, written through to memory): the data is
// the flag.  A granule holds an all-ones NaN until its owner publishes; three buffers rotate and
// each workgroup re-arms its own granules two exchanges ahead (drained with s_waitcnt vmcnt(0)
// before its next publish).  One wave per workgroup sweeps all granules and reduces them in a
// fixed order, so every workgroup obtains bit-identical results and takes identical branches.
// `order` brackets the exchange with an agent-scope release / acquire (cdna_hip_programming.md
// Guideline 16) so that plain stores issued before it are visible to every lane after it.
// Placement-independent; every spin is bounded and raises `dead`.
#pragma once
#include "solver_wg.hpp"

namespace rdis_hip {

constexpr int COOP_MAX_WG = 512;
constexpr int COOP_K = 3;  // values per exchange
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
    // [3 buffers][COOP_K][COOP_MAX_WG] granules, then an abort word
    unsigned long long granule[3][COOP_K][COOP_MAX_WG];
    unsigned int abort_flag;
    unsigned int pad[15];
};
inline size_t coop_state_bytes() { return sizeof(CoopState); }

struct GridSync {
    CoopState* st;
    int tid, nwg, wg;                   // lane in workgroup, #workgroups, my workgroup
    double (*red)[COOP_K][MAX_WAVES];   // LDS block-reduce scratch [2][K][waves]
    double* bcast;                      // LDS [2][4]
    int poll_delay;                     // x64 cycles between publishing and the first sweep
    int parity;
    unsigned epoch;
    bool dead;                          // a spin gave up: unwind quickly
    long long tm[12]; // cycles: 0 factor arithmetic, 1 workgroup reduce, 2 publish, 3 sweep, 4 tail,
                      // 5 #exchanges, 6 #sweeps, 7 whole kernel, 8 state-machine step, 9 request hand-over

    // ---- inter-workgroup exchange ------------------------------------------------
    __device__ gu64* gran(int buf, int k, int w) const { return (gu64*)&st->granule[buf][k][w]; }

    // Sum (k = 0,1) / max (k = 2) of one value per workgroup, delivered to every lane
    // of every workgroup, bit-identical everywhere.  `order` additionally makes all
    // plain global stores issued before the call visible to all lanes after it.
    __device__ void exchange(double& a, double& b, double& mx, bool order) {
        const long long t0 = coop_clock();
        long long t1 = t0, t2 = t0, t3 = t0;
        a = wave_sum(a); b = wave_sum(b); mx = wave_max(mx);
        const int w = tid >> 6, lane = tid & 63;
        if (lane == 0) { red[parity][0][w] = a; red[parity][1][w] = b; red[parity][2][w] = mx; }
        if (order) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave drains
        __syncthreads();
        const int buf = epoch % 3u;
        t1 = coop_clock();
        if (w == 0) {
            double ra = 0.0, rb = 0.0, rm = 0.0;
            const int nwv = blockDim.x >> 6;
            for (int i = 0; i < nwv; ++i) { ra += red[parity][0][i]; rb += red[parity][1][i]; rm = fmax(rm, red[parity][2][i]); }
            if (lane == 0) {
                if (order) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // re-arm stores + payload are out
                unsigned long long ua = __double_as_longlong(ra), ub = __double_as_longlong(rb), um = __double_as_longlong(rm);
                if (ua == COOP_SENTINEL) ua = COOP_CANON_NAN;
                if (ub == COOP_SENTINEL) ub = COOP_CANON_NAN;
                if (um == COOP_SENTINEL) um = COOP_CANON_NAN;
                __hip_atomic_store(gran(buf, 0, wg), ua, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(gran(buf, 1, wg), ub, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(gran(buf, 2, wg), um, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            // sweep: lane l looks after workgroups l, l+64, ...
            t2 = coop_clock();
            double sa = 0.0, sb = 0.0, sm = 0.0;
            unsigned spins = 0;
            bool ok = !dead;
            const int per = (nwg + 63) >> 6;
            for (int d = 0; d < poll_delay; d += 8) __builtin_amdgcn_s_sleep(8);
            while (!dead) {
                ++tm[6];
                // every load of the sweep is in flight before the first one is looked at: one
                // memory round trip per sweep (a lane's out-of-range slots read granule 0)
                constexpr int PER = COOP_MAX_WG / 64;
                unsigned long long va[PER], vb[PER], vm[PER];
#pragma unroll
                for (int j = 0; j < PER; ++j) {
                    if (j < per) {
                        const int ww = lane + (j << 6);
                        const int wc = ww < nwg ? ww : 0;
                        va[j] = __hip_atomic_load(gran(buf, 0, wc), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        vb[j] = __hip_atomic_load(gran(buf, 1, wc), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        vm[j] = __hip_atomic_load(gran(buf, 2, wc), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
                ok = true;
                sa = 0.0; sb = 0.0; sm = 0.0;
#pragma unroll
                for (int j = 0; j < PER; ++j) {
                    if (j < per) {
                        ok = ok && va[j] != COOP_SENTINEL && vb[j] != COOP_SENTINEL && vm[j] != COOP_SENTINEL;
                        if (lane + (j << 6) < nwg) {
                            sa += __longlong_as_double(va[j]); sb += __longlong_as_double(vb[j]);
                            sm = fmax(sm, __longlong_as_double(vm[j]));
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
            ok = __all(ok);
            t3 = coop_clock();
            sa = wave_sum(sa); sb = wave_sum(sb); sm = wave_max(sm);
            if (lane == 0) {
                bcast[parity * 4 + 0] = sa; bcast[parity * 4 + 1] = sb; bcast[parity * 4 + 2] = sm;
                bcast[parity * 4 + 3] = ok ? 1.0 : 0.0;
                if (order) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                // re-arm my granules two exchanges ahead (safe: everybody has consumed that buffer)
                const int nb = (epoch + 2u) % 3u;
                __hip_atomic_store(gran(nb, 0, wg), COOP_SENTINEL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(gran(nb, 1, wg), COOP_SENTINEL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(gran(nb, 2, wg), COOP_SENTINEL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        __syncthreads();
        a = bcast[parity * 4 + 0]; b = bcast[parity * 4 + 1]; mx = bcast[parity * 4 + 2];
        if (bcast[parity * 4 + 3] == 0.0) dead = true;
        parity ^= 1;
        ++epoch;
        const long long t4 = coop_clock();
        tm[1] += t1 - t0; tm[2] += t2 - t1; tm[3] += t3 - t2; tm[4] += t4 - t3; ++tm[5];
    }
    __device__ void barrier_ordered() {
        double a = 0.0, b = 0.0, c = 0.0;
        exchange(a, b, c, true);
    }

};

}  // namespace rdis_hip
