// wave_costs.hip -- cost model of a single gfx950 wave for latency-bound code (the solver's
// control step and exchange): cycles per dependent / independent fp64 op, select, LDS round
// trip, wave reduction, workgroup barrier, agent-scope store->load visibility.
//   hipcc --offload-arch=gfx950 -O3 -o wave_costs wave_costs.hip && ./wave_costs
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP 256

__device__ __forceinline__ long long now() { return clock64(); }

__device__ __forceinline__ double wave_sum_dpp(double v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__global__ void costs(long long* out, double* sink, double seed) {
    __shared__ double lds[512];
    const int tid = threadIdx.x;
    lds[tid] = seed + tid; lds[tid + 256] = seed;
    __syncthreads();
    double x = seed + tid * 1e-9, y = seed * 0.5, z = 1.0000001;
    long long t0, t1;
    int k = 0;
    // 0: dependent fp64 fma chain
    t0 = now();
#pragma unroll
    for (int i = 0; i < REP; ++i) { x = __builtin_fma(x, z, y); asm volatile("" : "+v"(x)); }
    t1 = now(); if (tid == 0) out[k] = t1 - t0; ++k;
    // 1: 4 independent fma chains
    double a0 = x, a1 = x + 1, a2 = x + 2, a3 = x + 3;
    t0 = now();
#pragma unroll
    for (int i = 0; i < REP / 4; ++i) { a0 = __builtin_fma(a0, z, y); a1 = __builtin_fma(a1, z, y); a2 = __builtin_fma(a2, z, y); a3 = __builtin_fma(a3, z, y); asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)); }
    t1 = now(); if (tid == 0) out[k] = t1 - t0; ++k;
    x = a0 + a1 + a2 + a3;
    // 2: dependent 64-bit selects
    t0 = now();
#pragma unroll
    for (int i = 0; i < REP; ++i) { x = (x > y) ? z : x; asm volatile("" : "+v"(x), "+v"(y), "+v"(z)); }
    t1 = now(); if (tid == 0) out[k] = t1 - t0; ++k;
    // 2b: 256 selects in 4 independent chains
    {
        double b0 = x, b1 = x + 1, b2 = x + 2, b3 = x + 3;
        t0 = now();
#pragma unroll
        for (int i = 0; i < REP / 4; ++i) {
            b0 = (b0 > y) ? z : b0; b1 = (b1 > y) ? z : b1; b2 = (b2 > y) ? z : b2; b3 = (b3 > y) ? z : b3;
            asm volatile("" : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(y), "+v"(z));
        }
        t1 = now(); if (tid == 0) out[10] = t1 - t0;
        x += b0 + b1 + b2 + b3;
    }
    // 3: dependent fp64 divisions (x16)
    t0 = now();
#pragma unroll
    for (int i = 0; i < 16; ++i) { x = y / (x + 3.0); asm volatile("" : "+v"(x)); }
    t1 = now(); if (tid == 0) out[k] = t1 - t0; ++k;
    // 4: LDS dependent round trips (x64)
    int idx = tid;
    t0 = now();
#pragma unroll
    for (int i = 0; i < 64; ++i) { idx = (int)lds[(idx & 255)] & 255; }
    t1 = now(); if (tid == 0) out[k] = t1 - t0; ++k;
    x += idx;
    // 5: wave_sum of a double (x16)
    t0 = now();
#pragma unroll
    for (int i = 0; i < 16; ++i) { x = wave_sum_dpp(x) * 1e-3; asm volatile("" : "+v"(x)); }
    t1 = now(); if (tid == 0) out[k] = t1 - t0; ++k;
    // 6: workgroup barrier (x64)
    t0 = now();
#pragma unroll
    for (int i = 0; i < 64; ++i) __syncthreads();
    t1 = now(); if (tid == 0) out[k] = t1 - t0; ++k;
    // 7: sincos (x16, dependent)
    t0 = now();
#pragma unroll
    for (int i = 0; i < 16; ++i) { double s, c; sincos(x, &s, &c); x = s + c; asm volatile("" : "+v"(x)); }
    t1 = now(); if (tid == 0) out[k] = t1 - t0; ++k;
    // 8: s_memtime overhead (x64)
    t0 = now();
    long long acc = 0;
#pragma unroll
    for (int i = 0; i < 64; ++i) acc += now();
    t1 = now(); if (tid == 0) out[k] = t1 - t0; ++k;
    // 9: sqrt (x16 dependent)
    t0 = now();
#pragma unroll
    for (int i = 0; i < 16; ++i) { x = sqrt(x * x + 1.0); asm volatile("" : "+v"(x)); }
    t1 = now(); if (tid == 0) out[k] = t1 - t0; ++k;
    sink[blockIdx.x * blockDim.x + tid] = x + y + z + (double)acc;
}

// store -> remote visibility ping-pong between two workgroups (different CUs, possibly XCDs)
__global__ void pingpong(unsigned long long* flag, long long* out, int rounds, int partner_block) {
    if (threadIdx.x != 0) return;
    if (blockIdx.x != 0 && blockIdx.x != partner_block) return;
    const bool first = blockIdx.x == 0;
    long long t0 = clock64(), spins = 0;
    for (int r = 1; r <= rounds; ++r) {
        if (first) {
            __hip_atomic_store(flag, (unsigned long long)(2 * r - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (__hip_atomic_load(flag + 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned long long)(2 * r)) { if (++spins > (1ll << 26)) return; }
        } else {
            while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned long long)(2 * r - 1)) { if (++spins > (1ll << 26)) return; }
            __hip_atomic_store(flag + 16, (unsigned long long)(2 * r), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (first) out[0] = clock64() - t0;
}

int main() {
    long long* out; double* sink; unsigned long long* flag;
    hipMalloc(&out, 64 * sizeof(long long)); hipMalloc(&sink, 1024 * sizeof(double)); hipMalloc(&flag, 4096);
    hipMemset(out, 0, 64 * sizeof(long long));
    const char* names[] = {"256 dependent fp64 fma", "256 fp64 fma in 4 chains", "256 dependent 64-bit selects", "16 dependent fp64 div",
                           "64 dependent LDS round trips", "16 wave_sum(double)", "64 workgroup barriers", "16 dependent sincos",
                           "64 clock reads", "16 dependent sqrt", "256 selects in 4 chains"};
    for (int threads : {64, 256}) {
        costs<<<1, threads>>>(out, sink, 1.25);
        hipDeviceSynchronize();
        long long h[16];
        hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
        printf("-- %d threads per workgroup (one wave per SIMD)\n", threads);
        for (int i = 0; i < 11; ++i) printf("%-32s %8lld ticks\n", names[i], h[i]);
    }
    // clock64 ticks at 100 MHz on gfx9?  calibrate against wall time
    {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int partner : {1, 2, 8, 9, 33, 64, 129}) {
            hipMemset(flag, 0, 4096);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            pingpong<<<256, 64>>>(flag, out, 2000, partner);
            hipEventRecord(e1);
            hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            long long h; hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost);
            printf("ping-pong block 0 <-> block %3d: %.3f us per round trip (2 hops), %lld ticks/round trip\n", partner, ms * 1e3 / 2000, h / 2000);
        }
    }
    return 0;
}
