// simd_fp64.hip -- what a SIMD of gfx950 gives to 1, 2, 3, 4 co-resident waves of fp64 arithmetic: wall cycles per
// wave-instruction when every wave runs C independent chains of dependent fp64 operations (fma / mul / add mixes).
//   hipcc --offload-arch=gfx950 -O3 -o simd_fp64 simd_fp64.hip && ./simd_fp64
// One workgroup of 256 w lanes (w waves per SIMD) per compute unit; cycles from the barrier before to the barrier after.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP 2048

template <int C, int MIX>
__global__ void run(long long* out, double* sink, double seed) {
    double x[C];
    const double z = 1.0000001 + seed * 1e-12, y = seed * 0.5, u = 0.999999;
#pragma unroll
    for (int c = 0; c < C; ++c) x[c] = seed + threadIdx.x * 1e-9 + c;
    __syncthreads();
    const long long t0 = clock64();
    for (int i = 0; i < REP / 8; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
#pragma unroll
            for (int c = 0; c < C; ++c) {
                if (MIX == 0) x[c] = __builtin_fma(x[c], z, y);
                else if (MIX == 1) x[c] = (j & 1) ? x[c] * z : x[c] + y;
                else x[c] = __builtin_fma(x[c], x[c], u);   // (two reads of one register)
            }
#pragma unroll
            for (int c = 0; c < C; ++c) asm volatile("" : "+v"(x[c]));
        }
    }
    __syncthreads();
    const long long t1 = clock64();
    double s = 0.0;
#pragma unroll
    for (int c = 0; c < C; ++c) s += x[c];
    if (s == 12345.678) sink[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}

template <int C, int MIX>
static void go(const char* name, long long* d_out, double* d_sink) {
    for (int w = 1; w <= 4; ++w) {
        run<C, MIX><<<256, 256 * w>>>(d_out, d_sink, 1.5);
        hipDeviceSynchronize();
        long long c = 0;
        hipMemcpy(&c, d_out, sizeof(c), hipMemcpyDeviceToHost);
        // per SIMD: w waves x C chains x REP instructions
        std::printf("%-28s chains %d  waves/SIMD %d: %8lld cycles = %.2f per wave-instruction of the SIMD, %.2f per instruction of a wave\n", name, C, w,
                    c, (double)c / ((double)w * C * REP), (double)c / ((double)C * REP));
    }
}

int main() {
    long long* d_out; double* d_sink;
    hipMalloc(&d_out, 64); hipMalloc(&d_sink, 64);
    go<1, 0>("fma (dependent)", d_out, d_sink);
    go<2, 0>("fma", d_out, d_sink);
    go<4, 0>("fma", d_out, d_sink);
    go<1, 1>("mul / add alternating", d_out, d_sink);
    go<2, 1>("mul / add alternating", d_out, d_sink);
    go<4, 1>("mul / add alternating", d_out, d_sink);
    go<2, 2>("fma x*x+u", d_out, d_sink);
    return 0;
}
