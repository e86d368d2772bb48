// Does a wave whose EXEC mask leaves whole 16-lane quarters empty issue its vector instructions
// faster?  (The solver's control step runs on one wave and is bound by instruction issue.)
//   hipcc --offload-arch=gfx950 -O3 -o exec_mask exec_mask.hip && ./exec_mask
#include <hip/hip_runtime.h>
#include <cstdio>

template <int CHAINS>
__global__ void k(double* out, long long* cyc, int active_lanes, int iters) {
    const int lane = threadIdx.x;
    double acc[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) acc[c] = 1.0 + lane + c;
    const double m = 1.0000001, a = 1e-9;
    long long t0 = 0, t1 = 0;
    if (lane < active_lanes) {
        t0 = clock64();
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_fma(acc[c], m, a);
        }
        t1 = clock64();
    }
    double s = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) s += acc[c];
    out[blockIdx.x * 64 + lane] = s;
    if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int CHAINS>
void run(const char* what) {
    double* out; long long* cyc;
    hipMalloc(&out, 64 * sizeof(double)); hipMalloc(&cyc, sizeof(long long));
    for (int active : {64, 32, 16, 1}) {
        k<CHAINS><<<1, 64>>>(out, cyc, active, 4096);
        k<CHAINS><<<1, 64>>>(out, cyc, active, 4096);
        hipDeviceSynchronize();
        long long c; hipMemcpy(&c, cyc, sizeof(c), hipMemcpyDeviceToHost);
        printf("%-28s active lanes %2d: %.2f clock64 ticks per fma\n", what, active, (double)c / (4096.0 * CHAINS));
    }
}
int main() {
    run<1>("1 dependent chain");
    run<8>("8 independent chains");
    return 0;
}
