// microbenchmark: cycles per CgdMachine::next() step (state in LDS vs registers), one wave
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../rdis_amd/csrc/minimizer.hpp"
using namespace rdis_hip;
template <bool LDS>
__global__ void __launch_bounds__(64) k(long long* out, int maxiters) {
    __shared__ CgdMachine Ms;
    __shared__ Request Qs[2];
    CgdMachine Mr;
    long long t0 = clock64(), tn = 0; int steps = 0, evals = 0;
  for (int rep = 0; rep < 200; ++rep) {
    if (LDS) Ms.init(maxiters, 3e-8); else Mr.init(maxiters, 3e-8);
    double r0 = 0, r1 = 0, r2 = 0;
    double p = 1.0 + 0.01 * rep, xi = 0.0, g = 0.0, h = 0.0;  // f(x) = (x-3)^4 + x^2
    auto F = [](double x) { return (x - 3) * (x - 3) * (x - 3) * (x - 3) + x * x; };
    auto D = [](double x) { return 4 * (x - 3) * (x - 3) * (x - 3) + 2 * x; };
    for (int round = 0;; ++round) {
        long long a0 = clock64();
        Request q;
        if (LDS) { step_machine(&Ms, &Qs[round & 1], r0, r1, r2); __syncthreads(); q = Qs[round & 1]; }
        else q = Mr.next(r0, r1, r2);
        tn += clock64() - a0; ++steps;
        if (q.kind == REQ_DONE) break;
        switch (q.kind) {
            case REQ_F: r0 = F(p + q.a * xi); ++evals; break;
            case REQ_FD: r0 = F(p + q.a * xi); r1 = D(p + q.a * xi) * xi; ++evals; break;
            case REQ_GRAD: xi = D(p); break;
            case REQ_CG_START: g = -xi; h = g; xi = g; break;
            case REQ_LINE_END: xi *= q.a; p += xi; break;
            case REQ_CG_REDUCE: r0 = fabs(xi) * fmax(fabs(p), 1.0) / fmax(fabs(q.a), 1.0); r1 = g * g; r2 = (xi + g) * xi; break;
            case REQ_CG_UPDATE: g = -xi; h = g + q.a * h; xi = h; break;
            default: break;
        }
    }
  }
    if (threadIdx.x == 0) { out[0] = clock64() - t0; out[1] = tn; out[2] = steps; out[3] = evals; }
}
int main() {
    long long* d; hipMalloc(&d, 64);
    for (int rep = 0; rep < 2; ++rep) {
        long long h[8];
        hipLaunchKernelGGL(k<true>, dim3(1), dim3(64), 0, 0, d, 2000);
        hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
        printf("LDS  state: next() %.0f cycles/step (%lld steps)\n", (double)h[1] / h[2], h[2]);
        hipLaunchKernelGGL(k<false>, dim3(1), dim3(64), 0, 0, d, 2000);
        hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
        printf("regs state: next() %.0f cycles/step (%lld steps)\n", (double)h[1] / h[2], h[2]);
    }
    return 0;
}
