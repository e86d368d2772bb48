// microbenchmark: cycles per CgdMachine::next() step on a quadratic line function
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../rdis_amd/csrc/minimizer.hpp"
using namespace rdis_hip;
__global__ void __launch_bounds__(64) k(long long* out, int maxiters) {
    __shared__ CgdMachine M;
    long long t0 = clock64(), tn = 0; int steps = 0, evals = 0;
  for (int rep = 0; rep < 200; ++rep) {
    M.init(maxiters, 3e-8);
    double r0 = 0, r1 = 0, r2 = 0;
    double p = 1.0 + 0.01 * rep, xi = 0.0, g = 0.0, h = 0.0;  // 1-D problem f(x) = (x-3)^4 + x^2
    auto F = [](double x) { return (x - 3) * (x - 3) * (x - 3) * (x - 3) + x * x; };
    auto D = [](double x) { return 4 * (x - 3) * (x - 3) * (x - 3) + 2 * x; };
    for (;;) {
        long long a0 = clock64();
        Request q = M.next(r0, r1, r2);
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
    if (threadIdx.x == 0) { out[0] = clock64() - t0; out[1] = tn; out[2] = steps; out[3] = evals; out[4] = M.iter; }
}
int main() {
    long long* d; hipMalloc(&d, 64);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, 2000);
        long long h[8]; hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
        printf("total %lld cycles, next() %lld cycles over %lld steps (%lld evals, iter %lld) => %.0f cycles/step\n", h[0], h[1], h[2], h[3], h[4], (double)h[1] / h[2]);
    }
    return 0;
}
