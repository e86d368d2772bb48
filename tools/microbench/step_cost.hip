// step_cost.hip -- cycles per step of the solver's control machine (rdis_amd/csrc/minimizer.hpp)
// in isolation: one wave steps CgdMachine::next on a cheap analytic line function, the way
// run_machine does (machine + request in LDS), and the time of every step is attributed to the
// state it started in.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../rdis_amd/csrc -o step_cost step_cost.hip
#include <hip/hip_runtime.h>
#include <cfloat>
#include <cstdio>
#include "minimizer.hpp"

using namespace rdis_hip;

__global__ void __launch_bounds__(64) drive(long long* cyc, long long* cnt, double* out, int maxiters) {
    __shared__ CgdMachine M;
    __shared__ Request Q[2];
    if (threadIdx.x == 0) M.init(maxiters, 3e-8);
    __syncthreads();
    double r0 = 0.0, r1 = 0.0, r2 = 0.0;
    long long c_hot = 0, n_hot = 0, c_other = 0, n_other = 0;
    double shift = 3.7e-5, curv = 1.0e6, level = 1.0e4;   // `level` drops with every line: the CG loop goes on
    for (int round = 0; round < 100000; ++round) {
        const int st0 = __builtin_amdgcn_readfirstlane(M.st);
        const long long t0 = clock64();
        if (threadIdx.x == 0) step_machine(&M, &Q[round & 1], r0, r1, r2, true);   // one lane: see run_machine
        const long long t1 = clock64();
        __syncthreads();
        if (st0 == CgdMachine::S_DB_EVAL) { c_hot += t1 - t0; ++n_hot; } else { c_other += t1 - t0; ++n_other; }
        const Request& q = Q[round & 1];
        const int kind = __builtin_amdgcn_readfirstlane(q.kind);
        const double a = uniform(q.a);
        if (kind == REQ_DONE) break;
        const int flags = __builtin_amdgcn_readfirstlane(q.flags);
        if (kind == REQ_EVAL && (flags & RF_PRE_BEGIN)) { level -= 10.0; shift = 1.0e-5 * (1.0 + (double)(round % 13)); curv = 1.0e6 * (1.0 + 0.5 * (double)(round % 5)); }
        if (kind == REQ_EVAL) {
            // a quartic with a single minimum at `shift`
            const double u = a - shift;
            r0 = uniform(level + curv * u * u + 0.3 * u * u * u * u + 0.05 * u * u * u);
            r1 = uniform(2.0 * curv * u + 1.2 * u * u * u + 0.15 * u * u);
        } else if (kind == REQ_GRAD && (flags & RF_POST_REDUCE)) {
            r0 = 1.0; r1 = 1.0; r2 = 0.5;
        }
    }
    if (threadIdx.x == 0) {
        cyc[0] = c_hot; cnt[0] = n_hot; cyc[1] = c_other; cnt[1] = n_other;
        out[0] = M.fret; out[1] = (double)M.nfeval; out[2] = (double)M.iter;
    }
}

int main() {
    long long *cyc, *cnt; double* out;
    hipMalloc(&cyc, 64); hipMalloc(&cnt, 64); hipMalloc(&out, 64);
    for (int rep = 0; rep < 2; ++rep) {
        drive<<<1, 64>>>(cyc, cnt, out, 200);
        hipDeviceSynchronize();
        long long hc[8], hn[2]; double ho[3];
        hipMemcpy(hc, cyc, 64, hipMemcpyDeviceToHost); hipMemcpy(hn, cnt, 16, hipMemcpyDeviceToHost);
        hipMemcpy(ho, out, 24, hipMemcpyDeviceToHost);
        printf("run %d: Brent-reply steps %lld, %.0f cycles each; other steps %lld, %.0f cycles each (fret %.6f nfeval %.0f iters %.0f)\n",
               rep, hn[0], (double)hc[0] / (hn[0] ? hn[0] : 1), hn[1], (double)hc[1] / (hn[1] ? hn[1] : 1), ho[0], ho[1], ho[2]);
    }
    return 0;
}
