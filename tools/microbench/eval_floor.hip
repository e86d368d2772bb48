// eval_floor.hip -- the floor of ONE dependent evaluation of the cooperative solvers (solver_pipe.hpp / solver_coop.hpp), piece by
// piece, in microseconds of wall clock (HIP events around many repetitions; no assumption about the shader clock):
//   arith   one wave alone on its SIMD: clamp(base + a dir) of its factor's twelve variables, value + twelve partials
//           (factors.hpp, rounded like the reference's build: the cooperative solvers' default), partials times direction
//   reduce  the wave's two sums (value, slope) across its 64 lanes (solver_wg.hpp: wave_sum)
//   hop     a store on one compute unit seen by a polling load on another (relaxed agent-scope atomics, the granules of
//           grid_sync.hpp): half a ping-pong round trip, the nearest and the farthest partner tried
//   step    one step of the control logic on the reply (minimizer.hpp: CgdMachine, one lane, machine and request in LDS)
// A trial of a line search cannot take less than step + arith + reduce + hop (+ the sweep's adds): what bench.py prints as
// roofline.latency.floor_us next to the measured microseconds per evaluation.  Built by __graft_entry__.build() into
// tools/microbench/bin/eval_floor; prints one JSON line.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../rdis_amd/csrc -o bin/eval_floor eval_floor.hip
#define RDIS_FACTORS_NO_CONTRACT 1
#include <hip/hip_runtime.h>
#include <cfloat>
#include <cstdio>
#include "solver_wg.hpp"

using namespace rdis_hip;

__global__ void __launch_bounds__(64) arith_kernel(const double* __restrict__ in, double* __restrict__ out, int reps) {
    double base[12], dir[12], lo[12], hi[12];
    for (int k = 0; k < 12; ++k) { base[k] = in[k] + 1e-9 * threadIdx.x; dir[k] = in[12 + k]; lo[k] = base[k] - 1e3; hi[k] = base[k] + 1e3; }
    const double ox = in[24], oy = in[25];
    double a = 1e-7, acc = 0.0;
    for (int r = 0; r < reps; ++r) {
        double v[12], g[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) v[k] = clampd(base[k] + a * dir[k], lo[k], hi[k]);
        const double f = ba_eval_grad(v, ox, oy, g);
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < 12; ++k) s += g[k] * dir[k];
        acc += f;
        a = a + 1e-300 * (f + s);   // (the next trial step depends on this one's reply)
        opaque(a);
    }
    out[threadIdx.x] = acc + a;
}
__global__ void __launch_bounds__(64) reduce_kernel(double* __restrict__ out, int reps) {
    double x = 1.0 + 1e-9 * threadIdx.x, y = 2.0 - 1e-9 * threadIdx.x;
    for (int r = 0; r < reps; ++r) {
        const double sx = wave_sum(x), sy = wave_sum(y);
        x = sx * 1e-2 + 1e-9 * threadIdx.x; y = sy * 1e-2;
        opaque(x); opaque(y);
    }
    out[threadIdx.x] = x + y;
}
__global__ void pingpong_kernel(unsigned long long* flag, int rounds, int partner_block) {
    if (threadIdx.x != 0) return;
    if (blockIdx.x != 0 && blockIdx.x != partner_block) return;
    const bool first = blockIdx.x == 0;
    long long spins = 0;
    for (int r = 1; r <= rounds; ++r) {
        if (first) {
            __hip_atomic_store(flag, (unsigned long long)(2 * r - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (__hip_atomic_load(flag + 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned long long)(2 * r)) { if (++spins > (1ll << 26)) return; }
        } else {
            while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned long long)(2 * r - 1)) { if (++spins > (1ll << 26)) return; }
            __hip_atomic_store(flag + 16, (unsigned long long)(2 * r), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}
// the control logic on a cheap analytic line function, as run_machine steps it; counts its steps
__global__ void __launch_bounds__(64) step_kernel(long long* __restrict__ nsteps, double* __restrict__ out, int maxiters) {
    __shared__ CgdMachine M;
    __shared__ Request Q[2];
    if (threadIdx.x == 0) M.init(maxiters, 3e-8);
    __syncthreads();
    double r0 = 0.0, r1 = 0.0, r2 = 0.0;
    long long n = 0;
    double shift = 3.7e-5, curv = 1.0e6, level = 1.0e4;
    for (int round = 0; round < 1000000; ++round) {
        if (threadIdx.x == 0) step_machine(&M, &Q[round & 1], r0, r1, r2, true);
        __syncthreads();
        ++n;
        const Request& q = Q[round & 1];
        const int kind = __builtin_amdgcn_readfirstlane(q.kind);
        const double a = uniform(q.a);
        if (kind == REQ_DONE) break;
        const int flags = __builtin_amdgcn_readfirstlane(q.flags);
        if (kind == REQ_EVAL && (flags & RF_PRE_BEGIN)) { level -= 10.0; shift = 1.0e-5 * (1.0 + (double)(round % 13)); curv = 1.0e6 * (1.0 + 0.5 * (double)(round % 5)); }
        if (kind == REQ_EVAL) {
            const double u = a - shift;
            r0 = uniform(level + curv * u * u + 0.3 * u * u * u * u + 0.05 * u * u * u);
            r1 = uniform(2.0 * curv * u + 1.2 * u * u * u + 0.15 * u * u);
        } else if (kind == REQ_GRAD && (flags & RF_POST_REDUCE)) {
            r0 = 1.0; r1 = 1.0; r2 = 0.5;
        }
    }
    if (threadIdx.x == 0) { nsteps[0] = n; out[0] = M.fret; }
}

template <class F>
static double time_us(F launch, int warm = 1, int runs = 3) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    double best = 1e30;
    for (int i = 0; i < warm + runs; ++i) {
        hipEventRecord(e0);
        launch();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        if (i >= warm && ms * 1e3 < best) best = ms * 1e3;
    }
    return best;
}

int main() {
    double h_in[26] = {0.0157, -0.0127, -0.0044, -0.034, -0.107, 1.12, 399.75, -3.18e-7, 5.88e-13, -0.612, 0.572, -1.847,
                       1e-3, -2e-3, 5e-4, 1e-2, 2e-2, -1e-2, 1.0, 1e-9, -1e-15, 3e-2, -1e-2, 2e-2, -332.65, 262.09};
    double *d_in, *d_out; long long* d_n; unsigned long long* flag;
    hipMalloc(&d_in, sizeof h_in); hipMalloc(&d_out, 4096); hipMalloc(&d_n, 64); hipMalloc(&flag, 4096);
    hipMemcpy(d_in, h_in, sizeof h_in, hipMemcpyHostToDevice);
    const int REPS = 20000;
    // (a launch costs some microseconds: two repetition counts, the difference)
    const double a1 = time_us([&] { arith_kernel<<<1, 64>>>(d_in, d_out, REPS); }), a2 = time_us([&] { arith_kernel<<<1, 64>>>(d_in, d_out, 2 * REPS); });
    const double r1 = time_us([&] { reduce_kernel<<<1, 64>>>(d_out, REPS); }), r2 = time_us([&] { reduce_kernel<<<1, 64>>>(d_out, 2 * REPS); });
    double hop_min = 1e30, hop_max = 0.0;
    for (int partner : {1, 2, 7, 8, 9, 33, 64, 129, 255}) {
        const int R = 2000;
        auto run = [&](int rounds) { hipMemset(flag, 0, 4096); hipDeviceSynchronize(); return time_us([&] { hipMemsetAsync(flag, 0, 4096); pingpong_kernel<<<256, 64>>>(flag, rounds, partner); }, 0, 2); };
        const double t = (run(2 * R) - run(R)) / R / 2.0;   // per hop
        if (t < hop_min) hop_min = t;
        if (t > hop_max) hop_max = t;
    }
    long long n1 = 0, n2 = 0;
    const double s1 = time_us([&] { step_kernel<<<1, 64>>>(d_n, d_out, 100); });
    hipMemcpy(&n1, d_n, 8, hipMemcpyDeviceToHost);
    const double s2 = time_us([&] { step_kernel<<<1, 64>>>(d_n, d_out, 400); });
    hipMemcpy(&n2, d_n, 8, hipMemcpyDeviceToHost);
    const double arith = (a2 - a1) / REPS, reduce = (r2 - r1) / REPS, step = (s2 - s1) / (double)(n2 - n1);
    printf("{\"arith_us\": %.4f, \"reduce_us\": %.4f, \"hop_us_min\": %.4f, \"hop_us_max\": %.4f, \"step_us\": %.4f, \"floor_us\": %.4f, "
           "\"what\": \"one wave's factor arithmetic (value + 12 partials + slope) + its two wave sums + one store-to-load hop between compute units (nearest partner) + one step of the control logic\"}\n",
           arith, reduce, hop_min, hop_max, step, arith + reduce + hop_min + step);
    return 0;
}
