// refround_api.hpp -- the solvers that BASELINE's configs 3 - 5 reach (the cooperative ones for ladybug as one component,
// the LDS-resident batch solver for ladybug 5 / 30 and the synthetic 3 x 40 components) a second time, with the factor
// arithmetic rounded like the reference's build: every product rounded before it is added, no fused multiply-add (g++ -O2 on
// x86-64 emits none; BundleAdjustmentFactor.cpp:160-185, 266-335, 351-554).  Plan option "factor_rounding" = 1 selects them.
// Why it exists: 25 unconverged CG iterations are a chaotic map of the start, and the DISTRIBUTION of end values over
// one-ulp starts depends on the evaluator's rounding -- the oracle compiled with contraction parts from itself with KS 0.21
// (DESIGN.md section 6); with this option the device's population is compared with the reference-faithful oracle's on equal terms.
// The kernels are the same headers compiled in a namespace of their own with -DRDIS_FACTORS_NO_CONTRACT (refround_kernels.hip);
// the views are passed as untyped pointers (the same structs, another namespace).
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>

namespace rdis_hip {

int refround_launch_pipe(hipStream_t stream, int kind, const void* P, const void* V, const void* first_group, const void* groups,
                         const int* wg_group, int ngroups, int total_wg, int maxiters, double ftol);
int refround_pipe_max_workgroups(int num_cus);
int refround_launch_coop(hipStream_t stream, int kind, const void* P, const void* V, const void* first_group, const void* groups,
                         const int* wg_group, int ngroups, int total_wg, int threads, int maxiters, double ftol);
int refround_coop_max_workgroups(int threads, int num_cus);
hipError_t refround_eval_each(int grid, hipStream_t stream, const void* P, int nf, const int* fac, double* out);
hipError_t refround_grad_each(int grid, hipStream_t stream, const void* P, int nf, const int* fac, double* out12);
hipError_t refround_launch_lds(int rot, int stale, int threads, int grid, size_t dyn, hipStream_t stream, const void* P, const void* V,
                               int maxiters, double ftol, int ns_cap, int ncb_cap, int chunk_cap);

}  // namespace rdis_hip
