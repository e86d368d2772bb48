// refround_kernels.hip -- see refround_api.hpp: solver_pipe.hpp / solver_coop.hpp / solver_lds.hpp compiled once more, in
// namespace rdis_hip_refround, with the factor arithmetic's contraction off.
#define RDIS_FACTORS_NO_CONTRACT 1
#define RDIS_REFERENCE_SLOPE 1      // solver_lds.hpp: a trial's slope as gradient times direction, like Df1dim::df
#define rdis_hip rdis_hip_refround
#include "solver_lds.hpp"
#include "solver_pipe.hpp"
#undef rdis_hip
#include "refround_api.hpp"

namespace rdis_hip {
namespace rr = ::rdis_hip_refround;

int refround_launch_pipe(hipStream_t stream, int kind, const void* P, const void* V, const void* first_group, const void* groups,
                         const int* wg_group, int ngroups, int total_wg, int maxiters, double ftol) {
    return rr::launch_pipe(stream, kind, *static_cast<const rr::ProblemView*>(P), *static_cast<const rr::PlanView*>(V),
                           *static_cast<const rr::CoopGroup*>(first_group), static_cast<const rr::CoopGroup*>(groups), wg_group,
                           ngroups, total_wg, maxiters, ftol);
}
int refround_pipe_max_workgroups(int num_cus) { return rr::pipe_max_workgroups(num_cus); }
int refround_launch_coop(hipStream_t stream, int kind, const void* P, const void* V, const void* first_group, const void* groups,
                         const int* wg_group, int ngroups, int total_wg, int threads, int maxiters, double ftol) {
    return rr::launch_coop(stream, kind, *static_cast<const rr::ProblemView*>(P), *static_cast<const rr::PlanView*>(V),
                           *static_cast<const rr::CoopGroup*>(first_group), static_cast<const rr::CoopGroup*>(groups), wg_group,
                           ngroups, total_wg, threads, maxiters, ftol);
}
int refround_coop_max_workgroups(int threads, int num_cus) { return rr::coop_max_workgroups(threads, num_cus); }

template <int ROT, bool STALE = false>
static hipError_t lds_rot(int threads, int grid, size_t dyn, hipStream_t stream, const rr::ProblemView& P, const rr::PlanView& V,
                          int maxiters, double ftol, int nsc, int ncc, int chc) {
#define RDIS_RR_LDS_LAUNCH(T)                                                                                                   \
    do {                                                                                                                        \
        if (dyn > 48 * 1024) {                                                                                                  \
            hipError_t e = hipFuncSetAttribute((const void*)rr::cgd_lds_kernel<T, ROT, STALE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn); \
            if (e != hipSuccess) return e;                                                                                      \
        }                                                                                                                       \
        rr::cgd_lds_kernel<T, ROT, STALE><<<grid, T, dyn, stream>>>(P, V, maxiters, ftol, nsc, ncc, chc);                        \
    } while (0)
    switch (threads) {
        case 64: RDIS_RR_LDS_LAUNCH(64); break;
        case 128: RDIS_RR_LDS_LAUNCH(128); break;
        case 256: RDIS_RR_LDS_LAUNCH(256); break;
        case 512: RDIS_RR_LDS_LAUNCH(512); break;
        case 768: RDIS_RR_LDS_LAUNCH(768); break;
        default: RDIS_RR_LDS_LAUNCH(1024); break;
    }
#undef RDIS_RR_LDS_LAUNCH
    return hipGetLastError();
}
hipError_t refround_launch_lds(int rot, int stale, int threads, int grid, size_t dyn, hipStream_t stream, const void* Pv, const void* Vv,
                               int maxiters, double ftol, int ns_cap, int ncb_cap, int chunk_cap) {
    const rr::ProblemView& P = *static_cast<const rr::ProblemView*>(Pv);
    const rr::PlanView& V = *static_cast<const rr::PlanView*>(Vv);
    // (the stale-cache emulation is instantiated for per-factor rotations only, like the default rounding's)
    if (stale) return lds_rot<rr::ROT_PER_FACTOR, true>(threads, grid, dyn, stream, P, V, maxiters, ftol, ns_cap, ncb_cap, chunk_cap);
    switch (rot) {
        case rr::ROT_CAMFIX: return lds_rot<rr::ROT_CAMFIX>(threads, grid, dyn, stream, P, V, maxiters, ftol, ns_cap, ncb_cap, chunk_cap);
        case rr::ROT_RECORDS: return lds_rot<rr::ROT_RECORDS>(threads, grid, dyn, stream, P, V, maxiters, ftol, ns_cap, ncb_cap, chunk_cap);
        default: return lds_rot<rr::ROT_PER_FACTOR>(threads, grid, dyn, stream, P, V, maxiters, ftol, ns_cap, ncb_cap, chunk_cap);
    }
}


// per-factor values / twelve partials in this rounding (rdis_hip_eval_each / rdis_hip_grad_each_ba after
// rdis_hip_set_factor_rounding(problem, 1): what the parity tests compare with the oracle's device arithmetic, ==)
namespace {
__global__ void __launch_bounds__(256)
rr_eval_each_kernel(rr::ProblemView P, int nf, const int* __restrict__ fac, double* __restrict__ out) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nf; i += gridDim.x * blockDim.x) {
        double f, s;
        rr::factor_value<rr::KIND_BA, false>(P, nullptr, fac ? fac[i] : i, f, s);
        out[i] = f;
    }
}
__global__ void __launch_bounds__(256)
rr_grad_each_kernel(rr::ProblemView P, int nf, const int* __restrict__ fac, double* __restrict__ out12) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nf; i += gridDim.x * blockDim.x) {
        const int fid = fac ? fac[i] : i;
        const int c = P.cam[fid], q = P.pt[fid];
        const double2 o = P.obs[fid];
        double v[12], g[12];
#pragma unroll
        for (int k = 0; k < 9; ++k) v[k] = P.x[c + k];
#pragma unroll
        for (int k = 0; k < 3; ++k) v[9 + k] = P.x[q + k];
        rr::ba_eval_grad(v, o.x, o.y, g);
#pragma unroll
        for (int k = 0; k < 12; ++k) out12[12ll * i + k] = g[k];
    }
}
}  // namespace
hipError_t refround_eval_each(int grid, hipStream_t stream, const void* Pv, int nf, const int* fac, double* out) {
    rr_eval_each_kernel<<<grid, 256, 0, stream>>>(*static_cast<const rr::ProblemView*>(Pv), nf, fac, out);
    return hipGetLastError();
}
hipError_t refround_grad_each(int grid, hipStream_t stream, const void* Pv, int nf, const int* fac, double* out12) {
    rr_grad_each_kernel<<<grid, 256, 0, stream>>>(*static_cast<const rr::ProblemView*>(Pv), nf, fac, out12);
    return hipGetLastError();
}

}  // namespace rdis_hip
