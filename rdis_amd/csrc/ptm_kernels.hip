// ptm_kernels.hip -- the point-major streaming solver's kernels (solver_ptm.hpp) and their launches, a translation
// unit of their own (they are the library's largest kernels; rdis_hip.hip sees them through ptm_api.hpp).
#include "solver_ptm.hpp"

namespace rdis_hip {

template <int ROT>
static hipError_t ptm_launch_rot(int threads, int grid, size_t dyn, hipStream_t stream, const ProblemView& P, const PlanView& V,
                                 int maxiters, double ftol, int ncb_cap) {
#define RDIS_PTM_LAUNCH(T)                                                                                                      \
    do {                                                                                                                        \
        if (dyn > 48 * 1024) {                                                                                                  \
            hipError_t e = hipFuncSetAttribute((const void*)cgd_ptm_kernel<T, ROT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn); \
            if (e != hipSuccess) return e;                                                                                      \
        }                                                                                                                       \
        cgd_ptm_kernel<T, ROT><<<grid, T, dyn, stream>>>(P, V, maxiters, ftol, ncb_cap);                                         \
    } while (0)
    switch (threads) {
        case 256: RDIS_PTM_LAUNCH(256); break;
        case 512: RDIS_PTM_LAUNCH(512); break;
        default: RDIS_PTM_LAUNCH(768); break;
    }
#undef RDIS_PTM_LAUNCH
    return hipGetLastError();
}

hipError_t ptm_launch(int rot, int threads, int grid, size_t dyn, hipStream_t stream, const ProblemView& P, const PlanView& V,
                      int maxiters, double ftol, int ncb_cap) {
    switch (rot) {
        case ROT_CAMFIX: return ptm_launch_rot<ROT_CAMFIX>(threads, grid, dyn, stream, P, V, maxiters, ftol, ncb_cap);
        default: return ptm_launch_rot<ROT_RECORDS>(threads, grid, dyn, stream, P, V, maxiters, ftol, ncb_cap);
    }
}

template <int ROT>
static const void* ptmg_kernel_ptr(int threads, bool wide, bool local) {
    if (wide && local) return (const void*)cgd_ptmg_kernel<PTM_WIDE_THREADS, ROT, true, true>;
    if (wide) return (const void*)cgd_ptmg_kernel<PTM_WIDE_THREADS, ROT, true>;
    switch (threads) {
        case 256: return (const void*)cgd_ptmg_kernel<256, ROT>;
        case 512: return (const void*)cgd_ptmg_kernel<512, ROT>;
        default: return (const void*)cgd_ptmg_kernel<768, ROT>;
    }
}
const void* ptmg_kernel_fn(int rot, int threads, bool wide, bool local) {
    switch (rot) {
        case ROT_CAMFIX: return ptmg_kernel_ptr<ROT_CAMFIX>(threads, wide, local);
        default: return ptmg_kernel_ptr<ROT_RECORDS>(threads, wide, local);
    }
}

hipError_t ptm_gather_launch(int grid, hipStream_t stream, int n, const int* jg, const unsigned* fidx, const double2* fobs,
                             short* pcam, double2* pobs) {
    ptm_gather_kernel<<<grid, 256, 0, stream>>>(n, jg, fidx, fobs, pcam, pobs);
    return hipGetLastError();
}

}  // namespace rdis_hip
