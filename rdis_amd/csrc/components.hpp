// components.hpp -- interface of components.hip (connected components of the residual factor
// graph, device side).  Raw device pointers in, host lists out.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <vector>

namespace rdis_hip {

struct ComponentLists {
    int64_t ncomp = 0, nfree = 0, nfac = 0;
    std::vector<int64_t> free_ptr, free_vid, fac_ptr, fac_id;
};

// device scratch of the labelling, kept by the caller between calls (grown on demand)
struct CcWorkspace {
    void* dev = nullptr;
    size_t bytes = 0;
    CcWorkspace() = default;
    CcWorkspace(const CcWorkspace&) = delete;
    CcWorkspace& operator=(const CcWorkspace&) = delete;
    ~CcWorkspace();
};

// kind 0: bundle adjustment (cam / pt: first variable id of each factor's camera and point block),
// otherwise nonlinear products (rowptr / vid: CSR of the factors' variables).  assigned_dev[N]:
// non-zero = the variable is assigned (fixed).  Returns 0 or a hipError_t.
int device_components(hipStream_t stream, int kind, int N, int F, const int* cam, const int* pt, const int* rowptr,
                      const int* vid, const unsigned char* assigned_dev, CcWorkspace* ws, ComponentLists* out);

}  // namespace rdis_hip
