// components.hip -- connected components of the residual factor graph, on the device.
//
// RDIS assigns a block of variables and then solves every connected component of what is left
// independently (reference src/Component.cpp:508-549 createChildren; the members come from its
// dynamic connectivity structure, ConnectivityGraph.h:255-261, an Euler-tour forest updated edge
// by edge).  The graph is bipartite: a factor is adjacent to its variables, an assigned variable
// has no edges.  For a batch solve the labelling is a static problem and is data-parallel:
//
//   1. lock-free union-find over the variables: one lane per factor joins the factor's
//      unassigned variables, hooking the larger root under the smaller, so that the root of a
//      component is its smallest variable id (a canonical label);
//   2. every variable / factor looks up its root; roots are counted;
//   3. components are ordered by (number of variables, smallest variable id) -- the reference
//      processes smaller components first (ComponentComparator, Component.cpp:603-608); among
//      equal sizes its order follows the internals of the Euler-tour structure, which is not a
//      function of the graph, so the smallest id is used;
//   4. stable radix sorts by component index give the member lists in ascending id order
//      (Component.cpp:78-79: variable and factor lists are sorted).
//
// Output convention (what rdis_hip_plan_create consumes): free_ptr / free_vid, fac_ptr / fac_id.
// Factors without an unassigned variable belong to no component; an unassigned variable that no
// factor touches is a component of its own with an empty factor list.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <climits>
#include <cstdint>

#include "components.hpp"

namespace rdis_hip {
namespace {

constexpr int KIND_BA_ = 0;

// parent[] is read and written with agent-scope relaxed atomics: a plain load could keep hitting
// a stale line of this CU's L1 after another CU hooked the root, and the retry loop would spin
__device__ __forceinline__ int uf_load(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ int uf_find(int* parent, int x) {
    // path halving; concurrent hooks only ever lower a parent, an older value is still an ancestor
    for (;;) {
        const int p = uf_load(parent + x);
        if (p == x) return x;
        const int gp = uf_load(parent + p);
        if (gp != p) __hip_atomic_store(parent + x, gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        x = p;
    }
}

__device__ __forceinline__ void uf_union(int* parent, int a, int b) {
    for (;;) {
        a = uf_find(parent, a);
        b = uf_find(parent, b);
        if (a == b) return;
        if (a < b) { const int t = a; a = b; b = t; }  // a is the larger root: hook it under b
        if (atomicCAS(&parent[a], a, b) == a) return;
    }
}

__global__ void __launch_bounds__(256)
cc_init_kernel(int N, int* parent, int* count) {
    for (int v = blockIdx.x * blockDim.x + threadIdx.x; v < N; v += gridDim.x * blockDim.x) { parent[v] = v; count[v] = 0; }
}

// one lane per factor: join its unassigned variables
__global__ void __launch_bounds__(256)
cc_union_kernel(int kind, int F, const int* __restrict__ cam, const int* __restrict__ pt,
                const int* __restrict__ rowptr, const int* __restrict__ vid,
                const unsigned char* __restrict__ assigned, int* parent) {
    for (int f = blockIdx.x * blockDim.x + threadIdx.x; f < F; f += gridDim.x * blockDim.x) {
        int first = -1;
        if (kind == KIND_BA_) {
            const int c = cam[f], q = pt[f];
#pragma unroll
            for (int k = 0; k < 12; ++k) {
                const int v = k < 9 ? c + k : q + (k - 9);
                if (assigned[v]) continue;
                if (first < 0) first = v; else uf_union(parent, first, v);
            }
        } else {
            for (int k = rowptr[f]; k < rowptr[f + 1]; ++k) {
                const int v = vid[k];
                if (assigned[v]) continue;
                if (first < 0) first = v; else uf_union(parent, first, v);
            }
        }
    }
}

// label[v] = root (smallest id of its component), INT_MAX for an assigned variable; roots counted
__global__ void __launch_bounds__(256)
cc_label_vars_kernel(int N, const unsigned char* __restrict__ assigned, int* parent, int* label, int* count) {
    for (int v = blockIdx.x * blockDim.x + threadIdx.x; v < N; v += gridDim.x * blockDim.x) {
        if (assigned[v]) { label[v] = INT_MAX; continue; }
        const int r = uf_find(parent, v);
        label[v] = r;
        atomicAdd(&count[r], 1);
    }
}

// sort key of a root: (number of variables, root); everything else sorts behind
__global__ void __launch_bounds__(256)
cc_root_keys_kernel(int N, const int* __restrict__ label, const int* __restrict__ count,
                    unsigned long long* key, int* val) {
    for (int v = blockIdx.x * blockDim.x + threadIdx.x; v < N; v += gridDim.x * blockDim.x) {
        const bool root = label[v] == v;
        key[v] = root ? (((unsigned long long)(unsigned)count[v] << 32) | (unsigned)v) : ~0ull;
        val[v] = v;
    }
}

// comp_of_root[root] = rank in the sorted order; free_ptr from the sorted counts
__global__ void __launch_bounds__(256)
cc_rank_kernel(int ncomp, const int* __restrict__ sorted_root, const int* __restrict__ count,
               int* comp_of_root, int* comp_nvars) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < ncomp; i += gridDim.x * blockDim.x) {
        const int r = sorted_root[i];
        comp_of_root[r] = i;
        comp_nvars[i] = count[r];
    }
}

__global__ void __launch_bounds__(256)
cc_var_keys_kernel(int N, const int* __restrict__ label, const int* __restrict__ comp_of_root, int* key, int* val) {
    for (int v = blockIdx.x * blockDim.x + threadIdx.x; v < N; v += gridDim.x * blockDim.x) {
        const int r = label[v];
        key[v] = r == INT_MAX ? INT_MAX : comp_of_root[r];
        val[v] = v;
    }
}

__global__ void __launch_bounds__(256)
cc_fac_keys_kernel(int kind, int F, const int* __restrict__ cam, const int* __restrict__ pt,
                   const int* __restrict__ rowptr, const int* __restrict__ vid,
                   const int* __restrict__ label, const int* __restrict__ comp_of_root,
                   int* key, int* val, int* comp_nfac) {
    for (int f = blockIdx.x * blockDim.x + threadIdx.x; f < F; f += gridDim.x * blockDim.x) {
        int r = INT_MAX;
        if (kind == KIND_BA_) {
            const int c = cam[f], q = pt[f];
#pragma unroll
            for (int k = 0; k < 12; ++k) {
                const int l = label[k < 9 ? c + k : q + (k - 9)];
                if (r == INT_MAX) r = l;
            }
        } else {
            for (int k = rowptr[f]; k < rowptr[f + 1] && r == INT_MAX; ++k) r = label[vid[k]];
        }
        const int comp = r == INT_MAX ? INT_MAX : comp_of_root[r];
        key[f] = comp;
        val[f] = f;
        if (comp != INT_MAX) atomicAdd(&comp_nfac[comp], 1);
    }
}

struct IsRoot {
    const int* label;
    __host__ __device__ int operator()(int v) const { return label[v] == v ? 1 : 0; }
};

struct Buf {   // a slice of the caller's workspace
    void* p = nullptr;
    template <class T> T* as() { return static_cast<T*>(p); }
};

#define CC_CHK(expr)                                   \
    do {                                               \
        const hipError_t e_ = (expr);                  \
        if (e_ != hipSuccess) return (int)e_;          \
    } while (0)

inline int grid_for(int64_t n) { return (int)std::min<int64_t>(std::max<int64_t>((n + 255) / 256, 1), 4096); }

}  // namespace

CcWorkspace::~CcWorkspace() { if (dev) (void)hipFree(dev); }

int device_components(hipStream_t stream, int kind, int N, int F, const int* cam, const int* pt, const int* rowptr,
                      const int* vid, const unsigned char* assigned_dev, CcWorkspace* ws, ComponentLists* out) {
    out->ncomp = out->nfree = out->nfac = 0;
    out->free_ptr.clear(); out->free_vid.clear(); out->fac_ptr.clear(); out->fac_id.clear();
    if (N <= 0) { out->free_ptr.assign(1, 0); out->fac_ptr.assign(1, 0); return 0; }
    const int M = std::max(N, std::max(F, 1));
    // every device buffer is a slice of one workspace kept by the caller between calls (a dozen
    // allocations and releases per labelling otherwise); the sizes of the library's scratch first
    Buf parent, count, label, key64, val, val2, val3, key64b, key32, key32b, comp_of_root, comp_nvars, comp_nfac, tmp, nsel, rtmp;
    size_t tb = 0, tb2 = 0, tb3 = 0, rb = 0;
    {
        unsigned long long* k64 = nullptr;
        int* i32 = nullptr;
        CC_CHK(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, k64, k64, i32, i32, N, 0, 64, stream));
        CC_CHK(hipcub::DeviceRadixSort::SortPairs(nullptr, tb2, i32, i32, i32, i32, N, 0, 32, stream));
        CC_CHK(hipcub::DeviceRadixSort::SortPairs(nullptr, tb3, i32, i32, i32, i32, std::max(F, 1), 0, 32, stream));
        hipcub::CountingInputIterator<int> idx(0);
        hipcub::TransformInputIterator<int, IsRoot, hipcub::CountingInputIterator<int>> it(idx, IsRoot{nullptr});
        CC_CHK(hipcub::DeviceReduce::Sum(nullptr, rb, it, i32, N, stream));
    }
    const size_t tbytes = std::max(tb, std::max(tb2, tb3));
    {
        const std::pair<Buf*, size_t> want[] = {
            {&parent, (size_t)N * 4}, {&count, (size_t)N * 4}, {&label, (size_t)N * 4}, {&key64, (size_t)N * 8}, {&key64b, (size_t)N * 8},
            {&val, (size_t)M * 4}, {&val2, (size_t)M * 4}, {&val3, (size_t)M * 4}, {&key32, (size_t)M * 4}, {&key32b, (size_t)M * 4},
            {&comp_of_root, (size_t)N * 4}, {&comp_nvars, (size_t)N * 4}, {&comp_nfac, (size_t)N * 4},
            {&tmp, tbytes}, {&nsel, 8}, {&rtmp, rb}};
        size_t total = 0;
        for (const auto& w : want) total += (std::max<size_t>(w.second, 8) + 255) / 256 * 256;
        if (ws->bytes < total) {
            CC_CHK(hipStreamSynchronize(stream));
            if (ws->dev) { CC_CHK(hipFree(ws->dev)); ws->dev = nullptr; ws->bytes = 0; }
            CC_CHK(hipMalloc(&ws->dev, total + total / 4));
            ws->bytes = total + total / 4;
        }
        char* base = static_cast<char*>(ws->dev);
        for (const auto& w : want) { w.first->p = base; base += (std::max<size_t>(w.second, 8) + 255) / 256 * 256; }
    }

    cc_init_kernel<<<grid_for(N), 256, 0, stream>>>(N, parent.as<int>(), count.as<int>());
    if (F > 0)
        cc_union_kernel<<<grid_for(F), 256, 0, stream>>>(kind, F, cam, pt, rowptr, vid, assigned_dev, parent.as<int>());
    cc_label_vars_kernel<<<grid_for(N), 256, 0, stream>>>(N, assigned_dev, parent.as<int>(), label.as<int>(), count.as<int>());
    cc_root_keys_kernel<<<grid_for(N), 256, 0, stream>>>(N, label.as<int>(), count.as<int>(),
                                                        key64.as<unsigned long long>(), val.as<int>());
    CC_CHK(hipGetLastError());
    // roots by (number of variables, id)
    size_t tbytes_ = tbytes;
    CC_CHK(hipcub::DeviceRadixSort::SortPairs(tmp.p, tbytes_, key64.as<unsigned long long>(), key64b.as<unsigned long long>(),
                                             val.as<int>(), val2.as<int>(), N, 0, 64, stream));
    // number of components = number of roots
    int ncomp = 0;
    {
        hipcub::CountingInputIterator<int> idx(0);
        hipcub::TransformInputIterator<int, IsRoot, hipcub::CountingInputIterator<int>> it(idx, IsRoot{label.as<int>()});
        CC_CHK(hipcub::DeviceReduce::Sum(rtmp.p, rb, it, nsel.as<int>(), N, stream));
        CC_CHK(hipMemcpyAsync(&ncomp, nsel.p, 4, hipMemcpyDeviceToHost, stream));
        CC_CHK(hipStreamSynchronize(stream));
    }
    out->ncomp = ncomp;
    out->free_ptr.assign((size_t)ncomp + 1, 0);
    out->fac_ptr.assign((size_t)ncomp + 1, 0);
    if (ncomp == 0) return 0;

    CC_CHK(hipMemsetAsync(comp_nfac.p, 0, (size_t)ncomp * 4, stream));
    cc_rank_kernel<<<grid_for(ncomp), 256, 0, stream>>>(ncomp, val2.as<int>(), count.as<int>(), comp_of_root.as<int>(), comp_nvars.as<int>());
    // variables grouped by component (stable: ascending id inside a component)
    cc_var_keys_kernel<<<grid_for(N), 256, 0, stream>>>(N, label.as<int>(), comp_of_root.as<int>(), key32.as<int>(), val.as<int>());
    CC_CHK(hipGetLastError());
    CC_CHK(hipcub::DeviceRadixSort::SortPairs(tmp.p, tbytes_, key32.as<int>(), key32b.as<int>(), val.as<int>(), val2.as<int>(), N, 0, 32, stream));
    // factors grouped by component (stable: ascending id), into a buffer of their own so that both
    // sorted lists can be fetched together
    if (F > 0) {
        cc_fac_keys_kernel<<<grid_for(F), 256, 0, stream>>>(kind, F, cam, pt, rowptr, vid, label.as<int>(), comp_of_root.as<int>(),
                                                           key32.as<int>(), val.as<int>(), comp_nfac.as<int>());
        CC_CHK(hipGetLastError());
        CC_CHK(hipcub::DeviceRadixSort::SortPairs(tmp.p, tbytes_, key32.as<int>(), key32b.as<int>(), val.as<int>(), val3.as<int>(), F, 0, 32, stream));
    }
    std::vector<int> h_nvars((size_t)ncomp), h_nfac((size_t)ncomp);
    CC_CHK(hipMemcpyAsync(h_nvars.data(), comp_nvars.p, (size_t)ncomp * 4, hipMemcpyDeviceToHost, stream));
    CC_CHK(hipMemcpyAsync(h_nfac.data(), comp_nfac.p, (size_t)ncomp * 4, hipMemcpyDeviceToHost, stream));
    CC_CHK(hipStreamSynchronize(stream));
    int64_t nfree = 0, nfac = 0;
    for (int i = 0; i < ncomp; ++i) {
        out->free_ptr[(size_t)i] = nfree; nfree += h_nvars[(size_t)i];
        out->fac_ptr[(size_t)i] = nfac; nfac += h_nfac[(size_t)i];
    }
    out->free_ptr[(size_t)ncomp] = nfree; out->nfree = nfree;
    out->fac_ptr[(size_t)ncomp] = nfac; out->nfac = nfac;
    std::vector<int> hv((size_t)nfree), hf((size_t)nfac);
    if (nfree > 0) CC_CHK(hipMemcpyAsync(hv.data(), val2.p, (size_t)nfree * 4, hipMemcpyDeviceToHost, stream));
    if (nfac > 0) CC_CHK(hipMemcpyAsync(hf.data(), val3.p, (size_t)nfac * 4, hipMemcpyDeviceToHost, stream));
    CC_CHK(hipStreamSynchronize(stream));
    out->free_vid.assign(hv.begin(), hv.end());
    out->fac_id.assign(hf.begin(), hf.end());
    return 0;
}

}  // namespace rdis_hip
