// lm_solver.hpp -- interface of lm_solver.hip: Levenberg-Marquardt for a bundle-adjustment
// sub-problem, normal equations reduced to the cameras by a Schur complement, on the device.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <string>
#include <vector>

namespace rdis_hip {

struct LmProblem {          // device arrays of an uploaded bundle-adjustment problem
    int N, F, ncams;        // ncams: variables 0 .. 9*ncams-1 are cameras, the rest points (3 each)
    double* x;
    const double *lo, *hi;
    const int *cam, *pt;    // first variable id of each factor's camera / point block (device)
    const int *h_cam, *h_pt; // the same on the host
    const double2* obs;
};

struct LmOptions {
    int maxiters;           // SSmaxit
    double tau, eps1, eps2, eps3;   // levmar opts[0..3] (reference: 1e-3, 1e-15, 1e-15, SSftol)
    int model;              // residual rows per factor: 1 = sqrt(2 E_j) (the reference's), 2 = the two pixel residuals
    int schur;              // the Schur product Z Z^T: 0 = by fill, 1 = dense on the matrix cores, 2 = block-sparse (camera pairs that share a point)
};

struct LmStep { double mu, dp_l2, f_trial; int accepted; };

struct LmResult {
    double fret = 0, finit = 0, mu = 0;
    int iters = 0, stop = 0, nfev = 0, njev = 0, nsolve = 0;
    int ncam_blocks = 0, npt_blocks = 0;
    int sparse_schur = 0;           // which Schur product ran
    std::vector<LmStep> history;    // one entry per linear solve
};

// device scratch and pinned host scalars of the solver, kept by the caller between solves (grown on demand)
struct LmWorkspace {
    void* dev = nullptr;
    size_t dev_bytes = 0;
    double* pinned = nullptr;
    LmWorkspace() = default;
    LmWorkspace(const LmWorkspace&) = delete;
    LmWorkspace& operator=(const LmWorkspace&) = delete;
    ~LmWorkspace();
};

// free_vid / fac: host arrays (sorted ascending not required).  x is updated in place: the free
// variables end at the clamped result.  Returns 0, a hipError_t (> 0), or -1 with *err set.
int device_lm_ba(hipStream_t stream, const LmProblem& P, int64_t nfree, const int64_t* free_vid, int64_t nf,
                 const int64_t* fac, const LmOptions& opt, LmWorkspace* ws, LmResult* out, std::string* err);

}  // namespace rdis_hip
