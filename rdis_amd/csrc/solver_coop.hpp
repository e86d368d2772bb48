// solver_coop.hpp -- cooperative multi-workgroup solver for one large component
// (placeholder until the grid-wide variant lands; the single-workgroup solver
// handles every component meanwhile).
#pragma once
#include "solver_wg.hpp"

namespace rdis_hip {
constexpr int COOP_MAX_WG = 256;
inline size_t coop_state_bytes() { return 4096; }
inline int launch_coop(hipStream_t, int, int, const ProblemView&, const PlanView&, int, int, int, void*, int, double) {
    return (int)hipErrorNotSupported;
}
}  // namespace rdis_hip
