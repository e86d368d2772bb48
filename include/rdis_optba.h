/*
 * rdis_optba.h -- C entry of the caller side of the path (rdis_amd/lib/librdis_host.so): what the reference's optBA
 * executable does between loading a BAL file and printing the optimum (src/bundleadjust/optBA.cpp:117-330:
 * BundleAdjustmentFunction::load, a SubspaceOptimizer with SSmaxit / SSftol, RDISOptimizer::optimize from the file's
 * values, "--randinit 0"), with rdis::HipRDISLevelOptimizer in RDISOptimizer's place and rdis::HipCGDSubspaceOptimizer
 * in CGDSubspaceOptimizer's (rdis_amd/host/rdis_levels.h, rdis_host.h).  This is the "(all components)" form of
 * BASELINE.json's metric: every subspace-optimizer call the recursion makes on the problem's real component mix --
 * separator blocks, leaves, single points -- instead of one component ("trace replay"; the reference's own
 * run is 731 calls on ladybug 5 / 30 and 13 850 on full ladybug).
 *
 * What is NOT the reference's: the cut (a degree-ordered separator in place of the binary-only PaToH call,
 * src/RDISOptimizer.cpp:779-865) and the random restarts' values (drawn per (node, restart, variable) instead of from
 * one shared generator in visiting order, :1196-1216) -- end-to-end parity with optBA is unpinned; the schedule itself
 * is checked decision by decision by oracle/levels.py (tests/test_host_dropin.py).
 *
 * schedule  0: sweeps over the levels (HipRDISLevelOptimizer::optimize), 1: the reference's per-node schedule with
 *              iterative improvement and random restarts (optimizeReferenceSchedule, src/RDISOptimizer.cpp:253-334,
 *              971-1147, 1507-1577).
 * options   nopts (name, value) pairs: SSmaxit, SSftol (src/SubspaceOptimizer.cpp:26-32), AVblkpct, steptol,
 *              maxSweeps, sepPiecePct, nRRperLvl, nRRatTop, minRR, maxNAtoRR, noAssignLimitAtTop, restartSeed,
 *              maxCalls (rdis_levels.h); anything else is an error.
 * out       RDIS_OPTBA_NOUT doubles:
 *              [0] final objective            [1] objective at the file's values
 *              [2] subspace-optimizer calls   [3] their CG iterations (Frprmn outer iterations), summed
 *              [4] launches: optimizeBatch calls (schedule 1) / plan launches (schedule 0)
 *              [5] seconds in the optimisation (decomposition and device upload outside)
 *              [6] seconds for the decomposition (separators + device labelling)
 *              [7] nodes of the decomposition tree   [8] objective evaluations of the calls, summed (schedule 1)
 *              [9] sweeps (schedule 0) / trace records (schedule 1)
 * x_out     may be NULL; else the final values of all variables (9 ncams + 3 npts doubles).
 * Returns 0, -1 (file cannot be loaded), -2 (an exception: message on stderr), -3 (bad arguments).
 */
#ifndef RDIS_OPTBA_H_
#define RDIS_OPTBA_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RDIS_OPTBA_NOUT 10
/* rdis_optba_run_hist: the same run, and (schedule 1) where its subspace-optimizer calls go: hist_rows rows, one per depth of the
 * decomposition tree, of RDIS_OPTBA_HIST_COLS doubles -- [0] nodes of that depth, [1] their free variables (a separator's /
 * a leaf's) in all, the steps of the reference's schedule at that depth by kind (src/RDISOptimizer.cpp:1131-1133): [2] "initial
 * values", [3] "iterative improvement", [4] "random restart", [5] steps that made no progress beyond steptol (:1086-1101),
 * [6] evaluations that were a new minimum (updateDomain, :1507-1577). */
#define RDIS_OPTBA_HIST_COLS 7

int rdis_optba_run(const char *bal_file, int64_t ncams, int64_t npts, int32_t schedule, int32_t nopts,
                   const char *const *opt_names, const double *opt_vals, int32_t device, double *out, double *x_out);
int rdis_optba_run_hist(const char *bal_file, int64_t ncams, int64_t npts, int32_t schedule, int32_t nopts,
                        const char *const *opt_names, const double *opt_vals, int32_t device, double *out, double *x_out,
                        double *hist, int32_t hist_rows);

#ifdef __cplusplus
}
#endif
#endif /* RDIS_OPTBA_H_ */
