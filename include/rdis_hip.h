/*
 * rdis_hip.h -- C ABI of the MI355X (gfx950) subspace-solver path for RDIS.
 *
 * The reference has no FFI: its boundary for this path is C++ virtual dispatch,
 *     Numeric SubspaceOptimizer::optimize(const VariablePtrVec& vars,
 *         const FactorPtrVec& factors, NumericVec& xval, Numeric& deltaFval,
 *         const bool printdbg)                       (src/SubspaceOptimizer.h:37-39)
 * called from RDISOptimizer::getValueFromDomain (src/RDISOptimizer.cpp:1067) and
 * BCDOptimizer::optimize (src/optimizers/BCDOptimizer.cpp:149).  The drop-in
 * `rdis::HipCGDSubspaceOptimizer` (rdis_amd/host/) implements that virtual and
 * reaches the GPU only through the entry points below (plain pointers and sizes,
 * no C++ or torch types).  INTEGRATION.md shows the reference-side binding.
 *
 * Conventions
 *  - every function returns 0 or a negative RDIS_HIP_E* code; nothing throws
 *    across the ABI; rdis_hip_last_error() gives the text for the last failure
 *    on a context.
 *  - variable / factor ids are int64 like the reference's VariableID / FactorID
 *    (src/common.h:30-32): variable ids dense 0..N-1 in creation order, factor id
 *    = index in the function's factor list.  (Narrowed to int32 on the device;
 *    N, F and nnz must be < 2^31.)
 *  - the caller owns all host buffers; device memory belongs to the handles.
 *  - one context per GPU; calls on a context are serialised by the caller (the
 *    reference path is single-threaded and not re-entrant either).
 *  - fp64 throughout ("Numeric = double", src/common.h:25).
 *  - several contexts may be driven from one thread (one per GPU: rdis::OptimizableFunction::setDevices): every entry point
 *    makes its context's device current first.  Environment RDIS_HIP_VIRTUAL_DEVICES=n (a test aid for boxes with one GPU):
 *    rdis_hip_device_count() reports n devices that all stand on GPU 0, and every HIP call the library issues for a context
 *    while another context's device is the current one fails with RDIS_HIP_EDEVICE -- what a forgotten "make current" would
 *    do on a real node.
 */
#ifndef RDIS_HIP_H_
#define RDIS_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RDIS_HIP_ABI_VERSION 1

enum {
    RDIS_HIP_OK = 0,
    RDIS_HIP_EINVAL = -1,   /* bad argument (NULL, negative size, id out of range) */
    RDIS_HIP_ENOMEM = -2,   /* host or device allocation failed */
    RDIS_HIP_EDEVICE = -3,  /* HIP runtime error (no device, launch failure, ...) */
    RDIS_HIP_EOVERLAP = -4, /* components of a plan are not independent */
    RDIS_HIP_ERANGE = -5    /* size exceeds the int32 device index range */
};

/* exit reason of one component solve = status & 0xff; these replace the
 * exceptions / asserts the reference uses inside optimize()
 * (src/optimizers/CGDSubspaceOptimizer.cpp:42-58, :175;
 *  external/include/minimize_nrc.h:649,663,675,690,403) */
enum {
    RDIS_HIP_EXIT_FTOL = 0,         /* 2|df| <= ftol(|f|+|fp|+1e-18) */
    RDIS_HIP_EXIT_GTOL = 1,         /* scaled gradient below 1e-8 */
    RDIS_HIP_EXIT_GGZERO = 2,       /* gradient exactly zero */
    RDIS_HIP_EXIT_ITMAX = 3,        /* "Too many iterations in frprmn" (normal for BA) */
    RDIS_HIP_EXIT_DBRENT_ITMAX = 4, /* "Too many iterations in routine dbrent" */
    RDIS_HIP_EXIT_NAN = 5,          /* a NaN objective (the reference asserts) */
    RDIS_HIP_EXIT_EMPTY = 6,        /* empty factor list: returns 0, touches nothing */
    RDIS_HIP_EXIT_SYNC_TIMEOUT = 7  /* device-side barrier gave up (never expected) */
};
#define RDIS_HIP_STATUS_ROLLED_BACK 0x100 /* negative progress: initial x restored
                                             (CGDSubspaceOptimizer.cpp:66-80) */

typedef struct rdis_hip_ctx rdis_hip_ctx;
typedef struct rdis_hip_problem rdis_hip_problem;
typedef struct rdis_hip_plan rdis_hip_plan;

/* ---- context ------------------------------------------------------------------ */
int rdis_hip_abi_version(void);
int rdis_hip_device_count(void);
int rdis_hip_create(int device, rdis_hip_ctx **out);
void rdis_hip_destroy(rdis_hip_ctx *ctx);
const char *rdis_hip_last_error(const rdis_hip_ctx *ctx);
/* run on a caller-provided hipStream_t (NULL: the context's own stream) */
int rdis_hip_set_stream(rdis_hip_ctx *ctx, void *hip_stream);
int rdis_hip_synchronize(rdis_hip_ctx *ctx);
/* copy bytes from a device pointer handed out by this library to host memory
 * (for FFI callers without HIP bindings); ordered after prior work on the stream */
int rdis_hip_copy_to_host(rdis_hip_ctx *ctx, void *dst, const void *dev_src, int64_t bytes);

/* ---- an OptimizableFunction in packed form --------------------------------------
 * x0/lo/hi: assigned value and single-interval domain of every variable
 * (VariableDomain::interval(), src/VariableDomain.h:53; CGD asserts one
 * sub-interval, CGDSubspaceOptimizer.cpp:119).
 *
 * upload_ba replaces F BundleAdjustmentFactor objects
 * (src/bundleadjust/BundleAdjustmentFactor.h:18-31): factor i reads variables
 * cam_vid0[i]..+8 = [rx ry rz tx ty tz f k1 k2] and pt_vid0[i]..+2 = [X Y Z]
 * (BundleAdjustmentFunction.h:88-96) and the observation obs[2i], obs[2i+1]. */
int rdis_hip_upload_ba(rdis_hip_ctx *ctx, int64_t nvars, const double *x0, const double *lo,
                       const double *hi, int64_t nfac, const int64_t *cam_vid0,
                       const int64_t *pt_vid0, const double *obs, rdis_hip_problem **out);
/* upload_nlp replaces F NonlinearProductFactor objects
 * (src/NonlinearProductFactor.h:21-113): factor i = coeff[i] * prod over
 * k in [rowptr[i], rowptr[i+1]) of g((x[vid[k]] - cons[k])^expo[k]), g = sin iff sine[k].
 * Arithmetic: exponents 0, 1, 2 as the reference special-cases them (src/util/numeric.cpp:12-23), 3 and 4 by multiplication,
 * others through pow; sine and cosine by the library's own routine of the rotation angle (below 1 ulp) -- last-place differences
 * from std::pow / std::sin / std::cos, chosen so that a host can compute the same bits. */
int rdis_hip_upload_nlp(rdis_hip_ctx *ctx, int64_t nvars, const double *x0, const double *lo,
                        const double *hi, int64_t nfac, const double *coeff,
                        const int64_t *rowptr, const int64_t *vid, const double *expo,
                        const double *cons, const uint8_t *sine, rdis_hip_problem **out);
/* NonlinearProductFactor's useExponential (src/NonlinearProductFactor.h:62, 113): factor i with use_exp[i] != 0
 * evaluates to coeff[i] * exp(-product) (src/NonlinearProductFactor.cpp:140, 204).  Values only: the reference's
 * computeGradient asserts the flag off (.cpp:110), so eval_grad, plan_create and cgd_batch over a list that holds
 * such a factor return RDIS_HIP_EINVAL.  use_exp == NULL or all zero: cleared (the state after upload_nlp). */
int rdis_hip_nlp_set_exponential(rdis_hip_problem *p, const uint8_t *use_exp);
void rdis_hip_free_problem(rdis_hip_problem *p);

/* Variable::assign for already-assigned variables (src/Variable.cpp:66-88): the
 * constants an outer optimiser fixes before a solve.  vid == NULL: vid = 0..n-1. */
int rdis_hip_set_x(rdis_hip_problem *p, int64_t n, const int64_t *vid, const double *val);
int rdis_hip_get_x(rdis_hip_problem *p, int64_t n, const int64_t *vid, double *out);

/* ---- batched factor evaluation at the currently assigned x ----------------------
 * fac == NULL: all factors 0..nf-1 (nf must then equal the factor count).
 * eval      : OptimizableFunction::evalFactors (src/OptimizableFunction.cpp:95-135)
 * eval_grad : ... + computeGradient(facs, pg) (src/OptimizableFunction.cpp:234-262, the merge of src/State.h:157-210);
 *             g is dense over all N variables (0 where no listed factor touches it).  Bundle adjustment: one
 *             pass over the factors, no per-factor partial leaves the compute unit (rdis_amd/csrc/grad_fused.hpp);
 *             a point variable's contributions are added in factor-list order like the reference's merge, a camera
 *             variable's in factor-list order within a tile of the list and then tile by tile -- a fixed order, the
 *             same bits run to run; the value is bit for bit what eval returns for the same list.  The tables of
 *             a list are built at its first use and kept (up to 64 lists within 1 GiB of device memory, least recently used
 *             first to go; all factors: fac == NULL; an explicit list of at most 512 entries needs no tables and builds none).
 *             Nonlinear-product functions: per-factor partials, then each g[v] in factor-list order.
 * eval_grad_device : the same, results left on the device: *f_dev points to one double, *g_dev to N doubles,
 *             valid until the next evaluation call on this problem; asynchronous on the context's stream
 *             (rdis_hip_synchronize / rdis_hip_copy_to_host order after it).
 * eval_each / grad_each_ba: per-factor values / 12 partials, for kernel parity tests
 *             (Factor::eval, BundleAdjustmentFactor::computeGradient). */
int rdis_hip_eval(rdis_hip_problem *p, int64_t nf, const int64_t *fac, double *f);
int rdis_hip_eval_grad(rdis_hip_problem *p, int64_t nf, const int64_t *fac, double *f, double *g);
int rdis_hip_eval_grad_device(rdis_hip_problem *p, int64_t nf, const int64_t *fac, void **f_dev, void **g_dev);
int rdis_hip_eval_each(rdis_hip_problem *p, int64_t nf, const int64_t *fac, double *fvals);
/* how eval_each / grad_each_ba round a bundle-adjustment factor's arithmetic: 0 (default) = one fused multiply-add where the
 * model has a * b + c, 1 = every product rounded before it is added, like the reference's x86-64 build -- the arithmetic of the
 * solvers' parity option (plan option "factor_rounding" = 1), so that tests can compare it factor by factor, bit for bit,
 * with a CPU restatement (BundleAdjustmentFactor.cpp:160-185, 266-335) */
int rdis_hip_set_factor_rounding(rdis_hip_problem *p, int32_t mode);
int rdis_hip_grad_each_ba(rdis_hip_problem *p, int64_t nf, const int64_t *fac, double *g12);

/* ---- a second subspace solver: Levenberg-Marquardt (bundle adjustment) ---------------
 * The least-squares problem LMSubspaceOptimizer::optimize hands to levmar
 * (src/optimizers/LMSubspaceOptimizer.cpp:28-147, 176-278: residual sqrt(2 E_j) per factor,
 * Jacobian row grad E_j / e_j, damping scale 1e-3, eps1 = eps2 = 1e-15, eps3 = ftol, itmax =
 * maxiters, result clamped into the domains), solved on the device: normal equations in
 * camera / point blocks, Schur complement onto the cameras, both contractions on the matrix
 * cores (rdis_amd/csrc/lm_solver.hip).  levmar is not vendored by the reference: parity is
 * unpinned, the iteration is checked step by step against oracle/lm_oracle.py.
 * residual_model 1 is that formulation; 2 replaces it by the two pixel residuals of every
 * observation (the usual bundle-adjustment Gauss-Newton model: J^T J of rank 2 per factor instead
 * of 1, same machinery) -- not something the reference offers, far faster convergence.
 * Bits 4-5 of residual_model choose how the Schur product Z Z^T is formed: 0 = by fill (block-sparse over the
 * camera pairs that share a point where cameras see few of the points -- every BAL problem --, dense on the
 * matrix cores otherwise), 1 = dense, 2 = block-sparse.
 * Same calling convention as one component of rdis_hip_cgd_batch; x_inout may be NULL (start at
 * the currently assigned x; the result is left assigned either way).
 * info[8] = {iterations, stop code (levmar's: 1 small gradient, 2 small step, 3 itmax,
 * 4 singular, 5 no further reduction, 6 small error, 7 invalid values), residual evaluations,
 * Jacobian evaluations, linear solves, final damping mu, camera blocks, point blocks};
 * hist (may be NULL): up to hist_cap records {mu, |Dp|^2, f(trial), accepted} per linear solve. */
int rdis_hip_lm_optimize(rdis_hip_problem *p, int64_t nfree, const int64_t *free_vid, int64_t nf,
                         const int64_t *fac_id, double *x_inout, int32_t maxiters, double ftol,
                         int32_t residual_model, double *fret, double *delta, double *info8, double *hist4, int64_t hist_cap,
                         int64_t *nhist);

/* ---- the step before the path: which independent sub-problems are there? -----
 * Connected components of the factor graph once the variables with assigned[v] != 0 are fixed:
 * what Component::createChildren (src/Component.cpp:508-549) obtains from the reference's dynamic
 * connectivity structure (ConnectivityGraph.h:255-261), computed on the device by a lock-free
 * union-find (rdis_amd/csrc/components.hip).  A component = unassigned variables connected
 * through factors.  Lists come out the way Component::init leaves them (Component.cpp:60-79):
 * variable ids ascending, factor ids ascending; components ordered by number of variables
 * ascending (ComponentComparator, Component.cpp:603-608), equal sizes by smallest variable id.
 * Factors all of whose variables are assigned are in no component; an unassigned variable that
 * no factor reads is a component with an empty factor list.  The four arrays are exactly the
 * arguments of rdis_hip_plan_create / rdis_hip_cgd_batch.
 *   rdis_hip_components        computes; returns the sizes
 *   rdis_hip_components_fetch  copies the lists of the last call (free_ptr, fac_ptr: ncomp + 1) */
int rdis_hip_components(rdis_hip_problem *p, const uint8_t *assigned, int64_t *ncomp, int64_t *nfree,
                        int64_t *nfac);
int rdis_hip_components_fetch(rdis_hip_problem *p, int64_t *free_ptr, int64_t *free_vid, int64_t *fac_ptr,
                              int64_t *fac_id);

/* ---- the solver: CGDSubspaceOptimizer::optimize for a batch of independent
 * components --------------------------------------------------------------------
 * Component c optimises the free variables free_vid[free_ptr[c] .. free_ptr[c+1])
 * over the factors fac_id[fac_ptr[c] .. fac_ptr[c+1]) (each list in the order the
 * reference would pass vars / factors).  Components must be independent: no
 * shared free variable and no factor of one reading a free variable of another
 * (they are connected components of the residual factor graph,
 * src/Component.cpp:508-549); otherwise RDIS_HIP_EOVERLAP.
 *
 * x_inout (free-variable order, concatenated over components): start values in,
 * final clamped values out; the problem's variables are left assigned to them
 * (post-condition of optimize(), CGDSubspaceOptimizer.cpp:84-86).  maxiters / ftol
 * are SubspaceOptimizer's SSmaxit / SSftol (src/SubspaceOptimizer.cpp:15-32).
 * Per component: fret (returned value), delta (deltaFval), iters (Frprmn::iter),
 * status (RDIS_HIP_EXIT_* | RDIS_HIP_STATUS_ROLLED_BACK), nfeval / ngeval (calls
 * of SubfunctionFD::operator() / ::df the reference would have made).  Any output
 * pointer may be NULL. */
int rdis_hip_cgd_batch(rdis_hip_problem *p, int64_t ncomp, const int64_t *free_ptr,
                       const int64_t *free_vid, const int64_t *fac_ptr, const int64_t *fac_id,
                       double *x_inout, int32_t maxiters, double ftol, double *fret,
                       double *delta, int32_t *iters, int32_t *status, int64_t *nfeval,
                       int64_t *ngeval);

/* The same in three steps, so that a decomposition is analysed and uploaded once
 * and solved many times with everything resident in HBM:
 *   plan_create : validate + upload the decomposition, build the per-variable
 *                 gather lists, allocate workspace
 *   plan_set_start : (optional) new start values, free-variable order; NULL = take
 *                 the problem's currently assigned x
 *   plan_solve  : reset the free variables to the start and run the solver
 *                 (asynchronous on the context's stream)
 *   plan_fetch  : wait and copy results out (any pointer may be NULL) */
int rdis_hip_plan_create(rdis_hip_problem *p, int64_t ncomp, const int64_t *free_ptr,
                         const int64_t *free_vid, const int64_t *fac_ptr, const int64_t *fac_id,
                         rdis_hip_plan **out);
void rdis_hip_plan_destroy(rdis_hip_plan *plan);
int rdis_hip_plan_set_start(rdis_hip_plan *plan, const double *x_start);
int rdis_hip_plan_solve(rdis_hip_plan *plan, int32_t maxiters, double ftol);
int rdis_hip_plan_fetch(rdis_hip_plan *plan, double *x_out, double *fret, double *delta,
                        int32_t *iters, int32_t *status, int64_t *nfeval, int64_t *ngeval);
/* sum of fret over the plan's components, left on the device (for the RCCL
 * all-reduce of the top-level objective, src/RDISOptimizer.cpp:1491-1494);
 * returns a device pointer to one double valid until the next plan_solve. */
int rdis_hip_plan_objective_device(rdis_hip_plan *plan, void **dev_ptr);

/* ---- the path's one collective --------------------------------------------------------
 * Independent components of a recursion level are sharded over the GPUs of a node (rdis_amd/dist.py, DESIGN.md section 5:
 * no data-path collective); what the ranks exchange is the top-level objective -- the sum the reference forms on one host
 * (src/RDISOptimizer.cpp:1491-1494) -- one fp64 all-reduce over RCCL / xGMI, on the stream the solve ran on.
 *   comm_unique_id   rank 0 makes the 128-byte id the ranks of a communicator share (ncclGetUniqueId); the caller
 *                    distributes it (a file, a socket, MPI: its business)
 *   comm_create      one rank per process and GPU (ncclCommInitRank on the context's device); collective over the ranks
 *   comm_create_all  one process that drives several contexts (rdis::OptimizableFunction::setDevices): one communicator
 *                    per context, each on a different GPU (ncclCommInitAll)
 *   allreduce_objective      sum over the ranks of rdis_hip_plan_objective_device's double, in place, asynchronous on the
 *                    context's stream; sum_out != NULL: copied out (waits).  comm == NULL: a world of one.
 *   allreduce_objective_all  the same for the plans of one process's contexts (grouped: one thread drives every rank);
 *                    comms == NULL: the partial sums meet on the host in plan order (no RCCL, or one GPU listed twice)
 *   comm_allreduce_f64       a few doubles of host memory summed (op 0) or maximised (op 1) over the ranks: counters, and the
 *                    barrier + maximum of a timed region
 * RCCL is loaded when the first communicator is made (dlopen): a single-GPU user never maps it. */
#define RDIS_HIP_COMM_ID_BYTES 128
typedef struct rdis_hip_comm rdis_hip_comm;
int rdis_hip_comm_unique_id(void *id128);
int rdis_hip_comm_create(rdis_hip_ctx *ctx, int32_t world, int32_t rank, const void *id128, rdis_hip_comm **out);
int rdis_hip_comm_create_all(int32_t n, rdis_hip_ctx *const *ctxs, rdis_hip_comm **comms);
void rdis_hip_comm_destroy(rdis_hip_comm *comm);
int rdis_hip_allreduce_objective(rdis_hip_plan *plan, rdis_hip_comm *comm, double *sum_out);
int rdis_hip_allreduce_objective_all(int32_t n, rdis_hip_plan *const *plans, rdis_hip_comm *const *comms, double *sum_out);
int rdis_hip_comm_allreduce_f64(rdis_hip_comm *comm, double *inout, int32_t n, int32_t op);

/* tuning / introspection ---------------------------------------------------------- */
/* option names: "block_threads" (workgroup size of the per-component solver: 64, 128, 256, 512,
 * 768 or 1024; 0 = auto), "coop_min_factors" (bundle-adjustment components with at least this
 * many factors are solved by the multi-workgroup cooperative kernel, one launch each; 0 = never;
 * default 4096), "coop_max_components" (only when the plan has at most this many such
 * components, default 48 -- their groups are packed into launches of what is resident at once; a
 * plan with more of them is one batched launch, one workgroup per component), "coop_group_min_factors" (default 256: when the cooperative groups of ALL
 * components with at least this many factors fit the device together, each of them gets one and
 * they run side by side in one launch; 0 = off), "coop_workgroups" (cap, 0 = what fits),
 * "coop_threads" (128, 256 or 512), "coop_poll_delay" (x64 cycles between publishing and the
 * first granule sweep), "coop_pipeline" (default 1: cooperative groups run with the control logic,
 * the exchange and the factor arithmetic on waves of their own, solver_pipe.hpp, whenever every
 * group of the plan fits that layout of 128 factor lanes per workgroup; 0 = the plain cooperative
 * kernel; same bits either way as long as every variable fed by more than 48 partials has a wave;
 * the environment variable RDIS_HIP_COOP_PIPELINE=0 / 1 sets the default for plans the caller does
 * not see, e.g. the transient one of rdis_hip_cgd_batch),
 * "coop_speculate" (default 1: the pipelined groups evaluate guesses at the following trial steps
 * of a line search ahead of the control logic; results do not depend on it), "force_stream" (send large components to the streaming grid solver even
 * when they fit the register-resident one; large components that do not fit, and large
 * nonlinear-product components, always go there),
 * "quad_max_vars" / "quad_min_components" / "row_min_components" (bundle-adjustment components
 * with at most quad_max_vars free variables, default and maximum 4, are solved by groups of four
 * lanes, sixteen per wave, when the plan has at least quad_min_components of them, default 16384;
 * by groups of sixteen lanes from row_min_components, default 4096; below that by a workgroup
 * each; 0 variables = never),
 * "camera_records" (bundle adjustment, batched launches: 1 = default -- when no camera variable is
 * free in the launch its factors read per-camera rotation records (angle, axis, sine, cosine,
 * computed once per camera) and form only the point partials; with free cameras and more than 2048
 * factors in a component the component rewrites the records of its cameras at every trial point;
 * 2 = records wherever possible; 0 = every factor forms its camera's rotation itself.  Results
 * are bit-identical in all three settings),
 * "tiny_max_blocks" (cap on the grid of the persistent tiny-component kernels, 0 = what is
 * resident; for tests),
 * "overlap_batch" (default 1: the batched launch of a plan runs on a second stream, concurrently
 * with its cooperative launches when those take at most half of the compute units),
 * "lds_resident" (bundle adjustment, default 1: a component of the batched launch whose variables --
 * free ones and the constants its factors read -- fit a compute unit's LDS keeps them there as slots,
 * solver_lds.hpp; 0 = never: such components run on the plain batch solver, bit-identical where the
 * slots are the free variables), "lds_threads" / "lds_rot" / "lds_camera_sums" (its workgroup size,
 * 0 = auto; rotation records in it, -1 = auto; 0 = camera partials through memory like the plain
 * solver, for bit-for-bit comparisons),
 * "lds_matrix" (default 0; 1 = the LDS-resident solver's line-search trials in matrix form, a camera's rotation matrix and its
 * derivative along the direction formed once per camera and trial point -- what the point-major streaming solver does; measured
 * slower here, where a component has a few hundred factors per camera: the records are one lane's chain in front of every trial),
 * "ptm_stream" (default 1: components too large for the LDS whose CAMERA blocks fit it stream their
 * point blocks from HBM once per trial point, solver_ptm.hpp; 0 = never, 2 = every component whose
 * tables fit), "ptm_threads" (its workgroup size: 0 = auto, 256, 512 or 768), "ptm_group" (workgroups
 * that share one such component when the launch has fewer components than compute units: 0 = auto,
 * 1 = never, k <= 16 = k; for the WIDE groups of at most eight large components -- a component too large for a
 * cooperative group whose cameras fit the LDS streams through this solver on as many workgroups of 512 lanes as are resident,
 * instead of the grid solver -- up to 512),
 * "ptm_round_slots" (the slots of factors a round of that solver's gradient pass evaluates and stages: 0 = two where the workgroup
 * has at most 512 lanes and the LDS holds their staging rows, else one; 1; 2 -- the same bits on chunks of even slot counts),
 * "ptm_local_cameras" (a wide group whose component has more cameras than a compute unit's LDS holds, about 125: every
 * workgroup keeps only the cameras its own contiguous share of the camera-sorted chunk order meets, under local numbers;
 * -1 = default: where the cameras do not fit; 0 = never: such a component takes the grid solver; 1 = every wide group, for
 * tests.  One such component a plan; falls back to the grid solver when a workgroup's cameras would not fit either),
 * "factor_rounding" (how the factor arithmetic rounds a * b + c: 0 = one fused multiply-add; 1 = the PARITY option: the product
 * is rounded before it is added, like the reference's x86-64 build (g++ emits no fused multiply-add), and EVERY sum of a solve is
 * added in the reference's order by one lane (or one wave feeding one lane): the objective over the listed factors
 * (OptimizableFunction.cpp:95-135), every variable's partials in factor-list order (State.h:157-210), a trial's slope as gradient
 * times direction over the variables in list order (Df1dim::df, minimize_nrc.h:439-447), gg and dgg of the Polak-Ribiere step
 * (minimize_nrc.h:665-672).  With it -- and "emulate_stale_cache" -- a solve returns, bit for bit (==: fret, x, iterations, call
 * counts), what the CPU oracle returns when its three named switches for the device's factor arithmetic are on (its own sine /
 * cosine of the rotation angle, two reciprocals in place of five quotients, the adjoint sweep in place of the reference's forward
 * chain: oracle/rdis_oracle.h RO_ARITH_*, RO_BA_DERIV_ADJOINT_DEVICE), on BASELINE configs 3 and 4, from x0, 25 iterations, no
 * re-synchronisation (tests/test_gpu_parity.py); and the end values over one-ulp starts pass the plain two-sample test against the
 * reference-faithful oracle.  A parity option, not a fast path: the LDS-resident batch solver pays a full gradient and four
 * sequential sums per trial, a cooperative group runs in the plain layout with two sequential sums per trial (full ladybug: 0.4 s a
 * solve instead of 2.5 ms); refused where other solvers would run; -1 = default: the cooperative solvers round like the reference
 * (4 % slower) and keep their parallel sums, the batch solvers fuse.  After 25 unconverged CG iterations the DISTRIBUTION of end
 * values over one-ulp starts depends on all of that: DESIGN.md section 6),
 * "emulate_stale_cache" (default 0; 1 = the reference's factor cache, Variable.cpp:66-76 and
 * Factor.h:228-234 -- a factor keeps its value while its variables have moved by less than 1e-12 since
 * it was computed -- emulated in the LDS-resident batch solver and, with factor_rounding = 1, in the cooperative solver's plain
 * layout; refused where other solvers would run),
 * "trace_records" (per-component trace capacity, 0 = off), "dump_iters" (record p and
 * the search direction at the start of the first k line minimisations, 0 = off). */
int rdis_hip_plan_set_option(rdis_hip_plan *plan, const char *name, int64_t value);
/* which solver the plan's components go to (a test and tuning aid; the partition is computed on demand):
 * "components_cooperative", "components_grid_stream", "components_tiny", "components_lds",
 * "components_point_major", "components_plain" (counts), "pipelined" (0/1: cooperative groups use the
 * pipelined layout), "point_major_group" (workgroups per component in the last solve's point-major launch), "point_major_threads" (their lanes), "point_major_round_slots" (slots a gradient round staged), "grid_stream_workgroups" (workgroups of the first component on the grid solver),
 * "point_major_wide" (0/1: that launch was a wide group), "point_major_local_cameras" (0, or the most cameras a workgroup of
 * a wide group with local camera numbering holds) */
int rdis_hip_plan_get_info(rdis_hip_plan *plan, const char *name, int64_t *value);
/* device memory the plan holds beyond the problem's (index tables, workspace, per-factor
 * partials, results): what a host-side cache of plans budgets with
 * (rdis::HipCGDSubspaceOptimizer::setPlanCacheBytes).  The solvers' tables are built on demand: the call builds
 * them if they do not exist yet (it may return RDIS_HIP_ENOMEM); what the first solve adds for the launch shape
 * it picks shows in a query after that solve. */
int rdis_hip_plan_device_bytes(rdis_hip_plan *plan, int64_t *bytes);
/* device time of the solver kernel(s) of the last plan_solve, measured with HIP
 * events on the launch stream; launches = number of kernel launches it covers */
int rdis_hip_plan_last_kernel_ms(rdis_hip_plan *plan, double *ms, int32_t *launches);
/* solver trace of component c of the last solve (trace_records > 0): up to cap
 * records of 4 doubles {tag, a, b, c}; *nrec = records written by the device */
int rdis_hip_plan_get_trace(rdis_hip_plan *plan, int64_t comp, double *rec4, int64_t cap,
                            int64_t *nrec);

/* 32 shader-cycle accumulators of the last cooperative solve, as seen by lane 0 of
 * workgroup 0: {factor arithmetic, workgroup reduce, publish, granule sweep, tail,
 * #exchanges, #sweeps, whole kernel, control step, request hand-over, combine, release,
 * [12..20] cycles per request kind, [22..30] requests per kind}.  All zero unless the
 * library was built with -DRDIS_COOP_TIMING (profiling aid) */
int rdis_hip_plan_debug_counters(rdis_hip_plan *plan, int64_t *out32);
/* p and search direction at the start of each of the first dump_iters line
 * minimisations of component c: out[dump_iters][2][nfree_c] (dump_iters > 0) */
int rdis_hip_plan_get_vectors(rdis_hip_plan *plan, int64_t comp, double *out, int64_t cap_doubles);

#ifdef __cplusplus
}
#endif
#endif /* RDIS_HIP_H_ */
