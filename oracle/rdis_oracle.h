/*
 * rdis_oracle.h -- CPU ORACLE for the RDIS subspace-solver hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, the smoke check
 * in __graft_entry__.py and the cpu_baseline leg of bench.py may link or call
 * it.  The product path (rdis_amd/csrc, include/rdis_hip.h) never does.
 *
 * It is a plain-C restatement of the reference's algorithm for the path
 * (citations are path:line under /root/reference):
 *   - factor evaluation         src/bundleadjust/BundleAdjustmentFactor.cpp:160-185,266-335
 *                               src/NonlinearProductFactor.cpp:186-209, 149-178
 *   - sum over a factor list    src/OptimizableFunction.cpp:95-135
 *   - gradient of the sum       src/OptimizableFunction.cpp:234-262, src/State.h:157-210
 *   - clamp-assign              src/optimizers/CGDSubspaceOptimizer.cpp:160-184,
 *                               src/Variable.cpp:66-88, src/VariableDomain.cpp:158-163
 *   - Polak-Ribiere CG, line minimisation, Brent-with-derivatives, bracketing
 *                               external/include/minimize_nrc.h:80-151,284-404,410-513,585-692
 *   - the solver wrapper        src/optimizers/CGDSubspaceOptimizer.cpp:19-98
 *
 * Pinning status: see oracle/README.md (golden values of SURVEY.md section 8c,
 * the reference's own data/testpoly.txt minima, and bit-exact agreement of the
 * minimiser with the reference's minimize_nrc.h built as oracle/_ref).
 */
#ifndef RDIS_ORACLE_H_
#define RDIS_ORACLE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { RO_KIND_BA = 0, RO_KIND_NLP = 1 };

/* exit reason of the CG loop (low byte of status) */
enum {
    RO_EXIT_FTOL = 0,      /* minimize_nrc.h:649 */
    RO_EXIT_GTOL = 1,      /* minimize_nrc.h:663 */
    RO_EXIT_GGZERO = 2,    /* minimize_nrc.h:675 */
    RO_EXIT_ITMAX = 3,     /* minimize_nrc.h:690 (thrown, swallowed by CGD) */
    RO_EXIT_DBRENT_ITMAX = 4, /* minimize_nrc.h:403 (thrown, swallowed by CGD) */
    RO_EXIT_NAN = 5,       /* CGDSubspaceOptimizer.cpp:175 assert */
    RO_EXIT_EMPTY = 6      /* CGDSubspaceOptimizer.cpp:26-29 */
};
#define RO_STATUS_ROLLED_BACK 0x100 /* CGDSubspaceOptimizer.cpp:66-80 */

typedef struct ro_problem ro_problem;

/* ---- single-factor arithmetic ------------------------------------------- */
/* vals = [rx,ry,rz,tx,ty,tz,f,k1,k2,X,Y,Z] (BundleAdjustmentCommon.h:36-59) */
double ro_ba_factor_eval(const double vals[12], double obsx, double obsy);
/* two derivatives of the same factor: the reference's forward chain, operation by operation
 * (BundleAdjustmentFactor.cpp:351-554; rounds like the reference), and an adjoint sweep derived
 * independently from the camera model (the derivation the device kernels use) */
double ro_ba_factor_grad_ref(const double vals[12], double obsx, double obsy,
                             double grad[12]);
double ro_ba_factor_grad(const double vals[12], double obsx, double obsy,
                         double grad[12]);
enum { RO_BA_DERIV_REFCHAIN = 0, RO_BA_DERIV_ADJOINT = 1, RO_BA_DERIV_ADJOINT_DEVICE = 2 };
/* The three named last-place differences between the reference's factor arithmetic and the device's parity option (plan
 * option factor_rounding = 1, rdis_amd/csrc/factors.hpp compiled without contraction), as run-time switches of THIS
 * restatement -- so that a device run can be compared with a CPU run end to end with ==, and so that what separates that
 * CPU run from the reference-pinned one is a closed, named set:
 *   RO_ARITH_RECIPROCAL    the unit axis and the perspective divide through one reciprocal each (x * (1 / y) for x / y)
 *   RO_ARITH_SINCOS_ANGLE  sine / cosine of the rotation angle by the device's routine (ro_sincos_angle) for the C library's
 *   RO_BA_DERIV_ADJOINT_DEVICE (ro_set_ba_derivative)  the adjoint sweep in the device's association for the reference's
 *                          forward chain
 * All off = the reference's arithmetic = what every pin of this oracle is made with. */
#define RO_ARITH_RECIPROCAL 1
#define RO_ARITH_SINCOS_ANGLE 2
#define RO_ARITH_POW_SMALL_INT 4   /* nonlinear-product factors: x^3 = x x x, x^4 = (x x)(x x) (the device; the reference: std::pow) */
void ro_sincos_angle(double x, double *sn, double *cs);
/* value + 12 partials with all three switches on: what one lane of the device's parity option computes */
double ro_ba_factor_grad_device(const double vals[12], double obsx, double obsy, double grad[12]);

/* ---- problems -------------------------------------------------------------*/
/* BA: factor i reads the 9 variables cam_vid0[i]..+8 and pt_vid0[i]..+2. */
ro_problem *ro_create_ba(int64_t nvars, const double *x0, const double *lo,
                         const double *hi, int64_t nfac, const int64_t *cam_vid0,
                         const int64_t *pt_vid0, const double *obs);
/* NLP: factor i = coeff[i] * prod_k g((x[vid[k]]-cons[k])^expo[k]),
 * k in rowptr[i]..rowptr[i+1], g = sin iff sine[k]. */
ro_problem *ro_create_nlp(int64_t nvars, const double *x0, const double *lo,
                          const double *hi, int64_t nfac, const double *coeff,
                          const int64_t *rowptr, const int64_t *vid,
                          const double *expo, const double *cons,
                          const uint8_t *sine);
/* useExponential per factor (src/NonlinearProductFactor.h:62): value = coeff * exp(-product); values only
 * (the reference's gradient asserts it off, src/NonlinearProductFactor.cpp:110).  NULL clears.  0 on success. */
int ro_nlp_set_exponential(ro_problem *p, const uint8_t *use_exp);
void ro_destroy(ro_problem *p);

/* emulate Variable::assign's "|delta| < 1e-12 => factors not notified" rule
 * (src/Variable.cpp:70-76) together with the cached factor value
 * (src/Factor.h:228-234).  Default 1 (reference-faithful). */
void ro_set_emulate_stale_cache(ro_problem *p, int on);

/* which of the two derivatives the gradient / solver entry points use.  Default
 * RO_BA_DERIV_REFCHAIN (reference-faithful rounding). */
void ro_set_ba_derivative(ro_problem *p, int which);
/* RO_ARITH_* flags of the factor arithmetic (default 0: the reference's) */
void ro_set_arithmetic(ro_problem *p, int flags);
/* order of the objective's sum over the listed factors: RO_SUM_LIST = list order, one after the other (the
 * reference, src/OptimizableFunction.cpp:95-135; default), RO_SUM_PAIRWISE = a tree over runs of 64 -- an
 * experiment's switch (the rounding a device's reduction has), never used to pin anything */
#define RO_SUM_LIST 0
#define RO_SUM_PAIRWISE 1
void ro_set_sum_order(ro_problem *p, int which);
/* The fourth named difference (round 6): how the sums of a solve are ADDED.  RO_SUM_TOPOLOGY_REFERENCE (default): every sum in
 * the reference's order.  RO_SUM_TOPOLOGY_COOPERATIVE: the trees of the device's cooperative solvers (solver_coop.hpp /
 * solver_pipe.hpp: waves of 64 as balanced trees, the waves' sums as entries taken l, l + 64, ... by lane l, a wave sum over the
 * lanes; the slope factor by factor; gg / dgg by owner lane; a wave-owned variable's partials strided over a wave), restated entry
 * for entry in rdis_oracle.c.  wave_vid: the variables a wave owns, in wave order -- those fed by more than 48 listed partials,
 * the longest runs first, ties in list order (rdis_hip.hip: prepare_partition).  With it, RO_ARITH_* and
 * RO_BA_DERIV_ADJOINT_DEVICE on, and the stale cache off, ro_cgd_optimize returns what the device's DEFAULT cooperative path
 * returns, bit for bit (tests/test_gpu_parity.py).  Bundle adjustment only. */
#define RO_SUM_TOPOLOGY_REFERENCE 0
#define RO_SUM_TOPOLOGY_COOPERATIVE 1
#define RO_SUM_TOPOLOGY_LDS 2
void ro_set_sum_topology(ro_problem *p, int kind, int64_t nwave_owned, const int64_t *wave_vid);
/* RO_SUM_TOPOLOGY_LDS: the trees of the device's LDS-resident batch solver (solver_lds.hpp: a workgroup of nt lanes a component;
 * BASELINE configs 3 and 5-S), restated entry for entry: lane l adds the terms l, l + nt, ... in order, a wave's 64 lanes as a
 * balanced tree, the waves' sums as a balanced tree; a camera variable's partials per wave of 64 listed factors of its camera as a
 * tree, the waves' sums in order.  slot_vid: the component's slots in the solver's order (camera blocks ascending, nine each, then
 * point blocks ascending, three each).  That solver's DEFAULT arithmetic contracts a * b + c into fused multiply-adds where the
 * source has them in one expression -- which no C restatement compiled by another compiler reproduces; ro_set_factor_arithmetic
 * plugs in the factor arithmetic from outside: tests/cpp/factors_host.hip is rdis_amd/csrc/factors.hpp itself compiled for the
 * HOST by the same front end (per factor == the device, tests/test_gpu_parity.py).  With both, ro_cgd_optimize returns what the
 * default LDS-resident path returns, bit for bit. */
typedef struct {
    double (*value)(const double *x12, double ox, double oy);
    double (*eval_grad)(const double *x12, double ox, double oy, double *g12);
    double (*value_slope)(const double *x12, const double *d12, double ox, double oy, int camfix, double *slope);
} ro_factor_arith;
void ro_set_factor_arithmetic(ro_problem *p, const ro_factor_arith *ext);   /* ext must outlive the problem's use; NULL: built in */
void ro_set_lds_topology(ro_problem *p, int nt, int64_t nslots, const int64_t *slot_vid);
/* the device's public evaluation entry points on bundle adjustment (rdis_hip_eval, rdis_hip_eval_grad), their sums restated entry
 * for entry (rdis_oracle.c): lanes = 512 entries a chunk; tile_chunks = max(1, min(8, chunks / (8 x the device's compute units))) */
double ro_eval_device_ba(ro_problem *p, int64_t nf, const int64_t *fac, int lanes);
double ro_eval_device_grid(ro_problem *p, int64_t nf, const int64_t *fac, int blocks);   /* nonlinear products: blocks = min(ceil(nf / 256), 8 x compute units) */
double ro_eval_grad_device_ba(ro_problem *p, int64_t nf, const int64_t *fac, double *g, int lanes, int tile_chunks);
/* RO_SUM_TOPOLOGY_GROUP: the sums of the device's solver of tiny components (solver_quad.hpp: G = 4 or 16 lanes a component of at
 * most four free variables -- a point against constant cameras, thousands of them a launch), restated entry for entry
 * (rdis_oracle.c); bundle adjustment, the factor arithmetic from outside (ro_set_factor_arithmetic). */
#define RO_SUM_TOPOLOGY_GROUP 5
void ro_set_group_topology(ro_problem *p, int G);
/* RO_SUM_TOPOLOGY_WG: the sums of the device's plain one-workgroup solver (solver_wg.hpp: cgd_wg_kernel, nt lanes) on a
 * nonlinear-product problem -- BASELINE configs 1 and 2 --, restated entry for entry (rdis_oracle.c).  The device's factor
 * arithmetic there differs from the reference's in two named places: sine and cosine (factors.hpp: nlp_sin / nlp_cos, the routine
 * of the rotation angle, fused; plugged in from outside with ro_set_trig -- tests/cpp/factors_host.hip: fh_sincos) and the third and
 * fourth power by multiplication (RO_ARITH_POW_SMALL_INT) where the reference calls std::pow. */
#define RO_SUM_TOPOLOGY_WG 4
void ro_set_wg_topology(ro_problem *p, int nt);
/* ... and the grid solver (solver_stream.hpp: nwg workgroups of nt lanes on one component): the same sums over the grid's lanes, every
 * wave an entry of the exchange */
void ro_set_stream_topology(ro_problem *p, int nt, int nwg);
void ro_set_trig(ro_problem *p, void (*sincos_fn)(double x, double *sn, double *cs));
/* RO_SUM_TOPOLOGY_PTM: the sums of the device's point-major streaming solver (solver_ptm.hpp: a workgroup of nt lanes a component
 * whose camera blocks stay in LDS while its point blocks stream -- BASELINE config 5-L), restated entry for entry (rdis_oracle.c).
 * cam_vid0 / pt_vid0: the component's camera blocks (ascending) and point blocks in the plan's order (by number of listed factors
 * descending, then by their cameras, whole wave-chunks dealt over sixteen runs: the caller restates rdis_hip.hip's rule); blk: the
 * slots a wave asks for together (ptm_api.hpp: PTM_BLK); K: the workgroups that share the component (1, or a group of up to 16:
 * cgd_ptmg_kernel -- chunk c is workgroup c mod K's, every wave of the group an entry of the exchange, the workgroups' partial camera
 * gradients added in rank order; K < 0: -K workgroups as a WIDE group -- one large component on a large share of the device: a
 * workgroup's waves are added first, as the 16-tree, and the workgroup is one entry of the exchange).  That solver evaluates trials in MATRIX form against per-camera records
 * (factors.hpp: ba_camera_trial, ba_camera_trial_dir, ba_trial_value, ba_trial_slope): `ar` plugs that arithmetic in from outside like
 * ro_set_factor_arithmetic does for the vector form (which the gradient uses: both must be set). */
#define RO_SUM_TOPOLOGY_PTM 3
typedef struct {
    void (*camera_trial)(const double *xc9, double *TR16);
    void (*camera_trial_dir)(const double *xc9, const double *dc9, double *DR10);
    /* value of one factor at point q3; with DR10 and slope != NULL also its slope along (DR10, e3) */
    double (*trial)(const double *TR16, const double *DR10, const double *q3, const double *e3, double ox, double oy, double *slope);
} ro_ptm_arith;
void ro_set_ptm_local(ro_problem *p, const int64_t *wg_chunk0);   /* after ro_set_ptm_topology with K < 0: [K + 1] chunk ranges, LOCAL camera numbering */
void ro_set_ptm_round_slots(ro_problem *p, int round_slots);   /* after ro_set_ptm_topology; 1 (default) or 2: rdis_oracle.c, "gradient" */
void ro_set_ptm_topology(ro_problem *p, int nt, int blk, int K, int64_t ncb, const int64_t *cam_vid0, int64_t npb, const int64_t *pt_vid0,
                         const ro_ptm_arith *ar);
/* process-wide experiment flags; never used to pin anything.  bit 0: reciprocals in place of the projection's divisions, the
 * device's form.  bit 1 (round 5): the slope of a line-search trial added factor by factor, sum_f (sum_k partial_fk xi_k) -- the
 * association the device's fused trials use -- in place of the reference's gradient times direction (Df1dim::df,
 * minimize_nrc.h:439-447); bundle adjustment only.  This is the switch that turns the oracle's population of end values on
 * ladybug 5 / 30 into the device's (tests/test_oracle.py::test_slope_association_moves_the_population).  bit 2 (with bit 1): those
 * per-factor terms added as a balanced tree instead of in list order -- the shape of a device's reduction. */
void ro_set_experiment(int flags);

void ro_assign(ro_problem *p, int64_t nvid, const int64_t *vid, const double *val);
void ro_get_x(const ro_problem *p, int64_t nvid, const int64_t *vid, double *out);

/* sum over the listed factors in list order (fac == NULL: all factors 0..n-1) */
double ro_eval_factors(ro_problem *p, int64_t nf, const int64_t *fac);
/* dense gradient (length nvars, zero where untouched); per-variable
 * contributions are added in factor-list order.  merge != 0 runs the
 * reference's sorted-(vid,value)-vector merge per factor (same result,
 * reference cost model). */
void ro_compute_gradient(ro_problem *p, int64_t nf, const int64_t *fac,
                         double *g, int merge);
/* per-factor values and 12 (BA) / arity (NLP, CSR order) partials */
void ro_eval_each(ro_problem *p, int64_t nf, const int64_t *fac, double *fvals);
void ro_grad_each_ba(ro_problem *p, int64_t nf, const int64_t *fac, double *g12);
/* pixel residuals (2 per factor) and their Jacobian rows (2 x 12 per factor): the usual
 * bundle-adjustment least-squares model, used by the LM oracle's residual model 2 */
void ro_ba_factor_resjac(const double vals[12], double obsx, double obsy, double res[2], double J[24]);
void ro_resjac_each_ba(ro_problem *p, int64_t nf, const int64_t *fac, double *res2, double *J24);

/* ---- the solver ------------------------------------------------------------*/
typedef struct {
    double fret;     /* returned value */
    double delta;    /* deltaFval */
    double finit;    /* initialFval */
    int32_t iters;   /* Frprmn::iter (index of the last iteration started) */
    int32_t status;  /* RO_EXIT_* | RO_STATUS_ROLLED_BACK */
    int64_t nfeval;  /* calls of SubfunctionFD::operator() */
    int64_t ngeval;  /* calls of SubfunctionFD::df */
} ro_result;

/* CGDSubspaceOptimizer::optimize over free variables free_vid[0..nfree) and
 * factors fac[0..nf) (NULL = all).  xval is in free_vid order, overwritten
 * with the final clamped values; the problem's variables are left assigned to
 * them.  merge selects the gradient cost model (see ro_compute_gradient). */
void ro_cgd_optimize(ro_problem *p, int64_t nfree, const int64_t *free_vid,
                     int64_t nf, const int64_t *fac, double *xval,
                     int32_t maxiters, double ftol, int merge, ro_result *out);

/* generic entry to the restated minimiser, for the bit-exact comparison with
 * oracle/_ref (the reference's own minimize_nrc.h): minimise func over n
 * variables starting from x. */
typedef double (*ro_func_cb)(void *ctx, const double *x);
typedef void (*ro_grad_cb)(void *ctx, const double *x, double *g);
/* returns exit reason; x <- Frprmn::p, *fret <- Frprmn::fret, *iter <- Frprmn::iter */
int ro_frprmn(int n, double *x, ro_func_cb f, ro_grad_cb df, void *ctx,
              int maxiters, double ftol, double *fret, int *iter);

/* ---- replay of a device trace (see rdis_oracle.c) ---------------------------*/
#define RO_NEAR_AMPLIFICATION 1.0e5

typedef struct {
    int64_t consumed;        /* trace records consumed */
    int64_t first_mismatch;  /* index of the first record that did not line up, or -1 */
    int64_t step_mismatches; /* step lengths / line minima not bit-identical */
    int64_t tag_mismatches;  /* record kinds out of sequence */
    int32_t underrun;        /* the oracle wanted more records than the trace holds */
    int32_t reason;          /* oracle's exit reason when fed the device's values */
    int32_t iters;
    double fret, finit;
    double max_step_rel;     /* worst |a_own - a_dev| / |a_own| among mismatching steps */
    double max_f_rel;        /* worst |f_own - f_dev| / sum|factor values| */
    double max_slope_rel;    /* worst |s_own - s_dev| / sum|g_j xi_j| */
    double max_iter_rel;     /* worst relative difference of test / gg / dgg */
    double max_vec_rel;      /* worst drift of p / xi (inf-norm relative) before each re-sync */
    double max_f_rel_near;   /* max_f_rel / max_slope_rel restricted to the ordinary trial points:
                                |f| <= 4 |f(x0)| + 1 and a first-order rounding bound of the sum of at most
                                RO_NEAR_AMPLIFICATION x eps x sum|factor values| (typically 2e3..3e4).
                                Far-out bracketing steps (f up to 1e4 x f(x0)) and steps that put a point
                                next to a camera's pole (P_z -> 0) are ill-conditioned */
    double max_slope_rel_near;
    int32_t last_near;       /* internal */
    int64_t synced_iters;    /* line searches started from the device's dumped p, xi */
    double pending_slope;    /* internal */
    double max_f_far_ulps;   /* far-out trial points: worst |f_own - f_dev| in units of (the change of f
                                under a one-ulp move of every free variable + eps * sum|factor values|) */
    double max_f_bound;      /* worst |f_own - f_dev| / (eps * first-order rounding bound of the sum at that
                                point), all trial points: the conditioning-aware form of max_f_rel */
    double max_slope_bound;  /* the same for the slope along the line */
} ro_replay_report;

/* vdump (may be NULL): the device's p and xi at the start of each of the first
 * dump_iters line searches, [dump_iters][2][nfree]; the oracle adopts them so
 * that both sides evaluate at bit-identical points.
 * x_end (may be NULL) receives the oracle's unclamped end point Frprmn::p */
void ro_cgd_replay(ro_problem *p, int64_t nfree, const int64_t *free_vid, int64_t nf,
                   const int64_t *fac, const double *xstart, int32_t maxiters, double ftol,
                   const double *trace, int64_t nrec, const double *vdump, int32_t dump_iters,
                   double *x_end, ro_replay_report *rep);

/* the oracle's own trace in the same record format; returns the record count */
int64_t ro_cgd_record(ro_problem *p, int64_t nfree, const int64_t *free_vid, int64_t nf,
                      const int64_t *fac, const double *xstart, int32_t maxiters, double ftol,
                      double *trace, int64_t cap);

#ifdef __cplusplus
}
#endif
/* Connected components of the factor graph once the variables with assigned[v] != 0 are fixed
 * (Component::createChildren / Component::init, src/Component.cpp:508-549, 60-79): variable and
 * factor lists ascending, components by (number of variables, smallest variable id).  Returns
 * the number of components; free_ptr / fac_ptr: nvars + 1 entries, free_vid: nvars, fac_id: nfac. */
int64_t ro_components(const ro_problem *p, const uint8_t *assigned, int64_t *free_ptr, int64_t *free_vid,
                      int64_t *fac_ptr, int64_t *fac_id);

#endif /* RDIS_ORACLE_H_ */
