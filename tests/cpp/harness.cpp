// Test harness for the C++ drop-in surface (rdis_amd/host): small C entry points that
// use the mirrored classes exactly the way the reference's callers do
// (BCDOptimizer with one block = all variables, src/optimizers/BCDOptimizer.cpp:149;
//  RDISOptimizer::getValueFromDomain, src/RDISOptimizer.cpp:1039-1083).
#include <chrono>
#include <cmath>
#include <cstring>
#include <iostream>
#include <memory>

#include "../../rdis_amd/host/rdis_host.h"
#include "../../rdis_amd/host/rdis_levels.h"

using namespace rdis;

extern "C" {

// ---- host-only: what the loaders build (no GPU needed) --------------------------------
// out_sizes = {nvars, nfac, kind}; arrays may be NULL to query sizes
int harness_pack_bal(const char* path, long long ncams, long long npts, long long* out_sizes, double* xinit,
                     double* lo, double* hi, long long* cam_vid0, long long* pt_vid0, double* obs) {
    try {
        BundleAdjustmentFunction f;
        if (!f.load(path, ncams, npts)) return -1;
        const OptimizableFunction::Packed& P = f.packed();
        out_sizes[0] = f.getNumVars(); out_sizes[1] = (long long)f.getFactors().size(); out_sizes[2] = P.kind;
        if (xinit) std::memcpy(xinit, f.getInitialState().data(), sizeof(double) * f.getInitialState().size());
        if (lo) std::memcpy(lo, P.lo.data(), sizeof(double) * P.lo.size());
        if (hi) std::memcpy(hi, P.hi.data(), sizeof(double) * P.hi.size());
        if (cam_vid0) for (size_t i = 0; i < P.cam_vid0.size(); ++i) { cam_vid0[i] = P.cam_vid0[i]; pt_vid0[i] = P.pt_vid0[i]; }
        if (obs) std::memcpy(obs, P.obs.data(), sizeof(double) * P.obs.size());
        return 0;
    } catch (const std::exception& e) { std::cerr << "harness_pack_bal: " << e.what() << std::endl; return -2; }
}

// load -> assign the initial state -> save -> load again: sizes and values survive
int harness_bal_round_trip(const char* path, long long ncams, long long npts, const char* out_path) {
    try {
        BundleAdjustmentFunction f, g;
        if (!f.load(path, ncams, npts)) return -1;
        f.assignAll(f.getInitialState());
        if (!f.save(out_path)) return -3;
        if (!g.load(out_path)) return -4;
        if (g.getNumVars() != f.getNumVars() || g.getFactors().size() != f.getFactors().size()) return 1;
        if (g.getInitialState() != f.getInitialState()) return 2;
        const OptimizableFunction::Packed &a = f.packed(), &b = g.packed();
        if (a.cam_vid0 != b.cam_vid0 || a.pt_vid0 != b.pt_vid0 || a.obs != b.obs || a.lo != b.lo || a.hi != b.hi) return 3;
        return 0;
    } catch (const std::exception& e) { std::cerr << "harness_bal_round_trip: " << e.what() << std::endl; return -2; }
}

// which = 0: polynomial file, 1: default high-dimensional sinusoid. out_sizes = {nvars, nfac, nnz}
int harness_pack_nlp(int which, const char* path, long long* out_sizes, double* lo, double* hi, double* coeff,
                     long long* rowptr, long long* vid, double* expo, double* cons, unsigned char* sine) {
    try {
        std::unique_ptr<PolynomialFunction> f;
        if (which == 0) { f.reset(new PolynomialFunction()); if (!f->load(path)) return -1; }
        else f = PolynomialFunction::makeHighDimSinusoid();
        const OptimizableFunction::Packed& P = f->packed();
        out_sizes[0] = f->getNumVars(); out_sizes[1] = (long long)f->getFactors().size(); out_sizes[2] = (long long)P.vid.size();
        if (lo) {
            std::memcpy(lo, P.lo.data(), sizeof(double) * P.lo.size());
            std::memcpy(hi, P.hi.data(), sizeof(double) * P.hi.size());
            std::memcpy(coeff, P.coeff.data(), sizeof(double) * P.coeff.size());
            for (size_t i = 0; i < P.rowptr.size(); ++i) rowptr[i] = P.rowptr[i];
            for (size_t i = 0; i < P.vid.size(); ++i) { vid[i] = P.vid[i]; expo[i] = P.expo[i]; cons[i] = P.cons[i]; sine[i] = P.sine[i]; }
        }
        return 0;
    } catch (const std::exception& e) { std::cerr << "harness_pack_nlp: " << e.what() << std::endl; return -2; }
}

// ---- GPU: the reference's call pattern through the plugin surface ------------------------
// mode 0: all variables, all factors (BCD, one block).  mode 1: block = camera 0 + point 0 free,
// the other variables assigned to the file values, factors = those touching the block
// (getValueFromDomain).  out = {fret, deltaFval, f_before, f_after(eval of the same factor list),
// iters, status, nfeval, ngeval, post_ok}
int harness_ba_cgd(const char* path, long long ncams, long long npts, int mode, int maxit, double ftol,
                   double* out, double* xval_out, long long* nfree_out) {
    try {
        BundleAdjustmentFunction f;
        if (!f.load(path, ncams, npts)) return -1;
        f.assignAll(f.getInitialState());
        HipCGDSubspaceOptimizer ssopt(f);
        Options o; o.set("SSmaxit", maxit); o.set("SSftol", ftol);
        ssopt.setParameters(o);
        VariablePtrVec vars;
        FactorPtrVec facs;
        if (mode == 0) { vars = f.getVariables(); facs = f.getFactors(); }
        else {
            VariableID lo, hi;
            f.getBlockRangeByBlkId(0, lo, hi);
            for (VariableID v = lo; v <= hi; ++v) vars.push_back(f.getVariables()[(size_t)v]);
            f.getBlockRangeByBlkId(f.getNumCameras(), lo, hi);
            for (VariableID v = lo; v <= hi; ++v) vars.push_back(f.getVariables()[(size_t)v]);
            for (Factor* fa : f.getFactors()) {
                const BundleAdjustmentFactor* b = static_cast<const BundleAdjustmentFactor*>(fa);
                if (b->getCameraID() == 0 || b->getPointID() == 0) facs.push_back(fa);
            }
        }
        NumericVec xval(vars.size());
        for (size_t i = 0; i < vars.size(); ++i) xval[i] = vars[i]->eval();
        Numeric ferr = 0;
        const Numeric before = f.evalFactors(facs, ferr);
        Numeric delta = 0;
        const Numeric fret = ssopt.optimize(vars, facs, xval, delta, false);
        // post-conditions of optimize(): vars assigned to xval (clamped), others untouched
        bool ok = true;
        for (size_t i = 0; i < vars.size(); ++i) {
            ok = ok && vars[i]->isAssigned() && vars[i]->eval() == xval[i];
            ok = ok && xval[i] >= vars[i]->getDomain().min() && xval[i] <= vars[i]->getDomain().max();
        }
        if (mode == 1)
            for (size_t v = 9; v < (size_t)f.getNumCameras() * 9; ++v) ok = ok && f.getVariables()[v]->eval() == f.getInitialState()[v];
        const Numeric after = f.evalFactors(facs, ferr);
        out[0] = fret; out[1] = delta; out[2] = before; out[3] = after; out[4] = ssopt.lastIters();
        out[5] = ssopt.lastStatus(); out[6] = (double)ssopt.lastFEvals(); out[7] = (double)ssopt.lastGEvals(); out[8] = ok ? 1 : 0;
        *nfree_out = (long long)vars.size();
        if (xval_out) std::memcpy(xval_out, xval.data(), sizeof(double) * xval.size());
        // empty factor list: returns 0, delta 0, xval untouched (CGDSubspaceOptimizer.cpp:26-29)
        NumericVec keep = xval;
        Numeric d2 = 1;
        if (ssopt.optimize(vars, FactorPtrVec(), keep, d2, false) != 0 || d2 != 0 || keep != xval) out[8] = 0;
        return 0;
    } catch (const std::exception& e) { std::cerr << "harness_ba_cgd: " << e.what() << std::endl; return -2; }
}

// polynomial file, start (x0, x1); out = {fret, delta, x0, x1, iters, status}
int harness_poly_cgd(const char* path, double x0, double x1, int maxit, double* out) {
    try {
        PolynomialFunction f;
        if (!f.load(path)) return -1;
        NumericVec xval(2); xval[0] = x0; xval[1] = x1;
        f.assignAll(xval);
        HipCGDSubspaceOptimizer ssopt(f);
        Options o; o.set("SSmaxit", maxit);
        ssopt.setParameters(o);
        Numeric delta = 0;
        const Numeric fret = ssopt.optimize(f.getVariables(), f.getFactors(), xval, delta, false);
        out[0] = fret; out[1] = delta; out[2] = xval[0]; out[3] = xval[1]; out[4] = ssopt.lastIters(); out[5] = ssopt.lastStatus();
        return 0;
    } catch (const std::exception& e) { std::cerr << "harness_poly_cgd: " << e.what() << std::endl; return -2; }
}

// sibling components in one launch: every point of a BAL problem is its own component once the
// cameras are fixed (the shape RDIS produces on ladybug).  out = {sum fret, sum
// delta, ncomp, total iterations, f_before, f_after}
int harness_ba_points_batch(const char* path, long long ncams, long long npts, int maxit, double* out) {
    try {
        BundleAdjustmentFunction f;
        if (!f.load(path, ncams, npts)) return -1;
        f.assignAll(f.getInitialState());
        HipCGDSubspaceOptimizer ssopt(f);
        Options o; o.set("SSmaxit", maxit);
        ssopt.setParameters(o);
        std::vector<HipCGDSubspaceOptimizer::Component> comps((size_t)f.getNumPoints());
        for (Factor* fa : f.getFactors()) comps[(size_t)static_cast<BundleAdjustmentFactor*>(fa)->getPointID()].factors.push_back(fa);
        for (long long p = 0; p < f.getNumPoints(); ++p) {
            VariableID lo, hi;
            f.getBlockRangeByBlkId(f.getNumCameras() + p, lo, hi);
            for (VariableID v = lo; v <= hi; ++v) {
                comps[(size_t)p].vars.push_back(f.getVariables()[(size_t)v]);
                comps[(size_t)p].xval.push_back(f.getVariables()[(size_t)v]->eval());
            }
        }
        const Numeric before = f.eval();
        const Numeric total = ssopt.optimizeBatch(comps, false);
        const Numeric after = f.eval();
        double sd = 0, its = 0;
        for (const auto& c : comps) { sd += c.deltaFval; its += c.iters + 1; }
        out[0] = total; out[1] = sd; out[2] = (double)comps.size(); out[3] = its; out[4] = before; out[5] = after;
        return 0;
    } catch (const std::exception& e) { std::cerr << "harness_ba_points_batch: " << e.what() << std::endl; return -2; }
}

// The RDIS step in full: with the cameras assigned, un-assign the points, let the optimizer find
// the children (connected components on the device) and solve them in one launch.
// out = {sum fret, sum delta, ncomp, total iterations, f_before, f_after, min #vars, max #vars}
int harness_ba_children_batch(const char* path, long long ncams, long long npts, int maxit, double* out) {
    try {
        BundleAdjustmentFunction f;
        if (!f.load(path, ncams, npts)) return -1;
        const NumericVec x0 = f.getInitialState();
        f.assignAll(x0);
        const Numeric before = f.eval();
        HipCGDSubspaceOptimizer ssopt(f);
        Options o; o.set("SSmaxit", maxit);
        ssopt.setParameters(o);
        for (VariableID v = 9 * f.getNumCameras(); v < (VariableID)f.getNumVars(); ++v) f.getVariables()[(size_t)v]->unassign();
        std::vector<HipCGDSubspaceOptimizer::Component> comps = ssopt.createChildren();
        size_t mn = (size_t)-1, mx = 0;
        for (auto& c : comps) {
            mn = std::min(mn, c.vars.size()); mx = std::max(mx, c.vars.size());
            for (const Variable* v : c.vars) c.xval.push_back(x0[(size_t)v->getID()]);
        }
        const Numeric total = ssopt.optimizeBatch(comps, false);
        const Numeric after = f.eval();
        double sd = 0, its = 0;
        for (const auto& c : comps) { sd += c.deltaFval; its += c.iters + 1; }
        out[0] = total; out[1] = sd; out[2] = (double)comps.size(); out[3] = its; out[4] = before; out[5] = after;
        out[6] = (double)mn; out[7] = (double)mx;
        return 0;
    } catch (const std::exception& e) { std::cerr << "harness_ba_children_batch: " << e.what() << std::endl; return -2; }
}

// The same over SEVERAL devices from one process (OptimizableFunction::setDevices; ndev > 0: ndev contexts on device 0 --
// what one GPU can test; ndev < 0: -ndev distinct GPUs): which = 0 the point components (cameras assigned), 1 the camera components (points assigned).
// out = {sum fret, ncomp, f_before, f_after, plan cache entries}; per component (in createChildren's order): fret, iters,
// f-evaluations; x_out: every variable's value afterwards
int harness_ba_children_batch_devices(const char* path, long long ncams, long long npts, int maxit, int which, int ndev, int rounds,
                                      double* out, double* fret_out, long long* iters_out, long long* nfe_out, double* x_out) {
    try {
        BundleAdjustmentFunction f;
        if (!f.load(path, ncams, npts)) return -1;
        if (ndev > 1) f.setDevices(std::vector<int>((size_t)ndev, 0));
        if (ndev < -1) {   // ndev < 0: -ndev DISTINCT GPUs (a multi-GPU node)
            std::vector<int> devs;
            for (int d = 0; d < -ndev; ++d) devs.push_back(d);
            f.setDevices(devs);
        }
        const NumericVec x0 = f.getInitialState();
        f.assignAll(x0);
        const Numeric before = f.eval();
        HipCGDSubspaceOptimizer ssopt(f);
        Options o; o.set("SSmaxit", maxit);
        ssopt.setParameters(o);
        Numeric total = 0;
        size_t ncomp = 0;
        for (int rd = 0; rd < rounds; ++rd) {   // (a second round starts from what the first left: values cross between the devices)
            const int w = (which + rd) & 1;
            const VariableID lo = w == 0 ? 9 * f.getNumCameras() : 0, hi = w == 0 ? (VariableID)f.getNumVars() : 9 * f.getNumCameras();
            for (VariableID v = lo; v < hi; ++v) f.getVariables()[(size_t)v]->unassign();
            std::vector<HipCGDSubspaceOptimizer::Component> comps = ssopt.createChildren();
            for (auto& c : comps) for (const Variable* v : c.vars) c.xval.push_back(rd == 0 ? x0[(size_t)v->getID()] : x_out[(size_t)v->getID()]);
            total = ssopt.optimizeBatch(comps, false);
            ncomp = comps.size();
            for (size_t c = 0; c < comps.size(); ++c) { fret_out[c] = comps[c].fret; iters_out[c] = comps[c].iters; nfe_out[c] = comps[c].nfeval; }
            for (size_t i = 0; i < f.getVariables().size(); ++i) x_out[i] = f.getVariables()[i]->eval();
        }
        const Numeric after = f.eval();
        out[0] = total; out[1] = (double)ncomp; out[2] = before; out[3] = after; out[4] = (double)ssopt.planCacheEntries();
        return 0;
    } catch (const std::exception& e) { std::cerr << "harness_ba_children_batch_devices: " << e.what() << std::endl; return -2; }
}

// ... and the level driver over ndev devices.  out = {final value, value before, sweeps, nodes}
int harness_level_driver_devices(const char* path, long long ncams, long long npts, int maxit, int max_sweeps, double blkpct, int ndev,
                                 double* out, double* x_out) {
    try {
        BundleAdjustmentFunction f;
        if (!f.load(path, ncams, npts)) return -1;
        if (ndev > 1) f.setDevices(std::vector<int>((size_t)ndev, 0));
        f.assignAll(f.getInitialState());
        const Numeric before = f.eval();
        HipCGDSubspaceOptimizer ssopt(f);
        Options o; o.set("SSmaxit", maxit);
        ssopt.setParameters(o);
        HipRDISLevelOptimizer rdis(f, ssopt);
        Options ro; ro.set("AVblkpct", blkpct); ro.set("maxSweeps", max_sweeps); ro.set("batch", 1);
        rdis.setParameters(ro);
        const Numeric fin = rdis.optimize(false);
        out[0] = fin; out[1] = before; out[2] = rdis.sweepsDone(); out[3] = (double)rdis.nodes().size();
        for (size_t i = 0; i < f.getVariables().size(); ++i) x_out[i] = f.getVariables()[i]->eval();
        return 0;
    } catch (const std::exception& e) { std::cerr << "harness_level_driver_devices: " << e.what() << std::endl; return -2; }
}

// The level driver under the reference's schedule (HipRDISLevelOptimizer::optimizeReferenceSchedule): per-node iterative
// improvement and random restarts.  out = {final value, value before, nodes, steps}; trace rows of 10 doubles:
// node, kind, nrr, va, fret, delta, value, newMin, start hash (high 32 bits, low 32 bits).  Returns the number of steps.
long long harness_level_reference(const char* path, long long ncams, long long npts, int maxit, double blkpct, double seppct,
                                  int nrr_per_lvl, int max_na_to_rr, double seed, int ndev, double max_calls, double steptol, double* out, double* trace_out,
                                  long long trace_cap, double* x_out) {
    try {
        BundleAdjustmentFunction f;
        if (!f.load(path, ncams, npts)) return -1;
        if (ndev > 1) f.setDevices(std::vector<int>((size_t)ndev, 0));
        f.assignAll(f.getInitialState());
        const Numeric before = f.eval();
        HipCGDSubspaceOptimizer ssopt(f);
        Options o; o.set("SSmaxit", maxit);
        ssopt.setParameters(o);
        HipRDISLevelOptimizer rdis(f, ssopt);
        Options ro; ro.set("AVblkpct", blkpct); ro.set("sepPiecePct", seppct); ro.set("nRRperLvl", nrr_per_lvl);
        ro.set("maxNAtoRR", max_na_to_rr); ro.set("restartSeed", seed); ro.set("maxCalls", max_calls); ro.set("steptol", steptol);
        rdis.setParameters(ro);
        const Numeric fin = rdis.optimizeReferenceSchedule(false);
        const auto& tr = rdis.refTrace();
        out[0] = fin; out[1] = before; out[2] = (double)rdis.nodes().size(); out[3] = (double)tr.size();
        for (size_t i = 0; i < tr.size() && (long long)i < trace_cap; ++i) {
            double* r = trace_out + 10 * i;
            r[0] = tr[i].node; r[1] = tr[i].kind; r[2] = tr[i].nrr; r[3] = tr[i].va; r[4] = tr[i].fret; r[5] = tr[i].delta;
            r[6] = tr[i].value; r[7] = tr[i].newMin; r[8] = (double)(tr[i].startHash >> 32); r[9] = (double)(tr[i].startHash & 0xFFFFFFFFull);
        }
        if (x_out) for (size_t i = 0; i < f.getVariables().size(); ++i) x_out[i] = f.getVariables()[i]->eval();
        return (long long)tr.size();
    } catch (const std::exception& e) { std::cerr << "harness_level_reference: " << e.what() << std::endl; return -2; }
}

// LMSubspaceOptimizer's place taken by HipLMSubspaceOptimizer: all variables of a BAL subset.
// out = {fret, delta, f_before, f_after, iterations, stop, linear solves}
int harness_ba_lm(const char* path, long long ncams, long long npts, int maxit, double* out, double* x_out) {
    try {
        BundleAdjustmentFunction f;
        if (!f.load(path, ncams, npts)) return -1;
        NumericVec x = f.getInitialState();
        f.assignAll(x);
        const Numeric before = f.eval();
        HipLMSubspaceOptimizer lm(f);
        Options o; o.set("SSmaxit", maxit);
        lm.setParameters(o);
        Numeric delta = 0;
        const Numeric fret = lm.optimize(f.getVariables(), f.getFactors(), x, delta, false);
        const Numeric after = f.eval();
        out[0] = fret; out[1] = delta; out[2] = before; out[3] = after; out[4] = lm.lastIters(); out[5] = lm.lastStop(); out[6] = lm.lastLinearSolves();
        for (size_t i = 0; i < x.size(); ++i) x_out[i] = x[i];
        return 0;
    } catch (const std::exception& e) { std::cerr << "harness_ba_lm: " << e.what() << std::endl; return -2; }
}

// ---- host-only: the separator the level driver chooses for the whole function (no GPU needed).
// sep_out: capacity nvars; returns the number of separator variables (or < 0)
long long harness_separator(const char* path, long long ncams, long long npts, double blkpct, long long* sep_out) {
    try {
        BundleAdjustmentFunction f;
        if (!f.load(path, ncams, npts)) return -1;
        std::vector<VariableID> vars((size_t)f.getNumVars()), sep;
        std::vector<FactorID> facs(f.getFactors().size());
        for (size_t i = 0; i < vars.size(); ++i) vars[i] = (VariableID)i;
        for (size_t i = 0; i < facs.size(); ++i) facs[i] = (FactorID)i;
        HipRDISLevelOptimizer::chooseSeparator(f, vars, facs, (size_t)std::llround(blkpct * (double)vars.size()), sep);
        for (size_t i = 0; i < sep.size(); ++i) sep_out[i] = sep[i];
        return (long long)sep.size();
    } catch (const std::exception& e) { std::cerr << "harness_separator: " << e.what() << std::endl; return -2; }
}

// ---- GPU: the level driver (rdis_levels.h) on a BAL problem from the file's state.
// out = {final value, initial value, sweeps, #nodes, #leaves, #split nodes, wall ms of optimize(), decomposition ms,
//        largest separator, #trace steps, monotone (1/0), max |running sum - evaluated|};
// trace_out (may be NULL): up to trace_cap rows {sweep, depth, kind, ncomp, nvars, nfactors, objective, ms};
// x_out (may be NULL): final values of all variables
int harness_level_driver(const char* path, long long ncams, long long npts, int maxit, int max_sweeps, double blkpct,
                         int batch, double* out, double* trace_out, long long trace_cap, double* x_out) {
    try {
        BundleAdjustmentFunction f;
        if (!f.load(path, ncams, npts)) return -1;
        f.assignAll(f.getInitialState());
        const Numeric before = f.eval();
        HipCGDSubspaceOptimizer ssopt(f);
        Options o; o.set("SSmaxit", maxit);
        ssopt.setParameters(o);
        HipRDISLevelOptimizer rdis(f, ssopt);
        Options ro; ro.set("AVblkpct", blkpct); ro.set("maxSweeps", max_sweeps); ro.set("batch", batch);
        rdis.setParameters(ro);
        const auto t0 = std::chrono::steady_clock::now();
        const Numeric fin = rdis.optimize(false);
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        size_t nleaf = 0, nsplit = 0, maxsep = 0;
        for (const auto& nd : rdis.nodes()) { (nd.leaf ? nleaf : nsplit)++; maxsep = std::max(maxsep, nd.separator.size()); }
        bool mono = true; double prev = before;
        for (const auto& st : rdis.trace()) { mono = mono && st.objective <= prev; prev = st.objective; }
        out[0] = fin; out[1] = before; out[2] = rdis.sweepsDone(); out[3] = (double)rdis.nodes().size(); out[4] = (double)nleaf;
        out[5] = (double)nsplit; out[6] = ms; out[7] = rdis.decompositionMs(); out[8] = (double)maxsep; out[9] = (double)rdis.trace().size();
        out[10] = mono ? 1 : 0; out[11] = std::fabs(prev - fin);
        if (trace_out)
            for (size_t i = 0; i < rdis.trace().size() && (long long)i < trace_cap; ++i) {
                const auto& st = rdis.trace()[i];
                double* r = trace_out + 8 * i;
                r[0] = st.sweep; r[1] = st.depth; r[2] = st.kind; r[3] = (double)st.ncomp; r[4] = (double)st.nvars; r[5] = (double)st.nfactors;
                r[6] = st.objective; r[7] = st.ms;
            }
        if (x_out) for (size_t i = 0; i < f.getVariables().size(); ++i) x_out[i] = f.getVariables()[i]->eval();
        return 0;
    } catch (const std::exception& e) { std::cerr << "harness_level_driver: " << e.what() << std::endl; return -2; }
}

// ---- GPU: the unchanged-caller path.  RDISOptimizer calls ssopt.optimize one component at a time
// (src/RDISOptimizer.cpp:1067); on ladybug 5 cameras / 30 points its run makes 80 calls with the 48
// variables of the 5 cameras and one point and 651 calls with the 3 variables of a single point
// (SURVEY.md 3.2b).  The same call shapes, in rounds (one separator call, then the next few points in
// turn), every call starting from the values the previous ones left.  cache: plan cache entries (0 = off).
// out = {wall ms of all calls, #calls, final value of the function, sum of returned values, cache hits, cache misses}
static int call_shapes(const char* path, int nsep, int npt_calls, int maxit, int cache, double cache_bytes, double* out, double* x_out);
int harness_call_shapes(const char* path, int nsep, int npt_calls, int maxit, int cache, double* out, double* x_out) {
    return call_shapes(path, nsep, npt_calls, maxit, cache, -1.0, out, x_out);
}
// ... with the cache's byte budget set too (cache_bytes < 0: left at its default); out has three more entries:
// {.., calls served by the transient path because their plan could not be kept, plans resident at the end, their bytes}
int harness_call_shapes_budget(const char* path, int nsep, int npt_calls, int maxit, int cache, double cache_bytes, double* out, double* x_out) {
    return call_shapes(path, nsep, npt_calls, maxit, cache, cache_bytes, out, x_out);
}
// the optimizer outlives its function: the function's destructor drops the optimizer's cached plans (they belong to its
// device problem), the optimizer's own destructor then has nothing left to touch.  Returns the plans cached before.
int harness_optimizer_outlives_function(const char* path) {
    try {
        BundleAdjustmentFunction* f = new BundleAdjustmentFunction;
        if (!f->load(path, 5, 30)) { delete f; return -1; }
        f->assignAll(f->getInitialState());
        HipCGDSubspaceOptimizer* ssopt = new HipCGDSubspaceOptimizer(*f);
        Options o; o.set("SSmaxit", 3);
        ssopt->setParameters(o);
        const VariablePtrVec& V = f->getVariables();
        int made = 0;
        for (int p = 0; p < 4; ++p) {
            VariablePtrVec pv; FactorPtrVec pf;
            for (int k = 0; k < 3; ++k) pv.push_back(V[(size_t)(45 + 3 * p + k)]);
            for (Factor* fa : f->getFactors()) if (static_cast<BundleAdjustmentFactor*>(fa)->getPointID() == p) pf.push_back(fa);
            NumericVec y(3); Numeric d = 0;
            for (size_t i = 0; i < 3; ++i) y[i] = pv[i]->eval();
            ssopt->optimize(pv, pf, y, d, false);
            ++made;
        }
        const int cached = (int)ssopt->planCacheEntries();
        delete f;        // first the function (and its device problem) ...
        delete ssopt;    // ... then the optimizer
        return cached == made ? cached : -3;
    } catch (const std::exception& e) { std::cerr << "harness_optimizer_outlives_function: " << e.what() << std::endl; return -2; }
}
static int call_shapes(const char* path, int nsep, int npt_calls, int maxit, int cache, double cache_bytes, double* out, double* x_out) {
    try {
        BundleAdjustmentFunction f;
        if (!f.load(path, 5, 30)) return -1;
        f.assignAll(f.getInitialState());
        HipCGDSubspaceOptimizer ssopt(f);
        Options o; o.set("SSmaxit", maxit);
        ssopt.setParameters(o);
        ssopt.setPlanCache((size_t)cache);
        if (cache_bytes >= 0) ssopt.setPlanCacheBytes((size_t)cache_bytes);
        (void)f.eval();   // upload outside the timed region
        const VariablePtrVec& V = f.getVariables();
        VariablePtrVec sepv(V.begin(), V.begin() + 48);
        FactorPtrVec sepf = f.getFactors();   // every factor reads a camera
        std::vector<VariablePtrVec> pv(30); std::vector<FactorPtrVec> pf(30);
        for (int p = 0; p < 30; ++p) for (int k = 0; k < 3; ++k) pv[(size_t)p].push_back(V[(size_t)(45 + 3 * p + k)]);
        for (Factor* fa : f.getFactors()) pf[(size_t)static_cast<BundleAdjustmentFactor*>(fa)->getPointID()].push_back(fa);
        double sum = 0; long long calls = 0; int next_pt = 1, pts_done = 0;
        const auto t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < nsep; ++r) {
            NumericVec x(48); Numeric d = 0;
            for (size_t i = 0; i < 48; ++i) x[i] = sepv[i]->eval();
            sum += ssopt.optimize(sepv, sepf, x, d, false); ++calls;
            const int quota = (int)((long long)npt_calls * (r + 1) / nsep) - pts_done;
            for (int q = 0; q < quota; ++q) {
                NumericVec y(3);
                for (size_t i = 0; i < 3; ++i) y[i] = pv[(size_t)next_pt][i]->eval();
                sum += ssopt.optimize(pv[(size_t)next_pt], pf[(size_t)next_pt], y, d, false); ++calls;
                next_pt = next_pt == 29 ? 1 : next_pt + 1;   // point 0 is part of the separator
                ++pts_done;
            }
        }
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        out[0] = ms; out[1] = (double)calls; out[2] = f.eval(); out[3] = sum;
        out[4] = (double)ssopt.planCacheHits(); out[5] = (double)ssopt.planCacheMisses();
        if (cache_bytes >= 0) { out[6] = (double)ssopt.planCacheFallbacks(); out[7] = (double)ssopt.planCacheEntries(); out[8] = (double)ssopt.planCacheBytes(); }
        if (x_out) for (size_t i = 0; i < V.size(); ++i) x_out[i] = V[i]->eval();
        return 0;
    } catch (const std::exception& e) { std::cerr << "harness_call_shapes: " << e.what() << std::endl; return -2; }
}

// ---- GPU: the level driver on the default high-dimensional sinusoid (BASELINE config 2: 121 variables in a
// ternary tree, every variable its own block): a decomposition several levels deep.  x0: start (121 values).
// out = {final, initial, sweeps, #nodes, #leaves, #split, max depth, monotone, |running sum - evaluated|, wall ms}
int harness_level_driver_sinusoid(const double* x0, int maxit, int max_sweeps, double blkpct, double seppct, int batch, double* out, double* x_out) {
    try {
        std::unique_ptr<PolynomialFunction> f = PolynomialFunction::makeHighDimSinusoid();
        NumericVec x(x0, x0 + f->getNumVars());
        f->assignAll(x);
        const Numeric before = f->eval();
        HipCGDSubspaceOptimizer ssopt(*f);
        Options o; o.set("SSmaxit", maxit);
        ssopt.setParameters(o);
        HipRDISLevelOptimizer rdis(*f, ssopt);
        Options ro; ro.set("AVblkpct", blkpct); ro.set("maxSweeps", max_sweeps); ro.set("batch", batch); ro.set("sepPiecePct", seppct);
        rdis.setParameters(ro);
        const auto t0 = std::chrono::steady_clock::now();
        const Numeric fin = rdis.optimize(false);
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        size_t nleaf = 0, nsplit = 0; int maxd = 0;
        for (const auto& nd : rdis.nodes()) { (nd.leaf ? nleaf : nsplit)++; maxd = std::max(maxd, nd.depth); }
        // the tree is a partition: every variable in exactly one leaf or one separator
        std::vector<int> seen((size_t)f->getNumVars(), 0);
        for (const auto& nd : rdis.nodes()) {
            if (nd.leaf) for (VariableID v : nd.vars) ++seen[(size_t)v];
            else for (VariableID v : nd.separator) ++seen[(size_t)v];
        }
        bool part = true;
        for (int c : seen) part = part && c == 1;
        bool mono = true; double prev = before;
        for (const auto& st : rdis.trace()) { mono = mono && st.objective <= prev; prev = st.objective; }
        out[0] = fin; out[1] = before; out[2] = rdis.sweepsDone(); out[3] = (double)rdis.nodes().size(); out[4] = (double)nleaf;
        out[5] = (double)nsplit; out[6] = maxd; out[7] = (mono && part) ? 1 : 0; out[8] = std::fabs(prev - fin); out[9] = ms;
        if (x_out) for (size_t i = 0; i < f->getVariables().size(); ++i) x_out[i] = f->getVariables()[i]->eval();
        return 0;
    } catch (const std::exception& e) { std::cerr << "harness_level_driver_sinusoid: " << e.what() << std::endl; return -2; }
}

// ---- GPU (component labelling): the level driver's decomposition, nothing solved.  which: 0 = BAL file (path, ncams, npts),
// 1 = the default high-dimensional sinusoid.  out (capacity cap int64): #nodes, then per node {depth, parent, leaf, #vars,
// #factors, #separator, #sepFactors, vars.., factors.., separator.., sepFactors..}; then #plans, per plan {depth, kind, #components,
// free_ptr.. (ncomp + 1), #free, free_vid.., fac_ptr.. (ncomp + 1), #fac, fac_id..}.  Returns the number of entries written (< 0: error).
long long harness_level_tree(int which, const char* path, long long ncams, long long npts, double blkpct, double seppct, long long* out, long long cap) {
    try {
        std::unique_ptr<OptimizableFunction> owner;
        if (which == 0) {
            std::unique_ptr<BundleAdjustmentFunction> f(new BundleAdjustmentFunction);
            if (!f->load(path, ncams, npts)) return -1;
            f->assignAll(f->getInitialState());
            owner = std::move(f);
        } else {
            std::unique_ptr<PolynomialFunction> f = PolynomialFunction::makeHighDimSinusoid();
            NumericVec x((size_t)f->getNumVars(), 0.5);
            f->assignAll(x);
            owner = std::move(f);
        }
        HipCGDSubspaceOptimizer ssopt(*owner);
        HipRDISLevelOptimizer rdis(*owner, ssopt);
        Options ro; ro.set("AVblkpct", blkpct); ro.set("sepPiecePct", seppct);
        rdis.setParameters(ro);
        rdis.decompose();
        long long n = 0;
        auto put = [&](long long v) { if (n < cap) out[n] = v; ++n; };
        put((long long)rdis.nodes().size());
        for (const auto& nd : rdis.nodes()) {
            put(nd.depth); put(nd.parent); put(nd.leaf ? 1 : 0);
            put((long long)nd.vars.size()); put((long long)nd.factors.size()); put((long long)nd.separator.size()); put((long long)nd.sepFactors.size());
            for (auto v : nd.vars) put(v);
            for (auto v : nd.factors) put(v);
            for (auto v : nd.separator) put(v);
            for (auto v : nd.sepFactors) put(v);
        }
        put((long long)rdis.numPlans());
        for (size_t i = 0; i < rdis.numPlans(); ++i) {
            int depth, kind;
            std::vector<int64_t> fp, fv, cp, ci;
            rdis.planLists(i, depth, kind, fp, fv, cp, ci);
            put(depth); put(kind); put((long long)fp.size() - 1);
            for (auto v : fp) put(v);
            put((long long)fv.size());
            for (auto v : fv) put(v);
            for (auto v : cp) put(v);
            put((long long)ci.size());
            for (auto v : ci) put(v);
        }
        return n <= cap ? n : -3;
    } catch (const std::exception& e) { std::cerr << "harness_level_tree: " << e.what() << std::endl; return -2; }
}

// ---- GPU: a function built factor by factor through the mirrored classes, one factor exponential
// (NonlinearProductFactor's constructor argument useExponential, src/NonlinearProductFactor.h:61-63).
// f0 = 2 sin((x0 - 0.5)^3) x1, f1 = 1.5 exp(-(x1^2 x2)), f2 = -0.75 x2.  out = {eval(), f1 alone, gradient refused (1/0),
// gradient of {f0, f2} wrt x0, x1, x2}
namespace {
struct SmallFunction : OptimizableFunction {
    SmallFunction() {
        VariableID id = 0;
        Variable* x0 = addVariable("x0", VariableDomain(-5, 5), id);
        Variable* x1 = addVariable("x1", VariableDomain(-5, 5), id);
        Variable* x2 = addVariable("x2", VariableDomain(-5, 5), id);
        NonlinearProductFactor* f0 = new NonlinearProductFactor(0, 2.0);
        f0->addVariable(x0, 3.0, 0.5, true); f0->addVariable(x1);
        NonlinearProductFactor* f1 = new NonlinearProductFactor(1, 1.5, /*useExponential=*/true, 2);
        f1->addVariable(x1, 2.0); f1->addVariable(x2);
        NonlinearProductFactor* f2 = new NonlinearProductFactor(2, -0.75);
        f2->addVariable(x2);
        addFactor(f0); addFactor(f1); addFactor(f2);
    }
};
}  // namespace

int harness_nlp_exponential(const double* x, double* out) {
    try {
        SmallFunction f;
        f.assignAll(NumericVec(x, x + 3));
        out[0] = f.eval();
        Numeric ferr = 0;
        out[1] = f.evalFactors(FactorPtrVec(1, f.getFactors()[1]), ferr, true);
        out[2] = 0;
        try { NumericVec g; f.computeGradient(g); } catch (const HipError&) { out[2] = 1; }
        FactorPtrVec rest; rest.push_back(f.getFactors()[0]); rest.push_back(f.getFactors()[2]);
        PartialGradient pg;
        f.computeGradient(rest, pg);
        out[3] = out[4] = out[5] = 0;
        for (const auto& kv : pg) out[3 + kv.first] = kv.second;
        return 0;
    } catch (const std::exception& e) { std::cerr << "harness_nlp_exponential: " << e.what() << std::endl; return -2; }
}

}  // extern "C"
