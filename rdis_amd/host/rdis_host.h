// rdis_host.h -- host side of the drop-in: the reference's plugin surface for the
// subspace-solver path, mirrored class for class, over the C ABI of
// include/rdis_hip.h.  No factor arithmetic lives here: every evaluation goes to the
// HIP kernels, and there is no CPU fallback.
//
// Mirrored interfaces (reference file:line):
//   SubspaceOptimizer                 src/SubspaceOptimizer.h:24-60
//   CGDSubspaceOptimizer              src/optimizers/CGDSubspaceOptimizer.h:21-38
//       -> HipCGDSubspaceOptimizer    (same virtual, same pre/post-conditions)
//   OptimizableFunction               src/OptimizableFunction.h:26-230 (the parts the path uses)
//   Factor / Variable / VariableDomain  src/Factor.h, src/Variable.h, src/VariableDomain.h
//   BundleAdjustmentFactor / Function src/bundleadjust/BundleAdjustmentFactor.h, ...Function.h
//   NonlinearProductFactor            src/NonlinearProductFactor.h
//   PolynomialFunction + makeHighDimSinusoid  src/PolynomialFunction.h,
//                                     src/OptimizableFunctionGenerator.cpp:660-760
// Deliberate differences:
//   * Factor gains pack(): the descriptor the device needs (the reference exposes only
//     eval / computeGradient / getVariables; SURVEY.md section 7, hard part 7).
//   * setParameters takes rdis::Options (a string->value map) because Boost
//     program_options is not available here; INTEGRATION.md shows the one-line adapter.
//   * HipCGDSubspaceOptimizer::optimizeBatch solves sibling components in one launch.
#ifndef RDIS_HOST_H_
#define RDIS_HOST_H_

#include <cstdint>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

struct rdis_hip_ctx;
struct rdis_hip_problem;
struct rdis_hip_comm;

namespace rdis {

typedef double Numeric;                 // src/common.h:25
typedef long long int VariableCount;    // src/common.h:30-32
typedef long long int VariableID;
typedef long long int FactorID;
typedef std::vector<Numeric> NumericVec;
typedef std::vector<VariableID> VariableIDVec;
// sorted (vid, value) pairs, like the reference's flat_map (src/State.h:30)
typedef std::vector<std::pair<VariableID, Numeric> > PartialGradient;

class Factor;
class OptimizableFunction;
class HipCGDSubspaceOptimizer;

// SSmaxit / SSftol carrier (stands in for boost::program_options::variables_map)
class Options {
public:
    void set(const std::string& name, double v) { vals_[name] = v; }
    size_t count(const std::string& name) const { return vals_.count(name); }
    template <class T> T as(const std::string& name) const { return static_cast<T>(vals_.at(name)); }
private:
    std::map<std::string, double> vals_;
};

// single-interval domain (CGD asserts exactly one sub-interval, CGDSubspaceOptimizer.cpp:119)
class VariableDomain {
public:
    VariableDomain() : lo_(0), hi_(0), slo_(0), shi_(0) {}
    VariableDomain(Numeric lower, Numeric upper) : lo_(lower), hi_(upper), slo_(lower), shi_(upper) {}
    // ... with the interval random states are sampled from (src/VariableDomain.h: getSamplingInterval; the bundle-adjustment
    // loader sets it per variable type, src/bundleadjust/BundleAdjustmentFunction.cpp:419-471)
    VariableDomain(Numeric lower, Numeric upper, Numeric sampleLower, Numeric sampleUpper)
        : lo_(lower), hi_(upper), slo_(sampleLower), shi_(sampleUpper) {}
    Numeric samplingMin() const { return slo_; }
    Numeric samplingMax() const { return shi_; }
    explicit VariableDomain(const std::string& domain);  // "lo:hi" (src/VariableDomain.cpp:63-72)
    Numeric min() const { return lo_; }
    Numeric max() const { return hi_; }
    Numeric closestVal(Numeric val) const {  // src/VariableDomain.cpp:158-163
        if (lo_ <= val && val <= hi_) return val;
        return val < lo_ ? lo_ : hi_;
    }
private:
    Numeric lo_, hi_, slo_, shi_;
};

class Variable {
public:
    Variable(VariableID id, const std::string& name, const VariableDomain& dom, OptimizableFunction* owner)
        : id_(id), name_(name), dom_(dom), assigned_(false), value_(0), owner_(owner) {}
    void assign(Numeric newval);            // src/Variable.cpp:66-88
    void unassign() { assigned_ = false; }
    Numeric eval() const;                   // throws if unassigned (the reference asserts)
    bool isAssigned() const { return assigned_; }
    const VariableID& getID() const { return id_; }
    const std::string& getName() const { return name_; }
    const VariableDomain& getDomain() const { return dom_; }
    void setDomain(const VariableDomain& d) { dom_ = d; }
    std::vector<Factor*>& getFactors() { return factors_; }
private:
    friend class OptimizableFunction;
    friend class HipCGDSubspaceOptimizer;
    friend class HipLMSubspaceOptimizer;
    VariableID id_;
    std::string name_;
    VariableDomain dom_;
    bool assigned_;
    Numeric value_;
    OptimizableFunction* owner_;
    std::vector<Factor*> factors_;
};
typedef std::vector<Variable*> VariablePtrVec;

class Factor {
public:
    explicit Factor(FactorID id) : id_(id), owner_(nullptr) {}
    virtual ~Factor() {}
    FactorID getID() const { return id_; }
    const VariablePtrVec& getVariables() const { return variables_; }
    bool areAllVarsAssigned() const;
    // Factor::eval / computeGradient (src/Factor.h:72, :92-93), evaluated on the device
    Numeric eval() const;
    void computeGradient(PartialGradient& g) const;
protected:
    friend class OptimizableFunction;
    FactorID id_;
    VariablePtrVec variables_;
    OptimizableFunction* owner_;
};
typedef std::vector<Factor*> FactorPtrVec;

// vals = [rx ry rz tx ty tz f k1 k2 X Y Z] (BundleAdjustmentCommon.h:36-59)
class BundleAdjustmentFactor : public Factor {
public:
    BundleAdjustmentFactor(FactorID id, long long cameraID, long long pointID, Numeric obsX, Numeric obsY)
        : Factor(id), cam_(cameraID), pt_(pointID), ox_(obsX), oy_(obsY) {}
    void addVariable(Variable* v) { variables_.push_back(v); v->getFactors().push_back(this); }
    long long getCameraID() const { return cam_; }
    long long getPointID() const { return pt_; }
    void getObservation(Numeric& x, Numeric& y) const { x = ox_; y = oy_; }
private:
    long long cam_, pt_;
    Numeric ox_, oy_;
};

// coeff * prod_i g((x_i - k_i)^e_i), g = id or sin (src/NonlinearProductFactor.h:15-20)
class NonlinearProductFactor : public Factor {
public:
    struct Term { Numeric exponent, constant; bool useSine; };
    // (the reference's signature, src/NonlinearProductFactor.h:61-63; totalNVars only reserves)
    explicit NonlinearProductFactor(FactorID id, Numeric coefficient = 1, bool useExponential = false, VariableCount totalNVars = 0)
        : Factor(id), coeff_(coefficient), useExponential_(useExponential) { terms_.reserve((size_t)totalNVars); }
    // exponent 0 is dropped and a repeated variable ignored (src/NonlinearProductFactor.cpp:27-52)
    void addVariable(Variable* v, Numeric exponent = 1, Numeric constant = 0, bool useSine = false);
    void setCoeff(Numeric c) { coeff_ = c; }
    Numeric getCoeff() const { return coeff_; }
    // value = coeff * exp(-product) (src/NonlinearProductFactor.cpp:140); such a factor has no gradient in the reference
    // (computeGradient asserts, :110) and the device library refuses to form one
    bool usesExponential() const { return useExponential_; }
    const std::vector<Term>& terms() const { return terms_; }
private:
    Numeric coeff_;
    bool useExponential_;
    std::vector<Term> terms_;
};

class OptimizableFunction {
public:
    OptimizableFunction();
    virtual ~OptimizableFunction();
    OptimizableFunction(const OptimizableFunction&) = delete;
    OptimizableFunction& operator=(const OptimizableFunction&) = delete;

    virtual VariableCount getNumVars() const { return (VariableCount)variables.size(); }
    virtual VariablePtrVec& getVariables() { return variables; }
    virtual const FactorPtrVec& getFactors() const { return factors; }
    // hooks the solver calls after (un)assigning a variable (src/OptimizableFunction.h:65-69)
    virtual void onVarAssigned(const VariableID, const Numeric) const {}
    virtual void onVarUnassigned(const VariableID) const {}
    // MinSum semiring: product is +, sum is min => descent (src/Semiring.h:97-106)
    bool isMinSum() const { return true; }

    // src/OptimizableFunction.h:82-88, :109-110 -- evaluated by the HIP kernels
    virtual Numeric eval() const;
    virtual Numeric evalFactors(const FactorPtrVec& fctrs, Numeric& ferr, bool useCached = true) const;
    virtual void computeGradient(const FactorPtrVec& facs, PartialGradient& gradient, bool checkGrad = false) const;
    virtual void computeGradient(NumericVec& gradient, bool checkGrad = false) const;

    // variable blocks (src/OptimizableFunction.h:157-179): the variables an outer optimiser assigns
    // together.  Default: every variable is its own block.
    virtual bool hasBlockedVars() const { return false; }
    virtual void getBlockRangeByVid(VariableID vid, VariableID& lower, VariableID& upper) const { lower = upper = vid; }

    // assign every variable (State x) -- optBA's "assign the initial state"
    void assignAll(const NumericVec& x);
    // the device has assigned these values itself (a solve leaves its variables assigned there):
    // mirror them on the host side without marking them for upload
    void adoptDeviceValues(const std::vector<int64_t>& vids, const std::vector<double>& vals);
    void initDevice(int device = 0);          // upload the packed function (done lazily otherwise)
    rdis_hip_problem* deviceProblem() const;  // uploads + pushes pending assignments
    rdis_hip_ctx* deviceContext() const;
    // Several GPUs of one node from ONE process: the function is replicated, one context + device problem per listed
    // device (devices[0] is the primary, which serves every call that is not a batch); a batch of independent
    // components -- sibling components of a recursion level, src/Component.cpp:508-549 -- is shared out over them
    // by HipCGDSubspaceOptimizer::optimizeBatch.  The same device may be listed more than once (two contexts on one
    // GPU: what the tests do).  To be called before the first use of the device; replicas are uploaded when first used.
    void setDevices(const std::vector<int>& devices);
    size_t numDevices() const { return 1 + replicas_.size(); }
    int deviceOrdinal(size_t d) const { return d == 0 ? device_ : replicas_.at(d - 1).device; }   // the GPU behind entry d
    rdis_hip_problem* deviceProblem(size_t d) const;   // d = 0: the primary
    rdis_hip_ctx* deviceContext(size_t d) const;
    // One RCCL communicator per listed device (rdis_hip_comm_create_all), made at first use: what optimizeBatch's objective
    // all-reduce runs on.  nullptr when the list names a GPU twice or holds one device (RCCL wants one rank per GPU) or RCCL
    // is not there: the partial sums then meet on the host (rdis_hip_allreduce_objective_all without communicators).
    rdis_hip_comm* const* deviceComms() const;

    // packed (structure-of-arrays) view of the function; what rdis_hip_upload_* takes
    struct Packed {
        int kind;  // 0 = bundle adjustment, 1 = nonlinear product
        std::vector<double> lo, hi;
        std::vector<int64_t> cam_vid0, pt_vid0;
        std::vector<double> obs;
        std::vector<double> coeff, expo, cons;
        std::vector<int64_t> rowptr, vid;
        std::vector<uint8_t> sine;
        std::vector<uint8_t> useexp;  // empty, or one flag per factor (NonlinearProductFactor::usesExponential)
    };
    const Packed& packed() const;

protected:
    Variable* addVariable(const std::string& name, const VariableDomain& dom, VariableID& id);
    void addFactor(Factor* f);
    VariablePtrVec variables;  // indexed by id (creation order, src/OptimizableFunction.cpp:57-76)
    FactorPtrVec factors;      // indexed by id
    VariableDomain defaultDomain;

private:
    friend class Variable;
    friend class Factor;
    friend class HipCGDSubspaceOptimizer;
    friend class HipLMSubspaceOptimizer;
    void markDirty(VariableID id) const;
    void markDirtyElsewhere(VariableID id, size_t holder) const;   // device `holder` has assigned the value itself
    void pushAssignments() const;
    void ensureUploaded() const;
    struct Replica {
        int device = 0;
        rdis_hip_ctx* ctx = nullptr;
        rdis_hip_problem* prob = nullptr;
        std::vector<VariableID> dirty;
        std::vector<char> is_dirty;
    };
    void uploadTo(int device, rdis_hip_ctx*& ctx, rdis_hip_problem*& prob) const;
    mutable std::vector<Replica> replicas_;   // devices 1 .. (setDevices)
    mutable std::vector<rdis_hip_comm*> comms_;   // one per device once made (deviceComms)
    mutable bool comms_tried_ = false;
    void fillIds(const FactorPtrVec& f, std::vector<int64_t>& ids) const;
    mutable std::unique_ptr<Packed> packed_;
    mutable rdis_hip_ctx* ctx_;
    mutable rdis_hip_problem* prob_;
    mutable int device_;
    mutable std::vector<VariableID> dirty_;
    mutable std::vector<char> is_dirty_;
    std::unordered_map<std::string, Variable*> by_name_;
    mutable std::vector<HipCGDSubspaceOptimizer*> plan_holders_;   // optimisers with cached plans of prob_
};

class BundleAdjustmentFunction : public OptimizableFunction {
public:
    BundleAdjustmentFunction() : ncams_(0), npts_(0) {}
    // BAL text (src/bundleadjust/BundleAdjustmentFunction.cpp:50-250); numcams / numpoints > 0
    // keep the first k cameras / points
    bool load(const std::string& file, VariableCount numcams = 0, VariableCount numpoints = 0);
    // BAL file with the currently assigned values (or the given state) in place of the parameters
    // (src/bundleadjust/BundleAdjustmentFunction.cpp:253-320)
    bool save(const std::string& file, const NumericVec* state = nullptr) const;
    long long getNumCameras() const { return ncams_; }
    long long getNumPoints() const { return npts_; }
    VariableID getCamVID(long long cid, unsigned k) const { return cid * 9 + k; }                 // .h:88-96
    VariableID getPointVID(long long pid, unsigned k) const { return ncams_ * 9 + pid * 3 + k; }
    // blocks = camera (9) / point (3) (.h:51-78)
    VariableCount getNumBlocks() const { return ncams_ + npts_; }
    void getBlockRangeByBlkId(VariableCount b, VariableID& lo, VariableID& hi) const;
    virtual bool hasBlockedVars() const { return true; }
    virtual void getBlockRangeByVid(VariableID vid, VariableID& lower, VariableID& upper) const {
        if (vid < ncams_ * 9) { lower = vid / 9 * 9; upper = lower + 8; }
        else { lower = ncams_ * 9 + (vid - ncams_ * 9) / 3 * 3; upper = lower + 2; }
    }
    const NumericVec& getInitialState() const { return xinit; }
private:
    void setDomain(VariableID vid, Numeric initialVal);  // .cpp:402-477
    long long ncams_, npts_;
    NumericVec xinit;
};

class PolynomialFunction : public OptimizableFunction {
public:
    PolynomialFunction() {}
    explicit PolynomialFunction(const VariableDomain& dflt) { defaultDomain = dflt; }
    bool load(const std::string& file);  // src/PolynomialFunction.cpp:60-215
    // OptimizableFunctionGenerator::makeHighDimSinusoid (src/OptimizableFunctionGenerator.cpp:660-760)
    static std::unique_ptr<PolynomialFunction> makeHighDimSinusoid(VariableCount treeHeight = 4,
            VariableCount branches = 3, VariableCount maxArity = 3, bool allowOddArityFactors = false);
private:
    Variable* varByName(const std::string& name, VariableID& next_id);
};

// src/SubspaceOptimizer.h:24-60
class SubspaceOptimizer {
public:
    explicit SubspaceOptimizer(OptimizableFunction& f_);
    virtual ~SubspaceOptimizer() {}
    virtual void setParameters(const Options& options);  // SSmaxit, SSftol (src/SubspaceOptimizer.cpp:26-32)
    size_t getMaxIters() const { return maxiters; }
    Numeric getFtol() const { return ftol; }
    // vars / factors define the sub-function; xval: start values in the order of vars, overwritten
    // with the final (clamped) values; deltaFval = f(x_end) - f(x_init); returns f(x_end)
    virtual Numeric optimize(const VariablePtrVec& vars, const FactorPtrVec& factors, NumericVec& xval,
                             Numeric& deltaFval, const bool printdbg) = 0;
protected:
    OptimizableFunction& f;
    const bool doAscent;
    size_t maxiters;
    Numeric ftol;
};

// The drop-in for CGDSubspaceOptimizer: same contract, solved on the MI355X.
class HipCGDSubspaceOptimizer : public SubspaceOptimizer {
public:
    explicit HipCGDSubspaceOptimizer(OptimizableFunction& f_);
    virtual ~HipCGDSubspaceOptimizer();
    virtual Numeric optimize(const VariablePtrVec& vars, const FactorPtrVec& factors, NumericVec& xinit,
                             Numeric& deltaFval, const bool printdbg);

    // sibling components of one recursion level in one launch (they share no free variable
    // and no factor, src/Component.cpp:508-549).  Returns the sum of the components' values.
    struct Component {
        VariablePtrVec vars;
        FactorPtrVec factors;
        NumericVec xval;      // in: start, out: final clamped values
        Numeric fret, deltaFval;
        int iters, status;
        long long nfeval, ngeval;
    };
    Numeric optimizeBatch(std::vector<Component>& comps, const bool printdbg);
    // the batch's objective as the devices' all-reduce left it (rdis_hip_allreduce_objective_all: RCCL over xGMI between the
    // listed GPUs, the host where there is no communicator) -- what a caller that keeps its state on the devices reads; the
    // value optimizeBatch RETURNS is the host sum of the components' values in device order, the same bits whatever the sharing
    Numeric lastBatchObjective() const { return last_batch_objective_; }
    bool lastBatchObjectiveOverRccl() const { return last_batch_rccl_; }

    // The sibling components themselves: connected components of the factor graph over the
    // variables that are currently unassigned -- what Component::createChildren
    // (src/Component.cpp:508-549) obtains from the reference's dynamic connectivity structure,
    // labelled on the device (rdis_hip_components).  Variable and factor lists ascending by id,
    // components by number of variables ascending (Component.cpp:60-79, 603-608); xval of each
    // is left empty for the caller to fill with a start.
    std::vector<Component> createChildren();

    // Callers like RDISOptimizer come back with the same (variables, factors) lists over and over
    // (every revisit of a component, every restart).  The decomposition of such a call -- validated
    // lists, gather tables, workspace -- is kept on the device as a persistent plan, keyed by the id
    // lists, so a repeat pays for the start values, the launch and the results only.  At most
    // `entries` plans are kept (least recently used goes first); 0 switches the cache off.  Results
    // are bit-identical either way.
    void setPlanCache(size_t entries);
    // ... and at most `bytes` of device memory in all (default 4 GiB; rdis_hip_plan_device_bytes).  When
    // the device runs out of memory while a plan is created, least recently used plans are dropped and the
    // creation retried; a call whose plan cannot be kept at all is served by the transient path
    // (rdis_hip_cgd_batch), counted in planCacheFallbacks().
    void setPlanCacheBytes(size_t bytes);
    // a plan option of include/rdis_hip.h (rdis_hip_plan_set_option) for every plan this optimiser makes from now on -- e.g. the
    // parity option: setPlanOption("factor_rounding", 1); setPlanOption("emulate_stale_cache", 1).  Plans already
    // cached are dropped; calls too large to cache (the transient path) take the library's defaults.
    void setPlanOption(const std::string& name, long long value);
    size_t planCacheHits() const { return cache_hits_; }
    size_t planCacheMisses() const { return cache_misses_; }
    size_t planCacheFallbacks() const { return cache_fallbacks_; }
    size_t planCacheBytes() const { return cache_bytes_; }
    size_t planCacheEntries() const { return cache_.size(); }

    // results of the last optimize() beyond what the reference returns
    int lastIters() const { return last_iters_; }
    int lastStatus() const { return last_status_; }
    long long lastFEvals() const { return last_nfeval_; }
    long long lastGEvals() const { return last_ngeval_; }
private:
    struct CachedPlan;
    CachedPlan* cachedPlan(const std::vector<int64_t>& free_ptr, const std::vector<int64_t>& free_vid,
                           const std::vector<int64_t>& fac_ptr, const std::vector<int64_t>& fac_id, size_t dev = 0);
    struct Shard;
    void solveShards(std::vector<Shard>& shards);
    void dropPlans();
    bool evictOne(const CachedPlan* keep = nullptr, int gpu = -1);   // the least recently used plan other than `keep` goes (gpu >= 0: on that GPU)
    void forget(CachedPlan* e);
    friend class OptimizableFunction;
    void functionGone();   // the function is being destroyed: its device problem goes, and the plans with it
    int last_iters_, last_status_;
    long long last_nfeval_, last_ngeval_;
    std::vector<CachedPlan*> cache_;
    size_t cache_cap_, cache_hits_, cache_misses_;
    unsigned long long cache_tick_;
    size_t cache_byte_cap_, cache_bytes_, cache_fallbacks_;
    bool function_alive_;
    std::vector<unsigned> free_stamp_;   // optimizeBatch: variable id -> stamp of the component it is free in
    std::vector<std::pair<std::string, long long> > plan_options_;
    Numeric last_batch_objective_ = 0;
    bool last_batch_rccl_ = false, last_batch_valid_ = false;
    unsigned stamp_;
};

// The drop-in for LMSubspaceOptimizer (src/optimizers/LMSubspaceOptimizer.h): same contract and
// the same least-squares problem (one residual sqrt(2 E_j) per factor, levmar's options as set at
// LMSubspaceOptimizer.cpp:84-101), solved on the MI355X by rdis_hip_lm_optimize -- block normal
// equations, Schur complement onto the cameras, matrix-core contractions.  Bundle adjustment
// functions only.  levmar is not part of the reference tree: parity is unpinned (oracle/lm_oracle.py).
class HipLMSubspaceOptimizer : public SubspaceOptimizer {
public:
    explicit HipLMSubspaceOptimizer(OptimizableFunction& f_);
    virtual ~HipLMSubspaceOptimizer() {}
    virtual Numeric optimize(const VariablePtrVec& vars, const FactorPtrVec& factors, NumericVec& xinit,
                             Numeric& deltaFval, const bool printdbg);
    int lastIters() const { return last_iters_; }
    int lastStop() const { return last_stop_; }          // levmar's termination code (info[6])
    int lastLinearSolves() const { return last_nsolve_; }
    // 1 (default): the reference's residual sqrt(2 E_j) per factor; 2: the two pixel residuals
    void setResidualModel(int m) { model_ = m; }
private:
    int model_, last_iters_, last_stop_, last_nsolve_;
};

// error raised when the HIP library reports a failure (no silent fallback)
class HipError : public std::runtime_error {
public:
    HipError(int code, const std::string& what) : std::runtime_error(what), code_(code) {}
    int code() const { return code_; }
private:
    int code_;
};

}  // namespace rdis
#endif  // RDIS_HOST_H_
