// rdis_levels.h -- the caller side of the path, re-thought for a batch device: what
// RDISOptimizer's recursion does around ssopt.optimize (reference
// src/RDISOptimizer.cpp:253-334 doOptimization: choose variables, value them with the subspace
// optimizer, assign, decompose, recurse into the children; :971-1117 getValueFromDomain: the
// call itself, made with every other variable of the component held at its current value --
// alternating minimisation), as a LEVEL driver:
//
//   * the decomposition is computed once (the reference's staticDecomp, :1846-1856): a tree of
//     components, each either a LEAF (optimised as a whole: at most AVblkpct of the function's
//     variables, like :342) or split by a SEPARATOR -- a set of variable blocks chosen from the
//     degree structure of the factor graph (chooseSeparator below; stands in for the PaToH call
//     site, :779-865, whose library is binary-only) -- into children: the connected components of
//     what is left (Component::createChildren, src/Component.cpp:508-549), labelled on the device
//     (rdis_hip_components), children ordered as the reference orders them (:603-608);
//   * nodes of equal depth are independent of each other, so each depth becomes at most two
//     persistent device plans (all its separator solves; all its leaves) -- siblings in ONE launch
//     (HipCGDSubspaceOptimizer::optimizeBatch semantics) instead of the reference's serial loop over
//     children (:292-314);
//   * a sweep runs the plans top-down, everything resident on the device; sweeps repeat while the
//     objective improves by more than steptol (the reference's per-node "iterative improvement"
//     steps, :1091-1108), at most maxSweeps times.
//
// Not carried over (they are what makes sibling order observable in the reference, SURVEY.md 7
// hard part 6): branch and bound, the component cache, random restarts.  With them off the
// reference's own recursion is this alternation.
#ifndef RDIS_LEVELS_H_
#define RDIS_LEVELS_H_

#include "rdis_host.h"

struct rdis_hip_plan;

namespace rdis {

class HipRDISLevelOptimizer {
public:
    struct Node {
        int depth;
        int parent;                        // index into nodes(), -1 for a top component
        bool leaf;
        std::vector<VariableID> vars;      // ascending (Component.cpp:78-79)
        std::vector<FactorID> factors;     // every factor that reads one of vars, ascending
        std::vector<VariableID> separator; // ascending; empty for a leaf
        std::vector<FactorID> sepFactors;  // the factors that read a separator variable
    };
    struct Step {                          // one launch (or, unbatched, one group of calls)
        int sweep, depth, kind;            // kind 0: separators of this depth, 1: leaves of this depth
        long long ncomp, nvars, nfactors, iters;
        double objective;                  // of the whole function after the step
        double ms;                         // wall time of the step, host side
    };

    HipRDISLevelOptimizer(OptimizableFunction& f, HipCGDSubspaceOptimizer& ssopt);
    ~HipRDISLevelOptimizer();
    HipRDISLevelOptimizer(const HipRDISLevelOptimizer&) = delete;
    HipRDISLevelOptimizer& operator=(const HipRDISLevelOptimizer&) = delete;

    // AVblkpct (0.2), steptol (1e-4), maxSweeps (20), SSmaxit / SSftol (forwarded to the subspace
    // optimizer by the caller), batch (1; 0 = one ssopt.optimize call per component, in the
    // reference's child order: the same results bit for bit, for tests and comparison), sepPiecePct (0: a
    // separator leaves pieces that are leaves; > 0: pieces of up to that fraction of the node, which are
    // split again -- deeper trees, like the reference's two-way partitions)
    void setParameters(const Options& options);

    // Every variable must be assigned (the initial state, optBA.cpp:189-205).  Returns the final
    // value of the function; the variables are left assigned to the optimum found.
    Numeric optimize(bool printInfo = false);

    // The decomposition alone -- tree and per-depth plans, nothing solved (what optimize() does first): for callers
    // that want to look at it, and for the parity tests against oracle/levels.py.
    void decompose();
    // the launches of a sweep, in order: plan i solves the separators (kind 0) or the leaves (kind 1) of one depth;
    // its lists are the arguments of rdis_hip_plan_create
    size_t numPlans() const { return plans_.size(); }
    void planLists(size_t i, int& depth, int& kind, std::vector<int64_t>& free_ptr, std::vector<int64_t>& free_vid,
                   std::vector<int64_t>& fac_ptr, std::vector<int64_t>& fac_id) const;

    const std::vector<Node>& nodes() const { return nodes_; }
    const std::vector<Step>& trace() const { return trace_; }
    int sweepsDone() const { return sweeps_; }
    double decompositionMs() const { return decomp_ms_; }

    // The separator of one component: variable blocks (OptimizableFunction::getBlockRangeByVid) are
    // put back into an empty graph in order of ascending degree (number of the component's factors
    // that read the block; ties by block id) for as long as the largest connected piece of what has
    // been put back stays within maxPiece variables; from the first block that does not fit on, blocks are
    // separator blocks.
    // Then, like ensureFactorWillBeAssigned (RDISOptimizer.cpp:412-458), the remaining variables of the
    // first factor (in list order) that reads a separator variable and has the fewest other
    // variables are added, with their blocks.  On ladybug-49-7776 with maxPiece = 0.2 x 23769: 46 of
    // the 49 cameras and one point -- the shape PaToH's cut has in the reference's own run.
    static void chooseSeparator(const OptimizableFunction& f, const std::vector<VariableID>& vars,
                                const std::vector<FactorID>& factors, size_t maxPiece,
                                std::vector<VariableID>& separator);

private:
    struct LevelPlan;
    void buildTree();
    void buildPlans();
    void releasePlans();
    double runPlan(LevelPlan& lp, int sweep, double objective, bool printInfo);
    OptimizableFunction& f_;
    HipCGDSubspaceOptimizer& ss_;
    double blkpct_, steptol_, seppct_;
    int maxSweeps_;
    bool batch_;
    size_t ssmaxit_;
    double ssftol_;
    std::vector<Node> nodes_;
    std::vector<LevelPlan*> plans_;
    std::vector<Step> trace_;
    int sweeps_;
    double decomp_ms_;
};

}  // namespace rdis
#endif  // RDIS_LEVELS_H_
