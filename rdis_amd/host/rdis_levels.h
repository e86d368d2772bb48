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
// hard part 6): branch and bound, the component cache.  With them and the random restarts off the
// reference's own recursion is this alternation (optimize()).  optimizeReferenceSchedule() (round 4) runs the
// tree under the reference's own per-node schedule instead -- iterative improvement node by node and random
// restarts, with restart values drawn per (node, restart, variable) so that siblings stay independent.
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

    // The same tree under the REFERENCE'S schedule (round 4): what doOptimization / getValueFromDomain do at every node
    // (src/RDISOptimizer.cpp:253-334, 971-1147, 1507-1577) with branch and bound and the component cache off --
    //   * a node is valued again and again: first from its initial values (step type "initial values"), then, with its
    //     children re-optimised in between, from where it stands ("iterative improvement", :1131-1133) for as long as
    //     the subspace optimizer makes progress beyond steptol (:1086-1101); its children run their own such loops to
    //     their end before the node is valued again (the recursion of :292-314);
    //   * when a step makes no progress the node is restarted from a random state (:1088-1094, sampleRandomState
    //     :1196-1216: uniform over each variable's sampling interval), at most numRestarts = max(minRR, nRRperLvl >>
    //     depth) times (nRRatTop at depth 0; :978-983), and after maxNAtoRR assignments without one a restart is forced
    //     below the top (:992-994); the children of a node restarted after NO PROGRESS start from random states too
    //     (:1162-1171), those of a node whose FORCED restart made progress start from the values they had in the
    //     node's previous evaluation (setInitialValFromChildren, :1112-1114, 1713-1724);
    //   * every evaluation of a node is compared with its best so far (updateDomain :1507-1577: a new minimum only beyond
    //     steptol), a node is done when its last evaluation was no new minimum and its restarts are spent (:997-999),
    //     and it is left at its best.
    // What is NOT the reference's: the tree (chooseSeparator in place of PaToH), the random numbers -- a restart's
    // values are drawn per (node, restart, variable) from splitmix64(restartSeed ...), so siblings can be solved in any
    // order or together (the reference draws from one shared mt19937 in visiting order, :1139, 1223-1225) -- and that
    // every variable keeps a value throughout (the reference's alwaysUseAllFactors, :1029-1030, 1165-1182).  Independent
    // nodes run in lock-step: all first evaluations of a set of siblings in one optimizeBatch, all their children
    // together, and so on.  Options: nRRperLvl (2), nRRatTop (nRRperLvl), minRR (1), maxNAtoRR (10),
    // noAssignLimitAtTop (1), restartSeed, maxCalls (100000 subspace-optimizer calls: the reference's time limit, :317-321 --
    // when it is reached every node's loop ends where it stands, at its best).  End-to-end parity with the reference stays unpinned (PaToH, Boost's
    // generator); what is pinned is the schedule: oracle/levels.py restates it from the reference's lines and must take
    // the same decisions, and draw the same restart values, bit for bit, from the values this run's solves returned.
    struct RefStep {
        int node, kind;                    // kind 0: initial values, 1: iterative improvement, 2: random restart
        int nrr, va;                       // the node's restart count and assignments since the last one, after the step
        double fret, delta;                // what the subspace optimizer returned
        double value;                      // the node's value after its children ran (NaN: the step made no progress)
        int newMin;                        // updateDomain's verdict
        unsigned long long startHash;      // FNV-1a over the bits of the start vector
    };
    Numeric optimizeReferenceSchedule(bool printInfo = false);
    const std::vector<RefStep>& refTrace() const { return ref_trace_; }
    // of the last optimizeReferenceSchedule: subspace-optimizer calls (ssopt.optimize, src/RDISOptimizer.cpp:1067), their CG
    // iterations (Frprmn outer iterations) and objective evaluations, and the optimizeBatch launches that carried them
    long long refCalls() const { return ref_calls_; }
    long long refIterations() const { return ref_iters_; }
    long long refFEvals() const { return ref_fevals_; }
    long long refBatches() const { return ref_batches_; }
    static double restartValue(unsigned long long seed, int node, int restart, VariableID vid, const VariableDomain& dom);

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
    // reference schedule
    struct NodeState;
    void runSet(const std::vector<int>& set, const std::vector<char>& randomInit, std::vector<NodeState>& st, bool printInfo);
    unsigned nrr_per_lvl_, nrr_at_top_, min_rr_, max_na_to_rr_;
    bool no_assign_limit_at_top_, nrr_at_top_set_;
    unsigned long long restart_seed_;
    long long ref_iters_ = 0, ref_fevals_ = 0, ref_batches_ = 0;
    long long max_calls_, ref_calls_;   // budget of subspace-optimizer calls (the reference: a time limit, :317-321), calls so far
    std::vector<std::vector<int> > children_;
    std::vector<RefStep> ref_trace_;
};

}  // namespace rdis
#endif  // RDIS_LEVELS_H_
