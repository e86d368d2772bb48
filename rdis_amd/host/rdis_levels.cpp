// rdis_levels.cpp -- see rdis_levels.h.  Host bookkeeping of the decomposition tree; the component
// labelling and every solve go through include/rdis_hip.h.
#include "rdis_levels.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <iostream>
#include <numeric>

#include "../../include/rdis_hip.h"

namespace rdis {

namespace {
void check(rdis_hip_ctx* ctx, int rc, const char* where) {
    if (rc != 0) throw HipError(rc, std::string(where) + ": " + (ctx ? rdis_hip_last_error(ctx) : "no context"));
}
double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
struct UnionFind {
    std::vector<int> parent, size;
    explicit UnionFind(size_t n) : parent(n), size(n, 0) { std::iota(parent.begin(), parent.end(), 0); }
    int find(int x) { while (parent[(size_t)x] != x) { parent[(size_t)x] = parent[(size_t)parent[(size_t)x]]; x = parent[(size_t)x]; } return x; }
    int unite(int a, int b) {
        a = find(a); b = find(b);
        if (a == b) return a;
        if (size[(size_t)a] < size[(size_t)b]) std::swap(a, b);
        parent[(size_t)b] = a; size[(size_t)a] += size[(size_t)b];
        return a;
    }
};
}  // namespace

struct HipRDISLevelOptimizer::LevelPlan {
    int depth = 0, kind = 0;
    std::vector<int> node;                  // node index of each component
    std::vector<int64_t> free_ptr, free_vid, fac_ptr, fac_id;
    rdis_hip_plan* plan = nullptr;
    std::vector<double> x, fret, delta;
    std::vector<int32_t> iters, status;
};

HipRDISLevelOptimizer::HipRDISLevelOptimizer(OptimizableFunction& f, HipCGDSubspaceOptimizer& ssopt)
    : f_(f), ss_(ssopt), blkpct_(0.2), steptol_(1.0e-4), seppct_(0.0), maxSweeps_(20), batch_(true),
      ssmaxit_(ssopt.getMaxIters()), ssftol_(ssopt.getFtol()), sweeps_(0), decomp_ms_(0),
      nrr_per_lvl_(2), nrr_at_top_(2), min_rr_(1), max_na_to_rr_(10), no_assign_limit_at_top_(true), nrr_at_top_set_(false),
      restart_seed_(0x5D15ull), max_calls_(100000), ref_calls_(0) {}

HipRDISLevelOptimizer::~HipRDISLevelOptimizer() { releasePlans(); }

void HipRDISLevelOptimizer::releasePlans() {
    for (LevelPlan* lp : plans_) {
        if (lp->plan) rdis_hip_plan_destroy(lp->plan);
        delete lp;
    }
    plans_.clear();
}

void HipRDISLevelOptimizer::setParameters(const Options& o) {
    if (o.count("AVblkpct")) blkpct_ = o.as<double>("AVblkpct");     // RDISOptimizer.cpp:95
    if (o.count("steptol")) steptol_ = o.as<double>("steptol");      // :1089-1090
    if (o.count("maxSweeps")) maxSweeps_ = o.as<int>("maxSweeps");
    if (o.count("batch")) batch_ = o.as<int>("batch") != 0;
    if (o.count("sepPiecePct")) seppct_ = o.as<double>("sepPiecePct");
    if (o.count("nRRperLvl")) nrr_per_lvl_ = (unsigned)o.as<int>("nRRperLvl");          // RDISOptimizer.cpp:97, 1781-1782
    if (o.count("nRRatTop")) { nrr_at_top_ = (unsigned)o.as<int>("nRRatTop"); nrr_at_top_set_ = true; }   // :99, 1784-1785
    if (!nrr_at_top_set_) nrr_at_top_ = nrr_per_lvl_;
    if (o.count("minRR")) min_rr_ = (unsigned)o.as<int>("minRR");                       // :102, 1778-1779
    if (o.count("maxNAtoRR")) max_na_to_rr_ = (unsigned)o.as<int>("maxNAtoRR");         // :104, 1787-1788
    if (o.count("noAssignLimitAtTop")) no_assign_limit_at_top_ = o.as<int>("noAssignLimitAtTop") != 0;   // :115, 1793-1794
    if (o.count("restartSeed")) restart_seed_ = (unsigned long long)o.as<double>("restartSeed");
    if (o.count("maxCalls")) max_calls_ = (long long)o.as<double>("maxCalls");
    if (blkpct_ <= 0 || blkpct_ > 1 || maxSweeps_ < 1) throw std::invalid_argument("HipRDISLevelOptimizer: bad options");
}

void HipRDISLevelOptimizer::chooseSeparator(const OptimizableFunction& f, const std::vector<VariableID>& vars,
                                            const std::vector<FactorID>& factors, size_t maxPiece,
                                            std::vector<VariableID>& separator) {
    separator.clear();
    const FactorPtrVec& allf = f.getFactors();
    // blocks of the component: local block index per variable
    std::vector<VariableID> blk_lo;                 // first variable id of each block, ascending
    std::vector<int> blk_nv;                        // variables of the block that are in the component
    std::vector<std::pair<VariableID, int> > v2b;   // (vid, block) sorted by vid
    v2b.reserve(vars.size());
    for (VariableID v : vars) {
        VariableID lo = v, hi = v;
        f.getBlockRangeByVid(v, lo, hi);
        if (blk_lo.empty() || blk_lo.back() != lo) { blk_lo.push_back(lo); blk_nv.push_back(0); }
        ++blk_nv.back();
        v2b.push_back(std::make_pair(v, (int)blk_lo.size() - 1));
    }
    const size_t nb = blk_lo.size();
    auto block_of = [&](VariableID v) -> int {
        auto it = std::lower_bound(v2b.begin(), v2b.end(), std::make_pair(v, -1));
        return (it != v2b.end() && it->first == v) ? it->second : -1;
    };
    // per factor the distinct component blocks it reads; block degree
    std::vector<std::vector<int> > fblk(factors.size());
    std::vector<long long> deg(nb, 0);
    for (size_t j = 0; j < factors.size(); ++j) {
        for (const Variable* v : allf[(size_t)factors[j]]->getVariables()) {
            const int b = block_of(v->getID());
            if (b >= 0 && std::find(fblk[j].begin(), fblk[j].end(), b) == fblk[j].end()) fblk[j].push_back(b);
        }
        for (int b : fblk[j]) ++deg[(size_t)b];
    }
    std::vector<int> order(nb);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return deg[(size_t)a] < deg[(size_t)b]; });
    // factors by block, to connect a block to the blocks already put back
    std::vector<std::vector<int> > bfac(nb);
    for (size_t j = 0; j < factors.size(); ++j) for (int b : fblk[j]) bfac[(size_t)b].push_back((int)j);
    UnionFind uf(nb);
    std::vector<char> in(nb, 0);
    std::vector<int> seen_stamp(nb, -1);
    bool closed = false;   // once a block does not fit, every block of higher degree is a separator block too
    for (size_t k = 0; k < nb; ++k) {
        const int b = order[k];
        if (!closed) {
            // size of the piece this block would create
            long long total = blk_nv[(size_t)b];
            for (int j : bfac[(size_t)b])
                for (int o : fblk[(size_t)j]) {
                    if (o == b || !in[(size_t)o]) continue;
                    const int r = uf.find(o);
                    if (seen_stamp[(size_t)r] == (int)k) continue;
                    seen_stamp[(size_t)r] = (int)k;
                    total += uf.size[(size_t)r];
                }
            if ((size_t)total <= maxPiece) {
                in[(size_t)b] = 1;
                uf.size[(size_t)uf.find(b)] = blk_nv[(size_t)b];
                for (int j : bfac[(size_t)b])
                    for (int o : fblk[(size_t)j])
                        if (o != b && in[(size_t)o]) uf.unite(b, o);
                continue;
            }
            closed = true;
        }
    }
    std::vector<char> in_sep(nb, 0);
    bool any = false;
    for (size_t b = 0; b < nb; ++b) if (!in[b]) { in_sep[b] = 1; any = true; }
    if (any) {
        // ensureFactorWillBeAssigned: the first factor with the fewest variables outside the separator
        long long best = -1; size_t bestj = 0;
        for (size_t j = 0; j < factors.size(); ++j) {
            bool touches = false; long long left = 0;
            for (int b : fblk[j]) { if (in_sep[(size_t)b]) touches = true; else left += blk_nv[(size_t)b]; }
            if (!touches || left == 0) { if (touches && left == 0) { best = 0; break; } continue; }
            if (best < 0 || left < best) { best = left; bestj = j; }
        }
        if (best > 0) for (int b : fblk[bestj]) in_sep[(size_t)b] = 1;
    }
    for (const auto& vb : v2b) if (in_sep[(size_t)vb.second]) separator.push_back(vb.first);
}

void HipRDISLevelOptimizer::buildTree() {
    const double t0 = now_ms();
    nodes_.clear();
    const VariablePtrVec& vars = f_.getVariables();
    const FactorPtrVec& facs = f_.getFactors();
    const size_t N = vars.size();
    const size_t leafMax = std::max<size_t>(1, (size_t)std::llround(blkpct_ * (double)N));   // RDISOptimizer.cpp:342, :1761
    rdis_hip_problem* p = f_.deviceProblem();
    rdis_hip_ctx* ctx = f_.deviceContext();

    // components of a set of free variables (everything else assigned), on the device
    auto components = [&](const std::vector<uint8_t>& assigned, std::vector<int64_t>& free_ptr, std::vector<int64_t>& free_vid,
                          std::vector<int64_t>& fac_ptr, std::vector<int64_t>& fac_id) {
        int64_t nc = 0, nfree = 0, nfac = 0;
        check(ctx, rdis_hip_components(p, assigned.data(), &nc, &nfree, &nfac), "rdis_hip_components");
        free_ptr.assign((size_t)nc + 1, 0); fac_ptr.assign((size_t)nc + 1, 0);
        free_vid.assign((size_t)nfree, 0); fac_id.assign((size_t)nfac, 0);
        check(ctx, rdis_hip_components_fetch(p, free_ptr.data(), free_vid.data(), fac_ptr.data(), fac_id.data()), "rdis_hip_components_fetch");
        return (size_t)nc;
    };

    // depth 0: the connected components of the whole function (topccomp->decompose(), RDISOptimizer.cpp:173)
    std::vector<uint8_t> assigned(N, 0);
    std::vector<int64_t> fp, fv, cp, ci;
    size_t nc = components(assigned, fp, fv, cp, ci);
    std::vector<int> frontier;
    for (size_t c = 0; c < nc; ++c) {
        Node nd; nd.depth = 0; nd.parent = -1; nd.leaf = true;
        nd.vars.assign(fv.begin() + fp[c], fv.begin() + fp[c + 1]);
        nd.factors.assign(ci.begin() + cp[c], ci.begin() + cp[c + 1]);
        nodes_.push_back(nd);
        frontier.push_back((int)nodes_.size() - 1);
    }
    std::vector<int> owner(N, -1);
    for (int depth = 0; !frontier.empty(); ++depth) {
        // split what is too large to be optimised as a whole
        std::fill(assigned.begin(), assigned.end(), 1);
        std::vector<int> split;
        for (int ni : frontier) {
            Node& nd = nodes_[(size_t)ni];
            if (nd.factors.empty() || nd.vars.size() <= leafMax) continue;
            // (sepPiecePct > 0: pieces of up to that fraction of the NODE -- a bisection-like cut whose children
            // may be split again, the shape of the reference's two-way partitions; default: pieces that are leaves)
            const size_t piece = seppct_ > 0 ? std::max<size_t>(leafMax, (size_t)std::llround(seppct_ * (double)nd.vars.size())) : leafMax;
            chooseSeparator(f_, nd.vars, nd.factors, piece, nd.separator);
            if (nd.separator.empty() || nd.separator.size() >= nd.vars.size()) { nd.separator.clear(); continue; }
            nd.leaf = false;
            std::vector<char> is_sep(N, 0);   // (only this node's entries are read)
            for (VariableID v : nd.separator) is_sep[(size_t)v] = 1;
            for (FactorID fid : nd.factors) {
                for (const Variable* v : facs[(size_t)fid]->getVariables())
                    if (is_sep[(size_t)v->getID()]) { nd.sepFactors.push_back(fid); break; }
            }
            for (VariableID v : nd.vars) if (!is_sep[(size_t)v]) { assigned[(size_t)v] = 0; owner[(size_t)v] = ni; }
            split.push_back(ni);
        }
        frontier.clear();
        if (split.empty()) break;
        // the children of ALL nodes split at this depth: one labelling call (the nodes are disjoint)
        nc = components(assigned, fp, fv, cp, ci);
        for (size_t c = 0; c < nc; ++c) {
            Node ch; ch.depth = depth + 1; ch.leaf = true;
            ch.vars.assign(fv.begin() + fp[c], fv.begin() + fp[c + 1]);
            ch.factors.assign(ci.begin() + cp[c], ci.begin() + cp[c + 1]);
            ch.parent = owner[(size_t)ch.vars.front()];
            nodes_.push_back(ch);
            frontier.push_back((int)nodes_.size() - 1);
        }
    }
    decomp_ms_ = now_ms() - t0;
}

void HipRDISLevelOptimizer::buildPlans() {
    releasePlans();
    int maxDepth = 0;
    for (const Node& nd : nodes_) maxDepth = std::max(maxDepth, nd.depth);
    rdis_hip_problem* p = f_.deviceProblem();
    for (int d = 0; d <= maxDepth; ++d)
        for (int kind = 0; kind < 2; ++kind) {   // a node's separator first, then (one level down) its children
            LevelPlan* lp = new LevelPlan;
            lp->depth = d; lp->kind = kind;
            lp->free_ptr.push_back(0); lp->fac_ptr.push_back(0);
            for (size_t i = 0; i < nodes_.size(); ++i) {
                const Node& nd = nodes_[i];
                if (nd.depth != d || nd.factors.empty()) continue;
                if (kind == 0 && nd.leaf) continue;
                if (kind == 1 && !nd.leaf) continue;
                const std::vector<VariableID>& v = kind == 0 ? nd.separator : nd.vars;
                const std::vector<FactorID>& fc = kind == 0 ? nd.sepFactors : nd.factors;
                lp->node.push_back((int)i);
                lp->free_vid.insert(lp->free_vid.end(), v.begin(), v.end());
                lp->fac_id.insert(lp->fac_id.end(), fc.begin(), fc.end());
                lp->free_ptr.push_back((int64_t)lp->free_vid.size());
                lp->fac_ptr.push_back((int64_t)lp->fac_id.size());
            }
            if (lp->node.empty()) { delete lp; continue; }
            const size_t ncomp = lp->node.size();
            lp->x.resize(lp->free_vid.size()); lp->fret.resize(ncomp); lp->delta.resize(ncomp);
            lp->iters.resize(ncomp); lp->status.resize(ncomp);
            if (batch_ && f_.numDevices() == 1)
                check(f_.deviceContext(), rdis_hip_plan_create(p, (int64_t)ncomp, lp->free_ptr.data(), lp->free_vid.data(),
                                                               lp->fac_ptr.data(), lp->fac_id.data(), &lp->plan), "rdis_hip_plan_create");
            plans_.push_back(lp);
        }
}

double HipRDISLevelOptimizer::runPlan(LevelPlan& lp, int sweep, double objective, bool printInfo) {
    const double t0 = now_ms();
    const VariablePtrVec& vars = f_.getVariables();
    const FactorPtrVec& facs = f_.getFactors();
    const size_t ncomp = lp.node.size();
    long long iters = 0;
    double dsum = 0.0;
    if (batch_ && f_.numDevices() > 1) {
        // several devices (OptimizableFunction::setDevices): the level's components are shared out over them by
        // HipCGDSubspaceOptimizer::optimizeBatch (heaviest first by factor count; launches on all devices before any
        // result is fetched; its plan cache keeps every device's share resident from the first sweep on)
        std::vector<HipCGDSubspaceOptimizer::Component> comps(ncomp);
        for (size_t c = 0; c < ncomp; ++c) {
            HipCGDSubspaceOptimizer::Component& C = comps[c];
            for (int64_t i = lp.free_ptr[c]; i < lp.free_ptr[c + 1]; ++i) {
                Variable* v = vars[(size_t)lp.free_vid[(size_t)i]];
                C.vars.push_back(v); C.xval.push_back(v->eval());
            }
            for (int64_t j = lp.fac_ptr[c]; j < lp.fac_ptr[c + 1]; ++j) C.factors.push_back(facs[(size_t)lp.fac_id[(size_t)j]]);
        }
        ss_.optimizeBatch(comps, false);
        for (size_t c = 0; c < ncomp; ++c) {
            const HipCGDSubspaceOptimizer::Component& C = comps[c];
            lp.fret[c] = C.fret; lp.delta[c] = C.deltaFval; lp.iters[c] = C.iters; lp.status[c] = C.status;
            for (size_t i = 0; i < C.xval.size(); ++i) lp.x[(size_t)lp.free_ptr[c] + i] = C.xval[i];
            if ((C.status & 0xff) == RDIS_HIP_EXIT_EMPTY) continue;
            dsum += C.deltaFval;
            iters += C.iters + 1;
        }
    } else if (batch_) {
        rdis_hip_problem* p = f_.deviceProblem();   // (pushes pending host-side assignments)
        (void)p;
        rdis_hip_ctx* ctx = f_.deviceContext();
        check(ctx, rdis_hip_plan_set_start(lp.plan, nullptr), "rdis_hip_plan_set_start");   // from the values assigned on the device
        check(ctx, rdis_hip_plan_solve(lp.plan, (int32_t)ssmaxit_, ssftol_), "rdis_hip_plan_solve");
        check(ctx, rdis_hip_plan_fetch(lp.plan, lp.x.data(), lp.fret.data(), lp.delta.data(), lp.iters.data(), lp.status.data(), nullptr, nullptr),
              "rdis_hip_plan_fetch");
        for (size_t c = 0; c < ncomp; ++c) {
            const int st = lp.status[c] & 0xff;
            if (st == RDIS_HIP_EXIT_SYNC_TIMEOUT) throw HipError(RDIS_HIP_EDEVICE, "level driver: device-side exchange timed out");
            if (st == RDIS_HIP_EXIT_EMPTY) continue;
            dsum += lp.delta[c];
            iters += lp.iters[c] + 1;
        }
        // the variables are left assigned on the device; mirror the values on the host side
        f_.adoptDeviceValues(lp.free_vid, lp.x);
    } else {
        // the same calls one at a time, in the order the reference visits siblings (Component.cpp:603-608:
        // fewer variables first -- the order the labelling returns them in)
        for (size_t c = 0; c < ncomp; ++c) {
            VariablePtrVec cv; FactorPtrVec cf; NumericVec xv;
            for (int64_t i = lp.free_ptr[c]; i < lp.free_ptr[c + 1]; ++i) {
                Variable* v = vars[(size_t)lp.free_vid[(size_t)i]];
                cv.push_back(v); xv.push_back(v->eval());
            }
            for (int64_t j = lp.fac_ptr[c]; j < lp.fac_ptr[c + 1]; ++j) cf.push_back(facs[(size_t)lp.fac_id[(size_t)j]]);
            Numeric delta = 0;
            lp.fret[c] = ss_.optimize(cv, cf, xv, delta, false);
            lp.delta[c] = delta; lp.iters[c] = ss_.lastIters(); lp.status[c] = ss_.lastStatus();
            for (size_t i = 0; i < xv.size(); ++i) lp.x[(size_t)lp.free_ptr[c] + i] = xv[i];
            dsum += delta;
            iters += ss_.lastIters() + 1;
        }
    }
    objective += dsum;
    Step s;
    s.sweep = sweep; s.depth = lp.depth; s.kind = lp.kind; s.ncomp = (long long)ncomp;
    s.nvars = (long long)lp.free_vid.size(); s.nfactors = (long long)lp.fac_id.size(); s.iters = iters;
    s.objective = objective; s.ms = now_ms() - t0;
    trace_.push_back(s);
    if (printInfo)
        std::cout << "sweep " << sweep << " depth " << lp.depth << (lp.kind == 0 ? " separators: " : " leaves: ") << ncomp << " component(s), "
                  << lp.free_vid.size() << " variables, " << lp.fac_id.size() << " factors -> " << objective << " (" << s.ms << " ms)" << std::endl;
    return objective;
}

void HipRDISLevelOptimizer::decompose() {
    buildTree();
    buildPlans();
}

void HipRDISLevelOptimizer::planLists(size_t i, int& depth, int& kind, std::vector<int64_t>& free_ptr, std::vector<int64_t>& free_vid,
                                      std::vector<int64_t>& fac_ptr, std::vector<int64_t>& fac_id) const {
    const LevelPlan& lp = *plans_.at(i);
    depth = lp.depth; kind = lp.kind;
    free_ptr = lp.free_ptr; free_vid = lp.free_vid; fac_ptr = lp.fac_ptr; fac_id = lp.fac_id;
}

Numeric HipRDISLevelOptimizer::optimize(bool printInfo) {
    for (const Variable* v : f_.getVariables())
        if (!v->isAssigned()) throw std::logic_error("HipRDISLevelOptimizer::optimize: assign an initial state first");
    ssmaxit_ = ss_.getMaxIters(); ssftol_ = ss_.getFtol();
    buildTree();
    buildPlans();
    trace_.clear();
    double objective = f_.eval();
    if (printInfo) {
        size_t nleaf = 0, nsplit = 0;
        for (const Node& nd : nodes_) (nd.leaf ? nleaf : nsplit)++;
        std::cout << "level driver: " << nodes_.size() << " components (" << nsplit << " split by a separator, " << nleaf << " leaves) in "
                  << plans_.size() << " launches per sweep; decomposition " << decomp_ms_ << " ms; initial value " << objective << std::endl;
    }
    sweeps_ = 0;
    for (int sweep = 0; sweep < maxSweeps_; ++sweep) {
        const double before = objective;
        for (LevelPlan* lp : plans_) objective = runPlan(*lp, sweep, objective, printInfo);
        ++sweeps_;
        if (!(before - objective > steptol_)) break;   // no progress beyond steptol (RDISOptimizer.cpp:1089-1108)
    }
    const double check_f = f_.eval();   // the running sum of the launches' deltas IS the function value
    if (printInfo) std::cout << "level driver: final value " << check_f << " (sum of deltas: " << objective << ") after " << sweeps_ << " sweep(s)" << std::endl;
    return check_f;
}

// ---------------------------------------------------------------------------------------------------------------
// The reference's schedule on the static tree (header comment; every rule cites src/RDISOptimizer.cpp)
struct HipRDISLevelOptimizer::NodeState {
    unsigned nrr = 0, va = 0;        // Component::numRandomRestarts, numVAsinceLastRR (src/Component.h:186-196)
    bool assigned = false;           // the node's variables have been valued in this visit (vardata.isAssigned(repvid), :1023)
    bool lastOpt = false;            // Component::wasLastEvalOpt (:1554)
    bool havePrev = false, haveOpt = false;   // getPrevSD / getOptSD (:1025-1026, 1514)
    double opt = 0.0;
    std::vector<double> optx;        // the node's variables at its best evaluation (the optimum subdomain)
    bool randomInit = false;         // its initial values have been used up by an ancestor's restart (:1127, 1134-1136)
    bool redo = false;               // this step was made again from a random state after no progress (doAlternatingMin = false, :1088-1091)
    int kind = 0;
    double fret = 0.0, delta = 0.0;
    unsigned long long hash = 0;
};

double HipRDISLevelOptimizer::restartValue(unsigned long long seed, int node, int restart, VariableID vid, const VariableDomain& dom) {
    // sampleRandomState (:1196-1216): uniform over the sampling interval, clamped into the domain.  The number comes
    // from splitmix64 of (seed, node, restart, variable) instead of the reference's shared generator.
    unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (unsigned long long)(node + 1) + 0xBF58476D1CE4E5B9ull * (unsigned long long)(restart + 1) +
                           0x94D049BB133111EBull * (unsigned long long)(vid + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    const double u = (double)(z >> 11) * (1.0 / 9007199254740992.0);
    const double lo = dom.samplingMin(), hi = dom.samplingMax();
    return dom.closestVal(lo + u * (hi - lo));
}

void HipRDISLevelOptimizer::runSet(const std::vector<int>& set, const std::vector<char>& randomInit, std::vector<NodeState>& st, bool printInfo) {
    const VariablePtrVec& vars = f_.getVariables();
    const FactorPtrVec& facs = f_.getFactors();
    for (size_t i = 0; i < set.size(); ++i) { st[(size_t)set[i]] = NodeState(); st[(size_t)set[i]].randomInit = randomInit[i] != 0; }
    auto numRestarts = [&](int n) {   // :978-983 (level = depth + 1; a top component: depth 0)
        const unsigned lvl = nrr_per_lvl_ >> std::min(31, nodes_[(size_t)n].depth);
        return std::max(min_rr_, nodes_[(size_t)n].depth == 0 ? nrr_at_top_ : lvl);
    };
    auto valued = [&](int n) -> const std::vector<VariableID>& { return nodes_[(size_t)n].leaf ? nodes_[(size_t)n].vars : nodes_[(size_t)n].separator; };
    auto valuedFactors = [&](int n) -> const std::vector<FactorID>& { return nodes_[(size_t)n].leaf ? nodes_[(size_t)n].factors : nodes_[(size_t)n].sepFactors; };
    auto finish = [&](int n) {   // the node is left at its best evaluation (the optimum subdomain its parent multiplies in, :1484-1494)
        NodeState& s = st[(size_t)n];
        if (!s.haveOpt) return;
        const std::vector<VariableID>& nv = nodes_[(size_t)n].vars;
        for (size_t i = 0; i < nv.size(); ++i)
            if (vars[(size_t)nv[i]]->eval() != s.optx[i]) { vars[(size_t)nv[i]]->assign(s.optx[i]); f_.onVarAssigned(nv[i], s.optx[i]); }
    };
    std::vector<int> active(set);
    while (!active.empty()) {
        if (ref_calls_ >= max_calls_) {   // the reference's time limit (checkTimedOut, :317-321): every loop ends where it stands
            for (int n : active) finish(n);
            return;
        }
        // getValueFromDomain's entry (:986-1000) and getSSInitialVal (:1120-1147): who goes on, and from where
        std::vector<int> pending;
        std::vector<std::vector<double> > start(nodes_.size());
        for (int n : active) {
            NodeState& s = st[(size_t)n];
            const bool top = nodes_[(size_t)n].depth == 0;
            const bool forceRR = !(top && no_assign_limit_at_top_) && s.va >= max_na_to_rr_;          // :992-994
            if ((forceRR || !s.lastOpt) && s.nrr > numRestarts(n)) { finish(n); continue; }           // :997-999
            const std::vector<VariableID>& V = valued(n);
            std::vector<double>& x0 = start[(size_t)n];
            if (!s.assigned && !s.randomInit) { s.kind = 0; for (VariableID v : V) x0.push_back(vars[(size_t)v]->eval()); }        // xvalinit (:1127-1130)
            else if (s.assigned && !forceRR) { s.kind = 1; for (VariableID v : V) x0.push_back(vars[(size_t)v]->eval()); }        // :1131-1133
            else { s.kind = 2; for (VariableID v : V) x0.push_back(restartValue(restart_seed_, n, (int)s.nrr, v, vars[(size_t)v]->getDomain())); }   // :1134-1136
            if (s.kind != 1) { ++s.nrr; s.va = 0; }                                                     // :1047, Component.h:190-194
            s.redo = false;
            pending.push_back(n);
        }
        // the subspace optimizer, and again from a random state where it made no progress (:1032-1106)
        std::vector<int> succeeded;
        while (!pending.empty()) {
            std::vector<HipCGDSubspaceOptimizer::Component> comps(pending.size());
            for (size_t i = 0; i < pending.size(); ++i) {
                const int n = pending[i];
                for (VariableID v : valued(n)) comps[i].vars.push_back(vars[(size_t)v]);
                for (FactorID fa : valuedFactors(n)) comps[i].factors.push_back(facs[(size_t)fa]);
                comps[i].xval = start[(size_t)n];
            }
            ss_.optimizeBatch(comps, false);
            ref_calls_ += (long long)pending.size();
            ++ref_batches_;
            for (const HipCGDSubspaceOptimizer::Component& C : comps)
                if ((C.status & 0xff) != RDIS_HIP_EXIT_EMPTY) { ref_iters_ += C.iters + 1; ref_fevals_ += C.nfeval; }
            std::vector<int> again;
            const double ftol = steptol_;   // :1083-1084
            for (size_t i = 0; i < pending.size(); ++i) {
                const int n = pending[i];
                NodeState& s = st[(size_t)n];
                s.fret = comps[i].fret; s.delta = comps[i].deltaFval;
                unsigned long long h = 1469598103934665603ull;
                for (double v : start[(size_t)n]) { unsigned long long b; std::memcpy(&b, &v, 8); h = (h ^ b) * 1099511628211ull; }
                s.hash = h;
                if (s.havePrev && s.delta >= 0.0 - ftol) {       // approxgeq(deltafval, 0, ftol) (:1086, common.h:74-76)
                    RefStep rs{n, s.kind, (int)s.nrr, (int)s.va, s.fret, s.delta, std::nan(""), 0, s.hash};
                    ref_trace_.push_back(rs);
                    if (s.nrr < numRestarts(n)) {                 // :1087-1094: try again from a random position
                        s.kind = 2;
                        std::vector<double>& x0 = start[(size_t)n];
                        x0.clear();
                        for (VariableID v : valued(n)) x0.push_back(restartValue(restart_seed_, n, (int)s.nrr, v, vars[(size_t)v]->getDomain()));
                        ++s.nrr; s.va = 0;
                        s.redo = true;
                        again.push_back(n);
                    } else {
                        finish(n);                                // :1095-1099: a failure ends the node's loop (:279)
                    }
                } else {
                    succeeded.push_back(n);
                }
            }
            pending.swap(again);
        }
        std::sort(succeeded.begin(), succeeded.end());
        // assign (:282), decompose and recurse into the children (:289-314) -- all children of all these nodes together
        std::vector<int> kids;
        std::vector<char> kidsRandom;
        for (int n : succeeded) {
            NodeState& s = st[(size_t)n];
            s.assigned = true;
            ++s.va;                                               // Component.cpp:221
            // Where do the children start?  A step that succeeded with doAlternatingMin still set and a previous subdomain
            // -- every iterative improvement, and a FORCED restart that made progress at once -- hands the children the
            // values they had in that subdomain (setInitialValFromChildren, :1112-1114, 1713-1724): they start from where
            // they stand.  A restart after no progress (:1088-1091) and a first visit from a random state have nothing to
            // hand down: their children's initial values are random ones (:1162-1171).
            const bool inherit = s.havePrev && !s.redo;
            for (int c : children_[(size_t)n]) { kids.push_back(c); kidsRandom.push_back(s.kind == 2 && !inherit ? 1 : 0); }
        }
        if (!kids.empty()) runSet(kids, kidsRandom, st, printInfo);
        // updateDomain (:1507-1577)
        for (int n : succeeded) {
            NodeState& s = st[(size_t)n];
            const Node& nd = nodes_[(size_t)n];
            double value = s.fret;                                // a leaf's value IS what its solve returned
            if (!nd.leaf) {
                FactorPtrVec fl;
                for (FactorID fa : nd.factors) fl.push_back(facs[(size_t)fa]);
                Numeric ferr = 0;
                value = f_.evalFactors(fl, ferr, true);
            }
            bool isNewMin = false;
            if (!s.haveOpt || value < s.opt) {                                        // :1521-1525
                isNewMin = !s.haveOpt || !(std::fabs(value - s.opt) < steptol_);     // approxeq (common.h:66-68)
                s.opt = value; s.haveOpt = true;
                s.optx.resize(nd.vars.size());
                for (size_t i = 0; i < nd.vars.size(); ++i) s.optx[i] = vars[(size_t)nd.vars[i]]->eval();
            }
            s.lastOpt = isNewMin;                                                      // :1554
            s.havePrev = true;                                                         // :1576
            RefStep rs{n, s.kind, (int)s.nrr, (int)s.va, s.fret, s.delta, value, isNewMin ? 1 : 0, s.hash};
            ref_trace_.push_back(rs);
            if (printInfo && nd.depth == 0)
                std::cout << "node " << n << ": " << (s.kind == 0 ? "initial values" : s.kind == 1 ? "iterative improvement" : "random restart")
                          << ", nRR " << s.nrr << " / " << numRestarts(n) << " -> " << value << (isNewMin ? " (new minimum)" : "") << std::endl;
        }
        active.swap(succeeded);
    }
}

Numeric HipRDISLevelOptimizer::optimizeReferenceSchedule(bool printInfo) {
    for (const Variable* v : f_.getVariables())
        if (!v->isAssigned()) throw std::logic_error("HipRDISLevelOptimizer::optimizeReferenceSchedule: assign an initial state first");
    buildTree();
    children_.assign(nodes_.size(), std::vector<int>());
    std::vector<int> roots;
    for (size_t i = 0; i < nodes_.size(); ++i) {
        if (nodes_[i].factors.empty()) continue;   // checkEmpty (:262)
        if (nodes_[i].parent >= 0) children_[(size_t)nodes_[i].parent].push_back((int)i); else roots.push_back((int)i);
    }
    ref_trace_.clear();
    ref_calls_ = ref_iters_ = ref_fevals_ = ref_batches_ = 0;
    std::vector<NodeState> st(nodes_.size());
    runSet(roots, std::vector<char>(roots.size(), 0), st, printInfo);
    return f_.eval();
}

}  // namespace rdis
