// rdis_host.cpp -- see rdis_host.h.  Host bookkeeping only; all arithmetic on the
// path goes through include/rdis_hip.h.
#include "rdis_host.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <sstream>

#include "../../include/rdis_hip.h"

namespace rdis {

namespace {
void check(rdis_hip_ctx* ctx, int rc, const char* where) {
    if (rc != 0) throw HipError(rc, std::string(where) + ": " + (ctx ? rdis_hip_last_error(ctx) : "no context"));
}
// split on " \t,;" with token compression (OptimizableFunction::readAndSplit,
// src/OptimizableFunction.cpp:354-371); skips blank and '#' lines
bool readAndSplit(std::istream& in, std::vector<std::string>& out) {
    std::string line;
    while (std::getline(in, line)) {
        if (line.empty() || line[0] == '#') continue;
        out.clear();
        std::string cur;
        for (char ch : line) {
            if (ch == ' ' || ch == '\t' || ch == ',' || ch == ';' || ch == '\r') {
                if (!cur.empty()) { out.push_back(cur); cur.clear(); }
            } else cur.push_back(ch);
        }
        if (!cur.empty()) out.push_back(cur);
        if (out.empty()) continue;
        return true;
    }
    return false;
}
std::string trim(const std::string& s) {
    size_t a = s.find_first_not_of(" \t\r\n"), b = s.find_last_not_of(" \t\r\n");
    return a == std::string::npos ? std::string() : s.substr(a, b - a + 1);
}
std::string lower(std::string s) {
    for (char& c : s) c = (char)std::tolower((unsigned char)c);
    return s;
}
bool parseNumber(const std::string& s, double& v) {
    char* end = nullptr;
    v = std::strtod(s.c_str(), &end);
    return !s.empty() && end && *end == '\0';
}
}  // namespace

// ------------------------------------------------------------------ VariableDomain
VariableDomain::VariableDomain(const std::string& domain) : lo_(0), hi_(0), slo_(0), shi_(0) {
    std::vector<double> v;
    std::string cur;
    for (char ch : domain + " ") {
        if (ch == ' ' || ch == '~' || ch == ':' || ch == ',') {
            double d;
            if (!cur.empty() && parseNumber(cur, d)) v.push_back(d);
            cur.clear();
        } else cur.push_back(ch);
    }
    if (v.size() < 2) throw std::invalid_argument("VariableDomain: cannot parse '" + domain + "'");
    lo_ = v.front(); hi_ = v.back();
    slo_ = lo_; shi_ = hi_;
}

// ------------------------------------------------------------------ Variable
void Variable::assign(Numeric newval) {
    assigned_ = true;
    value_ = newval;
    if (owner_) owner_->markDirty(id_);
}
Numeric Variable::eval() const {
    if (!assigned_) throw std::logic_error("Variable::eval: variable " + name_ + " is not assigned");
    return value_;
}

// ------------------------------------------------------------------ Factor
bool Factor::areAllVarsAssigned() const {
    for (const Variable* v : variables_) if (!v->isAssigned()) return false;
    return true;
}
Numeric Factor::eval() const {
    if (!owner_) throw std::logic_error("Factor::eval: factor is not part of a function");
    double out = 0;
    int64_t id = id_;
    rdis_hip_problem* p = owner_->deviceProblem();
    check(owner_->deviceContext(), rdis_hip_eval_each(p, 1, &id, &out), "Factor::eval");
    return out;
}
void Factor::computeGradient(PartialGradient& g) const {
    FactorPtrVec one(1, const_cast<Factor*>(this));
    owner_->computeGradient(one, g);
}

void NonlinearProductFactor::addVariable(Variable* v, Numeric exponent, Numeric constant, bool useSine) {
    if (exponent == 0.0) return;
    for (const Variable* u : variables_) if (u == v) return;
    variables_.push_back(v);
    terms_.push_back(Term{exponent, constant, useSine});
    v->getFactors().push_back(this);
}

// ------------------------------------------------------------------ OptimizableFunction
OptimizableFunction::OptimizableFunction() : defaultDomain(0, 0), ctx_(nullptr), prob_(nullptr), device_(0) {}

OptimizableFunction::~OptimizableFunction() {
    // optimisers that outlive the function must not be left with plans of a freed device problem
    for (HipCGDSubspaceOptimizer* o : plan_holders_) o->functionGone();
    plan_holders_.clear();
    for (rdis_hip_comm* m : comms_) rdis_hip_comm_destroy(m);
    comms_.clear();
    if (prob_) rdis_hip_free_problem(prob_);
    if (ctx_) rdis_hip_destroy(ctx_);
    for (Replica& r : replicas_) {
        if (r.prob) rdis_hip_free_problem(r.prob);
        if (r.ctx) rdis_hip_destroy(r.ctx);
    }
    for (Variable* v : variables) delete v;
    for (Factor* f : factors) delete f;
}

Variable* OptimizableFunction::addVariable(const std::string& name, const VariableDomain& dom, VariableID& id) {
    // a name that exists returns the existing variable (src/OptimizableFunction.cpp:61-63); by index, not
    // by the reference's linear search: a BAL problem with 3e5 variables spent 70 s in it
    auto it = by_name_.find(name);
    if (it != by_name_.end()) return it->second;
    Variable* v = new Variable(id, name, dom, this);
    variables.push_back(v);
    by_name_.emplace(name, v);
    ++id;
    return v;
}
void OptimizableFunction::addFactor(Factor* f) {
    f->owner_ = this;
    factors.push_back(f);
}

const OptimizableFunction::Packed& OptimizableFunction::packed() const {
    if (packed_) return *packed_;
    std::unique_ptr<Packed> P(new Packed);
    const size_t N = variables.size(), F = factors.size();
    P->lo.resize(N); P->hi.resize(N);
    for (size_t i = 0; i < N; ++i) { P->lo[i] = variables[i]->getDomain().min(); P->hi[i] = variables[i]->getDomain().max(); }
    const bool ba = F > 0 && dynamic_cast<const BundleAdjustmentFactor*>(factors[0]) != nullptr;
    P->kind = ba ? 0 : 1;
    if (ba) {
        P->cam_vid0.resize(F); P->pt_vid0.resize(F); P->obs.resize(2 * F);
        for (size_t i = 0; i < F; ++i) {
            const BundleAdjustmentFactor* bf = dynamic_cast<const BundleAdjustmentFactor*>(factors[i]);
            if (!bf || bf->getVariables().size() != 12) throw std::logic_error("packed: mixed factor kinds");
            const VariablePtrVec& v = bf->getVariables();
            for (int k = 1; k < 9; ++k) if (v[k]->getID() != v[0]->getID() + k) throw std::logic_error("packed: camera block not contiguous");
            for (int k = 1; k < 3; ++k) if (v[9 + k]->getID() != v[9]->getID() + k) throw std::logic_error("packed: point block not contiguous");
            P->cam_vid0[i] = v[0]->getID(); P->pt_vid0[i] = v[9]->getID();
            bf->getObservation(P->obs[2 * i], P->obs[2 * i + 1]);
        }
    } else {
        P->rowptr.assign(1, 0);
        for (size_t i = 0; i < F; ++i) {
            const NonlinearProductFactor* nf = dynamic_cast<const NonlinearProductFactor*>(factors[i]);
            if (!nf) throw std::logic_error("packed: unsupported factor kind (no device descriptor)");
            P->coeff.push_back(nf->getCoeff());
            if (nf->usesExponential()) { P->useexp.resize(F, 0); P->useexp[i] = 1; }
            for (size_t k = 0; k < nf->terms().size(); ++k) {
                P->vid.push_back(nf->getVariables()[k]->getID());
                P->expo.push_back(nf->terms()[k].exponent);
                P->cons.push_back(nf->terms()[k].constant);
                P->sine.push_back(nf->terms()[k].useSine ? 1 : 0);
            }
            P->rowptr.push_back((int64_t)P->vid.size());
        }
    }
    packed_ = std::move(P);
    return *packed_;
}

void OptimizableFunction::initDevice(int device) { device_ = device; ensureUploaded(); }

void OptimizableFunction::uploadTo(int device, rdis_hip_ctx*& ctx, rdis_hip_problem*& prob) const {
    const Packed& P = packed();
    if (!ctx) {
        int rc = rdis_hip_create(device, &ctx);
        if (rc != 0) throw HipError(rc, "rdis_hip_create failed: no usable MI355X / HIP runtime (there is no CPU fallback)");
    }
    const size_t N = variables.size();
    std::vector<double> x0(N, 0.0);
    for (size_t i = 0; i < N; ++i) if (variables[i]->assigned_) x0[i] = variables[i]->value_;
    int rc;
    if (P.kind == 0)
        rc = rdis_hip_upload_ba(ctx, (int64_t)N, x0.data(), P.lo.data(), P.hi.data(), (int64_t)factors.size(),
                                P.cam_vid0.data(), P.pt_vid0.data(), P.obs.data(), &prob);
    else
        rc = rdis_hip_upload_nlp(ctx, (int64_t)N, x0.data(), P.lo.data(), P.hi.data(), (int64_t)factors.size(),
                                 P.coeff.data(), P.rowptr.data(), P.vid.data(), P.expo.data(), P.cons.data(),
                                 P.sine.data(), &prob);
    check(ctx, rc, "upload");
    if (P.kind != 0 && !P.useexp.empty()) check(ctx, rdis_hip_nlp_set_exponential(prob, P.useexp.data()), "nlp_set_exponential");
}

void OptimizableFunction::ensureUploaded() const {
    if (prob_) return;
    uploadTo(device_, ctx_, prob_);
    dirty_.clear();
    is_dirty_.assign(variables.size(), 0);
}

void OptimizableFunction::setDevices(const std::vector<int>& devices) {
    if (devices.empty()) throw std::invalid_argument("setDevices: empty device list");
    if (prob_ || !replicas_.empty()) throw std::logic_error("setDevices: the function is on a device already");
    device_ = devices[0];
    replicas_.resize(devices.size() - 1);
    for (size_t d = 1; d < devices.size(); ++d) replicas_[d - 1].device = devices[d];
}

void OptimizableFunction::markDirty(VariableID id) const {
    if (prob_ && !is_dirty_[(size_t)id]) { is_dirty_[(size_t)id] = 1; dirty_.push_back(id); }   // (everything is uploaded at first use)
    for (Replica& r : replicas_)
        if (r.prob && !r.is_dirty[(size_t)id]) { r.is_dirty[(size_t)id] = 1; r.dirty.push_back(id); }
}

void OptimizableFunction::markDirtyElsewhere(VariableID id, size_t holder) const {
    if (holder != 0 && prob_ && !is_dirty_[(size_t)id]) { is_dirty_[(size_t)id] = 1; dirty_.push_back(id); }
    for (size_t d = 1; d <= replicas_.size(); ++d) {
        Replica& r = replicas_[d - 1];
        if (d != holder && r.prob && !r.is_dirty[(size_t)id]) { r.is_dirty[(size_t)id] = 1; r.dirty.push_back(id); }
    }
}

rdis_hip_problem* OptimizableFunction::deviceProblem(size_t d) const {
    if (d == 0) return deviceProblem();
    Replica& r = replicas_.at(d - 1);
    if (!r.prob) {
        uploadTo(r.device, r.ctx, r.prob);
        r.dirty.clear();
        r.is_dirty.assign(variables.size(), 0);
    }
    if (!r.dirty.empty()) {
        std::vector<int64_t> ids(r.dirty.begin(), r.dirty.end());
        std::vector<double> vals(ids.size());
        for (size_t i = 0; i < ids.size(); ++i) { vals[i] = variables[(size_t)ids[i]]->value_; r.is_dirty[(size_t)ids[i]] = 0; }
        r.dirty.clear();
        check(r.ctx, rdis_hip_set_x(r.prob, (int64_t)ids.size(), ids.data(), vals.data()), "set_x");
    }
    return r.prob;
}
rdis_hip_ctx* OptimizableFunction::deviceContext(size_t d) const {
    if (d == 0) return deviceContext();
    (void)deviceProblem(d);
    return replicas_.at(d - 1).ctx;
}
rdis_hip_comm* const* OptimizableFunction::deviceComms() const {
    if (!comms_tried_) {
        comms_tried_ = true;
        const size_t n = numDevices();
        bool distinct = n > 1;
        for (size_t a = 0; a < n && distinct; ++a)
            for (size_t b = 0; b < a; ++b) if (deviceOrdinal(a) == deviceOrdinal(b)) distinct = false;
        if (distinct) {
            std::vector<rdis_hip_ctx*> ctxs(n);
            for (size_t d = 0; d < n; ++d) ctxs[d] = deviceContext(d);
            comms_.assign(n, nullptr);
            if (rdis_hip_comm_create_all((int32_t)n, ctxs.data(), comms_.data()) != 0) comms_.clear();   // (no RCCL: the host sums)
        }
    }
    return comms_.empty() ? nullptr : comms_.data();
}

void OptimizableFunction::pushAssignments() const {
    if (dirty_.empty()) return;
    std::vector<int64_t> ids(dirty_.begin(), dirty_.end());
    std::vector<double> vals(ids.size());
    for (size_t i = 0; i < ids.size(); ++i) { vals[i] = variables[(size_t)ids[i]]->value_; is_dirty_[(size_t)ids[i]] = 0; }
    dirty_.clear();
    check(ctx_, rdis_hip_set_x(prob_, (int64_t)ids.size(), ids.data(), vals.data()), "set_x");
}

rdis_hip_problem* OptimizableFunction::deviceProblem() const { ensureUploaded(); pushAssignments(); return prob_; }
rdis_hip_ctx* OptimizableFunction::deviceContext() const { ensureUploaded(); return ctx_; }

void OptimizableFunction::assignAll(const NumericVec& x) {
    if (x.size() != variables.size()) throw std::invalid_argument("assignAll: size mismatch");
    for (size_t i = 0; i < x.size(); ++i) { variables[i]->assign(x[i]); onVarAssigned((VariableID)i, x[i]); }
}

void OptimizableFunction::fillIds(const FactorPtrVec& fs, std::vector<int64_t>& ids) const {
    ids.resize(fs.size());
    for (size_t i = 0; i < fs.size(); ++i) {
        if (!fs[i]->areAllVarsAssigned()) throw std::logic_error("factor with unassigned variables handed to the device path");
        ids[i] = fs[i]->getID();
    }
}

Numeric OptimizableFunction::eval() const {
    Numeric ferr = 0;
    return evalFactors(factors, ferr, true);
}
Numeric OptimizableFunction::evalFactors(const FactorPtrVec& fctrs, Numeric& ferr, bool) const {
    ferr = 0.0;
    if (fctrs.empty()) return 0.0;
    std::vector<int64_t> ids;
    fillIds(fctrs, ids);
    double f = 0;
    rdis_hip_problem* p = deviceProblem();
    check(ctx_, rdis_hip_eval(p, (int64_t)ids.size(), ids.data(), &f), "evalFactors");
    return f;
}
void OptimizableFunction::computeGradient(const FactorPtrVec& facs, PartialGradient& gradient, bool) const {
    gradient.clear();
    if (facs.empty()) return;
    std::vector<int64_t> ids;
    fillIds(facs, ids);
    std::vector<double> g(variables.size());
    double f = 0;
    rdis_hip_problem* p = deviceProblem();
    check(ctx_, rdis_hip_eval_grad(p, (int64_t)ids.size(), ids.data(), &f, g.data()), "computeGradient");
    std::vector<char> touched(variables.size(), 0);
    for (const Factor* fa : facs) for (const Variable* v : fa->getVariables()) touched[(size_t)v->getID()] = 1;
    for (size_t i = 0; i < g.size(); ++i) if (touched[i]) gradient.push_back(std::make_pair((VariableID)i, g[i]));
}
void OptimizableFunction::computeGradient(NumericVec& gradient, bool) const {
    PartialGradient pg;
    computeGradient(factors, pg);
    gradient.assign(variables.size(), 0.0);
    for (const auto& kv : pg) gradient[(size_t)kv.first] = kv.second;
}

void OptimizableFunction::adoptDeviceValues(const std::vector<int64_t>& vids, const std::vector<double>& vals) {
    for (size_t i = 0; i < vids.size(); ++i) {
        Variable* v = variables[(size_t)vids[i]];
        v->assigned_ = true; v->value_ = vals[i];
        onVarAssigned(v->getID(), vals[i]);
        if (!replicas_.empty()) markDirtyElsewhere(v->getID(), 0);   // (the primary assigned them)
    }
}

// ------------------------------------------------------------------ BundleAdjustmentFunction
void BundleAdjustmentFunction::getBlockRangeByBlkId(VariableCount b, VariableID& lo, VariableID& hi) const {
    if (b < ncams_) { lo = getCamVID(b, 0); hi = getCamVID(b, 8); }
    else { lo = getPointVID(b - ncams_, 0); hi = getPointVID(b - ncams_, 2); }
}

void BundleAdjustmentFunction::setDomain(VariableID vid, Numeric init) {
    const long long ncv = ncams_ * 9;
    const int type = vid < ncv ? (int)(vid % 9) : 9 + (int)((vid - ncv) % 3);
    const double dsf = 1000.0, pi = 3.14159265358979323846;
    double slo, shi, dlo, dhi;
    if (type < 3) { slo = -1.0 * pi; shi = 1.0 * pi; dlo = slo * dsf; dhi = shi * dsf; }
    else if (type <= 5 || type >= 9) { slo = init + -100.0; shi = init + 100.0; dlo = slo * dsf; dhi = shi * dsf; }
    else if (type == 6) {
        slo = init + -100.0; shi = init + 100.0;
        dlo = std::max(std::min(slo, slo * dsf), 0.0);
        dhi = shi * dsf;
    } else if (type == 7) { slo = init + -1e-4; shi = init + 1e-4; dlo = -1e-1; dhi = 1e-1; }
    else { slo = init + -1e-6; shi = init + 1e-6; dlo = -1e-3; dhi = 1e-3; }
    variables[(size_t)vid]->setDomain(VariableDomain(std::min(dlo, slo), std::max(dhi, shi), slo, shi));  // hull; sampling interval
}

bool BundleAdjustmentFunction::load(const std::string& file, VariableCount numcams, VariableCount numpoints) {
    std::ifstream in(file.c_str());
    if (!in.is_open()) { std::cerr << "BundleAdjustmentFunction::load: cannot open " << file << std::endl; return false; }
    std::vector<std::string> sv;
    if (!readAndSplit(in, sv) || sv.size() != 3) return false;
    const long long fc = std::atoll(sv[0].c_str()), fp = std::atoll(sv[1].c_str()), nobs = std::atoll(sv[2].c_str());
    ncams_ = numcams <= 0 ? fc : numcams;
    npts_ = numpoints <= 0 ? fp : numpoints;
    if (ncams_ > fc || npts_ > fp) return false;
    VariableID id = 0;
    char name[64];
    for (long long c = 0; c < ncams_; ++c)
        for (int k = 0; k < 9; ++k) { std::snprintf(name, sizeof name, "C%lld.%d", c, k); addVariable(name, VariableDomain(0, 0), id); }
    for (long long p = 0; p < npts_; ++p)
        for (int k = 0; k < 3; ++k) { std::snprintf(name, sizeof name, "x%lld.%c", p, (char)('x' + k)); addVariable(name, VariableDomain(0, 0), id); }
    for (long long i = 0; i < nobs; ++i) {
        if (!readAndSplit(in, sv) || sv.size() != 4) return false;
        const long long cam = std::atoll(sv[0].c_str()), pt = std::atoll(sv[1].c_str());
        if (cam >= ncams_ || pt >= npts_) continue;
        BundleAdjustmentFactor* f = new BundleAdjustmentFactor((FactorID)factors.size(), cam, pt,
                                                               std::strtod(sv[2].c_str(), nullptr), std::strtod(sv[3].c_str(), nullptr));
        addFactor(f);
        for (int k = 0; k < 9; ++k) f->addVariable(variables[(size_t)getCamVID(cam, k)]);
        for (int k = 0; k < 3; ++k) f->addVariable(variables[(size_t)getPointVID(pt, k)]);
    }
    xinit.assign(variables.size(), 0.0);
    for (long long c = 0; c < fc; ++c)
        for (int k = 0; k < 9; ++k) {
            if (!readAndSplit(in, sv) || sv.size() != 1) return false;
            if (c >= ncams_) continue;
            const double v = std::strtod(sv[0].c_str(), nullptr);
            xinit[(size_t)getCamVID(c, k)] = v;
            setDomain(getCamVID(c, k), v);
        }
    for (long long p = 0; p < fp; ++p)
        for (int k = 0; k < 3; ++k) {
            if (!readAndSplit(in, sv) || sv.size() != 1) return false;
            if (p >= npts_) continue;
            const double v = std::strtod(sv[0].c_str(), nullptr);
            xinit[(size_t)getPointVID(p, k)] = v;
            setDomain(getPointVID(p, k), v);
        }
    return !variables.empty() && !factors.empty();
}

bool BundleAdjustmentFunction::save(const std::string& file, const NumericVec* state) const {
    std::ofstream out(file.c_str());
    if (!out.is_open()) { std::cerr << "BundleAdjustmentFunction::save: cannot open " << file << std::endl; return false; }
    if (state && state->size() != variables.size()) return false;
    char buf[128];
    out << ncams_ << " " << npts_ << " " << factors.size() << "\n";
    for (const Factor* f : factors) {
        const BundleAdjustmentFactor* b = static_cast<const BundleAdjustmentFactor*>(f);
        Numeric ox, oy;
        b->getObservation(ox, oy);
        std::snprintf(buf, sizeof buf, "%lld %lld     %.17e %.17e\n", b->getCameraID(), b->getPointID(), ox, oy);
        out << buf;
    }
    for (size_t i = 0; i < variables.size(); ++i) {
        std::snprintf(buf, sizeof buf, "%.17e\n", state ? (*state)[i] : variables[i]->eval());
        out << buf;
    }
    return out.good();
}

// ------------------------------------------------------------------ PolynomialFunction
Variable* PolynomialFunction::varByName(const std::string& name, VariableID& next_id) {
    return addVariable(name, defaultDomain, next_id);
}

bool PolynomialFunction::load(const std::string& file) {
    std::ifstream in(file.c_str());
    if (!in.is_open()) { std::cerr << "PolynomialFunction::load: cannot open " << file << std::endl; return false; }
    VariableID next_id = 0;
    std::string line;
    while (std::getline(in, line)) {
        if (!line.empty() && line[line.size() - 1] == '\r') line.erase(line.size() - 1);
        if (line.empty() || line[0] == '#') continue;
        const size_t eq = line.find('=');
        if (eq != std::string::npos) {  // "<name> = <lo>:<hi>", "default" sets the default domain
            const std::string name = trim(line.substr(0, eq));
            const VariableDomain dom(trim(line.substr(eq + 1)));
            if (lower(name) == "default") defaultDomain = dom;
            else addVariable(name, dom, next_id);
            continue;
        }
        NonlinearProductFactor* f = new NonlinearProductFactor((FactorID)factors.size());
        addFactor(f);
        std::stringstream ss(line);
        std::string part;
        while (std::getline(ss, part, ',')) {  // "[coeff][, var^exp ...]"
            const size_t hat = part.find('^');
            const std::string name = lower(trim(part.substr(0, hat)));
            double num;
            if (hat == std::string::npos) {
                if (parseNumber(name, num)) { f->setCoeff(num); continue; }
                f->addVariable(varByName(name, next_id), 1.0, 0, false);
            } else {
                double e = 1.0;
                if (!parseNumber(trim(part.substr(hat + 1)), e)) throw std::invalid_argument("PolynomialFunction: bad exponent in '" + line + "'");
                f->addVariable(varByName(name, next_id), e, 0, false);
            }
        }
    }
    return !variables.empty() && !factors.empty();
}

// A complete k-ary tree of variables numbered level by level (the root is 0, node v's parent is (v - 1) / k).  For every
// allowed arity a -- 1, and the even ones up to maxArity unless odd ones are allowed -- each node at depth >= a - 1 carries
// one factor over itself and its a - 1 nearest ancestors, listed from the highest ancestor down: 12 * prod sin(x) for
// a > 1, 0.6 * x for a = 1; nodes are visited from the last to the first (what fixes the factor ids).
// Then 0.1 * x^2 per node.  Same function as src/OptimizableFunctionGenerator.cpp:660-760; rdis_amd/problems.py builds
// the packed form of it independently and the tests compare the two.
std::unique_ptr<PolynomialFunction> PolynomialFunction::makeHighDimSinusoid(VariableCount treeHeight, VariableCount branches,
        VariableCount maxArity, bool allowOddArityFactors) {
    const double period = 2.000001 * 3.141592653;
    // the generator writes the domain through a six-significant-digit text form ("%g") and reads it back
    char text[64];
    std::snprintf(text, sizeof text, "%g", 10 * period);
    const double edge = std::strtod(text, nullptr);
    std::unique_ptr<PolynomialFunction> fn(new PolynomialFunction(VariableDomain(-edge, edge, -period, period)));

    // depth of every node from its parent's (nodes of one level are consecutive)
    std::vector<VariableCount> depth;
    for (VariableCount level = 0, width = 1; level <= treeHeight; ++level, width *= branches)
        depth.insert(depth.end(), (size_t)width, level);
    const VariableCount nodes = (VariableCount)depth.size();
    VariableID next = 0;
    for (VariableCount v = 0; v < nodes; ++v) fn->addVariable("x" + std::to_string(v), fn->defaultDomain, next);

    auto newFactor = [&](Numeric coefficient) {
        NonlinearProductFactor* fac = new NonlinearProductFactor((FactorID)fn->factors.size(), coefficient);
        fn->addFactor(fac);
        return fac;
    };
    std::vector<VariableID> path;
    for (VariableCount arity = 1; arity <= std::min(maxArity, treeHeight + 1); ++arity) {
        if (arity > 1 && arity % 2 == 1 && !allowOddArityFactors) continue;
        const bool sines = arity > 1;
        for (VariableID leaf = nodes - 1; leaf >= 0; --leaf) {
            if (depth[(size_t)leaf] < arity - 1) continue;   // not enough ancestors
            path.assign(1, leaf);
            while ((VariableCount)path.size() < arity) path.push_back((path.back() - 1) / branches);
            NonlinearProductFactor* fac = newFactor(sines ? 12 : 0.6);
            for (auto it = path.rbegin(); it != path.rend(); ++it) fac->addVariable(fn->variables[(size_t)*it], 1, 0, sines);
        }
    }
    for (VariableID v = 0; v < nodes; ++v) newFactor(0.1)->addVariable(fn->variables[(size_t)v], 2, 0, false);
    return fn;
}

// ------------------------------------------------------------------ SubspaceOptimizer
SubspaceOptimizer::SubspaceOptimizer(OptimizableFunction& f_)
    : f(f_), doAscent(!f_.isMinSum()), maxiters(50), ftol(3.0e-8) {}  // src/SubspaceOptimizer.cpp:12-17

void SubspaceOptimizer::setParameters(const Options& options) {
    if (options.count("SSmaxit")) maxiters = options.as<size_t>("SSmaxit");
    if (options.count("SSftol")) ftol = options.as<Numeric>("SSftol");
    if (maxiters == 0) throw std::invalid_argument("SSmaxit must be > 0");
}

// ------------------------------------------------------------------ HipCGDSubspaceOptimizer
struct HipCGDSubspaceOptimizer::CachedPlan {
    size_t dev = 0;   // the device (OptimizableFunction::setDevices) the plan lives on
    bool busy = false;   // a shard of the call in progress uses it
    unsigned long long hash = 0, used = 0;
    int64_t bytes = 0;
    std::vector<int64_t> free_ptr, free_vid, fac_ptr, fac_id;
    rdis_hip_plan* plan = nullptr;
};

HipCGDSubspaceOptimizer::HipCGDSubspaceOptimizer(OptimizableFunction& f_)
    : SubspaceOptimizer(f_), last_iters_(0), last_status_(0), last_nfeval_(0), last_ngeval_(0),
      cache_cap_(256), cache_hits_(0), cache_misses_(0), cache_tick_(0),
      cache_byte_cap_((size_t)4 << 30), cache_bytes_(0), cache_fallbacks_(0), function_alive_(true), stamp_(0) {
    if (doAscent) throw std::invalid_argument("HipCGDSubspaceOptimizer: only the MinSum (descent) semiring is supported");
    f.plan_holders_.push_back(this);
}

HipCGDSubspaceOptimizer::~HipCGDSubspaceOptimizer() {
    dropPlans();
    if (function_alive_) {
        std::vector<HipCGDSubspaceOptimizer*>& h = f.plan_holders_;
        h.erase(std::remove(h.begin(), h.end(), this), h.end());
    }
}

void HipCGDSubspaceOptimizer::functionGone() {   // called by ~OptimizableFunction while its device problem still exists
    dropPlans();
    function_alive_ = false;
}

void HipCGDSubspaceOptimizer::dropPlans() {
    for (CachedPlan* e : cache_) { if (e->plan) rdis_hip_plan_destroy(e->plan); delete e; }
    cache_.clear();
    cache_bytes_ = 0;
}

void HipCGDSubspaceOptimizer::forget(CachedPlan* e) {
    for (size_t i = 0; i < cache_.size(); ++i)
        if (cache_[i] == e) {
            if (e->plan) rdis_hip_plan_destroy(e->plan);
            cache_bytes_ -= std::min<size_t>(cache_bytes_, (size_t)e->bytes);
            delete e;
            cache_.erase(cache_.begin() + (long)i);
            return;
        }
}

// the least recently used plan goes -- never `keep`, never one a call in progress uses.  gpu >= 0: only plans that live on
// that GPU (a shortage of device memory is not helped by dropping what other GPUs hold)
bool HipCGDSubspaceOptimizer::evictOne(const CachedPlan* keep, int gpu) {
    size_t lru = cache_.size();
    for (size_t i = 0; i < cache_.size(); ++i) {
        const CachedPlan* e = cache_[i];
        if (e == keep || e->busy || (gpu >= 0 && f.deviceOrdinal(e->dev) != gpu)) continue;
        if (lru == cache_.size() || e->used < cache_[lru]->used) lru = i;
    }
    if (lru == cache_.size()) return false;
    if (cache_[lru]->plan) rdis_hip_plan_destroy(cache_[lru]->plan);
    cache_bytes_ -= std::min<size_t>(cache_bytes_, (size_t)cache_[lru]->bytes);
    delete cache_[lru];
    cache_.erase(cache_.begin() + (long)lru);
    return true;
}

void HipCGDSubspaceOptimizer::setPlanCache(size_t entries) {
    cache_cap_ = entries;
    if (cache_.size() > cache_cap_) dropPlans();
}

void HipCGDSubspaceOptimizer::setPlanCacheBytes(size_t bytes) {
    cache_byte_cap_ = bytes;
    while (cache_bytes_ > cache_byte_cap_ && evictOne()) {}
}

void HipCGDSubspaceOptimizer::setPlanOption(const std::string& name, long long value) {
    for (auto& kv : plan_options_) if (kv.first == name) { kv.second = value; while (evictOne()) {} return; }
    plan_options_.emplace_back(name, value);
    while (evictOne()) {}   // (plans made under the old options)
}

HipCGDSubspaceOptimizer::CachedPlan* HipCGDSubspaceOptimizer::cachedPlan(const std::vector<int64_t>& free_ptr, const std::vector<int64_t>& free_vid,
                                                                         const std::vector<int64_t>& fac_ptr, const std::vector<int64_t>& fac_id, size_t dev) {
    if (cache_cap_ == 0) return nullptr;
    unsigned long long h = 1469598103934665603ull + dev;   // FNV-1a over the four lists
    auto mix = [&](const std::vector<int64_t>& v) {
        for (int64_t x : v) { h ^= (unsigned long long)x; h *= 1099511628211ull; }
        h ^= 0x9e3779b97f4a7c15ull + v.size(); h *= 1099511628211ull;
    };
    mix(free_ptr); mix(free_vid); mix(fac_ptr); mix(fac_id);
    for (CachedPlan* e : cache_)
        if (e->hash == h && e->dev == dev && e->free_vid == free_vid && e->fac_id == fac_id && e->free_ptr == free_ptr && e->fac_ptr == fac_ptr) {
            e->used = ++cache_tick_;
            ++cache_hits_;
            return e;
        }
    ++cache_misses_;
    while (cache_.size() >= cache_cap_ && evictOne()) {}
    CachedPlan* e = new CachedPlan;
    e->hash = h; e->used = ++cache_tick_; e->dev = dev;
    e->free_ptr = free_ptr; e->free_vid = free_vid; e->fac_ptr = fac_ptr; e->fac_id = fac_id;
    // Device memory is the cache's real bound: when the device is out of it the least recently used plans
    // go, one at a time, and the creation is tried again; with nothing left to drop the call is served by
    // the transient path (rdis_hip_cgd_batch, in the problem's arena) like an uncached one.
    int rc;
    for (;;) {
        rc = rdis_hip_plan_create(f.deviceProblem(dev), (int64_t)free_ptr.size() - 1, free_ptr.data(), free_vid.data(),
                                  fac_ptr.data(), fac_id.data(), &e->plan);
        if (rc != RDIS_HIP_ENOMEM || !evictOne(nullptr, f.deviceOrdinal(dev))) break;
    }
    if (rc == RDIS_HIP_ENOMEM) { delete e; ++cache_fallbacks_; return nullptr; }
    if (rc != 0) { delete e; check(f.deviceContext(dev), rc, "rdis_hip_plan_create"); }
    for (const auto& kv : plan_options_) {
        const int ro = rdis_hip_plan_set_option(e->plan, kv.first.c_str(), (int64_t)kv.second);
        if (ro != 0) { rdis_hip_plan_destroy(e->plan); delete e; check(f.deviceContext(dev), ro, "rdis_hip_plan_set_option"); }
    }
    (void)rdis_hip_plan_device_bytes(e->plan, &e->bytes);
    if ((size_t)e->bytes > cache_byte_cap_) {   // larger than the whole budget: not kept
        rdis_hip_plan_destroy(e->plan);
        delete e;
        ++cache_fallbacks_;
        return nullptr;
    }
    while (cache_bytes_ + (size_t)e->bytes > cache_byte_cap_ && evictOne()) {}
    cache_bytes_ += (size_t)e->bytes;
    cache_.push_back(e);
    return e;
}

Numeric HipCGDSubspaceOptimizer::optimize(const VariablePtrVec& vars, const FactorPtrVec& gdfs, NumericVec& xval,
                                          Numeric& deltaFval, const bool printdbg) {
    std::vector<Component> one(1);
    one[0].vars = vars; one[0].factors = gdfs; one[0].xval = xval;
    if (xval.size() != vars.size()) throw std::invalid_argument("optimize: xval.size() != vars.size()");
    optimizeBatch(one, printdbg);
    if (!gdfs.empty()) xval = one[0].xval;  // an empty factor list leaves xval as it was (.cpp:26-29)
    deltaFval = one[0].deltaFval;
    last_iters_ = one[0].iters; last_status_ = one[0].status;
    last_nfeval_ = one[0].nfeval; last_ngeval_ = one[0].ngeval;
    return one[0].fret;
}

// the components of a batch that one device solves
struct HipCGDSubspaceOptimizer::Shard {
    size_t dev = 0;
    std::vector<size_t> comps;   // indices into the batch, ascending
    std::vector<int64_t> free_ptr, free_vid, fac_ptr, fac_id;
    std::vector<double> x, fret, delta;
    std::vector<int32_t> iters, status;
    std::vector<int64_t> nfe, nge;
    CachedPlan* cp = nullptr;
    bool launched = false;       // its plan's solve has been issued and not yet fetched
};

// Every shard's plan is found (or made), then every device gets its start values and its launch -- nothing here waits
// for a device, so the launches of all devices are in flight together -- and only then are the results fetched, device
// by device.  A shard whose plan cannot be kept is served by the transient path after the others have been launched.
// Whatever goes wrong on one device (check() throws), every launch already issued on the others is waited for and no
// plan stays marked as in use: the caller may catch the error and carry on with the optimiser.
void HipCGDSubspaceOptimizer::solveShards(std::vector<Shard>& shards) {
    struct Settle {
        HipCGDSubspaceOptimizer& self;
        std::vector<Shard>& shards;
        bool completed = false;
        ~Settle() {
            for (Shard& S : shards) {
                if (!completed && S.launched) (void)rdis_hip_synchronize(self.f.deviceContext(S.dev));
                if (S.cp) S.cp->busy = false;
            }
        }
    } settle{*this, shards};
    for (Shard& S : shards) {
        S.cp = S.fac_id.empty() ? nullptr : cachedPlan(S.free_ptr, S.free_vid, S.fac_ptr, S.fac_id, S.dev);
        if (S.cp) S.cp->busy = true;   // (a later shard's plan must not push this one out)
    }
    for (Shard& S : shards) {
        if (!S.cp) continue;
        rdis_hip_ctx* ctx = f.deviceContext(S.dev);
        (void)f.deviceProblem(S.dev);   // (pending assignments of constants reach this device)
        check(ctx, rdis_hip_plan_set_start(S.cp->plan, S.x.data()), "rdis_hip_plan_set_start");
        // A solve may still allocate (tables built for the launch shape it picks).  Out of device memory: the other
        // plans on that GPU go, least recently used first, and it is tried again; with nothing left to drop this plan
        // goes too and the shard is served by the transient path like an uncached one.
        int rc;
        for (;;) {
            rc = rdis_hip_plan_solve(S.cp->plan, (int32_t)maxiters, ftol);
            if (rc != RDIS_HIP_ENOMEM || !evictOne(S.cp, f.deviceOrdinal(S.dev))) break;
        }
        if (rc == RDIS_HIP_ENOMEM) { forget(S.cp); S.cp = nullptr; ++cache_fallbacks_; continue; }
        S.launched = true;   // (also when the solve failed half way: some of its launches may be queued)
        check(ctx, rc, "rdis_hip_plan_solve");
    }
    // the batch's objective: one fp64 all-reduce between the devices' partial sums (src/RDISOptimizer.cpp:1491-1494 is where the
    // reference couples siblings), enqueued behind the solves -- over RCCL when every listed device takes part with a resident
    // plan and the devices are distinct GPUs, on the host otherwise
    last_batch_objective_ = 0; last_batch_rccl_ = false; last_batch_valid_ = false;
    {
        bool all_plans = !shards.empty();
        for (Shard& S : shards) all_plans = all_plans && S.cp != nullptr && S.launched;
        if (all_plans) {
            std::vector<rdis_hip_plan*> plans;
            std::vector<rdis_hip_comm*> cm;
            rdis_hip_comm* const* comms = shards.size() == f.numDevices() && shards.size() > 1 ? f.deviceComms() : nullptr;
            for (Shard& S : shards) { plans.push_back(S.cp->plan); if (comms) cm.push_back(comms[S.dev]); }
            double sum = 0.0;
            const int rc = rdis_hip_allreduce_objective_all((int32_t)plans.size(), plans.data(), comms ? cm.data() : nullptr, &sum);
            check(f.deviceContext(shards[0].dev), rc, "rdis_hip_allreduce_objective_all");
            last_batch_objective_ = sum; last_batch_rccl_ = comms != nullptr; last_batch_valid_ = true;
        }
    }
    for (Shard& S : shards) {
        const size_t nc = S.comps.size();
        S.fret.resize(nc); S.delta.resize(nc); S.iters.resize(nc); S.status.resize(nc); S.nfe.resize(nc); S.nge.resize(nc);
        rdis_hip_ctx* ctx = f.deviceContext(S.dev);
        if (S.cp) {
            check(ctx, rdis_hip_plan_fetch(S.cp->plan, S.x.data(), S.fret.data(), S.delta.data(), S.iters.data(), S.status.data(), S.nfe.data(), S.nge.data()),
                  "rdis_hip_plan_fetch");
            S.launched = false;
            S.cp->busy = false;
            int64_t now = S.cp->bytes;   // (what the first solve added belongs to the plan's account)
            (void)rdis_hip_plan_device_bytes(S.cp->plan, &now);
            if (now != S.cp->bytes) {
                cache_bytes_ = cache_bytes_ - std::min<size_t>(cache_bytes_, (size_t)S.cp->bytes) + (size_t)now;
                S.cp->bytes = now;
                while (cache_bytes_ > cache_byte_cap_ && evictOne(S.cp)) {}
            }
        } else {
            const int rc = rdis_hip_cgd_batch(f.deviceProblem(S.dev), (int64_t)nc, S.free_ptr.data(), S.free_vid.data(), S.fac_ptr.data(), S.fac_id.data(),
                                              S.x.data(), (int32_t)maxiters, ftol, S.fret.data(), S.delta.data(), S.iters.data(),
                                              S.status.data(), S.nfe.data(), S.nge.data());
            check(ctx, rc, "rdis_hip_cgd_batch");
        }
    }
    settle.completed = true;
}

Numeric HipCGDSubspaceOptimizer::optimizeBatch(std::vector<Component>& comps, const bool printdbg) {
    const size_t nc = comps.size();
    // every variable of every listed factor must be assigned or free in this call (the caller guarantees it,
    // src/RDISOptimizer.cpp:1042, :1049-1059); a component's free set is a mark per variable id, valid for one stamp
    if (free_stamp_.size() != (size_t)f.getNumVars()) free_stamp_.assign((size_t)f.getNumVars(), 0);
    for (size_t c = 0; c < nc; ++c) {
        Component& C = comps[c];
        if (C.xval.size() != C.vars.size()) throw std::invalid_argument("optimizeBatch: xval.size() != vars.size()");
        if (++stamp_ == 0) { std::fill(free_stamp_.begin(), free_stamp_.end(), 0u); stamp_ = 1; }
        for (const Variable* u : C.vars) free_stamp_[(size_t)u->getID()] = stamp_;
        for (const Factor* fa : C.factors)
            for (const Variable* v : fa->getVariables())
                if (!v->isAssigned() && free_stamp_[(size_t)v->getID()] != stamp_)
                    throw std::logic_error("optimize: factor " + std::to_string(fa->getID()) + " reads the unassigned variable " + v->getName());
    }
    // Which device solves which component: sibling components share no free variable and no factor (src/Component.cpp:
    // 508-549), so a batch is shared out whole components at a time -- heaviest first, always onto the device with the
    // least load so far, by factor count; ties: lower component, lower device (rdis_amd/dist.py, SURVEY 8e: the same rule
    // the ranks of a multi-process job use).  One device, or one component: everything on the primary.
    const size_t ndev = (nc > 1) ? f.numDevices() : 1;
    std::vector<Shard> shards(ndev);
    for (size_t d = 0; d < ndev; ++d) shards[d].dev = d;
    if (ndev == 1) {
        for (size_t c = 0; c < nc; ++c) shards[0].comps.push_back(c);
    } else {
        std::vector<size_t> order(nc);
        for (size_t c = 0; c < nc; ++c) order[c] = c;
        std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return comps[a].factors.size() > comps[b].factors.size(); });
        std::vector<long long> load(ndev, 0);
        for (size_t c : order) {
            size_t d = 0;
            for (size_t k = 1; k < ndev; ++k) if (load[k] < load[d]) d = k;
            shards[d].comps.push_back(c);
            load[d] += (long long)comps[c].factors.size();
        }
        for (Shard& S : shards) std::sort(S.comps.begin(), S.comps.end());
    }
    size_t used = 0;
    for (Shard& S : shards) {
        if (S.comps.empty()) continue;
        S.free_ptr.assign(1, 0); S.fac_ptr.assign(1, 0);
        for (size_t c : S.comps) {
            const Component& C = comps[c];
            for (size_t i = 0; i < C.vars.size(); ++i) { S.free_vid.push_back(C.vars[i]->getID()); S.x.push_back(C.xval[i]); }
            for (const Factor* fa : C.factors) S.fac_id.push_back(fa->getID());
            S.free_ptr.push_back((int64_t)S.free_vid.size());
            S.fac_ptr.push_back((int64_t)S.fac_id.size());
        }
        if (&S != &shards[used]) shards[used] = std::move(S);
        ++used;
    }
    shards.resize(used);
    solveShards(shards);
    // the value of the batch: the devices' sums, added in device order (RDISOptimizer.cpp:1491-1494 couples siblings only here)
    Numeric total = 0;
    for (Shard& S : shards) {
        Numeric part = 0;
        for (size_t k = 0; k < S.comps.size(); ++k) {
            const size_t c = S.comps[k];
            Component& C = comps[c];
            C.fret = S.fret[k]; C.deltaFval = S.delta[k]; C.iters = S.iters[k]; C.status = S.status[k];
            C.nfeval = S.nfe[k]; C.ngeval = S.nge[k];
            part += S.fret[k];
            if ((C.status & 0xff) == RDIS_HIP_EXIT_SYNC_TIMEOUT)   // never expected; not a result the caller may build on
                throw HipError(RDIS_HIP_EDEVICE, "HipCGD: the device-side exchange of component " + std::to_string(c) + " timed out (start restored)");
            if ((C.status & 0xff) == RDIS_HIP_EXIT_NAN) std::cerr << "HipCGD: NaN objective in component " << c << ", start restored" << std::endl;
            if ((C.status & 0xff) == RDIS_HIP_EXIT_EMPTY) continue;  // nothing touched
            // the variables are left assigned to the final, clamped values (.cpp:84-86); the device that solved the
            // component already holds them, so they are not marked for re-upload there -- the other devices get them
            // with the next call that goes to them
            for (size_t i = 0; i < C.vars.size(); ++i) {
                Variable* v = C.vars[i];
                const double val = S.x[(size_t)S.free_ptr[k] + i];
                C.xval[i] = val;
                v->assigned_ = true; v->value_ = val;
                f.onVarAssigned(v->getID(), val);
                if (f.numDevices() > 1) f.markDirtyElsewhere(v->getID(), S.dev);
            }
            if (printdbg)
                std::cout << "CGD subspace result (steps " << C.iters << "): " << C.fret << ", diff: " << C.deltaFval
                          << ", init: " << C.fret - C.deltaFval << ((C.status & RDIS_HIP_STATUS_ROLLED_BACK) ? " [restored]" : "") << std::endl;
        }
        total += part;
    }
    if (!last_batch_valid_) last_batch_objective_ = total;   // (a shard went through the transient path: no resident objective to reduce)
    return total;
}

HipLMSubspaceOptimizer::HipLMSubspaceOptimizer(OptimizableFunction& f_)
    : SubspaceOptimizer(f_), model_(1), last_iters_(0), last_stop_(0), last_nsolve_(0) {}

Numeric HipLMSubspaceOptimizer::optimize(const VariablePtrVec& vars, const FactorPtrVec& factors, NumericVec& xval,
                                         Numeric& deltaFval, const bool printdbg) {
    if (xval.size() != vars.size()) throw std::invalid_argument("optimize: xval.size() != vars.size()");
    if (factors.empty() || vars.empty()) { deltaFval = 0; return 0; }
    std::vector<int64_t> free_vid, fac_id;
    for (const Variable* v : vars) free_vid.push_back(v->getID());
    for (const Factor* fa : factors) fac_id.push_back(fa->getID());
    rdis_hip_problem* p = f.deviceProblem();
    double fret = 0, delta = 0, info[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    check(f.deviceContext(), rdis_hip_lm_optimize(p, (int64_t)free_vid.size(), free_vid.data(), (int64_t)fac_id.size(), fac_id.data(),
                                                  xval.data(), (int32_t)maxiters, ftol, (int32_t)model_, &fret, &delta, info, nullptr, 0, nullptr),
          "rdis_hip_lm_optimize");
    // the variables are left assigned to the final, clamped values (LMSubspaceOptimizer.cpp:104-108)
    for (size_t i = 0; i < vars.size(); ++i) {
        Variable* v = vars[i];
        v->assigned_ = true; v->value_ = xval[i];
        f.onVarAssigned(v->getID(), xval[i]);
    }
    deltaFval = delta;
    last_iters_ = (int)info[0]; last_stop_ = (int)info[1]; last_nsolve_ = (int)info[4];
    if (printdbg)
        std::cout << "LM SS opt returned " << fret << " (init " << fret - delta << ", diff " << delta << ") after "
                  << last_iters_ << " iterations -- termination: " << last_stop_ << ", #linsolves " << last_nsolve_ << std::endl;
    return fret;
}

std::vector<HipCGDSubspaceOptimizer::Component> HipCGDSubspaceOptimizer::createChildren() {
    const VariablePtrVec& vars = f.getVariables();
    const FactorPtrVec& facs = f.getFactors();
    std::vector<uint8_t> assigned(vars.size());
    for (size_t i = 0; i < vars.size(); ++i) assigned[i] = vars[i]->isAssigned() ? 1 : 0;
    rdis_hip_problem* p = f.deviceProblem();
    int64_t nc = 0, nfree = 0, nfac = 0;
    check(f.deviceContext(), rdis_hip_components(p, assigned.data(), &nc, &nfree, &nfac), "rdis_hip_components");
    std::vector<int64_t> free_ptr((size_t)nc + 1), fac_ptr((size_t)nc + 1), free_vid((size_t)nfree), fac_id((size_t)nfac);
    check(f.deviceContext(), rdis_hip_components_fetch(p, free_ptr.data(), free_vid.data(), fac_ptr.data(), fac_id.data()),
          "rdis_hip_components_fetch");
    std::vector<Component> out((size_t)nc);
    for (int64_t c = 0; c < nc; ++c) {
        Component& C = out[(size_t)c];
        for (int64_t i = free_ptr[(size_t)c]; i < free_ptr[(size_t)c + 1]; ++i) C.vars.push_back(vars[(size_t)free_vid[(size_t)i]]);
        for (int64_t j = fac_ptr[(size_t)c]; j < fac_ptr[(size_t)c + 1]; ++j) C.factors.push_back(facs[(size_t)fac_id[(size_t)j]]);
        C.fret = C.deltaFval = 0; C.iters = C.status = 0; C.nfeval = C.ngeval = 0;
    }
    return out;
}

}  // namespace rdis
