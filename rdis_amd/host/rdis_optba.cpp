// rdis_optba.cpp -- include/rdis_optba.h: optBA's core over the level driver, as a C entry of librdis_host.so.
#include "../../include/rdis_optba.h"

#include <chrono>
#include <cstring>
#include <iostream>
#include <string>

#include "rdis_levels.h"

using namespace rdis;

extern "C" int rdis_optba_run(const char* bal_file, int64_t ncams, int64_t npts, int32_t schedule, int32_t nopts,
                              const char* const* opt_names, const double* opt_vals, int32_t device, double* out, double* x_out) {
    return rdis_optba_run_hist(bal_file, ncams, npts, schedule, nopts, opt_names, opt_vals, device, out, x_out, nullptr, 0);
}

extern "C" int rdis_optba_run_hist(const char* bal_file, int64_t ncams, int64_t npts, int32_t schedule, int32_t nopts,
                                   const char* const* opt_names, const double* opt_vals, int32_t device, double* out, double* x_out,
                                   double* hist, int32_t hist_rows) {
    if (hist_rows < 0 || (hist_rows > 0 && !hist)) return -3;
    if (!bal_file || !out || (schedule != 0 && schedule != 1) || nopts < 0 || (nopts > 0 && (!opt_names || !opt_vals))) return -3;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    try {
        BundleAdjustmentFunction f;
        if (!f.load(bal_file, ncams, npts)) return -1;
        f.initDevice(device);
        f.assignAll(f.getInitialState());                       // "--randinit 0" (optBA.cpp:189-193)
        Options ss, lv;
        static const char* const level_opts[] = {"AVblkpct", "steptol", "maxSweeps", "sepPiecePct", "nRRperLvl", "nRRatTop", "minRR",
                                                 "maxNAtoRR", "noAssignLimitAtTop", "restartSeed", "maxCalls", "batch"};
        for (int i = 0; i < nopts; ++i) {
            const std::string n(opt_names[i] ? opt_names[i] : "");
            if (n == "SSmaxit" || n == "SSftol") { ss.set(n, opt_vals[i]); continue; }
            bool known = false;
            for (const char* k : level_opts) known = known || n == k;
            if (!known) { std::cerr << "rdis_optba_run: unknown option '" << n << "'" << std::endl; return -3; }
            lv.set(n, opt_vals[i]);
        }
        HipCGDSubspaceOptimizer ssopt(f);
        ssopt.setParameters(ss);
        HipRDISLevelOptimizer rdis(f, ssopt);
        rdis.setParameters(lv);
        const Numeric before = f.eval();
        for (int i = 0; i < RDIS_OPTBA_NOUT; ++i) out[i] = 0.0;
        out[1] = before;
        const double t0 = now();
        if (schedule == 1) {
            out[0] = rdis.optimizeReferenceSchedule(false);
            const double t1 = now();
            out[2] = (double)rdis.refCalls(); out[3] = (double)rdis.refIterations(); out[4] = (double)rdis.refBatches();
            out[6] = rdis.decompositionMs() * 1e-3; out[5] = t1 - t0 - out[6];
            out[8] = (double)rdis.refFEvals(); out[9] = (double)rdis.refTrace().size();
            // where the calls go: per depth of the tree -- nodes, their free variables, and the steps of the reference's schedule
            // by kind (src/RDISOptimizer.cpp:1131-1133), those that made no progress beyond steptol, those that were a new minimum
            for (int r = 0; r < hist_rows; ++r) for (int k = 0; k < RDIS_OPTBA_HIST_COLS; ++k) hist[RDIS_OPTBA_HIST_COLS * r + k] = 0.0;
            for (const auto& nd : rdis.nodes())
                if (nd.depth < hist_rows) {
                    double* h = hist + RDIS_OPTBA_HIST_COLS * nd.depth;
                    h[0] += 1.0; h[1] += (double)(nd.leaf ? nd.vars.size() : nd.separator.size());
                }
            for (const auto& st : rdis.refTrace()) {
                const int d = rdis.nodes()[(size_t)st.node].depth;
                if (d >= hist_rows) continue;
                double* h = hist + RDIS_OPTBA_HIST_COLS * d;
                h[2 + st.kind] += 1.0;
                if (st.value != st.value) h[5] += 1.0;
                if (st.newMin) h[6] += 1.0;
            }
        } else {
            out[0] = rdis.optimize(false);
            const double t1 = now();
            long long calls = 0, iters = 0;
            for (const auto& st : rdis.trace()) { calls += st.ncomp; iters += st.iters; }
            out[2] = (double)calls; out[3] = (double)iters; out[4] = (double)rdis.trace().size();
            out[6] = rdis.decompositionMs() * 1e-3; out[5] = t1 - t0 - out[6];
            out[9] = rdis.sweepsDone();
        }
        out[7] = (double)rdis.nodes().size();
        if (x_out) for (size_t i = 0; i < f.getVariables().size(); ++i) x_out[i] = f.getVariables()[i]->eval();
        return 0;
    } catch (const std::exception& e) {
        std::cerr << "rdis_optba_run: " << e.what() << std::endl;
        return -2;
    }
}
