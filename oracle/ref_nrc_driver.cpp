/*
 * ref_nrc_driver.cpp -- builds the REFERENCE's own minimiser as oracle/_ref.
 *
 * TEST INFRASTRUCTURE ONLY.  The single reference source on the hot path that
 * has no Boost dependency is external/include/minimize_nrc.h (Frprmn, linmin,
 * Dbrent, bracket).  This driver includes it *where it lies* under
 * /root/reference (never copied into this repository) and exposes
 * nrc::Frprmn<T>::minimize through a C entry point so the restated minimiser
 * in rdis_oracle.c can be compared with it bit for bit on arbitrary functions.
 *
 * The header expects `rdis::Numeric` to exist (it is `typedef double Numeric`
 * in src/common.h:25, which cannot be included here because it pulls Boost);
 * that one typedef is the only thing supplied from outside the reference.
 * Everything else on the path (factors, OptimizableFunction, CGD wrapper) needs
 * Boost and is therefore unbuildable in this image -- see oracle/README.md.
 */
#include <algorithm>
#include <cmath>
#include <cstring>
#include <exception>
#include <limits>
#include <vector>

namespace rdis { typedef double Numeric; }
#include "external/include/minimize_nrc.h"

extern "C" {
typedef double (*ref_func_cb)(void *ctx, const double *x);
typedef void (*ref_grad_cb)(void *ctx, const double *x, double *g);
}

namespace {
struct Functor {
    ref_func_cb f; ref_grad_cb g; void *ctx;
    double operator()(const std::vector<double> &x) { return f(ctx, x.data()); }
    void df(const std::vector<double> &x, std::vector<double> &d) {
        d.resize(x.size());
        g(ctx, x.data(), d.data());
    }
};
}

/* returns 0 = returned normally, 3 = "Too many iterations in frprmn",
 * 4 = "Too many iterations in routine dbrent", 9 = other exception
 * (numbering follows RO_EXIT_* for the thrown cases). */
extern "C" int ref_frprmn(int n, double *x, ref_func_cb f, ref_grad_cb g, void *ctx,
                          int maxiters, double ftol, double *fret, int *iter)
{
    Functor fn{f, g, ctx};
    rdis::nrc::Frprmn<Functor> cg(fn, maxiters, ftol);
    std::vector<double> p(x, x + n);
    int rc = 0;
    try {
        cg.minimize(p);
    } catch (const char *s) {
        rc = std::strstr(s, "frprmn") ? 3 : (std::strstr(s, "dbrent") ? 4 : 9);
    } catch (const std::exception &) {
        rc = 9;
    }
    /* Frprmn::p / fret / iter are the public results the CGD wrapper reads
     * (src/optimizers/CGDSubspaceOptimizer.cpp:61-63) */
    std::copy(cg.p.begin(), cg.p.end(), x);
    *fret = cg.fret;
    *iter = cg.iter;
    return rc;
}
