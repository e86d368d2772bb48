// factors_forms_test.hip -- host-side check (no GPU) of rdis_amd/csrc/factors.hpp: a line-search trial in matrix form
// (ba_camera_trial / ba_camera_trial_dir / ba_trial_value / ba_trial_slope, what solver_ptm.hpp evaluates) against the
// vector form every other solver uses (ba_forward + ba_slope_dir), on random cameras, points and directions of
// ladybug's ranges, with free and with fixed cameras and at theta = 0.  Prints the worst relative deviations;
// exit code 1 when they exceed 1e-13 (value) / 1e-11 of the slope's scale.
#include <cmath>
#include <cstdio>
#include <cstdint>
#include "../../rdis_amd/csrc/factors.hpp"
using namespace rdis_hip;

static uint64_t st = 0x9E3779B97F4A7C15ull;
static double U() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (double)(st >> 11) / 9007199254740992.0; }

int main() {
    double worst_f = 0.0, worst_s = 0.0, worst_g = 0.0;
    for (int it = 0; it < 200000; ++it) {
        double x[12], d[12];
        const bool zero_rot = it % 1000 == 999;
        const double th = zero_rot ? 0.0 : 0.05 + 1.2 * U();
        double ax[3] = {U() - 0.5, U() - 0.5, U() - 0.5};
        const double n = std::sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
        for (int k = 0; k < 3; ++k) x[k] = ax[k] / n * th;
        x[3] = 0.2 * (U() - 0.5); x[4] = 0.2 * (U() - 0.5); x[5] = -(3.0 + 2.0 * U());
        x[6] = 375.0 + 50.0 * U(); x[7] = -3e-7 * (0.5 + U()); x[8] = 5e-13 * (0.5 + U());
        for (int k = 9; k < 12; ++k) x[k] = 2.0 * (U() - 0.5);
        const double scale[12] = {1e-2, 1e-2, 1e-2, 2e-2, 2e-2, 2e-2, 2.0, 2e-8, 2e-14, 2e-2, 2e-2, 2e-2};
        const bool camfix = it % 7 == 3;
        for (int k = 0; k < 12; ++k) d[k] = (k < 9 && camfix) ? 0.0 : scale[k] * (U() - 0.5) * (it % 3 == 0 ? 100.0 : 1.0);
        const double ox = 600.0 * (U() - 0.5), oy = 600.0 * (U() - 0.5);
        // vector form
        BaFwd t;
        const double f0 = ba_forward(x, ox, oy, t);
        const double s0 = camfix ? ba_slope_dir<true>(t, x, d) : ba_slope_dir<false>(t, x, d);
        // matrix form
        BaFwd rot;
        ba_rotation(x[0], x[1], x[2], rot);
        double xc[9], dc[9], TR[CAM_TRIAL], DR[CAM_DIR];
        for (int k = 0; k < 9; ++k) { xc[k] = x[k]; dc[k] = d[k]; }
        ba_camera_trial(rot, xc, TR);
        ba_camera_trial_dir(rot, dc, DR);
        const double q[3] = {x[9], x[10], x[11]}, e[3] = {d[9], d[10], d[11]};
        BaTrial bt;
        const double f1 = ba_trial_value(TR, q, ox, oy, bt);
        const double s1 = camfix ? ba_trial_slope<true>(bt, TR, DR, q, e) : ba_trial_slope<false>(bt, TR, DR, q, e);
        // the gradient in matrix form (ba_trial_adjoint + the rotation chain once per camera) against the adjoint sweep
        {
            double g[12], gq[3], gc[CAM_GRAD], gr[3];
            ba_adjoint(t, x, t.res0, t.res1, g);
            ba_trial_adjoint(bt, TR, q, gq, gc);
            const double M[9] = {gc[0], gc[1], gc[2], gc[3], gc[4], gc[5], gc[6], gc[7], gc[8]};
            ba_rotation_gradient(rot, M, gr);
            const double m12[12] = {gr[0], gr[1], gr[2], gc[9], gc[10], gc[11], gc[12], gc[13], gc[14], gq[0], gq[1], gq[2]};
            // (measured like the device's partials against the oracle's: against the row's largest entry -- the point's entries
            // are sums of products that cancel -- and the three rotation entries, sums of nine such products, against those)
            double rs = 0.0, rowmax = 1e-300;
            for (int k = 0; k < 9; ++k) rs = std::fmax(rs, std::fabs(M[k]));
            for (int k = 0; k < 12; ++k) rowmax = std::fmax(rowmax, std::fabs(g[k]));
            for (int k = 0; k < 12; ++k) {
                const double sc = k < 3 ? std::fmax(rowmax, rs) : rowmax;
                const double dg = std::fabs(m12[k] - g[k]) / sc;
                if (dg > worst_g) worst_g = dg;
                if (!(dg <= 1e-12)) {
                    std::printf("case %d: partial %d  %.17g (matrix form) vs %.17g (adjoint), rel %.3g, theta %.3g\n", it, k, m12[k], g[k], dg, th);
                    return 1;
                }
            }
        }
        const double df = std::fabs(f1 - f0) / std::fmax(std::fabs(f0), 1e-300);
        // the slope is a sum of terms that cancel: measured against the size of its terms (|res| times the pixel speed)
        double dn = 0.0;
        for (int k = 0; k < 12; ++k) dn += std::fabs(d[k] / scale[k]);
        const double sscale = std::sqrt(2.0 * f0) * 400.0 * dn * 0.02 + 1e-300;
        const double ds = std::fabs(s1 - s0) / std::fmax(std::fabs(s0), sscale);
        if (df > worst_f) worst_f = df;
        if (ds > worst_s) worst_s = ds;
        if (!(df <= 1e-13) || !(ds <= 1e-11)) {
            std::printf("case %d: f %.17g vs %.17g (rel %.3g), slope %.17g vs %.17g (rel %.3g), theta %.3g camfix %d\n", it, f1, f0, df, s1, s0, ds, th, (int)camfix);
            return 1;
        }
    }
    std::printf("matrix form against vector form over 200000 cases: worst relative deviation of the value %.3g, of the slope %.3g, of a partial %.3g\n", worst_f, worst_s, worst_g);
    return 0;
}
