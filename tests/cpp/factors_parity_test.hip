// factors_parity_test.hip -- host-side check (no GPU): rdis_amd/csrc/factors.hpp compiled for the HOST without contraction
// (-DRDIS_FACTORS_NO_CONTRACT, what refround_kernels.hip instantiates on the device: the parity option's factor arithmetic)
// against the oracle's restatement of it in plain C (oracle/rdis_oracle.c with RO_ARITH_RECIPROCAL | RO_ARITH_SINCOS_ANGLE and
// RO_BA_DERIV_ADJOINT_DEVICE, ro_ba_factor_grad_device): value and all twelve partials, bit for bit, on random cameras,
// points and observations of ladybug's ranges and at theta = 0.  Exit code 1 on the first difference.
#define RDIS_FACTORS_NO_CONTRACT 1
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include "../../rdis_amd/csrc/factors.hpp"
#include "../../oracle/rdis_oracle.h"
using namespace rdis_hip;

static uint64_t st = 0x9E3779B97F4A7C15ull;
static double U() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (double)(st >> 11) / 9007199254740992.0; }

int main() {
    long bad = 0, n = 0;
    for (int it = 0; it < 300000; ++it) {
        double x[12];
        const bool zero_rot = it % 1000 == 999;
        const double th = zero_rot ? 0.0 : (it % 50 == 7 ? 40.0 * U() : 0.001 + 3.2 * U());
        double ax[3] = {U() - 0.5, U() - 0.5, U() - 0.5};
        const double nn = std::sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
        for (int k = 0; k < 3; ++k) x[k] = ax[k] / nn * th;
        x[3] = 0.2 * (U() - 0.5); x[4] = 0.2 * (U() - 0.5); x[5] = -(3.0 + 2.0 * U());
        x[6] = 375.0 + 50.0 * U(); x[7] = -3e-7 * (0.5 + U()); x[8] = 5e-13 * (0.5 + U());
        for (int k = 9; k < 12; ++k) x[k] = 2.0 * (U() - 0.5);
        const double ox = 600.0 * (U() - 0.5), oy = 600.0 * (U() - 0.5);
        BaFwd t;
        double g[12], go[12];
        const double f = ba_forward(x, ox, oy, t);
        ba_adjoint(t, x, t.res0, t.res1, g);
        const double fo = ro_ba_factor_grad_device(x, ox, oy, go);
        ++n;
        if (std::memcmp(&f, &fo, 8) != 0 || std::memcmp(g, go, sizeof g) != 0) {
            if (bad < 5) {
                std::printf("case %d: f %.17g / %.17g\n", it, f, fo);
                for (int k = 0; k < 12; ++k) if (std::memcmp(&g[k], &go[k], 8)) std::printf("  g[%d] %.17g / %.17g\n", k, g[k], go[k]);
            }
            ++bad;
        }
        if (th > 0.0) {   // the angle routine on its own
            double s1, c1, s2, c2;
            sincos_angle(th, &s1, &c1);
            ro_sincos_angle(th, &s2, &c2);
            if (std::memcmp(&s1, &s2, 8) || std::memcmp(&c1, &c2, 8)) { if (bad < 5) std::printf("sincos(%.17g)\n", th); ++bad; }
            // ... and against the library: below one unit in the last place
            const double es = std::fabs(s1 - std::sin(th)), ec = std::fabs(c1 - std::cos(th));
            const double us = std::ldexp(1.0, std::ilogb(std::fabs(std::sin(th))) - 52), uc = std::ldexp(1.0, std::ilogb(std::fabs(std::cos(th))) - 52);
            if (es > us || ec > uc) { if (bad < 5) std::printf("sincos(%.17g) off the library by more than an ulp\n", th); ++bad; }
        }
    }
    std::printf("%ld cases, %ld differences\n", n, bad);
    return bad ? 1 : 0;
}
