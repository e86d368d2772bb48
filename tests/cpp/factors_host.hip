// factors_host.hip -- rdis_amd/csrc/factors.hpp compiled for the HOST as a small C library (tests only): the factor arithmetic of
// the device's DEFAULT (fused multiply-add) instantiation, formed by the same front end from the same source -- `#pragma clang fp
// contract(on)` fuses a * b + c within an expression in the front end, for the host as for gfx950.  Built with
//   hipcc --offload-arch=gfx950 -O2 -std=c++17 -mfma -ffp-contract=on -fPIC -shared -o libfactors_host.so factors_host.hip
// and compared with the device factor by factor (tests/test_gpu_parity.py).
#include "../../rdis_amd/csrc/factors.hpp"
using namespace rdis_hip;

extern "C" {
double fh_value(const double* x, double ox, double oy) {
    double xx[12];
    for (int k = 0; k < 12; ++k) xx[k] = x[k];
    BaFwd t;
    return ba_forward(xx, ox, oy, t);
}
// value + twelve partials (ba_forward + ba_adjoint), and the forward-mode slope along d (ba_slope_dir<false>)
double fh_eval_grad(const double* x, double ox, double oy, double* g) {
    double xx[12], gg[12];
    for (int k = 0; k < 12; ++k) xx[k] = x[k];
    BaFwd t;
    const double E = ba_forward(xx, ox, oy, t);
    ba_adjoint(t, xx, t.res0, t.res1, gg);
    for (int k = 0; k < 12; ++k) g[k] = gg[k];
    return E;
}
double fh_value_slope(const double* x, const double* d, double ox, double oy, int camfix, double* slope) {
    double xx[12], dd[12];
    for (int k = 0; k < 12; ++k) { xx[k] = x[k]; dd[k] = d[k]; }
    BaFwd t;
    const double E = ba_forward(xx, ox, oy, t);
    *slope = camfix ? ba_slope_dir<true>(t, xx, dd) : ba_slope_dir<false>(t, xx, dd);
    return E;
}
// the matrix form of the point-major streaming solver's trials (solver_ptm.hpp): a camera's records at its nine values
// [r t f k1 k2] (and along its nine direction entries), a factor's value (and slope) against them
void fh_camera_trial(const double* xc, double* TR) {
    double x[9];
    for (int k = 0; k < 9; ++k) x[k] = xc[k];
    BaFwd rot;
    ba_rotation(x[0], x[1], x[2], rot);
    ba_camera_trial(rot, x, TR);
}
void fh_camera_trial_dir(const double* xc, const double* dc, double* DR) {
    double d[9];
    for (int k = 0; k < 9; ++k) d[k] = dc[k];
    BaFwd rot;
    ba_rotation(xc[0], xc[1], xc[2], rot);
    ba_camera_trial_dir(rot, d, DR);
}
double fh_trial(const double* TR, const double* DR, const double* q, const double* e, double ox, double oy, double* slope) {
    double tr[CAM_TRIAL], dr[CAM_DIR], qq[3], ee[3];
    for (int k = 0; k < CAM_TRIAL; ++k) tr[k] = TR[k];
    for (int k = 0; k < CAM_DIR; ++k) dr[k] = DR ? DR[k] : 0.0;
    for (int k = 0; k < 3; ++k) { qq[k] = q[k]; ee[k] = e[k]; }
    BaTrial t;
    const double v = ba_trial_value(tr, qq, ox, oy, t);
    if (slope && DR) *slope = ba_trial_slope<false>(t, tr, dr, qq, ee);
    return v;
}
// the nonlinear-product factors' sine and cosine (factors.hpp: nlp_sin / nlp_cos -- sincos_angle with the default contraction)
void fh_sincos(double x, double* sn, double* cs) { *sn = nlp_sin(x); *cs = nlp_cos(x); }
void fh_eval_grad_each(long long n, const double* x12, const double* obs2, double* f, double* g12) {
    for (long long i = 0; i < n; ++i) f[i] = fh_eval_grad(x12 + 12 * i, obs2[2 * i], obs2[2 * i + 1], g12 + 12 * i);
}
}
