/*
 * rdis_oracle.c -- CPU ORACLE (test infrastructure only; see rdis_oracle.h).
 *
 * Plain C99, no dependencies beyond libm.  Build with -O2 -ffp-contract=off so
 * the arithmetic matches an x86-64 g++ -O2 build of the reference (no FMA).
 * Every function cites the reference lines it restates (paths are relative to
 * /root/reference).
 */
#include "rdis_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ===========================================================================
 * Bundle-adjustment reprojection factor
 * =========================================================================*/

typedef struct {
    double v[3];      /* unit rotation axis (or raw r when theta == 0) */
    double theta;     /* |r| */
    double s, c;      /* sin/cos theta (theta > 0 only) */
    double w[3];      /* v x q */
    double d;         /* v . q */
    double P[3];      /* point in the camera frame */
    double pp[2];     /* -P.xy / P.z */
    double r2, dstn;  /* |pp|^2 and 1 + k1 r2 + k2 r2^2 */
    double res[2];    /* pixel residual */
    double it, iz;    /* 1 / theta and 1 / P_z (the device's form, RO_ARITH_RECIPROCAL) */
} ba_fwd;

/* Forward projection.  Snavely camera: angle-axis rotation (Rodrigues),
 * translation, perspective divide with the -z convention, two-term radial
 * distortion, focal scaling.
 * src/bundleadjust/BundleAdjustmentFactor.cpp:266-335 (rotate+translate),
 * BundleAdjustmentFactor.h:80-107 (divide, distort, error),
 * BundleAdjustmentCommon.h:81-93 (normalize). */
/* An experiment's switch (process-wide; tests/golden/make_end_values.py and nothing else): bit 0 = the unit axis and the
 * perspective divide through ONE reciprocal each (x * (1 / y) in place of x / y), the device's form (rdis_amd/csrc/
 * factors.hpp) -- an equally valid rounding, not the reference's. */
/* bit 1 (value 2) = the slope of a line-search trial summed FACTOR by factor -- sum_f (sum_k partial_fk xi_k), the association
 * the device's fused trial uses -- in place of the reference's gradient-times-direction sum over the variables (Df1dim::df).  Also
 * an experiment's switch (round 5: where ladybug 5 / 30's population of end values parts from the device's). */
static int g_experiment = 0;
void ro_set_experiment(int flags) { g_experiment = flags; }
static double (*g_slope_by_factor)(void *ctx, const double *xi) = 0;
static double (*g_slope_topology)(void *ctx, const double *xi) = 0;   /* set for the duration of a solve with RO_SUM_TOPOLOGY_COOPERATIVE */

/* RO_ARITH_SINCOS_ANGLE: sine and cosine of the rotation angle by the DEVICE's routine (rdis_amd/csrc/factors.hpp
 * sincos_angle) instead of the C library's: one Cody-Waite reduction by pi/2 in three pieces (the first product exact in a
 * fused multiply-add, the tail carried along) and the fdlibm minimax kernels on [-pi/4, pi/4].  Restated operation by operation
 * -- with -ffp-contract=off this file rounds every other product before it is added, like the device's reference-rounding
 * instantiation (refround_kernels.hip) -- so that it returns the device's bits.  Below 1 ulp like the library's, and not the
 * reference's: one of the three named last-place differences between the device's parity option and the reference
 * (DESIGN.md section 6).  x >= 0; beyond 1e6 (never a rotation angle) the library is used, as on the device. */
void ro_sincos_angle(double x, double *sn, double *cs)
{
    if (!(x < 1.0e6)) { *sn = sin(x); *cs = cos(x); return; }
    const double fn = rint(x * 6.36619772367581382433e-01);
    const int n = (int)fn;
    double t = fma(-fn, 1.57079632673412561417e+00, x);
    double w = fn * 6.07710050630396597660e-11;
    const double r = t - w;
    w = fma(fn, 2.02226624879595063154e-21, -((t - r) - w));
    const double y = r - w;
    const double yt = (r - y) - w;
    const double z = y * y, z2 = z * z;
    const double sr = 8.33333333332248946124e-03 + z * (-1.98412698298579493134e-04 + z * 2.75573137070700676789e-06)
                    + z * z2 * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10);
    const double v = z * y;
    const double ks = y - ((z * (0.5 * yt - v * sr) - yt) - v * -1.66666666666666324348e-01);
    const double cr = z * (4.16666666666666019037e-02 + z * (-1.38888888888741095749e-03 + z * 2.48015872894767294178e-05))
                    + (z2 * z2) * (-2.75573143513906633035e-07 + z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11));
    const double hz = 0.5 * z, wc = 1.0 - hz;
    const double kc = wc + (((1.0 - wc) - hz) + (z * cr - y * yt));
    const double s0 = (n & 1) ? kc : ks, c0 = (n & 1) ? ks : kc;
    *sn = (n & 2) ? -s0 : s0;
    *cs = ((n + 1) & 2) ? -c0 : c0;
}

/* arith: RO_ARITH_* flags (0 = the reference's arithmetic) */
static double ba_forward(const double x[12], double ox, double oy, ba_fwd *t, int arith)
{
    const double *r = x, *tr = x + 3, *q = x + 9;
    const double f = x[6], k1 = x[7], k2 = x[8];
    if (g_experiment & 1) arith |= RO_ARITH_RECIPROCAL;

    t->theta = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    t->it = 1.0 / t->theta;
    if (t->theta != 0.0 && (arith & RO_ARITH_RECIPROCAL)) {
        const double it = t->it;
        t->v[0] = r[0] * it; t->v[1] = r[1] * it; t->v[2] = r[2] * it;
    } else if (t->theta != 0.0) {
        t->v[0] = r[0] / t->theta; t->v[1] = r[1] / t->theta; t->v[2] = r[2] / t->theta;
    } else {
        t->v[0] = r[0]; t->v[1] = r[1]; t->v[2] = r[2];
    }
    const double *v = t->v;
    t->w[0] = v[1] * q[2] - v[2] * q[1];
    t->w[1] = v[2] * q[0] - v[0] * q[2];
    t->w[2] = v[0] * q[1] - v[1] * q[0];
    if (t->theta > 0.0) {
        if (arith & RO_ARITH_SINCOS_ANGLE) ro_sincos_angle(t->theta, &t->s, &t->c);
        else { t->c = cos(t->theta); t->s = sin(t->theta); }
        const double omc = 1 - t->c;
        t->d = v[0] * q[0] + v[1] * q[1] + v[2] * q[2];
        for (int i = 0; i < 3; ++i)
            t->P[i] = q[i] * t->c + t->w[i] * t->s + v[i] * omc * t->d;
    } else {
        /* first-order rotation near zero: q + r x q (.cpp:304-329) */
        t->c = 1.0; t->s = 0.0; t->d = 0.0;
        for (int i = 0; i < 3; ++i) t->P[i] = q[i] + t->w[i];
    }
    for (int i = 0; i < 3; ++i) t->P[i] += tr[i];

    t->iz = 1.0 / t->P[2];
    if (arith & RO_ARITH_RECIPROCAL) {
        const double iz = t->iz;
        t->pp[0] = -t->P[0] * iz;
        t->pp[1] = -t->P[1] * iz;
    } else {
        t->pp[0] = -t->P[0] / t->P[2];
        t->pp[1] = -t->P[1] / t->P[2];
    }
    t->r2 = t->pp[0] * t->pp[0] + t->pp[1] * t->pp[1];
    t->dstn = 1 + t->r2 * (k1 + k2 * t->r2);
    const double pix0 = f * t->dstn * t->pp[0];
    const double pix1 = f * t->dstn * t->pp[1];
    t->res[0] = pix0 - ox;
    t->res[1] = pix1 - oy;
    return (t->res[0] * t->res[0] + t->res[1] * t->res[1]) / 2.0;
}

/* BundleAdjustmentFactor::evalFactor(NumericVec) (.cpp:160-185) */
double ro_ba_factor_eval(const double vals[12], double obsx, double obsy)
{
    ba_fwd t;
    return ba_forward(vals, obsx, obsy, &t, 0);
}

/* Analytic gradient of one reprojection factor.  The reference
 * (BundleAdjustmentFactor.cpp:351-554) expands the chain rule forward, one
 * variable at a time; this oracle back-propagates adjoints through the same
 * forward model instead (derived from the mathematics, SURVEY.md 8a note 8).
 * Both are the exact derivative; they differ only in rounding. */
/* adjoint sweep for the seed (s0, s1): g = d(s0 pix_x + s1 pix_y)/dx at the forward state t.  With
 * the residual as seed this is grad E; with unit seeds the two rows of the residual's Jacobian. */
static void ba_adjoint(const ba_fwd *tp, const double x[12], double s0, double s1, double g[12])
{
    const ba_fwd t = *tp;
    const double *q = x + 9, *v = t.v;
    const double f = x[6], k1 = x[7], k2 = x[8];

    /* pix = f * dstn * pp ; E = |pix - obs|^2 / 2 */
    const double rp = s0 * t.pp[0] + s1 * t.pp[1];
    g[6] = t.dstn * rp;                 /* dE/df    */
    const double adst = f * rp;         /* dE/ddstn */
    g[7] = adst * t.r2;                 /* dE/dk1   */
    g[8] = adst * t.r2 * t.r2;          /* dE/dk2   */
    const double ar2 = adst * (k1 + 2.0 * k2 * t.r2);
    const double app0 = f * t.dstn * s0 + 2.0 * ar2 * t.pp[0];
    const double app1 = f * t.dstn * s1 + 2.0 * ar2 * t.pp[1];

    /* pp = -P.xy / P.z */
    double a[3];
    a[0] = -app0 / t.P[2];
    a[1] = -app1 / t.P[2];
    a[2] = -(app0 * t.pp[0] + app1 * t.pp[1]) / t.P[2];

    /* P = R q + t */
    g[3] = a[0]; g[4] = a[1]; g[5] = a[2];

    const double av = a[0] * v[0] + a[1] * v[1] + a[2] * v[2];
    const double aq = a[0] * q[0] + a[1] * q[1] + a[2] * q[2];
    /* q x a and v x a */
    const double qxa[3] = { q[1] * a[2] - q[2] * a[1], q[2] * a[0] - q[0] * a[2],
                            q[0] * a[1] - q[1] * a[0] };
    const double vxa[3] = { v[1] * a[2] - v[2] * a[1], v[2] * a[0] - v[0] * a[2],
                            v[0] * a[1] - v[1] * a[0] };
    if (t.theta > 0.0) {
        const double omc = 1 - t.c;
        /* dE/dq = R^T a (rotation by -theta about v) */
        for (int i = 0; i < 3; ++i)
            g[9 + i] = a[i] * t.c - vxa[i] * t.s + v[i] * omc * av;
        /* R q as a function of (v, theta), v = r/theta, theta = |r| */
        const double aw = a[0] * t.w[0] + a[1] * t.w[1] + a[2] * t.w[2];
        const double gth = -aq * t.s + aw * t.c + av * t.d * t.s;
        double gv[3];
        for (int i = 0; i < 3; ++i)
            gv[i] = t.s * qxa[i] + omc * (a[i] * t.d + q[i] * av);
        const double vgv = v[0] * gv[0] + v[1] * gv[1] + v[2] * gv[2];
        for (int i = 0; i < 3; ++i)
            g[i] = (gv[i] - v[i] * vgv) / t.theta + v[i] * gth;
    } else {
        /* theta == 0: P = q + r x q.  (The reference divides by theta here and
         * has no guard, BundleAdjustmentFactor.cpp:376,407-409.) */
        for (int i = 0; i < 3; ++i) {
            g[9 + i] = a[i] - vxa[i];
            g[i] = qxa[i];
        }
    }
}

static double ba_grad_adjoint(const double x[12], double ox, double oy, double g[12], int arith)
{
    ba_fwd t;
    const double E = ba_forward(x, ox, oy, &t, arith);
    ba_adjoint(&t, x, t.res[0], t.res[1], g);
    return E;
}

double ro_ba_factor_grad(const double x[12], double ox, double oy, double g[12])
{
    return ba_grad_adjoint(x, ox, oy, g, 0);
}

/* RO_BA_DERIV_ADJOINT_DEVICE: the same adjoint sweep with the DEVICE's association of every sum and product
 * (rdis_amd/csrc/factors.hpp ba_adjoint, restated statement by statement: the reciprocals 1 / P_z and 1 / theta of the forward
 * pass in place of quotients, (1 - c) (a . v) formed once, ((a . v) d - a . q) s + (a . w) c): the bits of the device's
 * reference-rounding instantiation.  Meant to be used with RO_ARITH_RECIPROCAL | RO_ARITH_SINCOS_ANGLE. */
static double ba_grad_adjoint_device(const double x[12], double ox, double oy, double g[12], int arith)
{
    ba_fwd t;
    const double E = ba_forward(x, ox, oy, &t, arith);
    const double q0 = x[9], q1 = x[10], q2 = x[11];
    const double v0 = t.v[0], v1 = t.v[1], v2 = t.v[2];
    const double f = x[6], s0 = t.res[0], s1 = t.res[1];
    const double pp0 = t.pp[0], pp1 = t.pp[1];
    const double rp = s0 * pp0 + s1 * pp1;
    g[6] = t.dstn * rp;
    const double adst = f * rp;
    g[7] = adst * t.r2;
    g[8] = adst * t.r2 * t.r2;
    const double ar2 = adst * (x[7] + 2.0 * x[8] * t.r2);
    const double fd = f * t.dstn;
    const double app0 = fd * s0 + 2.0 * ar2 * pp0;
    const double app1 = fd * s1 + 2.0 * ar2 * pp1;
    const double a0 = -app0 * t.iz, a1 = -app1 * t.iz;
    const double a2 = -(app0 * pp0 + app1 * pp1) * t.iz;
    g[3] = a0; g[4] = a1; g[5] = a2;
    const double av = a0 * v0 + a1 * v1 + a2 * v2;
    const double qxa0 = q1 * a2 - q2 * a1, qxa1 = q2 * a0 - q0 * a2, qxa2 = q0 * a1 - q1 * a0;
    const double vxa0 = v1 * a2 - v2 * a1, vxa1 = v2 * a0 - v0 * a2, vxa2 = v0 * a1 - v1 * a0;
    if (t.theta > 0.0) {
        const double omc = 1.0 - t.c;
        const double k = omc * av;
        g[9] = a0 * t.c - vxa0 * t.s + v0 * k;
        g[10] = a1 * t.c - vxa1 * t.s + v1 * k;
        g[11] = a2 * t.c - vxa2 * t.s + v2 * k;
        const double aq = a0 * q0 + a1 * q1 + a2 * q2;
        const double aw = a0 * t.w[0] + a1 * t.w[1] + a2 * t.w[2];
        const double gth = (av * t.d - aq) * t.s + aw * t.c;
        const double gv0 = t.s * qxa0 + omc * (a0 * t.d + q0 * av);
        const double gv1 = t.s * qxa1 + omc * (a1 * t.d + q1 * av);
        const double gv2 = t.s * qxa2 + omc * (a2 * t.d + q2 * av);
        const double vgv = v0 * gv0 + v1 * gv1 + v2 * gv2;
        g[0] = (gv0 - v0 * vgv) * t.it + v0 * gth;
        g[1] = (gv1 - v1 * vgv) * t.it + v1 * gth;
        g[2] = (gv2 - v2 * vgv) * t.it + v2 * gth;
    } else {
        g[9] = a0 - vxa0; g[10] = a1 - vxa1; g[11] = a2 - vxa2;
        g[0] = qxa0; g[1] = qxa1; g[2] = qxa2;
    }
    return E;
}

double ro_ba_factor_grad_device(const double x[12], double ox, double oy, double g[12])
{
    return ba_grad_adjoint_device(x, ox, oy, g, RO_ARITH_RECIPROCAL | RO_ARITH_SINCOS_ANGLE);
}

/* The reference's own derivative: the chain rule expanded forward, one variable at a time,
 * every product and quotient in the order BundleAdjustmentFactor.cpp:351-554 forms it, so that
 * an x86-64 build without FMA rounds every partial like the reference does.  This is what
 * makes the oracle's CG trajectories reproduce the reference's recorded runs bit for bit
 * (tests/test_oracle.py::test_reference_recorded_cgd_runs_*); the adjoint sweep above is an
 * independent second derivation, compared with this one on every factor of ladybug.
 *   col[0..2] = dP/d(that variable)  ->  dpp (quotient rule, .cpp:413-414)  ->  pixel  ->  E */
static double chain_to_E(const ba_fwd *t, double f, const double dp[4], const double dP[3], double P22)
{
    /* dp = { dpixx/dppx, dpixx/dppy, dpixy/dppx, dpixy/dppy } */
    const double dppx = (t->P[0] * dP[2] - t->P[2] * dP[0]) / P22;
    const double dppy = (t->P[1] * dP[2] - t->P[2] * dP[1]) / P22;
    const double drx = t->res[0] * (dp[0] * dppx + dp[1] * dppy);
    const double dry = t->res[1] * (dp[2] * dppx + dp[3] * dppy);
    return f * (drx + dry);
}

static double ba_grad_refchain(const double x[12], double ox, double oy, double g[12], int arith);
double ro_ba_factor_grad_ref(const double x[12], double ox, double oy, double g[12])
{
    return ba_grad_refchain(x, ox, oy, g, 0);
}

static double ba_grad_refchain(const double x[12], double ox, double oy, double g[12], int arith)
{
    ba_fwd t;
    const double E = ba_forward(x, ox, oy, &t, arith);
    const double *q = x + 9, *v = t.v, *P = t.P, *pp = t.pp;
    const double f = x[6], k1 = x[7], k2 = x[8];
    const double r2 = t.r2, dstn = t.dstn, vdp = t.d;
    const double *vxp = t.w;

    const double tmp1 = 2.0 * (k1 + 2.0 * k2 * r2);                       /* .cpp:365 */
    /* the reference evaluates sin/cos of theta again here (.cpp:368), also for theta == 0,
     * where the forward pass above has the branch values 0 and 1: same numbers */
    const double sinth = sin(t.theta), costh = cos(t.theta);
    const double vnorm = t.theta;                                         /* .cpp:376 */
    const double dEdpx = t.res[0], dEdpy = t.res[1];
    const double P22 = P[2] * P[2];
    const double pp00 = pp[0] * pp[0], pp01 = pp[0] * pp[1], pp11 = pp[1] * pp[1];
    const double dp[4] = { dstn + tmp1 * pp00, tmp1 * pp01, tmp1 * pp01, dstn + tmp1 * pp11 };

    /* dP/d(unit axis), dP/dtheta (.cpp:391-404) */
    const double dPdvp[3][3] = {
        { (vdp + v[0] * q[0]) * (1 - costh),  q[2] * sinth + v[0] * q[1] * (1 - costh), -q[1] * sinth + v[0] * q[2] * (1 - costh) },
        { -q[2] * sinth + v[1] * q[0] * (1 - costh), (vdp + v[1] * q[1]) * (1 - costh),  q[0] * sinth + v[1] * q[2] * (1 - costh) },
        {  q[1] * sinth + v[2] * q[0] * (1 - costh), -q[0] * sinth + v[2] * q[1] * (1 - costh), (vdp + v[2] * q[2]) * (1 - costh) } };
    const double dPdth[3] = { -q[0] * sinth + vxp[0] * costh + v[0] * vdp * sinth,
                              -q[1] * sinth + vxp[1] * costh + v[1] * vdp * sinth,
                              -q[2] * sinth + vxp[2] * costh + v[2] * vdp * sinth };
    /* d(unit axis)/d(rotation vector) (.cpp:407-409, 425-427, 443-445); dtheta/dr = v */
    const double dvpdv[3][3] = {
        { (v[1] * v[1] + v[2] * v[2]) / vnorm, -v[0] * v[1] / vnorm, -v[0] * v[2] / vnorm },
        { -v[0] * v[1] / vnorm, (v[0] * v[0] + v[2] * v[2]) / vnorm, -v[1] * v[2] / vnorm },
        { -v[0] * v[2] / vnorm, -v[1] * v[2] / vnorm, (v[0] * v[0] + v[1] * v[1]) / vnorm } };
    for (int k = 0; k < 3; ++k) {
        double dP[3];
        for (int i = 0; i < 3; ++i)
            dP[i] = dPdvp[i][0] * dvpdv[k][0] + dPdvp[i][1] * dvpdv[k][1] + dPdvp[i][2] * dvpdv[k][2] + dPdth[i] * v[k];
        g[k] = chain_to_E(&t, f, dp, dP, P22);
    }
    /* point (.cpp:482-516) */
    {
        const double dPdq[3][3] = {
            { costh * (1.0 - v[0] * v[0]) + v[0] * v[0], v[2] * sinth + v[0] * v[1] * (1.0 - costh), -v[1] * sinth + v[0] * v[2] * (1.0 - costh) },
            { -v[2] * sinth + v[0] * v[1] * (1.0 - costh), costh * (1.0 - v[1] * v[1]) + v[1] * v[1], v[0] * sinth + v[1] * v[2] * (1.0 - costh) },
            { v[1] * sinth + v[0] * v[2] * (1.0 - costh), -v[0] * sinth + v[1] * v[2] * (1.0 - costh), costh * (1.0 - v[2] * v[2]) + v[2] * v[2] } };
        for (int k = 0; k < 3; ++k) g[9 + k] = chain_to_E(&t, f, dp, dPdq[k], P22);
    }
    /* translation (.cpp:518-530) */
    g[3] = (dEdpx * dp[0] + dEdpy * dp[2]) * -f / P[2];
    g[4] = (dEdpx * dp[1] + dEdpy * dp[3]) * -f / P[2];
    {
        const double dpxdtz = dp[0] * P[0] + dp[1] * P[1];
        const double dpydtz = dp[2] * P[0] + dp[3] * P[1];
        g[5] = (dEdpx * dpxdtz + dEdpy * dpydtz) * f / P22;
    }
    /* focal length and the two distortion coefficients (.cpp:532-543) */
    g[6] = dEdpx * (dstn * pp[0]) + dEdpy * (dstn * pp[1]);
    g[7] = dEdpx * (f * r2 * pp[0]) + dEdpy * (f * r2 * pp[1]);
    g[8] = dEdpx * (f * r2 * r2 * pp[0]) + dEdpy * (f * r2 * r2 * pp[1]);
    return E;
}

/* the two pixel residuals of one factor and their Jacobian rows J[0..11] (x), J[12..23] (y) */
void ro_ba_factor_resjac(const double x[12], double ox, double oy, double res[2], double J[24])
{
    ba_fwd t;
    ba_forward(x, ox, oy, &t, 0);
    res[0] = t.res[0]; res[1] = t.res[1];
    ba_adjoint(&t, x, 1.0, 0.0, J);
    ba_adjoint(&t, x, 0.0, 1.0, J + 12);
}

/* ===========================================================================
 * Nonlinear product factor
 * =========================================================================*/

/* rdis::power, src/util/numeric.cpp:12-23 */
static double nlp_power_arith(double val, double e, int arith)
{
    if (e == 0.) return 1.;
    if (e == 1.) return val;
    if (e == 2.) return val * val;
    if (arith & RO_ARITH_POW_SMALL_INT) {   /* the device's third and fourth power (factors.hpp: nlp_power) */
        if (e == 3.) return val * val * val;
        if (e == 4.) { const double q = val * val; return q * q; }
    }
    return pow(val, e);
}

/* ===========================================================================
 * Problem container
 * =========================================================================*/

struct ro_problem {
    int kind;
    int64_t nvars, nfac;
    double *x, *lo, *hi;
    /* BA */
    int64_t *cam, *pt;
    double *obs;
    /* NLP */
    double *coeff, *expo, *cons;
    int64_t *rowptr, *vid;
    uint8_t *sine;
    uint8_t *useexp;  /* NULL, or per factor: NonlinearProductFactor::useExponential */
    /* cached factor values + variable->factor adjacency (Factor.h:228-234,
     * Variable.cpp:66-88) */
    int emulate;
    int ba_deriv;     /* RO_BA_DERIV_REFCHAIN (default), RO_BA_DERIV_ADJOINT or RO_BA_DERIV_ADJOINT_DEVICE */
    int arith;        /* RO_ARITH_* flags of the factor arithmetic (default 0: the reference's) */
    int topo;         /* RO_SUM_TOPOLOGY_*: 0 = every sum in the reference's order; 1 = the cooperative device solver's trees */
    int64_t nwave_owned, *wave_vid;   /* ... its wave-owned variables (ids), in wave order */
    int64_t *wave_of; /* [nvars] wave that owns the variable, or -1 */
    const ro_factor_arith *ext;       /* factor arithmetic supplied from outside (ro_set_factor_arithmetic), or NULL */
    int lds_nt;                       /* RO_SUM_TOPOLOGY_LDS: lanes of the workgroup */
    int stream_nwg;                   /* RO_SUM_TOPOLOGY_WG on the GRID solver (ro_set_stream_topology): its workgroups, 0 = one workgroup */
    int64_t lds_nslots, *lds_slot_vid;/* ... its slots in order: the variable behind each (cameras first, then points) */
    /* RO_SUM_TOPOLOGY_PTM (ro_set_ptm_topology): lanes of the workgroup, slots a block of slots holds, the component's camera blocks
     * and point blocks (first variable ids) in the solver's order, the trial arithmetic in matrix form; ptm: tables of the solve at hand */
    int ptm_nt, ptm_blk, ptm_K;       /* ... K workgroups share the component */
    int ptm_wide;                     /* ... as a wide group: one entry of the exchange a workgroup */
    int ptm_round_slots;              /* slots a gradient round stages: 1, or 2 (a block of slots) */
    int64_t *ptm_wg_chunk0;           /* LOCAL (ro_set_ptm_local): [K + 1] workgroup r owns the chunks [r], [r + 1]) of the order, or NULL */
    int64_t ptm_ncb, ptm_npb, *ptm_cam, *ptm_pt;
    const ro_ptm_arith *ptm_ar;
    void (*trig)(double x, double *sn, double *cs);   /* ro_set_trig: sine and cosine of the nonlinear-product factors from outside, or NULL */
    struct ptm_tab *ptm;
    int ptm_at_start;                 /* the value asked for is the rollback's (other lanes take the factors) */
    int sum_order;    /* RO_SUM_LIST (default: the reference's order) or RO_SUM_PAIRWISE */
    double *fcache;
    uint8_t *fdirty;
    int64_t *v2f_ptr, *v2f_idx;
};

static void *dup_mem(const void *src, size_t bytes)
{
    void *d = malloc(bytes ? bytes : 1);
    if (src && bytes) memcpy(d, src, bytes);
    return d;
}

static int64_t fac_arity(const ro_problem *p, int64_t f)
{
    return p->kind == RO_KIND_BA ? 12 : p->rowptr[f + 1] - p->rowptr[f];
}

static int64_t fac_var(const ro_problem *p, int64_t f, int64_t k)
{
    if (p->kind == RO_KIND_BA) return k < 9 ? p->cam[f] + k : p->pt[f] + (k - 9);
    return p->vid[p->rowptr[f] + k];
}

static void build_adjacency(ro_problem *p)
{
    p->v2f_ptr = calloc((size_t)p->nvars + 1, sizeof(int64_t));
    for (int64_t f = 0; f < p->nfac; ++f)
        for (int64_t k = 0, a = fac_arity(p, f); k < a; ++k) p->v2f_ptr[fac_var(p, f, k) + 1]++;
    for (int64_t v = 0; v < p->nvars; ++v) p->v2f_ptr[v + 1] += p->v2f_ptr[v];
    p->v2f_idx = malloc(sizeof(int64_t) * (size_t)(p->v2f_ptr[p->nvars] ? p->v2f_ptr[p->nvars] : 1));
    int64_t *fill = calloc((size_t)p->nvars + 1, sizeof(int64_t));
    for (int64_t f = 0; f < p->nfac; ++f)
        for (int64_t k = 0, a = fac_arity(p, f); k < a; ++k) {
            int64_t v = fac_var(p, f, k);
            p->v2f_idx[p->v2f_ptr[v] + fill[v]++] = f;
        }
    free(fill);
    p->fcache = calloc((size_t)p->nfac + 1, sizeof(double));
    p->fdirty = malloc((size_t)p->nfac + 1);
    memset(p->fdirty, 1, (size_t)p->nfac + 1);
}

static ro_problem *alloc_common(int kind, int64_t nvars, const double *x0,
                                const double *lo, const double *hi, int64_t nfac)
{
    ro_problem *p = calloc(1, sizeof(*p));
    p->kind = kind; p->nvars = nvars; p->nfac = nfac; p->emulate = 1;
    p->x = dup_mem(x0, sizeof(double) * (size_t)nvars);
    p->lo = dup_mem(lo, sizeof(double) * (size_t)nvars);
    p->hi = dup_mem(hi, sizeof(double) * (size_t)nvars);
    return p;
}

ro_problem *ro_create_ba(int64_t nvars, const double *x0, const double *lo,
                         const double *hi, int64_t nfac, const int64_t *cam_vid0,
                         const int64_t *pt_vid0, const double *obs)
{
    ro_problem *p = alloc_common(RO_KIND_BA, nvars, x0, lo, hi, nfac);
    p->cam = dup_mem(cam_vid0, sizeof(int64_t) * (size_t)nfac);
    p->pt = dup_mem(pt_vid0, sizeof(int64_t) * (size_t)nfac);
    p->obs = dup_mem(obs, sizeof(double) * 2 * (size_t)nfac);
    build_adjacency(p);
    return p;
}

ro_problem *ro_create_nlp(int64_t nvars, const double *x0, const double *lo,
                          const double *hi, int64_t nfac, const double *coeff,
                          const int64_t *rowptr, const int64_t *vid,
                          const double *expo, const double *cons,
                          const uint8_t *sine)
{
    ro_problem *p = alloc_common(RO_KIND_NLP, nvars, x0, lo, hi, nfac);
    const size_t nnz = (size_t)rowptr[nfac];
    p->coeff = dup_mem(coeff, sizeof(double) * (size_t)nfac);
    p->rowptr = dup_mem(rowptr, sizeof(int64_t) * ((size_t)nfac + 1));
    p->vid = dup_mem(vid, sizeof(int64_t) * nnz);
    p->expo = dup_mem(expo, sizeof(double) * nnz);
    p->cons = dup_mem(cons, sizeof(double) * nnz);
    p->sine = dup_mem(sine, nnz);
    build_adjacency(p);
    return p;
}

void ro_destroy(ro_problem *p)
{
    if (!p) return;
    free(p->x); free(p->lo); free(p->hi); free(p->cam); free(p->pt); free(p->obs);
    free(p->coeff); free(p->expo); free(p->cons); free(p->rowptr); free(p->vid);
    free(p->sine); free(p->useexp); free(p->fcache); free(p->fdirty); free(p->v2f_ptr); free(p->v2f_idx);
    free(p->wave_vid); free(p->wave_of); free(p->lds_slot_vid); free(p->ptm_cam); free(p->ptm_pt); free(p->ptm_wg_chunk0);
    free(p);
}

/* NonlinearProductFactor's constructor argument useExponential (src/NonlinearProductFactor.cpp:14-19), per
 * factor.  Only values may then be asked for: the reference's computeGradient asserts the flag off (.cpp:110),
 * and this restatement's derivative entry points return NaN for such a factor. */
int ro_nlp_set_exponential(ro_problem *p, const uint8_t *use_exp)
{
    if (!p || p->kind != RO_KIND_NLP) return -1;
    free(p->useexp);
    p->useexp = use_exp ? dup_mem(use_exp, (size_t)p->nfac) : NULL;
    if (p->fdirty) memset(p->fdirty, 1, (size_t)p->nfac);
    return 0;
}

void ro_set_ba_derivative(ro_problem *p, int which)
{
    p->ba_deriv = which;
}

void ro_set_factor_arithmetic(ro_problem *p, const ro_factor_arith *ext)
{
    p->ext = ext;
    if (p->fdirty) memset(p->fdirty, 1, (size_t)p->nfac);
}

void ro_set_arithmetic(ro_problem *p, int flags)
{
    p->arith = flags;
    if (p->fdirty) memset(p->fdirty, 1, (size_t)p->nfac);
}

static double ba_grad(const ro_problem *p, const double x[12], double ox, double oy, double g[12])
{
    if (p->ext) return p->ext->eval_grad(x, ox, oy, g);
    if (p->ba_deriv == RO_BA_DERIV_ADJOINT_DEVICE) return ba_grad_adjoint_device(x, ox, oy, g, p->arith);
    return p->ba_deriv == RO_BA_DERIV_ADJOINT ? ba_grad_adjoint(x, ox, oy, g, p->arith)
                                              : ba_grad_refchain(x, ox, oy, g, p->arith);
}

void ro_set_sum_order(ro_problem *p, int which)
{
    p->sum_order = which;
}

/* ---------------------------------------------------------------------------------------------------------------------
 * RO_SUM_TOPOLOGY_COOPERATIVE: the sums of the DEVICE's cooperative solvers (rdis_amd/csrc/solver_coop.hpp, and
 * solver_pipe.hpp, which gives the same bits), restated entry for entry -- the fourth and last named difference between the
 * benchmarked default path and the reference (the other three: RO_ARITH_*, RO_BA_DERIV_ADJOINT_DEVICE).  The device adds as
 * trees, in a fixed order:
 *   wave     64 lanes as a balanced binary tree in lane order (the DPP butterfly of solver_wg.hpp's wave_sum: every step
 *            adds a value and its mirror image, and a + b == b + a, so each lane ends with the natural tree's bits);
 *   grid     wave w of the group publishes its sum as entry w (lane j of the group is factor j, so entry w covers the listed
 *            factors 64 w .. 64 w + 63); the sweeping wave's lane l adds entries l, l + 64, l + 128, ... one after the
 *            other from 0.0, then a wave sum over the 64 lanes (grid_sync.hpp: to_wave0_n);
 *   slope    a factor's term is sum_k partial_k * direction_k over its twelve slots in slot order (from 0.0, every product
 *            rounded), the terms then added like the values;
 *   gg, dgg  lane i carries the terms of free variable i unless a wave owns it; wave w's first lane adds the terms of the
 *            variable it owns; the lanes' terms then added like the values (solver_coop.hpp: cg_reduce);
 *   gradient a variable fed by more than 48 partials is owned by a wave (the longest runs first, rdis_hip.hip:
 *            prepare_partition): lane l adds the partials l, l + 64, ... of its run in order from 0.0, then a wave sum; every
 *            other variable adds its partials in factor-list order like the reference.
 * ------------------------------------------------------------------------------------------------------------------- */
static double tree64(const double *v)
{
    double a[64];
    memcpy(a, v, sizeof a);
    for (int w = 1; w < 64; w *= 2)
        for (int i = 0; i < 64; i += 2 * w) a[i] = a[i] + a[i + w];
    return a[0];
}

/* values[0 .. count) as the group's lanes in order: waves of 64, their sums as entries, the sweep */
static double coop_tree_sum(const double *values, int64_t count)
{
    const int64_t nent = (count + 63) / 64;
    double lane[64];
    for (int l = 0; l < 64; ++l) {
        double acc = 0.0;
        for (int64_t e = l; e < nent; e += 64) {
            double v[64];
            for (int k = 0; k < 64; ++k) v[k] = (64 * e + k < count) ? values[64 * e + k] : 0.0;
            acc = acc + tree64(v);
        }
        lane[l] = acc;
    }
    return tree64(lane);
}

/* a wave-owned variable's run of partials: lane l takes l, l + 64, ... in order from 0.0; then the wave sum */
static double strided_tree_sum(const double *values, int64_t count)
{
    double lane[64];
    for (int l = 0; l < 64; ++l) {
        double s = 0.0;
        for (int64_t k = l; k < count; k += 64) s = s + values[k];
        lane[l] = s;
    }
    return tree64(lane);
}

/* ---------------------------------------------------------------------------------------------------------------------
 * RO_SUM_TOPOLOGY_LDS: the sums of the device's LDS-resident batch solver (rdis_amd/csrc/solver_lds.hpp: one workgroup of nt
 * lanes per component -- BASELINE configs 3 and 5-S), restated entry for entry:
 *   lanes    lane l adds the terms l, l + nt, l + 2 nt, ... one after the other from 0.0 (eval_partial over the listed factors,
 *            cg_reduce over the component's slots);
 *   waves    64 lanes as a balanced tree (wave_sum); the waves' sums as a balanced tree over 4 entries (up to four waves) or 16
 *            (more), zero-padded (combine_waves);
 *   gradient a camera variable: the listed factors of its camera block, in listed order, padded to whole waves of 64 -- each
 *            wave's 64 partials as a balanced tree, the waves' sums one after the other (gradient_to_xi: gperm, CG); a point
 *            variable: in factor-list order like the reference.
 * ------------------------------------------------------------------------------------------------------------------- */
static double lds_tree_sum(const double *terms, int64_t count, int nt)
{
    const int nw = nt / 64;
    double wsum[16];
    for (int w = 0; w < 16; ++w) wsum[w] = 0.0;
    for (int w = 0; w < nw; ++w) {
        double lane[64];
        for (int l = 0; l < 64; ++l) {
            double acc = 0.0;
            for (int64_t j = 64 * w + l; j < count; j += nt) acc = acc + terms[j];
            lane[l] = acc;
        }
        wsum[w] = tree64(lane);
    }
    if (nw == 1) return wsum[0];
    if (nw <= 4) return (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
    for (int w = 1; w < 16; w *= 2)
        for (int i = 0; i < 16; i += 2 * w) wsum[i] = wsum[i] + wsum[i + w];
    return wsum[0];
}

void ro_set_lds_topology(ro_problem *p, int nt, int64_t nslots, const int64_t *slot_vid)
{
    ro_set_sum_topology(p, RO_SUM_TOPOLOGY_REFERENCE, 0, NULL);
    free(p->lds_slot_vid);
    p->topo = RO_SUM_TOPOLOGY_LDS;
    p->lds_nt = nt;
    p->lds_nslots = nslots;
    p->lds_slot_vid = dup_mem(slot_vid, sizeof(int64_t) * (size_t)nslots);
}

/* ---------------------------------------------------------------------------------------------------------------------
 * RO_SUM_TOPOLOGY_PTM: the sums of the device's point-major streaming solver (rdis_amd/csrc/solver_ptm.hpp, one workgroup of nt
 * lanes per component -- BASELINE config 5-L, the strong-scaling workload's one-device case), restated entry for entry.
 *   layout   the component's point blocks in the plan's order (by number of listed factors descending, then by their cameras;
 *            whole wave-chunks of 64 dealt out over sixteen runs: rdis_hip.hip prepare_partition -- the caller passes the order);
 *            chunk ch holds blocks 64 ch .. 64 ch + 63, lane l its l-th; a chunk has as many SLOTS as its first block has listed
 *            factors, slot t of lane l is the t-th listed factor of the lane's block;
 *   trials   the chunks' slots in blocks of `blk`, the blocks in chunk order dealt to the waves in equal contiguous shares (rows:
 *            rdis_hip.hip ptm_build_segments); a lane adds its factors' terms row by row, slot by slot, from 0.0; a wave's 64 lanes
 *            as a balanced tree, the waves' sums as a balanced tree over 16 entries (zero-padded; 4 up to four waves).  Values and
 *            slopes in matrix form against per-camera records (factors.hpp: ba_camera_trial, ba_trial_value, ba_trial_slope);
 *   rollback the value at clamp(x_start): wave w takes the chunks w, w + waves, ... whole (eval_start);
 *   gradient a point variable's partials in factor-list order like the reference (the first copied).  A camera variable's: wave w
 *            takes the chunks w, w + waves, ..., slot by slot -- its sequence of steps; round r is every wave's r-th step; a
 *            camera's entry adds, from 0.0, round by round, the partials of the round's factors of that camera by wave, then lane.
 *            With two slots a round (ro_set_ptm_round_slots: where the device's LDS holds the staging rows) a step is a block of
 *            up to two slots of one chunk, and within a round the first slots come first, then the second ones;
 *   gg, dgg  a lane adds the terms of its blocks' variables as the blocks are finished (the chunks with factors first, then those
 *            without), then those of the camera slots tid, tid + nt, ... (ten slots a camera: [t f k1 k2 | r | pad]); then the
 *            trees of the trials.
 * ------------------------------------------------------------------------------------------------------------------- */
struct ptm_tab {
    int64_t nf;
    const int64_t *fac;
    int64_t npc;
    int64_t *fcam;          /* [nf] camera block (index into ptm_cam) of listed factor i */
    int64_t *pptr, *pidx;   /* block -> its listed factors (indices into the list), in listed order */
    int64_t *cp;            /* [npc + 1] a chunk's first entry: 64 entries a slot */
    int64_t *rptr, *rows;   /* wave w of workgroup r: rows rptr[r nw + w] .. rptr[r nw + w + 1]: triples (chunk, first entry, end entry) */
    int64_t *cam_of_var, *pt_of_var;   /* [nvars] block index or -1 */
};

static void ptm_free(struct ptm_tab *T)
{
    if (!T) return;
    free(T->fcam); free(T->pptr); free(T->pidx); free(T->cp); free(T->rptr); free(T->rows); free(T->cam_of_var); free(T->pt_of_var);
    free(T);
}

/* the chunks of workgroup rk's wave w: first, step, end -- chunk c is workgroup c mod K's and wave (c / K) mod waves', or, with LOCAL
 * camera numbering, the workgroup owns a contiguous range of the order and its wave w takes the w-th, w + waves-th, ... of it */
static void ptm_wave_chunks(const ro_problem *p, int64_t npc, int rk, int w, int64_t *c0, int64_t *step, int64_t *cend)
{
    const int nw = p->ptm_nt / 64, K = p->ptm_K;
    if (p->ptm_wg_chunk0) { *c0 = p->ptm_wg_chunk0[rk] + w; *step = nw; *cend = p->ptm_wg_chunk0[rk + 1]; }
    else { *c0 = rk + (int64_t)K * w; *step = (int64_t)K * nw; *cend = npc; }
}
/* ... and of the workgroup as a whole (a trial's rows are dealt from them) */
static void ptm_wg_chunks(const ro_problem *p, int64_t npc, int rk, int64_t *c0, int64_t *step, int64_t *cend)
{
    if (p->ptm_wg_chunk0) { *c0 = p->ptm_wg_chunk0[rk]; *step = 1; *cend = p->ptm_wg_chunk0[rk + 1]; }
    else { *c0 = rk; *step = p->ptm_K; *cend = npc; }
}

static struct ptm_tab *ptm_build(const ro_problem *p, int64_t nf, const int64_t *fac)
{
    struct ptm_tab *T = calloc(1, sizeof *T);
    const int64_t npb = p->ptm_npb, ncb = p->ptm_ncb, npc = (npb + 63) / 64;
    const int nw = p->ptm_nt / 64, blk = p->ptm_blk, K = p->ptm_K;
    T->nf = nf; T->fac = fac; T->npc = npc;
    T->cam_of_var = malloc(sizeof(int64_t) * (size_t)(p->nvars + 1));
    T->pt_of_var = malloc(sizeof(int64_t) * (size_t)(p->nvars + 1));
    for (int64_t v = 0; v < p->nvars; ++v) T->cam_of_var[v] = T->pt_of_var[v] = -1;
    for (int64_t c = 0; c < ncb; ++c) T->cam_of_var[p->ptm_cam[c]] = c;
    for (int64_t b = 0; b < npb; ++b) T->pt_of_var[p->ptm_pt[b]] = b;
    T->fcam = malloc(sizeof(int64_t) * (size_t)(nf + 1));
    T->pptr = calloc((size_t)npb + 2, sizeof(int64_t));
    T->pidx = malloc(sizeof(int64_t) * (size_t)(nf + 1));
    for (int64_t i = 0; i < nf; ++i) {
        const int64_t f = fac ? fac[i] : i;
        T->fcam[i] = T->cam_of_var[p->cam[f]];
        T->pptr[T->pt_of_var[p->pt[f]] + 1]++;
    }
    for (int64_t b = 0; b < npb; ++b) T->pptr[b + 1] += T->pptr[b];
    int64_t *fill = calloc((size_t)npb + 1, sizeof(int64_t));
    for (int64_t i = 0; i < nf; ++i) {
        const int64_t f = fac ? fac[i] : i, b = T->pt_of_var[p->pt[f]];
        T->pidx[T->pptr[b] + fill[b]++] = i;
    }
    free(fill);
    T->cp = calloc((size_t)npc + 2, sizeof(int64_t));
    for (int64_t ch = 0; ch < npc; ++ch) T->cp[ch + 1] = T->cp[ch] + 64 * (T->pptr[64 * ch + 1] - T->pptr[64 * ch]);
    /* the waves' rows: workgroup r's chunks r, r + K, ... cut into blocks of slots, the blocks dealt out in equal contiguous shares */
    T->rptr = calloc((size_t)K * nw + 1, sizeof(int64_t));
    T->rows = malloc(sizeof(int64_t) * 3 * (size_t)(npc + 2 * (int64_t)K * nw + 2));
    int64_t nr = 0;
    for (int rk = 0; rk < K; ++rk) {
        int64_t units = 0, wc0, wstep, wend;
        ptm_wg_chunks(p, npc, rk, &wc0, &wstep, &wend);
        for (int64_t ch = wc0; ch < wend; ch += wstep) units += ((T->cp[ch + 1] - T->cp[ch]) / 64 + blk - 1) / blk;
        int64_t u = 0, ch = wc0, done = 0;
        for (int w = 0; w < nw; ++w) {
            const int64_t end = units * (w + 1) / nw;
            T->rptr[rk * nw + w] = nr;
            while (u < end) {
                const int64_t nb = ((T->cp[ch + 1] - T->cp[ch]) / 64 + blk - 1) / blk;
                if (done >= nb) { ch += wstep; done = 0; continue; }
                const int64_t take = nb - done < end - u ? nb - done : end - u;
                const int64_t e0 = T->cp[ch] + 64 * blk * done;
                int64_t e1 = e0 + 64 * blk * take;
                if (e1 > T->cp[ch + 1]) e1 = T->cp[ch + 1];
                T->rows[3 * nr] = ch; T->rows[3 * nr + 1] = e0; T->rows[3 * nr + 2] = e1; ++nr;
                done += take; u += take;
            }
        }
    }
    T->rptr[K * nw] = nr;
    return T;
}

/* the waves' sums (entries: workgroup r's wave w at r nw + w) put together: one workgroup -- a balanced tree over its waves (4
 * entries up to four waves, else 16, zero-padded: solver_lds.hpp combine_waves); a group -- the sweep of grid_sync.hpp: lane l adds
 * the entries l, l + 64, ... from 0.0, then a wave sum over the lanes */
static double ptm_combine(const double *wsum, int nw, int K, int wide)
{
    if (K > 1) {
        /* (a wide group: a workgroup's waves first, as the 16-tree -- one entry a workgroup) */
        double *ent = NULL;
        int nent = nw * K;
        if (wide) {
            ent = malloc(sizeof(double) * (size_t)K);
            for (int r = 0; r < K; ++r) {
                double t[16];
                for (int w = 0; w < 16; ++w) t[w] = w < nw ? wsum[r * nw + w] : 0.0;
                for (int w = 1; w < 16; w *= 2)
                    for (int i = 0; i < 16; i += 2 * w) t[i] = t[i] + t[i + w];
                ent[r] = t[0];
            }
            wsum = ent; nent = K;
        }
        double lane[64];
        for (int l = 0; l < 64; ++l) {
            double acc = 0.0;
            for (int e = l; e < nent; e += 64) acc = acc + wsum[e];
            lane[l] = acc;
        }
        free(ent);
        return tree64(lane);
    }
    double t[16];
    for (int w = 0; w < 16; ++w) t[w] = w < nw ? wsum[w] : 0.0;
    if (nw == 1) return t[0];
    if (nw <= 4) return (t[0] + t[1]) + (t[2] + t[3]);
    for (int w = 1; w < 16; w *= 2)
        for (int i = 0; i < 16; i += 2 * w) t[i] = t[i] + t[i + w];
    return t[0];
}

/* the component's value (and slope along dir, a dense vector by variable id, or NULL) at the assigned point */
static double ptm_eval(ro_problem *p, const double *dir, double *slope_out)
{
    const struct ptm_tab *T = p->ptm;
    const ro_ptm_arith *ar = p->ptm_ar;
    const int64_t ncb = p->ptm_ncb, npb = p->ptm_npb;
    const int nw = p->ptm_nt / 64, K = p->ptm_K;
    double *TR = malloc(sizeof(double) * 16 * (size_t)(ncb + 1)), *DR = calloc(10 * (size_t)(ncb + 1), sizeof(double));
    for (int64_t c = 0; c < ncb; ++c) {
        const double *xc = p->x + p->ptm_cam[c];
        ar->camera_trial(xc, TR + 16 * c);
        if (dir) ar->camera_trial_dir(xc, dir + p->ptm_cam[c], DR + 10 * c);
    }
    double *wf = calloc((size_t)K * nw + 16, sizeof(double)), *ws = calloc((size_t)K * nw + 16, sizeof(double));
    const double zero3[3] = {0.0, 0.0, 0.0};
    for (int rk = 0; rk < K; ++rk)
    for (int w = 0; w < nw; ++w) {
        double af[64], as[64];
        for (int l = 0; l < 64; ++l) af[l] = as[l] = 0.0;
        if (!p->ptm_at_start) {
            for (int64_t r = T->rptr[rk * nw + w]; r < T->rptr[rk * nw + w + 1]; ++r) {
                const int64_t ch = T->rows[3 * r], t0 = (T->rows[3 * r + 1] - T->cp[ch]) / 64, t1 = (T->rows[3 * r + 2] - T->cp[ch]) / 64;
                for (int l = 0; l < 64; ++l) {
                    const int64_t b = 64 * ch + l;
                    if (b >= npb) continue;
                    const double *q = p->x + p->ptm_pt[b], *e = dir ? dir + p->ptm_pt[b] : zero3;
                    const int64_t deg = T->pptr[b + 1] - T->pptr[b];
                    for (int64_t t = t0; t < t1 && t < deg; ++t) {
                        const int64_t i = T->pidx[T->pptr[b] + t], f = T->fac ? T->fac[i] : i, c = T->fcam[i];
                        double sl = 0.0;
                        af[l] = af[l] + ar->trial(TR + 16 * c, dir ? DR + 10 * c : NULL, q, e, p->obs[2 * f], p->obs[2 * f + 1], dir ? &sl : NULL);
                        if (dir) as[l] = as[l] + sl;
                    }
                }
            }
        } else {
            int64_t c0, cstep, cend;
            ptm_wave_chunks(p, T->npc, rk, w, &c0, &cstep, &cend);
            for (int64_t ch = c0; ch < cend; ch += cstep)
                for (int l = 0; l < 64; ++l) {
                    const int64_t b = 64 * ch + l;
                    if (b >= npb) continue;
                    const double *q = p->x + p->ptm_pt[b];
                    for (int64_t k = T->pptr[b]; k < T->pptr[b + 1]; ++k) {
                        const int64_t i = T->pidx[k], f = T->fac ? T->fac[i] : i, c = T->fcam[i];
                        af[l] = af[l] + ar->trial(TR + 16 * c, NULL, q, zero3, p->obs[2 * f], p->obs[2 * f + 1], NULL);
                    }
                }
        }
        wf[rk * nw + w] = tree64(af); ws[rk * nw + w] = tree64(as);
    }
    free(TR); free(DR);
    if (slope_out) *slope_out = ptm_combine(ws, nw, K, p->ptm_wide);
    const double r = ptm_combine(wf, nw, K, p->ptm_wide);
    free(wf); free(ws);
    return r;
}

/* the camera variables' gradient entries the rounds' way; gq: the listed factors' twelve partials */
static void ptm_camera_gradient(const ro_problem *p, const double *gq, double *g)
{
    const struct ptm_tab *T = p->ptm;
    const int64_t ncb = p->ptm_ncb, npb = p->ptm_npb;
    const int nw = p->ptm_nt / 64, K = p->ptm_K;
    double *tot = calloc(9 * (size_t)(ncb + 1), sizeof(double)), *acc = malloc(sizeof(double) * 9 * (size_t)(ncb + 1));
    uint8_t *held = calloc((size_t)ncb + 1, 1), *seen = calloc((size_t)ncb + 1, 1);
    const int rs = p->ptm_round_slots == 2 ? 2 : 1;
    int64_t *sptr = calloc((size_t)nw + 1, sizeof(int64_t));
    int64_t *sch = malloc(sizeof(int64_t) * (size_t)(T->cp[T->npc] / 64 + 1)), *sslot = malloc(sizeof(int64_t) * (size_t)(T->cp[T->npc] / 64 + 1));
    for (int rk = 0; rk < K; ++rk) {   /* a workgroup's partial sums; the workgroups' then in rank order */
        int64_t nrounds = 0, at = 0;
        for (int w = 0; w < nw; ++w) {   /* a wave's steps: (chunk, first slot) -- one slot, or a block of up to rs slots of one chunk */
            sptr[w] = at;
            int64_t c0, cstep, cend;
            ptm_wave_chunks(p, T->npc, rk, w, &c0, &cstep, &cend);
            for (int64_t ch = c0; ch < cend; ch += cstep)
                for (int64_t t = 0; t < (T->cp[ch + 1] - T->cp[ch]) / 64; t += rs) { sch[at] = ch; sslot[at] = t; ++at; }
            if (at - sptr[w] > nrounds) nrounds = at - sptr[w];
        }
        sptr[nw] = at;
        for (int64_t k = 0; k < 9 * ncb; ++k) acc[k] = 0.0;
        for (int64_t rr = 0; rr < nrounds; ++rr)
            for (int sl = 0; sl < rs; ++sl)   /* (within a camera: the round's first slots by wave and lane, then its second slots) */
            for (int w = 0; w < nw; ++w) {
                if (rr >= sptr[w + 1] - sptr[w]) continue;
                const int64_t ch = sch[sptr[w] + rr], t = sslot[sptr[w] + rr] + sl;
                if (t >= (T->cp[ch + 1] - T->cp[ch]) / 64) continue;
                for (int l = 0; l < 64; ++l) {
                    const int64_t b = 64 * ch + l;
                    if (b >= npb || t >= T->pptr[b + 1] - T->pptr[b]) continue;
                    const int64_t i = T->pidx[T->pptr[b] + t], c = T->fcam[i];
                    /* (within a round a camera's rows stand by wave, then lane: this loop's order, camera by camera) */
                    for (int k = 0; k < 9; ++k) acc[9 * c + k] = acc[9 * c + k] + gq[12 * i + k];
                }
            }
        /* the workgroups' partial sums in rank order; LOCAL: only the workgroups that HOLD the camera (whose chunks meet it) take part,
         * the first one's copied */
        if (!p->ptm_wg_chunk0) { for (int64_t k = 0; k < 9 * ncb; ++k) tot[k] = rk == 0 ? acc[k] : tot[k] + acc[k]; }
        else {
            memset(held, 0, (size_t)ncb + 1);
            int64_t wc0, wstep, wend;
            ptm_wg_chunks(p, T->npc, rk, &wc0, &wstep, &wend);
            for (int64_t b = 64 * wc0; b < 64 * wend && b < npb; ++b)
                for (int64_t k = T->pptr[b]; k < T->pptr[b + 1]; ++k) held[T->fcam[T->pidx[k]]] = 1;
            for (int64_t c = 0; c < ncb; ++c) {
                if (!held[c]) continue;
                for (int k = 0; k < 9; ++k) tot[9 * c + k] = seen[c] ? tot[9 * c + k] + acc[9 * c + k] : acc[9 * c + k];
                seen[c] = 1;
            }
        }
    }
    for (int64_t c = 0; c < ncb; ++c)
        for (int k = 0; k < 9; ++k) g[p->ptm_cam[c] + k] = tot[9 * c + k];
    free(acc); free(tot); free(sptr); free(sch); free(sslot); free(held); free(seen);
}

/* LOCAL camera numbering (a wide group whose component has more cameras than a compute unit's LDS holds): workgroup r owns the chunks
 * [wg_chunk0[r], wg_chunk0[r + 1]) of the order and keeps only the cameras they meet; a camera's partial sums come from the
 * workgroups that hold it, in rank order; its terms of gg / dgg from the first of them.  After ro_set_ptm_topology (K < 0). */
void ro_set_ptm_local(ro_problem *p, const int64_t *wg_chunk0)
{
    free(p->ptm_wg_chunk0);
    p->ptm_wg_chunk0 = wg_chunk0 ? dup_mem(wg_chunk0, sizeof(int64_t) * (size_t)(p->ptm_K + 1)) : NULL;
}

void ro_set_ptm_round_slots(ro_problem *p, int round_slots)
{
    p->ptm_round_slots = round_slots;
}

void ro_set_ptm_topology(ro_problem *p, int nt, int blk, int K, int64_t ncb, const int64_t *cam_vid0, int64_t npb, const int64_t *pt_vid0,
                         const ro_ptm_arith *ar)
{
    p->ptm_round_slots = 1;
    free(p->ptm_wg_chunk0); p->ptm_wg_chunk0 = NULL;
    p->ptm_wide = K < 0;   /* (K < 0: -K workgroups as a wide group) */
    K = K < 0 ? -K : K;
    p->ptm_K = K < 1 ? 1 : K;
    ro_set_sum_topology(p, RO_SUM_TOPOLOGY_REFERENCE, 0, NULL);
    free(p->ptm_cam); free(p->ptm_pt);
    p->topo = RO_SUM_TOPOLOGY_PTM;
    p->ptm_nt = nt; p->ptm_blk = blk; p->ptm_ncb = ncb; p->ptm_npb = npb; p->ptm_ar = ar;
    p->ptm_cam = dup_mem(cam_vid0, sizeof(int64_t) * (size_t)ncb);
    p->ptm_pt = dup_mem(pt_vid0, sizeof(int64_t) * (size_t)npb);
}

/* ---------------------------------------------------------------------------------------------------------------------
 * RO_SUM_TOPOLOGY_WG: the sums of the device's plain one-workgroup solver (rdis_amd/csrc/solver_wg.hpp: cgd_wg_kernel, nt lanes --
 * where BASELINE configs 1 and 2, the nonlinear-product functions, run), restated entry for entry:
 *   values   lane l adds the listed factors l, l + nt, ... from 0.0; waves and workgroup as trees (lds_tree_sum);
 *   slope    per factor sum_k (d_k c) dir_k over its variables in row order, each step one fused multiply-add, from 0.0; the
 *            factors' terms added like the values;
 *   gradient a variable's partials in factor-list order like the reference, the first copied -- unless more than 64 listed
 *            partials feed it: then strided over a wave (lane l: l, l + 64, ... from 0.0; a wave sum);
 *   gg, dgg  lane l adds the terms of the free variables l, l + nt, ...; the trees of the values.
 * Nonlinear-product problems; bundle adjustment (the fallback of components no other solver takes) with the factor arithmetic from
 * outside (ro_set_factor_arithmetic): the slope per factor in forward mode, as in RO_SUM_TOPOLOGY_LDS. */
/* the grid solver's sums (solver_stream.hpp: nwg workgroups of nt lanes on ONE component): lane l of the grid adds the terms l,
 * l + lanes, ... from 0.0; every wave of 64 as a balanced tree -- an entry of the exchange; the sweep: lane l adds the entries l,
 * l + 64, ... from 0.0, then a wave sum (grid_sync.hpp) */
static double stream_tree_sum(const double *terms, int64_t count, int nt, int nwg)
{
    const int64_t lanes = (int64_t)nt * nwg, nent = lanes / 64;
    double *ent = malloc(sizeof(double) * (size_t)(nent + 1));
    for (int64_t e = 0; e < nent; ++e) {
        double lane[64];
        for (int l = 0; l < 64; ++l) {
            double acc = 0.0;
            for (int64_t j = 64 * e + l; j < count; j += lanes) acc = acc + terms[j];
            lane[l] = acc;
        }
        ent[e] = tree64(lane);
    }
    double sw[64];
    for (int l = 0; l < 64; ++l) {
        double acc = 0.0;
        for (int64_t e = l; e < nent; e += 64) acc = acc + ent[e];
        sw[l] = acc;
    }
    free(ent);
    return tree64(sw);
}
/* one workgroup (solver_wg.hpp) or the grid (solver_stream.hpp) */
static double wg_tree_sum(const ro_problem *p, const double *terms, int64_t count)
{
    return p->stream_nwg > 0 ? stream_tree_sum(terms, count, p->lds_nt, p->stream_nwg) : lds_tree_sum(terms, count, p->lds_nt);
}

/* ... the same solver family on a grid of nwg workgroups of nt lanes (solver_stream.hpp: one component too large for the
 * register-resident cooperative solver that the point-major solver does not take) */
void ro_set_stream_topology(ro_problem *p, int nt, int nwg)
{
    ro_set_wg_topology(p, nt);
    p->stream_nwg = nwg;
}

void ro_set_wg_topology(ro_problem *p, int nt)
{
    p->stream_nwg = 0;
    ro_set_sum_topology(p, RO_SUM_TOPOLOGY_REFERENCE, 0, NULL);
    p->topo = RO_SUM_TOPOLOGY_WG;
    p->lds_nt = nt;
}

/* ---------------------------------------------------------------------------------------------------------------------
 * RO_SUM_TOPOLOGY_GROUP: the sums of the device's solver of tiny components (rdis_amd/csrc/solver_quad.hpp: G = 4 or 16 lanes a
 * component -- a point against constant cameras: thousands of them a launch), restated entry for entry:
 *   values   lane l of the G adds the listed factors l, l + G, ... from 0.0; the G lanes as a balanced tree in lane order;
 *   slope    per factor sum_k partial_k direction_k over its twelve slots, each step one fused multiply-add, from 0.0 (a slot
 *            that is not free has direction 0 and leaves the sum as it is); the factors' terms added like the values;
 *   gradient a variable's partials: lane l adds those of ITS factors in list order from 0.0, the G lanes as the tree;
 *   gg, dgg  one after the other over the (at most four) variables, like the reference.
 * Bundle adjustment; the factor arithmetic from outside (ro_set_factor_arithmetic). */
static double group_tree_sum(const double *terms, int64_t count, int G)
{
    double lane[16];
    for (int l = 0; l < G; ++l) {
        double acc = 0.0;
        for (int64_t j = l; j < count; j += G) acc = acc + terms[j];
        lane[l] = acc;
    }
    for (int w = 1; w < G; w *= 2)
        for (int i = 0; i < G; i += 2 * w) lane[i] = lane[i] + lane[i + w];
    return lane[0];
}

void ro_set_group_topology(ro_problem *p, int G)
{
    ro_set_sum_topology(p, RO_SUM_TOPOLOGY_REFERENCE, 0, NULL);
    p->topo = RO_SUM_TOPOLOGY_GROUP;
    p->lds_nt = G == 4 ? 4 : 16;
}

void ro_set_trig(ro_problem *p, void (*sincos_fn)(double x, double *sn, double *cs))
{
    p->trig = sincos_fn;
    if (p->fdirty) memset(p->fdirty, 1, (size_t)p->nfac);
}

void ro_set_sum_topology(ro_problem *p, int kind, int64_t nwave_owned, const int64_t *wave_vid)
{
    free(p->wave_vid); free(p->wave_of);
    p->wave_vid = NULL; p->wave_of = NULL; p->nwave_owned = 0;
    p->topo = kind;
    if (kind == RO_SUM_TOPOLOGY_COOPERATIVE) {
        p->nwave_owned = nwave_owned;
        p->wave_vid = dup_mem(wave_vid, sizeof(int64_t) * (size_t)nwave_owned);
        p->wave_of = malloc(sizeof(int64_t) * (size_t)(p->nvars + 1));
        for (int64_t v = 0; v < p->nvars; ++v) p->wave_of[v] = -1;
        for (int64_t w = 0; w < nwave_owned; ++w) p->wave_of[wave_vid[w]] = w;
    }
}

void ro_set_emulate_stale_cache(ro_problem *p, int on)
{
    p->emulate = on;
    memset(p->fdirty, 1, (size_t)p->nfac + 1);
}

/* Variable::assign for an already-assigned variable (src/Variable.cpp:66-88):
 * the value is always stored; the variable's factors are told to recompute
 * only when |new - old| >= 1e-12. */
static void assign_one(ro_problem *p, int64_t v, double val)
{
    if (!(fabs(val - p->x[v]) < 1e-12)) {
        for (int64_t k = p->v2f_ptr[v]; k < p->v2f_ptr[v + 1]; ++k) p->fdirty[p->v2f_idx[k]] = 1;
    }
    p->x[v] = val;
}

void ro_assign(ro_problem *p, int64_t nvid, const int64_t *vid, const double *val)
{
    for (int64_t i = 0; i < nvid; ++i) assign_one(p, vid ? vid[i] : i, val[i]);
}

void ro_get_x(const ro_problem *p, int64_t nvid, const int64_t *vid, double *out)
{
    for (int64_t i = 0; i < nvid; ++i) out[i] = p->x[vid ? vid[i] : i];
}

/* VariableDomain::closestVal for a single-interval domain
 * (src/VariableDomain.cpp:158-163; CGD asserts one sub-interval, CGDSubspaceOptimizer.cpp:119) */
static double closest_val(double val, double lo, double hi)
{
    if (lo <= val && val <= hi) return val;
    if (val < lo) return lo;
    return hi;
}

static void gather_ba(const ro_problem *p, int64_t f, double vals[12])
{
    const double *c = p->x + p->cam[f], *q = p->x + p->pt[f];
    for (int k = 0; k < 9; ++k) vals[k] = c[k];
    vals[9] = q[0]; vals[10] = q[1]; vals[11] = q[2];
}

static double nlp_sin(const ro_problem *p, double v)
{
    if (!p->trig) return sin(v);
    double sn, cs;
    p->trig(v, &sn, &cs);
    return sn;
}
static double nlp_cos(const ro_problem *p, double v)
{
    if (!p->trig) return cos(v);
    double sn, cs;
    p->trig(v, &sn, &cs);
    return cs;
}

/* NonlinearProductFactor::evalFactor (src/NonlinearProductFactor.cpp:186-209; the same arithmetic as
 * evalFactorNoCache, :119-145), with useExponential (:140, :204) */
static double nlp_eval(const ro_problem *p, int64_t f)
{
    double prod = 1;
    for (int64_t k = p->rowptr[f]; k < p->rowptr[f + 1]; ++k) {
        double val = p->x[p->vid[k]];
        if (p->cons[k] != 0) val -= p->cons[k];
        if (p->expo[k] != 1) val = nlp_power_arith(val, p->expo[k], p->arith);
        if (p->sine[k]) val = nlp_sin(p, val);
        prod *= val;
    }
    if (p->useexp && p->useexp[f]) prod = exp(-prod);
    return prod * p->coeff[f];
}

/* NonlinearProductFactor::getDerivative (src/NonlinearProductFactor.cpp:149-178) */
static double nlp_deriv(const ro_problem *p, int64_t f, int64_t wrt)
{
    if (p->useexp && p->useexp[f]) return NAN; /* the reference asserts here (.cpp:110) */
    double prod = 1;
    for (int64_t k = p->rowptr[f]; k < p->rowptr[f + 1]; ++k) {
        double val = p->x[p->vid[k]];
        if (p->vid[k] == wrt) {
            if (p->expo[k] == 1 && !p->sine[k]) continue; /* d/dx (x-k) = 1 */
            val -= p->cons[k];
            const double inner_e = nlp_power_arith(val, p->expo[k], p->arith);
            val = nlp_power_arith(val, p->expo[k] - 1.0, p->arith);
            val *= p->expo[k];
            if (p->sine[k]) val *= nlp_cos(p, inner_e);
            prod *= val;
        } else {
            if (p->cons[k] != 0) val -= p->cons[k];
            if (p->expo[k] != 1) val = nlp_power_arith(val, p->expo[k], p->arith);
            if (p->sine[k]) val = nlp_sin(p, val);
            prod *= val;
        }
    }
    return prod * p->coeff[f];
}

/* First-order bounds, in units of the machine epsilon, on how far two correct fp64 evaluations
 * of one factor can lie apart: what a difference between this oracle and another implementation
 * of the same formulas is measured against (parity tests; not part of the reference).
 *
 * Bundle adjustment: the camera-frame point P is a sum of O(|q|) terms; an absolute error
 * eps*T_i in P_i becomes a relative error a_i = T_i/|P_i| + T_2/|P_2| (+2 for the roundings
 * that follow) in the projection pp_i = -P_i/P_2 and in the pixel, i.e. |pix_i| a_i in the
 * residual and |res_i| |pix_i| a_i in E.  The partials are proportional to the residual, whose
 * relative error is (|pix|/|res|) a, and carry one more power of 1/P_2.
 *   vb = |E| + sum_i |res_i| |pix_i| a_i          (bound on the value, / eps)
 *   gb = 1 + a (|pix|/|res| + 2)                  (relative bound on each partial, / eps) */
static void ba_bounds(const double x[12], double ox, double oy, double *vb, double *gb)
{
    ba_fwd t;
    const double E = ba_forward(x, ox, oy, &t, 0);
    const double *q = x + 9, *tr = x + 3, *v = t.v;
    const double omc = 1 - t.c;
    double a[2], T[3];
    for (int i = 0; i < 3; ++i)
        T[i] = fabs(q[i]) + fabs(t.w[i]) + fabs(v[i] * t.d) + fabs(tr[i])
             + (fabs(v[0] * q[0]) + fabs(v[1] * q[1]) + fabs(v[2] * q[2])) * (fabs(t.s) + fabs(v[i] * omc));
    /* the pixel is pp times a polynomial in |pp|^2: up to the fifth power of pp */
    const double sens = 1.0 + 2.0 * fabs(t.r2 * (x[7] + 2.0 * x[8] * t.r2)) / fabs(t.dstn);
    for (int i = 0; i < 2; ++i) a[i] = (T[i] / fabs(t.P[i]) + T[2] / fabs(t.P[2])) * sens + 2.0;
    const double pix0 = t.res[0] + ox, pix1 = t.res[1] + oy;
    *vb = fabs(E) + fabs(t.res[0]) * fabs(pix0) * a[0] + fabs(t.res[1]) * fabs(pix1) * a[1];
    const double am = a[0] > a[1] ? a[0] : a[1];
    const double rr = hypot(t.res[0], t.res[1]);
    *gb = 1.0 + am * (hypot(pix0, pix1) / rr + 2.0);
    if (!(*vb == *vb)) *vb = INFINITY;
    if (!(*gb == *gb)) *gb = INFINITY;
}

/* Nonlinear product: each term (x-k)^e carries about e+1 roundings; under a sine the absolute
 * error of its argument u becomes |u|(e+1)/|sin u| relative.  The value bound is |f| times the
 * sum over terms (+1 per multiplication); a partial is a product of the same kind. */
static void nlp_bounds(const ro_problem *p, int64_t f, double *vb, double *gb)
{
    double rel = 1.0;
    for (int64_t k = p->rowptr[f]; k < p->rowptr[f + 1]; ++k) {
        double val = p->x[p->vid[k]];
        double r = 1.0 + p->expo[k];
        if (p->cons[k] != 0) { const double d = val - p->cons[k]; r += (fabs(val) + fabs(p->cons[k])) / fabs(d); val = d; }
        if (p->expo[k] != 1) val = nlp_power_arith(val, p->expo[k], p->arith);
        if (p->sine[k]) {
            const double sn = fabs(sin(val)), cs = fabs(cos(val));
            const double m = sn < cs ? sn : cs;   /* the derivative has the cosine */
            r = r * fabs(val) / (m > 0 ? m : DBL_MIN) + 2.0;
        }
        rel += r + 1.0;
    }
    *vb = fabs(nlp_eval(p, f)) * rel;
    *gb = rel + 2.0;
    if (!(*vb == *vb)) *vb = INFINITY;
    if (!(*gb == *gb)) *gb = INFINITY;
}

static void gather_ba(const ro_problem *p, int64_t f, double vals[12]);
static void factor_bounds(const ro_problem *p, int64_t f, double *vb, double *gb)
{
    if (p->kind == RO_KIND_BA) {
        double vals[12];
        gather_ba(p, f, vals);
        ba_bounds(vals, p->obs[2 * f], p->obs[2 * f + 1], vb, gb);
    } else {
        nlp_bounds(p, f, vb, gb);
    }
}

static double factor_value_nocache(const ro_problem *p, int64_t f)
{
    if (p->kind == RO_KIND_BA) {
        double vals[12];
        ba_fwd t;
        gather_ba(p, f, vals);
        if (p->ext) return p->ext->value(vals, p->obs[2 * f], p->obs[2 * f + 1]);
        return ba_forward(vals, p->obs[2 * f], p->obs[2 * f + 1], &t, p->arith);
    }
    return nlp_eval(p, f);
}

/* Factor::eval -> evalFactorCached (src/Factor.cpp:110-119, Factor.h:228-234) */
static double factor_value(ro_problem *p, int64_t f)
{
    if (!p->emulate) return factor_value_nocache(p, f);
    if (p->fdirty[f]) {
        p->fcache[f] = factor_value_nocache(p, f);
        p->fdirty[f] = 0;
    }
    return p->fcache[f];
}

/* OptimizableFunction::evalFactors (src/OptimizableFunction.cpp:95-135):
 * MinSum product is '+', identity 0, accumulated in list order. */
static double eval_pairwise(ro_problem *p, int64_t lo, int64_t hi, const int64_t *fac)
{
    if (hi - lo <= 64) {
        double s = 0.0;
        for (int64_t i = lo; i < hi; ++i) s = s + factor_value(p, fac ? fac[i] : i);
        return s;
    }
    const int64_t mid = lo + (hi - lo) / 2;
    const double a = eval_pairwise(p, lo, mid, fac);
    return a + eval_pairwise(p, mid, hi, fac);
}

double ro_eval_factors(ro_problem *p, int64_t nf, const int64_t *fac)
{
    /* RO_SUM_PAIRWISE: the same terms added as a tree over runs of 64 (an experiment's switch: how the rounding of the
     * objective sum -- 1e-12 relative in list order over 3e4 terms, 1e-15 as a tree, which is what a device computes --
     * moves the distribution of end values; tests/golden/make_end_values.py).  Not the reference's order. */
    if (p->sum_order == RO_SUM_PAIRWISE) return eval_pairwise(p, 0, nf, fac);
    if (p->topo == RO_SUM_TOPOLOGY_PTM && p->ptm && p->ptm->nf == nf && p->ptm->fac == fac) return ptm_eval(p, NULL, NULL);
    if (p->topo == RO_SUM_TOPOLOGY_COOPERATIVE || p->topo == RO_SUM_TOPOLOGY_LDS || p->topo == RO_SUM_TOPOLOGY_WG || p->topo == RO_SUM_TOPOLOGY_GROUP) {
        double *vals = malloc(sizeof(double) * (size_t)(nf + 1));
        for (int64_t i = 0; i < nf; ++i) vals[i] = factor_value(p, fac ? fac[i] : i);
        const double r = p->topo == RO_SUM_TOPOLOGY_GROUP ? group_tree_sum(vals, nf, p->lds_nt)
                       : p->topo == RO_SUM_TOPOLOGY_WG ? wg_tree_sum(p, vals, nf)
                       : p->topo != RO_SUM_TOPOLOGY_COOPERATIVE ? lds_tree_sum(vals, nf, p->lds_nt) : coop_tree_sum(vals, nf);
        free(vals);
        return r;
    }
    double feval = 0.0;
    for (int64_t i = 0; i < nf; ++i) feval = feval + factor_value(p, fac ? fac[i] : i);
    return feval;
}

void ro_eval_each(ro_problem *p, int64_t nf, const int64_t *fac, double *fvals)
{
    for (int64_t i = 0; i < nf; ++i) fvals[i] = factor_value_nocache(p, fac ? fac[i] : i);
}

void ro_grad_each_ba(ro_problem *p, int64_t nf, const int64_t *fac, double *g12)
{
    for (int64_t i = 0; i < nf; ++i) {
        const int64_t f = fac ? fac[i] : i;
        double vals[12];
        gather_ba(p, f, vals);
        ba_grad(p, vals, p->obs[2 * f], p->obs[2 * f + 1], g12 + 12 * i);
    }
}

/* one factor's PartialGradient as (vid, value) pairs sorted by vid
 * (BundleAdjustmentFactor.cpp:338-348, Factor.cpp:142-151; flat_map order) */
static int factor_partials(const ro_problem *p, int64_t f, int64_t *vids, double *vals)
{
    int n;
    if (p->kind == RO_KIND_BA) {
        double x[12], g[12];
        gather_ba(p, f, x);
        ba_grad(p, x, p->obs[2 * f], p->obs[2 * f + 1], g);
        for (int k = 0; k < 12; ++k) { vids[k] = fac_var(p, f, k); vals[k] = g[k]; }
        n = 12;
    } else {
        n = (int)fac_arity(p, f);
        for (int k = 0; k < n; ++k) {
            vids[k] = p->vid[p->rowptr[f] + k];
            vals[k] = nlp_deriv(p, f, vids[k]);
        }
    }
    for (int i = 1; i < n; ++i) { /* insertion sort by vid */
        int64_t kv = vids[i]; double kx = vals[i]; int j = i - 1;
        while (j >= 0 && vids[j] > kv) { vids[j + 1] = vids[j]; vals[j + 1] = vals[j]; --j; }
        vids[j + 1] = kv; vals[j + 1] = kx;
    }
    return n;
}

/* OptimizableFunction::computeGradientOfSum (src/OptimizableFunction.cpp:248-262)
 * with productGradient (src/State.h:157-210).  The first contribution to a
 * variable is copied, later ones are added, always in factor-list order. */
void ro_compute_gradient(ro_problem *p, int64_t nf, const int64_t *fac, double *g, int merge)
{
    int64_t vids[64]; double vals[64];
    int64_t maxar = 12;
    if (p->kind == RO_KIND_NLP)
        for (int64_t f = 0; f < p->nfac; ++f) if (fac_arity(p, f) > maxar) maxar = fac_arity(p, f);
    int64_t *vb = vids; double *xb = vals;
    if (maxar > 64) { vb = malloc(sizeof(int64_t) * (size_t)maxar); xb = malloc(sizeof(double) * (size_t)maxar); }

    if (!merge) {
        uint8_t *seen = calloc((size_t)p->nvars + 1, 1);
        memset(g, 0, sizeof(double) * (size_t)p->nvars);
        /* RO_SUM_TOPOLOGY_COOPERATIVE: the runs of the wave-owned variables are kept and added the device's way afterwards */
        double **run = NULL; int64_t *rlen = NULL, *rcap = NULL;
        if ((p->topo == RO_SUM_TOPOLOGY_COOPERATIVE || p->topo == RO_SUM_TOPOLOGY_WG) && p->nwave_owned > 0) {
            run = calloc((size_t)p->nwave_owned, sizeof(double *));
            rlen = calloc((size_t)p->nwave_owned, sizeof(int64_t));
            rcap = calloc((size_t)p->nwave_owned, sizeof(int64_t));
        }
        for (int64_t i = 0; i < nf; ++i) {
            const int n = factor_partials(p, fac ? fac[i] : i, vb, xb);
            for (int k = 0; k < n; ++k) {
                const int64_t w = run ? p->wave_of[vb[k]] : -1;
                if (w >= 0) {
                    if (rlen[w] == rcap[w]) { rcap[w] = rcap[w] ? 2 * rcap[w] : 1024; run[w] = realloc(run[w], sizeof(double) * (size_t)rcap[w]); }
                    run[w][rlen[w]++] = xb[k];
                    continue;
                }
                if (!seen[vb[k]]) { g[vb[k]] = xb[k]; seen[vb[k]] = 1; }
                else g[vb[k]] = g[vb[k]] + xb[k];
            }
        }
        if (p->topo == RO_SUM_TOPOLOGY_LDS && p->kind == RO_KIND_BA) {
            /* camera variables again, the LDS solver's way: per camera block its listed factors in listed order, whole waves of 64
             * (the last one padded with zeros) as balanced trees, the waves' sums one after the other */
            int64_t *cnt = calloc((size_t)p->nvars + 1, sizeof(int64_t));
            for (int64_t i = 0; i < nf; ++i) cnt[p->cam[fac ? fac[i] : i]]++;
            for (int64_t cb = 0; cb < p->nvars; ++cb) {
                if (!cnt[cb]) continue;
                const int64_t m = cnt[cb], nch = (m + 63) / 64;
                double *part = calloc((size_t)(64 * nch) * 9, sizeof(double));
                int64_t at = 0;
                for (int64_t i = 0; i < nf; ++i) {
                    const int64_t f = fac ? fac[i] : i;
                    if (p->cam[f] != cb) continue;
                    double x[12], gq[12];
                    gather_ba(p, f, x);
                    ba_grad(p, x, p->obs[2 * f], p->obs[2 * f + 1], gq);
                    for (int k = 0; k < 9; ++k) part[(size_t)(9 * at + k)] = gq[k];
                    ++at;
                }
                for (int k = 0; k < 9; ++k) {
                    double sm = 0.0;
                    for (int64_t ch = 0; ch < nch; ++ch) {
                        double v[64];
                        for (int l = 0; l < 64; ++l) v[l] = part[(size_t)(9 * (64 * ch + l) + k)];
                        const double t = tree64(v);
                        sm = ch == 0 ? t : sm + t;
                    }
                    g[cb + k] = sm;
                }
                free(part);
            }
            free(cnt);
        }
        if (p->topo == RO_SUM_TOPOLOGY_GROUP && p->kind == RO_KIND_BA) {
            /* every variable again, the group's way: lane l the partials of its factors l, l + G, ... in list order, then the tree */
            const int G = p->lds_nt;
            double *gq = malloc(sizeof(double) * 12 * (size_t)(nf + 1));
            for (int64_t i = 0; i < nf; ++i) {
                const int64_t f = fac ? fac[i] : i;
                double x[12];
                gather_ba(p, f, x);
                ba_grad(p, x, p->obs[2 * f], p->obs[2 * f + 1], gq + 12 * i);
            }
            for (int64_t v = 0; v < p->nvars; ++v) {
                if (!seen[v]) continue;
                double lane[16];
                for (int l = 0; l < G; ++l) {
                    double acc = 0.0;
                    for (int64_t i = l; i < nf; i += G) {
                        const int64_t f = fac ? fac[i] : i;
                        for (int k = 0; k < 12; ++k)
                            if (fac_var(p, f, k) == v) acc = acc + gq[12 * i + k];
                    }
                    lane[l] = acc;
                }
                for (int w = 1; w < G; w *= 2)
                    for (int i = 0; i < G; i += 2 * w) lane[i] = lane[i] + lane[i + w];
                g[v] = lane[0];
            }
            free(gq);
        }
        if (p->topo == RO_SUM_TOPOLOGY_PTM && p->ptm && p->ptm->nf == nf && p->ptm->fac == fac) {
            double *gq = malloc(sizeof(double) * 12 * (size_t)(nf + 1));
            for (int64_t i = 0; i < nf; ++i) {
                const int64_t f = fac ? fac[i] : i;
                double x[12];
                gather_ba(p, f, x);
                ba_grad(p, x, p->obs[2 * f], p->obs[2 * f + 1], gq + 12 * i);
            }
            ptm_camera_gradient(p, gq, g);
            free(gq);
        }
        if (run) {
            for (int64_t w = 0; w < p->nwave_owned; ++w) {
                if (rlen[w] > 0) g[p->wave_vid[w]] = strided_tree_sum(run[w], rlen[w]);
                free(run[w]);
            }
            free(run); free(rlen); free(rcap);
        }
        free(seen);
    } else {
        /* sorted (vid,value) vector, merged factor by factor: the reference's
         * cost model (linear walk + vector insert per factor). */
        int64_t cap = 1024, len = 0;
        int64_t *mv = malloc(sizeof(int64_t) * (size_t)cap);
        double *mx = malloc(sizeof(double) * (size_t)cap);
        for (int64_t i = 0; i < nf; ++i) {
            const int n = factor_partials(p, fac ? fac[i] : i, vb, xb);
            if (len + n > cap) {
                while (len + n > cap) cap *= 2;
                mv = realloc(mv, sizeof(int64_t) * (size_t)cap);
                mx = realloc(mx, sizeof(double) * (size_t)cap);
            }
            if (len == 0) {
                memcpy(mv, vb, sizeof(int64_t) * (size_t)n);
                memcpy(mx, xb, sizeof(double) * (size_t)n);
                len = n;
                continue;
            }
            int64_t i1 = 0; int i2 = 0;
            while (i2 < n) {
                if (i1 == len || mv[i1] > vb[i2]) {
                    memmove(mv + i1 + 1, mv + i1, sizeof(int64_t) * (size_t)(len - i1));
                    memmove(mx + i1 + 1, mx + i1, sizeof(double) * (size_t)(len - i1));
                    mv[i1] = vb[i2]; mx[i1] = xb[i2]; ++len; ++i1; ++i2;
                } else if (mv[i1] == vb[i2]) {
                    mx[i1] = mx[i1] + xb[i2]; ++i1; ++i2;
                } else {
                    ++i1;
                }
            }
        }
        memset(g, 0, sizeof(double) * (size_t)p->nvars);
        for (int64_t k = 0; k < len; ++k) g[mv[k]] = mx[k];
        free(mv); free(mx);
    }
    if (vb != vids) { free(vb); free(xb); }
}

/* ---------------------------------------------------------------------------------------------------------------------
 * The device's PUBLIC evaluation entry points on bundle adjustment (rdis_hip_eval / rdis_hip_eval_grad: rdis_amd/csrc/grad_fused.hip,
 * eval_kernels.hpp), restated entry for entry -- the batched factor / gradient evaluation of the path's boundary:
 *   value    the list in chunks of `lanes` (512) entries, a lane each: a chunk's waves of 64 as balanced trees, the waves' sums one
 *            after the other from 0.0; the chunks' sums: lane t of 256 adds the chunks t, t + 256, ... from 0.0, the four waves as
 *            trees, their sums one after the other from 0.0;
 *   gradient a point block's three entries: per chunk its listed factors' partials one after the other from 0.0 (list order),
 *            the chunks' sums one after the other from 0.0 where several chunks hold the block; a camera block's nine: per TILE (up
 *            to tile_chunks consecutive chunks, fewer where a tile would meet more than 192 cameras) one after the other from
 *            0.0 over the tile's listed factors, the tiles' sums one after the other from 0.0 where several tiles hold the camera.
 * The factor arithmetic is the one plugged in (ro_set_factor_arithmetic: factors.hpp for the host) or the built-in one. */
static double chunk_value_sum(const double *vals, int64_t count, int lanes)
{
    const int64_t nchunks = (count + lanes - 1) / lanes;
    double *part = malloc(sizeof(double) * (size_t)(nchunks + 1));
    for (int64_t ch = 0; ch < nchunks; ++ch) {
        double r = 0.0;
        for (int w = 0; w < lanes / 64; ++w) {
            double v[64];
            for (int l = 0; l < 64; ++l) { const int64_t j = ch * lanes + 64 * w + l; v[l] = j < count ? vals[j] : 0.0; }
            r = r + tree64(v);
        }
        part[ch] = r;
    }
    double red[4];
    for (int w = 0; w < 4; ++w) {
        double lane[64];
        for (int l = 0; l < 64; ++l) {
            double acc = 0.0;
            for (int64_t i = 64 * w + l; i < nchunks; i += 256) acc = acc + part[i];
            lane[l] = acc;
        }
        red[w] = tree64(lane);
    }
    free(part);
    double r = 0.0;
    for (int w = 0; w < 4; ++w) r = r + red[w];
    return r;
}

/* ... and on nonlinear products (eval_kernels.hpp: eval_sum_kernel): `blocks` workgroups of 256 lanes stride over the list -- lane t
 * of the grid adds the entries t, t + 256 blocks, ... from 0.0 --, a workgroup's four waves as trees added one after the other from
 * 0.0, the workgroups' sums like the chunks' above.  (The gradient there: per-factor partials, every variable's added in list
 * order -- the reference's order.) */
double ro_eval_device_grid(ro_problem *p, int64_t nf, const int64_t *fac, int blocks)
{
    const int64_t lanes = 256ll * blocks;
    double *part = malloc(sizeof(double) * (size_t)(blocks + 1));
    for (int b = 0; b < blocks; ++b) {
        double r = 0.0;
        for (int w = 0; w < 4; ++w) {
            double v[64];
            for (int l = 0; l < 64; ++l) {
                double acc = 0.0;
                for (int64_t j = 256ll * b + 64 * w + l; j < nf; j += lanes) acc = acc + factor_value_nocache(p, fac ? fac[j] : j);
                v[l] = acc;
            }
            r = r + tree64(v);
        }
        part[b] = r;
    }
    double red[4];
    for (int w = 0; w < 4; ++w) {
        double lane[64];
        for (int l = 0; l < 64; ++l) {
            double acc = 0.0;
            for (int64_t i = 64 * w + l; i < blocks; i += 256) acc = acc + part[i];
            lane[l] = acc;
        }
        red[w] = tree64(lane);
    }
    free(part);
    double r = 0.0;
    for (int w = 0; w < 4; ++w) r = r + red[w];
    return r;
}

double ro_eval_device_ba(ro_problem *p, int64_t nf, const int64_t *fac, int lanes)
{
    double *vals = malloc(sizeof(double) * (size_t)(nf + 1));
    for (int64_t i = 0; i < nf; ++i) vals[i] = factor_value_nocache(p, fac ? fac[i] : i);
    const double r = chunk_value_sum(vals, nf, lanes);
    free(vals);
    return r;
}

double ro_eval_grad_device_ba(ro_problem *p, int64_t nf, const int64_t *fac, double *g, int lanes, int tile_chunks)
{
    const int64_t nchunks = (nf + lanes - 1) / lanes;
    double *vals = malloc(sizeof(double) * (size_t)(nf + 1)), *gq = malloc(sizeof(double) * 12 * (size_t)(nf + 1));
    for (int64_t i = 0; i < nf; ++i) {
        const int64_t f = fac ? fac[i] : i;
        double x[12];
        gather_ba(p, f, x);
        vals[i] = ba_grad(p, x, p->obs[2 * f], p->obs[2 * f + 1], gq + 12 * i);
    }
    const double value = chunk_value_sum(vals, nf, lanes);
    memset(g, 0, sizeof(double) * (size_t)p->nvars);
    /* point blocks: chunk by chunk */
    {
        double *acc = calloc((size_t)p->nvars + 3, sizeof(double));      /* the chunk at hand's sum of a block */
        double *tot = calloc((size_t)p->nvars + 3, sizeof(double));
        int64_t *last = malloc(sizeof(int64_t) * (size_t)(p->nvars + 1)), *nparts = calloc((size_t)p->nvars + 1, sizeof(int64_t));
        for (int64_t v = 0; v < p->nvars; ++v) last[v] = -1;
        for (int64_t ch = 0; ch < nchunks; ++ch) {
            const int64_t j0 = ch * lanes, j1 = j0 + lanes < nf ? j0 + lanes : nf;
            for (int64_t j = j0; j < j1; ++j) {
                const int64_t q = p->pt[fac ? fac[j] : j];
                if (last[q] != ch) {   /* the block's first row of this chunk; what the chunk before left goes to the total first */
                    if (last[q] >= 0) { for (int k = 0; k < 3; ++k) tot[q + k] = nparts[q] == 0 ? 0.0 + acc[q + k] : tot[q + k] + acc[q + k]; nparts[q]++; }
                    last[q] = ch;
                    for (int k = 0; k < 3; ++k) acc[q + k] = 0.0;
                }
                for (int k = 0; k < 3; ++k) acc[q + k] = acc[q + k] + gq[12 * j + 9 + k];
            }
        }
        for (int64_t j = 0; j < nf; ++j) {
            const int64_t q = p->pt[fac ? fac[j] : j];
            if (last[q] < 0) continue;
            /* one chunk: its sum IS the entry; several: the chunks' sums one after the other from 0.0 */
            for (int k = 0; k < 3; ++k) g[q + k] = nparts[q] == 0 ? acc[q + k] : tot[q + k] + acc[q + k];
            last[q] = -1;
        }
        free(acc); free(tot); free(last); free(nparts);
    }
    /* camera blocks: tile by tile */
    {
        double *acc = calloc((size_t)p->nvars + 9, sizeof(double)), *tot = calloc((size_t)p->nvars + 9, sizeof(double));
        int64_t *tile_of = malloc(sizeof(int64_t) * (size_t)(p->nvars + 1)), *nparts = calloc((size_t)p->nvars + 1, sizeof(int64_t));
        for (int64_t v = 0; v < p->nvars; ++v) tile_of[v] = -1;
        int64_t tile = 0;
        for (int64_t ch0 = 0; ch0 < nchunks; ++tile) {
            /* the tile's chunks: up to tile_chunks, fewer where the cameras met would pass 192 (a chunk alone always stands) */
            int64_t ch1 = ch0, ncam = 0;
            int64_t *seen_list = malloc(sizeof(int64_t) * (size_t)(lanes * (tile_chunks + 1) + 1));
            while (ch1 < nchunks && ch1 - ch0 < tile_chunks) {
                const int64_t j0 = ch1 * lanes, j1 = j0 + lanes < nf ? j0 + lanes : nf, before = ncam;
                for (int64_t j = j0; j < j1; ++j) {
                    const int64_t c = p->cam[fac ? fac[j] : j];
                    if (tile_of[c] != tile) {
                        if (tile_of[c] >= 0 && tile_of[c] != tile) {   /* what an earlier tile left: to the total */
                            for (int k = 0; k < 9; ++k) tot[c + k] = nparts[c] == 0 ? 0.0 + acc[c + k] : tot[c + k] + acc[c + k];
                            nparts[c]++;
                        }
                        tile_of[c] = tile;
                        for (int k = 0; k < 9; ++k) acc[c + k] = 0.0;
                        seen_list[ncam++] = c;
                    }
                }
                if (ch1 > ch0 && ncam > 192) {   /* too many: this chunk starts the next tile -- undo what it added */
                    for (int64_t k = before; k < ncam; ++k) tile_of[seen_list[k]] = -2 - tile;   /* (not of this tile; its old sums were already moved) */
                    ncam = before;
                    break;
                }
                ++ch1;
            }
            for (int64_t j = ch0 * lanes; j < (ch1 * lanes < nf ? ch1 * lanes : nf); ++j) {
                const int64_t c = p->cam[fac ? fac[j] : j];
                for (int k = 0; k < 9; ++k) acc[c + k] = acc[c + k] + gq[12 * j + k];
            }
            free(seen_list);
            ch0 = ch1;
        }
        for (int64_t j = 0; j < nf; ++j) {
            const int64_t c = p->cam[fac ? fac[j] : j];
            if (tile_of[c] == -1) continue;
            for (int k = 0; k < 9; ++k) g[c + k] = nparts[c] == 0 ? acc[c + k] : tot[c + k] + acc[c + k];
            tile_of[c] = -1;
        }
        free(acc); free(tot); free(tile_of); free(nparts);
    }
    free(vals); free(gq);
    return value;
}

/* ===========================================================================
 * The minimiser: Polak-Ribiere conjugate gradients with a derivative-aware
 * Brent line search (external/include/minimize_nrc.h).  Restated from the
 * published algorithm with the reference's constants and evaluation order.
 * =========================================================================*/

/* Optional overrides of the scalars the control logic sees.  The replay check
 * (ro_cgd_replay) feeds the device's recorded values through these so that the
 * oracle's decisions can be compared with the device's step by step. */
typedef struct {
    void *ctx;
    double (*on_f)(void *ctx, double a, double f_own);
    double (*on_slope)(void *ctx, double s_own, double s_abs, const double *xi, int n);
    double (*on_start)(void *ctx, double fp_own);
    void (*on_linmin)(void *ctx, double amin, double fmin);
    void (*on_iter)(void *ctx, double *test, double *gg, double *dgg);
    /* start of line minimisation `its`: p, xi (= h) may be inspected / replaced */
    void (*on_vectors)(void *ctx, int its, int n, double *p, double *xi, double *h);
    /* gg and dgg of the Polak-Ribiere step formed another way (the device's trees) */
    void (*on_cg_sums)(void *ctx, int n, const double *g, const double *xi, double *gg, double *dgg);
} ro_hooks;

typedef struct {
    int n;
    ro_func_cb f;
    ro_grad_cb df;
    void *ctx;
    const double *p, *xi; /* line: p + a * xi */
    double *xt, *dft;
    const ro_hooks *hk;   /* optional value overrides (replay); NULL normally */
} line_t;

/* Df1dim::operator() (minimize_nrc.h:432-436) */
static double line_f(line_t *L, double a)
{
    for (int j = 0; j < L->n; ++j) L->xt[j] = L->p[j] + a * L->xi[j];
    double f = L->f(L->ctx, L->xt);
    if (L->hk && L->hk->on_f) f = L->hk->on_f(L->hk->ctx, a, f);
    return f;
}

/* Df1dim::df (minimize_nrc.h:439-447): slope at the xt left by line_f */
static double line_df(line_t *L)
{
    double s = 0.0;
    L->df(L->ctx, L->xt, L->dft);
    for (int j = 0; j < L->n; ++j) s += L->dft[j] * L->xi[j];
    if ((g_experiment & 2) && g_slope_by_factor) s = g_slope_by_factor(L->ctx, L->xi);
    if (g_slope_topology) s = g_slope_topology(L->ctx, L->xi);
    if (L->hk && L->hk->on_slope) {
        double sabs = 0.0;
        for (int j = 0; j < L->n; ++j) sabs += fabs(L->dft[j] * L->xi[j]);
        s = L->hk->on_slope(L->hk->ctx, s, sabs, L->xi, L->n);
    }
    return s;
}

typedef struct { double ax, bx, cx, fa, fb, fc; } bracket_t;

/* Bracketmethod::bracket (minimize_nrc.h:80-151): walk downhill from (a,b)
 * with golden-ratio growth and bounded parabolic extrapolation until
 * f(bx) <= f(cx). */
static void bracket_min(bracket_t *B, double a, double b, line_t *L)
{
    const double GOLD = 1.618034, GLIMIT = 100.0, TINY = 1.0e-20;
    double ax = a, bx = b, cx, fa, fb, fc, fu, tmp;
    fa = line_f(L, ax);
    fb = line_f(L, bx);
    if (fb > fa) {
        tmp = ax; ax = bx; bx = tmp;
        tmp = fa; fa = fb; fb = tmp;
    }
    cx = bx + GOLD * (bx - ax);
    fc = line_f(L, cx);
    while (fb > fc) {
        const double r = (bx - ax) * (fb - fc);
        const double q = (bx - cx) * (fb - fa);
        const double qr = q - r;
        double u = bx - ((bx - cx) * q - (bx - ax) * r) /
                            (2.0 * copysign(fmax(fabs(qr), TINY), qr));
        const double ulim = bx + GLIMIT * (cx - bx);
        if ((bx - u) * (u - cx) > 0.0) {          /* u between b and c */
            fu = line_f(L, u);
            if (fu < fc) { ax = bx; bx = u; fa = fb; fb = fu; break; }
            if (fu > fb) { cx = u; fc = fu; break; }
            u = cx + GOLD * (cx - bx);
            fu = line_f(L, u);
        } else if ((cx - u) * (u - ulim) > 0.0) { /* u between c and the limit */
            fu = line_f(L, u);
            if (fu < fc) {
                const double unew = u + GOLD * (u - cx);
                bx = cx; cx = u; u = unew;
                fb = fc; fc = fu; fu = line_f(L, u);
            }
        } else if ((u - ulim) * (ulim - cx) >= 0.0) { /* clip to the limit */
            u = ulim;
            fu = line_f(L, u);
        } else {
            u = cx + GOLD * (cx - bx);
            fu = line_f(L, u);
        }
        ax = bx; bx = cx; cx = u;
        fa = fb; fb = fc; fc = fu;
    }
    B->ax = ax; B->bx = bx; B->cx = cx; B->fa = fa; B->fb = fb; B->fc = fc;
}

/* Dbrent::minimize (minimize_nrc.h:284-404).  Returns 0 and sets xmin/fmin,
 * or 1 when 100 iterations did not converge (the reference throws). */
static int dbrent_min(const bracket_t *B, line_t *L, double *xmin, double *fmin)
{
    const int ITMAX = 100;
    const double tol = 3.0e-8; /* Dbrent default ctor; independent of ftol (:288,:499) */
    const double ZEPS = DBL_EPSILON * 1.0e-3;
    double a = (B->ax < B->cx ? B->ax : B->cx);
    double b = (B->ax > B->cx ? B->ax : B->cx);
    double x, w, v, fx, fw, fv, dx, dw, dv, u, fu, du;
    double d = 0.0, e = 0.0;
    x = w = v = B->bx;
    fw = fv = fx = line_f(L, x);
    dw = dv = dx = line_df(L);
    for (int it = 0; it < ITMAX; ++it) {
        const double xm = 0.5 * (a + b);
        const double tol1 = tol * fabs(x) + ZEPS;
        const double tol2 = 2.0 * tol1;
        if (fabs(x - xm) <= (tol2 - 0.5 * (b - a))) { *fmin = fx; *xmin = x; return 0; }
        int bisect = 1;
        if (fabs(e) > tol1) {
            /* secant steps through (x,dx) and each of (w,dw), (v,dv) */
            double d1 = 2.0 * (b - a), d2 = d1;
            if (dw != dx) d1 = (w - x) * dx / (dx - dw);
            if (dv != dx) d2 = (v - x) * dx / (dx - dv);
            const double u1 = x + d1, u2 = x + d2;
            const int ok1 = (a - u1) * (u1 - b) > 0.0 && dx * d1 <= 0.0;
            const int ok2 = (a - u2) * (u2 - b) > 0.0 && dx * d2 <= 0.0;
            const double olde = e;
            e = d;
            if (ok1 || ok2) {
                if (ok1 && ok2) d = (fabs(d1) < fabs(d2) ? d1 : d2);
                else if (ok1) d = d1;
                else d = d2;
                if (fabs(d) <= fabs(0.5 * olde)) {
                    u = x + d;
                    if (u - a < tol2 || b - u < tol2) d = copysign(tol1, xm - x);
                    bisect = 0;
                }
            }
        }
        if (bisect) { e = (dx >= 0.0 ? a - x : b - x); d = 0.5 * e; }
        if (fabs(d) >= tol1) {
            u = x + d;
            fu = line_f(L, u);
        } else {
            u = x + copysign(tol1, d);
            fu = line_f(L, u);
            if (fu > fx) { *fmin = fx; *xmin = x; return 0; }
        }
        du = line_df(L);
        if (fu <= fx) {
            if (u >= x) a = x; else b = x;
            v = w; fv = fw; dv = dw;
            w = x; fw = fx; dw = dx;
            x = u; fx = fu; dx = du;
        } else {
            if (u < x) a = u; else b = u;
            if (fu <= fw || w == x) {
                v = w; fv = fw; dv = dw;
                w = u; fw = fu; dw = du;
            } else if (fu < fv || v == x || v == w) {
                v = u; fv = fu; dv = du;
            }
        }
    }
    return 1;
}

/* Frprmn::minimize (minimize_nrc.h:619-691) with Dlinemethod::linmin (:492-513) */
static int frprmn_ex(int n, double *x, ro_func_cb f, ro_grad_cb df, void *ctx,
                     int maxiters, double ftol, double *fret_out, int *iter_out,
                     const ro_hooks *hk)
{
    const double EPS = 1.0e-18, GTOL = 1.0e-8;
    double *buf = malloc(sizeof(double) * (size_t)(6 * n + 6));
    double *p = buf, *xi = p + n, *g = xi + n, *h = g + n, *xt = h + n, *dft = xt + n;
    line_t L = { n, f, df, ctx, p, xi, xt, dft, hk };
    int reason = RO_EXIT_ITMAX, iter = 0;
    double fret = DBL_MAX;

    memcpy(p, x, sizeof(double) * (size_t)n);
    double fp = f(ctx, p);
    if (hk && hk->on_start) fp = hk->on_start(hk->ctx, fp);
    df(ctx, p, xi);
    for (int j = 0; j < n; ++j) { g[j] = -xi[j]; xi[j] = h[j] = g[j]; }

    for (int its = 0; its < maxiters; ++its) {
        iter = its;
        /* linmin: bracket from (0,1), Brent with derivatives, move p */
        bracket_t B;
        double amin, fmin;
        if (hk && hk->on_vectors) hk->on_vectors(hk->ctx, its, n, p, xi, h);
        bracket_min(&B, 0.0, 1.0, &L);
        if (dbrent_min(&B, &L, &amin, &fmin)) { reason = RO_EXIT_DBRENT_ITMAX; goto done; }
        for (int j = 0; j < n; ++j) { xi[j] *= amin; p[j] += xi[j]; }
        fret = fmin;
        if (hk && hk->on_linmin) hk->on_linmin(hk->ctx, amin, fmin);

        if (2.0 * fabs(fret - fp) <= ftol * (fabs(fret) + fabs(fp) + EPS)) { reason = RO_EXIT_FTOL; goto done; }
        fp = fret;
        df(ctx, p, xi);
        double test = 0.0;
        const double den = fmax(fabs(fp), 1.0);
        for (int j = 0; j < n; ++j) {
            const double t = fabs(xi[j]) * fmax(fabs(p[j]), 1.0) / den;
            if (t > test) test = t;
        }
        double gg = 0.0, dgg = 0.0;
        for (int j = 0; j < n; ++j) {
            gg += g[j] * g[j];
            dgg += (xi[j] + g[j]) * xi[j];
        }
        if (hk && hk->on_cg_sums) hk->on_cg_sums(hk->ctx, n, g, xi, &gg, &dgg);
        if (hk && hk->on_iter) hk->on_iter(hk->ctx, &test, &gg, &dgg);
        if (test < GTOL) { reason = RO_EXIT_GTOL; goto done; }
        if (gg == 0.0) { reason = RO_EXIT_GGZERO; goto done; }
        const double gam = dgg / gg;
        for (int j = 0; j < n; ++j) {
            g[j] = -xi[j];
            xi[j] = h[j] = g[j] + gam * h[j];
        }
    }
done:
    memcpy(x, p, sizeof(double) * (size_t)n);
    *fret_out = fret;
    *iter_out = iter;
    free(buf);
    return reason;
}

int ro_frprmn(int n, double *x, ro_func_cb f, ro_grad_cb df, void *ctx,
              int maxiters, double ftol, double *fret_out, int *iter_out)
{
    return frprmn_ex(n, x, f, df, ctx, maxiters, ftol, fret_out, iter_out, NULL);
}

/* ===========================================================================
 * CGDSubspaceOptimizer::optimize and its SubfunctionFD functor
 * (src/optimizers/CGDSubspaceOptimizer.cpp:19-98, 124-184)
 * =========================================================================*/

typedef struct {
    ro_problem *p;
    int64_t nfree, nf;
    const int64_t *free_vid, *fac;
    double *gdense;
    int merge, saw_nan;
    int64_t nfeval, ngeval;
} sub_t;

/* SubfunctionFD::quickAssignVals (.cpp:160-184) */
static void sub_assign(sub_t *S, const double *x)
{
    for (int64_t i = 0; i < S->nfree; ++i) {
        const int64_t v = S->free_vid[i];
        if (isnan(x[i])) S->saw_nan = 1;
        assign_one(S->p, v, closest_val(x[i], S->p->lo[v], S->p->hi[v]));
    }
}

/* SubfunctionFD::operator() (.cpp:124-132); coeff = +1 for MinSum */
static double sub_f(void *ctx, const double *x)
{
    sub_t *S = ctx;
    ++S->nfeval;
    sub_assign(S, x);
    return ro_eval_factors(S->p, S->nf, S->fac);
}

static double sum_tree(const double *t, int64_t lo, int64_t hi)
{
    if (hi - lo <= 8) {
        double s = 0.0;
        for (int64_t i = lo; i < hi; ++i) s += t[i];
        return s;
    }
    const int64_t mid = lo + (hi - lo) / 2;
    const double a = sum_tree(t, lo, mid);
    return a + sum_tree(t, mid, hi);
}

/* (experiment, ro_set_experiment bit 1) the slope at the assigned point, factor by factor in list order; bundle adjustment only */
static double sub_slope_by_factor(void *ctx, const double *xi)
{
    sub_t *S = ctx;
    int64_t vids[64]; double vals[64];
    double *dir = calloc((size_t)S->p->nvars + 1, sizeof(double));
    for (int64_t i = 0; i < S->nfree; ++i) dir[S->free_vid[i]] = xi[i];
    double s = 0.0;
    double *terms = (g_experiment & 4) ? malloc(sizeof(double) * (size_t)(S->nf + 1)) : NULL;
    for (int64_t i = 0; i < S->nf; ++i) {
        const int n = factor_partials(S->p, S->fac ? S->fac[i] : i, vids, vals);
        double t = 0.0;
        for (int k = 0; k < n && k < 64; ++k) t += vals[k] * dir[vids[k]];
        if (terms) terms[i] = t; else s += t;
    }
    if (terms) {   /* bit 2: the factors' terms added as a tree (a device's reduction) instead of in list order */
        s = sum_tree(terms, 0, S->nf);
        free(terms);
    }
    free(dir);
    return s;
}

/* RO_SUM_TOPOLOGY_COOPERATIVE: a trial's slope the device's way -- per factor sum_k partial_k direction_k over its twelve slots in
 * slot order, the factors' terms added like the values (coop_tree_sum) */
static double sub_slope_topology(void *ctx, const double *xi)
{
    sub_t *S = ctx;
    ro_problem *p = S->p;
    double *dir = calloc((size_t)p->nvars + 1, sizeof(double));
    double *terms = malloc(sizeof(double) * (size_t)(S->nf + 1));
    for (int64_t i = 0; i < S->nfree; ++i) dir[S->free_vid[i]] = xi[i];
    for (int64_t i = 0; i < S->nf; ++i) {
        const int64_t f = S->fac ? S->fac[i] : i;
        double x[12], gq[12], acc = 0.0;
        gather_ba(p, f, x);
        ba_grad(p, x, p->obs[2 * f], p->obs[2 * f + 1], gq);
        /* (the device's loop stands outside the factor arithmetic's no-contraction region: its multiply-adds are fused) */
        for (int k = 0; k < 12; ++k) acc = fma(gq[k], dir[fac_var(p, f, k)], acc);
        terms[i] = acc;
    }
    const double s = coop_tree_sum(terms, S->nf);
    free(terms); free(dir);
    return s;
}

/* RO_SUM_TOPOLOGY_LDS: a trial's slope in the LDS-resident solver -- per factor the forward-mode slope along the direction (the
 * external arithmetic's value_slope: factors.hpp ba_slope_dir), the factors' terms added like the values (lds_tree_sum) */
static double sub_slope_lds(void *ctx, const double *xi)
{
    sub_t *S = ctx;
    ro_problem *p = S->p;
    double *dir = calloc((size_t)p->nvars + 1, sizeof(double));
    double *terms = malloc(sizeof(double) * (size_t)(S->nf + 1));
    for (int64_t i = 0; i < S->nfree; ++i) dir[S->free_vid[i]] = xi[i];
    for (int64_t i = 0; i < S->nf; ++i) {
        const int64_t f = S->fac ? S->fac[i] : i;
        double x[12], d[12], sl = 0.0;
        gather_ba(p, f, x);
        for (int k = 0; k < 12; ++k) d[k] = dir[fac_var(p, f, k)];
        (void)p->ext->value_slope(x, d, p->obs[2 * f], p->obs[2 * f + 1], 0, &sl);
        terms[i] = sl;
    }
    const double s = p->topo == RO_SUM_TOPOLOGY_WG ? wg_tree_sum(p, terms, S->nf) : lds_tree_sum(terms, S->nf, p->lds_nt);
    free(terms); free(dir);
    return s;
}

/* ... and gg, dgg: lane l adds the terms of the free slots l, l + nt, ... (slot order: camera blocks, then point blocks) */
static void sub_cg_sums_lds(void *ctx, int n, const double *g, const double *xi, double *gg, double *dgg)
{
    sub_t *S = ctx;
    ro_problem *p = S->p;
    int64_t *li = malloc(sizeof(int64_t) * (size_t)(p->nvars + 1));
    for (int64_t v = 0; v < p->nvars; ++v) li[v] = -1;
    for (int64_t i = 0; i < n; ++i) li[S->free_vid[i]] = i;
    /* (a slot that is not free adds nothing and is skipped by the device's loop: the lane's chain simply does not see it) */
    const int nt = p->lds_nt, nw = nt / 64;
    double wa[16], wb[16];
    for (int w = 0; w < 16; ++w) wa[w] = wb[w] = 0.0;
    for (int w = 0; w < nw; ++w) {
        double la[64], lb[64];
        for (int l = 0; l < 64; ++l) {
            double a = 0.0, b = 0.0;
            for (int64_t sidx = 64 * w + l; sidx < p->lds_nslots; sidx += nt) {
                const int64_t i = li[p->lds_slot_vid[sidx]];
                if (i < 0) continue;
                a = a + g[i] * g[i];
                b = b + (xi[i] + g[i]) * xi[i];
            }
            la[l] = a; lb[l] = b;
        }
        wa[w] = tree64(la); wb[w] = tree64(lb);
    }
    if (nw == 1) { *gg = wa[0]; *dgg = wb[0]; }
    else if (nw <= 4) { *gg = (wa[0] + wa[1]) + (wa[2] + wa[3]); *dgg = (wb[0] + wb[1]) + (wb[2] + wb[3]); }
    else {
        for (int w = 1; w < 16; w *= 2)
            for (int i = 0; i < 16; i += 2 * w) { wa[i] = wa[i] + wa[i + w]; wb[i] = wb[i] + wb[i + w]; }
        *gg = wa[0]; *dgg = wb[0];
    }
    free(li);
}

/* RO_SUM_TOPOLOGY_PTM: a trial's slope in the point-major streaming solver -- with the value, from the cameras' records */
static double sub_slope_ptm(void *ctx, const double *xi)
{
    sub_t *S = ctx;
    ro_problem *p = S->p;
    double *dir = calloc((size_t)p->nvars + 12, sizeof(double));
    for (int64_t i = 0; i < S->nfree; ++i) dir[S->free_vid[i]] = xi[i];
    double s = 0.0;
    (void)ptm_eval(p, dir, &s);
    free(dir);
    return s;
}

/* ... and gg, dgg: a lane's point blocks as the gradient pass finishes them, then its camera slots (a group: the first workgroup's
 * lanes speak for the cameras) */
static void sub_cg_sums_ptm(void *ctx, int n, const double *g, const double *xi, double *gg, double *dgg)
{
    sub_t *S = ctx;
    ro_problem *p = S->p;
    const struct ptm_tab *T = p->ptm;
    const int nt = p->ptm_nt, nw = nt / 64, K = p->ptm_K;
    int64_t *li = malloc(sizeof(int64_t) * (size_t)(p->nvars + 1));
    for (int64_t v = 0; v < p->nvars; ++v) li[v] = -1;
    for (int64_t i = 0; i < n; ++i) li[S->free_vid[i]] = i;
    static const int var_of_slot[10] = {3, 4, 5, 6, 7, 8, 0, 1, 2, -1};   /* [t f k1 k2 | r | pad] */
    double *wa = calloc((size_t)K * nw + 16, sizeof(double)), *wb = calloc((size_t)K * nw + 16, sizeof(double));
    /* LOCAL: every workgroup's cameras (ascending) and who holds a camera first */
    int64_t *lptr = NULL, *lcount = NULL, *lcam = NULL, *first_holder = NULL;
    if (p->ptm_wg_chunk0) {
        const int64_t ncb = p->ptm_ncb;
        lptr = calloc((size_t)K + 1, sizeof(int64_t)); lcount = calloc((size_t)K + 1, sizeof(int64_t));
        lcam = malloc(sizeof(int64_t) * (size_t)K * (size_t)(ncb + 1));
        first_holder = malloc(sizeof(int64_t) * (size_t)(ncb + 1));
        for (int64_t c = 0; c < ncb; ++c) first_holder[c] = -1;
        uint8_t *held = malloc((size_t)ncb + 1);
        for (int rk = 0; rk < K; ++rk) {
            memset(held, 0, (size_t)ncb + 1);
            for (int64_t b = 64 * p->ptm_wg_chunk0[rk]; b < 64 * p->ptm_wg_chunk0[rk + 1] && b < p->ptm_npb; ++b)
                for (int64_t k = T->pptr[b]; k < T->pptr[b + 1]; ++k) held[T->fcam[T->pidx[k]]] = 1;
            lptr[rk] = rk * (ncb + 1);
            for (int64_t c = 0; c < ncb; ++c)
                if (held[c]) { lcam[lptr[rk] + lcount[rk]++] = c; if (first_holder[c] < 0) first_holder[c] = rk; }
            if (lcount[rk] == 0) { lcam[lptr[rk]] = 0; lcount[rk] = 1; }   /* (a workgroup without chunks is laid out for one camera) */
        }
        free(held);
    }
    for (int rk = 0; rk < K; ++rk)
    for (int w = 0; w < nw; ++w) {
        double la[64], lb[64];
        for (int l = 0; l < 64; ++l) {
            double a = 0.0, b = 0.0;
            int64_t c0, cstep, cend;
            ptm_wave_chunks(p, T->npc, rk, w, &c0, &cstep, &cend);
            for (int pass = 0; pass < 2; ++pass)   /* the chunks with factors as the rounds finish them, then those without */
                for (int64_t ch = c0; ch < cend; ch += cstep) {
                    const int empty = T->cp[ch] >= T->cp[ch + 1];
                    if (empty != pass) continue;
                    const int64_t blkid = 64 * ch + l;
                    if (blkid >= p->ptm_npb) continue;
                    for (int k = 0; k < 3; ++k) {
                        const int64_t i = li[p->ptm_pt[blkid] + k];
                        if (i < 0) continue;
                        a = a + g[i] * g[i];
                        b = b + (xi[i] + g[i]) * xi[i];
                    }
                }
            /* the camera slots: the first workgroup's lanes over all cameras -- or, LOCAL, every workgroup's lanes over the cameras it
             * holds (ascending, its own numbering) where it is the first to hold them */
            const int64_t nloc = p->ptm_wg_chunk0 ? lcount[rk] : p->ptm_ncb;
            for (int64_t sidx = 64 * w + l; (p->ptm_wg_chunk0 || rk == 0) && sidx < 10 * nloc; sidx += nt) {
                const int k = var_of_slot[sidx % 10];
                if (k < 0) continue;
                const int64_t cam = p->ptm_wg_chunk0 ? lcam[lptr[rk] + sidx / 10] : sidx / 10;
                if (p->ptm_wg_chunk0 && first_holder[cam] != rk) continue;
                const int64_t i = li[p->ptm_cam[cam] + k];
                if (i < 0) continue;
                a = a + g[i] * g[i];
                b = b + (xi[i] + g[i]) * xi[i];
            }
            la[l] = a; lb[l] = b;
        }
        wa[rk * nw + w] = tree64(la); wb[rk * nw + w] = tree64(lb);
    }
    *gg = ptm_combine(wa, nw, K, p->ptm_wide); *dgg = ptm_combine(wb, nw, K, p->ptm_wide);
    free(li); free(wa); free(wb); free(lptr); free(lcount); free(lcam); free(first_holder);
}

/* RO_SUM_TOPOLOGY_GROUP: a trial's slope in the solver of tiny components */
static double sub_slope_group(void *ctx, const double *xi)
{
    sub_t *S = ctx;
    ro_problem *p = S->p;
    double *dir = calloc((size_t)p->nvars + 1, sizeof(double));
    double *terms = malloc(sizeof(double) * (size_t)(S->nf + 1));
    for (int64_t i = 0; i < S->nfree; ++i) dir[S->free_vid[i]] = xi[i];
    for (int64_t i = 0; i < S->nf; ++i) {
        const int64_t f = S->fac ? S->fac[i] : i;
        double x[12], gq[12], acc = 0.0;
        gather_ba(p, f, x);
        ba_grad(p, x, p->obs[2 * f], p->obs[2 * f + 1], gq);
        for (int k = 0; k < 12; ++k) acc = fma(gq[k], dir[fac_var(p, f, k)], acc);
        terms[i] = acc;
    }
    const double s = group_tree_sum(terms, S->nf, p->lds_nt);
    free(terms); free(dir);
    return s;
}

/* RO_SUM_TOPOLOGY_WG: a trial's slope in the plain workgroup solver, nonlinear-product factors */
static double sub_slope_wg(void *ctx, const double *xi)
{
    sub_t *S = ctx;
    ro_problem *p = S->p;
    double *dir = calloc((size_t)p->nvars + 1, sizeof(double));
    double *terms = malloc(sizeof(double) * (size_t)(S->nf + 1));
    for (int64_t i = 0; i < S->nfree; ++i) dir[S->free_vid[i]] = xi[i];
    for (int64_t i = 0; i < S->nf; ++i) {
        const int64_t f = S->fac ? S->fac[i] : i;
        double acc = 0.0;
        for (int64_t k = p->rowptr[f]; k < p->rowptr[f + 1]; ++k) acc = fma(nlp_deriv(p, f, p->vid[k]), dir[p->vid[k]], acc);
        terms[i] = acc;
    }
    const double s = wg_tree_sum(p, terms, S->nf);
    free(terms); free(dir);
    return s;
}

static void sub_cg_sums_wg(void *ctx, int n, const double *g, const double *xi, double *gg, double *dgg)
{
    sub_t *S = ctx;
    double *a = malloc(sizeof(double) * (size_t)(n + 1)), *b = malloc(sizeof(double) * (size_t)(n + 1));
    for (int i = 0; i < n; ++i) { a[i] = g[i] * g[i]; b[i] = (xi[i] + g[i]) * xi[i]; }
    *gg = wg_tree_sum(S->p, a, n);
    *dgg = wg_tree_sum(S->p, b, n);
    free(a); free(b);
}

/* ... and gg, dgg: lane i carries variable i's terms unless a wave owns it, wave w's first lane those of its variable */
static void sub_cg_sums_topology(void *ctx, int n, const double *g, const double *xi, double *gg, double *dgg)
{
    sub_t *S = ctx;
    ro_problem *p = S->p;
    int64_t lanes = n;
    if (64 * p->nwave_owned > lanes) lanes = 64 * p->nwave_owned;
    double *a = calloc((size_t)lanes + 64, sizeof(double)), *b = calloc((size_t)lanes + 64, sizeof(double));
    for (int64_t i = 0; i < n; ++i) {
        if (p->wave_of[S->free_vid[i]] >= 0) continue;
        a[i] = g[i] * g[i];
        b[i] = (xi[i] + g[i]) * xi[i];
    }
    int64_t *li = malloc(sizeof(int64_t) * (size_t)(p->nvars + 1));
    for (int64_t v = 0; v < p->nvars; ++v) li[v] = -1;
    for (int64_t i = 0; i < n; ++i) li[S->free_vid[i]] = i;
    for (int64_t w = 0; w < p->nwave_owned; ++w) {
        const int64_t i = li[p->wave_vid[w]];
        if (i < 0) continue;
        a[64 * w] = a[64 * w] + g[i] * g[i];
        b[64 * w] = b[64 * w] + (xi[i] + g[i]) * xi[i];
    }
    *gg = coop_tree_sum(a, lanes);
    *dgg = coop_tree_sum(b, lanes);
    free(a); free(b); free(li);
}

/* SubfunctionFD::df (.cpp:135-157) */
static void sub_df(void *ctx, const double *x, double *deriv)
{
    sub_t *S = ctx;
    ++S->ngeval;
    sub_assign(S, x);
    ro_compute_gradient(S->p, S->nf, S->fac, S->gdense, S->merge);
    for (int64_t i = 0; i < S->nfree; ++i) deriv[i] = S->gdense[S->free_vid[i]];
}

void ro_cgd_optimize(ro_problem *p, int64_t nfree, const int64_t *free_vid,
                     int64_t nf, const int64_t *fac, double *xval,
                     int32_t maxiters, double ftol, int merge, ro_result *out)
{
    memset(out, 0, sizeof(*out));
    if (nf == 0) { out->status = RO_EXIT_EMPTY; return; } /* .cpp:26-29 */

    sub_t S = { p, nfree, nf, free_vid, fac, NULL, merge, 0, 0, 0 };
    g_slope_by_factor = p->kind == RO_KIND_BA ? sub_slope_by_factor : 0;
    S.gdense = malloc(sizeof(double) * (size_t)(p->nvars + 1));
    double *xinit = malloc(sizeof(double) * (size_t)(nfree + 1));
    double *xw = malloc(sizeof(double) * (size_t)(nfree + 1));

    if (p->topo == RO_SUM_TOPOLOGY_PTM && p->kind == RO_KIND_BA) { ptm_free(p->ptm); p->ptm = ptm_build(p, nf, fac); p->ptm_at_start = 0; }
    /* SubspaceOptimizer::quickAssignVals(vars, xval, true) then sfd(xval) */
    const double finit = sub_f(&S, xval);
    memcpy(xinit, xval, sizeof(double) * (size_t)nfree);
    memcpy(xw, xval, sizeof(double) * (size_t)nfree);

    double fret; int iter;
    int reason;
    if (p->topo == RO_SUM_TOPOLOGY_COOPERATIVE && p->kind == RO_KIND_BA) {
        ro_hooks hk;
        memset(&hk, 0, sizeof hk);
        hk.ctx = &S; hk.on_cg_sums = sub_cg_sums_topology;
        g_slope_topology = sub_slope_topology;
        reason = frprmn_ex((int)nfree, xw, sub_f, sub_df, &S, maxiters, ftol, &fret, &iter, &hk);
        g_slope_topology = 0;
    } else if (p->topo == RO_SUM_TOPOLOGY_GROUP && p->kind == RO_KIND_BA && p->ext) {
        g_slope_topology = sub_slope_group;   /* (gg and dgg: the reference's order) */
        reason = frprmn_ex((int)nfree, xw, sub_f, sub_df, &S, maxiters, ftol, &fret, &iter, NULL);
        g_slope_topology = 0;
    } else if (p->topo == RO_SUM_TOPOLOGY_WG && (p->kind == RO_KIND_NLP || p->ext)) {
        /* the variables fed by more than 64 listed partials: their runs strided over a wave (solver_wg.hpp: WG_LONG_LIST) */
        int64_t *cnt = calloc((size_t)p->nvars + 1, sizeof(int64_t)), nlong = 0;
        for (int64_t i = 0; i < nf; ++i) {
            const int64_t f = fac ? fac[i] : i;
            for (int64_t k = 0, a = fac_arity(p, f); k < a; ++k) cnt[fac_var(p, f, k)]++;
        }
        int64_t *lv = malloc(sizeof(int64_t) * (size_t)(nfree + 1));
        for (int64_t i = 0; i < nfree; ++i) if (cnt[free_vid[i]] > 64) lv[nlong++] = free_vid[i];
        const int nt = p->lds_nt;
        ro_set_sum_topology(p, RO_SUM_TOPOLOGY_COOPERATIVE, nlong, lv);   /* (borrows the wave-owned variables' bookkeeping) */
        p->topo = RO_SUM_TOPOLOGY_WG; p->lds_nt = nt;   /* (stream_nwg is left as it was) */
        free(cnt); free(lv);
        ro_hooks hk;
        memset(&hk, 0, sizeof hk);
        hk.ctx = &S; hk.on_cg_sums = sub_cg_sums_wg;
        /* (bundle adjustment on this solver: the forward-mode slope per factor of the batch solvers, the terms added as the values are) */
        g_slope_topology = p->kind == RO_KIND_NLP ? sub_slope_wg : sub_slope_lds;
        reason = frprmn_ex((int)nfree, xw, sub_f, sub_df, &S, maxiters, ftol, &fret, &iter, &hk);
        g_slope_topology = 0;
    } else if (p->topo == RO_SUM_TOPOLOGY_PTM && p->kind == RO_KIND_BA && p->ext && p->ptm_ar) {
        ro_hooks hk;
        memset(&hk, 0, sizeof hk);
        hk.ctx = &S; hk.on_cg_sums = sub_cg_sums_ptm;
        g_slope_topology = sub_slope_ptm;
        reason = frprmn_ex((int)nfree, xw, sub_f, sub_df, &S, maxiters, ftol, &fret, &iter, &hk);
        g_slope_topology = 0;
    } else if (p->topo == RO_SUM_TOPOLOGY_LDS && p->kind == RO_KIND_BA && p->ext) {
        ro_hooks hk;
        memset(&hk, 0, sizeof hk);
        hk.ctx = &S; hk.on_cg_sums = sub_cg_sums_lds;
        g_slope_topology = sub_slope_lds;
        reason = frprmn_ex((int)nfree, xw, sub_f, sub_df, &S, maxiters, ftol, &fret, &iter, &hk);
        g_slope_topology = 0;
    } else {
        reason = ro_frprmn((int)nfree, xw, sub_f, sub_df, &S, maxiters, ftol, &fret, &iter);
    }
    if (S.saw_nan) reason = RO_EXIT_NAN;

    sub_assign(&S, xw); /* assign gdmin.p with sanitisation (.cpp:61) */
    int status = reason;
    if (fret > finit) { /* negative progress: restore (.cpp:66-80) */
        status |= RO_STATUS_ROLLED_BACK;
        p->ptm_at_start = 1;
        fret = sub_f(&S, xinit);
        p->ptm_at_start = 0;
    }
    for (int64_t i = 0; i < nfree; ++i) xval[i] = p->x[free_vid[i]]; /* .cpp:84-86 */

    out->fret = fret; out->finit = finit; out->delta = fret - finit;
    out->iters = iter; out->status = status;
    out->nfeval = S.nfeval; out->ngeval = S.ngeval;
    free(S.gdense); free(xinit); free(xw);
    if (p->ptm) { ptm_free(p->ptm); p->ptm = NULL; }
}

/* ===========================================================================
 * Replay check.  The device solver can record what its control logic saw: one
 * record {tag, a, b, c} per event (tags as in rdis_amd/csrc/minimizer.hpp):
 *   1 line value      a = step, b = f          2 line value+slope  c = slope
 *   3 CG reductions   a = test, b = gg, c = dgg
 *   4 start           a = fp                   5 line minimum a = amin, b = fmin
 * ro_cgd_replay runs THIS file's solver on the same inputs, evaluates every
 * quantity itself, compares it with the record, and then continues with the
 * recorded value.  If the device implements the same algorithm, every step
 * length the oracle asks for is bit-identical to the recorded one (the
 * decisions are a pure function of the values seen), and the values agree to
 * rounding.  This separates "same algorithm" from the chaotic sensitivity of
 * the iterates to rounding.
 * =========================================================================*/
typedef struct {
    const double *rec;
    int64_t nrec, pos;
    sub_t *S;
    ro_replay_report *rep;
    const double *vdump; /* [dump_iters][2][n]: the device's p and xi at the start of each line search */
    int dump_iters;
} replay_t;

static const double *replay_next(replay_t *R, int tag_a, int tag_b)
{
    if (R->pos >= R->nrec) { R->rep->underrun = 1; return NULL; }
    const double *r = R->rec + 4 * R->pos;
    const int tag = (int)r[0];
    if (tag != tag_a && tag != tag_b) {
        if (R->rep->first_mismatch < 0) R->rep->first_mismatch = R->pos;
        R->rep->tag_mismatches++;
        return NULL;
    }
    R->pos++;
    return r;
}

static void track(double *worst, double own, double dev, double scale)
{
    const double d = fabs(own - dev) / (scale > 0 ? scale : 1.0);
    if (d > *worst || d != d) *worst = d;
}

static double replay_on_f(void *ctx, double a, double f_own)
{
    replay_t *R = ctx;
    const double *r = replay_next(R, 1, 2);
    if (!r) return f_own;
    if (memcmp(&r[1], &a, sizeof(double)) != 0) {
        if (R->rep->first_mismatch < 0) R->rep->first_mismatch = R->pos - 1;
        R->rep->step_mismatches++;
        track(&R->rep->max_step_rel, a, r[1], fabs(a));
    }
    /* scale: sum of |factor values| at the oracle's own point */
    double sabs = 0.0;
    for (int64_t i = 0; i < R->S->nf; ++i)
        sabs += fabs(factor_value_nocache(R->S->p, R->S->fac ? R->S->fac[i] : i));
    {
        double B = 0.0, vb, gb;
        for (int64_t i = 0; i < R->S->nf; ++i) { factor_bounds(R->S->p, R->S->fac ? R->S->fac[i] : i, &vb, &gb); B += vb; }
        track(&R->rep->max_f_bound, f_own, r[2], DBL_EPSILON * B);
        R->rep->last_near = B <= RO_NEAR_AMPLIFICATION * sabs && fabs(f_own) <= 4.0 * fabs(R->rep->finit) + 1.0;
    }
    track(&R->rep->max_f_rel, f_own, r[2], sabs);
    if (R->rep->last_near) track(&R->rep->max_f_rel_near, f_own, r[2], sabs);
    else {
        /* a far-out trial point: how much does the objective move when every free variable
         * moves by one unit in the last place?  The device's value is expected within a small
         * multiple of that (plus the ordinary rounding of the sum). */
        sub_t *S = R->S;
        ro_problem *p = S->p;
        double *save = malloc((size_t)(S->nfree > 0 ? S->nfree : 1) * sizeof(double));
        for (int64_t i = 0; i < S->nfree; ++i) {
            const int64_t v = S->free_vid[i];
            save[i] = p->x[v];
            p->x[v] = nextafter(p->x[v], (i & 1) ? INFINITY : -INFINITY);
        }
        double fpert = 0.0;
        for (int64_t i = 0; i < S->nf; ++i) fpert = fpert + factor_value_nocache(p, S->fac ? S->fac[i] : i);
        for (int64_t i = 0; i < S->nfree; ++i) p->x[S->free_vid[i]] = save[i];
        free(save);
        track(&R->rep->max_f_far_ulps, f_own, r[2], fabs(fpert - f_own) + DBL_EPSILON * sabs);
    }
    R->rep->pending_slope = ((int)r[0] == 2) ? r[3] : NAN;
    return r[2];
}

static double replay_on_slope(void *ctx, double s_own, double s_abs, const double *xi, int n)
{
    replay_t *R = ctx;
    const double dev = R->rep->pending_slope;
    /* scale = sum over factors and slots of |dE_i/dx_j * xi_j| at the currently
     * assigned point: the slope is a sum of these terms, so this is what its
     * rounding error is proportional to (the summed gradient entries can cancel) */
    {
        sub_t *S = R->S;
        ro_problem *p = S->p;
        double *dir = calloc((size_t)p->nvars + 1, sizeof(double));
        int64_t vids[64]; double vals[64];
        for (int i = 0; i < n; ++i) dir[S->free_vid[i]] = xi[i];
        double sc = 0.0, sb = 0.0;
        for (int64_t i = 0; i < S->nf; ++i) {
            const int64_t f = S->fac ? S->fac[i] : i;
            if (fac_arity(p, f) > 64) continue;
            const int k = factor_partials(p, f, vids, vals);
            double t = 0.0, vb, gb;
            for (int j = 0; j < k; ++j) t += fabs(vals[j] * dir[vids[j]]);
            factor_bounds(p, f, &vb, &gb);
            sc += t;
            sb += t * gb;
        }
        free(dir);
        if (sc > s_abs) s_abs = sc;
        if (R->rep->pending_slope == R->rep->pending_slope)
            track(&R->rep->max_slope_bound, s_own, R->rep->pending_slope, DBL_EPSILON * (sb > 0 ? sb : s_abs));
    }
    if (dev != dev) { /* the device recorded a value-only evaluation here */
        if (R->rep->first_mismatch < 0) R->rep->first_mismatch = R->pos - 1;
        R->rep->tag_mismatches++;
        return s_own;
    }
    track(&R->rep->max_slope_rel, s_own, dev, s_abs);
    if (R->rep->last_near) track(&R->rep->max_slope_rel_near, s_own, dev, s_abs);
    return dev;
}

static double replay_on_start(void *ctx, double fp_own)
{
    replay_t *R = ctx;
    const double *r = replay_next(R, 4, 4);
    if (!r) return fp_own;
    track(&R->rep->max_f_rel, fp_own, r[1], fabs(fp_own));
    return r[1];
}

static void replay_on_linmin(void *ctx, double amin, double fmin)
{
    replay_t *R = ctx;
    const double *r = replay_next(R, 5, 5);
    if (!r) return;
    if (memcmp(&r[1], &amin, sizeof(double)) != 0 || memcmp(&r[2], &fmin, sizeof(double)) != 0) {
        if (R->rep->first_mismatch < 0) R->rep->first_mismatch = R->pos - 1;
        R->rep->step_mismatches++;
    }
}

static void replay_on_iter(void *ctx, double *test, double *gg, double *dgg)
{
    replay_t *R = ctx;
    const double *r = replay_next(R, 3, 3);
    if (!r) return;
    track(&R->rep->max_iter_rel, *test, r[1], fabs(*test));
    track(&R->rep->max_iter_rel, *gg, r[2], fabs(*gg));
    track(&R->rep->max_iter_rel, *dgg, r[3], fabs(*gg)); /* dgg can cancel; gg is its scale */
    *test = r[1]; *gg = r[2]; *dgg = r[3];
}

/* Adopt the device's point and direction at the start of every recorded line
 * search (after measuring how far the oracle's own had drifted): evaluations are
 * then made at bit-identical points, so values can be compared to rounding over
 * the whole run instead of through the problem's condition number. */
static void replay_on_vectors(void *ctx, int its, int n, double *p, double *xi, double *h)
{
    replay_t *R = ctx;
    if (!R->vdump || its >= R->dump_iters) return;
    const double *dp = R->vdump + (size_t)2 * its * n, *dx = dp + n;
    double pn = 0.0, xn = 0.0, pd = 0.0, xd = 0.0;
    for (int j = 0; j < n; ++j) {
        pn = fmax(pn, fabs(p[j])); xn = fmax(xn, fabs(xi[j]));
        pd = fmax(pd, fabs(p[j] - dp[j])); xd = fmax(xd, fabs(xi[j] - dx[j]));
    }
    track(&R->rep->max_vec_rel, 0.0, pd, pn);
    track(&R->rep->max_vec_rel, 0.0, xd, xn);
    memcpy(p, dp, sizeof(double) * (size_t)n);
    memcpy(xi, dx, sizeof(double) * (size_t)n);
    memcpy(h, dx, sizeof(double) * (size_t)n);
    R->rep->synced_iters = its + 1;
}

void ro_cgd_replay(ro_problem *p, int64_t nfree, const int64_t *free_vid, int64_t nf,
                   const int64_t *fac, const double *xstart, int32_t maxiters, double ftol,
                   const double *trace, int64_t nrec, const double *vdump, int32_t dump_iters,
                   double *x_end, ro_replay_report *rep)
{
    memset(rep, 0, sizeof(*rep));
    rep->first_mismatch = -1;
    rep->pending_slope = NAN;
    const int emulate = p->emulate;
    ro_set_emulate_stale_cache(p, 0); /* the device recomputes every factor */
    sub_t S = { p, nfree, nf, free_vid, fac, NULL, 0, 0, 0, 0 };
    S.gdense = malloc(sizeof(double) * (size_t)(p->nvars + 1));
    double *xw = malloc(sizeof(double) * (size_t)(nfree + 1));
    memcpy(xw, xstart, sizeof(double) * (size_t)nfree);
    replay_t R = { trace, nrec, 0, &S, rep, vdump, dump_iters };
    ro_hooks hk = { &R, replay_on_f, replay_on_slope, replay_on_start, replay_on_linmin, replay_on_iter,
                    replay_on_vectors, NULL };

    rep->finit = sub_f(&S, xw);
    rep->reason = frprmn_ex((int)nfree, xw, sub_f, sub_df, &S, maxiters, ftol, &rep->fret, &rep->iters, &hk);
    rep->consumed = R.pos;
    if (x_end) memcpy(x_end, xw, sizeof(double) * (size_t)nfree);
    free(S.gdense); free(xw);
    ro_set_emulate_stale_cache(p, emulate);
}

/* ---- recording the oracle's own trace (self-check of the replay machinery) --- */
typedef struct { double *rec; int64_t cap, n; } recorder_t;

static void rec_put(recorder_t *T, int tag, double a, double b, double c)
{
    if (T->n < T->cap) { double *r = T->rec + 4 * T->n; r[0] = tag; r[1] = a; r[2] = b; r[3] = c; }
    T->n++;
}
static double rec_on_f(void *ctx, double a, double f) { rec_put(ctx, 1, a, f, 0.0); return f; }
static double rec_on_slope(void *ctx, double s, double sabs, const double *xi, int n)
{
    recorder_t *T = ctx; (void)sabs; (void)xi; (void)n;
    if (T->n >= 1 && T->n <= T->cap) { double *r = T->rec + 4 * (T->n - 1); r[0] = 2; r[3] = s; }
    return s;
}
static double rec_on_start(void *ctx, double fp) { rec_put(ctx, 4, fp, 0.0, 0.0); return fp; }
static void rec_on_linmin(void *ctx, double amin, double fmin) { rec_put(ctx, 5, amin, fmin, 0.0); }
static void rec_on_iter(void *ctx, double *t, double *gg, double *dgg) { rec_put(ctx, 3, *t, *gg, *dgg); }

int64_t ro_cgd_record(ro_problem *p, int64_t nfree, const int64_t *free_vid, int64_t nf,
                      const int64_t *fac, const double *xstart, int32_t maxiters, double ftol,
                      double *trace, int64_t cap)
{
    const int emulate = p->emulate;
    ro_set_emulate_stale_cache(p, 0);
    sub_t S = { p, nfree, nf, free_vid, fac, NULL, 0, 0, 0, 0 };
    S.gdense = malloc(sizeof(double) * (size_t)(p->nvars + 1));
    double *xw = malloc(sizeof(double) * (size_t)(nfree + 1));
    memcpy(xw, xstart, sizeof(double) * (size_t)nfree);
    recorder_t T = { trace, cap, 0 };
    ro_hooks hk = { &T, rec_on_f, rec_on_slope, rec_on_start, rec_on_linmin, rec_on_iter, NULL, NULL };
    double fret; int iter;
    (void)sub_f(&S, xw);
    (void)frprmn_ex((int)nfree, xw, sub_f, sub_df, &S, maxiters, ftol, &fret, &iter, &hk);
    free(S.gdense); free(xw);
    ro_set_emulate_stale_cache(p, emulate);
    return T.n;
}

void ro_resjac_each_ba(ro_problem *p, int64_t nf, const int64_t *fac, double *res2, double *J24)
{
    for (int64_t i = 0; i < nf; ++i) {
        const int64_t f = fac ? fac[i] : i;
        double vals[12];
        gather_ba(p, f, vals);
        ro_ba_factor_resjac(vals, p->obs[2 * f], p->obs[2 * f + 1], res2 + 2 * i, J24 + 24 * i);
    }
}

/* ------------------------------------------------------------------------
 * Connected components of the residual factor graph
 *
 * Component::createChildren (src/Component.cpp:508-549) asks the dynamic connectivity structure
 * (ConnectivityGraph.h:255-261) for the component label of every unassigned variable of the
 * parent and builds one child per label; Component::init (Component.cpp:60-79) fills a child
 * with the variables and factors of that connected component and sorts both lists by id.
 * Children are kept ordered by number of variables (ComponentComparator, :603-608).  The graph
 * is bipartite: a factor is adjacent to each of its variables, edges of assigned variables are
 * removed.  Restated as a static labelling: sequential union-find over the variables, one pass
 * over the factors.  Ties between components of equal size are broken by the smallest variable
 * id (the reference's order among them follows the internals of its Euler-tour forest).
 * Returns the number of components; free_ptr / fac_ptr need nvars + 1 entries, free_vid nvars,
 * fac_id nfac.
 * ------------------------------------------------------------------------ */
static int64_t cc_find(int64_t *parent, int64_t x)
{
    while (parent[x] != x) { parent[x] = parent[parent[x]]; x = parent[x]; }
    return x;
}

typedef struct { int64_t nvars, root; } cc_key;
static int cc_key_cmp(const void *a, const void *b)
{
    const cc_key *x = a, *y = b;
    if (x->nvars != y->nvars) return x->nvars < y->nvars ? -1 : 1;
    return x->root < y->root ? -1 : (x->root > y->root ? 1 : 0);
}

int64_t ro_components(const ro_problem *p, const uint8_t *assigned, int64_t *free_ptr, int64_t *free_vid,
                      int64_t *fac_ptr, int64_t *fac_id)
{
    const int64_t N = p->nvars, F = p->nfac;
    int64_t *parent = malloc((size_t)(N + 1) * sizeof(int64_t));
    int64_t *count = calloc((size_t)(N + 1), sizeof(int64_t));
    int64_t *comp_of = malloc((size_t)(N + 1) * sizeof(int64_t));
    for (int64_t v = 0; v < N; ++v) parent[v] = v;
    for (int64_t f = 0; f < F; ++f) {
        int64_t first = -1;
        for (int64_t k = 0, a = fac_arity(p, f); k < a; ++k) {
            const int64_t v = fac_var(p, f, k);
            if (assigned[v]) continue;
            if (first < 0) { first = v; continue; }
            int64_t ra = cc_find(parent, first), rb = cc_find(parent, v);
            if (ra == rb) continue;
            if (ra < rb) parent[rb] = ra; else parent[ra] = rb;   /* the smaller id stays root */
        }
    }
    int64_t ncomp = 0;
    for (int64_t v = 0; v < N; ++v) if (!assigned[v]) count[cc_find(parent, v)]++;
    cc_key *keys = malloc((size_t)(N + 1) * sizeof(cc_key));
    for (int64_t v = 0; v < N; ++v)
        if (!assigned[v] && parent[v] == v) { keys[ncomp].nvars = count[v]; keys[ncomp].root = v; ++ncomp; }
    qsort(keys, (size_t)ncomp, sizeof(cc_key), cc_key_cmp);
    free_ptr[0] = 0;
    for (int64_t c = 0; c < ncomp; ++c) { comp_of[keys[c].root] = c; free_ptr[c + 1] = free_ptr[c] + keys[c].nvars; }
    /* members in ascending id: a forward pass with per-component cursors */
    int64_t *cur = malloc((size_t)(ncomp + 1) * sizeof(int64_t));
    for (int64_t c = 0; c < ncomp; ++c) cur[c] = free_ptr[c];
    for (int64_t v = 0; v < N; ++v) if (!assigned[v]) free_vid[cur[comp_of[cc_find(parent, v)]]++] = v;
    for (int64_t c = 0; c <= ncomp; ++c) fac_ptr[c] = 0;
    for (int pass = 0; pass < 2; ++pass) {
        if (pass == 1) {
            for (int64_t c = 0; c < ncomp; ++c) fac_ptr[c + 1] += fac_ptr[c];
            for (int64_t c = 0; c < ncomp; ++c) cur[c] = fac_ptr[c];
        }
        for (int64_t f = 0; f < F; ++f) {
            int64_t r = -1;
            for (int64_t k = 0, a = fac_arity(p, f); k < a && r < 0; ++k) {
                const int64_t v = fac_var(p, f, k);
                if (!assigned[v]) r = cc_find(parent, v);
            }
            if (r < 0) continue;   /* every variable assigned: a constant, in no component */
            if (pass == 0) fac_ptr[comp_of[r] + 1]++; else fac_id[cur[comp_of[r]]++] = f;
        }
    }
    free(parent); free(count); free(comp_of); free(keys); free(cur);
    return ncomp;
}
