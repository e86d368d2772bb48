// factors.hpp -- per-factor arithmetic of the hot path as gfx950 device code.
//
// One lane evaluates one factor in fp64.  Replaces, on the device:
//   BundleAdjustmentFactor::evalFactor / computeGradient
//       (reference src/bundleadjust/BundleAdjustmentFactor.cpp:160-185, 351-554)
//   NonlinearProductFactor::evalFactor / getDerivative
//       (reference src/NonlinearProductFactor.cpp:186-209, 149-178; power():
//        src/util/numeric.cpp:12-23)
// The bundle-adjustment derivative is the adjoint (reverse) sweep of the
// Snavely projection -- about 140 flops instead of the ~420 of a
// variable-by-variable forward expansion -- derived from the model, not from
// the reference's expression tree.
#pragma once
#include <hip/hip_runtime.h>

// Contraction of a * b + c into one fused multiply-add inside the factor arithmetic: on (the product: a third fewer fp64
// instructions).  -DRDIS_FACTORS_NO_CONTRACT builds the library with every product rounded before it is added, like the
// reference's x86-64 build (g++ -O2 emits no FMA) -- a measurement build: over one-ulp-perturbed starts the end values
// after 25 CG iterations then have the oracle's distribution, while with contraction they have the distribution the
// ORACLE has when it is compiled with contraction (tests/golden/make_end_values.py).
#ifdef RDIS_FACTORS_NO_CONTRACT
#define RDIS_FACTORS_FP_CONTRACT _Pragma("clang fp contract(off)")
#else
#define RDIS_FACTORS_FP_CONTRACT _Pragma("clang fp contract(on)")
#endif

namespace rdis_hip {

// sin and cos of a non-negative angle in one pass: Cody-Waite reduction by pi/2 in three pieces
// (exact up to 2^20 quarter turns; rotation angles are a few radians, their domain +-1000 pi)
// and the fdlibm minimax kernels on [-pi/4, pi/4] with the reduction's tail carried through --
// below 1 ulp, like the library routine it replaces, at about a third of its instructions and
// without its large-argument branch.  Beyond the exact range the library is used.
// A 64-bit constant where it is used, in scalar registers.  Left to itself the compiler forms the polynomials' constants
// once per kernel in vector registers; in a large kernel they do not stay there (they cannot be re-formed by one
// instruction, so they are spilled) and every use becomes a load from scratch memory with a wait of its own -- twenty of
// them in a row made the cameras' records of a trial point three times as slow as their arithmetic (solver_ptm.hpp).
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ double kconst(double v) { asm volatile("" : "+s"(v)); return v; }
#define KC(x) ::rdis_hip::kconst(x)
#else
#define KC(x) (x)
#endif
// (Measurement builds, one switch at a time, for tracing where the device's population of end values parts from the oracle's --
// tools/gpu_probe_population.py, DESIGN.md section 6: -DRDIS_BISECT_LIBM_SINCOS the library's sincos in place of this routine;
// -DRDIS_BISECT_IEEE_DIV quotients where the unit axis and the perspective divide use one reciprocal each.)
__host__ __device__ __forceinline__ void sincos_angle(double x, double* sn, double* cs) {
RDIS_FACTORS_FP_CONTRACT
#ifdef RDIS_BISECT_LIBM_SINCOS
    sincos(x, sn, cs); return;
#endif
    if (!(x < 1.0e6)) { sincos(x, sn, cs); return; }
    const double fn = rint(x * KC(6.36619772367581382433e-01));
    const int n = (int)fn;
    // x - fn*pi/2 as y + yt: pi/2 = 1.57079632673412561417 + 6.07710050630396597660e-11 + 2.02226624879595063154e-21
    double t = __builtin_fma(-fn, KC(1.57079632673412561417e+00), x);       // exact (33-bit constant)
    double w = fn * KC(6.07710050630396597660e-11);
    const double r = t - w;
    w = __builtin_fma(fn, KC(2.02226624879595063154e-21), -((t - r) - w));
    const double y = r - w;
    const double yt = (r - y) - w;
    const double z = y * y, z2 = z * z;
    // sine kernel
    const double sr = KC(8.33333333332248946124e-03) + z * (KC(-1.98412698298579493134e-04) + z * KC(2.75573137070700676789e-06))
                    + z * z2 * (KC(-2.50507602534068634195e-08) + z * KC(1.58969099521155010221e-10));
    const double v = z * y;
    const double ks = y - ((z * (0.5 * yt - v * sr) - yt) - v * KC(-1.66666666666666324348e-01));
    // cosine kernel
    const double cr = z * (KC(4.16666666666666019037e-02) + z * (KC(-1.38888888888741095749e-03) + z * KC(2.48015872894767294178e-05)))
                    + (z2 * z2) * (KC(-2.75573143513906633035e-07) + z * (KC(2.08757232129817482790e-09) + z * KC(-1.13596475577881948265e-11)));
    const double hz = 0.5 * z, wc = 1.0 - hz;
    const double kc = wc + (((1.0 - wc) - hz) + (z * cr - y * yt));
    // quadrant n mod 4: (s, c), (c, -s), (-s, -c), (-c, s)
    const double s0 = (n & 1) ? kc : ks, c0 = (n & 1) ? ks : kc;
    *sn = (n & 2) ? -s0 : s0;
    *cs = ((n + 1) & 2) ? -c0 : c0;
}

// variable order inside a BA factor: [rx ry rz | tx ty tz | f k1 k2 | X Y Z]
// (reference BundleAdjustmentCommon.h:36-59)
struct BaFwd {
    double v0, v1, v2, theta, itheta, s, c, w0, w1, w2, d;
    double P0, P1, P2, iz, pp0, pp1, r2, dstn, res0, res1;
};

// The rotation part of the forward pass: angle, unit axis, sine and cosine of a camera's
// angle-axis vector.  It depends on the camera alone, so a launch whose components leave every
// camera constant computes it once per camera (camera_rotations_kernel) instead of once per
// factor and trial point; the arithmetic is the same either way.
__host__ __device__ __forceinline__ void ba_rotation(double r0, double r1, double r2, BaFwd& t) {
RDIS_FACTORS_FP_CONTRACT
    const double th2 = r0 * r0 + r1 * r1 + r2 * r2;
    t.theta = sqrt(th2);
    const bool rot = t.theta > 0.0;
    // unit axis (reference BundleAdjustmentCommon.h:81-93); one reciprocal instead of three quotients
    t.itheta = 1.0 / t.theta;
#ifdef RDIS_BISECT_IEEE_DIV
    t.v0 = rot ? r0 / t.theta : r0;
    t.v1 = rot ? r1 / t.theta : r1;
    t.v2 = rot ? r2 / t.theta : r2;
#else
    t.v0 = rot ? r0 * t.itheta : r0;
    t.v1 = rot ? r1 * t.itheta : r1;
    t.v2 = rot ? r2 * t.itheta : r2;
#endif
    if (rot) {
        sincos_angle(t.theta, &t.s, &t.c);
    } else {
        t.s = 0.0; t.c = 1.0;
    }
}

// ... and the rest, given the rotation fields of t: rotate, translate, project, distort, residual.
// x[0..2] are not read.
__host__ __device__ __forceinline__ double ba_project(const double (&x)[12], double ox, double oy, BaFwd& t) {
RDIS_FACTORS_FP_CONTRACT
    const double q0 = x[9], q1 = x[10], q2 = x[11];
    t.w0 = t.v1 * q2 - t.v2 * q1;
    t.w1 = t.v2 * q0 - t.v0 * q2;
    t.w2 = t.v0 * q1 - t.v1 * q0;
    if (t.theta > 0.0) {
        const double omc = 1.0 - t.c;
        t.d = t.v0 * q0 + t.v1 * q1 + t.v2 * q2;
        t.P0 = q0 * t.c + t.w0 * t.s + t.v0 * omc * t.d;
        t.P1 = q1 * t.c + t.w1 * t.s + t.v1 * omc * t.d;
        t.P2 = q2 * t.c + t.w2 * t.s + t.v2 * omc * t.d;
    } else {  // first-order rotation at theta == 0 (reference .cpp:304-329)
        t.d = 0.0;
        t.P0 = q0 + t.w0; t.P1 = q1 + t.w1; t.P2 = q2 + t.w2;
    }
    t.P0 += x[3]; t.P1 += x[4]; t.P2 += x[5];
    t.iz = 1.0 / t.P2;
#ifdef RDIS_BISECT_IEEE_DIV
    t.pp0 = -t.P0 / t.P2;
    t.pp1 = -t.P1 / t.P2;
#else
    t.pp0 = -t.P0 * t.iz;
    t.pp1 = -t.P1 * t.iz;
#endif
    t.r2 = t.pp0 * t.pp0 + t.pp1 * t.pp1;
    t.dstn = 1.0 + t.r2 * (x[7] + x[8] * t.r2);
    t.res0 = x[6] * t.dstn * t.pp0 - ox;
    t.res1 = x[6] * t.dstn * t.pp1 - oy;
    return (t.res0 * t.res0 + t.res1 * t.res1) * 0.5;
}

__host__ __device__ __forceinline__ double ba_forward(const double (&x)[12], double ox, double oy, BaFwd& t) {
    ba_rotation(x[0], x[1], x[2], t);
    return ba_project(x, ox, oy, t);
}

// Rotation record of a camera block as camera_rotations_kernel leaves it: seven doubles at the
// block's first variable id in a shadow array of x.
constexpr int ROT_V0 = 0, ROT_THETA = 3, ROT_ITHETA = 4, ROT_SIN = 5, ROT_COS = 6;
__host__ __device__ __forceinline__ void store_rotation(double r0, double r1, double r2, double* __restrict__ r) {
    BaFwd t;
    ba_rotation(r0, r1, r2, t);
    r[ROT_V0] = t.v0; r[ROT_V0 + 1] = t.v1; r[ROT_V0 + 2] = t.v2;
    r[ROT_THETA] = t.theta; r[ROT_ITHETA] = t.itheta; r[ROT_SIN] = t.s; r[ROT_COS] = t.c;
}
__host__ __device__ __forceinline__ void ba_load_rotation(const double* __restrict__ r, BaFwd& t) {
    t.v0 = r[ROT_V0]; t.v1 = r[ROT_V0 + 1]; t.v2 = r[ROT_V0 + 2];
    t.theta = r[ROT_THETA]; t.itheta = r[ROT_ITHETA]; t.s = r[ROT_SIN]; t.c = r[ROT_COS];
}

__device__ __forceinline__ double ba_eval(const double (&x)[12], double ox, double oy) {
    BaFwd t;
    return ba_forward(x, ox, oy, t);
}

// Adjoint (reverse) sweep of the projection: g = d(s0 * pix_x + s1 * pix_y) / dx for the forward
// state t.  With (s0, s1) = the residual this is the gradient of E = |res|^2 / 2; with unit
// seeds it yields the two rows of the residual's Jacobian.
__host__ __device__ __forceinline__ void ba_adjoint(const BaFwd& t, const double (&x)[12], double s0, double s1, double (&g)[12]) {
RDIS_FACTORS_FP_CONTRACT
    const double q0 = x[9], q1 = x[10], q2 = x[11];
    const double f = x[6];
    const double rp = s0 * t.pp0 + s1 * t.pp1;
    g[6] = t.dstn * rp;
    const double adst = f * rp;
    g[7] = adst * t.r2;
    g[8] = adst * t.r2 * t.r2;
    const double ar2 = adst * (x[7] + 2.0 * x[8] * t.r2);
    const double fd = f * t.dstn;
    const double app0 = fd * s0 + 2.0 * ar2 * t.pp0;
    const double app1 = fd * s1 + 2.0 * ar2 * t.pp1;
    const double a0 = -app0 * t.iz, a1 = -app1 * t.iz;
    const double a2 = -(app0 * t.pp0 + app1 * t.pp1) * t.iz;
    g[3] = a0; g[4] = a1; g[5] = a2;
    const double av = a0 * t.v0 + a1 * t.v1 + a2 * t.v2;
    const double qxa0 = q1 * a2 - q2 * a1, qxa1 = q2 * a0 - q0 * a2, qxa2 = q0 * a1 - q1 * a0;
    const double vxa0 = t.v1 * a2 - t.v2 * a1, vxa1 = t.v2 * a0 - t.v0 * a2, vxa2 = t.v0 * a1 - t.v1 * a0;
    if (t.theta > 0.0) {
        const double omc = 1.0 - t.c;
        const double k = omc * av;
        g[9] = a0 * t.c - vxa0 * t.s + t.v0 * k;   // R^T a
        g[10] = a1 * t.c - vxa1 * t.s + t.v1 * k;
        g[11] = a2 * t.c - vxa2 * t.s + t.v2 * k;
        const double aq = a0 * q0 + a1 * q1 + a2 * q2;
        const double aw = a0 * t.w0 + a1 * t.w1 + a2 * t.w2;
        const double gth = (av * t.d - aq) * t.s + aw * t.c;
        const double gv0 = t.s * qxa0 + omc * (a0 * t.d + q0 * av);
        const double gv1 = t.s * qxa1 + omc * (a1 * t.d + q1 * av);
        const double gv2 = t.s * qxa2 + omc * (a2 * t.d + q2 * av);
        const double vgv = t.v0 * gv0 + t.v1 * gv1 + t.v2 * gv2;
        g[0] = (gv0 - t.v0 * vgv) * t.itheta + t.v0 * gth;
        g[1] = (gv1 - t.v1 * vgv) * t.itheta + t.v1 * gth;
        g[2] = (gv2 - t.v2 * vgv) * t.itheta + t.v2 * gth;
    } else {
        g[9] = a0 - vxa0; g[10] = a1 - vxa1; g[11] = a2 - vxa2;
        g[0] = qxa0; g[1] = qxa1; g[2] = qxa2;
    }
}

// Forward-mode derivative of E along ONE direction d (the search direction of a line minimisation):
// the slope d/da E(x + a d) at the forward state t, without forming the 12 partials -- about two
// thirds of the adjoint sweep's instructions and no 12-entry gradient live in registers.  The
// batched solvers use it for line-search trials (they need the slope only; the full gradient, once
// per CG iteration, stays with the adjoint).  CAMFIX: the camera's nine entries of d are zero (and
// are not read).  Same model, differentiated in the other order: agrees with sum_k g_k d_k to rounding.
template <bool CAMFIX>
__host__ __device__ __forceinline__ double ba_slope_dir(const BaFwd& t, const double (&x)[12], const double (&d)[12]) {
RDIS_FACTORS_FP_CONTRACT
    const double q0 = x[9], q1 = x[10], q2 = x[11];
    const double e0 = d[9], e1 = d[10], e2 = d[11];          // direction of the point
    double dP0, dP1, dP2;
    if (t.theta > 0.0) {
        const double omc = 1.0 - t.c;
        // v x dq, v . dq
        double dw0 = t.v1 * e2 - t.v2 * e1, dw1 = t.v2 * e0 - t.v0 * e2, dw2 = t.v0 * e1 - t.v1 * e0;
        double ddot = t.v0 * e0 + t.v1 * e1 + t.v2 * e2;
        dP0 = e0 * t.c + dw0 * t.s + t.v0 * (omc * ddot);
        dP1 = e1 * t.c + dw1 * t.s + t.v1 * (omc * ddot);
        dP2 = e2 * t.c + dw2 * t.s + t.v2 * (omc * ddot);
        if constexpr (!CAMFIX) {
            // rotation vector r = theta v: d theta = v . dr, dv = (dr - v d theta) / theta
            const double dth = t.v0 * d[0] + t.v1 * d[1] + t.v2 * d[2];
            const double u0 = (d[0] - t.v0 * dth) * t.itheta, u1 = (d[1] - t.v1 * dth) * t.itheta, u2 = (d[2] - t.v2 * dth) * t.itheta;
            const double uw0 = u1 * q2 - u2 * q1, uw1 = u2 * q0 - u0 * q2, uw2 = u0 * q1 - u1 * q0;   // dv x q
            const double ud = u0 * q0 + u1 * q1 + u2 * q2;                                             // dv . q
            const double sd = t.s * dth, cd = t.c * dth;
            const double k = sd * t.d + omc * ud;
            const double od = omc * t.d;
            dP0 += uw0 * t.s + t.w0 * cd - q0 * sd + u0 * od + t.v0 * k + d[3];
            dP1 += uw1 * t.s + t.w1 * cd - q1 * sd + u1 * od + t.v1 * k + d[4];
            dP2 += uw2 * t.s + t.w2 * cd - q2 * sd + u2 * od + t.v2 * k + d[5];
        }
    } else {   // P = q + r x q + t
        dP0 = e0 + (t.v1 * e2 - t.v2 * e1);
        dP1 = e1 + (t.v2 * e0 - t.v0 * e2);
        dP2 = e2 + (t.v0 * e1 - t.v1 * e0);
        if constexpr (!CAMFIX) {
            dP0 += (d[1] * q2 - d[2] * q1) + d[3];
            dP1 += (d[2] * q0 - d[0] * q2) + d[4];
            dP2 += (d[0] * q1 - d[1] * q0) + d[5];
        }
    }
    const double dpp0 = -(dP0 + t.pp0 * dP2) * t.iz;
    const double dpp1 = -(dP1 + t.pp1 * dP2) * t.iz;
    const double dr2 = 2.0 * (t.pp0 * dpp0 + t.pp1 * dpp1);
    double ddst = dr2 * (x[7] + 2.0 * x[8] * t.r2);
    double scale = x[6] * ddst;          // d(f dstn) = df dstn + f ddstn
    if constexpr (!CAMFIX) {
        ddst += t.r2 * (d[7] + d[8] * t.r2);
        scale = d[6] * t.dstn + x[6] * ddst;
    }
    const double fd = x[6] * t.dstn;
    const double dpix0 = scale * t.pp0 + fd * dpp0;
    const double dpix1 = scale * t.pp1 + fd * dpp1;
    return t.res0 * dpix0 + t.res1 * dpix1;
}

// ---- a line-search trial in matrix form (solver_ptm.hpp) ----------------------------------------
// Everything about a camera that is the same for all its factors at one trial point is formed ONCE per camera and trial
// instead of once per factor: the rotation matrix R of its angle-axis vector (Rodrigues: R = c I + s [v]x + (1 - c) v v^T;
// I + [r]x at theta = 0, the reference's first-order branch, BundleAdjustmentFactor.cpp:304-329) and, for the slope of a
// line search, the derivative dR of that matrix along the camera's part of the search direction.  A factor then costs
//   P = R q + t (9 fused multiply-adds) ... E                                  about 30 fp64 operations + one division,
//   dP = w x (R q) + R e + dt (24) ... dE/da                                   about 45,
// with dR q written as a cross product: the derivative of a rotation along a direction rho of its angle-axis vector is
// dR = [w]x R, w = J rho (J the left Jacobian of the rotation group: (s/theta) I + (1 - s/theta) v v^T + ((1 - c)/theta) [v]x),
// so a camera's direction record is three numbers instead of a matrix: ten doubles -- five 16-byte LDS reads a factor
// instead of eight, which is what bounds the trial loop (tools/microbench/trial_loop.hip: 13 reads instead of 16 a factor,
// the SIMD takes a slot of 64 factors every 474 instead of 511 cycles at three waves, 531 instead of 604 at two),
// where the vector form above (ba_project + ba_slope_dir: cross products, dot products and the chain through the unit
// axis, per factor) takes about 150: the streaming solver's trials are bound by dependent fp64 issue, not by bytes.
// Same model, other association of the same sums: values and slopes agree with the vector form to a few ulp of their
// terms (tests/cpp/factors_forms_test.hip), far inside the 1e-12 the parity tests allow per evaluation.
// Records (read by the factors as 16-byte pairs):
//   TR = [R00 R01 R02 t0 | R10 R11 R12 t1 | R20 R21 R22 t2 | f k1 k2 -]        16 doubles
//   DR = [w0 w1 | w2 dt0 | dt1 dt2 | df dk1 | dk2 -]                           10 doubles
constexpr int CAM_TRIAL = 16;
constexpr int CAM_DIR = 10;
// x: the camera's nine values at the trial point [r t f k1 k2]; rot: ba_rotation(x[0..2])
__host__ __device__ __forceinline__ void ba_camera_trial(const BaFwd& rot, const double (&x)[9], double* __restrict__ TR) {
RDIS_FACTORS_FP_CONTRACT
    const double v0 = rot.v0, v1 = rot.v1, v2 = rot.v2, s = rot.s, c = rot.c;
    if (rot.theta > 0.0) {
        const double omc = 1.0 - c;
        const double a0 = omc * v0, a1 = omc * v1, a2 = omc * v2;
        TR[0] = c + a0 * v0;      TR[1] = a0 * v1 - s * v2;  TR[2] = a0 * v2 + s * v1;
        TR[4] = a1 * v0 + s * v2; TR[5] = c + a1 * v1;       TR[6] = a1 * v2 - s * v0;
        TR[8] = a2 * v0 - s * v1; TR[9] = a2 * v1 + s * v0;  TR[10] = c + a2 * v2;
    } else {   // P = q + r x q
        TR[0] = 1.0; TR[1] = -v2; TR[2] = v1;
        TR[4] = v2;  TR[5] = 1.0; TR[6] = -v0;
        TR[8] = -v1; TR[9] = v0;  TR[10] = 1.0;
    }
    TR[3] = x[3]; TR[7] = x[4]; TR[11] = x[5];
    TR[12] = x[6]; TR[13] = x[7]; TR[14] = x[8]; TR[15] = 0.0;
}
// d: the camera's nine entries of the search direction
__host__ __device__ __forceinline__ void ba_camera_trial_dir(const BaFwd& rot, const double (&d)[9], double* __restrict__ DR) {
RDIS_FACTORS_FP_CONTRACT
    const double v0 = rot.v0, v1 = rot.v1, v2 = rot.v2;
    if (rot.theta > 0.0) {
        const double dth = v0 * d[0] + v1 * d[1] + v2 * d[2];          // d theta = v . dr
        const double a = rot.s * rot.itheta, b = (1.0 - rot.c) * rot.itheta;
        const double l = (1.0 - a) * dth;
        DR[0] = a * d[0] + (l * v0 + b * (v1 * d[2] - v2 * d[1]));
        DR[1] = a * d[1] + (l * v1 + b * (v2 * d[0] - v0 * d[2]));
        DR[2] = a * d[2] + (l * v2 + b * (v0 * d[1] - v1 * d[0]));
    } else {   // R = I + [r]x at r = 0: d(r x q) = dr x q
        DR[0] = d[0]; DR[1] = d[1]; DR[2] = d[2];
    }
    DR[3] = d[3]; DR[4] = d[4]; DR[5] = d[5];
    DR[6] = d[6]; DR[7] = d[7]; DR[8] = d[8]; DR[9] = 0.0;
}
// the derivative of the rotation matrix itself along d (row-major 3 x 3; the gradient's chain, ba_rotation_gradient)
__host__ __device__ __forceinline__ void ba_rotation_matrix_dir(const BaFwd& rot, const double (&d)[3], double (&M)[9]) {
RDIS_FACTORS_FP_CONTRACT
    const double v0 = rot.v0, v1 = rot.v1, v2 = rot.v2, s = rot.s, c = rot.c;
    if (rot.theta > 0.0) {
        const double omc = 1.0 - c;
        const double dth = v0 * d[0] + v1 * d[1] + v2 * d[2];
        const double u0 = (d[0] - v0 * dth) * rot.itheta, u1 = (d[1] - v1 * dth) * rot.itheta, u2 = (d[2] - v2 * dth) * rot.itheta;   // dv
        const double ds = c * dth, dc = -(s * dth), domc = s * dth;
        const double a0 = omc * v0, a1 = omc * v1, a2 = omc * v2;
        const double b0 = domc * v0 + omc * u0, b1 = domc * v1 + omc * u1, b2 = domc * v2 + omc * u2;   // d(a)
        const double w0 = ds * v0 + s * u0, w1 = ds * v1 + s * u1, w2 = ds * v2 + s * u2;               // d(s v)
        M[0] = dc + b0 * v0 + a0 * u0;    M[1] = b0 * v1 + a0 * u1 - w2;  M[2] = b0 * v2 + a0 * u2 + w1;
        M[3] = b1 * v0 + a1 * u0 + w2;    M[4] = dc + b1 * v1 + a1 * u1;  M[5] = b1 * v2 + a1 * u2 - w0;
        M[6] = b2 * v0 + a2 * u0 - w1;    M[7] = b2 * v1 + a2 * u1 + w0;  M[8] = dc + b2 * v2 + a2 * u2;
    } else {
        M[0] = 0.0;   M[1] = -d[2]; M[2] = d[1];
        M[3] = d[2];  M[4] = 0.0;   M[5] = -d[0];
        M[6] = -d[1]; M[7] = d[0];  M[8] = 0.0;
    }
}
struct BaTrial {   // what the slope needs of the value's evaluation
    double iz, pp0, pp1, r2, dstn, fd, res0, res1;
};
// the factor's value at point q against the camera record TR; observation (ox, oy)
__host__ __device__ __forceinline__ double ba_trial_value(const double (&TR)[CAM_TRIAL], const double (&q)[3], double ox, double oy, BaTrial& t) {
RDIS_FACTORS_FP_CONTRACT
    const double P0 = TR[0] * q[0] + (TR[1] * q[1] + (TR[2] * q[2] + TR[3]));
    const double P1 = TR[4] * q[0] + (TR[5] * q[1] + (TR[6] * q[2] + TR[7]));
    const double P2 = TR[8] * q[0] + (TR[9] * q[1] + (TR[10] * q[2] + TR[11]));
    t.iz = 1.0 / P2;
    t.pp0 = -P0 * t.iz;
    t.pp1 = -P1 * t.iz;
    t.r2 = t.pp0 * t.pp0 + t.pp1 * t.pp1;
    t.dstn = 1.0 + t.r2 * (TR[13] + TR[14] * t.r2);
    t.fd = TR[12] * t.dstn;
    t.res0 = t.fd * t.pp0 - ox;
    t.res1 = t.fd * t.pp1 - oy;
    return (t.res0 * t.res0 + t.res1 * t.res1) * 0.5;
}
// ... and its slope along (camera direction as DR, point direction e); CAMFIX: the camera does not move (DR is not read)
template <bool CAMFIX>
__host__ __device__ __forceinline__ double ba_trial_slope(const BaTrial& t, const double (&TR)[CAM_TRIAL], const double (&DR)[CAM_DIR],
                                                          const double (&q)[3], const double (&e)[3]) {
RDIS_FACTORS_FP_CONTRACT
    double dP0, dP1, dP2;
    if constexpr (CAMFIX) {
        dP0 = TR[0] * e[0] + (TR[1] * e[1] + TR[2] * e[2]);
        dP1 = TR[4] * e[0] + (TR[5] * e[1] + TR[6] * e[2]);
        dP2 = TR[8] * e[0] + (TR[9] * e[1] + TR[10] * e[2]);
    } else {
        const double Q0 = TR[0] * q[0] + (TR[1] * q[1] + TR[2] * q[2]);      // R q
        const double Q1 = TR[4] * q[0] + (TR[5] * q[1] + TR[6] * q[2]);
        const double Q2 = TR[8] * q[0] + (TR[9] * q[1] + TR[10] * q[2]);
        dP0 = TR[0] * e[0] + (TR[1] * e[1] + (TR[2] * e[2] + (DR[1] * Q2 + (DR[3] - DR[2] * Q1))));
        dP1 = TR[4] * e[0] + (TR[5] * e[1] + (TR[6] * e[2] + (DR[2] * Q0 + (DR[4] - DR[0] * Q2))));
        dP2 = TR[8] * e[0] + (TR[9] * e[1] + (TR[10] * e[2] + (DR[0] * Q1 + (DR[5] - DR[1] * Q0))));
    }
    const double dpp0 = -(dP0 + t.pp0 * dP2) * t.iz;
    const double dpp1 = -(dP1 + t.pp1 * dP2) * t.iz;
    const double dr2 = 2.0 * (t.pp0 * dpp0 + t.pp1 * dpp1);
    double ddst = dr2 * (TR[13] + 2.0 * TR[14] * t.r2);
    double scale = TR[12] * ddst;          // d(f dstn) = df dstn + f ddstn
    if constexpr (!CAMFIX) {
        ddst += t.r2 * (DR[7] + DR[8] * t.r2);
        scale = DR[6] * t.dstn + TR[12] * ddst;
    }
    const double dpix0 = scale * t.pp0 + t.fd * dpp0;
    const double dpix1 = scale * t.pp1 + t.fd * dpp1;
    return t.res0 * dpix0 + t.res1 * dpix1;
}

// The factor's GRADIENT in matrix form (solver_ptm.hpp's gradient pass; round 5), from the value's evaluation t against the
// camera record TR at point q.  With a = dE/dP (the adjoint of the projection, the same expressions as ba_adjoint's):
//   the point's entries   dE/dq = R^T a                                        (nine multiply-adds)
//   the camera's          dE/dR = a q^T (nine products), dE/dt = a, dE/df, dE/dk1, dE/dk2 -- fifteen numbers that are
//                         LINEAR in the factor, so a camera's factors are summed first and the chain from dE/dR to the three
//                         rotation variables (dE/dr_k = <sum a q^T, dR/dr_k>, ba_rotation_gradient) is applied once per
//                         camera and gradient instead of once per factor: about 75 fp64 operations a factor where the adjoint
//                         sweep of the vector form (ba_project + ba_adjoint) takes about 185.
// gc = [a0 q0, a0 q1, a0 q2, a1 q0, ... a2 q2 | a0 a1 a2 | dE/df dE/dk1 dE/dk2]
constexpr int CAM_GRAD = 15;
__host__ __device__ __forceinline__ void ba_trial_adjoint(const BaTrial& t, const double (&TR)[CAM_TRIAL], const double (&q)[3],
                                                          double (&gq)[3], double (&gc)[CAM_GRAD]) {
RDIS_FACTORS_FP_CONTRACT
    const double rp = t.res0 * t.pp0 + t.res1 * t.pp1;
    gc[12] = t.dstn * rp;
    const double adst = TR[12] * rp;
    gc[13] = adst * t.r2;
    gc[14] = adst * t.r2 * t.r2;
    const double ar2 = adst * (TR[13] + 2.0 * TR[14] * t.r2);
    const double app0 = t.fd * t.res0 + 2.0 * ar2 * t.pp0;
    const double app1 = t.fd * t.res1 + 2.0 * ar2 * t.pp1;
    const double a0 = -app0 * t.iz, a1 = -app1 * t.iz;
    const double a2 = -(app0 * t.pp0 + app1 * t.pp1) * t.iz;
    gq[0] = TR[0] * a0 + (TR[4] * a1 + TR[8] * a2);
    gq[1] = TR[1] * a0 + (TR[5] * a1 + TR[9] * a2);
    gq[2] = TR[2] * a0 + (TR[6] * a1 + TR[10] * a2);
    gc[0] = a0 * q[0]; gc[1] = a0 * q[1]; gc[2] = a0 * q[2];
    gc[3] = a1 * q[0]; gc[4] = a1 * q[1]; gc[5] = a1 * q[2];
    gc[6] = a2 * q[0]; gc[7] = a2 * q[1]; gc[8] = a2 * q[2];
    gc[9] = a0; gc[10] = a1; gc[11] = a2;
}
// ... and the chain for one camera: M = the sum of its factors' a q^T (row-major 3 x 3), rot = ba_rotation of its angle-axis
// vector; gr[k] = sum_ij M_ij dR_ij/dr_k (the derivative of the rotation matrix along the k-th unit vector: ba_rotation_matrix_dir)
__host__ __device__ __forceinline__ void ba_rotation_gradient(const BaFwd& rot, const double (&M)[9], double (&gr)[3]) {
RDIS_FACTORS_FP_CONTRACT
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        double d[3] = {0.0, 0.0, 0.0}, D[9];
        d[k] = 1.0;
        ba_rotation_matrix_dir(rot, d, D);
        gr[k] = (M[0] * D[0] + M[1] * D[1] + M[2] * D[2]) + (M[3] * D[3] + M[4] * D[4] + M[5] * D[5]) + (M[6] * D[6] + M[7] * D[7] + M[8] * D[8]);
    }
}

// value + the 12 partials
__device__ __forceinline__ double ba_eval_grad(const double (&x)[12], double ox, double oy,
                                               double (&g)[12]) {
    BaFwd t;
    const double E = ba_forward(x, ox, oy, t);
    ba_adjoint(t, x, t.res0, t.res1, g);
    return E;
}

// the two pixel residuals and their Jacobian rows
__device__ __forceinline__ void ba_residual_jacobian(const double (&x)[12], double ox, double oy, double (&res)[2],
                                                     double (&jx)[12], double (&jy)[12]) {
    BaFwd t;
    ba_forward(x, ox, oy, t);
    res[0] = t.res0; res[1] = t.res1;
    ba_adjoint(t, x, 1.0, 0.0, jx);
    ba_adjoint(t, x, 0.0, 1.0, jy);
}

// ---- nonlinear product factor -------------------------------------------------
// (round 6) Two last-place differences from the reference's std::pow / std::sin / std::cos, so that a CPU can compute what the device
// computes (the tests' CPU checker has both as named switches): the third and fourth power by multiplication, and sine /
// cosine by sincos_angle above (below 1 ulp, one source for host and device) instead of the device library's routines, whose bits no
// host library promises.
__host__ __device__ __forceinline__ double nlp_power(double v, double e) {
    if (e == 0.0) return 1.0;
    if (e == 1.0) return v;
    if (e == 2.0) return v * v;
    if (e == 3.0) return v * v * v;
    if (e == 4.0) { const double q = v * v; return q * q; }
    return pow(v, e);
}
__host__ __device__ __forceinline__ double nlp_sin(double v) {
    if (!(fabs(v) < 1.0e6)) return sin(v);
    double sn, cs;
    sincos_angle(v, &sn, &cs);
    return sn;
}
__host__ __device__ __forceinline__ double nlp_cos(double v) {
    if (!(fabs(v) < 1.0e6)) return cos(v);
    double sn, cs;
    sincos_angle(v, &sn, &cs);
    return cs;
}

__host__ __device__ __forceinline__ double nlp_term(double v, double e, double k, bool sine) {
    if (k != 0.0) v -= k;
    if (e != 1.0) v = nlp_power(v, e);
    if (sine) v = nlp_sin(v);
    return v;
}

// d/dx of one term; `skip` reproduces the reference's "exponent 1, no sine => 1"
__host__ __device__ __forceinline__ double nlp_dterm(double v, double e, double k, bool sine) {
    v -= k;
    const double inner_e = nlp_power(v, e);
    double d = nlp_power(v, e - 1.0) * e;
    if (sine) d *= nlp_cos(inner_e);
    return d;
}

}  // namespace rdis_hip
