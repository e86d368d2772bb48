// brent_step.hip -- how long is ONE reply-to-next-trial step of Brent's method with derivatives
// (the hot path of CgdMachine::next, minimizer.hpp) when its state stays in registers?
// One wave, a dependent loop: the trial point the step asks for is "evaluated" by a cheap analytic
// function and fed back.  Compared with tools/microbench/step_cost.hip (the machine as the solvers
// run it: state in LDS, request written for the other waves).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o brent_step brent_step.hip
#include <hip/hip_runtime.h>
#include <cfloat>
#include <cstdio>

struct BS { double a, b, x, w, v, fx, fw, fv, dx, dw, dv, d, e, uu; bool tiny; int it; };

__device__ __forceinline__ bool step(BS& S, double fu, double du, double& un) {
#pragma clang fp contract(off)
    const double TOL = 3.0e-8, ZEPS = DBL_EPSILON * 1.0e-3;
    const bool le = fu <= S.fx;
    const bool right = S.uu >= S.x, left = S.uu < S.x;
    const double a1 = le ? (right ? S.x : S.a) : (left ? S.uu : S.a);
    const double b1 = le ? (right ? S.b : S.x) : (left ? S.b : S.uu);
    const bool c1 = !le && (fu <= S.fw || S.w == S.x);
    const bool c2 = !le && !c1 && (fu < S.fv || S.v == S.x || S.v == S.w);
    const bool vw = le || c1;
    const double v1 = vw ? S.w : (c2 ? S.uu : S.v), fv1 = vw ? S.fw : (c2 ? fu : S.fv), dv1 = vw ? S.dw : (c2 ? du : S.dv);
    const double w1 = le ? S.x : (c1 ? S.uu : S.w), fw1 = le ? S.fx : (c1 ? fu : S.fw), dw1 = le ? S.dx : (c1 ? du : S.dw);
    const double x1 = le ? S.uu : S.x, fx1 = le ? fu : S.fx, dx1 = le ? du : S.dx;
    const double xm = 0.5 * (a1 + b1);
    const double tol1 = TOL * fabs(x1) + ZEPS;
    const double tol2 = 2.0 * tol1;
    const bool conv = fabs(x1 - xm) <= (tol2 - 0.5 * (b1 - a1));
    const bool big = fabs(S.e) > tol1;
    const double dflt = 2.0 * (b1 - a1);
    const double q1 = (w1 - x1) * dx1 / (dx1 - dw1);
    const double q2 = (v1 - x1) * dx1 / (dx1 - dv1);
    const double d1 = (dw1 != dx1) ? q1 : dflt;
    const double d2 = (dv1 != dx1) ? q2 : dflt;
    const double u1 = x1 + d1, u2 = x1 + d2;
    const bool ok1 = (a1 - u1) * (u1 - b1) > 0.0 && dx1 * d1 <= 0.0;
    const bool ok2 = (a1 - u2) * (u2 - b1) > 0.0 && dx1 * d2 <= 0.0;
    const double dsel = (ok1 && ok2) ? (fabs(d1) < fabs(d2) ? d1 : d2) : (ok1 ? d1 : d2);
    const bool accept = big && (ok1 || ok2) && (fabs(dsel) <= fabs(0.5 * S.e));
    const double ut = x1 + dsel;
    const double dacc = (ut - a1 < tol2 || b1 - ut < tol2) ? copysign(tol1, xm - x1) : dsel;
    const double ebis = (dx1 >= 0.0 ? a1 - x1 : b1 - x1);
    const double enew = accept ? S.d : ebis;
    const double dnew = accept ? dacc : 0.5 * ebis;
    const bool tn = !(fabs(dnew) >= tol1);
    un = tn ? x1 + copysign(tol1, dnew) : x1 + dnew;
    S.a = a1; S.b = b1; S.v = v1; S.fv = fv1; S.dv = dv1; S.w = w1; S.fw = fw1; S.dw = dw1; S.x = x1; S.fx = fx1; S.dx = dx1;
    S.e = enew; S.d = dnew; S.tiny = tn; S.uu = un; S.it += 1;
    return conv;
}

struct Req { int kind, flags, ncand; double cand[3]; int pre_tag, tr_tag; double a, b, pa, pb, pc, ta, tb, tc; };

// the same step with its state in LDS: loaded up front, stored at the end (as CgdMachine::hot does);
// MODE 1: every lane of the wave loads and stores; MODE 2: lane 0 only; MODE 3: lane 0 + a request record written
template <int MODE>
__global__ void __launch_bounds__(64) drive_lds(long long* cyc, double* out, int n) {
    __shared__ BS M;
    __shared__ Req Q[2];
    const bool lane0 = threadIdx.x == 0;
    if (lane0) M = BS{-1.618034, 1.0, 0.0, 0.0, 0.0, 10.0, 10.0, 10.0, -3.0, -3.0, -3.0, 0.0, 0.0, 0.5, false, 0};
    __syncthreads();
    double shift = 0.0371, curv = 40.0;
    double u = 0.5, acc = 0.0;
    long long total = 0; int steps = 0;
    for (int i = 0; i < n; ++i) {
        const double t = u - shift;
        const double fu = 10.0 - 3.0 * shift + curv * t * t, du = 2.0 * curv * t;
        const long long t0 = clock64();
        double un = 0.0; bool conv = false; int itn = 0;
        if (MODE == 1 || lane0) {
            BS S = M;
            conv = step(S, fu, du, un);
            M = S;
            itn = S.it;
            if (MODE == 3) {
                Req* q = &Q[i & 1];
                q->kind = 0; q->flags = 5; q->ncand = 0; q->pre_tag = 0; q->tr_tag = 0; q->a = un;
            }
        }
        const long long t1 = clock64();
        __syncthreads();
        un = M.uu; itn = M.it;
        conv = __builtin_amdgcn_readfirstlane((int)conv) != 0;
        total += t1 - t0; ++steps;
        acc += un;
        u = un;
        if (conv || itn > 60) {
            shift = 0.011 + 0.003 * (double)(i % 17);
            if (lane0) M = BS{-1.618034, 1.0, 0.0, 0.0, 0.0, 10.0, 10.0, 10.0, -3.0, -3.0, -3.0, 0.0, 0.0, 0.5, false, 0};
            __syncthreads();
            u = 0.5;
        }
    }
    if (threadIdx.x == 0) { cyc[0] = total; cyc[1] = steps; out[0] = acc; }
}

__global__ void __launch_bounds__(64) drive(long long* cyc, double* out, int n) {
    BS S{-1.618034, 1.0, 0.0, 0.0, 0.0, 10.0, 10.0, 10.0, -3.0, -3.0, -3.0, 0.0, 0.0, 0.5, false, 0};
    double shift = 0.0371, curv = 40.0;
    double u = S.uu, acc = 0.0;
    long long total = 0; int steps = 0;
    for (int i = 0; i < n; ++i) {
        const double t = u - shift;
        const double fu = 10.0 - 3.0 * shift + curv * t * t, du = 2.0 * curv * t;
        const long long t0 = clock64();
        double un;
        const bool conv = step(S, fu, du, un);
        const long long t1 = clock64();
        total += t1 - t0; ++steps;
        acc += un;
        u = un;
        if (conv || S.it > 60) {   // restart another line
            shift = 0.011 + 0.003 * (double)(i % 17);
            S = BS{-1.618034, 1.0, 0.0, 0.0, 0.0, 10.0, 10.0, 10.0, -3.0, -3.0, -3.0, 0.0, 0.0, 0.5, false, 0};
            u = S.uu;
        }
    }
    if (threadIdx.x == 0) { cyc[0] = total; cyc[1] = steps; out[0] = acc; }
}

int main() {
    long long* cyc; double* out;
    hipMalloc(&cyc, 64); hipMalloc(&out, 64);
    for (int rep = 0; rep < 2; ++rep) {
        drive<<<1, 64>>>(cyc, out, 20000);
        hipDeviceSynchronize();
        long long h[2]; double o;
        hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost); hipMemcpy(&o, out, 8, hipMemcpyDeviceToHost);
        printf("run %d: %lld steps, %.0f cycles each incl. two clock reads (checksum %.6f)\n", rep, h[1], (double)h[0] / h[1], o);
    }
    for (int mode = 1; mode <= 3; ++mode) {
        if (mode == 1) drive_lds<1><<<1, 64>>>(cyc, out, 20000);
        if (mode == 2) drive_lds<2><<<1, 64>>>(cyc, out, 20000);
        if (mode == 3) drive_lds<3><<<1, 64>>>(cyc, out, 20000);
        hipDeviceSynchronize();
        long long h[2]; double o;
        hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost); hipMemcpy(&o, out, 8, hipMemcpyDeviceToHost);
        printf("state in LDS, mode %d (1 all lanes, 2 lane 0, 3 lane 0 + request): %lld steps, %.0f cycles each (checksum %.6f)\n", mode, h[1], (double)h[0] / h[1], o);
    }
    return 0;
}
