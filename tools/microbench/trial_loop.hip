// trial_loop.hip -- the factor phase of the streaming solver's line-search trial (solver_ptm.hpp: eval_line) on its own:
// w waves per SIMD, each lane a point, slot after slot a factor against a camera's trial records in LDS, the (camera,
// observation) entries streamed from HBM a block of two slots ahead.  What bounds it?  Variants leave one ingredient out.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o trial_loop trial_loop.hip && ./trial_loop
// MODE 0 the loop as it is; 1 no stream from HBM (indices from the lane, observation constant); 2 every lane reads camera 0
// (LDS broadcast); 3 the records of ONE camera held in registers (no LDS reads in the loop); 4 as 0 with the reciprocal by
// v_rcp_f64 + two Newton steps instead of the IEEE division; 5 value only (no slope); 6 no stream, every lane of a wave the
// same camera, another one each slot (LDS broadcast); 7 no stream, ten lanes a camera, neighbours consecutive cameras (what
// the plan's order gives the solver on the bench's components: no bank conflicts); 8 as 7 with the stream; 9 as 0 with round
// 4's direction record (the matrix dR: sixteen doubles, eight reads -- the library's is [w dt df dk1 dk2], ten doubles, five
// reads, since round 5); 10 as 8 with that record; 11 as 8 with the stream asked for two blocks (four slots) ahead;
// 12 as 8 with the wave's priority set by the work it has left (s_setprio: four levels), 13 as 12 with eight steps (level = step & 3... no: two bits of the count).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../../rdis_amd/csrc/factors.hpp"
using namespace rdis_hip;

constexpr int NCAM = 49, TS = 18, SLOTS = 48;

template <int MODE>
__global__ void __launch_bounds__(1024) run(long long* out, double* sink, const short* pcam, const double2* pobs) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* CTR = lds;
    double* CDR = lds + TS * NCAM;
    for (int i = threadIdx.x; i < TS * NCAM; i += blockDim.x) {
        const int c = i / TS, k = i % TS;
        double v = 0.0, d = 0.0;
        if (k == 0 || k == 5 || k == 10) v = 0.9 + 0.001 * c; else if (k < 12 && (k & 3) != 3) v = 0.01 * (k - 5);
        if (k == 3 || k == 7) v = 0.1; if (k == 11) v = -5.0 - 0.01 * c;
        if (k == 12) v = 400.0; if (k == 13) v = -1e-7; if (k == 14) v = 1e-13;
        if (k < 15) d = 1e-3 * (k + 1);
        CTR[i] = v; CDR[i] = d;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const long long base = ((long long)blockIdx.x * nw + wave) * SLOTS * 64;
    double x[3] = {0.1 + 0.001 * lane, -0.2, 0.3}, dp[3] = {1e-3, 2e-3, -1e-3};
    double af = 0.0, as = 0.0;
    __syncthreads();
    int c0, c1; double2 o0, o1;
    auto load = [&](int s, int& a, int& b, double2& oa, double2& ob) {
        if (MODE == 6) { a = (7 * s) % NCAM; b = (7 * s + 3) % NCAM; oa = make_double2(1.0, 2.0); ob = make_double2(-1.0, 0.5); }
        else if (MODE == 7) { a = (lane / 10 + s) % NCAM; b = (lane / 10 + s + 1) % NCAM; oa = make_double2(1.0, 2.0); ob = make_double2(-1.0, 0.5); }
        else if (MODE == 8 || MODE >= 10) { a = (lane / 10 + s) % NCAM; b = (lane / 10 + s + 1) % NCAM; oa = pobs[base + 64 * s + lane]; ob = pobs[base + 64 * (s + 1) + lane];
                              a += pcam[base + 64 * s + lane] >> 12; b += pcam[base + 64 * (s + 1) + lane] >> 12; }
        else if (MODE == 1) { a = (lane + s) % NCAM; b = (lane + 7 * s + 3) % NCAM; oa = make_double2(1.0, 2.0); ob = make_double2(-1.0, 0.5); }
        else { a = pcam[base + 64 * s + lane]; b = pcam[base + 64 * (s + 1) + lane]; oa = pobs[base + 64 * s + lane]; ob = pobs[base + 64 * (s + 1) + lane]; }
    };
    load(0, c0, c1, o0, o1);
    int c2 = 0, c3 = 0; double2 o2 = o0, o3 = o1;
    if (MODE == 11) load(2, c2, c3, o2, o3);
    double TRr[CAM_TRIAL], DRr[CAM_DIR];
    if (MODE == 3) { for (int k = 0; k < CAM_TRIAL; ++k) TRr[k] = CTR[TS * (lane % NCAM) + k]; for (int k = 0; k < CAM_DIR; ++k) DRr[k] = CDR[TS * (lane % NCAM) + k]; }
    const long long t0 = clock64();
    for (int s = 0; s < SLOTS; s += 2) {
        if (MODE == 12) {   // the more slots are left, the higher the wave's priority: the SIMD's laggard goes first
            const int left = (SLOTS - s) * 4 / (SLOTS + 1);
            if (left >= 3) __builtin_amdgcn_s_setprio(3); else if (left == 2) __builtin_amdgcn_s_setprio(2);
            else if (left == 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
        }
        if (MODE == 13) {   // ... by the low bits of the count of blocks left: a wave one block behind is one level up (mod 4)
            const int left = ((SLOTS - s) >> 1) & 3;
            if (left == 3) __builtin_amdgcn_s_setprio(3); else if (left == 2) __builtin_amdgcn_s_setprio(2);
            else if (left == 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
        }
        int b0 = c0, b1 = c1; double2 p0 = o0, p1 = o1;
        asm volatile("" : "+v"(b0), "+v"(b1), "+v"(p0.x), "+v"(p0.y), "+v"(p1.x), "+v"(p1.y));
        if (MODE == 11) {
            c0 = c2; c1 = c3; o0 = o2; o1 = o3;
            asm volatile("" : "+v"(c0), "+v"(c1), "+v"(o0.x), "+v"(o0.y), "+v"(o1.x), "+v"(o1.y));
            load(s + 4 < SLOTS ? s + 4 : s, c2, c3, o2, o3);
        } else
        load(s + 2 < SLOTS ? s + 2 : s, c0, c1, o0, o1);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            int cc = k ? b1 : b0;
            const double2 o = k ? p1 : p0;
            if (MODE == 2) cc = 0;
            double TR[CAM_TRIAL], DR[CAM_DIR];
            if (MODE == 3) { for (int q = 0; q < CAM_TRIAL; ++q) TR[q] = TRr[q]; for (int q = 0; q < CAM_DIR; ++q) DR[q] = DRr[q]; TR[3] += 1e-9 * cc; }
            else {
                const double2* tc = reinterpret_cast<const double2*>(CTR + TS * cc);
#pragma unroll
                for (int q = 0; q < CAM_TRIAL / 2; ++q) { const double2 v = tc[q]; TR[2 * q] = v.x; TR[2 * q + 1] = v.y; }
            }
            BaTrial t;
            if (MODE == 4) {
                RDIS_FACTORS_FP_CONTRACT
                const double P0 = TR[0] * x[0] + (TR[1] * x[1] + (TR[2] * x[2] + TR[3]));
                const double P1 = TR[4] * x[0] + (TR[5] * x[1] + (TR[6] * x[2] + TR[7]));
                const double P2 = TR[8] * x[0] + (TR[9] * x[1] + (TR[10] * x[2] + TR[11]));
                double r = __builtin_amdgcn_rcp(P2);
                r = r + r * (1.0 - P2 * r);
                t.iz = r + r * (1.0 - P2 * r);
                t.pp0 = -P0 * t.iz; t.pp1 = -P1 * t.iz;
                t.r2 = t.pp0 * t.pp0 + t.pp1 * t.pp1;
                t.dstn = 1.0 + t.r2 * (TR[13] + TR[14] * t.r2);
                t.fd = TR[12] * t.dstn;
                t.res0 = t.fd * t.pp0 - o.x; t.res1 = t.fd * t.pp1 - o.y;
                af += (t.res0 * t.res0 + t.res1 * t.res1) * 0.5;
            } else af += ba_trial_value(TR, x, o.x, o.y, t);
            if (MODE == 9 || MODE == 10) {   // round 4's direction record: the matrix dR and dt, sixteen doubles -- eight reads
                RDIS_FACTORS_FP_CONTRACT
                const double2* dc = reinterpret_cast<const double2*>(CDR + TS * cc);
                double D[16];
#pragma unroll
                for (int q = 0; q < 8; ++q) { const double2 v = dc[q]; D[2 * q] = v.x; D[2 * q + 1] = v.y; }
                double dP0 = TR[0] * dp[0] + (TR[1] * dp[1] + TR[2] * dp[2]);
                double dP1 = TR[4] * dp[0] + (TR[5] * dp[1] + TR[6] * dp[2]);
                double dP2 = TR[8] * dp[0] + (TR[9] * dp[1] + TR[10] * dp[2]);
                dP0 += D[0] * x[0] + (D[1] * x[1] + (D[2] * x[2] + D[3]));
                dP1 += D[4] * x[0] + (D[5] * x[1] + (D[6] * x[2] + D[7]));
                dP2 += D[8] * x[0] + (D[9] * x[1] + (D[10] * x[2] + D[11]));
                const double dpp0 = -(dP0 + t.pp0 * dP2) * t.iz;
                const double dpp1 = -(dP1 + t.pp1 * dP2) * t.iz;
                const double dr2 = 2.0 * (t.pp0 * dpp0 + t.pp1 * dpp1);
                double ddst = dr2 * (TR[13] + 2.0 * TR[14] * t.r2);
                ddst += t.r2 * (D[13] + D[14] * t.r2);
                const double scale = D[12] * t.dstn + TR[12] * ddst;
                const double dpix0 = scale * t.pp0 + t.fd * dpp0;
                const double dpix1 = scale * t.pp1 + t.fd * dpp1;
                as += t.res0 * dpix0 + t.res1 * dpix1;
            } else if (MODE != 5) {
                if (MODE != 3) {
                    const double2* dc = reinterpret_cast<const double2*>(CDR + TS * cc);
#pragma unroll
                    for (int q = 0; q < CAM_DIR / 2; ++q) { const double2 v = dc[q]; DR[2 * q] = v.x; DR[2 * q + 1] = v.y; }
                }
                as += ba_trial_slope<false>(t, TR, DR, x, dp);
            }
        }
    }
    const long long t1 = clock64();
    __syncthreads();
    const long long t2 = clock64();
    if (af + as == 12345.678) sink[0] = af;
    if (blockIdx.x == 0 && lane == 0) { out[2 * wave] = t1 - t0; out[2 * wave + 1] = t2 - t0; }
}

template <int MODE>
static void go(const char* name, long long* d_out, double* d_sink, const short* pcam, const double2* pobs) {
    for (int w = 1; w <= 4; ++w) {
        hipMemset(d_out, 0, 64 * sizeof(long long));
        run<MODE><<<256, 256 * w, 2 * TS * NCAM * sizeof(double)>>>(d_out, d_sink, pcam, pobs);
        hipDeviceSynchronize();
        long long c[64];
        hipMemcpy(c, d_out, sizeof(c), hipMemcpyDeviceToHost);
        long long mn = 1ll << 62, mx = 0;
        for (int i = 0; i < 4 * w; ++i) { if (c[2 * i] < mn) mn = c[2 * i]; if (c[2 * i] > mx) mx = c[2 * i]; }
        std::printf("%-34s waves/SIMD %d: per slot and wave %5.0f .. %5.0f cycles; the SIMD takes a slot every %4.0f cycles\n", name, w,
                    (double)mn / SLOTS, (double)mx / SLOTS, (double)c[1] / (SLOTS * w));
    }
}

int main() {
    const size_t n = (size_t)256 * 16 * SLOTS * 64 + 256;
    std::vector<short> hc(n); std::vector<double2> ho(n);
    unsigned s = 12345u;
    for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; hc[i] = (short)((s >> 8) % NCAM); ho[i] = make_double2(1.0 + (s & 255) * 0.01, -2.0); }
    short* pcam; double2* pobs; long long* d_out; double* d_sink;
    hipMalloc(&pcam, n * sizeof(short)); hipMalloc(&pobs, n * sizeof(double2)); hipMalloc(&d_out, 64 * sizeof(long long)); hipMalloc(&d_sink, 64);
    hipMemcpy(pcam, hc.data(), n * sizeof(short), hipMemcpyHostToDevice);
    hipMemcpy(pobs, ho.data(), n * sizeof(double2), hipMemcpyHostToDevice);
    go<0>("the loop", d_out, d_sink, pcam, pobs);
    go<1>("no stream from HBM", d_out, d_sink, pcam, pobs);
    go<2>("every lane camera 0", d_out, d_sink, pcam, pobs);
    go<3>("records in registers", d_out, d_sink, pcam, pobs);
    go<4>("reciprocal by rcp + 2 Newton steps", d_out, d_sink, pcam, pobs);
    go<5>("value only", d_out, d_sink, pcam, pobs);
    go<6>("no stream, a camera per wave", d_out, d_sink, pcam, pobs);
    go<7>("no stream, neighbouring cameras", d_out, d_sink, pcam, pobs);
    go<8>("neighbouring cameras", d_out, d_sink, pcam, pobs);
    go<9>("round 4's direction record (8 reads)", d_out, d_sink, pcam, pobs);
    go<10>("... neighbouring cameras", d_out, d_sink, pcam, pobs);
    go<11>("neighbouring cameras, stream 2 ahead", d_out, d_sink, pcam, pobs);
    go<12>("... priority by the work left", d_out, d_sink, pcam, pobs);
    go<13>("... priority by blocks left mod 4", d_out, d_sink, pcam, pobs);
    return 0;
}
