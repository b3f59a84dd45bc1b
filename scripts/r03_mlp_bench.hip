// r03_mlp_bench.hip -- standalone check + timing of the fused row-panel kernels k_mlp_fwd / k_mlp_bwd (panel_kernels.h):
// fp64 host reference at a small row count (every output incl. the records), then back-to-back timing at the CFFM-B1 size.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/r03_mlp_bench.hip -o build/r03_mlp_bench
#define CFFM_EXPERIMENTS 1
#include "../vss_cffm_amd/csrc/panel_kernels.h"
#include "r03_panel_experiments.h"
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
#include <string.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
static float frand() { return (float)rand() / RAND_MAX * 2.f - 1.f; }
template <class T> static T* dev(const std::vector<T>& h) { T* d; CK(hipMalloc(&d, h.size() * sizeof(T))); CK(hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice)); return d; }
template <class T> static T* devz(size_t n) { T* d; CK(hipMalloc(&d, n * sizeof(T))); CK(hipMemset(d, 0, n * sizeof(T))); return d; }
static std::vector<float> host(const float* d, size_t n) { std::vector<float> h(n); CK(hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost)); return h; }
static f32x4* pack(const float* dW, int N, int K, int nn) {
    f32x4* d; CK(hipMalloc(&d, (size_t)N * K * 4));
    hipLaunchKernelGGL(k_pnl_pack_weight, dim3((unsigned)(((long)N * K / 8 + 255) / 256)), dim3(256), 0, 0, dW, N, K, nn, d);
    CK(hipGetLastError());
    return d;
}
static void unsplit(const std::vector<float>& s, std::vector<double>& out) {   // split-4 storage -> values
    out.resize(s.size());
    for (size_t q = 0; q < s.size() / 4; ++q) {
        unsigned short h[4], l[4];
        memcpy(h, (const void*)&s[4 * q], 8); memcpy(l, (const void*)&s[4 * q + 2], 8);
        for (int e = 0; e < 4; ++e) { unsigned uh = (unsigned)h[e] << 16, ul = (unsigned)l[e] << 16; float fh, fl; memcpy((void*)&fh, (const void*)&uh, 4); memcpy((void*)&fl, (const void*)&ul, 4); out[4 * q + e] = (double)fh + fl; }
    }
}
struct Err { double e = 0, r = 0; void add(double got, double ref) { e = fmax(e, fabs(got - ref)); r = fmax(r, fabs(ref)); } double rel() const { return e / (r > 0 ? r : 1); } };

template <int MT, int D, bool WS = false>
static void run(int NP, bool check, int HW, bool with_act = true) {
    const int B = (NP + HW - 1) / HW;
    std::vector<float> ao((size_t)NP * 256), xt((size_t)B * 4 * HW * 256), wp(256 * 256), w1(1024 * 256), w2(256 * 1024), bp(256), b1(1024), b2(256), g2(256), be2(256), dout((size_t)NP * 256);
    for (auto& v : ao) v = frand(); for (auto& v : xt) v = frand() * 1.5f; for (auto& v : dout) v = frand();
    for (auto& v : wp) v = frand() * 0.08f; for (auto& v : w1) v = frand() * 0.08f; for (auto& v : w2) v = frand() * 0.05f;
    for (auto& v : bp) v = frand() * 0.1f; for (auto& v : b1) v = frand() * 0.1f; for (auto& v : b2) v = frand() * 0.1f;
    for (auto& v : g2) v = 1.f + 0.2f * frand(); for (auto& v : be2) v = 0.1f * frand();
    float *d_ao = dev(ao), *d_xt = dev(xt), *d_wp = dev(wp), *d_w1 = dev(w1), *d_w2 = dev(w2), *d_bp = dev(bp), *d_b1 = dev(b1), *d_b2 = dev(b2), *d_g2 = dev(g2), *d_be2 = dev(be2), *d_dout = dev(dout);
    const int grid = (NP + 16 * MT - 1) / (16 * MT);
    MlpFwdArgs f;
    f.ao = d_ao; f.xt = d_xt + 3L * HW * 256; f.xt_bs = 4L * HW * 256; f.rows_per_batch = HW;     // the target frame of a [B,4,HW,256] stack
    f.wp = pack(d_wp, 256, 256, 0); f.w1 = pack(d_w1, 1024, 256, 0); f.w2 = pack(d_w2, 256, 1024, 0);
    f.bp = d_bp; f.b1 = d_b1; f.b2 = d_b2; f.g2 = d_g2; f.be2 = d_be2;
    f.x1 = devz<float>((size_t)NP * 256); f.z2s = devz<float>((size_t)NP * 256); f.mean2 = devz<float>(NP); f.rstd2 = devz<float>(NP);
    f.hraw = devz<float>((size_t)NP * 1024); f.acts = devz<float>((size_t)NP * 1024); float* acts_keep = f.acts; if (!with_act) f.acts = nullptr; f.x2 = devz<float>((size_t)NP * 256); f.NP = NP;
    MlpBwdArgs b;
    b.dout = d_dout; b.hraw = f.hraw; b.b1 = d_b1; b.x1 = f.x1; b.mean2 = f.mean2; b.rstd2 = f.rstd2; b.g2 = d_g2;
    b.w2n = pack(d_w2, 256, 1024, 1); b.w1n = pack(d_w1, 1024, 256, 1); b.wpn = pack(d_wp, 256, 256, 1);
    b.dhs = devz<float>((size_t)NP * 1024); b.dx1 = devz<float>((size_t)NP * 256); b.dao = devz<float>((size_t)NP * 256);
    b.rec_b1 = devz<float>((size_t)grid * 1024); b.rec_ln = devz<float>((size_t)grid * 1024); b.NP = NP;
    auto kf = WS ? k_mlp_fwd_ws<MT, D> : k_mlp_fwd<MT, D>; auto kb = WS ? k_mlp_bwd_ws<MT, D> : k_mlp_bwd<MT, D>;
    const int lds = WS ? PNL_WS_FWD_LDS(MT) : PNL_FUSED_LDS(MT), ldsb = WS ? PNL_WS_BWD_LDS(MT) : PNL_FUSED_LDS(MT);
    const int PT = WS ? PNL_WS_THREADS : PNL_THREADS;
    CK(hipFuncSetAttribute((const void*)kf, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    CK(hipFuncSetAttribute((const void*)kb, hipFuncAttributeMaxDynamicSharedMemorySize, ldsb));
    hipLaunchKernelGGL(kf, dim3(grid), dim3(PT), lds, 0, f); CK(hipGetLastError());
    hipLaunchKernelGGL(kb, dim3(grid), dim3(PT), ldsb, 0, b); CK(hipGetLastError());
    CK(hipDeviceSynchronize());
    if (check) {
        auto x1 = host(f.x1, (size_t)NP * 256), x2 = host(f.x2, (size_t)NP * 256), hraw = host(f.hraw, (size_t)NP * 1024), mean = host(f.mean2, NP), rstd = host(f.rstd2, NP);
        auto dx1 = host(b.dx1, (size_t)NP * 256), dao = host(b.dao, (size_t)NP * 256), rb1 = host(b.rec_b1, (size_t)grid * 1024), rln = host(b.rec_ln, (size_t)grid * 1024);
        std::vector<double> z2, act, dh;
        unsplit(host(f.z2s, (size_t)NP * 256), z2); unsplit(host(acts_keep, (size_t)NP * 1024), act); unsplit(host(b.dhs, (size_t)NP * 1024), dh);
        Err ex1, ez2, eh, ea, ex2, emu, ers, edh, edx1, edao, eb1, eg, ebt, ec2, ecp;
        std::vector<double> sb1(1024, 0), sg(256, 0), sbt(256, 0), sc2(256, 0), scp(256, 0);
        for (int m = 0; m < NP; ++m) {
            double X1[256], Z[256], H[1024], A[1024], X2[256], DA[1024], DH[1024], DZ[256], DX[256];
            const float* xr = &xt[((size_t)(m / HW) * 4 + 3) * HW * 256 + (size_t)(m % HW) * 256];
            double mu = 0;
            for (int n = 0; n < 256; ++n) { double s = bp[n] + xr[n]; for (int k = 0; k < 256; ++k) s += (double)ao[(size_t)m * 256 + k] * wp[n * 256 + k]; X1[n] = s; mu += s; }
            mu /= 256; double var = 0; for (int n = 0; n < 256; ++n) var += (X1[n] - mu) * (X1[n] - mu); var /= 256;
            const double rs = 1 / sqrt(var + 1e-5);
            for (int n = 0; n < 256; ++n) Z[n] = (X1[n] - mu) * rs * g2[n] + be2[n];
            for (int j = 0; j < 1024; ++j) { double s = 0; for (int k = 0; k < 256; ++k) s += Z[k] * w1[j * 256 + k]; H[j] = s; const double u = s + b1[j]; A[j] = 0.5 * u * (1 + erf(u / sqrt(2.0))); }
            for (int n = 0; n < 256; ++n) { double s = b2[n] + X1[n]; for (int j = 0; j < 1024; ++j) s += A[j] * w2[n * 1024 + j]; X2[n] = s; }
            for (int j = 0; j < 1024; ++j) { double s = 0; for (int n = 0; n < 256; ++n) s += (double)dout[(size_t)m * 256 + n] * w2[n * 1024 + j]; DA[j] = s;
                const double u = H[j] + b1[j]; DH[j] = s * (0.5 * (1 + erf(u / sqrt(2.0))) + u * exp(-u * u / 2) / sqrt(2 * M_PI)); sb1[j] += DH[j]; }
            for (int k = 0; k < 256; ++k) { double s = 0; for (int j = 0; j < 1024; ++j) s += DH[j] * w1[j * 256 + k]; DZ[k] = s; }
            double m1 = 0, m2 = 0;
            for (int n = 0; n < 256; ++n) { const double xh = (X1[n] - mu) * rs; m1 += DZ[n] * g2[n]; m2 += DZ[n] * g2[n] * xh; sg[n] += DZ[n] * xh; sbt[n] += DZ[n]; }
            m1 /= 256; m2 /= 256;
            for (int n = 0; n < 256; ++n) { const double xh = (X1[n] - mu) * rs; DX[n] = (DZ[n] * g2[n] - m1 - xh * m2) * rs + dout[(size_t)m * 256 + n]; sc2[n] += dout[(size_t)m * 256 + n]; scp[n] += DX[n]; }
            emu.add(mean[m], mu); ers.add(rstd[m], rs);
            for (int n = 0; n < 256; ++n) { ex1.add(x1[(size_t)m * 256 + n], X1[n]); ez2.add(z2[(size_t)m * 256 + n], Z[n]); ex2.add(x2[(size_t)m * 256 + n], X2[n]); edx1.add(dx1[(size_t)m * 256 + n], DX[n]); }
            for (int j = 0; j < 1024; ++j) { eh.add(hraw[(size_t)m * 1024 + j], H[j]); ea.add(act[(size_t)m * 1024 + j], A[j]); edh.add(dh[(size_t)m * 1024 + j], DH[j]); }
            for (int k = 0; k < 256; ++k) { double s = 0; for (int n = 0; n < 256; ++n) s += DX[n] * wp[n * 256 + k]; edao.add(dao[(size_t)m * 256 + k], s); }
        }
        for (int j = 0; j < 1024; ++j) { double s = 0; for (int w = 0; w < grid; ++w) s += rb1[(size_t)w * 1024 + j]; eb1.add(s, sb1[j]); }
        for (int n = 0; n < 256; ++n) {
            double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
            for (int w = 0; w < grid; ++w) { s0 += rln[(size_t)w * 1024 + n]; s1 += rln[(size_t)w * 1024 + 256 + n]; s2 += rln[(size_t)w * 1024 + 512 + n]; s3 += rln[(size_t)w * 1024 + 768 + n]; }
            eg.add(s0, sg[n]); ebt.add(s1, sbt[n]); ec2.add(s2, sc2[n]); ecp.add(s3, scp[n]);
        }
        printf("WS=%d MT=%d D=%d NP=%d check (max|err|/max|ref|): x1 %.1e mean %.1e rstd %.1e z2 %.1e hraw %.1e act %.1e x2 %.1e | dh %.1e dx1 %.1e dao %.1e db1 %.1e dg %.1e dbeta %.1e db2 %.1e dbp %.1e\n",
               (int)WS, MT, D, NP, ex1.rel(), emu.rel(), ers.rel(), ez2.rel(), eh.rel(), ea.rel(), ex2.rel(), edh.rel(), edx1.rel(), edao.rel(), eb1.rel(), eg.rel(), ebt.rel(), ec2.rel(), ecp.rel());
    } else {
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        const int R = 30;
        float msf, msb;
        // rotating output buffers (8 sets, > the 256 MiB Infinity Cache): a step of the real layer never rewrites a warm buffer
        const int NROT = 8;
        float *rh[NROT], *ra[NROT], *rx1[NROT], *rz[NROT], *rx2[NROT];
        for (int q = 0; q < NROT; ++q) { rh[q] = devz<float>((size_t)NP * 1024); ra[q] = devz<float>((size_t)NP * 1024); rx1[q] = devz<float>((size_t)NP * 256); rz[q] = devz<float>((size_t)NP * 256); rx2[q] = devz<float>((size_t)NP * 256); }
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kf, dim3(grid), dim3(PT), lds, 0, f);
        CK(hipEventRecord(e0, 0)); for (int i = 0; i < R; ++i) hipLaunchKernelGGL(kf, dim3(grid), dim3(PT), lds, 0, f); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&msf, e0, e1));
        {
            float msr;
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < R; ++i) { MlpFwdArgs fr = f; const int q = i % NROT; fr.hraw = rh[q]; fr.acts = with_act ? ra[q] : nullptr; fr.x1 = rx1[q]; fr.z2s = rz[q]; fr.x2 = rx2[q]; hipLaunchKernelGGL(kf, dim3(grid), dim3(PT), lds, 0, fr); }
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&msr, e0, e1));
            printf("   k_mlp_fwd with rotating output buffers: %.2f us\n", msr * 1e3 / R);
        }
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kb, dim3(grid), dim3(PT), ldsb, 0, b);
        CK(hipEventRecord(e0, 0)); for (int i = 0; i < R; ++i) hipLaunchKernelGGL(kb, dim3(grid), dim3(PT), ldsb, 0, b); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&msb, e0, e1));
        printf("WS=%d MT=%d D=%d NP=%d grid=%d act-store=%d: k_mlp_fwd %.2f us   k_mlp_bwd %.2f us\n", (int)WS, MT, D, NP, grid, (int)with_act, msf * 1e3 / R, msb * 1e3 / R);
    }
}

int main(int argc, char** argv) {
    srand(2);
    const bool quick = argc > 1;     // any argument: only the checks
    printf("PNL_ABLATE=%d\n", PNL_ABLATE);
    if (!PNL_ABLATE) { run<2, 4>(500, true, 170); run<2, 2, true>(777, true, 300); run<2, 2, true>(500, true, 170); }
    if (quick) return 0;
    run<2, 4>(7200, false, 3600, true);
    run<2, 2, true>(7200, false, 3600, true);
    run<2, 2, true>(7200, false, 3600, false);
    return 0;
}
