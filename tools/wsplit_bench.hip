// tools/wsplit_bench.hip — where the row-split Winograd kernel's item overhead goes: wall time and per-phase cycle
// timeline of conv_wino_split_k for the epilogues the per-frame path uses, with ablations.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/wsplit_bench.hip -o gpurun_out/wsplit_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <math.h>
#include "../rerevst-code_amd/csrc/conv_mfma.h"
#include "../rerevst-code_amd/csrc/conv_wino.h"
#include "conv_wino_split_ab.h"
#include "../rerevst-code_amd/csrc/prep_kernels.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int EPI, int ABL>
float run(ConvP p, int iters, long long* dbg = nullptr) {
    p.xcd_slabs = 1;
    p.tiles_y = (p.H + 15) / 16;
    p.dbg = dbg;
    const int slabs = p.Cout / 32;
    int items = p.tiles_x * p.tiles_y * p.B * slabs;
    dim3 grid(items < 256 ? items : 256, 1);
    p.xcd_slabs = (grid.x % 8 == 0 && (grid.x / 8) % slabs == 0) ? 1 : 0;
    CK(hipFuncSetAttribute((const void*)conv_wino_split_ab_k<EPI, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, WSPLIT_SMEM_BYTES));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((conv_wino_split_ab_k<EPI, ABL>), grid, dim3(512), WSPLIT_SMEM_BYTES, 0, p);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((conv_wino_split_ab_k<EPI, ABL>), grid, dim3(512), WSPLIT_SMEM_BYTES, 0, p);
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / iters;
}

template <int EPI>
void layer(const char* name, int B, int H, int W, int Cin, int Cout) {
    const bool pool = EPI & E_POOL;
    const int Ho = pool ? H / 2 : H, Wo = pool ? W / 2 : W;
    const size_t in_f = (size_t)B * (H + 2) * (W + 2) * Cin + (size_t)40 * (W + 22) * Cin;
    const size_t out_f = (size_t)B * (Ho + 2) * (Wo + 2) * Cout + (size_t)40 * (W + 22) * Cout;
    const size_t res_f = (size_t)B * (H / 2 + 2) * (W / 2 + 2) * Cout + 4096;
    float *in, *out, *w, *wraw, *bias, *n1, *n2, *sty, *res;
    CK(hipMalloc(&in, in_f * 4)); CK(hipMalloc(&out, out_f * 4)); CK(hipMalloc(&res, res_f * 4));
    CK(hipMalloc(&w, (size_t)Cout * Cin * 16 * 4)); CK(hipMalloc(&wraw, (size_t)Cout * Cin * 9 * 4)); CK(hipMalloc(&bias, Cout * 4));
    CK(hipMalloc(&n1, 4 * Cout * 4)); CK(hipMalloc(&n2, 4 * Cout * 4)); CK(hipMalloc(&sty, 2 * Cout * 4));
    std::vector<float> hin(in_f), hw((size_t)Cout * Cin * 9), hn(4 * Cout), hs(2 * Cout), hr(res_f);
    for (auto& v : hin) v = (rand() / (float)RAND_MAX) - 0.5f;
    for (auto& v : hr) v = (rand() / (float)RAND_MAX) - 0.5f;
    for (auto& v : hw) v = ((rand() / (float)RAND_MAX) - 0.5f) * 0.05f;
    for (int c = 0; c < Cout; ++c) { hn[c] = 0.1f; hn[Cout + c] = 1.5f; hn[2 * Cout + c] = -3.f; hn[3 * Cout + c] = 3.f; hs[c] = 0.2f; hs[Cout + c] = 0.9f; }
    CK(hipMemcpy(in, hin.data(), in_f * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(res, hr.data(), res_f * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(wraw, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(n1, hn.data(), hn.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(n2, hn.data(), hn.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(sty, hs.data(), hs.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(pack_wino_k, dim3(4096), dim3(256), 0, 0, (const float*)wraw, w, Cout, Cin, 0, (const float*)nullptr);
    CK(hipMemset(bias, 0, Cout * 4)); CK(hipMemset(out, 0, out_f * 4));
    ConvP p{};
    p.in = in; p.Hi = H; p.Wi = W; p.Cin = Cin; p.out = out; p.H = H; p.W = W; p.Cout = Cout; p.B = B; p.in_bstride0 = 1;
    p.wpk = w; p.bias = bias; p.n1 = n1; p.n2 = n2; p.sty = sty; p.res = res; p.Hr = H / 2; p.Wr = W / 2;
    p.tiles_x = (W + 15) / 16;
    const double fl = 2.0 * B * H * W * (double)Cin * Cout * 4;      // executed (Winograd) FLOPs
    const int it = 10;
    run<EPI, 0>(p, 40);      // clocks up before anything is compared
    float t0 = 1e9f, t4 = 1e9f, t32 = 1e9f, t1 = 1e9f, t2 = 1e9f, t64 = 1e9f, t8 = 1e9f, t128 = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {      // round-robin, best of three: no variant profits from running later
        t0 = fminf(t0, run<EPI, 0>(p, it)); t4 = fminf(t4, run<EPI, 4>(p, it)); t32 = fminf(t32, run<EPI, 32>(p, it));
        t1 = fminf(t1, run<EPI, 1>(p, it)); t2 = fminf(t2, run<EPI, 2>(p, it)); t64 = fminf(t64, run<EPI, 64>(p, it));
        t8 = fminf(t8, run<EPI, 8>(p, it)); t128 = fminf(t128, run<EPI, 128>(p, it));
    }
    printf("%-24s EPI %3d: s_setprio(1) for waves 4-7: %.1f TF | for the odd wave of each pair: %.1f TF (as is %.1f)\n", name, EPI, fl / t8 / 1e9, fl / t128 / 1e9, fl / t0 / 1e9);
    printf("%-24s EPI %3d: %.4f ms = %.1f TF executed (%.3f of peak) | no stores %.1f | no residual loads %.1f | no epilogue %.1f | no loads %.1f | no K-loop barriers %.1f\n",
           name, EPI, t0, fl / t0 / 1e9, fl / t0 / 1e9 / 157.3, fl / t4 / 1e9, fl / t64 / 1e9, fl / t32 / 1e9, fl / t1 / 1e9, fl / t2 / 1e9);
    long long* dbg; CK(hipMalloc(&dbg, (size_t)256 * 8 * 6 * 8));
    for (int variant = 0; variant < 2; ++variant) {
        if (variant) run<EPI, 20>(p, 1, dbg); else run<EPI, 16>(p, 1, dbg);
        const int slabs = Cout / 32;
        const int items = p.tiles_x * ((H + 15) / 16) * B * slabs;
        const int g = items < 256 ? items : 256;
        std::vector<long long> h((size_t)g * 8 * 6);
        CK(hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost));
        double s6[6] = {0};
        for (size_t i = 0; i < h.size(); ++i) s6[i % 6] += h[i];
        const char* nm[6] = {"item setup", "chunks 1..n-1", "row sums+xch write", "reads+epilogue", "xch barrier", "chunk 0"};
        const double ipw = (double)items / g;
        double tot = 0;
        for (int k = 0; k < 6; ++k) tot += s6[k] / (g * 8) / ipw;
        printf("     timeline%s (clk/item):", variant ? " NO STORES" : "");
        for (int k : {0, 5, 1, 2, 4, 3}) printf(" %s %.0f |", nm[k], s6[k] / (g * 8) / ipw);
        printf(" total %.0f; %d chunks/item = %d MFMA clk, %.1f items/WG\n", tot, Cin / 16, Cin / 16 * 4096, ipw);
    }
    CK(hipFree(dbg));
    for (float* q : {in, out, w, wraw, bias, n1, n2, sty, res}) CK(hipFree(q));
}

int main() {
    constexpr int E54 = E_LRELU | E_NORM1 | E_RES_UPS | E_NORM2;
    layer<E54>("64->64 @640^2 B8", 8, 640, 640, 64, 64);
    layer<E_RELU | E_POOL>("64->64 @640^2 B8", 8, 640, 640, 64, 64);
    layer<E_RELU>("64->64 @640^2 B8", 8, 640, 640, 64, 64);
    layer<E54>("128->128 @320^2 B8", 8, 320, 320, 128, 128);
    layer<E_RELU | E_POOL>("128->128 @320^2 B8", 8, 320, 320, 128, 128);
    layer<E54>("256->256 @160^2 B8", 8, 160, 160, 256, 256);
    layer<E_RELU>("256->256 @160^2 B8", 8, 160, 160, 256, 256);
    return 0;
}
