// tools/upw_bench.hip — the dominant kernel of the frame, conv_wino_k<E_LRELU | E_NORM1, ABL, 4, UPS = 1, SC = 1>
// (ResidualBlock.conv1 behind the nearest-x2 upsample + the 1x1 shortcut; rerevst-code_amd/csrc/conv_wino.h), on its three
// layers at eight 640 x 640 frames per launch, with the library kernel's own ablation switches (template ABL: 1 no LDS-DMA
// after the first stage, 2 no K-loop barriers, 4 no stores, 32 no epilogue, 128 the MFMA stream alone; the library
// instantiates ABL = 0 only).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/upw_bench.hip -o tools/bin/upw_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <math.h>
#include "../rerevst-code_amd/csrc/conv_mfma.h"
#include "../rerevst-code_amd/csrc/conv_wino.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
constexpr int EPI = E_LRELU | E_NORM1;
using Geo = WinoGeo<4, 1, 1>;

template <int ABL>
float run(ConvP p, int iters) {
    p.tiles_x = (p.W + 15) / 16; p.tiles_y = (p.H + 15) / 16;
    const int slabs = p.Cout / 32;
    const int items = p.tiles_x * p.tiles_y * p.B * slabs;
    dim3 grid(items < 512 ? items : 512, 1);       // two workgroups per CU
    p.xcd_slabs = (grid.x % 8 == 0 && (grid.x / 8) % slabs == 0) ? 1 : 0;
    CK(hipFuncSetAttribute((const void*)conv_wino_k<EPI, ABL, 4, 1, 1, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, Geo::SMEM));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((conv_wino_k<EPI, ABL, 4, 1, 1, 0>), grid, dim3(256), Geo::SMEM, 0, p);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((conv_wino_k<EPI, ABL, 4, 1, 1, 0>), grid, dim3(256), Geo::SMEM, 0, p);
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / iters;
}

static void layer(const char* name, int B, int Hi, int Wi, int Cin, int Cout) {
    const int H = 2 * Hi, W = 2 * Wi;
    const size_t in_f = (size_t)B * (Hi + 2) * (Wi + 2) * Cin + (size_t)40 * (Wi + 22) * Cin;
    const size_t out_f = (size_t)B * (H + 2) * (W + 2) * Cout + (size_t)40 * (W + 22) * Cout;
    const size_t sc_f = (size_t)B * (Hi + 2) * (Wi + 2) * Cout + 4096;
    float *in, *out, *sc, *w, *wraw, *wsc, *bias, *n1;
    CK(hipMalloc(&in, in_f * 4)); CK(hipMalloc(&out, out_f * 4)); CK(hipMalloc(&sc, sc_f * 4));
    CK(hipMalloc(&w, (size_t)Cout * Cin * 10 * 4)); CK(hipMalloc(&wraw, (size_t)Cout * Cin * 9 * 4)); CK(hipMalloc(&wsc, (size_t)Cout * Cin * 4));
    CK(hipMalloc(&bias, Cout * 4)); CK(hipMalloc(&n1, 4 * Cout * 4));
    std::vector<float> hin(in_f), hw((size_t)Cout * Cin * 9), hs((size_t)Cout * Cin), hn(4 * Cout);
    for (auto& v : hin) v = (rand() / (float)RAND_MAX) - 0.5f;
    for (auto& v : hw) v = ((rand() / (float)RAND_MAX) - 0.5f) * 0.05f;
    for (auto& v : hs) v = ((rand() / (float)RAND_MAX) - 0.5f) * 0.05f;
    for (int c = 0; c < Cout; ++c) { hn[c] = 0.1f; hn[Cout + c] = 1.5f; hn[2 * Cout + c] = -3.f; hn[3 * Cout + c] = 3.f; }
    CK(hipMemcpy(in, hin.data(), in_f * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(wraw, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(wsc, hs.data(), hs.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(n1, hn.data(), hn.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(bias, 0, Cout * 4)); CK(hipMemset(out, 0, out_f * 4)); CK(hipMemset(sc, 0, sc_f * 4));
    hipLaunchKernelGGL(pack_wino_k, dim3(4096), dim3(256), 0, 0, (const float*)wraw, w, Cout, Cin, 1, (const float*)wsc);
    CK(hipDeviceSynchronize());
    ConvP p{};
    p.in = in; p.Hi = Hi; p.Wi = Wi; p.Cin = Cin; p.out = out; p.H = H; p.W = W; p.Cout = Cout; p.B = B; p.in_bstride0 = 1;
    p.wpk = w; p.bias = bias; p.n1 = n1; p.n2 = n1; p.sty = n1; p.sc_out = sc;
    const double fl = 2.0 * B * H * W * (double)Cin * Cout * 2.5;      // executed: 9 positions + the shortcut per 2x2 outputs
    run<0>(p, 30);
    float t0 = 1e9f, t4 = 1e9f, t32 = 1e9f, t33 = 1e9f, t34 = 1e9f, t128 = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        t0 = fminf(t0, run<0>(p, 10)); t4 = fminf(t4, run<4>(p, 10)); t32 = fminf(t32, run<32>(p, 10));
        t33 = fminf(t33, run<32 | 1>(p, 10)); t34 = fminf(t34, run<32 | 2>(p, 10)); t128 = fminf(t128, run<32 | 128>(p, 10));
    }
    auto f = [&](float ms) { return fl / ms / 1e9 / 157.3; };
    printf("%-26s as is %.4f ms = %.3f of the fp32-MFMA peak (executed) | no stores %.3f | no epilogue %.3f -> and no LDS-DMA %.3f | and no barriers %.3f | MFMA stream alone %.3f\n",
           name, t0, f(t0), f(t4), f(t32), f(t33), f(t34), f(t128));
    for (float* q : {in, out, sc, w, wraw, wsc, bias, n1}) CK(hipFree(q));
}

int main() {
    layer("512->256 @80^2->160^2 B8", 8, 80, 80, 512, 256);      // slice4.conv1
    layer("256->128 @160^2->320^2 B8", 8, 160, 160, 256, 128);   // slice3.conv1
    layer("128->64 @320^2->640^2 B8", 8, 320, 320, 128, 64);     // slice2.conv1
    layer("128->64 @576^2->1152^2 B1", 1, 576, 576, 128, 64);    // config 5, slice2.conv1
    return 0;
}
