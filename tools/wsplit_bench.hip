// tools/wsplit_bench.hip — where the row-split F(2x2,3x3) kernel's time goes: wall time of the LIBRARY kernel
// conv_wino_split_k (rerevst-code_amd/csrc/conv_wino_split.h) for the epilogues the per-frame path uses.  No kernel
// lives here: ablations are the library header compiled with -DWSPLIT_ABL=n (1 no LDS-DMA, 2 no K-loop barriers, 4 no
// stores, 32 no epilogue), one binary per value:
//   for v in 0 1 2 4 32; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -DWSPLIT_ABL=$v tools/wsplit_bench.hip -o tools/bin/wsplit_bench_$v; done
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <math.h>
#include "../rerevst-code_amd/csrc/conv_mfma.h"
#include "../rerevst-code_amd/csrc/conv_wino.h"
#include "../rerevst-code_amd/csrc/conv_wino_split.h"
#include "../rerevst-code_amd/csrc/prep_kernels.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int EPI>
float run(ConvP p, int iters) {
    p.tiles_x = (p.W + 15) / 16; p.tiles_y = (p.H + 15) / 16;
    const int slabs = p.Cout / 32;
    const int items = p.tiles_x * p.tiles_y * p.B * slabs;
    dim3 grid(items < 256 ? items : 256, 1);
    p.xcd_slabs = (grid.x % 8 == 0 && (grid.x / 8) % slabs == 0) ? 1 : 0;
    CK(hipFuncSetAttribute((const void*)conv_wino_split_k<EPI, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, WSPLIT_SMEM_BYTES));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((conv_wino_split_k<EPI, 0>), grid, dim3(512), WSPLIT_SMEM_BYTES, 0, p);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((conv_wino_split_k<EPI, 0>), grid, dim3(512), WSPLIT_SMEM_BYTES, 0, p);
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
    const size_t res_f = (size_t)B * (H + 2) * (W + 2) * Cout + 4096;
    float *in, *out, *w, *wraw, *bias, *n1, *sty, *res;
    CK(hipMalloc(&in, in_f * 4)); CK(hipMalloc(&out, out_f * 4)); CK(hipMalloc(&res, res_f * 4));
    CK(hipMalloc(&w, (size_t)Cout * Cin * 16 * 4)); CK(hipMalloc(&wraw, (size_t)Cout * Cin * 9 * 4)); CK(hipMalloc(&bias, Cout * 4));
    CK(hipMalloc(&n1, 4 * Cout * 4)); CK(hipMalloc(&sty, 2 * Cout * 4));
    std::vector<float> hin(in_f), hw((size_t)Cout * Cin * 9), hn(4 * Cout), hs(2 * Cout), hr(res_f);
    for (auto& v : hin) v = (rand() / (float)RAND_MAX) - 0.5f;
    for (auto& v : hr) v = (rand() / (float)RAND_MAX) - 0.5f;
    for (auto& v : hw) v = ((rand() / (float)RAND_MAX) - 0.5f) * 0.05f;
    for (int c = 0; c < Cout; ++c) { hn[c] = 0.1f; hn[Cout + c] = 1.5f; hn[2 * Cout + c] = -3.f; hn[3 * Cout + c] = 3.f; hs[c] = 0.2f; hs[Cout + c] = 0.9f; }
    CK(hipMemcpy(in, hin.data(), in_f * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(res, hr.data(), res_f * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(wraw, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(n1, hn.data(), hn.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(sty, hs.data(), hs.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(pack_wino_k, dim3(4096), dim3(256), 0, 0, (const float*)wraw, w, Cout, Cin, 0, (const float*)nullptr);
    CK(hipMemset(bias, 0, Cout * 4)); CK(hipMemset(out, 0, out_f * 4));
    ConvP p{};
    p.in = in; p.Hi = H; p.Wi = W; p.Cin = Cin; p.out = out; p.H = H; p.W = W; p.Cout = Cout; p.B = B; p.in_bstride0 = 1;
    p.wpk = w; p.bias = bias; p.n1 = n1; p.n2 = n1; p.sty = sty; p.res = res;
    p.Hr = (EPI & E_RES_UPS) ? H / 2 : H; p.Wr = (EPI & E_RES_UPS) ? W / 2 : W;
    const double fl = 2.0 * B * H * W * (double)Cin * Cout * 4;      // executed (Winograd) FLOPs
    run<EPI>(p, 40);      // clocks up
    float t = 1e9f;
    for (int rep = 0; rep < 3; ++rep) t = fminf(t, run<EPI>(p, 10));
    printf("%-24s EPI %3d WSPLIT_ABL %2d: %.4f ms = %.1f TF executed (%.3f of the fp32-MFMA peak)\n", name, EPI, WSPLIT_ABL, t, fl / t / 1e9, fl / t / 1e9 / 157.3);
    for (float* q : {in, out, w, wraw, bias, n1, sty, res}) CK(hipFree(q));
}

int main() {
    constexpr int E54 = E_LRELU | E_NORM1 | E_RES_UPS | E_NORM2;
    layer<E54>("64->64 @640^2 B8", 8, 640, 640, 64, 64);
    layer<E_RELU | E_POOL>("64->64 @640^2 B8", 8, 640, 640, 64, 64);
    layer<E54>("128->128 @320^2 B8", 8, 320, 320, 128, 128);
    layer<E_RELU | E_POOL>("128->128 @320^2 B8", 8, 320, 320, 128, 128);
    layer<E54>("256->256 @160^2 B8", 8, 160, 160, 256, 256);
    layer<E_RELU>("256->256 @160^2 B8", 8, 160, 160, 256, 256);
    layer<E_RELU | E_NORM1>("256->512 @80^2 B8", 8, 80, 80, 256, 512);
    layer<E_RES>("32->512 @80^2 B8", 8, 80, 80, 32, 512);                  // KernelFilter up-conv (two chunks per item)
    layer<E_RES>("32->512 @144^2 B2", 2, 144, 144, 32, 512);
    return 0;
}
