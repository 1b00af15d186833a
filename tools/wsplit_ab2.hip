// tools/wsplit_ab2.hip — A/B of the library's row-split kernel (conv_wino_split_k) against the frozen round-2 form
// (tools/conv_wino_split_ab.h, ABL = 0): same inputs, outputs compared BIT FOR BIT (the rework only moves instructions),
// then timed round-robin.     hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/wsplit_ab2.hip -o /tmp/wsplit_ab2
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <math.h>
#include "../rerevst-code_amd/csrc/conv_mfma.h"
#include "../rerevst-code_amd/csrc/conv_wino.h"
#include "conv_wino_split_ab.h"
#include "../rerevst-code_amd/csrc/prep_kernels.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

static ConvP geom(ConvP p) {
    p.tiles_y = (p.H + 15) / 16;
    return p;
}
template <int EPI, int NEW>
float run(ConvP p, int iters) {
    p = geom(p);
    const int slabs = p.Cout / 32;
    int items = p.tiles_x * p.tiles_y * p.B * slabs;
    dim3 grid(items < 256 ? items : 256, 1);
    p.xcd_slabs = (grid.x % 8 == 0 && (grid.x / 8) % slabs == 0) ? 1 : 0;
    auto launch = [&]() {
        if (NEW) hipLaunchKernelGGL((conv_wino_split_k<EPI>), grid, dim3(512), WSPLIT_SMEM_BYTES, 0, p);
        else hipLaunchKernelGGL((conv_wino_split_ab_k<EPI, 0>), grid, dim3(512), WSPLIT_SMEM_BYTES, 0, p);
    };
    if (NEW) CK(hipFuncSetAttribute((const void*)conv_wino_split_k<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, WSPLIT_SMEM_BYTES));
    else CK(hipFuncSetAttribute((const void*)conv_wino_split_ab_k<EPI, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, WSPLIT_SMEM_BYTES));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) launch();
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    CK(hipGetLastError());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return ms / iters;
}

template <int EPI>
void layer(const char* name, int B, int H, int W, int Cin, int Cout) {
    const bool pool = EPI & E_POOL;
    const int Ho = pool ? H / 2 : H, Wo = pool ? W / 2 : W;
    const int Hr = (EPI & E_RES_UPS) ? H / 2 : H, Wr = (EPI & E_RES_UPS) ? W / 2 : W;
    const size_t in_f = (size_t)B * (H + 2) * (W + 2) * Cin + (size_t)40 * (W + 22) * Cin;
    const size_t out_f = (size_t)B * (Ho + 2) * (Wo + 2) * Cout + (size_t)40 * (W + 22) * Cout;
    const size_t res_f = (size_t)B * (Hr + 2) * (Wr + 2) * Cout + (size_t)40 * (W + 22) * Cout;
    float *in, *out, *out2, *w, *wraw, *bias, *n1, *n2, *sty, *res;
    CK(hipMalloc(&in, in_f * 4)); CK(hipMalloc(&out, out_f * 4)); CK(hipMalloc(&out2, out_f * 4)); CK(hipMalloc(&res, res_f * 4));
    CK(hipMalloc(&w, (size_t)Cout * Cin * 16 * 4)); CK(hipMalloc(&wraw, (size_t)Cout * Cin * 9 * 4)); CK(hipMalloc(&bias, Cout * 4));
    CK(hipMalloc(&n1, 4 * Cout * 4)); CK(hipMalloc(&n2, 4 * Cout * 4)); CK(hipMalloc(&sty, 2 * Cout * 4));
    std::vector<float> hin(in_f, 0.f), hw((size_t)Cout * Cin * 9), hn(4 * Cout), hs(2 * Cout), hr(res_f), hb(Cout);
    // valid pixels only: the ring and the slack stay zero, as in the library's tensors
    for (int b = 0; b < B; ++b)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x)
                for (int c = 0; c < Cin; ++c) hin[(((size_t)b * (H + 2) + y + 1) * (W + 2) + x + 1) * Cin + c] = (rand() / (float)RAND_MAX) - 0.5f;
    for (auto& v : hr) v = (rand() / (float)RAND_MAX) - 0.5f;
    for (auto& v : hw) v = ((rand() / (float)RAND_MAX) - 0.5f) * 0.05f;
    for (auto& v : hb) v = ((rand() / (float)RAND_MAX) - 0.5f) * 0.1f;
    for (int c = 0; c < Cout; ++c) { hn[c] = 0.1f + 0.01f * (c % 7); hn[Cout + c] = 1.5f; hn[2 * Cout + c] = -3.f; hn[3 * Cout + c] = 3.f; hs[c] = 0.2f; hs[Cout + c] = 0.9f; }
    CK(hipMemcpy(in, hin.data(), in_f * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(res, hr.data(), res_f * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(wraw, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(bias, hb.data(), Cout * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(n1, hn.data(), hn.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(n2, hn.data(), hn.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(sty, hs.data(), hs.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(pack_wino_k, dim3(4096), dim3(256), 0, 0, (const float*)wraw, w, Cout, Cin, 0, (const float*)nullptr);
    CK(hipMemset(out, 0, out_f * 4)); CK(hipMemset(out2, 0, out_f * 4));
    ConvP p{};
    p.in = in; p.Hi = H; p.Wi = W; p.Cin = Cin; p.out = out; p.H = H; p.W = W; p.Cout = Cout; p.B = B; p.in_bstride0 = 1;
    p.wpk = w; p.bias = bias; p.n1 = n1; p.n2 = n2; p.sty = sty; p.res = res; p.Hr = Hr; p.Wr = Wr;
    p.tiles_x = (W + 15) / 16;
    ConvP p2 = p; p2.out = out2;
    run<EPI, 0>(p, 1); run<EPI, 1>(p2, 1);
    std::vector<float> a(out_f), b(out_f);
    CK(hipMemcpy(a.data(), out, out_f * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), out2, out_f * 4, hipMemcpyDeviceToHost));
    size_t diff = 0, nz = 0;
    for (size_t i = 0; i < out_f; ++i) { diff += memcmp(&a[i], &b[i], 4) != 0; nz += a[i] != 0.f; }
    const double fl = 2.0 * B * H * W * (double)Cin * Cout * 4;      // executed (Winograd) FLOPs
    const int it = H * W * B > 500000 ? 10 : 40;
    run<EPI, 0>(p, 20);
    float t0 = 1e9f, t1 = 1e9f;
    for (int rep = 0; rep < 4; ++rep) { t0 = fminf(t0, run<EPI, 0>(p, it)); t1 = fminf(t1, run<EPI, 1>(p2, it)); }
    printf("%-30s EPI %3d: round-2 kernel %.4f ms = %6.1f TF (%.3f) | now %.4f ms = %6.1f TF (%.3f)  x%.3f | %s (%zu of %zu floats differ, %zu non-zero)\n",
           name, EPI, t0, fl / t0 / 1e9, fl / t0 / 1e9 / 157.3, t1, fl / t1 / 1e9, fl / t1 / 1e9 / 157.3, t0 / t1, diff ? "OUTPUT DIFFERS" : "bit-identical", diff, out_f, nz);
    for (float* q : {in, out, out2, w, wraw, bias, n1, n2, sty, res}) CK(hipFree(q));
}

int main() {
    constexpr int E54 = E_LRELU | E_NORM1 | E_RES_UPS | E_NORM2;
    layer<E54>("64->64 @200x136 B3 (partial)", 3, 200, 136, 64, 64);
    layer<E_RELU | E_POOL>("64->64 @200x136 B3 (partial)", 3, 200, 136, 64, 64);
    layer<E_RES | E_NORM2>("32->512 @72x40 B2 (partial)", 2, 72, 40, 32, 512);
    layer<E_RELU>("64->128 @100x36 B1 (partial)", 1, 100, 36, 64, 128);
    layer<E54>("64->64 @640^2 B8", 8, 640, 640, 64, 64);
    layer<E_RELU | E_POOL>("64->64 @640^2 B8", 8, 640, 640, 64, 64);
    layer<E_RELU>("64->128 @320^2 B8", 8, 320, 320, 64, 128);
    layer<E54>("128->128 @320^2 B8", 8, 320, 320, 128, 128);
    layer<E_RELU | E_POOL>("128->128 @320^2 B8", 8, 320, 320, 128, 128);
    layer<E_RELU>("128->256 @160^2 B8", 8, 160, 160, 128, 256);
    layer<E54>("256->256 @160^2 B8", 8, 160, 160, 256, 256);
    layer<E_RELU>("256->256 @160^2 B8", 8, 160, 160, 256, 256);
    layer<E_RELU | E_POOL>("256->256 @160^2 B8", 8, 160, 160, 256, 256);
    layer<E_RELU | E_NORM1>("256->512 @80^2 B8", 8, 80, 80, 256, 512);
    layer<E_RES>("32->512 @80^2 B8", 8, 80, 80, 32, 512);
    layer<E_RES | E_NORM2>("32->512 @80^2 B8", 8, 80, 80, 32, 512);
    layer<E_RES>("32->512 @144^2 B1", 1, 144, 144, 32, 512);
    layer<E54>("64->64 @640^2 B1", 1, 640, 640, 64, 64);
    layer<E54>("64->64 @1152^2 B1", 1, 1152, 1152, 64, 64);
    return 0;
}
