// tools/f43_bench.hip — the F(4x4,3x3) prototype kernel (tools/conv_f43.h) against a scalar CPU convolution, and
// its rate next to the shipped F(2x2,3x3) row-split kernel on the layers it would replace, with ablations.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/f43_bench.hip -o tools/bin/f43_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <math.h>
#include "../rerevst-code_amd/csrc/conv_mfma.h"
#include "../rerevst-code_amd/csrc/conv_wino.h"
#include "conv_wino_split_ab.h"
#include "conv_f43.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

static ConvP geo43(ConvP p) {
    p.tiles_x = (p.W + 31) / 32; p.tiles_y = (p.H + 31) / 32;
    return p;
}
static dim3 grid43(ConvP& p) {
    const int slabs = p.Cout / 32;
    const int items = p.tiles_x * p.tiles_y * p.B * slabs;
    dim3 grid(items < 256 ? items : 256, 1);
    p.xcd_slabs = (grid.x % 8 == 0 && (grid.x / 8) % slabs == 0) ? 1 : 0;
    return grid;
}

template <int EPI, int ABL, int PK = 0>
float run43(ConvP p, int iters, long long* dbg = nullptr) {
    p = geo43(p);
    p.dbg = dbg;
    dim3 grid = grid43(p);
    CK(hipFuncSetAttribute((const void*)conv_f43_k<EPI, ABL, PK>, hipFuncAttributeMaxDynamicSharedMemorySize, F43Geo::SMEM));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((conv_f43_k<EPI, ABL, PK>), grid, dim3(256), F43Geo::SMEM, 0, p);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((conv_f43_k<EPI, ABL, PK>), grid, dim3(256), F43Geo::SMEM, 0, p);
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / iters;
}

template <int EPI, int ABL>
float run23(ConvP p, int iters) {      // the shipped row-split F(2x2,3x3) kernel on the same tensors
    p.tiles_x = (p.W + 15) / 16; p.tiles_y = (p.H + 15) / 16;
    const int slabs = p.Cout / 32;
    const int items = p.tiles_x * p.tiles_y * p.B * slabs;
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

struct Layer {
    int B, H, W, Cin, Cout;
    float *in, *out, *out23, *w43, *w23, *wraw, *bias, *n1;
    std::vector<float> hin, hw, hb;
};

static Layer make_layer(int B, int H, int W, int Cin, int Cout, float wscale) {
    Layer L{B, H, W, Cin, Cout};
    const size_t in_f = (size_t)B * (H + 2) * (W + 2) * Cin + (size_t)48 * (W + 50) * Cin;
    const size_t out_f = (size_t)B * (H + 2) * (W + 2) * Cout + (size_t)48 * (W + 50) * Cout;
    CK(hipMalloc(&L.in, in_f * 4)); CK(hipMalloc(&L.out, out_f * 4)); CK(hipMalloc(&L.out23, out_f * 4));
    CK(hipMalloc(&L.w43, (size_t)Cout * Cin * 36 * 4)); CK(hipMalloc(&L.w23, (size_t)Cout * Cin * 16 * 4)); CK(hipMalloc(&L.wraw, (size_t)Cout * Cin * 9 * 4));
    CK(hipMalloc(&L.bias, Cout * 4)); CK(hipMalloc(&L.n1, 4 * Cout * 4));
    L.hin.assign(in_f, 0.f); L.hw.resize((size_t)Cout * Cin * 9); L.hb.resize(Cout);
    for (int b = 0; b < B; ++b)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x)
                for (int c = 0; c < Cin; ++c) {
                    float v = (rand() / (float)RAND_MAX) * 2.f - 0.6f;       // post-activation like: mostly positive, O(1)
                    L.hin[(((size_t)b * (H + 2) + y + 1) * (W + 2) + x + 1) * Cin + c] = v > 0 ? v : 0.f;
                }
    for (auto& v : L.hw) v = ((rand() / (float)RAND_MAX) - 0.5f) * wscale;
    for (auto& v : L.hb) v = ((rand() / (float)RAND_MAX) - 0.5f) * 0.1f;
    std::vector<float> hn(4 * Cout);
    for (int c = 0; c < Cout; ++c) { hn[c] = 0.1f; hn[Cout + c] = 1.5f; hn[2 * Cout + c] = -3.f; hn[3 * Cout + c] = 3.f; }
    CK(hipMemcpy(L.in, L.hin.data(), in_f * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(L.wraw, L.hw.data(), L.hw.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(L.bias, L.hb.data(), Cout * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(L.n1, hn.data(), hn.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(L.out, 0, out_f * 4)); CK(hipMemset(L.out23, 0, out_f * 4));
    hipLaunchKernelGGL(pack_f43_k, dim3(4096), dim3(256), 0, 0, (const float*)L.wraw, L.w43, Cout, Cin);
    hipLaunchKernelGGL(pack_wino_k, dim3(4096), dim3(256), 0, 0, (const float*)L.wraw, L.w23, Cout, Cin, 0, (const float*)nullptr);
    CK(hipDeviceSynchronize());
    return L;
}
static void free_layer(Layer& L) { for (float* q : {L.in, L.out, L.out23, L.w43, L.w23, L.wraw, L.bias, L.n1}) CK(hipFree(q)); }

static ConvP params(const Layer& L, bool f43) {
    ConvP p{};
    p.in = L.in; p.Hi = L.H; p.Wi = L.W; p.Cin = L.Cin; p.out = f43 ? L.out : L.out23; p.H = L.H; p.W = L.W; p.Cout = L.Cout; p.B = L.B; p.in_bstride0 = 1;
    p.wpk = f43 ? L.w43 : L.w23; p.bias = L.bias; p.n1 = L.n1; p.n2 = L.n1; p.sty = L.n1;
    return p;
}

// correctness: F(4x4,3x3) and F(2x2,3x3) outputs against a double-precision direct convolution (+ bias, ReLU)
static void check(int B, int H, int W, int Cin, int Cout) {
    Layer L = make_layer(B, H, W, Cin, Cout, 0.08f);
    run43<E_RELU, 0>(params(L, true), 1);
    run23<E_RELU, 0>(params(L, false), 1);
    const size_t out_f = (size_t)B * (H + 2) * (W + 2) * Cout;
    std::vector<float> o43(out_f), o23(out_f);
    CK(hipMemcpy(o43.data(), L.out, out_f * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(o23.data(), L.out23, out_f * 4, hipMemcpyDeviceToHost));
    double e43 = 0, e23 = 0, ring = 0, rms = 0;
    size_t n = 0;
    for (int b = 0; b < B; ++b)
        for (int y = -1; y <= H; ++y)
            for (int x = -1; x <= W; ++x)
                for (int co = 0; co < Cout; ++co) {
                    const size_t oi = (((size_t)b * (H + 2) + y + 1) * (W + 2) + x + 1) * Cout + co;
                    if (y < 0 || y >= H || x < 0 || x >= W) { ring = fmax(ring, fabs((double)o43[oi])); continue; }
                    if ((x * 7 + y * 13 + co * 3) % 5) continue;      // a fifth of the outputs
                    double s = L.hb[co];
                    for (int ky = 0; ky < 3; ++ky)
                        for (int kx = 0; kx < 3; ++kx) {
                            const float* ip = &L.hin[(((size_t)b * (H + 2) + y + ky) * (W + 2) + x + kx) * Cin];
                            const float* wp = &L.hw[(size_t)co * Cin * 9 + ky * 3 + kx];
                            for (int c = 0; c < Cin; ++c) s += (double)ip[c] * wp[(size_t)c * 9];
                        }
                    if (s < 0) s = 0;
                    e43 = fmax(e43, fabs(s - o43[oi])); e23 = fmax(e23, fabs(s - o23[oi]));
                    rms += s * s; ++n;
                }
    printf("check B%d %dx%d %d->%d: max|err| F(4x4) %.3e, F(2x2) %.3e (output rms %.3f); ring max %.1e  %s\n", B, H, W, Cin, Cout, e43, e23,
           sqrt(rms / n), ring, (e43 < 2e-4 && ring == 0.0) ? "OK" : "MISMATCH");
    free_layer(L);
}

template <int EPI>
static void bench(const char* name, int B, int H, int W, int Cin, int Cout) {
    Layer L = make_layer(B, H, W, Cin, Cout, 0.05f);
    const ConvP p43 = params(L, true), p23 = params(L, false);
    const double fl = 2.0 * B * H * W * (double)Cin * Cout * 9;      // direct-form FLOPs
    const int it = 10;
    run23<EPI, 0>(p23, 30);      // clocks up
    float a0 = 1e9f, apk = 1e9f, a32 = 1e9f, a4 = 1e9f, e1 = 1e9f, e2 = 1e9f, e8 = 1e9f, e18 = 1e9f, a43 = 1e9f, b0 = 1e9f, b32 = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        a0 = fminf(a0, run43<EPI, 0>(p43, it)); apk = fminf(apk, run43<EPI, 0, 1>(p43, it)); a32 = fminf(a32, run43<EPI, 32>(p43, it)); a4 = fminf(a4, run43<EPI, 4>(p43, it));
        e1 = fminf(e1, run43<EPI, 32 | 1>(p43, it)); e2 = fminf(e2, run43<EPI, 32 | 2>(p43, it)); e8 = fminf(e8, run43<EPI, 32 | 8>(p43, it));
        e18 = fminf(e18, run43<EPI, 32 | 1 | 8>(p43, it)); a43 = fminf(a43, run43<EPI, 32 | 1 | 2 | 8>(p43, it));
        b0 = fminf(b0, run23<EPI, 0>(p23, it)); b32 = fminf(b32, run23<EPI, 32>(p23, it));
    }
    auto tf = [&](float ms) { return fl / ms / 1e9; };
    printf("%-20s EPI %2d direct-form TF: F(4x4,3x3) %.1f (%.4f ms; MFMA pipe %.3f of peak; packed-VALU transform %.1f) | no stores %.1f | no epilogue %.1f"
           " -> and no LDS-DMA %.1f | and no barriers %.1f | and no transform %.1f | and no DMA, no transform %.1f | loop only %.1f"
           "  ||  F(2x2,3x3) row split %.1f (%.4f ms) | no epilogue %.1f  ||  speed-up %.2fx\n",
           name, EPI, tf(a0), a0, tf(a0) / 4.0 / 157.3, tf(apk), tf(a4), tf(a32), tf(e1), tf(e2), tf(e8), tf(e18), tf(a43), tf(b0), b0, tf(b32), b0 / a0);
    // per-phase cycle timeline of the shipped form
    ConvP pg = geo43(p43);
    const int slabs = Cout / 32, items = pg.tiles_x * pg.tiles_y * B * slabs, g = items < 256 ? items : 256;
    long long* dbg; CK(hipMalloc(&dbg, (size_t)256 * 4 * 6 * 8));
    CK(hipMemset(dbg, 0, (size_t)256 * 4 * 6 * 8));
    run43<EPI, 16>(p43, 1, dbg);
    std::vector<long long> h((size_t)g * 4 * 6);
    CK(hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost));
    double s6[6] = {0};
    for (size_t i = 0; i < h.size(); ++i) s6[i % 6] += h[i];              // (every launch overwrites the counters: these are the last launch's)
    const double ipw = (double)items / g;
    const char* nm[6] = {"item setup", "MFMA runs + gaps", "-", "barriers", "output transform + stores", "input transforms (+ next item's patch)"};
    printf("     timeline (clk/item, %d chunks = %d MFMA clk):", Cin / 8, Cin / 8 * 4608);
    for (int k : {0, 1, 3, 5, 4}) printf(" %s %.0f |", nm[k], s6[k] / (g * 4) / ipw);
    printf(" %.1f items/WG\n", ipw);
    CK(hipFree(dbg));
    free_layer(L);
}

int main() {
    check(2, 40, 72, 64, 64);
    check(3, 32, 32, 32, 32);
    check(1, 100, 36, 128, 64);
    bench<E_RELU>("64->64 @640^2 B8", 8, 640, 640, 64, 64);
    bench<E_RELU | E_NORM1>("64->64 @640^2 B8", 8, 640, 640, 64, 64);
    bench<E_RELU>("128->128 @320^2 B8", 8, 320, 320, 128, 128);
    bench<E_RELU>("256->256 @160^2 B8", 8, 160, 160, 256, 256);
    bench<E_RELU>("64->64 @640^2 B1", 1, 640, 640, 64, 64);
    return 0;
}
