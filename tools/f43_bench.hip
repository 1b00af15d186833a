// tools/f43_bench.hip — rate of the LIBRARY kernels conv_f43_k (F(4x4,3x3), rerevst-code_amd/csrc/conv_f43.h) and
// conv_wino_split_k (F(2x2,3x3)) on the layers / epilogues of the per-frame path where the library chooses between them.
// No kernel lives here: ablations are the library header compiled with -DF43_ABL=n (one binary per value, see
// tools/f43_ablations.sh); F43_ABL & 16 prints the per-phase clock64 timeline instead of rates.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DF43_ABL=n] [-DF43_LAY=l] [-DF43_SPAN=s] tools/f43_bench.hip -o tools/bin/f43_bench[_n]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <math.h>
#include "../rerevst-code_amd/csrc/conv_mfma.h"
#include "../rerevst-code_amd/csrc/conv_wino.h"
#include "../rerevst-code_amd/csrc/conv_wino_split.h"
#include "../rerevst-code_amd/csrc/conv_f43.h"
#include "../rerevst-code_amd/csrc/prep_kernels.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

static dim3 grid_for(ConvP& p, int tile) {
    p.tiles_x = (p.W + tile - 1) / tile; p.tiles_y = (p.H + tile - 1) / tile;
    const int slabs = p.Cout / 32;
    const int items = p.tiles_x * p.tiles_y * p.B * slabs;
    dim3 grid(items < 256 ? items : 256, 1);
    p.xcd_slabs = (grid.x % 8 == 0 && (grid.x / 8) % slabs == 0) ? 1 : 0;
    return grid;
}

#ifndef F43_LAY
#define F43_LAY 0      // tensor layouts of the conv_f43_k under test (conv_f43.h LAY: 1 channel-chunk-major input, 2 output; the buffers are the same, timing only)
#endif
template <int EPI>
float run43(ConvP p, int iters, long long* dbg = nullptr) {
    p.dbg = dbg;
    dim3 grid = grid_for(p, 32);
    CK(hipFuncSetAttribute((const void*)conv_f43_k<EPI, F43_LAY>, hipFuncAttributeMaxDynamicSharedMemorySize, F43Geo::SMEM));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((conv_f43_k<EPI, F43_LAY>), grid, dim3(256), F43Geo::SMEM, 0, p);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((conv_f43_k<EPI, F43_LAY>), grid, dim3(256), F43Geo::SMEM, 0, p);
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / iters;
}

template <int EPI>
float run23(ConvP p, int iters) {
    dim3 grid = grid_for(p, 16);
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
static void bench(const char* name, int B, int H, int W, int Cin, int Cout) {
    const bool pool = EPI & E_POOL;
    const int Ho = pool ? H / 2 : H, Wo = pool ? W / 2 : W;
    const size_t in_f = (size_t)B * (H + 2) * (W + 2 + P8_PAD) * Cin + (size_t)48 * (W + 50) * Cin;      // (room for the wider rows of a channel-chunk-major interpretation, F43_LAY)
    const size_t out_f = (size_t)B * (Ho + 2) * (Wo + 2 + P8_PAD) * Cout + (size_t)48 * (W + 50) * Cout;
    const size_t res_f = (size_t)B * (H / 2 + 2) * (W / 2 + 2) * Cout + 4096;
    float *in, *out, *w43, *w23, *wraw, *bias, *n1, *sty, *res;
    CK(hipMalloc(&in, in_f * 4)); CK(hipMalloc(&out, out_f * 4)); CK(hipMalloc(&res, res_f * 4));
    CK(hipMalloc(&w43, (size_t)Cout * Cin * 36 * 4)); CK(hipMalloc(&w23, (size_t)Cout * Cin * 16 * 4)); CK(hipMalloc(&wraw, (size_t)Cout * Cin * 9 * 4));
    CK(hipMalloc(&bias, Cout * 4)); CK(hipMalloc(&n1, 4 * Cout * 4)); CK(hipMalloc(&sty, 2 * Cout * 4));
    std::vector<float> hin(in_f, 0.f), hw((size_t)Cout * Cin * 9), hn(4 * Cout), hs(2 * Cout), hr(res_f);
    for (int b = 0; b < B; ++b)
        for (int y = 0; y < H; ++y)
            for (size_t i = 0; i < (size_t)W * Cin; ++i) {
                const float v = (rand() / (float)RAND_MAX) * 2.f - 0.6f;       // post-activation like: mostly positive, O(1)
                hin[(((size_t)b * (H + 2) + y + 1) * (W + 2) + 1) * Cin + i] = v > 0 ? v : 0.f;
            }
    for (auto& v : hr) v = (rand() / (float)RAND_MAX) - 0.5f;
    for (auto& v : hw) v = ((rand() / (float)RAND_MAX) - 0.5f) * 0.05f;
    for (int c = 0; c < Cout; ++c) { hn[c] = 0.1f; hn[Cout + c] = 1.5f; hn[2 * Cout + c] = -3.f; hn[3 * Cout + c] = 3.f; hs[c] = 0.2f; hs[Cout + c] = 0.9f; }
    CK(hipMemcpy(in, hin.data(), in_f * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(res, hr.data(), res_f * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(wraw, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(n1, hn.data(), hn.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(sty, hs.data(), hs.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(bias, 0, Cout * 4)); CK(hipMemset(out, 0, out_f * 4));
    hipLaunchKernelGGL(pack_f43_k, dim3(4096), dim3(256), 0, 0, (const float*)wraw, w43, Cout, Cin);
    hipLaunchKernelGGL(pack_wino_k, dim3(4096), dim3(256), 0, 0, (const float*)wraw, w23, Cout, Cin, 0, (const float*)nullptr);
    CK(hipDeviceSynchronize());
    ConvP p{};
    p.in = in; p.Hi = H; p.Wi = W; p.Cin = Cin; p.out = out; p.H = H; p.W = W; p.Cout = Cout; p.B = B; p.in_bstride0 = 1;
    p.bias = bias; p.n1 = n1; p.n2 = n1; p.sty = sty; p.res = res; p.Hr = H / 2; p.Wr = W / 2;
    ConvP p43 = p, p23 = p;
    p43.wpk = w43; p23.wpk = w23;
    const double fl = 2.0 * B * H * W * (double)Cin * Cout * 9;      // direct-form FLOPs
    if (F43_ABL & 16) {
        long long* dbg; CK(hipMalloc(&dbg, (size_t)256 * 4 * 6 * 8));
        CK(hipMemset(dbg, 0, (size_t)256 * 4 * 6 * 8));
        run43<EPI>(p43, 1, dbg);
        ConvP pg = p43; dim3 gd = grid_for(pg, 32);
        const int items = pg.tiles_x * pg.tiles_y * B * (Cout / 32), g = gd.x;
        std::vector<long long> h((size_t)g * 4 * 6);
        CK(hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost));
        double s6[6] = {0};
        for (size_t i = 0; i < h.size(); ++i) s6[i % 6] += h[i];
        const double ipw = (double)items / g;
        const char* nm[6] = {"item setup", "MFMA runs + gaps", "-", "barriers", "output transform + stores", "input transforms (+ next item's patch)"};
        printf("%-22s EPI %2d timeline (clk/item, %d chunks = %d MFMA clk):", name, EPI, Cin / 8, Cin / 8 * 4608);
        for (int k : {0, 1, 3, 5, 4}) printf(" %s %.0f |", nm[k], s6[k] / (g * 4) / ipw);
        printf(" %.1f items/WG\n", ipw);
        CK(hipFree(dbg));
    } else {
        const int it = 10;
        run23<EPI>(p23, 30);      // clocks up
        float a = 1e9f, b = 1e9f;
        for (int rep = 0; rep < 3; ++rep) { a = fminf(a, run43<EPI>(p43, it)); b = fminf(b, run23<EPI>(p23, it)); }
        printf("%-22s EPI %2d F43_ABL %2d: F(4x4,3x3) %.4f ms = %.1f TF direct-form (MFMA pipe %.3f of peak) | F(2x2,3x3) row split %.4f ms = %.1f TF (%.3f) | speed-up %.3fx\n",
               name, EPI, F43_ABL, a, fl / a / 1e9, fl / a / 1e9 / 4.0 / 157.3, b, fl / b / 1e9, fl / b / 1e9 / 2.25 / 157.3, b / a);
    }
    for (float* q : {in, out, w43, w23, wraw, bias, n1, sty, res}) CK(hipFree(q));
}

int main() {
    constexpr int E54 = E_LRELU | E_NORM1 | E_RES_UPS | E_NORM2;
    bench<E_RELU | E_POOL>("64->64 @640^2 B8", 8, 640, 640, 64, 64);        // conv1_2
    bench<E_RELU>("64->128 @320^2 B8", 8, 320, 320, 64, 128);               // conv2_1
    bench<E_RELU | E_POOL>("128->128 @320^2 B8", 8, 320, 320, 128, 128);    // conv2_2
    bench<E_RELU>("128->256 @160^2 B8", 8, 160, 160, 128, 256);             // conv3_1
    bench<E_RELU>("256->256 @160^2 B8", 8, 160, 160, 256, 256);             // conv3_2, conv3_3
    bench<E_RELU | E_POOL>("256->256 @160^2 B8", 8, 160, 160, 256, 256);    // conv3_4
    bench<E54>("256->256 @160^2 B8", 8, 160, 160, 256, 256);                // slice4.conv2
    bench<E54>("128->128 @320^2 B8", 8, 320, 320, 128, 128);                // slice3.conv2
    bench<E54>("64->64 @640^2 B8", 8, 640, 640, 64, 64);                    // slice2.conv2
    bench<E_RELU>("256->256 @160^2 B4", 4, 160, 160, 256, 256);
    bench<E54>("64->64 @640^2 B4", 4, 640, 640, 64, 64);
    bench<E_RELU>("256->256 @96^2 B22", 22, 96, 96, 256, 256);              // 256x256 frames (384 padded), 22 per sub-batch
    bench<E54>("64->64 @384^2 B22", 22, 384, 384, 64, 64);
    bench<E_RELU>("256->256 @288^2 B2", 2, 288, 288, 256, 256);             // 1024x1024 frames (1152 padded), 2 per sub-batch
    // the sub-batch of the host entries at 512x512 (16 frames per launch: nothing stays in the memory-side cache between
    // launches, as in the pipeline).  `base` of use_f43 (rerevst_hip.hip) = the F(2x2) / F(4x4) time ratio of these rows
    // with the partially filled last round taken out: x ceil(r43) / r43 for the 256-channel rows (12.5 rounds).
    bench<E_RELU | E_POOL>("64->64 @640^2 B16", 16, 640, 640, 64, 64);
    bench<E_RELU>("64->128 @320^2 B16", 16, 320, 320, 64, 128);
    bench<E_RELU | E_POOL>("128->128 @320^2 B16", 16, 320, 320, 128, 128);
    bench<E_RELU>("128->256 @160^2 B16", 16, 160, 160, 128, 256);
    bench<E_RELU>("256->256 @160^2 B16", 16, 160, 160, 256, 256);
    bench<E_RELU | E_POOL>("256->256 @160^2 B16", 16, 160, 160, 256, 256);
    bench<E54>("256->256 @160^2 B16", 16, 160, 160, 256, 256);
    bench<E54>("128->128 @320^2 B16", 16, 320, 320, 128, 128);
    bench<E54>("64->64 @640^2 B16", 16, 640, 640, 64, 64);
    return 0;
}
