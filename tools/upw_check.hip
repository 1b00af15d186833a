// tools/upw_check.hip — conv_wino_ab_k<.., UPS = 1> (nearest-x2 upsample + 3x3 conv) against a scalar CPU loop.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/upw_check.hip -o tools/bin/upw_check
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
#include "../rerevst-code_amd/csrc/conv_mfma.h"
#include "conv_wino_ab.h"
#include "../rerevst-code_amd/csrc/conv_wino_split.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int NW, int UPS = 1>
int check(int B, int Hl, int Wl, int Cin, int Cout, int tap = -1) {
    const int H = (UPS ? 2 : 1) * Hl, W = (UPS ? 2 : 1) * Wl;
    const size_t in_f = (size_t)B * (Hl + 2) * (Wl + 2) * Cin + (size_t)40 * (Wl + 22) * Cin;
    const size_t out_f = (size_t)B * (H + 2) * (W + 2) * Cout;
    std::vector<float> hin(in_f, 0.f), hw((size_t)Cout * Cin * 9), hb(Cout), ref(out_f, 0.f), got(out_f);
    for (int b = 0; b < B; ++b) for (int y = 0; y < Hl; ++y) for (int x = 0; x < Wl; ++x) for (int c = 0; c < Cin; ++c)
        hin[(((size_t)b * (Hl + 2) + y + 1) * (Wl + 2) + x + 1) * Cin + c] = (rand() / (float)RAND_MAX) - 0.5f;
    for (auto& v : hw) v = ((rand() / (float)RAND_MAX) - 0.5f) * 0.1f;
    for (auto& v : hb) v = (rand() / (float)RAND_MAX) - 0.5f;
    if (tap >= 0) { for (auto& v : hw) v = 0.f; if (tap < 9) for (auto& v : hb) v = 0.f; for (int c = 0; c < Cout && c < Cin; ++c) hw[((size_t)c * Cin + c) * 9 + tap % 9] = 1.f; }
    if (tap == -2) for (auto& v : hb) v = 0.f;
    for (int b = 0; b < B; ++b) for (int y = 0; y < H; ++y) for (int x = 0; x < W; ++x) for (int co = 0; co < Cout; ++co) {
        double a = hb[co];
        for (int ky = 0; ky < 3; ++ky) for (int kx = 0; kx < 3; ++kx) {
            const int uy = y + ky - 1, ux = x + kx - 1;
            if (uy < 0 || uy >= H || ux < 0 || ux >= W) continue;
            const float* px = &hin[(((size_t)b * (Hl + 2) + (UPS ? uy / 2 : uy) + 1) * (Wl + 2) + (UPS ? ux / 2 : ux) + 1) * Cin];
            for (int ci = 0; ci < Cin; ++ci) a += (double)px[ci] * hw[((size_t)co * Cin + ci) * 9 + ky * 3 + kx];
        }
        ref[(((size_t)b * (H + 2) + y + 1) * (W + 2) + x + 1) * Cout + co] = (float)(a >= 0 ? a : 0.2 * a);
    }
    float *in, *out, *w, *wp, *bias;
    CK(hipMalloc(&in, in_f * 4)); CK(hipMalloc(&out, out_f * 4 + 65536)); CK(hipMalloc(&w, hw.size() * 4)); CK(hipMalloc(&wp, (size_t)Cout * Cin * 16 * 4)); CK(hipMalloc(&bias, Cout * 4));
    CK(hipMemcpy(in, hin.data(), in_f * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(bias, hb.data(), Cout * 4, hipMemcpyHostToDevice)); CK(hipMemset(out, 0, out_f * 4));
    hipLaunchKernelGGL(pack_wino_k, dim3(1024), dim3(256), 0, 0, (const float*)w, wp, Cout, Cin, UPS);
    ConvP p{};
    p.in = in; p.Hi = Hl; p.Wi = Wl; p.Cin = Cin; p.out = out; p.H = H; p.W = W; p.Cout = Cout; p.B = B; p.in_bstride0 = 1;
    p.wpk = wp; p.bias = bias; p.tiles_x = (W + 15) / 16; p.tiles_y = (H + 15) / 16;
    using Geo = WinoGeo<NW == 9 ? 8 : NW, UPS>;
    const int items = p.tiles_x * p.tiles_y * B * (Cout / 32);
    const int resident = 256 * Geo::OCC;
    dim3 grid(items < resident ? items : resident, 1);
    p.xcd_slabs = (grid.x % 8 == 0 && (grid.x / 8) % (Cout / 32) == 0) ? 1 : 0;
    if (NW == 9) {      // the row-split 8-wave kernel
        CK(hipFuncSetAttribute((const void*)conv_wino_split_k<E_LRELU>, hipFuncAttributeMaxDynamicSharedMemorySize, WSPLIT_SMEM_BYTES));
        hipLaunchKernelGGL((conv_wino_split_k<E_LRELU>), grid, dim3(512), WSPLIT_SMEM_BYTES, 0, p);
    } else {
        CK(hipFuncSetAttribute((const void*)conv_wino_ab_k<E_LRELU, 0, NW == 9 ? 8 : NW, UPS>, hipFuncAttributeMaxDynamicSharedMemorySize, Geo::SMEM));
        hipLaunchKernelGGL((conv_wino_ab_k<E_LRELU, 0, NW == 9 ? 8 : NW, UPS>), grid, dim3(NW * 64), Geo::SMEM, 0, p);
    }
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(got.data(), out, out_f * 4, hipMemcpyDeviceToHost));
    double md = 0; size_t bad = 0, first = 0;
    for (size_t i = 0; i < out_f; ++i) { const double d = fabs((double)got[i] - ref[i]); if (d > 1e-4) { if (!bad) first = i; ++bad; } if (d > md) md = d; }
    printf("UPS=%d NW=%d B=%d %dx%d -> %dx%d  %d->%d: max |diff| %.3g, %zu bad of %zu", UPS, NW, B, Hl, Wl, H, W, Cin, Cout, md, bad, out_f);
    if (bad) { const size_t px = first / Cout; printf("  first: ch %zu x %zu y %zu got %.5f ref %.5f", first % Cout, px % (W + 2), (px / (W + 2)) % (H + 2), got[first], ref[first]); }
    printf("\n");
    if (tap >= 0 && bad) {
        for (int y = 0; y < 4; ++y) { for (int x = 0; x < 6; ++x) { size_t i = (((size_t)0 * (H + 2) + y + 1) * (W + 2) + x + 1) * Cout + 5; printf("  %7.3f/%7.3f", got[i], ref[i]); } printf("\n"); }
    }
    CK(hipFree(in)); CK(hipFree(out)); CK(hipFree(w)); CK(hipFree(wp)); CK(hipFree(bias));
    return bad != 0;
}

int main() {
    int f = 0;
    f |= check<4, 0>(1, 16, 16, 64, 64);
    f |= check<8, 0>(2, 24, 40, 64, 64);
    f |= check<9, 0>(1, 16, 16, 64, 64);
    f |= check<9, 0>(2, 24, 40, 64, 64);
    f |= check<9, 0>(1, 37, 53, 128, 96);
    for (int tap = 0; tap < 9; ++tap) { printf("tap %d: ", tap); f |= check<4>(1, 8, 8, 64, 64, tap); }
    printf("delta tap 4 + random bias: "); f |= check<4>(1, 8, 8, 64, 64, 13);
    printf("random weights, zero bias: "); f |= check<4>(1, 8, 8, 64, 64, -2);
    f |= check<4>(1, 8, 8, 64, 64);
    f |= check<4>(2, 24, 40, 64, 64);
    f |= check<4>(1, 13, 21, 128, 64);
    f |= check<8>(2, 24, 40, 64, 64);
    return f;
}
