// tools/conv_microbench.hip — standalone timing of conv_mfma_k variants on one layer shape.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/conv_microbench.hip -o gpurun_out/conv_microbench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <math.h>
#include "../rerevst-code_amd/csrc/conv_mfma.h"
#include "conv_wino_ab.h"
#include "../rerevst-code_amd/csrc/conv_wino_split.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int BN, int TAPS, int EPI, int ABL, int MSUB = 0, int LD = 1>
float run(ConvP p, int iters) {
    if (MSUB) p.tiles_y = (p.H + MSUB * 4 - 1) / (MSUB * 4);
    dim3 grid(p.tiles_x * p.tiles_y * p.B, p.Cout / BN);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((conv_mfma_k<BN, TAPS, EPI, ABL, MSUB, LD>), grid, dim3(256), 0, 0, p);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((conv_mfma_k<BN, TAPS, EPI, ABL, MSUB, LD>), grid, dim3(256), 0, 0, p);
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / iters;
}

template <int ABL, int NW = 4>
float run_wino(ConvP p, int iters, int xcd = 0) {
    p.xcd_slabs = xcd;
    p.tiles_y = (p.H + 15) / 16;
    int items = p.tiles_x * p.tiles_y * p.B * (p.Cout / 32);
    dim3 grid(items < 256 ? items : 256, 1);
    CK(hipFuncSetAttribute((const void*)conv_wino_ab_k<E_RELU, ABL, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (WinoGeo<NW, 0>::SMEM)));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((conv_wino_ab_k<E_RELU, ABL, NW>), grid, dim3(NW * 64), (WinoGeo<NW, 0>::SMEM), 0, p);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((conv_wino_ab_k<E_RELU, ABL, NW>), grid, dim3(NW * 64), (WinoGeo<NW, 0>::SMEM), 0, p);
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / iters;
}

template <int ABL>
float run_split(ConvP p, int iters) {
    p.xcd_slabs = 1;
    p.tiles_y = (p.H + 15) / 16;
    int items = p.tiles_x * p.tiles_y * p.B * (p.Cout / 32);
    dim3 grid(items < 256 ? items : 256, 1);
    CK(hipFuncSetAttribute((const void*)conv_wino_split_k<E_RELU, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, WSPLIT_SMEM_BYTES));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((conv_wino_split_k<E_RELU, ABL>), grid, dim3(512), WSPLIT_SMEM_BYTES, 0, p);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((conv_wino_split_k<E_RELU, ABL>), grid, dim3(512), WSPLIT_SMEM_BYTES, 0, p);
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / iters;
}

template <int BN>
void layer(const char* name, int B, int H, int W, int Cin, int Cout) {
    const size_t in_f = (size_t)B * (H + 2) * (W + 2) * Cin + (size_t)40 * (W + 22) * Cin;
    const size_t out_f = (size_t)B * (H + 2) * (W + 2) * Cout + (size_t)40 * (W + 22) * Cout;
    float *in, *out, *w, *bias;
    CK(hipMalloc(&in, in_f * 4)); CK(hipMalloc(&out, out_f * 4));
    CK(hipMalloc(&w, (size_t)Cout * Cin * 16 * 4)); CK(hipMemset(w, 0, (size_t)Cout * Cin * 16 * 4)); CK(hipMalloc(&bias, Cout * 4));
    std::vector<float> hin(in_f), hw((size_t)Cout * Cin * 9);
    for (auto& v : hin) v = (rand() / (float)RAND_MAX) - 0.5f;
    for (auto& v : hw) v = ((rand() / (float)RAND_MAX) - 0.5f) * 0.05f;
    CK(hipMemcpy(in, hin.data(), in_f * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(bias, 0, Cout * 4)); CK(hipMemset(out, 0, out_f * 4));
    ConvP p{};
    p.in = in; p.Hi = H; p.Wi = W; p.Cin = Cin; p.out = out; p.H = H; p.W = W; p.Cout = Cout; p.B = B; p.in_bstride0 = 1;
    p.wpk = w; p.bias = bias; p.tiles_x = (W + 15) / 16; p.tiles_y = (H + 7) / 8;
    const double fl = 2.0 * B * H * W * (double)Cin * Cout * 9;
    const int it = 10;
    float t0 = run<BN, 9, E_RELU, 0>(p, it), t1 = run<BN, 9, E_RELU, 1>(p, it), t2 = run<BN, 9, E_RELU, 2>(p, it),
          t3 = run<BN, 9, E_RELU, 3>(p, it), t4 = run<BN, 9, E_RELU, 4>(p, it), t7 = run<BN, 9, E_RELU, 7>(p, it);
    float u1 = run<BN, 9, E_RELU, 0, 0, 1>(p, it), u2 = run<BN, 9, E_RELU, 0, 4, 0>(p, it), u3 = run<BN, 9, E_RELU, 0, 4, 1>(p, it),
          u4 = run<BN, 9, E_RELU, 1, 4, 0>(p, it), u5 = run<BN, 9, E_RELU, 4, 4, 0>(p, it);
    printf("%-28s bufferlds %.1f TF | BM256 %.1f | BM256+bufferlds %.1f | BM256 noload %.1f | BM256 nostore %.1f\n", name, fl / u1 / 1e9, fl / u2 / 1e9,
           fl / u3 / 1e9, fl / u4 / 1e9, fl / u5 / 1e9);
    float w0 = run_wino<0>(p, it), w1 = run_wino<1>(p, it), w2 = run_wino<2>(p, it), w4 = run_wino<4>(p, it), w7 = run_wino<7>(p, it), wx = run_wino<0>(p, it, 1), wm = run_wino<128 + 7>(p, it, 1);
    printf("%-28s WINOGRAD %.3f ms = %.1f TF-equivalent (direct FLOPs) | noload %.1f | nobarrier %.1f | nostore %.1f | none %.1f | XCD-SLABS %.1f | MFMA-only %.1f\n", name, w0,
           fl / w0 / 1e9, fl / w1 / 1e9, fl / w2 / 1e9, fl / w4 / 1e9, fl / w7 / 1e9, fl / wx / 1e9, fl / wm / 1e9);
    {
        float a0 = run_wino<0, 8>(p, it, 1), a1 = run_wino<1, 8>(p, it, 1), a2 = run_wino<2, 8>(p, it, 1), a4 = run_wino<4, 8>(p, it, 1), a7 = run_wino<7, 8>(p, it, 1);
        printf("%-28s WINO 8 waves %.3f ms = %.1f TF-eq | noload %.1f | nobarrier %.1f | nostore %.1f | none %.1f\n", name, a0, fl / a0 / 1e9, fl / a1 / 1e9,
               fl / a2 / 1e9, fl / a4 / 1e9, fl / a7 / 1e9);
    }
    printf("%-28s base %.3f ms %.1f TF | noload %.1f | nobarrier %.1f | noload+nobar %.1f | nostore %.1f | none %.1f TF  (WGs=%d)\n", name, t0,
           fl / t0 / 1e9, fl / t1 / 1e9, fl / t2 / 1e9, fl / t3 / 1e9, fl / t4 / 1e9, fl / t7 / 1e9,
           p.tiles_x * p.tiles_y * B * (Cout / BN));
    {
        float a0 = run_split<0>(p, it), a1 = run_split<1>(p, it), a2 = run_split<2>(p, it), a4 = run_split<4>(p, it), a7 = run_split<7>(p, it), a32 = run_split<32>(p, it);
        printf("%-28s WINO row-split %.3f ms = %.1f TF-eq | noload %.1f | nobarrier %.1f | nostore %.1f | none %.1f | NO EPILOGUE %.1f\n", name, a0, fl / a0 / 1e9, fl / a1 / 1e9,
               fl / a2 / 1e9, fl / a4 / 1e9, fl / a7 / 1e9, fl / a32 / 1e9);
    }
    {   // row-split kernel: per-phase cycles (wave averages)
        long long* dbg; CK(hipMalloc(&dbg, (size_t)256 * 8 * 6 * 8));
        ConvP q = p; q.dbg = dbg; q.xcd_slabs = 1; q.tiles_y = (q.H + 15) / 16;
        int items = q.tiles_x * q.tiles_y * q.B * (q.Cout / 32);
        dim3 grid(items < 256 ? items : 256, 1);
        CK(hipFuncSetAttribute((const void*)conv_wino_split_k<E_RELU, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, WSPLIT_SMEM_BYTES));
        CK(hipFuncSetAttribute((const void*)conv_wino_split_k<E_RELU, 20>, hipFuncAttributeMaxDynamicSharedMemorySize, WSPLIT_SMEM_BYTES));
        for (int variant = 0; variant < 2; ++variant) {
            for (int rep = 0; rep < 2; ++rep) {
                if (variant) hipLaunchKernelGGL((conv_wino_split_k<E_RELU, 20>), grid, dim3(512), WSPLIT_SMEM_BYTES, 0, q);
                else hipLaunchKernelGGL((conv_wino_split_k<E_RELU, 16>), grid, dim3(512), WSPLIT_SMEM_BYTES, 0, q);
            }
            CK(hipDeviceSynchronize());
            std::vector<long long> h((size_t)grid.x * 8 * 6);
            CK(hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost));
            double s6[6] = {0}, tot = 0;
            for (size_t i = 0; i < h.size(); ++i) { s6[i % 6] += h[i]; tot += h[i]; }
            const char* nm[6] = {"item setup", "chunks 1..n-1", "row sums+xch write", "reads+epilogue", "xch barrier", "chunk 0"};
            const double ipw = (double)items / grid.x;
            printf("     row-split timeline%s (clk/item):", variant ? " NO STORES" : "");
            for (int k : {0, 5, 1, 2, 4, 3}) printf(" %s %.0f |", nm[k], s6[k] / (grid.x * 8) / ipw);
            printf(" %d chunks/item, %.1f items/WG\n", q.Cin / 16, ipw);
        }
        CK(hipFree(dbg));
    }
    {   // 8-wave form against the 4-wave form (same transforms, same summation order per accumulator)
        ConvP q = p; q.xcd_slabs = 1; q.tiles_y = (q.H + 15) / 16;
        int items = q.tiles_x * q.tiles_y * q.B * (q.Cout / 32);
        dim3 grid(items < 256 ? items : 256, 1);
        std::vector<float> o4(out_f), o8(out_f);
        CK(hipMemset(out, 0, out_f * 4));
        hipLaunchKernelGGL((conv_wino_ab_k<E_RELU, 0, 4>), grid, dim3(256), (WinoGeo<4, 0>::SMEM), 0, q);
        CK(hipMemcpy(o4.data(), out, out_f * 4, hipMemcpyDeviceToHost));
        CK(hipMemset(out, 0, out_f * 4));
        hipLaunchKernelGGL((conv_wino_ab_k<E_RELU, 0, 8>), grid, dim3(512), (WinoGeo<8, 0>::SMEM), 0, q);
        CK(hipMemcpy(o8.data(), out, out_f * 4, hipMemcpyDeviceToHost));
        double md = 0; size_t bad = 0, first = 0;
        for (size_t i = 0; i < out_f; ++i) { double d = fabs((double)o4[i] - o8[i]); if (d > 1e-5) { if (!bad) first = i; ++bad; } if (d > md) md = d; }
        printf("     8-wave vs 4-wave: max |diff| %.3g, %zu mismatching of %zu", md, bad, out_f);
        if (bad) { size_t pxl = first / Cout; printf(" first at ch %zu, x %zu, y %zu (o4 %.5f o8 %.5f)", first % Cout, pxl % (W + 2), (pxl / (W + 2)) % (H + 2), o4[first], o8[first]); }
        printf("\n");
    }
    {   // Winograd per-phase cycles (wave averages)
        long long* dbg; CK(hipMalloc(&dbg, (size_t)256 * 8 * 6 * 8));
        ConvP q = p; q.dbg = dbg; q.xcd_slabs = 1; q.tiles_y = (q.H + 15) / 16;
        int items = q.tiles_x * q.tiles_y * q.B * (q.Cout / 32);
        dim3 grid(items < 256 ? items : 256, 1);
        CK(hipFuncSetAttribute((const void*)conv_wino_ab_k<E_RELU, 16, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (WinoGeo<4, 0>::SMEM)));
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((conv_wino_ab_k<E_RELU, 16, 4>), grid, dim3(256), (WinoGeo<4, 0>::SMEM), 0, q);
        CK(hipDeviceSynchronize());
        std::vector<long long> h((size_t)grid.x * 4 * 6);
        CK(hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost));
        double s[6] = {0}, tot = 0;
        for (size_t i = 0; i < h.size(); ++i) { s[i % 6] += h[i]; tot += h[i]; }
        const char* nm[6] = {"zero-acc", "item barrier", "V(0)", "K loop", "setup+prefetch", "epilogue"};
        printf("     wino timeline (%% of wave time):");
        for (int k = 0; k < 6; ++k) printf(" %s %.1f%% |", nm[k], 100.0 * s[k] / tot);
        printf(" total %.0f clk/wave, %.1f items/WG\n", tot / (grid.x * 4), (double)items / grid.x);
        CK(hipFree(dbg));
    }
    {   // per-wave timeline (s_memtime): loop time vs epilogue time, and the spread of WG start times
        const int nw = p.tiles_x * p.tiles_y * B * (Cout / BN) * 4;
        long long* dbg; CK(hipMalloc(&dbg, (size_t)nw * 4 * 8));
        ConvP q = p; q.dbg = dbg;
        dim3 grid(p.tiles_x * p.tiles_y * p.B, p.Cout / BN);
        hipLaunchKernelGGL((conv_mfma_k<BN, 9, E_RELU, 16>), grid, dim3(256), 0, 0, q);
        CK(hipDeviceSynchronize());
        std::vector<long long> h((size_t)nw * 4);
        CK(hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost));
        double loop = 0, epi = 0, drain = 0;
        for (int i = 0; i < nw; ++i) { loop += h[i*4+1]-h[i*4]; epi += h[i*4+2]-h[i*4+1]; drain += h[i*4+3]-h[i*4+2]; }
        printf("     timeline: avg loop %.0f clk, epilogue issue %.0f clk, store drain %.0f clk\n", loop/nw, epi/nw, drain/nw);
        CK(hipFree(dbg));
    }
    CK(hipFree(in)); CK(hipFree(out)); CK(hipFree(w)); CK(hipFree(bias));
}

int main() {
    layer<128>("256->256 @160^2 B8", 8, 160, 160, 256, 256);
    layer<128>("128->128 @320^2 B8", 8, 320, 320, 128, 128);
    layer<64>("64->64 @640^2 B8", 8, 640, 640, 64, 64);
    layer<128>("256->256 @160^2 B1", 1, 160, 160, 256, 256);
    layer<128>("256->512 @80^2 B8", 8, 80, 80, 256, 512);
    return 0;
}
