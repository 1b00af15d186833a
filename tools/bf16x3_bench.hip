// tools/bf16x3_bench.hip — split-bf16 study, rate side (numerics side: tools/bf16x3_study.py).
// What would a three-piece bfloat16 product stage cost inside an F(4x4,3x3) kernel of conv_f43_k's shape (a wave owns 16
// tiles x 32 couts x 36 positions)?  Not a convolution: the INSTRUCTION STREAMS such a kernel needs per 32-channel step of a
// work item, timed with clock64 on resident data, one workgroup per CU as conv_f43_k runs:
//   mfma   432 v_mfma_f32_16x16x32_bf16 per wave = 36 positions x 2 cout blocks x 6 products (v1 u1, v1 u2, v2 u1, v1 u3, v2 u2,
//          v3 u1) over 32 channels — what the matrix pipe is busy with.  conv_f43_k spends 4 x 144 fp32 MFMAs of 32 clocks
//          = 18 432 clocks on the same 32 channels;
//   valu   the VALU work in front of them, per lane: the float32 input transform of 4 channel pairs (4 x 144 packed ops — unchanged
//          from conv_f43_k) + the split of the 288 transformed values into three bfloat16 pieces (convert, widen back, subtract,
//          twice) + their way into LDS in operand layout (a lane cannot hold 36 positions x 8 channels x 3 pieces = 432 operand
//          registers beside 288 accumulators: with K = 32 per MFMA the transformed input has to go through LDS);
//   both   one wave doing both (compiler-scheduled): does the bf16 matrix pipe hide the VALU stream, which the fp32 MFMA —
//          executed on the vector ALU itself — cannot?
//   pair   two waves per SIMD, one running `valu`, the other `mfma` (producer / consumer specialisation; the consumer keeps 144
//          accumulators — 16 couts — because two waves per SIMD have 256 registers each), free-running side by side (no hand-over
//          barriers: what the SIMD can co-issue, an upper bound for a real double-buffered pipeline); both waves' clocks are printed.
// Bytes per 32-channel step and CU that would have to arrive in LDS (conv_f43_k: 4 chunks x (37 KB halo + 36 KB U) = 292 KB):
//   halo 34 x 34 x 32 channels x 4 B = 148 KB (float32: the split comes after the transform) + U 36 x 32 couts x 32 channels x 3 pieces
//   x 2 B = 221 KB: 369 KB — at the 13-20 B/clock the L2 -> LDS path of a CU sustains (profiles/r05_f43_timeline.txt) that is
//   18 400 - 28 400 clocks, whatever the matrix pipe does.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/bf16x3_bench.hip -o tools/bin/bf16x3_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// three bfloat16 pieces of a packed pair, each as one dword {piece(x), piece(y)}
__device__ __forceinline__ void split3(const f32x2 v, unsigned& p1, unsigned& p2, unsigned& p3) {
    auto cvt = [](const f32x2 a) { unsigned r; asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a[0]), "v"(a[1])); return r; };
    auto widen = [](const unsigned p) { return f32x2{__builtin_bit_cast(float, p << 16), __builtin_bit_cast(float, p & 0xffff0000u)}; };
    p1 = cvt(v);
    const f32x2 r1 = v - widen(p1);
    p2 = cvt(r1);
    const f32x2 r2 = r1 - widen(p2);
    p3 = cvt(r2);
}

// the 1-D input transform of conv_f43.h on one line of six packed pairs (twelve packed ops)
__device__ __forceinline__ void line(f32x2& d0, f32x2& d1, f32x2& d2, f32x2& d3, f32x2& d4, f32x2& d5) {
    const float A = 0.75f, B = 1.5f, A2 = A * A, B2 = B * B, P = A2 * B2, S = A2 + B2;
    const f32x2 a = d4 - B2 * d2, b = d3 - B2 * d1, c = d4 - A2 * d2, f = d3 - A2 * d1;
    const f32x2 g0 = P * d0 + d4, g1 = P * d1 + d5;
    d0 = g0 - S * d2; d5 = g1 - S * d3;
    d1 = a + A * b; d2 = a - A * b; d3 = c + B * f; d4 = c - B * f;
}

// MODE 1 mfma, 2 valu, 3 both in one wave, 4 producer / consumer wave pairs (512 threads)
template <int MODE>
__global__ __launch_bounds__(MODE == 4 ? 512 : 256, 1) void step_k(const float* __restrict__ in, float* __restrict__ out, int steps, long long* clk) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool do_valu = MODE == 2 || MODE == 3 || (MODE == 4 && wave < 4);
    const bool do_mfma = MODE == 1 || MODE == 3 || (MODE == 4 && wave >= 4);
    constexpr int NB = MODE == 4 ? 1 : 2;      // cout blocks of 16 per consumer wave
    // resident data: a raw "halo" of packed pairs per lane and operand pieces, all in LDS
    f32x2* raw = (f32x2*)smem;                                    // [36][64 lanes]: a lane's 6 x 6 patch of one channel pair
    unsigned* pieces = (unsigned*)(smem + 36 * 64 * 8);      // [wave 4][36 pos][3 pieces][64 lanes] dwords
    for (int i = tid; i < 36 * 64; i += blockDim.x) raw[i] = f32x2{in[i & 4095], in[(i + 7) & 4095]};
    for (int i = tid; i < 4 * 36 * 3 * 64; i += blockDim.x) pieces[i] = 0x3f803f80u;
    __syncthreads();
    f32x4 acc[36][NB];
#pragma unroll
    for (int p = 0; p < 36; ++p)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[p][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int w4 = wave & 3;
    f32x2 keep = {0.f, 0.f};
    const long long t0 = clock64();
    for (int s = 0; s < steps; ++s) {
        asm volatile("" ::: "memory");      // the operands are re-read from LDS every step (they would be new data)
        if (do_valu) {
#pragma unroll 1
            for (int pair = 0; pair < 4; ++pair) {       // the lane's 8 channels of this 32-channel step, two at a time
                f32x2 d[36];
#pragma unroll
                for (int i = 0; i < 36; ++i) d[i] = raw[i * 64 + lane] + keep;
#pragma unroll
                for (int c = 0; c < 6; ++c) line(d[c * 6 + 0], d[c * 6 + 1], d[c * 6 + 2], d[c * 6 + 3], d[c * 6 + 4], d[c * 6 + 5]);
#pragma unroll
                for (int r = 0; r < 6; ++r) line(d[r], d[6 + r], d[12 + r], d[18 + r], d[24 + r], d[30 + r]);
#pragma unroll
                for (int i = 0; i < 36; ++i) {
                    unsigned p1, p2, p3;
                    split3(d[i], p1, p2, p3);
                    unsigned* dst = pieces + ((w4 * 36 + i) * 3) * 64 + lane;       // (a real kernel writes the operand layout: same count of LDS stores)
                    dst[0] = p1; dst[64] = p2; dst[128] = p3;
                }
                keep = d[7] * 1e-30f;
            }
        }
        if (do_mfma) {
#pragma unroll
            for (int p = 0; p < 36; p += 2) {
                // operands of positions p, p + 1 from LDS: three pieces of the transformed input (A) and of U for each cout block (B)
                bf16x8 a[2][3], b[2][NB][3];
#pragma unroll
                for (int pp = 0; pp < 2; ++pp) {
                    const uint4* src = (const uint4*)(pieces + (w4 * 36 + p + pp) * 3 * 64) + (lane & 15);
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        a[pp][k] = __builtin_bit_cast(bf16x8, src[k * 16]);
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb) b[pp][nb][k] = __builtin_bit_cast(bf16x8, src[(k + 1 + nb) % 3 * 16 + ((lane >> 4) & 1)]);
                    }
                }
                constexpr int REP = MODE == 4 ? 2 : 1;      // one cout block per consumer wave: the same 432 MFMAs per wave and step
                constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};      // v3 u1, v2 u2, v1 u3, v2 u1, v1 u2, v1 u1: small terms first
#pragma unroll
                for (int rep = 0; rep < REP; ++rep)
#pragma unroll
                    for (int k = 0; k < 6; ++k)      // product-major: dependent MFMAs on one accumulator are 2 x NB instructions apart
#pragma unroll
                        for (int pp = 0; pp < 2; ++pp)
#pragma unroll
                            for (int nb = 0; nb < NB; ++nb)
                                acc[p + pp][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[pp][PA[k]], b[pp][nb][PB[k]], acc[p + pp][nb], 0, 0, 0);
            }
        }
    }
    const long long t1 = clock64();
    f32x4 sum = {keep[0], keep[1], 0.f, 0.f};
#pragma unroll
    for (int p = 0; p < 36; ++p)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) sum += acc[p][nb];
    out[blockIdx.x * blockDim.x + tid] = sum[0] + sum[1] + sum[2] + sum[3];
    if (lane == 0) clk[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int MODE>
static double run(const char* what, const float* in, float* out, long long* clk, int steps) {
    const int threads = MODE == 4 ? 512 : 256, waves = threads / 64;
    const size_t smem = 36 * 64 * 8 + 4 * 36 * 3 * 64 * 4;      // 18 432 + 110 592 B
    CK(hipFuncSetAttribute((const void*)step_k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    CK(hipMemset(clk, 0, 256 * 8 * 8));
    hipLaunchKernelGGL(step_k<MODE>, dim3(256), dim3(threads), smem, 0, in, out, steps, clk);
    hipLaunchKernelGGL(step_k<MODE>, dim3(256), dim3(threads), smem, 0, in, out, steps, clk);
    CK(hipDeviceSynchronize());
    std::vector<long long> h(256 * 8);
    CK(hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost));
    double s = 0, s2 = 0; int n = 0, n2 = 0;
    for (int b = 0; b < 256; ++b) for (int w = 0; w < waves; ++w) { if (MODE == 4 && w >= 4) { s2 += (double)h[b * 8 + w]; ++n2; } else { s += (double)h[b * 8 + w]; ++n; } }
    double per = s / n / steps;
    if (MODE == 4) {
        const double per2 = s2 / n2 / steps;
        printf("%-74s %8.0f clocks per 32-channel step (producer waves), %8.0f (consumer waves)\n", what, per, per2);
        per = per > per2 ? per : per2;
    } else
        printf("%-74s %8.0f clocks per 32-channel step and wave\n", what, per);
    return per;
}

int main() {
    float *in, *out; long long* clk;
    CK(hipMalloc(&in, 4096 * 4)); CK(hipMalloc(&out, 256 * 512 * 4)); CK(hipMalloc(&clk, 256 * 8 * 8));
    std::vector<float> h(4096);
    for (auto& v : h) v = (rand() / (float)RAND_MAX) - 0.3f;
    CK(hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    const int steps = 64;
    printf("# tools/bf16x3_bench.hip: instruction streams of a three-piece bfloat16 F(4x4,3x3) product stage, per 32 input channels of a work item (16 tiles x 32 couts per wave, one wave per SIMD unless said)\n");
    printf("# conv_f43_k on the same 32 channels: 4 chunks x 144 fp32 MFMAs x 32 clocks = 18 432 MFMA clocks, ~29 000 clocks as measured (profiles/r05_f43_timeline.txt: 221 000 per 256-channel item)\n");
    const double m = run<1>("mfma: 432 v_mfma_f32_16x16x32_bf16, operands from LDS", in, out, clk, steps);
    const double v = run<2>("valu: float32 input transform of 8 channels + split into 3 bf16 pieces + LDS stores", in, out, clk, steps);
    const double b = run<3>("both in one wave (compiler-scheduled)", in, out, clk, steps);
    const double p = run<4>("pair: producer wave (valu) + consumer wave (mfma, 16 couts x 2 passes) per SIMD", in, out, clk, steps);
    printf("# matrix pipe alone: %.2fx the fp32 MFMA clocks of conv_f43_k (18 432); one wave doing both: %.2fx its measured 29 000; wave pairs: %.2fx\n", 18432.0 / m, 29000.0 / b, 29000.0 / p);
    printf("# operand delivery: 369 KB per step and CU into LDS (halo 148 KB float32 + U 221 KB in three bf16 pieces) against conv_f43_k's 292 KB: at 13-20 B/clock per CU = 18 400-28 400 clocks\n");
    (void)v;
    return 0;
}
