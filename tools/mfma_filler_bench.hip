// tools/mfma_filler_bench.hip — what one extra instruction costs beside v_mfma_f32_16x16x4_f32 with ONE wave per SIMD.
// The F(4x4,3x3) kernel (rerevst-code_amd/csrc/conv_f43.h) runs one 512-register wave per SIMD; its K loop interleaves, per MFMA, about one
// packed VALU op of the input transform, 0.75 ds_read_b64 and 0.13 LDS-DMA instructions.  This bench issues a stream of
// independent MFMAs (8 accumulators round-robin) with N fillers of one kind between consecutive MFMAs and reports the
// cycles per MFMA slot, i.e. how much of a filler hides in the 32-cycle shadow of the f32 MFMA.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_filler_bench.hip -o tools/bin/mfma_filler_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// KIND: 0 none, 1 independent v_pk_fma_f32, 2 dependent chain of v_pk_fma_f32, 3 independent v_fma_f32, 4 dependent v_fma_f32,
//       5 ds_read_b64 (counted wait far behind), 6 v_pk_add_f32 independent, 7 s_nop 0 (issue slot only), 8 v_mov_b32
template <int KIND, int N>
__global__ __launch_bounds__(256, 1) void k(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[4096];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
    float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
    f32x2 v[8], c = {0.5f, 0.25f};
    for (int i = 0; i < 8; ++i) v[i] = f32x2{a + i, b + i};
    float s[8];
    for (int i = 0; i < 8; ++i) s[i] = a * i;
    f32x2 ld[4] = {};
    const unsigned laddr = (unsigned)(size_t)(__attribute__((address_space(3))) float*)lds + (threadIdx.x & 63) * 8;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[m & 7]) : "v"(a), "v"(b));
#pragma unroll
            for (int f = 0; f < N; ++f) {
                const int j = (m * N + f) & 7;
                if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(v[j]) : "v"(c));
                if (KIND == 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(v[0]) : "v"(c));
                if (KIND == 3) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(s[j]) : "v"(a));
                if (KIND == 4) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(s[0]) : "v"(a));
                if (KIND == 5) asm volatile("ds_read_b64 %0, %1" : "=v"(ld[f & 3]) : "v"(laddr));
                if (KIND == 6) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(v[j]) : "v"(c));
                if (KIND == 7) asm volatile("s_nop 0");
                if (KIND == 8) asm volatile("v_mov_b32 %0, %1" : "=v"(s[j]) : "v"(a));
            }
            if (KIND == 5 && (m & 3) == 3) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    long long t1 = clock64();
    float r = 0;
    for (int i = 0; i < 8; ++i) r += acc[i][0] + acc[i][1] + v[i][0] + v[i][1] + s[i];
    r += ld[0][0] + ld[1][0] + ld[2][0] + ld[3][0];
    if (r == 123.456f) out[0] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) ((long long*)out)[1] = t1 - t0;
}

template <int KIND, int N>
void run(const char* name, float* out) {
    const int iters = 2000;
    hipLaunchKernelGGL((k<KIND, N>), dim3(256), dim3(256), 0, 0, out, 200);
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL((k<KIND, N>), dim3(256), dim3(256), 0, 0, out, iters);
    CK(hipDeviceSynchronize());
    long long clk;
    CK(hipMemcpy(&clk, (char*)out + 8, 8, hipMemcpyDeviceToHost));
    const double per = (double)clk / (iters * 16.0);
    printf("%-34s x%d per MFMA: %6.1f clk per MFMA slot  (+%5.1f over the bare stream, %4.1f per filler)\n", name, N, per, per - 32.0, N ? (per - 32.0) / N : 0.0);
}

// Two waves per SIMD (512 threads): every wave repeats [16 MFMAs + 16 N fillers], the fillers either one group of N behind
// each MFMA (interleaved) or all 16 N behind the 16 MFMAs (batched).  Reports SIMD cycles per MFMA (both waves' MFMAs
// share the pipe: 32 = the matrix pipe never waits).  PHASE: the odd waves start with the filler block (out of phase).
template <int KIND, int N, int BATCHED, int PHASE>
__global__ __launch_bounds__(512, 1) void k2(float* out, int iters) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
    float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
    f32x2 v[8], c = {0.5f, 0.25f};
    for (int i = 0; i < 8; ++i) v[i] = f32x2{a + i, b + i};
    const bool odd = (threadIdx.x >> 8) & 1;        // waves 4..7 = the second wave of each SIMD
    int sc[4] = {__builtin_amdgcn_readfirstlane((int)blockIdx.x + 3), 5, 7, 11}, sk = __builtin_amdgcn_readfirstlane(iters | 1);
    auto fill = [&](int j) {
        if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(v[j & 7]) : "v"(c));
        if (KIND == 3) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[j & 7][0]) : "v"(a));
        if (KIND == 9) asm volatile("s_mul_i32 %0, %0, %1" : "+s"(sc[j & 3]) : "s"(sk));                 // scalar ALU, four independent chains
        if (KIND == 11) asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(sc[j & 3]) : "v"(a));            // what a spilled SGPR costs
    };
    __syncthreads();
    long long t0 = clock64();
    if (PHASE && odd) {
#pragma unroll
        for (int j = 0; j < 8 * N; ++j) fill(j);
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[m & 7]) : "v"(a), "v"(b));
            if (!BATCHED) {
#pragma unroll
                for (int f = 0; f < N; ++f) fill(m * N + f);
            }
        }
        if (BATCHED) {
#pragma unroll
            for (int j = 0; j < 16 * N; ++j) fill(j);
        }
    }
    long long t1 = clock64();
    float r = 0;
    for (int i = 0; i < 8; ++i) r += acc[i][0] + acc[i][1] + v[i][0] + v[i][1];
    r += (float)(sc[0] ^ sc[1] ^ sc[2] ^ sc[3] ^ sk);
    if (r == 123.456f) out[0] = r;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) { ((long long*)out)[2 + 2 * (threadIdx.x >> 6)] = t0; ((long long*)out)[3 + 2 * (threadIdx.x >> 6)] = t1; }
}
template <int KIND, int N, int BATCHED, int PHASE>
void run2(const char* name, float* out) {
    const int iters = 2000;
    hipLaunchKernelGGL((k2<KIND, N, BATCHED, PHASE>), dim3(256), dim3(512), 0, 0, out, 200);
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL((k2<KIND, N, BATCHED, PHASE>), dim3(256), dim3(512), 0, 0, out, iters);
    CK(hipDeviceSynchronize());
    long long tt[16];
    CK(hipMemcpy(tt, (char*)out + 16, sizeof tt, hipMemcpyDeviceToHost));
    long long lo = tt[0], hi = tt[1];
    for (int w = 0; w < 8; ++w) { if (tt[2 * w] < lo) lo = tt[2 * w]; if (tt[2 * w + 1] > hi) hi = tt[2 * w + 1]; }
    const double per = (double)(hi - lo) / (iters * 32.0);      // 32 MFMAs per SIMD and iteration; first wave in to last wave out
    printf("2 waves/SIMD %-26s x%d per MFMA, %-11s%s: %6.1f clk per MFMA of the SIMD (+%5.1f; %4.1f per filler)\n", name, N,
           BATCHED ? "batched" : "interleaved", PHASE ? ", out of phase" : "", per, per - 32.0, N ? (per - 32.0) / N : 0.0);
}

int main() {
    float* out;
    CK(hipMalloc(&out, 4096));
    run2<0, 0, 0, 0>("bare", out);
    run2<1, 1, 0, 0>("v_pk_fma_f32", out); run2<1, 1, 1, 0>("v_pk_fma_f32", out); run2<1, 1, 1, 1>("v_pk_fma_f32", out);
    run2<1, 2, 0, 0>("v_pk_fma_f32", out); run2<1, 2, 1, 0>("v_pk_fma_f32", out); run2<1, 2, 1, 1>("v_pk_fma_f32", out);
    run2<3, 2, 0, 0>("v_fma_f32", out); run2<3, 2, 1, 0>("v_fma_f32", out); run2<3, 2, 1, 1>("v_fma_f32", out);
    // scalar work in the shadow of the MFMAs (item bookkeeping): how many SALU instructions per MFMA are free?
    run2<9, 1, 0, 0>("s_mul_i32", out); run2<9, 2, 0, 0>("s_mul_i32", out); run2<9, 4, 0, 0>("s_mul_i32", out); run2<9, 8, 0, 0>("s_mul_i32", out);
    run2<9, 12, 0, 0>("s_mul_i32", out); run2<9, 8, 1, 0>("s_mul_i32", out);
    run2<11, 1, 0, 0>("v_readlane_b32", out); run2<11, 2, 0, 0>("v_readlane_b32", out); run2<11, 4, 0, 0>("v_readlane_b32", out); run2<11, 4, 1, 0>("v_readlane_b32", out);
    run<0, 0>("bare MFMA stream", out);
    run<7, 1>("s_nop 0", out); run<7, 4>("s_nop 0", out);
    run<8, 1>("v_mov_b32", out); run<8, 2>("v_mov_b32", out); run<8, 4>("v_mov_b32", out);
    run<3, 1>("v_fma_f32 independent", out); run<3, 2>("v_fma_f32 independent", out); run<3, 4>("v_fma_f32 independent", out); run<3, 8>("v_fma_f32 independent", out);
    run<4, 1>("v_fma_f32 dependent chain", out); run<4, 2>("v_fma_f32 dependent chain", out); run<4, 4>("v_fma_f32 dependent chain", out);
    run<1, 1>("v_pk_fma_f32 independent", out); run<1, 2>("v_pk_fma_f32 independent", out); run<1, 4>("v_pk_fma_f32 independent", out);
    run<2, 1>("v_pk_fma_f32 dependent chain", out); run<2, 2>("v_pk_fma_f32 dependent chain", out); run<2, 4>("v_pk_fma_f32 dependent chain", out);
    run<6, 1>("v_pk_add_f32 independent", out); run<6, 2>("v_pk_add_f32 independent", out);
    run<5, 1>("ds_read_b64", out); run<5, 2>("ds_read_b64", out); run<5, 4>("ds_read_b64", out);
    return 0;
}
