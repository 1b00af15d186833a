// tools/pkfma_bench.hip — issue cost of the packed-fp32 ops of conv_f43_k's input transform (one wave per SIMD, as there):
// clocks per v_pk_fma_f32 in a stream of six independent lines, with the multiplier (a) in an SGPR pair loaded by s_mov_b64
// every six ops (the library's form), (b) in an SGPR pair loaded once, (c) in a VGPR pair, (d) plain v_pk_fma without op_sel,
// (e) scalar v_fma_f32 pairs; each also interleaved behind an MFMA run of 24 (does the first op after an MFMA run pay extra?).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/pkfma_bench.hip -o tools/bin/pkfma_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

#define SIX(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5)

template <int MODE>
__global__ __launch_bounds__(256, 1) void k(float* out, long long* clk, int iters) {
    f32x2 x[6], c[6], r[6];
    for (int i = 0; i < 6; ++i) { x[i] = f32x2{1.0f + threadIdx.x * 1e-6f + i, 0.5f}; c[i] = f32x2{0.25f * i, 0.125f}; r[i] = c[i]; }
    f32x2 kv = {0.5625f, 0.5625f};
    asm volatile("" : "+v"(kv));
    unsigned long long ks = 0;
    if (MODE == 1) asm volatile("s_mov_b64 %0, 0x3f100000" : "=s"(ks));
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 24; ++rep) {       // 24 x 6 = 144 ops, as one input transform
            if (MODE == 0) {
                asm volatile("s_mov_b64 %6, 0x3f100000\n\t"
                             "v_pk_fma_f32 %0, %7, %6, %13 op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 %1, %8, %6, %14 op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 %2, %9, %6, %15 op_sel_hi:[1,0,1]\n\t"
                             "v_pk_fma_f32 %3, %10, %6, %16 op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 %4, %11, %6, %17 op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 %5, %12, %6, %18 op_sel_hi:[1,0,1]"
                             : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&s"(ks)
                             : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(c[0]), "v"(c[1]), "v"(c[2]), "v"(c[3]), "v"(c[4]), "v"(c[5]));
            } else if (MODE == 1) {
                asm volatile("v_pk_fma_f32 %0, %7, %6, %13 op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 %1, %8, %6, %14 op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 %2, %9, %6, %15 op_sel_hi:[1,0,1]\n\t"
                             "v_pk_fma_f32 %3, %10, %6, %16 op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 %4, %11, %6, %17 op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 %5, %12, %6, %18 op_sel_hi:[1,0,1]"
                             : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5])
                             : "s"(ks), "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(c[0]), "v"(c[1]), "v"(c[2]), "v"(c[3]), "v"(c[4]), "v"(c[5]));
            } else if (MODE == 2) {
                asm volatile("v_pk_fma_f32 %0, %7, %6, %13\n\tv_pk_fma_f32 %1, %8, %6, %14\n\tv_pk_fma_f32 %2, %9, %6, %15\n\t"
                             "v_pk_fma_f32 %3, %10, %6, %16\n\tv_pk_fma_f32 %4, %11, %6, %17\n\tv_pk_fma_f32 %5, %12, %6, %18"
                             : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5])
                             : "v"(kv), "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(c[0]), "v"(c[1]), "v"(c[2]), "v"(c[3]), "v"(c[4]), "v"(c[5]));
            } else if (MODE == 3) {      // scalar pairs: 12 v_fma_f32 with a literal-in-SGPR multiplier
                asm volatile("s_mov_b32 %6, 0x3f100000\n\t"
                             "v_fma_f32 %0, %7, %6, %13\n\tv_fma_f32 %1, %8, %6, %14\n\tv_fma_f32 %2, %9, %6, %15\n\t"
                             "v_fma_f32 %3, %10, %6, %16\n\tv_fma_f32 %4, %11, %6, %17\n\tv_fma_f32 %5, %12, %6, %18"
                             : "=&v"(r[0][0]), "=&v"(r[1][0]), "=&v"(r[2][0]), "=&v"(r[3][0]), "=&v"(r[4][0]), "=&v"(r[5][0]), "=&s"(*(unsigned*)&ks)
                             : "v"(x[0][0]), "v"(x[1][0]), "v"(x[2][0]), "v"(x[3][0]), "v"(x[4][0]), "v"(x[5][0]), "v"(c[0][0]), "v"(c[1][0]), "v"(c[2][0]), "v"(c[3][0]), "v"(c[4][0]), "v"(c[5][0]));
                asm volatile("v_fma_f32 %0, %7, %6, %13\n\tv_fma_f32 %1, %8, %6, %14\n\tv_fma_f32 %2, %9, %6, %15\n\t"
                             "v_fma_f32 %3, %10, %6, %16\n\tv_fma_f32 %4, %11, %6, %17\n\tv_fma_f32 %5, %12, %6, %18"
                             : "=&v"(r[0][1]), "=&v"(r[1][1]), "=&v"(r[2][1]), "=&v"(r[3][1]), "=&v"(r[4][1]), "=&v"(r[5][1])
                             : "s"(*(unsigned*)&ks), "v"(x[0][1]), "v"(x[1][1]), "v"(x[2][1]), "v"(x[3][1]), "v"(x[4][1]), "v"(x[5][1]), "v"(c[0][1]), "v"(c[1][1]), "v"(c[2][1]), "v"(c[3][1]), "v"(c[4][1]), "v"(c[5][1]));
            } else if (MODE == 4) {      // v_pk_mul / v_pk_add only (no third operand)
                asm volatile("v_pk_add_f32 %0, %6, %12\n\tv_pk_add_f32 %1, %7, %13\n\tv_pk_add_f32 %2, %8, %14\n\t"
                             "v_pk_add_f32 %3, %9, %15\n\tv_pk_add_f32 %4, %10, %16\n\tv_pk_add_f32 %5, %11, %17"
                             : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5])
                             : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(c[0]), "v"(c[1]), "v"(c[2]), "v"(c[3]), "v"(c[4]), "v"(c[5]));
            }
#pragma unroll
            for (int i = 0; i < 6; ++i) { f32x2 t = x[i]; x[i] = r[i]; c[i] = t; }      // renaming only: the next group reads this group's results (144 ops later at the earliest? no: 6 ops later)
        }
    }
    const long long t1 = clock64();
    float s = 0;
    for (int i = 0; i < 6; ++i) s += r[i][0] + r[i][1] + x[i][0] + c[i][1];
    if (s == 123.456f) out[threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) clk[MODE * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

int main() {
    float* out; long long* clk;
    CK(hipMalloc(&out, 4096)); CK(hipMalloc(&clk, 64 * 8)); CK(hipMemset(clk, 0, 64 * 8));
    const int iters = 200;
    hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, out, clk, iters);
    hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, out, clk, iters);
    hipLaunchKernelGGL(k<2>, dim3(256), dim3(256), 0, 0, out, clk, iters);
    hipLaunchKernelGGL(k<3>, dim3(256), dim3(256), 0, 0, out, clk, iters);
    hipLaunchKernelGGL(k<4>, dim3(256), dim3(256), 0, 0, out, clk, iters);
    CK(hipDeviceSynchronize());
    long long h[64];
    CK(hipMemcpy(h, clk, 64 * 8, hipMemcpyDeviceToHost));
    const char* names[5] = {"v_pk_fma_f32, SGPR-pair multiplier re-loaded by s_mov_b64 every six ops (conv_f43_k today)", "v_pk_fma_f32, SGPR-pair multiplier loaded once",
                            "v_pk_fma_f32, multiplier in a VGPR pair", "v_fma_f32 scalar pairs (two per packed op), SGPR multiplier", "v_pk_add_f32 (two VGPR-pair operands)"};
    for (int m = 0; m < 5; ++m) {
        const double ops = (double)iters * 24 * 6;
        printf("%-100s: %.2f clk per packed op (wave 0; waves 1-3: %.2f %.2f %.2f)\n", names[m], h[m * 4] / ops, h[m * 4 + 1] / ops, h[m * 4 + 2] / ops, h[m * 4 + 3] / ops);
    }
    return 0;
}
