// tools/store_align_bench.hip — why is conv_first_k slower into a channel-chunk-major tensor (0.0375 ms per 640 x 640 frame) than into
// NHWC (0.0234)?  The store pattern alone, no arithmetic: 256-thread workgroups, one per 16 x 16 pixel tile of B frames, every thread
// sixteen 16-byte non-temporal stores, in four variants of where they go:
//   nhwc      [B][H+2][W+2][64]: a wave instruction = 4 pixel rows x 256 contiguous bytes (whole 128-byte lines)
//   p8        [B][8][H+2][W+2][8]: a wave instruction = 2 planes x 512 contiguous bytes starting at stored column 16 tx + 1 (32 bytes past a line start)
//   p8-align  the same with the plane's rows pitched W + 8 pixels and the image at stored column 4: every 512-byte run starts on a line
//   p8-64x4   the aligned planes, tiles of 64 x 4 pixels: a wave instruction = 1 KB of one plane row
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/store_align_bench.hip -o tools/bin/store_align_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void store_k(float* out, int H, int W, int B) {
    const int tid = threadIdx.x;
    int bx = blockIdx.x;
    const int TW = MODE == 3 ? 64 : 16, TH = MODE == 3 ? 4 : 16;
    const int tiles_x = W / TW, tiles_y = H / TH;
    const int tx = bx % tiles_x; bx /= tiles_x;
    const int ty = bx % tiles_y;
    const int b = bx / tiles_y;
    const int x0 = tx * TW, y0 = ty * TH;
    const f32x4 v = {1.f + tid, 2.f, 3.f, 4.f};
    if (MODE == 0) {
        const int q = tid & 15, prow = tid >> 4;
        float* orow = out + (size_t)b * (size_t)(H + 2) * (W + 2) * 64 + ((size_t)(y0 + prow + 1) * (W + 2) + x0 + 1) * 64 + q * 4;
#pragma unroll
        for (int pc = 0; pc < 16; ++pc) __builtin_nontemporal_store(v, (f32x4*)&orow[pc * 64]);
    } else if (MODE == 1 || MODE == 2) {
        const int pitch = MODE == 2 ? W + 8 : W + 2, col0 = MODE == 2 ? 4 : 1;
        const int half = tid & 1, pc = (tid >> 1) & 15, k8 = tid >> 5;
        float* plane = out + ((size_t)b * 8 + k8) * (size_t)(H + 2) * pitch * 8 + ((size_t)(y0 + 1) * pitch + x0 + pc + col0) * 8 + half * 4;
#pragma unroll
        for (int prow = 0; prow < 16; ++prow) __builtin_nontemporal_store(v, (f32x4*)&plane[(size_t)prow * pitch * 8]);
    } else {
        const int pitch = W + 8;
        const int half = tid & 1, pc = (tid >> 1) & 31, k2 = tid >> 6;      // a wave = 32 pixels x 32 bytes of one plane row; four waves = chunks k2, k2 + 4
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int k8 = k2 + 4 * (it & 1), xh = (it >> 1) & 1, row = it >> 2;
            float* dst = out + ((size_t)b * 8 + k8) * (size_t)(H + 2) * pitch * 8 + ((size_t)(y0 + row + 1) * pitch + x0 + 32 * xh + pc + 4) * 8 + half * 4;
            __builtin_nontemporal_store(v, (f32x4*)dst);
        }
    }
}

template <int MODE>
static void run(const char* what, float* out, int H, int W, int B) {
    const int tiles = MODE == 3 ? (W / 64) * (H / 4) : (W / 16) * (H / 16);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(store_k<MODE>, dim3(tiles * B), dim3(256), 0, 0, out, H, W, B);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    const int it = 20;
    for (int i = 0; i < it; ++i) hipLaunchKernelGGL(store_k<MODE>, dim3(tiles * B), dim3(256), 0, 0, out, H, W, B);
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= it;
    printf("%-10s %.4f ms per frame = %.2f TB/s written\n", what, ms / B, (double)B * H * W * 256 / ms / 1e9);
}

int main() {
    const int H = 640, W = 640, B = 16;
    float* out;
    CK(hipMalloc(&out, (size_t)B * (H + 2) * (W + 8) * 64 * 4 + (64 << 20)));
    printf("# tools/store_align_bench.hip: the stores of conv_first_k alone (sixteen 640 x 640 frames, 256 bytes per pixel), by destination layout\n");
    run<0>("nhwc", out, H, W, B);
    run<1>("p8", out, H, W, B);
    run<2>("p8-align", out, H, W, B);
    run<3>("p8-64x4", out, H, W, B);
    return 0;
}
