// tools/fetch_calib.hip — what rocprofv3's FETCH_SIZE counts for the access patterns of this library.
// MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reports HALF the bytes of a wide coalesced streaming read (128-byte
// requests tallied at 64 B) and says other widths are uncalibrated.  The transform-domain kernels stage their input by
// buffer_load_dwordx4 ... lds in 64-byte segments (four lanes = the 16 channels of one pixel's chunk) that are
// Cin * 4 bytes apart, so the factor must be measured for THAT pattern before FETCH_SIZE is turned into bytes.
// Every kernel below reads a known number of bytes exactly once from a buffer larger than the Infinity Cache:
//   stream      : float4 per lane, 1 KB contiguous per wave (the guide's case)
//   seg64_sNNN  : LDS-DMA, 16 B per lane, 64-byte segments NNN bytes apart (256 = 64-channel tensors, 512, 1024, 2048)
//   seg256      : LDS-DMA, 256-byte segments (conv_last_k's reads of a 64-channel pixel)
//   chunkmajor_C: conv_f43_k's pattern — a workgroup owns a block of 1024 pixels of a C-channel tensor and reads it CHUNK-MAJOR:
//                 for every 8-channel chunk, the 32-byte piece of each pixel (two lanes per pixel); every byte of the
//                 tensor is read exactly once, each 128-byte line is touched by four consecutive chunk passes
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/fetch_calib.hip -o tools/bin/fetch_calib
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out -o p -- tools/bin/fetch_calib ; python tools/fetch_calib_summary.py <db>
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include "../rerevst-code_amd/csrc/conv_mfma.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void stream_k(const float4* __restrict__ in, float* out, size_t n4) {
    float4 s = {0, 0, 0, 0};
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = in[i];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    if (s.x + s.y + s.z + s.w == 123.456f) out[0] = s.x;
}

// every wave-instruction fetches 64 lanes x 16 B; lane l reads segment (l / LPS) of the instruction, piece l % LPS.
// Segment j of the whole launch starts at j * STRIDE bytes and only its first SEG bytes are ever read.
template <int SEG, int STRIDE>
__global__ __launch_bounds__(256) void seg_k(const char* __restrict__ in, float* out, size_t nseg) {
    __shared__ __attribute__((aligned(16))) char lds[4096];
    constexpr int LPS = SEG / 16;                  // lanes per segment
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const size_t seg_per_inst = 64 / LPS;
    const size_t n_inst = nseg / seg_per_inst;
    for (size_t it = blockIdx.x * 4 + wave; it < n_inst; it += (size_t)gridDim.x * 4) {
        const char* base = in + it * seg_per_inst * STRIDE;           // wave-uniform 64-bit base, 32-bit lane offset
        bufld16(base, lds + wave * 1024, (lane / LPS) * STRIDE + (lane % LPS) * 16, 0);
    }
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (((const float*)lds)[threadIdx.x] == 123.456f) out[0] = 1.f;
}

template <int C>
__global__ __launch_bounds__(256) void chunkmajor_k(const char* __restrict__ in, float* out, size_t nblocks) {
    __shared__ __attribute__((aligned(16))) char lds[4096];
    constexpr int PIX = 1024, ROW = C * 4;           // pixels per block, bytes per pixel
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    for (size_t blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
        const char* base = in + blk * (size_t)PIX * ROW;
        for (int chunk = 0; chunk < C / 8; ++chunk) {
            // 1024 pixels x 2 lanes = 2048 lanes = 8 instructions per wave
            for (int it = 0; it < 8; ++it) {
                const int pixel = (it * 4 + wave) * 32 + (lane >> 1);
                bufld16(base, lds + wave * 1024, pixel * ROW + (lane & 1) * 16, chunk * 32);
            }
            __builtin_amdgcn_s_waitcnt(0);
            __syncthreads();                         // one chunk pass of the whole block after the other, as the kernel's K loop
        }
    }
    if (((const float*)lds)[threadIdx.x] == 123.456f) out[0] = 1.f;
}

// conv_f43_k's requests into a channel-chunk-major plane (round 6, conv_f43.h LAY & 1): a tile's halo = 34 rows x 36 pixels x 32 bytes, the rows
// 1 152 contiguous bytes one plane row (648 pixels = 20 736 bytes) apart; lane order as the kernel's asrc[]: (16-byte half, column group of 4, column
// phase, row).  The tiles partition the planes, so every byte is read exactly once.
__global__ __launch_bounds__(256) void p8row_k(const char* __restrict__ in, float* out, size_t ntiles) {
    __shared__ __attribute__((aligned(16))) char lds[4096];
    constexpr int PITCH = 648, RAW_PIECES = 34 * 4 * 9 * 2;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), tid = threadIdx.x;
    for (size_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const size_t trow = t / 18, tcol = t % 18;
        const char* base = in + (trow * 34 * PITCH + tcol * 36) * 32;
        for (int it = 0; it < 10; ++it) {
            int e = it * 256 + tid;
            if (e >= RAW_PIECES) continue;
            const int half = e & 1, xd = (e >> 1) % 9, ph = ((e >> 1) / 9) & 3, y = (e >> 1) / 36;
            bufld16(base, lds + wave * 1024, ((y * PITCH + 4 * xd + ph) * 8 + 4 * half) * 4, 0);
        }
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
    }
    if (((const float*)lds)[threadIdx.x] == 123.456f) out[0] = 1.f;
}

template <int C>
void run_chunkmajor(const char* buf, float* out, size_t bytes) {
    const size_t nblocks = bytes / ((size_t)1024 * C * 4);
    hipLaunchKernelGGL((chunkmajor_k<C>), dim3(256), dim3(256), 0, 0, buf, out, nblocks);
    CK(hipDeviceSynchronize());
    printf("chunkmajor_k<%d>: %zu blocks, %zu useful bytes\n", C, nblocks, nblocks * 1024 * C * 4);
}

template <int SEG, int STRIDE>
void run_seg(const char* buf, float* out, size_t bytes) {
    const size_t nseg = bytes / STRIDE;
    hipLaunchKernelGGL((seg_k<SEG, STRIDE>), dim3(2048), dim3(256), 0, 0, buf, out, nseg);
    CK(hipDeviceSynchronize());
    printf("seg_k<%d,%d>: %zu segments, %zu useful bytes\n", SEG, STRIDE, nseg, nseg * SEG);
}

int main() {
    const size_t bytes = (size_t)2 << 30;          // 2 GiB: eight times the Infinity Cache
    char* buf; float* out;
    CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&out, 64));
    CK(hipMemset(buf, 0, bytes));
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(stream_k, dim3(4096), dim3(256), 0, 0, (const float4*)buf, out, bytes / 16);
    CK(hipDeviceSynchronize());
    printf("stream_k: %zu useful bytes\n", bytes);
    run_seg<64, 256>(buf, out, bytes);
    run_seg<64, 512>(buf, out, bytes);
    run_seg<64, 1024>(buf, out, bytes);
    run_seg<64, 2048>(buf, out, bytes);
    run_seg<256, 256>(buf, out, bytes);
    run_seg<32, 256>(buf, out, bytes);
    run_seg<32, 512>(buf, out, bytes);
    run_chunkmajor<64>(buf, out, bytes);
    run_chunkmajor<128>(buf, out, bytes);
    run_chunkmajor<256>(buf, out, bytes);
    {
        const size_t ntiles = bytes / ((size_t)34 * 648 * 32) * 18;
        hipLaunchKernelGGL(p8row_k, dim3(256), dim3(256), 0, 0, buf, out, ntiles);
        CK(hipDeviceSynchronize());
        printf("p8row_k: %zu tiles, %zu useful bytes\n", ntiles, ntiles * 34 * 36 * 32);
    }
    return 0;
}
