// conv_thin.h — the two HBM-bound ends of the per-frame path, on the vector ALUs.
//
// conv_first_k: uint8 BGR HWC frame -> [greyscale] -> ImageNet normalise -> conv3x3 3->64
//   + bias + ReLU.  Fuses numpy2tensor/transform_image (test/framework.py:26-35),
//   TransformerNet.RGB2Gray (test/style_network_global.py:487-497, quirk Q5) and
//   vgg19.features[0:2].  Reads 3 B/pixel, writes 256 B/pixel.
// conv_last_k: conv3x3 64->3 + bias (Decoder.slice1, :341,450) fused with
//   transform_back_image/tensor2numpy (test/framework.py:39-49): *std+mean, clamp(0,1),
//   *255, RGB->BGR, HWC float32.  Reads 256 B/pixel, writes 12 (+12) B/pixel.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "conv_mfma.h"   // bufld16

struct FirstP {
    const uint8_t* img;   // [B][H][W][3] BGR
    int H, W, B;
    float* out;           // [B,H,W,64] ring layout
    const float* w;       // [27][64]: row (ky*3+kx)*3 + c_rgb
    const float* bias;    // [64]
    int grey;             // 1: content frame (greyscaled), 0: style image (colour)
    int tiles_x, tiles_y;
    const float* wg;      // grey fold (pack_first_grey_k): W1 [9][64] | W0 [9][64] | bias + sum W0 [64]; null: never fold
    // optional on-device ReshapeTool.process (test/generate_real_video.py:61-83): img is the UNPADDED [B][src_H][src_W][3]
    // frame and padded pixel (y, x) reads source pixel (reflect(y - pad_top), reflect(x - pad_left)), edge-inclusive
    // (cv2.BORDER_REFLECT = numpy 'symmetric').  src_H == 0: img already has the padded geometry.
    int src_H, src_W, pad_top, pad_left;
    int p8;               // 1: `out` is channel-chunk-major [B][8 chunks][H+2][W+8][8], pixel x at column x + 4 (conv_f43.h LAY: what conv1_2 on conv_f43_k reads 12-19 % faster); same values
};

// symmetric (edge-inclusive) reflection of t into [0, n), any distance
__device__ __forceinline__ int reflect_sym(int t, int n) {
    const int period = 2 * n;
    t %= period;
    if (t < 0) t += period;
    return t < n ? t : period - 1 - t;
}

__global__ __launch_bounds__(256) void conv_first_k(const FirstP p) {
    __shared__ __attribute__((aligned(16))) float s_in[18 * 18 * 4];
    __shared__ __attribute__((aligned(16))) float s_w[27 * 64];
    const int tid = threadIdx.x;
    int bx = blockIdx.x;
    const int tx = bx % p.tiles_x;
    bx /= p.tiles_x;
    const int ty = bx % p.tiles_y;
    const int b = bx / p.tiles_y;
    const int y0 = ty * 16, x0 = tx * 16;
    const int SH = p.src_H ? p.src_H : p.H, SW = p.src_H ? p.src_W : p.W;
    const uint8_t* img = p.img + (size_t)b * SH * SW * 3;
    auto src_px = [&](int y, int x) {       // padded-frame pixel -> address in the source frame
        if (p.src_H) { y = reflect_sym(y - p.pad_top, SH); x = reflect_sym(x - p.pad_left, SW); }
        return img + ((size_t)y * SW + x) * 3;
    };

    const float mean[3] = {0.485f, 0.456f, 0.406f}, sd[3] = {0.229f, 0.224f, 0.225f};
    // A greyscaled frame feeds the three input channels with affine functions of one value g (RGB2Gray + the
    // re-normalisation, test/style_network_global.py:487-497): 9 multiplies per output instead of 27.  The fold
    // assumes all nine taps inside the image, so tiles that touch the image border take the general path below.
    if (p.grey && p.wg && y0 > 0 && x0 > 0 && y0 + 16 < p.H && x0 + 16 < p.W) {
        float* s_g = s_in;      // [18][18] grey values
        for (int i = tid; i < 18 * 18; i += 256) {
            const int hy = i / 18, hx = i - hy * 18;
            const uint8_t* px = src_px(y0 + hy - 1, x0 + hx - 1);
            float d[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) d[c] = (((float)px[2 - c] / 255.0f - mean[c]) / sd[c]) * sd[c] + mean[c];   // as the reference rounds it
            s_g[i] = d[2] * 0.299f + d[1] * 0.587f + d[0] * 0.114f;   // :493 (sic)
        }
        __syncthreads();
        if (p.p8) {      // the same arithmetic under another thread mapping: thread = (half of a chunk: 4 channels, pixel column, chunk), a loop over the 16 rows —
                         // consecutive lanes store consecutive 16-byte pieces: 32 lanes = 16 pixels x 32 bytes = 512 contiguous bytes of one chunk plane
            const int half = tid & 1, pc = (tid >> 1) & 15, k8 = tid >> 5, q4 = (k8 * 2 + half) * 4;
            f32x4 w1[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) w1[k] = *(const f32x4*)&p.wg[k * 64 + q4];
            const f32x4 bias = *(const f32x4*)&p.wg[1152 + q4];
            float* const plane = p.out + ((size_t)b * 8 + k8) * (size_t)(p.H + 2) * (p.W + 8) * 8 + ((size_t)(y0 + 1) * (p.W + 8) + x0 + pc + 4) * 8 + half * 4;      // rows pitched W + 8, pixel x at column x + 4: every 512-byte run starts on a line (conv_f43.h P8_PAD / P8_COL0)
            float g[3][3];
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c) g[r + 1][c] = s_g[r * 18 + pc + c];
#pragma unroll
            for (int prow = 0; prow < 16; ++prow) {
#pragma unroll
                for (int c = 0; c < 3; ++c) { g[0][c] = g[1][c]; g[1][c] = g[2][c]; g[2][c] = s_g[(prow + 2) * 18 + pc + c]; }
                f32x4 acc = bias;
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) acc += w1[ky * 3 + kx] * g[ky][kx];
                f32x4 r;
#pragma unroll
                for (int e = 0; e < 4; ++e) r[e] = fmaxf(acc[e], 0.f);
                __builtin_nontemporal_store(r, (f32x4*)&plane[(size_t)prow * (p.W + 8) * 8]);
            }
            return;
        }
        const int q = tid & 15, prow = tid >> 4;
        f32x4 w1[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) w1[k] = *(const f32x4*)&p.wg[k * 64 + q * 4];
        const f32x4 bias = *(const f32x4*)&p.wg[1152 + q * 4];
        float g[3][18];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 18; ++c) g[r][c] = s_g[(prow + r) * 18 + c];
        float* orow = p.out + (size_t)b * (size_t)(p.H + 2) * (p.W + 2) * 64 + ((size_t)(y0 + prow + 1) * (p.W + 2) + x0 + 1) * 64 + q * 4;
#pragma unroll
        for (int pc = 0; pc < 16; ++pc) {
            f32x4 acc = bias;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) acc += w1[ky * 3 + kx] * g[ky][pc + kx];
            f32x4 r;
#pragma unroll
            for (int e = 0; e < 4; ++e) r[e] = fmaxf(acc[e], 0.f);
            __builtin_nontemporal_store(r, (f32x4*)&orow[pc * 64]);
        }
        return;
    }
    for (int i = tid; i < 27 * 64; i += 256) s_w[i] = p.w[i];
    for (int i = tid; i < 18 * 18; i += 256) {
        const int hy = i / 18, hx = i - hy * 18;
        const int y = y0 + hy - 1, x = x0 + hx - 1;
        float o[3] = {0.f, 0.f, 0.f};
        if (y >= 0 && y < p.H && x >= 0 && x < p.W) {
            const uint8_t* px = src_px(y, x);
            float n[3];   // normalised, RGB order (framework.py:27,33-34)
#pragma unroll
            for (int c = 0; c < 3; ++c) n[c] = ((float)px[2 - c] / 255.0f - mean[c]) / sd[c];
            if (p.grey) {
                float d[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) d[c] = n[c] * sd[c] + mean[c];
                const float g = d[2] * 0.299f + d[1] * 0.587f + d[0] * 0.114f;   // :493 (sic)
#pragma unroll
                for (int c = 0; c < 3; ++c) o[c] = (g - mean[c]) / sd[c];
            } else {
#pragma unroll
                for (int c = 0; c < 3; ++c) o[c] = n[c];
            }
        }
        *(f32x4*)&s_in[i * 4] = f32x4{o[0], o[1], o[2], 0.f};
    }
    __syncthreads();

    if (p.p8) {      // border tiles of a channel-chunk-major output: the general arithmetic below under the thread mapping of the fast path above
        const int half = tid & 1, pc = (tid >> 1) & 15, k8 = tid >> 5, q4 = (k8 * 2 + half) * 4;
        f32x4 w[27];
#pragma unroll
        for (int k = 0; k < 27; ++k) w[k] = *(const f32x4*)&s_w[k * 64 + q4];
        const f32x4 bias = *(const f32x4*)&p.bias[q4];
        float* const plane = p.out + ((size_t)b * 8 + k8) * (size_t)(p.H + 2) * (p.W + 8) * 8 + half * 4;
        const int x = x0 + pc;
        for (int prow = 0; prow < 16; ++prow) {
            f32x4 acc = bias;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const f32x4 v = *(const f32x4*)&s_in[((prow + ky) * 18 + pc + kx) * 4];
                    const int k = (ky * 3 + kx) * 3;
                    acc += w[k] * v[0];
                    acc += w[k + 1] * v[1];
                    acc += w[k + 2] * v[2];
                }
            const int y = y0 + prow;
            if (y < p.H && x < p.W) {
                f32x4 r;
#pragma unroll
                for (int e = 0; e < 4; ++e) r[e] = fmaxf(acc[e], 0.f);
                *(f32x4*)&plane[((size_t)(y + 1) * (p.W + 8) + x + 4) * 8] = r;
            }
        }
        return;
    }
    const int q = tid & 15;      // output channels 4q..4q+3
    const int prow = tid >> 4;   // pixel row inside the tile
    f32x4 w[27];
#pragma unroll
    for (int k = 0; k < 27; ++k) w[k] = *(const f32x4*)&s_w[k * 64 + q * 4];
    const f32x4 bias = *(const f32x4*)&p.bias[q * 4];
    float* out_b = p.out + (size_t)b * (size_t)(p.H + 2) * (p.W + 2) * 64;
    const int y = y0 + prow;
    for (int pc = 0; pc < 16; ++pc) {
        f32x4 acc = bias;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const f32x4 v = *(const f32x4*)&s_in[((prow + ky) * 18 + pc + kx) * 4];
                const int k = (ky * 3 + kx) * 3;
                acc += w[k] * v[0];
                acc += w[k + 1] * v[1];
                acc += w[k + 2] * v[2];
            }
        const int x = x0 + pc;
        if (y < p.H && x < p.W) {
            f32x4 r;
#pragma unroll
            for (int e = 0; e < 4; ++e) r[e] = fmaxf(acc[e], 0.f);
            *(f32x4*)&out_b[((size_t)(y + 1) * (p.W + 2) + x + 1) * 64 + q * 4] = r;
        }
    }
}

struct LastP {
    const float* in;      // [B,H,W,64] ring layout
    int H, W, B;
    const float* w;       // pack_last_k: [blk 2][c 4][lane 64][s 4] MFMA A operands of the 27 x 64 tap-rgb matrix
    const float* bias;    // [4]
    float* out_img;       // [B][H][W][3] BGR float32 0..255
    float* out_pre;       // optional [B][H][W][3] RGB pre-clamp (normalised units), may be null
    int tiles_x, tiles_y;
    // optional on-device crop (generate_real_video.py:167): out_img is [B][out_H][out_W][3] and receives the window
    // that starts at (crop_top, crop_left) of the padded frame.  out_H == 0: the whole padded frame.
    int out_H, out_W, crop_top, crop_left;
    int ty0, tx0;         // first tile row / column of the computed window (tiles_x, tiles_y count its tiles)
};

// GEMM first, taps second.  out[y][x][rgb] = sum_tap sum_c w[tap][c][rgb] in[y+ky][x+kx][c] is evaluated as
//   G[p][tap*3+rgb] = sum_c W[tap*3+rgb][c] in[p][c]     for the 18 x 18 halo pixels p of a 16 x 16 tile — one
//                     [27 -> 32] x [64] x [324] GEMM on v_mfma_f32_16x16x4_f32, the input straight from global memory
//                     into the B operand (each value is read ONCE), the weights resident in 32 registers per lane,
//   out[y][x][rgb] = bias + sum_tap G[(y+ky, x+kx)][tap*3+rgb]      27 shifted LDS reads per output value (planes of
//                     324 floats: consecutive lanes = consecutive pixels, conflict-free).
// Round 2's direct form re-read every input value from LDS for each of its nine taps (288 ds_read_b128 per wave and
// tile, LDS-bound at 0.38 of the HBM peak); here LDS carries 27 floats per halo pixel once in and once out.
// Workgroups are persistent (the weight registers are loaded once) and walk the tiles of the window.
#define LAST_GP 330       /* floats per G plane (324 used); 4 planes = 8 banks on: the four row groups of a wave write disjoint banks */
__global__ __launch_bounds__(256) void conv_last_k(const LastP p) {
    __shared__ __attribute__((aligned(16))) float s_g[27 * LAST_GP];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int t = lane & 15, kq = lane >> 4;
    f32x4 wreg[2][4];
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
        for (int c = 0; c < 4; ++c) wreg[blk][c] = *(const f32x4*)(p.w + ((blk * 4 + c) * 64 + lane) * 4);
    const float mean[3] = {0.485f, 0.456f, 0.406f}, sd[3] = {0.229f, 0.224f, 0.225f};
    const float bias[3] = {p.bias[0], p.bias[1], p.bias[2]};
    const int ntiles = p.tiles_x * p.tiles_y * p.B;
    const int OH = p.out_H ? p.out_H : p.H, OW = p.out_H ? p.out_W : p.W;
    // Tile walk.  Workgroup w runs on XCD w % 8 (observed dispatch order; locality only): each XCD gets ONE contiguous band of
    // the tile list, and the workgroups of an XCD take consecutive tiles of it — horizontally adjacent tiles run on the same
    // XCD at the same time and the rows above / below a round or two apart, so the 18 x 18 halos (1.27x the tensor) are
    // fetched from HBM once and hit that XCD's L2 afterwards (round 3's round-robin over the XCDs measured 1.22x).
    int tile, tile_end, tile_step;
    if ((gridDim.x & 7) == 0) {
        const int band = (ntiles + 7) >> 3, xcd = blockIdx.x & 7;
        tile = xcd * band + (blockIdx.x >> 3); tile_step = gridDim.x >> 3;
        tile_end = (xcd + 1) * band < ntiles ? (xcd + 1) * band : ntiles;
    } else { tile = blockIdx.x; tile_step = gridDim.x; tile_end = ntiles; }
    auto origin_of = [&](int tl) {      // 64-bit tile origin (halo pixel (0,0) = ring pixel (y0, x0)); lane offsets are tile-relative and 32-bit
        const int tx = tl % p.tiles_x, ty = (tl / p.tiles_x) % p.tiles_y, b = tl / (p.tiles_x * p.tiles_y);
        return p.in + ((size_t)b * (size_t)(p.H + 2) + (ty + p.ty0) * 16) * (size_t)(p.W + 2) * 64 + (size_t)((tx + p.tx0) * 16) * 64 + 4 * kq;
    };
    auto halo_off = [&](int g) {
        int P = 16 * g + t;
        P = P < 324 ? P : 323;
        const int hy = P / 18, hx = P - 18 * hy;
        return (hy * (p.W + 2) + hx) * 64;
    };
    f32x4 x[4], xn[4];
    if (tile < tile_end) {               // the first tile's first group; every later tile's is requested under the previous tile's taps
        const float* src = origin_of(tile) + halo_off(wave);
#pragma unroll
        for (int c = 0; c < 4; ++c) x[c] = *(const f32x4*)(src + 16 * c);
    }
    for (; tile < tile_end; tile += tile_step) {
        const int tx = tile % p.tiles_x, ty = (tile / p.tiles_x) % p.tiles_y, b = tile / (p.tiles_x * p.tiles_y);
        const int y0 = (ty + p.ty0) * 16, x0 = (tx + p.tx0) * 16;
        const float* in_t = origin_of(tile);
        const bool more = tile + tile_step < tile_end;
        const float* in_n = more ? origin_of(tile + tile_step) : in_t;
        // ---- G = W . in over the halo: 21 groups of 16 pixels, round-robin over the waves
        for (int g = wave; g < 21; g += 4) {
            if (g + 4 < 21) {                 // next group's pixels while this one is multiplied
                const float* src = in_t + halo_off(g + 4);
#pragma unroll
                for (int c = 0; c < 4; ++c) xn[c] = *(const f32x4*)(src + 16 * c);
            } else if (more) {                // the NEXT tile's first group: its latency lies under this tile's taps and stores
                const float* src = in_n + halo_off(wave);
#pragma unroll
                for (int c = 0; c < 4; ++c) xn[c] = *(const f32x4*)(src + 16 * c);
            }
            f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int blk = 0; blk < 2; ++blk) acc[blk] = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[blk][c][s], x[c][s], acc[blk], 0, 0, 0);
            // D: register i of lane (pixel t, row group kq) = row n = 16 blk + 4 kq + i of that pixel
            const int P = 16 * g + t;
            if (P < 324) {
#pragma unroll
                for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int n = 16 * blk + 4 * kq + i;
                        if (n < 27) s_g[n * LAST_GP + P] = acc[blk][i];
                    }
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) x[c] = xn[c];
        }
        __syncthreads();
        // ---- taps: one output pixel per lane (wave w = tile rows 4w .. 4w+3)
        const int row = 4 * wave + (lane >> 4), col = lane & 15;
        float o[3] = {bias[0], bias[1], bias[2]};
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap - 3 * ky;
            const float* gp = &s_g[(tap * 3) * LAST_GP + (row + ky) * 18 + col + kx];
#pragma unroll
            for (int c = 0; c < 3; ++c) o[c] += gp[c * LAST_GP];
        }
        __syncthreads();                      // G is free for the next tile
        const int y = y0 + row, xx = x0 + col;
        if (y < p.H && xx < p.W) {
            typedef float f32x3 __attribute__((ext_vector_type(3)));
            if (p.out_pre) *(f32x3*)(p.out_pre + (((size_t)b * p.H + y) * p.W + xx) * 3) = f32x3{o[0], o[1], o[2]};
            float im[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) im[c] = fminf(fmaxf(o[c] * sd[c] + mean[c], 0.f), 1.f) * 255.f;
            const int cy = p.out_H ? y - p.crop_top : y, cx = p.out_H ? xx - p.crop_left : xx;
            // RGB -> BGR; a lane stores its pixel's 12 bytes, a row of the tile leaves as one 192-byte burst
            if (cy >= 0 && cy < OH && cx >= 0 && cx < OW) *(f32x3*)(p.out_img + (((size_t)b * OH + cy) * OW + cx) * 3) = f32x3{im[2], im[1], im[0]};
        }
    }
}
