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
            *(f32x4*)&orow[pc * 64] = r;
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
    const float* w;       // [9][64][4]: tap, cin, cout(rgb, padded to 4)
    const float* bias;    // [4]
    float* out_img;       // [B][H][W][3] BGR float32 0..255
    float* out_pre;       // optional [B][H][W][3] RGB pre-clamp (normalised units), may be null
    int tiles_x, tiles_y;
    // optional on-device crop (generate_real_video.py:167): out_img is [B][out_H][out_W][3] and receives the window
    // that starts at (crop_top, crop_left) of the padded frame.  out_H == 0: the whole padded frame.
    int out_H, out_W, crop_top, crop_left;
    int ty0, tx0;         // first tile row / column of the computed window (tiles_x, tiles_y count its tiles)
};

// Matrix-core form: v_mfma_f32_4x4x1_16B_f32 runs 16 independent 4x4 outer products per instruction
// (lane 4b+i feeds A-row i of block b, lane 4b+j feeds B-column j; D[i][j] of block b = register i of lane
// 4b+j), so a wave evaluates 64 pixels x 4 output channels per K step with no padding to 16/32 channels:
// A = the pixel's input value (one lane per pixel), B = w[k][rgb] replicated over the blocks.
// LDS: 18x18x16-channel halo chunks (double-buffered LDS-DMA) + the whole weight table [tap][chunk][4][16].
__global__ __launch_bounds__(256) void conv_last_k(const LastP p) {
    __shared__ __attribute__((aligned(16))) float s_in[2][18 * 18 * 16];
    __shared__ __attribute__((aligned(16))) float s_w[9 * 4 * 4 * 16];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    int bx = blockIdx.x;
    const int tx = bx % p.tiles_x;
    bx /= p.tiles_x;
    const int ty = bx % p.tiles_y, b = bx / p.tiles_y;
    const int y0 = (ty + p.ty0) * 16, x0 = (tx + p.tx0) * 16;
    // 64-bit tile origin, tile-relative 32-bit lane offsets: no limit on the frame size from the descriptor's range
    const float* in_t = p.in + ((size_t)b * (size_t)(p.H + 2) + y0) * (size_t)(p.W + 2) * 64 + (size_t)x0 * 64;

    auto stage = [&](int chunk, int buf) {
        for (int e = tid; e < 18 * 18 * 4; e += 256) {
            const int pp = e >> 2, qq = e & 3;
            const int hy = pp / 18, hx = pp - hy * 18;
            const int off = ((hy * (p.W + 2) + hx) * 64 + 4 * (qq ^ ((pp >> 2) & 3))) * 4;
            bufld16(in_t, (char*)&s_in[buf][0] + (e - (tid & 63)) * 16, off, chunk * 64);
        }
    };
    stage(0, 0);
    // weights: p.w is [9][64][4] (tap, cin, rgb-padded); LDS image [tap][chunk][j][16 cin]
    for (int i = tid; i < 9 * 64 * 4; i += 256) {
        const int j = i & 3, ci = (i >> 2) & 63, tap = i >> 8;
        s_w[((tap * 4 + (ci >> 4)) * 4 + j) * 16 + (ci & 15)] = p.w[i];
    }
    const int prow = lane >> 4, pcol = lane & 15;   // this lane's pixel as the A operand: wave rows 4w..4w+3
    const int jb = lane & 3;                        // this lane's output channel as the B operand / D column
    f32x4 acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int chunk = 0; chunk < 4; ++chunk) {
        __syncthreads();
        if (chunk + 1 < 4) stage(chunk + 1, (chunk + 1) & 1);
        const char* buf = (const char*)&s_in[chunk & 1][0];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap - ky * 3;
            const int pp = (4 * wave + prow + ky) * 18 + pcol + kx;
            const int sw = (pp >> 2) & 3;
            const float* wt = &s_w[((tap * 4 + chunk) * 4 + jb) * 16];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 a = *(const f32x4*)(buf + pp * 64 + ((q ^ sw) << 4));
                const f32x4 w4 = *(const f32x4*)(wt + q * 4);
#pragma unroll
                for (int s = 0; s < 4; ++s) acc[q] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[s], w4[s], acc[q], 0, 0, 0);
            }
        }
    }
    const f32x4 r = acc[0] + acc[1] + acc[2] + acc[3];
    // D: register i of lane 4*blk + j  =  pixel 4*blk + i of this wave, output channel j
    const float mean[3] = {0.485f, 0.456f, 0.406f}, sd[3] = {0.229f, 0.224f, 0.225f};
    const int jc = jb < 3 ? jb : 0;
    const float bias = p.bias[jc], mj = mean[jc], sj = sd[jc];
    const int blk = lane >> 2;
    // Interior tiles (all 16 x 16 pixels delivered): the wave's 4 rows x 16 pixels x BGR go through LDS and leave as
    // 48 lanes x 16 bytes = four contiguous 192-byte row pieces, instead of 4-byte stores 48 bytes apart — whole
    // bursts for HBM and, when the output is mapped host memory (zero-copy, look-ahead tickets), for PCIe.
    const int oy0 = p.out_H ? y0 - p.crop_top : y0, ox0 = p.out_H ? x0 - p.crop_left : x0;      // tile origin in the delivered image
    const int OH = p.out_H ? p.out_H : p.H, OW = p.out_H ? p.out_W : p.W;
    const bool whole = y0 + 16 <= p.H && x0 + 16 <= p.W && oy0 >= 0 && ox0 >= 0 && oy0 + 16 <= OH && ox0 + 16 <= OW;     // block-uniform
    if (whole) {
        // s_in[0] is free: the barrier at the top of the last chunk (which lives in s_in[1]) was passed by every wave
        float* so = &s_in[0][0] + wave * 192;     // [4 rows][16 pixels][3]
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float t = r[i] + bias;
            if (p.out_pre && jb < 3) {
                const int pix = 4 * blk + i;
                p.out_pre[(((size_t)b * p.H + y0 + 4 * wave + (pix >> 4)) * p.W + x0 + (pix & 15)) * 3 + jb] = t;
            }
            float im = t * sj + mj;
            im = fminf(fmaxf(im, 0.f), 1.f) * 255.f;
            if (jb < 3) so[(4 * blk + i) * 3 + 2 - jb] = im;      // RGB -> BGR
        }
        // the wave's own 768 bytes: written and read by the same wave (LDS operations of a wave complete in order)
        if (lane < 48) {
            const int row = lane / 12, piece = lane - row * 12;
            const f32x4 v = *(const f32x4*)(so + row * 48 + piece * 4);
            *(f32x4*)(p.out_img + (((size_t)b * OH + oy0 + 4 * wave + row) * OW + ox0) * 3 + piece * 4) = v;
        }
        return;
    }
    if (jb < 3) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int pix = 4 * blk + i;
            const int y = y0 + 4 * wave + (pix >> 4), x = x0 + (pix & 15);
            if (y < p.H && x < p.W) {
                const size_t o = (((size_t)b * p.H + y) * p.W + x) * 3;
                const float t = r[i] + bias;
                if (p.out_pre) p.out_pre[o + jb] = t;
                float im = t * sj + mj;
                im = fminf(fmaxf(im, 0.f), 1.f) * 255.f;
                if (!p.out_H) {
                    p.out_img[o + 2 - jb] = im;   // RGB -> BGR
                } else {
                    const int cy = y - p.crop_top, cx = x - p.crop_left;
                    if (cy >= 0 && cy < p.out_H && cx >= 0 && cx < p.out_W)
                        p.out_img[(((size_t)b * p.out_H + cy) * p.out_W + cx) * 3 + 2 - jb] = im;
                }
            }
        }
    }
}
