// conv_wino.h — 3x3 / pad-1 / stride-1 convolution by Winograd F(2x2,3x3) on the fp32 matrix cores.
//
// Same layers and the same fused epilogues as conv_mfma_k (vgg19.features convs,
// test/style_network_global.py:271-281; ResidualBlock.conv2 :104,119-122), but the 9-tap
// contraction is replaced by 16 element-wise GEMMs in the transform domain:
//   Y = A^T [ sum_c (G g_c G^T) .* (B^T d_c B) ] A          (2x2 outputs from a 4x4 input patch)
// 16 multiplies per 4 outputs instead of 36: 2.25x fewer MFMA FLOPs on a path that is bound by the
// fp32 MFMA rate.  Exact in real arithmetic; in fp32 the rounding differs from the direct form at the
// 1e-7 level (checked against the same goldens / tolerances).
//
// Mapping (one 256-thread workgroup = 16x16 output pixels x 32 output channels):
//  * a wave owns 16 Winograd tiles (2 x 8 tiles = 4 x 16 pixels) and ALL 16 transform positions, using
//    v_mfma_f32_16x16x4_f32: M = 16 output channels, N = 16 tiles, K = 4 input channels per step.
//  * lane (t = lane&15, q = lane>>4) reads the 4x4 raw patch of tile t for input channels 4q..4q+3 of
//    the staged 16-channel chunk (16 ds_read_b128), computes B^T d B in registers (32 float4 adds) and
//    the result IS its MFMA "B" operand for all 16 positions: the transformed input never touches LDS
//    or HBM.  Transformed weights U (pre-computed once at weight-pack time) are the "A" operand.
//  * the accumulator of a lane holds, for its tile, 4 consecutive output channels x 2 blocks x 16
//    positions, so A^T M A, bias, activation, saved-stat normalise, residual, AdaIN and the 2x2 max
//    pool (the Winograd output tile IS the pooling window) are all in-register; stores are 16-byte.
//  * per 16-channel chunk: one barrier, 18x18x16 raw halo + 16x32x16 U block staged by buffer_load..lds,
//    double buffered (114 KB LDS, one workgroup per CU).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "conv_mfma.h"

#define WINO_RAW_BYTES 24576            /* 18*18 pixels * 64 B, rounded up to 6 x 256 pieces */
#define WINO_U_BYTES 32768              /* 16 positions * 32 couts * 16 channels * 4 B */
#define WINO_SMEM_BYTES (2 * WINO_RAW_BYTES + 2 * WINO_U_BYTES)

template <int EPI, int ABL = 0>
__global__ __launch_bounds__(256) void conv_wino_k(const ConvP p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63, t = lane & 15, q = lane >> 4;
    const int tr = t >> 3, tc = t & 7;

    int bx = blockIdx.x;
    const int tx = bx % p.tiles_x;
    bx /= p.tiles_x;
    const int ty = bx % p.tiles_y;
    const int b = bx / p.tiles_y;
    const int n_tile = blockIdx.y;
    const int y0 = ty * 16, x0 = tx * 16;
    const int nchunks = p.Cin >> 4;

    const float* in_b = p.in + (size_t)b * (size_t)(p.Hi + 2) * (p.Wi + 2) * p.Cin;
    int asrc[6];
#pragma unroll
    for (int it = 0; it < 6; ++it) {
        const int e = it * 256 + tid;
        int pp = e >> 2;
        const int qq = e & 3;
        if (pp >= 324) pp = 0;
        const int hy = pp / 18, hx = pp - hy * 18;
        asrc[it] = (((y0 + hy) * (p.Wi + 2) + (x0 + hx)) * p.Cin + 4 * (qq ^ ((pp >> 2) & 3))) * 4;
    }
    const float* w_tile = p.wpk + (size_t)n_tile * nchunks * (16 * 32 * 16);

    auto stage = [&](int chunk) {
        char* rdst = smem + (chunk & 1) * WINO_RAW_BYTES;
        char* udst = smem + 2 * WINO_RAW_BYTES + (chunk & 1) * WINO_U_BYTES;
#pragma unroll
        for (int it = 0; it < 8; ++it) bufld16(w_tile, udst + (it * 256 + wave * 64) * 16, tid * 16 + it * 4096, chunk * WINO_U_BYTES);
#pragma unroll
        for (int it = 0; it < 6; ++it) bufld16(in_b, rdst + (it * 256 + wave * 64) * 16, asrc[it], chunk * 64);
    };

    // LDS byte offsets of this lane's 4x4 raw patch (pixel index pp, 16-byte piece q, XOR swizzle)
    int offD[4][4];
#pragma unroll
    for (int dy = 0; dy < 4; ++dy)
#pragma unroll
        for (int dx = 0; dx < 4; ++dx) {
            const int pp = (4 * wave + 2 * tr + dy) * 18 + 2 * tc + dx;
            offD[dy][dx] = pp * 64 + ((q ^ ((pp >> 2) & 3)) << 4);
        }
    // U fragment: row = pos*32 + nb*16 + t ; (row>>2)&3 == (t>>2)&3 because 32 | pos*32 and 16 | nb*16
    const int offU = t * 64 + ((q ^ ((t >> 2) & 3)) << 4);

    f32x4 acc[16][2];
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) acc[i][nb] = f32x4{0.f, 0.f, 0.f, 0.f};

    stage(0);
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        if (!(ABL & 2) || chunk == 0) __syncthreads();
        if (!(ABL & 1) && chunk + 1 < nchunks) stage(chunk + 1);
        const char* raw = smem + (chunk & 1) * WINO_RAW_BYTES;
        const char* ub = smem + 2 * WINO_RAW_BYTES + (chunk & 1) * WINO_U_BYTES;
        // ---- input transform V = B^T d B, in registers (4 channels per lane)
        f32x4 v[4][4];
        {
            f32x4 tmp[4][4];
#pragma unroll
            for (int dx = 0; dx < 4; ++dx) {
                const f32x4 d0 = *(const f32x4*)(raw + offD[0][dx]), d1 = *(const f32x4*)(raw + offD[1][dx]);
                const f32x4 d2 = *(const f32x4*)(raw + offD[2][dx]), d3 = *(const f32x4*)(raw + offD[3][dx]);
                tmp[0][dx] = d0 - d2; tmp[1][dx] = d1 + d2; tmp[2][dx] = d2 - d1; tmp[3][dx] = d1 - d3;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[r][0] = tmp[r][0] - tmp[r][2]; v[r][1] = tmp[r][1] + tmp[r][2];
                v[r][2] = tmp[r][2] - tmp[r][1]; v[r][3] = tmp[r][1] - tmp[r][3];
            }
        }
        // ---- 16 positions x 2 cout blocks x 4 k-steps
#pragma unroll
        for (int pos = 0; pos < 16; ++pos) {
            const f32x4 u0 = *(const f32x4*)(ub + pos * 2048 + offU);
            const f32x4 u1 = *(const f32x4*)(ub + pos * 2048 + 1024 + offU);
            const f32x4 vv = v[pos >> 2][pos & 3];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                acc[pos][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(u0[s], vv[s], acc[pos][0], 0, 0, 0);
                acc[pos][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(u1[s], vv[s], acc[pos][1], 0, 0, 0);
            }
        }
    }

    // ---- output transform + fused epilogue (all in registers)
    const int Ho = (EPI & E_POOL) ? (p.H >> 1) : p.H, Wo = (EPI & E_POOL) ? (p.W >> 1) : p.W;
    float* out_b = p.out + (size_t)b * (size_t)(Ho + 2) * (Wo + 2) * p.Cout;
    const float* res_b = nullptr;
    if (EPI & (E_RES | E_RES_UPS)) res_b = p.res + (size_t)b * (size_t)(p.Hr + 2) * (p.Wr + 2) * p.Cout;
    const int yb = y0 + 4 * wave + 2 * tr, xb = x0 + 2 * tc;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        const int co = n_tile * 32 + nb * 16 + 4 * q;
        f32x4 Y[2][2];
        {
            f32x4 T[2][4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                T[0][c] = acc[0 + c][nb] + acc[4 + c][nb] + acc[8 + c][nb];
                T[1][c] = acc[4 + c][nb] - acc[8 + c][nb] - acc[12 + c][nb];
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                Y[i][0] = T[i][0] + T[i][1] + T[i][2];
                Y[i][1] = T[i][1] - T[i][2] - T[i][3];
            }
        }
        const f32x4 bias = *(const f32x4*)(p.bias + co);
        f32x4 m1, r1, lo1, hi1, m2, r2, lo2, hi2, smean, sstd;
        if (EPI & E_NORM1) {
            m1 = *(const f32x4*)(p.n1 + co); r1 = *(const f32x4*)(p.n1 + p.Cout + co);
            lo1 = *(const f32x4*)(p.n1 + 2 * p.Cout + co); hi1 = *(const f32x4*)(p.n1 + 3 * p.Cout + co);
        }
        if (EPI & E_NORM2) {
            m2 = *(const f32x4*)(p.n2 + co); r2 = *(const f32x4*)(p.n2 + p.Cout + co);
            lo2 = *(const f32x4*)(p.n2 + 2 * p.Cout + co); hi2 = *(const f32x4*)(p.n2 + 3 * p.Cout + co);
            smean = *(const f32x4*)(p.sty + co); sstd = *(const f32x4*)(p.sty + p.Cout + co);
        }
        f32x4 pooled;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int y = yb + i, x = xb + j;
                const bool valid = (y < p.H) && (x < p.W);
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float tv = Y[i][j][e] + bias[e];
                    if (EPI & E_RELU) tv = fmaxf(tv, 0.f);
                    if (EPI & E_LRELU) tv = (tv >= 0.f) ? tv : tv * 0.2f;
                    if (EPI & E_NORM1) {
                        tv = (tv - m1[e]) * r1[e];
                        tv = fminf(hi1[e], fmaxf(lo1[e], tv));
                    }
                    o[e] = tv;
                }
                if (EPI & (E_RES | E_RES_UPS)) {
                    if (valid) {
                        const int ry = (EPI & E_RES_UPS) ? (y >> 1) : y, rx = (EPI & E_RES_UPS) ? (x >> 1) : x;
                        o += *(const f32x4*)(res_b + ((ry + 1) * (p.Wr + 2) + rx + 1) * p.Cout + co);
                    }
                }
                if (EPI & E_NORM2) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float tv = (o[e] - m2[e]) * r2[e];
                        tv = fminf(hi2[e], fmaxf(lo2[e], tv));
                        o[e] = tv * sstd[e] + smean[e];
                    }
                }
                if (EPI & E_POOL) {
                    if (i == 0 && j == 0) pooled = o;
                    else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) pooled[e] = fmaxf(pooled[e], o[e]);
                    }
                } else if (valid) {
                    if (ABL & 4) { if (o[0] == 123.456f) out_b[co] = o[0]; }
                    else *(f32x4*)(out_b + ((y + 1) * (p.W + 2) + x + 1) * p.Cout + co) = o;
                }
            }
        if (EPI & E_POOL) {
            const int y2 = yb >> 1, x2 = xb >> 1;
            if (y2 < Ho && x2 < Wo) {
                if (ABL & 4) { if (pooled[0] == 123.456f) out_b[co] = pooled[0]; }
                else *(f32x4*)(out_b + ((y2 + 1) * (Wo + 2) + x2 + 1) * p.Cout + co) = pooled;
            }
        }
    }
}

// Weight transform U = G g G^T, G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]], packed as
// [Cout/32][Cin/16][pos 16][32 couts][16 floats, 16-byte pieces XOR (cout>>2)&3].
__global__ void pack_wino_k(const float* __restrict__ w, float* __restrict__ dst, int Cout, int Cin) {
    const size_t total = (size_t)Cout * Cin * 16;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        size_t r = i;
        const int e = r & 3; r >>= 2;
        const int qs = r & 3; r >>= 2;
        const int j = r & 31; r >>= 5;
        const int pos = r & 15; r >>= 4;
        const int nchunks = Cin >> 4;
        const int chunk = r % nchunks; r /= nchunks;
        const int n_tile = (int)r;
        const int qq = qs ^ ((j >> 2) & 3);
        const int co = n_tile * 32 + j, ci = chunk * 16 + qq * 4 + e;
        const float* g = w + ((size_t)co * Cin + ci) * 9;
        const int pr = pos >> 2, pc = pos & 3;
        float rowv[3];   // (G g)[pr][kx]
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const float g0 = g[0 * 3 + kx], g1 = g[1 * 3 + kx], g2 = g[2 * 3 + kx];
            rowv[kx] = pr == 0 ? g0 : (pr == 1 ? 0.5f * (g0 + g1 + g2) : (pr == 2 ? 0.5f * (g0 - g1 + g2) : g2));
        }
        const float u = pc == 0 ? rowv[0] : (pc == 1 ? 0.5f * (rowv[0] + rowv[1] + rowv[2]) : (pc == 2 ? 0.5f * (rowv[0] - rowv[1] + rowv[2]) : rowv[2]));
        dst[i] = u;
    }
}
