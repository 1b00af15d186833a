// prep_kernels.h — once-per-video / once-per-style kernels (not on the per-frame path):
// per-channel statistics over (batch,H,W) (InstanceNorm.compute,
// test/style_network_global.py:59-77; EncoderStyle.cal_mean_std :304-315), the pointwise
// normalise / residual / AdaIN steps of Decoder.compute (:383-439), FilterPredictor's
// 64->1024 FC (:161-172), the fold of the predicted 32x32 filters into the KernelFilter
// convolutions, and the OIHW -> kernel-native weight repack.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// ---- weight repack: OIHW -> [Cout/BN][Cin/16][TAPS][BN][16 floats, 16B pieces XOR (j>>2)&3]
__global__ void pack_conv_k(const float* __restrict__ w, float* __restrict__ dst, int Cout, int Cin, int taps, int BN) {
    const size_t total = (size_t)Cout * Cin * taps;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        // i indexes the destination
        size_t r = i;
        const int e = r & 3; r >>= 2;
        const int qs = r & 3; r >>= 2;          // stored piece position
        const int j = r % BN; r /= BN;
        const int tap = r % taps; r /= taps;
        const int nchunks = Cin >> 4;
        const int chunk = r % nchunks; r /= nchunks;
        const int n_tile = (int)r;
        const int q = qs ^ ((j >> 2) & 3);      // logical piece
        const int co = n_tile * BN + j, ci = chunk * 16 + q * 4 + e;
        dst[i] = w[((size_t)co * Cin + ci) * taps + tap];
    }
}

// conv_first weights: OIHW [64][3][3][3] -> [27][64] rows (ky*3+kx)*3 + c
__global__ void pack_first_k(const float* __restrict__ w, float* __restrict__ dst) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 27 * 64) {
        const int co = i & 63, k = i >> 6, c = k % 3, tap = k / 3;
        dst[i] = w[(co * 3 + c) * 9 + tap];
    }
}

// conv_first on a greyscaled frame: the three input channels are affine functions of ONE grey value g,
// x_c = (g - mean_c) / sd_c, so sum_c w[tap][c] x_c = W1[tap] g + W0[tap]:
// dst[0..575] = W1 [9][64], dst[576..1151] = W0 [9][64], dst[1152..1215] = bias + sum_tap W0 (all nine taps valid).
__global__ void pack_first_grey_k(const float* __restrict__ w, const float* __restrict__ bias, float* __restrict__ dst) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const float mean[3] = {0.485f, 0.456f, 0.406f}, sd[3] = {0.229f, 0.224f, 0.225f};
    if (i < 9 * 64) {
        const int co = i & 63, tap = i >> 6;
        float w1 = 0.f, w0 = 0.f;
        for (int c = 0; c < 3; ++c) {
            const float wc = w[(co * 3 + c) * 9 + tap];
            w1 += wc / sd[c];
            w0 -= wc * (mean[c] / sd[c]);
        }
        dst[i] = w1;
        dst[576 + i] = w0;
    }
    if (i < 64) {
        float b = bias[i];
        for (int tap = 0; tap < 9; ++tap)
            for (int c = 0; c < 3; ++c) b -= w[(i * 3 + c) * 9 + tap] * (mean[c] / sd[c]);
        dst[1152 + i] = b;
    }
}

// conv_last weights: OIHW [3][64][3][3] -> the A-operand registers of conv_last_k: [blk 2][c 4][lane 64][s 4], the value
// lane (m = lane & 15, kq = lane >> 4) feeds to the MFMA of channel chunk c, step s, row block blk:
// W[n = 16 blk + m][channel 16 c + 4 kq + s] with n = tap * 3 + rgb (27 rows, padded to 32 with zeros)
__global__ void pack_last_k(const float* __restrict__ w, float* __restrict__ dst) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 2 * 4 * 64 * 4) {
        const int s = i & 3, lane = (i >> 2) & 63, c = (i >> 8) & 3, blk = i >> 10;
        const int n = 16 * blk + (lane & 15), ch = 16 * c + 4 * (lane >> 4) + s;
        dst[i] = (n < 27) ? w[((n % 3) * 64 + ch) * 9 + n / 3] : 0.f;
    }
}

// ---- folds of a dynamic 32x32 filter F (weight[out=i][in=j] = F[i][j], quirk Q2) -----
// down' = F . Wd :  Wd'[o][ci][t] = sum_m F[o][m] Wd[m][ci][t],  bd'[o] = sum_m F[o][m] bd[m]
// blockIdx.y = state set (the grouped multi-style launches fold a group's sets at once): F, Wout, bout advance by
// f_stride, 32 * CinT, b_stride floats per set
__global__ void fold_down_k(const float* __restrict__ F, const float* __restrict__ Wd, const float* __restrict__ bd,
                            float* __restrict__ Wout, float* __restrict__ bout, int CinT /* Cin*taps */, int f_stride, int b_stride) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;   // over 32*CinT
    F += (size_t)blockIdx.y * f_stride; Wout += (size_t)blockIdx.y * 32 * CinT; bout += (size_t)blockIdx.y * b_stride;
    if (i < 32 * CinT) {
        const int o = i / CinT, k = i - o * CinT;
        float s = 0.f;
        for (int m = 0; m < 32; ++m) s += F[o * 32 + m] * Wd[(size_t)m * CinT + k];
        Wout[i] = s;
    }
    if (i < 32) {
        float s = 0.f;
        for (int m = 0; m < 32; ++m) s += F[i * 32 + m] * bd[m];
        bout[i] = s;
    }
}
// up' = Wu . F :  Wu'[o][j][t] = sum_m Wu[o][m][t] F[m][j]
__global__ void fold_up_k(const float* __restrict__ F, const float* __restrict__ Wu, float* __restrict__ Wout, int Cout, int f_stride) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;   // over Cout*32*9
    F += (size_t)blockIdx.y * f_stride; Wout += (size_t)blockIdx.y * Cout * 32 * 9;      // blockIdx.y = state set
    if (i < Cout * 32 * 9) {
        const int t = i % 9, j = (i / 9) & 31, o = i / (9 * 32);
        float s = 0.f;
        for (int m = 0; m < 32; ++m) s += Wu[((size_t)o * 32 + m) * 9 + t] * F[m * 32 + j];
        Wout[i] = s;
    }
}

// ---- FilterPredictor FC: out[1024] = W[1024][64] . [cmean(32), smean(32)] + b ---------
__global__ void fc_filter_k(const float* __restrict__ W, const float* __restrict__ bias, const float* __restrict__ cmean,
                            const float* __restrict__ smean, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 1024) {
        float s = 0.f;
        for (int k = 0; k < 32; ++k) s += W[i * 64 + k] * cmean[k];
        for (int k = 0; k < 32; ++k) s += W[i * 64 + 32 + k] * smean[k];
        out[i] = s + bias[i];
    }
}

// ---- per-channel statistics over the valid pixels of a ring-layout tensor -------------
// pass 0: sum;  pass 1 (needs mean): centred sum of squares, min, max.
// part layout: [nblk][3][C] doubles.
struct StatP {
    const float* x;
    int B, H, W, C;
    const float* mean;   // pass 1
    double* part;
    int pass;
    int pix_per_blk;
};

__global__ __launch_bounds__(256) void chan_stat_k(const StatP p) {
    // a thread owns 4 consecutive channels (16-byte loads) of every nsub-th pixel of the block's pixel range; the
    // (b, y, x) decomposition happens once, then advances incrementally
    __shared__ double s_sum[256][4];
    __shared__ float s_mn[256][4], s_mx[256][4];
    const int tid = threadIdx.x;
    const int CQ = p.C >> 2;                       // channel quads (C is a multiple of 4)
    const int Qb = CQ < 64 ? CQ : 64;
    const int nsub = 256 / Qb;
    const int ql = tid % Qb, sub = tid / Qb;
    const long npix = (long)p.B * p.H * p.W;
    const long p0 = (long)blockIdx.x * p.pix_per_blk;
    long p1 = p0 + p.pix_per_blk;
    if (p1 > npix) p1 = npix;
    for (int cg = 0; cg < CQ; cg += Qb) {
        const int c = 4 * (cg + ql);
        double s[4] = {0.0, 0.0, 0.0, 0.0};
        f32x4 mn = {3.4e38f, 3.4e38f, 3.4e38f, 3.4e38f}, mx = {-3.4e38f, -3.4e38f, -3.4e38f, -3.4e38f};
        f32x4 m = {0.f, 0.f, 0.f, 0.f};
        if (p.pass) m = *(const f32x4*)(p.mean + c);
        if (sub < nsub) {
            long q = p0 + sub;
            int x = (int)(q % p.W);
            long r = q / p.W;
            int y = (int)(r % p.H);
            long b = r / p.H;
            for (; q < p1; q += nsub) {
                const f32x4 v = *(const f32x4*)(p.x + ((b * (p.H + 2) + y + 1) * (p.W + 2) + x + 1) * (long)p.C + c);
                if (p.pass) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float d = v[e] - m[e];
                        s[e] += (double)(d * d);
                        mn[e] = fminf(mn[e], v[e]);
                        mx[e] = fmaxf(mx[e], v[e]);
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) s[e] += (double)v[e];
                }
                x += nsub;
                while (x >= p.W) { x -= p.W; if (++y >= p.H) { y = 0; ++b; } }
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) { s_sum[tid][e] = s[e]; s_mn[tid][e] = mn[e]; s_mx[tid][e] = mx[e]; }
        __syncthreads();
        if (sub == 0) {
            for (int k = 1; k < nsub; ++k)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    s[e] += s_sum[k * Qb + ql][e];
                    mn[e] = fminf(mn[e], s_mn[k * Qb + ql][e]);
                    mx[e] = fmaxf(mx[e], s_mx[k * Qb + ql][e]);
                }
            double* o = p.part + (size_t)blockIdx.x * 3 * p.C;
#pragma unroll
            for (int e = 0; e < 4; ++e) { o[c + e] = s[e]; o[p.C + c + e] = mn[e]; o[2 * p.C + c + e] = mx[e]; }
        }
        __syncthreads();
    }
}

// final reduce.  mode 0: mean only -> out[c]
//               mode 1: content stats -> n[4][C] = mean, rsqrt(ss/N + 1e-8), (min-mean)*rstd, (max-mean)*rstd
//               mode 2: style stats   -> sty[2][C] = mean, sqrt(ss/(N-1) + 1e-5)
// launch: C/16 blocks of 256 threads — 16 channels x 16 lanes that each fold every 16th partial, then one LDS step
__global__ __launch_bounds__(256) void chan_final_k(const double* __restrict__ part, int nblk, int C, double N, int mode,
                                                    const float* __restrict__ mean_in, float* __restrict__ out) {
    __shared__ double s_s[16][16];
    __shared__ float s_mn[16][16], s_mx[16][16];
    const int cl = threadIdx.x & 15, kl = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    double s = 0.0;
    float mn = 3.4e38f, mx = -3.4e38f;
    if (c < C)
        for (int k = kl; k < nblk; k += 16) {
            const double* o = part + (size_t)k * 3 * C;
            s += o[c];
            mn = fminf(mn, (float)o[C + c]);
            mx = fmaxf(mx, (float)o[2 * C + c]);
        }
    s_s[kl][cl] = s; s_mn[kl][cl] = mn; s_mx[kl][cl] = mx;
    __syncthreads();
    if (kl != 0 || c >= C) return;
    for (int k = 1; k < 16; ++k) { s += s_s[k][cl]; mn = fminf(mn, s_mn[k][cl]); mx = fmaxf(mx, s_mx[k][cl]); }
    if (mode == 0) {
        out[c] = (float)(s / N);
    } else if (mode == 1) {
        const float m = mean_in[c];
        const float var = (float)(s / N) + 1e-8f;
        const float r = 1.0f / sqrtf(var);
        out[c] = m; out[C + c] = r; out[2 * C + c] = (mn - m) * r; out[3 * C + c] = (mx - m) * r;
    } else {
        const float var = (float)(s / (N - 1.0)) + 1e-5f;
        out[c] = mean_in[c]; out[C + c] = sqrtf(var);
    }
}

// ---- one-pass statistics (frame mode: per-frame InstanceNorm, test/style_network_frame.py:39-43) ------------------
// One read of the tensor.  Every thread accumulates sum(x - s) and sum((x - s)^2) in fp64 about its own first value s
// (inside the data's range, so nothing cancels), turns them into (n, mean, M2) and the block merges its threads with the
// pairwise update of Chan et al.; per block [n | mean | M2] x C doubles go to `part` ([nblk][3][C]).
__global__ __launch_bounds__(256) void chan_stat1_k(const StatP p) {
    // a block walks whole image rows (blockIdx.x, += gridDim.x): inside a row a thread's pixels are a constant stride
    // apart, so the loads are independent of the accumulation and of each other (the compiler keeps several in flight)
    __shared__ double s_n[256], s_mean[256][4], s_m2[256][4];
    const int tid = threadIdx.x;
    const int CQ = p.C >> 2;
    const int Qb = CQ < 64 ? CQ : 64;
    const int nsub = 256 / Qb;
    const int ql = tid % Qb, sub = tid / Qb;
    const int nrows = p.B * p.H;
    for (int cg = 0; cg < CQ; cg += Qb) {
        const int c = 4 * (cg + ql);
        double sd[4] = {0.0, 0.0, 0.0, 0.0}, sq[4] = {0.0, 0.0, 0.0, 0.0};
        f32x4 s0 = {0.f, 0.f, 0.f, 0.f};
        long n = 0;
        if (sub < nsub)
            for (int r = blockIdx.x; r < nrows; r += gridDim.x) {
                const int b = r / p.H, y = r - b * p.H;
                const float* row = p.x + (((long)b * (p.H + 2) + y + 1) * (p.W + 2) + 1) * (long)p.C + c;
                if (n == 0 && sub < p.W) s0 = *(const f32x4*)(row + (long)sub * p.C);
#pragma unroll 4
                for (int x = sub; x < p.W; x += nsub) {
                    const f32x4 v = *(const f32x4*)(row + (long)x * p.C);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { const double d = (double)(v[e] - s0[e]); sd[e] += d; sq[e] += d * d; }
                    ++n;
                }
            }
        s_n[tid] = (double)n;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const double m = n ? sd[e] / (double)n : 0.0;
            s_mean[tid][e] = (double)s0[e] + m;
            s_m2[tid][e] = n ? sq[e] - sd[e] * m : 0.0;
        }
        __syncthreads();
        if (sub == 0) {
            double na = s_n[tid], mean[4], m2[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) { mean[e] = s_mean[tid][e]; m2[e] = s_m2[tid][e]; }
            for (int k = 1; k < nsub; ++k) {
                const int o = k * Qb + ql;
                const double nb = s_n[o];
                if (nb == 0.0) continue;
                const double nn = na + nb;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const double d = s_mean[o][e] - mean[e];
                    mean[e] += d * (nb / nn);
                    m2[e] += s_m2[o][e] + d * d * (na * nb / nn);
                }
                na = nn;
            }
            double* o = p.part + (size_t)blockIdx.x * 3 * p.C;
#pragma unroll
            for (int e = 0; e < 4; ++e) { o[c + e] = na; o[p.C + c + e] = mean[e]; o[2 * p.C + c + e] = m2[e]; }
        }
        __syncthreads();
    }
}
// merge of the block partials; out = norm params [4][C]: mean, rsqrt(M2/N + 1e-8), -3e38, +3e38 (frame mode does not clamp).
// launch: C/4 blocks of 256 threads = 4 channels x 64 lanes; a lane folds every 64th block, then a butterfly of Chan merges.
__global__ __launch_bounds__(256) void chan_stat1_final_k(const double* __restrict__ part, int nblk, int C, float* __restrict__ out) {
    const int cl = threadIdx.x >> 6, kl = threadIdx.x & 63;
    const int c = blockIdx.x * 4 + cl;
    double na = 0.0, mean = 0.0, m2 = 0.0;
    if (c < C)
        for (int k = kl; k < nblk; k += 64) {
            const double* o = part + (size_t)k * 3 * C;
            const double nb = o[c];
            if (nb == 0.0) continue;
            const double nn = na + nb, d = o[C + c] - mean;
            mean += d * (nb / nn);
            m2 += o[2 * C + c] + d * d * (na * nb / nn);
            na = nn;
        }
    for (int off = 32; off > 0; off >>= 1) {        // the wave IS the channel: xor butterfly, both partners compute the same merge
        const double nb = __shfl_xor(na, off), mb = __shfl_xor(mean, off), qb = __shfl_xor(m2, off);
        const double nn = na + nb;
        if (nn > 0.0) {
            const double d = mb - mean;
            const double wb = nb / nn;
            m2 += qb + d * d * (na * wb);
            mean += d * wb;
        }
        na = nn;
    }
    if (kl != 0 || c >= C) return;
    const float m = (float)mean;
    const float var = (float)(m2 / na) + 1e-8f;
    out[c] = m; out[C + c] = 1.0f / sqrtf(var); out[2 * C + c] = -3.0e38f; out[3 * C + c] = 3.0e38f;
}

// ---- FilterPredictor means without the convolution (frame mode) ---------------------------------------------------
// FilterPredictor needs only mean_{HW} of down_sample(x) (test/style_network_frame.py:53-62).  The mean of a zero-padded
// 3x3 convolution is linear in nine RECTANGLE SUMS of its input: tap (dy,dx) sees every pixel except the row / column
// that falls outside:  mean_o = bias_o + (1/HW) sum_{c,tap} w[o][c][tap] * S[tap][c],
//   S[(dy,dx)][c] = sum of x[y'][x'][c] over y' in [max(0,dy), H+min(0,dy)), x' likewise.
// rect_sums_k: one block per channel quad of a [1,H,W,C] ring tensor -> S[9][C] (floats, accumulated in fp64).
__global__ __launch_bounds__(256) void rect_sums_k(const float* __restrict__ x, int H, int W, int C, float* __restrict__ S) {
    // grid (C/4, NY): block (q, j) sums every NY-th 256-pixel slice of channel quad q into the partial S[j][9][C]
    __shared__ double red[9][4][4];     // [tap][wave][e]
    const int c = blockIdx.x * 4;
    double acc[9][4];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[t][e] = 0.0;
#pragma unroll 4
    for (int i = blockIdx.y * 256 + threadIdx.x; i < H * W; i += 256 * gridDim.y) {
        const int y = i / W, xx = i - y * W;
        const f32x4 v = *(const f32x4*)(x + ((size_t)(y + 1) * (W + 2) + xx + 1) * C + c);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int dy = t / 3 - 1, dx = t % 3 - 1;
            const bool in = (dy < 0 ? y < H - 1 : (dy > 0 ? y > 0 : true)) && (dx < 0 ? xx < W - 1 : (dx > 0 ? xx > 0 : true));
            if (in) {
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[t][e] += (double)v[e];
            }
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            double a = acc[t][e];
            for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
            if (lane == 0) red[t][wave][e] = a;
        }
    __syncthreads();
    if (threadIdx.x < 36) {
        const int t = threadIdx.x / 4, e = threadIdx.x & 3;
        S[((size_t)blockIdx.y * 9 + t) * C + c + e] = (float)(red[t][0][e] + red[t][1][e] + red[t][2][e] + red[t][3][e]);
    }
}
// out[o] = bias[o] + (1/HW) sum_{c,tap} w[o][c][tap] S[tap][c]   (w OIHW [32][C][3][3]); one block per output
__global__ __launch_bounds__(256) void pred_mean_k(const float* __restrict__ w, const float* __restrict__ bias, const float* __restrict__ S, int nparts, int C,
                                                   double inv_hw, float* __restrict__ out) {
    __shared__ double red[4];
    const int o = blockIdx.x;
    double a = 0.0;
    for (int i = threadIdx.x; i < C * 9; i += 256) {
        const int c = i / 9, t = i - c * 9;
        float sv = S[t * C + c];
        for (int j = 1; j < nparts; ++j) sv += S[((size_t)j * 9 + t) * C + c];
        a += (double)w[((size_t)o * C + c) * 9 + t] * (double)sv;
    }
    for (int k = 32; k > 0; k >>= 1) a += __shfl_xor(a, k);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) out[o] = (float)((red[0] + red[1] + red[2] + red[3]) * inv_hw + (double)bias[o]);
}

// ---- streaming statistics (compute() over groups of sampled frames) -------------------------------------------
// Per channel the running (mean, M2 = centred sum of squares, min, max) of everything seen so far lives in
// acc[4][C] doubles.  A group contributes its own two-pass partials — part0: sums, part1: squares about the group's
// fp32 mean m32, min, max — which are merged with the pairwise update of Chan et al.:
//   d = mean_b - mean_a;  mean = mean_a + d n_b/n;  M2 = M2_a + M2_b + d^2 n_a n_b/n       (n = n_a + n_b)
// (M2_b about the exact group mean = sum (x - m32)^2 - n_b (mean_b - m32)^2).  part1 == nullptr: means only.
// launch: C/16 blocks of 256 threads, as chan_final_k.
__global__ __launch_bounds__(256) void chan_merge_k(const double* __restrict__ part0, const double* __restrict__ part1, int nblk, int C,
                                                    double n_b, const float* __restrict__ m32, double* __restrict__ acc, double n_a) {
    __shared__ double s_s[16][16], s_q[16][16];
    __shared__ float s_mn[16][16], s_mx[16][16];
    const int cl = threadIdx.x & 15, kl = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    double s = 0.0, q = 0.0;
    float mn = 3.4e38f, mx = -3.4e38f;
    if (c < C)
        for (int k = kl; k < nblk; k += 16) {
            s += part0[(size_t)k * 3 * C + c];
            if (part1) {
                const double* o = part1 + (size_t)k * 3 * C;
                q += o[c];
                mn = fminf(mn, (float)o[C + c]);
                mx = fmaxf(mx, (float)o[2 * C + c]);
            }
        }
    s_s[kl][cl] = s; s_q[kl][cl] = q; s_mn[kl][cl] = mn; s_mx[kl][cl] = mx;
    __syncthreads();
    if (kl != 0 || c >= C) return;
    for (int k = 1; k < 16; ++k) { s += s_s[k][cl]; q += s_q[k][cl]; mn = fminf(mn, s_mn[k][cl]); mx = fmaxf(mx, s_mx[k][cl]); }
    const double mean_b = s / n_b;
    double M2_b = 0.0;
    if (part1) { const double e = mean_b - (double)m32[c]; M2_b = q - n_b * e * e; if (M2_b < 0.0) M2_b = 0.0; }
    if (n_a == 0.0) {
        acc[c] = mean_b; acc[C + c] = M2_b; acc[2 * C + c] = mn; acc[3 * C + c] = mx;
    } else {
        const double n = n_a + n_b, d = mean_b - acc[c];
        acc[c] += d * (n_b / n);
        acc[C + c] += M2_b + d * d * (n_a * n_b / n);
        acc[2 * C + c] = fmin(acc[2 * C + c], (double)mn);
        acc[3 * C + c] = fmax(acc[3 * C + c], (double)mx);
    }
}

// mode 0: out[c] = mean;  mode 1: out = norm params [4][C] (mean, rsqrt(M2/N + 1e-8), (min-mean) rstd, (max-mean) rstd),
// with the same fp32 roundings as chan_final_k.
__global__ void chan_finish_k(const double* __restrict__ acc, int C, double N, int mode, float* __restrict__ out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float m = (float)acc[c];
    if (mode == 0) { out[c] = m; return; }
    const float var = (float)(acc[C + c] / N) + 1e-8f;
    const float r = 1.0f / sqrtf(var);
    out[c] = m; out[C + c] = r; out[2 * C + c] = ((float)acc[2 * C + c] - m) * r; out[3 * C + c] = ((float)acc[3 * C + c] - m) * r;
}

// ---- pointwise step of the preparation pass (valid pixels only, ring preserved) -------
struct PointP {
    const float* x; float* y;      // y may alias x
    int B, H, W, C;
    const float* mean; const float* scale;   // normalise: (v-mean)*scale, or /scale when div
    int div;
    const float* res; int res_mode;          // 0 none, 1 same-res image 0 broadcast, 2 half-res per batch
    int Hr, Wr;
    const float* smean; const float* sstd;   // affine after
    const float* lo; const float* hi;        // clamp of the normalised value (saved-stat forward), may be null
    int segs;                                // blocks per image row (small tensors still fill the chip)
};
__global__ __launch_bounds__(256) void pointwise_k(const PointP p) {
    // Blocks walk (image row, row segment) pairs; 256 is a multiple of C/4, so a thread keeps ONE channel quad: its
    // per-channel parameters are loaded once, and inside a row its pixels are a constant stride apart.
    const int CQ = p.C >> 2;
    const int nrows = p.B * p.H;
    const int sg = blockIdx.x % p.segs, rstep = gridDim.x / p.segs;
    const int t0 = sg * 256 + threadIdx.x;
    const int c4 = (t0 % CQ) * 4, x0 = t0 / CQ, xstep = p.segs * 256 / CQ;
    f32x4 mean = {0.f, 0.f, 0.f, 0.f}, scale = {1.f, 1.f, 1.f, 1.f}, lo, hi, smean, sstd;
    if (p.mean) { mean = *(const f32x4*)(p.mean + c4); scale = *(const f32x4*)(p.scale + c4); }
    if (p.lo) { lo = *(const f32x4*)(p.lo + c4); hi = *(const f32x4*)(p.hi + c4); }
    if (p.smean) { smean = *(const f32x4*)(p.smean + c4); sstd = *(const f32x4*)(p.sstd + c4); }
    for (int r = blockIdx.x / p.segs; r < nrows; r += rstep) {
        const int b = r / p.H, y = r - b * p.H;
        const long row = (((long)b * (p.H + 2) + y + 1) * (p.W + 2) + 1) * (long)p.C + c4;
        const long rrow = (p.res_mode == 1 ? ((long)(y + 1) * (p.Wr + 2) + 1) * (long)p.C
                         : p.res_mode == 2 ? (((long)b * (p.Hr + 2) + (y >> 1) + 1) * (p.Wr + 2) + 1) * (long)p.C : 0) + c4;
#pragma unroll 2
        for (int x = x0; x < p.W; x += xstep) {
            const long idx = row + (long)x * p.C;
            f32x4 v = *(const f32x4*)(p.x + idx);
            f32x4 rv = {0.f, 0.f, 0.f, 0.f};
            if (p.res_mode == 1) rv = *(const f32x4*)(p.res + rrow + (long)x * p.C);
            if (p.res_mode == 2) rv = *(const f32x4*)(p.res + rrow + (long)(x >> 1) * p.C);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float tv = v[e];
                if (p.mean) tv = p.div ? (tv - mean[e]) / scale[e] : (tv - mean[e]) * scale[e];
                if (p.lo) tv = fminf(hi[e], fmaxf(lo[e], tv));
                if (p.res_mode) tv += rv[e];
                if (p.smean) tv = tv * sstd[e] + smean[e];
                v[e] = tv;
            }
            *(f32x4*)(p.y + idx) = v;
        }
    }
}

// split-K finish of the 512->32 KernelFilter convolution: part holds `split` partial sums per pixel as channels
// [32 s, 32 s + 32) of a [.., 32 * split] ring-layout tensor; out[c] = LeakyReLU(sum_s part[32 s + c]) over every pixel
// slot of the buffers (the ring is zero in every part and LeakyReLU(0) = 0).
__global__ void sum_parts_lrelu_k(const float* __restrict__ part, float* __restrict__ out, int split, long npix) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < npix * 8; i += (long)gridDim.x * blockDim.x) {
        const long px = i >> 3;
        const int c4 = (int)(i & 7) * 4;
        const float* src = part + px * 32 * split + c4;
        f32x4 v = *(const f32x4*)src;
        for (int k = 1; k < split; ++k) v += *(const f32x4*)(src + 32 * k);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = v[e] >= 0.f ? v[e] : v[e] * 0.2f;
        *(f32x4*)(out + px * 32 + c4) = v;
    }
}

// identity saved-statistics entry (mean 0, rstd 1, lo -inf, hi +inf): frame mode has no second
// normalisation after the filters (test/style_network_frame.py AdaIN_filter: results*std+mean)
__global__ void identity_norm_k(float* __restrict__ n, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) { n[c] = 0.f; n[C + c] = 1.f; n[2 * C + c] = -3.0e38f; n[3 * C + c] = 3.0e38f; }
}

// blended state for multi-style interpolation: out = sum_s w[s] * state_s
// ("Multi-style Interpolation/style_network.py":41-45,137-138,354-356)
// (blend_states_k: blockIdx.y = image of a grouped multi-style launch, its weights w[image][style], its state set out + image * count)
struct BlendManyP { const float* st[8]; float w[16][8]; int n; float* out; int count; };
struct BlendP { const float* st[8]; float w[8]; int n; float* out; int count; };
__global__ void blend_states_k(const BlendManyP p) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    if (i < p.count) {
        float s = 0.f;
        for (int k = 0; k < p.n; ++k) s += p.w[b][k] * p.st[k][i];      // the same sum, in the same order, as blend_state_k
        p.out[(size_t)b * p.count + i] = s;
    }
}
__global__ void blend_state_k(const BlendP p) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < p.count) {
        float s = 0.f;
        for (int k = 0; k < p.n; ++k) s += p.w[k] * p.st[k][i];
        p.out[i] = s;
    }
}
