// conv_wino_split.h — Winograd F(2x2,3x3) 3x3 convolution, 8 waves, transform rows split inside wave pairs.
//
// Same layers, work items, LDS staging and item stream as conv_wino_k<.., UPS = 0> (conv_wino.h; reference:
// vgg19.features convs, test/style_network_global.py:271-281; ResidualBlock.conv2 :104,119-122; KernelFilter
// convs :210-217).  What differs is how the two waves of a SIMD divide one tile group (16 Winograd tiles):
//   conv_wino_k, NW = 8 : each wave owns 16 of the 32 output channels and ALL 16 transform positions — both waves
//                         read the whole 4x4 patch and run the whole B^T d B (duplicated);
//   here                : each wave owns all 32 output channels and the 8 positions of TWO transform rows —
//                         wave "half 0" rows r = 0,1 (needs patch rows 0..2), wave "half 1" rows r = 3,2 (patch
//                         rows 3..1).  12 patch reads and 32 packed VALU ops per chunk instead of 16 and 64.
// With the patch rows taken in the order (b0,b1,b2) = rows (0,1,2) / (3,2,1) both halves run ONE instruction stream:
//   c0 = b0 - b2          (r = 0: d0 - d2      | r = 3: d3 - d1 = -(B^T d)[3], the sign returns in the output step)
//   c1 = b1 + sgn * b2    (r = 1: d1 + d2      | r = 2: d2 - d1),   sgn = +1 / -1 per wave
// followed by the usual 4 -> 4 pass along the other axis.  The accumulators hold M[r][k] for the wave's two rows;
// Y = A^T M A couples the four rows, so at the end of an item the partner's row sum T'[r][j] = sum_k M[r][k] A[k][j]
// crosses through LDS (one 16-byte vector per output pixel and channel block, one barrier per item):
//   output row i = half:  Y[i][j] = T'c0[j] + sgn * T'c1[j] + (partner's T'c1[j])
// (max-pool layers need all four pixels of a tile in one lane: wave `half` finishes channel block `half` and gets
// the partner's two row sums of that block instead — the same four vectors, the epilogue stays balanced).
#pragma once
#include "conv_wino.h"

// WSPLIT_ABL (default 0; tools/wsplit_bench.hip builds one binary per value): microbenchmark switches of THIS kernel —
// 1 no LDS-DMA after an item's first stage, 2 no K-loop barriers, 4 no stores, 32 no epilogue.  The library is compiled
// with 0: every hook is a discarded constexpr branch (the instruction stream is unchanged, checked with hipcc -S).
#ifndef WSPLIT_ABL
#define WSPLIT_ABL 0
#endif

#define WSPLIT_XCH_BYTES 32768          /* exchange: 8 waves x 4 vectors x 64 lanes x 16 B */
#define WSPLIT_SMEM_BYTES (WinoGeo<8, 0>::SMEM + WSPLIT_XCH_BYTES)   /* 146 KB */

// c = a * s + b on the packed-fp32 pipe
__device__ __forceinline__ f32x4 f4fma(const f32x4 a, const float s, const f32x4 b) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const f32x2 sv = {s, s};
    f32x2 lo, hi;
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(lo) : "v"(__builtin_shufflevector(a, a, 0, 1)), "v"(sv), "v"(__builtin_shufflevector(b, b, 0, 1)));
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(hi) : "v"(__builtin_shufflevector(a, a, 2, 3)), "v"(sv), "v"(__builtin_shufflevector(b, b, 2, 3)));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3);
}

struct WSplitSched {
    // MFMA-loop iteration i (= local position, 0..7): first up to three patch pieces, then the U fragments of
    // position i+2.  LDS returns in order, so U(i) complete => everything issued before it is complete.
    static constexpr int pieces_in(int i) { return (i >= 0 && i < 4) ? 3 : 0; }
    static constexpr int issued(int i) { return pieces_in(i) + (i + 2 < 8 ? 2 : 0); }
    static constexpr int younger(int i) { return i == 0 ? 2 + issued(0) : issued(i - 1) + issued(i); }
};

// PERIMG = 1: per-image state (ConvP::par_bstride / bias_bstride / w_bstride; the grouped multi-style decoder).  A
// separate instantiation: the extra scalar state costs the shared-state kernels ~1 % (profiles/r03_wsplit_ab2.txt).
template <int EPI, int PERIMG = 0>
__global__ __launch_bounds__(512, 1) void conv_wino_split_k(const ConvP p) {
    using G = WinoGeo<8, 0>;
    using S = WSplitSched;
    constexpr int RAW_BYTES = G::RAW_BYTES, U_BYTES = G::U_BYTES, U_LDS = G::U_LDS, NT = G::NT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform: LDS-DMA bases stay in SGPRs
    const int lane = tid & 63, t = lane & 15, q = lane >> 4;
    const int tr = t >> 3, tc = t & 7;
    const int tg = wave >> 1;                    // tile group: output rows 4*tg .. 4*tg+3 of the workgroup tile
    const int half = wave & 1;                   // transform rows {0,1} / {3,2}; finishes output row `half` of each tile
    const float sgn = half ? -1.f : 1.f;
    const int nchunks = p.Cin >> 4;              // even
    const int cs = p.cstride ? p.cstride : p.Cin;    // channels per pixel in memory (split K: the launch contracts a slice of them)
    const int n_ntiles = p.Cout >> 5;

    // ---- work items: identical walk to conv_wino_k (XCD-aware, incremental, scalar)
    struct Item { int tx, ty, b, nt; };
    Item cur, nxt, dlt;
    {
        const int GD = gridDim.x, w = blockIdx.x;
        int pix, dpix;
        if (p.xcd_slabs) {
            const int PT = (GD >> 3) / n_ntiles;
            pix = (w & 7) * PT + (w >> 3) / n_ntiles; dpix = 8 * PT;
            cur.nt = (w >> 3) % n_ntiles; dlt.nt = 0;
        } else {
            cur.nt = w % n_ntiles; pix = w / n_ntiles;
            dlt.nt = GD % n_ntiles; dpix = GD / n_ntiles;
        }
        cur.tx = pix % p.tiles_x; cur.ty = (pix / p.tiles_x) % p.tiles_y; cur.b = pix / (p.tiles_x * p.tiles_y);
        dlt.tx = dpix % p.tiles_x; dlt.ty = (dpix / p.tiles_x) % p.tiles_y; dlt.b = dpix / (p.tiles_x * p.tiles_y);
    }
    auto advance = [&](const Item& a) {
        Item r = a;
        r.nt += dlt.nt;
        int carry = 0;
        if (r.nt >= n_ntiles) { r.nt -= n_ntiles; carry = 1; }
        r.tx += dlt.tx + carry;
        if (r.tx >= p.tiles_x) { r.tx -= p.tiles_x; r.ty += 1; }
        r.ty += dlt.ty;
        if (r.ty >= p.tiles_y) { r.ty -= p.tiles_y; r.b += 1; }
        r.b += dlt.b;
        return r;
    };
    auto in_of = [&](const Item& a) {
        return p.in + (size_t)a.b * (size_t)(p.Hi + 2) * (p.Wi + 2) * cs + (size_t)(((a.ty + p.ty0) * 16) * (p.Wi + 2) + (a.tx + p.tx0) * 16) * cs +
               (size_t)a.nt * p.cin_slab_step;
    };
    auto w_of = [&](const Item& a) { return p.wpk + (size_t)a.nt * nchunks * (16 * 32 * 16) + (PERIMG ? (size_t)a.b * p.w_bstride : (size_t)0); };
    int asrc[G::RAW_IT];
#pragma unroll
    for (int it = 0; it < G::RAW_IT; ++it) {     // same LDS image of the 18x18 halo as conv_wino_k (even/odd column split)
        const int e = it * NT + tid;
        int P = e >> 2;
        const int qq = e & 3;
        if (P >= 324) P = 0;
        const int hf = P >= G::HALF, rem = P - hf * G::HALF;
        const int hy = rem / 9, hx = 2 * (rem - hy * 9) + hf;
        asrc[it] = ((hy * (p.Wi + 2) + hx) * cs + 4 * (qq ^ ((hx >> 1) & 3))) * 4;
    }
    const int raw_last_num = ((G::RAW_IT - 1) * NT + wave * 64 < G::PIECES) ? 0x7fffffff : 0;
    bool have = cur.b < p.B, have_nxt = false;
    const float* in_t = in_of(cur);
    const float* w_t = w_of(cur);
    const float* in_n = in_t;
    const float* w_n = w_t;
    auto stage_u = [&](int chunk) {
        char* udst = smem + 2 * RAW_BYTES + (chunk & 1) * U_LDS;
#pragma unroll
        for (int it = 0; it < G::U_IT; ++it) bufld16(w_t, udst + (it * NT + wave * 64) * 16, tid * 16, chunk * U_BYTES + it * NT * 16);
    };
    auto stage_raw = [&](int chunk) {
        char* rdst = smem + (chunk & 1) * RAW_BYTES;
#pragma unroll
        for (int it = 0; it < G::RAW_IT; ++it)
            if (it * NT + wave * 64 < G::PIECES) bufld16(in_t, rdst + (it * NT + wave * 64) * 16, asrc[it], chunk * 64);
    };
    // per-channel epilogue parameters of the item's cout slab: rows of 32 floats,
    // 0 bias | 1-4 n1 (mean, rstd, lo, hi) | 5-8 n2 | 9-10 style mean, std
    char* const par = smem + 2 * RAW_BYTES + 2 * U_LDS;
    char* const xch = par + WINO_PAR_BYTES;
    auto stage_params = [&](int ntile, int img) {      // img: the item's image (per-image state: p.par_bstride)
        if (wave < 2) {
            const int e = tid;
            const int row = e >> 3, col = (e & 7) * 4;
            const float* src = p.bias;
            const int pb = PERIMG ? img * p.par_bstride : 0;
            int off = ntile * 32 + col + (PERIMG ? img * p.bias_bstride : 0);
            if (row >= 1 && row <= 4) { src = (EPI & E_NORM1) ? p.n1 : p.bias; off = (EPI & E_NORM1) ? ntile * 32 + col + pb + (row - 1) * p.Cout : off; }
            if (row >= 5 && row <= 8) { src = (EPI & E_NORM2) ? p.n2 : p.bias; off = (EPI & E_NORM2) ? ntile * 32 + col + pb + (row - 5) * p.Cout : off; }
            if (row >= 9) { src = (EPI & E_NORM2) ? p.sty : p.bias; off = (EPI & E_NORM2) ? ntile * 32 + col + pb + (row - 9) * p.Cout : off; }
            if (row > 10) { src = p.bias; off = ntile * 32 + (PERIMG ? img * p.bias_bstride : 0); }
            glds16(src + off, par + wave * 1024);
        }
    };

    // LDS byte addresses.  Patch piece dx*3 + b = halo pixel (row 4 tg + 2 tr + (half ? 3-b : b), column 2 tc + dx),
    // 16-byte piece q; U fragment of local position rl*4 + k = global position (half ? (rl ? 2 : 3) : rl)*4 + k.
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    unsigned offD[12];
#pragma unroll
    for (int dx = 0; dx < 4; ++dx)
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            const int hy = 4 * tg + 2 * tr + (half ? 3 - b : b), hx = 2 * tc + dx;
            const int P = (hx & 1) * G::HALF + hy * 9 + (hx >> 1);
            offD[dx * 3 + b] = lds0 + P * 64 + ((q ^ ((hx >> 1) & 3)) << 4);
        }
    const unsigned offU = lds0 + 2 * RAW_BYTES + t * 64 + ((q ^ ((0 - (t >> 2)) & 3)) << 4);
    const unsigned ubA = offU + (half ? 12 : 0) * 2048;      // local row 0: transform row 0 / 3
    const unsigned ubB = offU + (half ? 8 : 4) * 2048;       // local row 1: transform row 1 / 2

    f32x4 acc[8][2];
    f32x4 va[8], vb[8];          // V of the current / next chunk; local position (rl, k) lives in element k*2 + rl
    auto col_pass = [&](const f32x4 (&d)[12], f32x4 (&v)[8], int dx) {
        v[dx * 2 + 0] = f4sub(d[dx * 3 + 0], d[dx * 3 + 2]);
        v[dx * 2 + 1] = f4fma(d[dx * 3 + 2], sgn, d[dx * 3 + 1]);
    };
    auto row_pass = [&](f32x4 (&v)[8], int rl) {
        const f32x4 e0 = v[0 + rl], e1 = v[2 + rl], e2 = v[4 + rl], e3 = v[6 + rl];
        v[0 + rl] = f4sub(e0, e2); v[2 + rl] = f4add(e1, e2); v[4 + rl] = f4sub(e2, e1); v[6 + rl] = f4sub(e1, e3);
    };

    auto chunk_body = [&](int c, auto par_c, auto first_c, f32x4 (&vcur)[8], f32x4 (&vnext)[8], auto&& hook) {
        constexpr int PAR = decltype(par_c)::value;
        constexpr bool FIRST = decltype(first_c)::value;    // first chunk of an item: accumulators start from zero
        // U(c+1) -> U buffer (c+1)&1, raw(c+2) -> raw buffer c&1; past the end of the item the same slots carry the
        // next item's U(0), raw(0), raw(1).  LDS-DMA instructions are spread over the first MFMA-loop iterations.
        const bool own_u = c + 1 < nchunks, own_r = c + 2 < nchunks;
        const rsrc_t rs_u = make_rsrc(own_u ? w_t : w_n);
        const rsrc_t rs_r = make_rsrc(own_r ? in_t : in_n);
        const rsrc_t rs_rl = make_rsrc(own_r ? in_t : in_n, raw_last_num);   // last raw slot: waves past the tile's end are switched off
        const int usoff = own_u ? (c + 1) * U_BYTES : 0;
        const int rsoff = (own_r ? c + 2 : c + 2 - nchunks) * 64;
        char* const udst = smem + 2 * RAW_BYTES + (1 - PAR) * U_LDS;
        char* const rdst = smem + PAR * RAW_BYTES;
        constexpr int UI = PAR * U_LDS;                    // U buffer c&1, folded into the read immediates
        constexpr int RB = (1 - PAR) * RAW_BYTES;          // raw buffer (c+1)&1
        f32x4 u[4][2];       // U fragments in flight, slot = position & 3
        f32x4 d[12];         // raw patch rows of the next chunk
        u[0][0] = lds_rd128<UI + 0>(ubA); u[0][1] = lds_rd128<UI + 1024>(ubA);
        u[1][0] = lds_rd128<UI + 2048>(ubA); u[1][1] = lds_rd128<UI + 2048 + 1024>(ubA);
        static_for([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            if constexpr (i < 4) {
                d[3 * i + 0] = lds_rd128<RB>(offD[3 * i + 0]);
                d[3 * i + 1] = lds_rd128<RB>(offD[3 * i + 1]);
                d[3 * i + 2] = lds_rd128<RB>(offD[3 * i + 2]);
            }
            if constexpr (i + 2 < 8) {
                constexpr int n = i + 2, k = n & 3;
                u[n & 3][0] = lds_rd128<UI + k * 2048>(n < 4 ? ubA : ubB);
                u[n & 3][1] = lds_rd128<UI + k * 2048 + 1024>(n < 4 ? ubA : ubB);
            }
            // column dx = i-2 of the patch was issued in iteration dx, ahead of U(i): complete with it
            if constexpr (i >= 2 && i <= 5)
                asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(d[3 * (i - 2) + 0]), "+v"(d[3 * (i - 2) + 1]), "+v"(d[3 * (i - 2) + 2]), "+v"(u[i & 3][0]), "+v"(u[i & 3][1]) : "i"(S::younger(i)));
            else lds_release2<S::younger(i)>(u[i & 3][0], u[i & 3][1]);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (!(WSPLIT_ABL & 1)) {
                {
                    if constexpr (i < G::U_IT) bufld16_rs(rs_u, udst + (i * NT + wave * 64) * 16, tid * 16, usoff + i * NT * 16);
                    if constexpr (i < G::RAW_IT) bufld16_rs(i == G::RAW_IT - 1 ? rs_rl : rs_r, rdst + (i * NT + wave * 64) * 16, asrc[i], rsoff);
                }
            }
            const f32x4 vv = vcur[(i & 3) * 2 + (i >> 2)];
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
                    acc[i][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(u[i & 3][nb][s], vv[s], (FIRST && s == 0) ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[i][nb], 0, 0, 0);
            // input transform of the next chunk, sliced under the MFMAs
            if constexpr (i >= 2 && i <= 5) col_pass(d, vnext, i - 2);
            if constexpr (i >= 6) row_pass(vnext, i - 6);
            hook(ic);
        }, std::make_integer_sequence<int, 8>{});
    };
    auto no_hook = [](auto) {};
    // Barrier behind the item's first chunk: its LDS-DMA must have landed, the residual loads issued AFTER them (in the
    // same chunk) need not — vector loads return in order, so "at most NRES outstanding" leaves exactly those in flight
    // (stores of the previous item may still be among the outstanding ones: then even fewer loads are, which is safe).
    constexpr int NRES = (EPI & E_RES_UPS) ? 2 : (EPI & E_RES) ? 4 : 0;
    auto barrier_first_chunk = [&]() {
        if constexpr ((WSPLIT_ABL & 2) != 0) return;
        if constexpr (NRES) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" :: "n"(NRES) : "memory");
        else __syncthreads();
    };
    auto barrier_chunk = [&]() { if constexpr (!(WSPLIT_ABL & 2)) __syncthreads(); };

    // ---- persistent loop over (pixel tile, cout slab) work items; only the first item has a prologue (see conv_wino_k)
    int par_ntile = -1, par_img = -1;
    if (have) {
        stage_raw(0);
        stage_u(0);
        stage_raw(1);
        stage_params(cur.nt, cur.b);
        par_ntile = cur.nt; par_img = cur.b;
        __syncthreads();
        f32x4 d[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) d[k] = *(const f32x4*)(smem + (offD[k] - lds0));
#pragma unroll
        for (int dx = 0; dx < 4; ++dx) col_pass(d, va, dx);
#pragma unroll
        for (int rl = 0; rl < 2; ++rl) row_pass(va, rl);
        // Every wave has read its raw(0) patch before ANY wave's first chunk requests raw(2) into the same LDS buffer.
        // (Without this barrier a wave delayed past the LDS-DMA latency — another stream's kernel competing for its SIMD —
        // read a patch partly overwritten by chunk 2's channels: one frame in ~10^4..10^5 slightly wrong with two or
        // more streams in flight; found and located in round 3 with tools/device_stream_stress.py / layer_hunt.py.)
        __syncthreads();
    }
    while (have) {
        const int e_y0 = (cur.ty + p.ty0) * 16, e_x0 = (cur.tx + p.tx0) * 16, e_b = cur.b, e_ntile = cur.nt;
        nxt = advance(cur);
        have_nxt = nxt.b < p.B;
        in_n = have_nxt ? in_of(nxt) : in_t;      // no next item: the last two chunks re-request this item's first tiles
        w_n = have_nxt ? w_of(nxt) : w_t;         // (valid memory, free LDS buffers, nobody reads them)
        if (par_ntile != e_ntile || (PERIMG && par_img != e_b)) {      // (never re-staged when gridDim.x is a multiple of the slab count and the state is shared)
            __syncthreads();                       // slower waves may still read the old slab's parameters
            stage_params(e_ntile, e_b);            // lands before the first K-loop barrier
            par_ntile = e_ntile; par_img = e_b;
        }
        // output geometry of this item
        const int Ho = (EPI & E_POOL) ? (p.H >> 1) : p.H, Wo = (EPI & E_POOL) ? (p.W >> 1) : p.W;
        float* out_b = p.out + (size_t)e_b * (size_t)(Ho + 2) * (Wo + 2) * p.Cout;
        const float* res_b = nullptr;
        if (EPI & (E_RES | E_RES_UPS)) res_b = p.res + (size_t)e_b * (size_t)(p.Hr + 2) * (p.Wr + 2) * p.Cout;
        const int yb = e_y0 + 4 * tg + 2 * tr, xb = e_x0 + 2 * tc;
        const int y = yb + half;                  // the output row this wave finishes (no-pool layers)
        f32x4 resv[2][2];                         // [nb][j]
        const float* rsrc[2] = {nullptr, nullptr};      // [j]: this lane's residual pixels (channel 4q of the slab's first block)
        if (EPI & (E_RES | E_RES_UPS)) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int x = xb + j;
                const int ry = (EPI & E_RES_UPS) ? (y >> 1) : y, rx = (EPI & E_RES_UPS) ? (x >> 1) : x;
                // lanes outside the image read the tensor's first pixel instead (valid memory; their outputs are never stored)
                const int pix = (y < p.H && x < p.W) ? (ry + 1) * (p.Wr + 2) + rx + 1 : 0;
                rsrc[j] = res_b + pix * p.Cout + e_ntile * 32 + 4 * q;
            }
        }
        // the residual values are requested behind the first chunk's LDS-DMA instructions (iterations 0..3): the chunk's
        // barrier does not wait for them (barrier_first_chunk), so their HBM latency has the whole K loop to hide in
        auto res_hook = [&](auto ic) {
            if constexpr (decltype(ic)::value == 4 && NRES != 0) {
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) {
                    resv[nb][0] = *(const f32x4*)(rsrc[0] + nb * 16);
                    if (EPI & E_RES_UPS) resv[nb][1] = resv[nb][0];      // both output columns of a lane lie in ONE low-resolution pixel
                    else resv[nb][1] = *(const f32x4*)(rsrc[1] + nb * 16);
                }
            }
        };
        chunk_body(0, std::integral_constant<int, 0>{}, std::true_type{}, va, vb, res_hook);
        barrier_first_chunk();    // U(c+1), raw(c+2) landed and visible; buffers of chunk c free
        chunk_body(1, std::integral_constant<int, 1>{}, std::false_type{}, vb, va, no_hook);
        barrier_chunk();
        for (int c = 2; c < nchunks; c += 2) {
            chunk_body(c, std::integral_constant<int, 0>{}, std::false_type{}, va, vb, no_hook);
            barrier_chunk();
            chunk_body(c + 1, std::integral_constant<int, 1>{}, std::false_type{}, vb, va, no_hook);
            barrier_chunk();
        }
        cur = nxt; have = have_nxt; in_t = in_n; w_t = w_n;
        if constexpr ((WSPLIT_ABL & 32) != 0) {      // microbench only: no epilogue (the accumulators stay alive)
            f32x4 sacc = acc[0][0];
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) if (i || nb) sacc += acc[i][nb];
            if (sacc[0] + sacc[1] + sacc[2] + sacc[3] == 123.456f) p.out[tid] = sacc[0];
            continue;
        }

        // ---- output transform: row sums of the wave's two rows, partner's row through LDS, fused epilogue
        f32x4 T[2][2][2];                         // [rl][j][nb]: T'[r][j] = sum_k M[r][k] A[k][j]
#pragma unroll
        for (int rl = 0; rl < 2; ++rl)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                T[rl][0][nb] = e4add(e4add(acc[rl * 4 + 0][nb], acc[rl * 4 + 1][nb]), acc[rl * 4 + 2][nb]);
                T[rl][1][nb] = f4sub(f4sub(acc[rl * 4 + 1][nb], acc[rl * 4 + 2][nb]), acc[rl * 4 + 3][nb]);
            }
        // exchange slot (j*2 + nb): 64 lanes x 16 B = 1 KB each
        if (EPI & E_POOL) {       // slot rl*2 + j: both rows of the channel block the PARTNER finishes
            char* mine = xch + wave * 4096 + lane * 16;
#pragma unroll
            for (int rl = 0; rl < 2; ++rl)
#pragma unroll
                for (int j = 0; j < 2; ++j) *(f32x4*)(mine + (rl * 2 + j) * 1024) = half ? T[rl][j][0] : T[rl][j][1];
        } else {
            char* mine = xch + wave * 4096 + lane * 16;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) *(f32x4*)(mine + (j * 2 + nb) * 1024) = T[1][j][nb];
        }
        __syncthreads();
        auto finish = [&](const f32x4 Yv, const f32x4& bias, const f32x4& m1, const f32x4& r1, const f32x4& lo1, const f32x4& hi1) {
            f32x4 v = e4add(Yv, bias);
            if (EPI & E_RELU) v = f4relu(v);
            if (EPI & E_LRELU) v = f4lrelu(v);
            if (EPI & E_NORM1) v = f4norm_clamp(v, m1, r1, lo1, hi1);
            return v;
        };
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            const int co = e_ntile * 32 + nb * 16 + 4 * q;
            const char* pl = par + (nb * 16 + 4 * q) * 4;      // this lane's 4 channels inside a 128-byte parameter row
            const f32x4 bias = *(const f32x4*)(pl);
            f32x4 m1, r1, lo1, hi1, m2, r2, lo2, hi2, smean, sstd;
            if (EPI & E_NORM1) {
                m1 = *(const f32x4*)(pl + 128); r1 = *(const f32x4*)(pl + 256);
                lo1 = *(const f32x4*)(pl + 384); hi1 = *(const f32x4*)(pl + 512);
            }
            if (EPI & E_NORM2) {
                m2 = *(const f32x4*)(pl + 640); r2 = *(const f32x4*)(pl + 768);
                lo2 = *(const f32x4*)(pl + 896); hi2 = *(const f32x4*)(pl + 1024);
                smean = *(const f32x4*)(pl + 1152); sstd = *(const f32x4*)(pl + 1280);
            }
            if (EPI & E_POOL) {
                if (nb == half) {
                    // rows r = 0,1 live in the even wave (A), rows -r3, r2 in the odd one (B):
                    // Y[0][j] = A0 + A1 + B1,  Y[1][j] = A1 - B1 + B0
                    const char* theirs = xch + (wave ^ 1) * 4096 + lane * 16;
                    f32x4 pooled;
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const f32x4 p0 = *(const f32x4*)(theirs + (0 * 2 + j) * 1024), p1 = *(const f32x4*)(theirs + (1 * 2 + j) * 1024);
                        const f32x4 A0 = half ? p0 : T[0][j][nb], A1 = half ? p1 : T[1][j][nb];
                        const f32x4 B0 = half ? T[0][j][nb] : p0, B1 = half ? T[1][j][nb] : p1;
                        const f32x4 Y0 = e4add(e4add(A0, A1), B1);
                        const f32x4 Y1 = e4add(f4sub(A1, B1), B0);
                        const f32x4 a = finish(Y0, bias, m1, r1, lo1, hi1), b = finish(Y1, bias, m1, r1, lo1, hi1);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float m = fmaxf(a[e], b[e]);
                            pooled[e] = j == 0 ? m : fmaxf(pooled[e], m);
                        }
                    }
                    const int y2 = yb >> 1, x2 = xb >> 1;
                    if constexpr ((WSPLIT_ABL & 4) != 0) { if (pooled[0] == 123.456f) out_b[co] = pooled[0]; }
                    else if (y2 < Ho && x2 < Wo) *(f32x4*)(out_b + ((y2 + 1) * (Wo + 2) + x2 + 1) * p.Cout + co) = pooled;
                }
            } else {
                const char* theirs = xch + (wave ^ 1) * 4096 + lane * 16;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int x = xb + j;
                    const f32x4 rc = *(const f32x4*)(theirs + (j * 2 + nb) * 1024);
                    const f32x4 Y = e4add(f4fma(T[1][j][nb], sgn, T[0][j][nb]), rc);
                    f32x4 o = finish(Y, bias, m1, r1, lo1, hi1);
                    if (EPI & (E_RES | E_RES_UPS)) o = e4add(o, resv[nb][j]);
                    if (EPI & E_NORM2) o = e4fma(f4norm_clamp(o, m2, r2, lo2, hi2), sstd, smean);
                    if constexpr ((WSPLIT_ABL & 4) != 0) { if (o[0] == 123.456f) out_b[co] = o[0]; }
                    else if (y < p.H && x < p.W) *(f32x4*)(out_b + ((y + 1) * (p.W + 2) + x + 1) * p.Cout + co) = o;
                }
            }
        }
    }
    // The last two chunks of the LAST item re-requested tiles into the free LDS buffers (nobody reads them): those
    // LDS-DMA transfers must land before the workgroup gives its LDS back — a workgroup of another stream's kernel can
    // start on this CU at once, and a stray LDS-DMA write then lands in ITS shared memory (found in round 3: one frame
    // in ~10^5 wrong with two or more streams in flight, tools/device_stream_stress.py / layer_hunt.py).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
