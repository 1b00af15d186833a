// tools/conv_wino_ab.h — A/B reference forms of the transform-domain convolution, for tools/conv_microbench.hip and
// tools/upw_check.hip only (NOT part of the library): the general kernel the library's conv_wino_k was specialised from.
// UPS = 0: F(2x2,3x3) with 4 waves (one per SIMD, both 16-cout blocks) or 8 waves split by channel block — the
// forms the row-split kernel (csrc/conv_wino_split.h) superseded; UPS = 1: the upsample-fused form (same as the library).
// Geometry, LDS layouts and helpers are the library's (csrc/conv_wino.h).
#pragma once
#include "../rerevst-code_amd/csrc/conv_wino.h"

template <int EPI, int ABL = 0, int NW = 4, int UPS = 0, int SC = 0>
__global__ __launch_bounds__(NW * 64, (WinoGeo<NW, UPS, SC>::OCC)) void conv_wino_ab_k(const ConvP p) {
    static_assert(!SC || UPS, "the shortcut rides on the upsample-fused form");
    using G = WinoGeo<NW, UPS, SC>;
    constexpr int NPU = G::NPU;
    constexpr int RAW_BYTES = G::RAW_BYTES, U_BYTES = G::U_BYTES, U_LDS = G::U_LDS, NT = G::NT, NB = G::NB, NP = G::NP, PW = G::PW, NPIECE = G::NPIECE;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform: LDS-DMA bases stay in SGPRs
    const int lane = tid & 63, t = lane & 15, q = lane >> 4;
    const int tr = t >> 3, tc = t & 7;
    const int tg = NW == 8 ? wave >> 1 : wave;          // tile group: output rows 4*tg .. 4*tg+3 of the workgroup tile
    const int nb0 = NW == 8 ? wave & 1 : 0;             // first 16-cout block of this wave
    const int nchunks = p.Cin >> 4;      // even (Cin >= 64)
    const int n_ntiles = p.Cout >> 5;

    // ---- work items.  Workgroup w runs on XCD w % 8 (observed dispatch order; used for locality only).
    // p.xcd_slabs != 0: the S cout slabs of one pixel tile are given to workgroups of the SAME XCD in the same
    // round, so the raw input tile is fetched from HBM once and then hits that XCD's L2 (otherwise every XCD owns
    // one slab — its U stays L2-resident — and re-fetches every input tile).  The walk is incremental: all the
    // divisions happen once, a round advances (tx, ty, b, slab) with carries on scalars.
    struct Item { int tx, ty, b, nt; };
    Item cur, nxt, dlt;
    {
        const int GD = gridDim.x, w = blockIdx.x;
        int pix, dpix;
        if (p.xcd_slabs) {
            const int PT = (GD >> 3) / n_ntiles;            // pixel tiles per XCD per round
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
    // scalar bases of an item: its input tile origin (the per-thread halo offsets asrc[] are tile-relative and
    // never change) and its U slab
    auto in_of = [&](const Item& a) {
        return p.in + (size_t)a.b * (size_t)(p.Hi + 2) * (p.Wi + 2) * p.Cin + (size_t)(((a.ty + p.ty0) * G::TIN) * (p.Wi + 2) + (a.tx + p.tx0) * G::TIN) * p.Cin;
    };
    auto w_of = [&](const Item& a) { return p.wpk + (size_t)a.nt * nchunks * (NPU * 32 * 16); };
    int asrc[G::RAW_IT];
#pragma unroll
    for (int it = 0; it < G::RAW_IT; ++it) {
        // UPS = 0, LDS pixel slot P: even halo columns first, then odd ones (HALF slots each), row-major inside;
        // stored piece qq holds channels 4*(qq ^ ((hx>>1)&3)).. : conflict-free for the stride-2 patch reads.
        // UPS = 1: row-major 10x10, piece qq holds channels 4*(qq ^ (hx&3)): conflict-free for the stride-1 reads
        const int e = it * NT + tid;
        int P = e >> 2;
        const int qq = e & 3;
        if (P >= G::HALO * G::HALO) P = 0;
        int hy, hx, swz;
        if (UPS) { hy = P / 10; hx = P - hy * 10; swz = hx & 3; }
        else { const int half = P >= G::HALF, rem = P - half * G::HALF; hy = rem / 9; hx = 2 * (rem - hy * 9) + half; swz = (hx >> 1) & 3; }
        asrc[it] = ((hy * (p.Wi + 2) + hx) * p.Cin + 4 * (qq ^ swz)) * 4;
    }
    const int raw_last_num = ((G::RAW_IT - 1) * NT + wave * 64 < G::PIECES) ? 0x7fffffff : 0;
    const int u_last_num = ((G::U_IT - 1) * NT + wave * 64 < G::U_PIECES) ? 0x7fffffff : 0;
    bool have = cur.b < p.B, have_nxt = false;
    const float* in_t = in_of(cur);
    const float* w_t = w_of(cur);
    const float* in_n = in_t;
    const float* w_n = w_t;
    auto stage_u = [&](int chunk) {
        char* udst = smem + 2 * RAW_BYTES + (chunk & 1) * U_LDS;
#pragma unroll
        for (int it = 0; it < G::U_IT; ++it)
            if (it * NT + wave * 64 < G::U_PIECES) bufld16(w_t, udst + (it * NT + wave * 64) * 16, tid * 16, chunk * U_BYTES + it * NT * 16);
    };
    auto stage_raw = [&](int chunk) {
        char* rdst = smem + (chunk & 1) * RAW_BYTES;
#pragma unroll
        for (int it = 0; it < G::RAW_IT; ++it)
            if (it * NT + wave * 64 < G::PIECES) bufld16(in_t, rdst + (it * NT + wave * 64) * 16, asrc[it], chunk * 64);
    };

    // per-channel epilogue parameters of the item's cout slab, parked in LDS while the K loop runs:
    // rows of 32 floats: 0 bias | 1-4 n1 (mean, rstd, lo, hi) | 5-8 n2 | 9-10 style mean, std
    char* const par = smem + 2 * RAW_BYTES + 2 * U_LDS;
    auto stage_params = [&](int ntile) {
        if (wave < 2) {
            const int e = tid;                       // 16-byte piece: row e>>3, floats 4*(e&7)..
            const int row = e >> 3, col = (e & 7) * 4;
            const float* src = p.bias;
            int off = ntile * 32 + col;
            if (row >= 1 && row <= 4) { src = (EPI & E_NORM1) ? p.n1 : p.bias; off += (EPI & E_NORM1) ? (row - 1) * p.Cout : 0; }
            if (row >= 5 && row <= 8) { src = (EPI & E_NORM2) ? p.n2 : p.bias; off += (EPI & E_NORM2) ? (row - 5) * p.Cout : 0; }
            if (row >= 9) { src = (EPI & E_NORM2) ? p.sty : p.bias; off += (EPI & E_NORM2) ? (row - 9) * p.Cout : 0; }
            if (row > 10) { src = p.bias; off = ntile * 32; }
            glds16(src + off, par + wave * 1024);
        }
    };

    // LDS byte addresses: this lane's 4x4 raw patch (slot P, 16-byte piece q, XOR swizzle), relative to
    // the raw buffer; and its U fragment (row = pos*32 + nb*16 + t, (row>>2)&3 == (t>>2)&3)
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    unsigned offD[NPIECE];   // index dx*PW + dy
#pragma unroll
    for (int dx = 0; dx < PW; ++dx)
#pragma unroll
        for (int dy = 0; dy < PW; ++dy) {
            if (UPS) {
                const int hy = 2 * tg + tr + dy, hx = tc + dx;
                offD[dx * PW + dy] = lds0 + (hy * 10 + hx) * 64 + ((q ^ (hx & 3)) << 4);
            } else {
                const int hy = 4 * tg + 2 * tr + dy, hx = 2 * tc + dx;
                const int P = (hx & 1) * G::HALF + hy * 9 + (hx >> 1);
                offD[dx * PW + dy] = lds0 + P * 64 + ((q ^ ((hx >> 1) & 3)) << 4);
            }
        }
    const unsigned offU = lds0 + 2 * RAW_BYTES + nb0 * 1024 + t * 64 + ((q ^ ((0 - (t >> 2)) & 3)) << 4);
    const unsigned offU1 = offU + U_LDS;

    f32x4 acc[NPU][NB];
    // transformed input B^T d B of the current / next chunk (ping-pong); V[r][k] lives in element k*PW + r: the
    // raw patch is read straight into the "next" array and both transform passes run in place.
    // UPS = 0: B^T = [[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]];  UPS = 1: B^T = [[0,1,0],[1,-1,0],[0,-1,1]]
    f32x4 va[NPIECE], vb[NPIECE];
    auto pass = [](f32x4& d0, f32x4& d1, f32x4& d2, f32x4& d3) {
        const f32x4 a0 = d0, a1 = d1, a2 = d2, a3 = d3;
        if (UPS) { d0 = a1; d1 = f4sub(a0, a1); d2 = f4sub(a2, a1); }
        else { d0 = f4sub(a0, a2); d1 = f4add(a1, a2); d2 = f4sub(a2, a1); d3 = f4sub(a1, a3); }
    };
    auto col_pass = [&](f32x4 (&d)[NPIECE], int dx) {   // d[dx*PW + dy] -> (B^T d)[r][dx] at d[dx*PW + r]
        pass(d[dx * PW + 0], d[dx * PW + 1], d[dx * PW + 2], d[dx * PW + PW - 1]);
    };
    auto row_pass = [&](f32x4 (&d)[NPIECE], int r) {    // (B^T d)[r][.] -> V[r][k] at d[k*PW + r]
        pass(d[0 * PW + r], d[1 * PW + r], d[2 * PW + r], d[(PW - 1) * PW + r]);
    };

    // One chunk: MFMAs of chunk c with V(c) = vcur, while the raw patch of chunk c+1 is read and
    // transformed into vnext.  Issue order per iteration i: U(i+2) x2, then (i<8) patch pieces 2i, 2i+1.
    // c is even in the first body of the unrolled chunk loop and odd in the second (PAR = c & 1), so the
    // buffer selection folds into the 16-bit immediate of every ds_read: no address arithmetic in the loop
    auto chunk_body = [&](int c, auto par_c, auto first_c, f32x4 (&vcur)[NPIECE], f32x4 (&vnext)[NPIECE]) {
        constexpr int PAR = decltype(par_c)::value;
        constexpr bool FIRST = decltype(first_c)::value;    // first chunk of an item: accumulators start from zero
        // U(c+1) goes to U buffer (c+1)&1 (last read in chunk c-1), raw(c+2) to raw buffer c&1 (its patch was read
        // in chunk c-1).  Past the end of the item the same slots carry the NEXT item's U(0), raw(0), raw(1) (nchunks
        // is even, so the buffer parities line up): the K loops of consecutive items form one stream.  The LDS-DMA
        // instructions are spread over the first iterations of the MFMA loop: the four waves share one address
        // unit (~16 clk per 1 KB instruction); issued back to back they stall there.
        const bool own_u = c + 1 < nchunks, own_r = c + 2 < nchunks;
        // one descriptor per stream and chunk; in the last slot of a stream the waves past the tile's end are switched off
        const rsrc_t rs_u = make_rsrc(own_u ? w_t : w_n);
        const rsrc_t rs_ul = make_rsrc(own_u ? w_t : w_n, u_last_num);
        const rsrc_t rs_r = make_rsrc(own_r ? in_t : in_n);
        const rsrc_t rs_rl = make_rsrc(own_r ? in_t : in_n, raw_last_num);
        const int usoff = own_u ? (c + 1) * U_BYTES : 0;
        const int rsoff = (own_r ? c + 2 : c + 2 - nchunks) * 64;
        char* const udst = smem + 2 * RAW_BYTES + (1 - PAR) * U_LDS;
        char* const rdst = smem + PAR * RAW_BYTES;
        if (ABL & 128) {   // microbench only: the MFMA stream alone (no LDS reads, no transform)
            f32x4 u01[2] = {vcur[0], vcur[1]};
#pragma unroll
            for (int i = 0; i < NPU; ++i) {
                const f32x4 vv = vcur[i < NP ? i : 0];
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb)
                        acc[i][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(u01[nb][s], vv[s], (FIRST && s == 0) ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[i][nb], 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < NPIECE; ++i) vnext[i] = vcur[i];
            return;
        }
        const unsigned ub = PAR ? offU1 : offU;
        constexpr int RB = (1 - PAR) * RAW_BYTES;    // raw buffer (c+1)&1
        f32x4 u[4][NB];      // U fragments in flight, slot = pos & 3
        f32x4 (&d)[NPIECE] = vnext;   // raw patch of the next chunk, index dx*PW + dy; transformed in place
        u[0][0] = lds_rd128<0>(ub);                      // issue order = completion order: U(0) blocks, then U(1)
        if constexpr (NB == 2) u[0][NB - 1] = lds_rd128<1024>(ub);
        u[1][0] = lds_rd128<2048>(ub);
        if constexpr (NB == 2) u[1][NB - 1] = lds_rd128<2048 + 1024>(ub);
        static_for([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            if constexpr (i + 2 < NPU) {
                u[(i + 2) & 3][0] = lds_rd128<(i + 2) * 2048>(ub);
                if constexpr (NB == 2) u[(i + 2) & 3][NB - 1] = lds_rd128<(i + 2) * 2048 + 1024>(ub);
            }
            static_for([&](auto kc) {
                constexpr int pc = i * G::PPI + decltype(kc)::value;
                if constexpr (pc < NPIECE) d[pc] = lds_rd128<RB>(offD[pc]);
            }, std::make_integer_sequence<int, G::PPI>{});
            // U(i) is complete when at most younger(i) younger reads are outstanding; in-order return also
            // completes every patch piece issued before U(i): those of iterations <= i-3
            constexpr int cdx = i - 3;      // UPS: col_iter(dx) = dx + 3 -> the patch column released in this iteration
            if constexpr (UPS && NB == 2 && cdx >= 0 && cdx < PW) {   // one s_waitcnt for the column and the U fragments
                asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(d[cdx * 3 + 0]), "+v"(d[cdx * 3 + 1]), "+v"(d[cdx * 3 + 2]), "+v"(u[i & 3][0]), "+v"(u[i & 3][1]) : "i"(G::younger(i)));
            } else {
                static_for([&](auto xc) {
                    constexpr int dx = decltype(xc)::value;
                    if constexpr (G::col_iter(dx) == i) {
                        if constexpr (PW == 4) lds_release4<G::younger(i)>(d[dx * 4 + 0], d[dx * 4 + 1], d[dx * 4 + 2], d[dx * 4 + 3]);
                        else lds_release3<G::younger(i)>(d[dx * 3 + 0], d[dx * 3 + 1], d[dx * 3 + 2]);
                    }
                }, std::make_integer_sequence<int, PW>{});
                if constexpr (NB == 2) lds_release2<G::younger(i)>(u[i & 3][0], u[i & 3][1]);
                else lds_release1<G::younger(i)>(u[i & 3][0]);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (!(ABL & 1)) {
                if constexpr (i < G::U_IT) bufld16_rs(i == G::U_IT - 1 ? rs_ul : rs_u, udst + (i * NT + wave * 64) * 16, tid * 16, usoff + i * NT * 16);
                if constexpr (i < G::RAW_IT) bufld16_rs(i == G::RAW_IT - 1 ? rs_rl : rs_r, rdst + (i * NT + wave * 64) * 16, asrc[i], rsoff);
            }
            const f32x4 vv = vcur[i < NP ? (i % PW) * PW + i / PW : 0];       // position NP (shortcut): V[0][0], the centre pixel
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
                    acc[i][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(u[i & 3][nb][s], vv[s], (FIRST && s == 0) ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[i][nb], 0, 0, 0);
            // input transform of the next chunk, sliced under the MFMAs
            static_for([&](auto xc) {
                constexpr int k = decltype(xc)::value;
                if constexpr (G::col_iter(k) == i) col_pass(d, k);
                if constexpr (G::row_iter(k) == i) row_pass(d, k);
            }, std::make_integer_sequence<int, PW>{});
        }, std::make_integer_sequence<int, NPU>{});
    };

    // ---- persistent loop over (pixel tile, cout slab) work items.  Only the first item has a prologue: the last
    // two chunks of every item request the next item's U(0), raw(0), raw(1), and the last chunk body, which reads
    // and transforms "the next chunk's" patch, thereby leaves V(0) of the next item in va.
    int par_ntile = -1;
    long long tl[6] = {0, 0, 0, 0, 0, 0}, tl_t = 0;      // ABL & 16 (microbench): cycles per phase, summed over items
    auto tick = [&](int k) { if (ABL & 16) { const long long n = clock64(); tl[k] += n - tl_t; tl_t = n; } };
    if (ABL & 16) tl_t = clock64();
    if (have) {
        stage_raw(0);
        stage_u(0);
        stage_raw(1);
        stage_params(cur.nt);
        par_ntile = cur.nt;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < NPIECE; ++k) va[k] = *(const f32x4*)(smem + (offD[k] - lds0));
#pragma unroll
        for (int dx = 0; dx < PW; ++dx) col_pass(va, dx);
#pragma unroll
        for (int r = 0; r < PW; ++r) row_pass(va, r);
        __syncthreads();      // round-3 fix, as in the library kernels: raw(0) is read by every wave before the first chunk's LDS-DMA reuses its buffer
    }
    while (have) {
        const int e_y0 = (cur.ty + p.ty0) * 16, e_x0 = (cur.tx + p.tx0) * 16, e_b = cur.b, e_ntile = cur.nt;
        nxt = advance(cur);
        have_nxt = nxt.b < p.B;
        in_n = have_nxt ? in_of(nxt) : in_t;      // no next item: the last two chunks re-request this item's first tiles
        w_n = have_nxt ? w_of(nxt) : w_t;         // (valid memory, free LDS buffers, nobody reads them)
        if (par_ntile != e_ntile) {            // (never re-staged when gridDim.x is a multiple of the slab count)
            __syncthreads();                       // slower waves may still read the old slab's parameters
            stage_params(e_ntile);                 // lands before the first K-loop barrier
            par_ntile = e_ntile;
        }
        tick(0);                              // zero acc (+ previous epilogue tail)
        chunk_body(0, std::integral_constant<int, 0>{}, std::true_type{}, va, vb);
        if (!(ABL & 2)) __syncthreads();          // U(c+1), raw(c+2) landed and visible; buffers of chunk c free
        chunk_body(1, std::integral_constant<int, 1>{}, std::false_type{}, vb, va);
        if (!(ABL & 2)) __syncthreads();
        for (int c = 2; c < nchunks; c += 2) {
            chunk_body(c, std::integral_constant<int, 0>{}, std::false_type{}, va, vb);
            if (!(ABL & 2)) __syncthreads();
            chunk_body(c + 1, std::integral_constant<int, 1>{}, std::false_type{}, vb, va);
            if (!(ABL & 2)) __syncthreads();
        }
        tick(3);                              // K loop
        // the next item's first tiles were requested by the last two chunks
        cur = nxt; have = have_nxt; in_t = in_n; w_t = w_n;
        // ---- output transform + fused epilogue (all in registers)
        const int Ho = (EPI & E_POOL) ? (p.H >> 1) : p.H, Wo = (EPI & E_POOL) ? (p.W >> 1) : p.W;
        float* out_b = p.out + (size_t)e_b * (size_t)(Ho + 2) * (Wo + 2) * p.Cout;
        const float* res_b = nullptr;
        if (EPI & (E_RES | E_RES_UPS)) res_b = p.res + (size_t)e_b * (size_t)(p.Hr + 2) * (p.Wr + 2) * p.Cout;
        const int yb = e_y0 + 4 * tg + 2 * tr, xb = e_x0 + 2 * tc;
        f32x4 resv[NB][2][2];     // residual values requested before the output transform hides their latency
        if (EPI & (E_RES | E_RES_UPS)) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int y = yb + i, x = xb + j;
                        const int ry = (EPI & E_RES_UPS) ? (y >> 1) : y, rx = (EPI & E_RES_UPS) ? (x >> 1) : x;
                        resv[nb][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                        if (y < p.H && x < p.W)
                            resv[nb][i][j] = *(const f32x4*)(res_b + ((ry + 1) * (p.Wr + 2) + rx + 1) * p.Cout + e_ntile * 32 + (nb0 + nb) * 16 + 4 * q);
                    }
        }
        if constexpr (SC) {      // shortcut output: one low-resolution pixel per tile, no bias (conv_shortcut has none)
            const int ly = yb >> 1, lx = xb >> 1;
            if (ly < p.Hi && lx < p.Wi) {
                float* sc_b = p.sc_out + (size_t)e_b * (size_t)(p.Hi + 2) * (p.Wi + 2) * p.Cout + ((size_t)(ly + 1) * (p.Wi + 2) + lx + 1) * p.Cout;
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) *(f32x4*)(sc_b + e_ntile * 32 + (nb0 + nb) * 16 + 4 * q) = acc[NP][nb];
            }
        }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int co = e_ntile * 32 + (nb0 + nb) * 16 + 4 * q;
            f32x4 Y[2][2];
            if constexpr (UPS) {   // A^T = [[1,1,0],[1,0,1]]
                f32x4 T[2][3];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    T[0][c] = acc[0 + c][nb] + acc[3 + c][nb];
                    T[1][c] = acc[0 + c][nb] + acc[6 + c][nb];
                }
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    Y[i][0] = T[i][0] + T[i][1];
                    Y[i][1] = T[i][0] + T[i][2];
                }
            } else {               // A^T = [[1,1,1,0],[0,1,-1,-1]]
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
            const char* pl = par + ((nb0 + nb) * 16 + 4 * q) * 4;      // this lane's 4 channels inside a 128-byte parameter row
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
            f32x4 pooled;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int y = yb + i, x = xb + j;
                    const bool valid = (y < p.H) && (x < p.W);
                    f32x4 o = e4add(Y[i][j], bias);
                    if (EPI & E_RELU) o = f4relu(o);
                    if (EPI & E_LRELU) o = f4lrelu(o);
                    if (EPI & E_NORM1) o = f4norm_clamp(o, m1, r1, lo1, hi1);
                    if (EPI & (E_RES | E_RES_UPS)) o = e4add(o, resv[nb][i][j]);
                    if (EPI & E_NORM2) o = e4fma(f4norm_clamp(o, m2, r2, lo2, hi2), sstd, smean);
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
        tick(5);                              // epilogue issue
    }
    if ((ABL & 16) && lane == 0) {
        long long* dbg = p.dbg;   // microbench only
#pragma unroll
        for (int k = 0; k < 6; ++k) dbg[(blockIdx.x * NW + wave) * 6 + k] = tl[k];
    }
}

