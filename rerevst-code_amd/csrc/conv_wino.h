// conv_wino.h — 3x3 convolutions in a transform domain on the fp32 matrix cores: shared helpers, geometry (WinoGeo,
// both forms) and the kernel conv_wino_k = the UPSAMPLE-FUSED form (UPS = 1, 4 waves, two workgroups per CU:
// ResidualBlock.conv1 behind the nearest-x2 upsample, test/style_network_global.py:100-103,116-118).  The F(2x2,3x3)
// layers run on conv_wino_split.h, which reuses everything here.  (The general kernel this one was specialised from —
// with the superseded 4-wave / channel-split F(2x2,3x3) forms — is in the git history up to round 3, tools/conv_wino_ab.h.)
//
// The 9-tap contraction becomes element-wise GEMMs over transform positions:
//   Y = A^T [ sum_c (G g_c G^T) .* (B^T d_c B) ] A
// UPS = 0: 2x2 outputs from a 4x4 patch, 16 multiplies instead of 36.  UPS = 1: see WinoGeo below, 9 instead of 36.
// Exact in real arithmetic; in fp32 the rounding differs from the direct form at the 1e-7 level (checked against
// the reference goldens with the stated tolerances).
//
// Mapping (one workgroup = 16x16 output pixels x 32 output channels):
//  * v_mfma_f32_16x16x4_f32 with M = 16 output channels (U = G g G^T, pre-packed, "A" operand), N = 16 tiles,
//    K = 4 input channels per step.  A wave owns 16 tiles (4 x 16 output pixels).
//  * lane (t = lane&15, q = lane>>4) reads the raw patch of tile t for input channels 4q..4q+3 of the staged
//    16-channel chunk, computes B^T d B in registers (packed fp32) and the result IS its MFMA "B" operand: the
//    transformed input never touches LDS or HBM.
//  * the accumulators of a lane are positions x 4 consecutive output channels of its own tile, so A^T M A, bias,
//    activation, saved-stat normalise, residual, AdaIN and the 2x2 max pool (the output tile IS the pooling window)
//    are in-register; stores are 16-byte.
//  * per 16-channel chunk: one barrier; raw halo + U block staged by buffer_load..lds, double buffered; the raw tile
//    is staged TWO chunks ahead so that the patch reads + transform of chunk c+1 are sliced under the MFMAs of
//    chunk c; LDS reads are issued by inline asm two positions ahead and released by counted s_waitcnt.
//  * workgroups are persistent and walk (pixel tile, cout slab) items as ONE stream: the last two chunks of an item
//    request the next item's first tiles and the last chunk body leaves its V(0) in registers (no prologue).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <utility>
#include <type_traits>
#include "conv_mfma.h"

#define WINO_PAR_BYTES 2048             /* per-channel epilogue parameters of the item's 32 couts: 11 vectors x 128 B */
// A workgroup is 16x16 output pixels x 32 output channels, run by NW = 4 or 8 waves.  NW = 4: wave w owns tile
// group w (4 x 16 output pixels) and both 16-channel blocks.  NW = 8: waves 2g and 2g+1 share tile group g and own
// one 16-channel block each (64 accumulators, two waves per SIMD that fill each other's issue gaps).
//
// UPS = 0: F(2x2,3x3) on the input itself — 16 positions, 4x4 patches two pixels apart, 18x18 halo.
// UPS = 1: nearest-x2 upsample fused with the 3x3 conv (ResidualBlock.conv1 behind Upsample,
//   test/style_network_global.py:100-103,116-118).  One LOW-resolution pixel produces 2x2 outputs; per axis
//     y[2i]   = g0 x[i-1] + (g1+g2) x[i]   = s x[i] + g0 (x[i-1] - x[i])
//     y[2i+1] = (g0+g1) x[i] + g2 x[i+1]   = s x[i] + g2 (x[i+1] - x[i]),     s = g0+g1+g2
//   i.e. 3 multiplies per 2 outputs: 9 positions per 2x2 outputs in 2-D (36 for the direct form, 16 for the
//   a parity-folded 2x2 conv), 3x3 patches one pixel apart, 10x10 low-resolution halo, all coefficients +-1.
// SC = 1 (with UPS): the ResidualBlock's 1x1 shortcut (test/style_network_global.py:105,113-114) rides along as a
// TENTH position: conv1x1(up(x)) = up(conv1x1(x)) is one more GEMM on V[0][0] = x[i][j], the centre pixel the
// upsample-fused transform already holds, with the shortcut weights in the U slot; its output is the low-resolution
// tensor that conv2's epilogue adds (E_RES_UPS).

template <int NW, int UPS, int SC = 0>
struct WinoGeo {
    static constexpr int NT = NW * 64;                       // threads
    static constexpr int NB = NW == 8 ? 1 : 2;               // 16-cout blocks per wave
    static constexpr int NP = UPS ? 9 : 16;                  // transform positions
    static constexpr int NPU = NP + SC;                      // GEMM positions = U blocks per chunk
    static constexpr int PW = UPS ? 3 : 4;                   // patch width; pieces are indexed dx*PW + dy
    static constexpr int NPIECE = PW * PW;
    static constexpr int PPI = UPS ? 3 : 2;                  // patch pieces read per MFMA-loop iteration
    static constexpr int TIN = UPS ? 8 : 16;                 // input pixels per workgroup tile edge
    static constexpr int HALO = UPS ? 10 : 18;
    static constexpr int HALF = 18 * 9;                      // UPS = 0: halo pixels in even (= odd) columns
    static constexpr int PIECES = HALO * HALO * 4;           // 16-byte pieces of one 16-channel raw halo tile
    static constexpr int RAW_IT = (PIECES + NT - 1) / NT;    // LDS-DMA instructions per thread per raw tile
    static constexpr int RAW_BYTES = RAW_IT * NT * 16;       // 24576 / 8192
    static constexpr int U_BYTES = NPU * 32 * 16 * 4;        // 32768 / 18432 / 20480
    static constexpr int U_PIECES = U_BYTES / 16;
    static constexpr int U_IT = (U_PIECES + NT - 1) / NT;
    static constexpr int U_LDS = U_IT * NT * 16;             // LDS bytes per U buffer: a disabled LDS-DMA slot still writes zeros
    static constexpr int SMEM = 2 * RAW_BYTES + 2 * U_LDS + WINO_PAR_BYTES;   // 114 KB (one workgroup per CU) / 58 KB
    static constexpr int OCC = (UPS && NW == 4) ? 2 : 1;     // workgroups per CU
    // MFMA-loop iteration in which column dx of the patch has landed (its last piece was issued three
    // iterations earlier, before U(i)), and the iterations of the row passes
    static constexpr int col_iter(int dx) { return (dx * PW + PW - 1) / PPI + 3; }      // UPS: dx + 3
    static constexpr int row_iter(int r) { return col_iter(PW - 1) + 1 + r; }
    // LDS reads issued in iteration i: U fragments of position i+2 and up to PPI patch pieces
    static constexpr int pieces_in(int i) { return i < 0 ? 0 : (NPIECE - i * PPI <= 0 ? 0 : (NPIECE - i * PPI < PPI ? NPIECE - i * PPI : PPI)); }
    static constexpr int issued(int i) { return (i + 2 < NPU ? NB : 0) + pieces_in(i); }
    // LDS reads younger than U(i) when iteration i waits for it (LDS returns in order)
    static constexpr int younger(int i) {
        return i == 0 ? NB + issued(0) : i == 1 ? issued(0) + issued(1) : pieces_in(i - 2) + issued(i - 1) + issued(i);
    }
};

// LDS reads of the hand-pipelined main loop: issued early by inline asm, released by counted
// s_waitcnt lgkmcnt(N) (LDS returns in order), so the single wave per SIMD never parks on LDS latency.
template <int IMM>
__device__ __forceinline__ f32x4 lds_rd128(unsigned addr) {     // LDS byte address = addr + IMM (16-bit immediate)
    static_assert(IMM >= 0 && IMM < 65536, "ds_read offset is a 16-bit unsigned immediate");
    f32x4 r;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "i"(IMM));
    return r;
}
template <int N>
__device__ __forceinline__ void lds_release1(f32x4& a) {   // (never pass one variable as two operands: the
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a) : "i"(N));   //  second one is copied BEFORE the wait)
}
template <int N>
__device__ __forceinline__ void lds_release2(f32x4& a, f32x4& b) {   // a, b become valid here
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "i"(N));
}
template <int N>
__device__ __forceinline__ void lds_release3(f32x4& a, f32x4& b, f32x4& c) {
    asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(a), "+v"(b), "+v"(c) : "i"(N));
}
template <int N>
__device__ __forceinline__ void lds_release4(f32x4& a, f32x4& b, f32x4& c, f32x4& d) {
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "i"(N));
}

// a - b on the packed-fp32 pipe (two v_pk_add_f32 with the neg modifier; the compiler only forms v_pk_add_f32 for
// additions and would emit four v_sub_f32)
__device__ __forceinline__ f32x4 f4sub(const f32x4 a, const f32x4 b) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 lo, hi;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(lo) : "v"(__builtin_shufflevector(a, a, 0, 1)), "v"(__builtin_shufflevector(b, b, 0, 1)));
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(hi) : "v"(__builtin_shufflevector(a, a, 2, 3)), "v"(__builtin_shufflevector(b, b, 2, 3)));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3);
}

__device__ __forceinline__ f32x4 f4add(const f32x4 a, const f32x4 b) {   // (the compiler picks four v_add_f32 here)
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 lo, hi;
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(lo) : "v"(__builtin_shufflevector(a, a, 0, 1)), "v"(__builtin_shufflevector(b, b, 0, 1)));
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(hi) : "v"(__builtin_shufflevector(a, a, 2, 3)), "v"(__builtin_shufflevector(b, b, 2, 3)));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3);
}

// ---- epilogue arithmetic on 4-channel vectors.  Written on 2-wide halves the compiler selects v_pk_add_f32 /
// v_pk_mul_f32 / v_pk_fma_f32 by itself (two elements per instruction, freely scheduled, no inline-asm hazard nops);
// it has no packed form for a subtraction or for a literal operand, so those go through f4sub / an opaque register;
// the saved-range clamp is ONE v_med3_f32 (lo <= hi by construction: med3(v, lo, hi) == min(hi, max(lo, v))).
// Same roundings as the scalar forms: (v - m) * r is still a subtraction followed by a multiplication.
typedef float f32x2_t __attribute__((ext_vector_type(2)));
#define F4_LO(v) __builtin_shufflevector(v, v, 0, 1)
#define F4_HI(v) __builtin_shufflevector(v, v, 2, 3)
__device__ __forceinline__ f32x4 e4add(const f32x4 a, const f32x4 b) {
    const f32x2_t lo = F4_LO(a) + F4_LO(b), hi = F4_HI(a) + F4_HI(b);
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3);
}
__device__ __forceinline__ f32x4 e4mul(const f32x4 a, const f32x4 b) {
    const f32x2_t lo = F4_LO(a) * F4_LO(b), hi = F4_HI(a) * F4_HI(b);
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3);
}
__device__ __forceinline__ f32x4 e4fma(const f32x4 a, const f32x4 b, const f32x4 c) {   // a * b + c, fused
    const f32x2_t lo = __builtin_elementwise_fma(F4_LO(a), F4_LO(b), F4_LO(c)), hi = __builtin_elementwise_fma(F4_HI(a), F4_HI(b), F4_HI(c));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3);
}
__device__ __forceinline__ float med3(float v, float lo, float hi) {
    float r;
    asm("v_med3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(v), "v"(lo), "v"(hi));   // (the builtin canonicalises its inputs first: +1 v_max each)
    return r;
}
__device__ __forceinline__ f32x4 f4relu(f32x4 v) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
    return v;
}
__device__ __forceinline__ f32x4 f4lrelu(const f32x4 v) {      // LeakyReLU(0.2): v >= 0 ? v : 0.2 v == max(v, 0.2 v)
    float k = 0.2f;
    asm volatile("" : "+s"(k));                                 // opaque: a literal would be scalarised into four v_mul_f32
    const f32x2_t k2 = {k, k};
    const f32x2_t lo = F4_LO(v) * k2, hi = F4_HI(v) * k2;
    const f32x4 s = __builtin_shufflevector(lo, hi, 0, 1, 2, 3);
    f32x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = fmaxf(v[e], s[e]);
    return r;
}
__device__ __forceinline__ f32x4 f4norm_clamp(const f32x4 v, const f32x4 m, const f32x4 r, const f32x4 lo, const f32x4 hi) {
    const f32x4 n = e4mul(f4sub(v, m), r);      // InstanceNorm.forward: (x - mean) * rstd, clamped to the saved range
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = med3(n[e], lo[e], hi[e]);
    return o;
}

template <class F, int... I>
__device__ __forceinline__ void static_for(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}

// PERIMG = 1: per-image epilogue parameters (ConvP::par_bstride; weights and bias are shared by the images of a launch), a separate instantiation as in conv_wino_split.h
template <int EPI, int ABL = 0, int NW = 4, int UPS = 1, int SC = 0, int PERIMG = 0>
__global__ __launch_bounds__(NW * 64, (WinoGeo<NW, UPS, SC>::OCC)) void conv_wino_k(const ConvP p) {
    static_assert(UPS == 1 && NW == 4, "library kernel: upsample-fused form, 4 waves");
    static_assert(!(EPI & E_POOL), "no pooling behind an upsample");
    using G = WinoGeo<NW, UPS, SC>;
    constexpr int NPU = G::NPU;
    constexpr int RAW_BYTES = G::RAW_BYTES, U_BYTES = G::U_BYTES, U_LDS = G::U_LDS, NT = G::NT, NB = G::NB, NP = G::NP, PW = G::PW, NPIECE = G::NPIECE;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform: LDS-DMA bases stay in SGPRs
    const int lane = tid & 63, t = lane & 15, q = lane >> 4;
    const int tr = t >> 3, tc = t & 7;
    const int tg = wave;                                // tile group: output rows 4*tg .. 4*tg+3 of the workgroup tile
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
    auto w_of = [&](const Item& a) { return p.wpk + (size_t)a.nt * nchunks * (NPU * 32 * 16); };      // (PERIMG: the images of a launch share weights and bias here — only the saved statistics are per image; conv() checks)
    int asrc[G::RAW_IT];
#pragma unroll
    for (int it = 0; it < G::RAW_IT; ++it) {
        // LDS image: row-major 10x10 low-resolution halo, piece qq holds channels 4*(qq ^ (hx&3)): conflict-free for the stride-1 reads
        const int e = it * NT + tid;
        int P = e >> 2;
        const int qq = e & 3;
        if (P >= G::HALO * G::HALO) P = 0;
        const int hy = P / 10, hx = P - hy * 10, swz = hx & 3;
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
    auto stage_params = [&](int ntile, int img) {      // img: the item's image (per-image state: p.par_bstride)
        if (wave < 2) {
            const int e = tid;                       // 16-byte piece: row e>>3, floats 4*(e&7)..
            const int row = e >> 3, col = (e & 7) * 4;
            const float* src = p.bias;
            const int pb = PERIMG ? img * p.par_bstride : 0;
            int off = ntile * 32 + col;
            if (row >= 1 && row <= 4) { src = (EPI & E_NORM1) ? p.n1 : p.bias; off = (EPI & E_NORM1) ? ntile * 32 + col + pb + (row - 1) * p.Cout : off; }
            if (row >= 5 && row <= 8) { src = (EPI & E_NORM2) ? p.n2 : p.bias; off = (EPI & E_NORM2) ? ntile * 32 + col + pb + (row - 5) * p.Cout : off; }
            if (row >= 9) { src = (EPI & E_NORM2) ? p.sty : p.bias; off = (EPI & E_NORM2) ? ntile * 32 + col + pb + (row - 9) * p.Cout : off; }
            if (row > 10) { src = p.bias; off = ntile * 32; }
            glds16(src + off, par + wave * 1024);
        }
    };

    // LDS byte addresses: this lane's 3x3 raw patch (16-byte piece q, XOR swizzle), relative to
    // the raw buffer; and its U fragment (row = pos*32 + nb*16 + t, (row>>2)&3 == (t>>2)&3)
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    unsigned offD[NPIECE];   // index dx*PW + dy
#pragma unroll
    for (int dx = 0; dx < PW; ++dx)
#pragma unroll
        for (int dy = 0; dy < PW; ++dy) {
            const int hy = 2 * tg + tr + dy, hx = tc + dx;
            offD[dx * PW + dy] = lds0 + (hy * 10 + hx) * 64 + ((q ^ (hx & 3)) << 4);
        }
    const unsigned offU = lds0 + 2 * RAW_BYTES + t * 64 + ((q ^ ((0 - (t >> 2)) & 3)) << 4);
    const unsigned offU1 = offU + U_LDS;

    f32x4 acc[NPU][NB];
    // transformed input B^T d B of the current / next chunk (ping-pong); V[r][k] lives in element k*PW + r: the
    // raw patch is read straight into the "next" array and both transform passes run in place.
    // B^T = [[0,1,0],[1,-1,0],[0,-1,1]]
    f32x4 va[NPIECE], vb[NPIECE];
    auto pass = [](f32x4& d0, f32x4& d1, f32x4& d2) {
        const f32x4 a0 = d0, a1 = d1, a2 = d2;
        d0 = a1; d1 = f4sub(a0, a1); d2 = f4sub(a2, a1);
    };
    auto col_pass = [&](f32x4 (&d)[NPIECE], int dx) {   // d[dx*PW + dy] -> (B^T d)[r][dx] at d[dx*PW + r]
        pass(d[dx * PW + 0], d[dx * PW + 1], d[dx * PW + 2]);
    };
    auto row_pass = [&](f32x4 (&d)[NPIECE], int r) {    // (B^T d)[r][.] -> V[r][k] at d[k*PW + r]
        pass(d[0 * PW + r], d[1 * PW + r], d[2 * PW + r]);
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
        u[0][1] = lds_rd128<1024>(ub);
        u[1][0] = lds_rd128<2048>(ub);
        u[1][1] = lds_rd128<2048 + 1024>(ub);
        static_for([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            if constexpr (i + 2 < NPU) {
                u[(i + 2) & 3][0] = lds_rd128<(i + 2) * 2048>(ub);
                u[(i + 2) & 3][1] = lds_rd128<(i + 2) * 2048 + 1024>(ub);
            }
            static_for([&](auto kc) {
                constexpr int pc = i * G::PPI + decltype(kc)::value;
                if constexpr (pc < NPIECE) d[pc] = lds_rd128<RB>(offD[pc]);
            }, std::make_integer_sequence<int, G::PPI>{});
            // U(i) is complete when at most younger(i) younger reads are outstanding; in-order return also
            // completes every patch piece issued before U(i): those of iterations <= i-3
            constexpr int cdx = i - 3;      // col_iter(dx) = dx + 3 -> the patch column released in this iteration
            static_assert(G::col_iter(0) == 3 && G::col_iter(PW - 1) == PW + 2, "column release schedule");
            if constexpr (cdx >= 0 && cdx < PW) {   // one s_waitcnt for the column and the U fragments
                asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(d[cdx * 3 + 0]), "+v"(d[cdx * 3 + 1]), "+v"(d[cdx * 3 + 2]), "+v"(u[i & 3][0]), "+v"(u[i & 3][1]) : "i"(G::younger(i)));
            } else {
                lds_release2<G::younger(i)>(u[i & 3][0], u[i & 3][1]);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (!(ABL & 1)) {
                // one U and one raw request per iteration from the first on (spread evenly over the chunk, as conv_f43_k paces them, measured +-0
                // with two waves per SIMD in flight: profiles/r05_wino_dma_spread.txt)
                {
                    if constexpr (i < G::U_IT) bufld16_rs(i == G::U_IT - 1 ? rs_ul : rs_u, udst + (i * NT + wave * 64) * 16, tid * 16, usoff + i * NT * 16);
                    if constexpr (i < G::RAW_IT) bufld16_rs(i == G::RAW_IT - 1 ? rs_rl : rs_r, rdst + (i * NT + wave * 64) * 16, asrc[i], rsoff);
                }
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
    int par_ntile = -1, par_img = -1;
    long long tl[6] = {0, 0, 0, 0, 0, 0}, tl_t = 0;      // ABL & 16 (microbench): cycles per phase, summed over items
    auto tick = [&](int k) { if (ABL & 16) { const long long n = clock64(); tl[k] += n - tl_t; tl_t = n; } };
    if (ABL & 16) tl_t = clock64();
    if (have) {
        stage_raw(0);
        stage_u(0);
        stage_raw(1);
        stage_params(cur.nt, cur.b);
        par_ntile = cur.nt; par_img = cur.b;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < NPIECE; ++k) va[k] = *(const f32x4*)(smem + (offD[k] - lds0));
#pragma unroll
        for (int dx = 0; dx < PW; ++dx) col_pass(va, dx);
#pragma unroll
        for (int r = 0; r < PW; ++r) row_pass(va, r);
        __syncthreads();      // raw(0) has been read by every wave before the first chunk requests raw(2) into its buffer (see conv_wino_split_k)
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
        if constexpr ((ABL & 32) != 0) {      // microbench only: no epilogue (the accumulators stay alive)
            f32x4 sacc = acc[0][0];
#pragma unroll
            for (int i = 0; i < NPU; ++i)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) if (i || nb) sacc += acc[i][nb];
            if (sacc[0] + sacc[1] + sacc[2] + sacc[3] == 123.456f) p.out[tid] = sacc[0];
            continue;
        }
        // ---- output transform + fused epilogue (all in registers)
        float* out_b = p.out + (size_t)e_b * (size_t)(p.H + 2) * (p.W + 2 + (p.out_p8 ? 6 : 0)) * p.Cout;      // (a channel-chunk-major output's rows are pitched W + 8: conv_f43.h P8_PAD)
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
                            resv[nb][i][j] = *(const f32x4*)(res_b + ((ry + 1) * (p.Wr + 2) + rx + 1) * p.Cout + e_ntile * 32 + nb * 16 + 4 * q);
                    }
        }
        if constexpr (SC) {      // shortcut output: one low-resolution pixel per tile, no bias (conv_shortcut has none)
            const int ly = yb >> 1, lx = xb >> 1;
            if (ly < p.Hi && lx < p.Wi) {
                float* sc_b = p.sc_out + (size_t)e_b * (size_t)(p.Hi + 2) * (p.Wi + 2) * p.Cout + ((size_t)(ly + 1) * (p.Wi + 2) + lx + 1) * p.Cout;
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) *(f32x4*)(sc_b + e_ntile * 32 + nb * 16 + 4 * q) = acc[NP][nb];
            }
        }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int co = e_ntile * 32 + nb * 16 + 4 * q;
            f32x4 Y[2][2];      // A^T = [[1,1,0],[1,0,1]]
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
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int y = yb + i, x = xb + j;
                    f32x4 o = e4add(Y[i][j], bias);
                    if (EPI & E_RELU) o = f4relu(o);
                    if (EPI & E_LRELU) o = f4lrelu(o);
                    if (EPI & E_NORM1) o = f4norm_clamp(o, m1, r1, lo1, hi1);
                    if (EPI & (E_RES | E_RES_UPS)) o = e4add(o, resv[nb][i][j]);
                    if (EPI & E_NORM2) o = e4fma(f4norm_clamp(o, m2, r2, lo2, hi2), sstd, smean);
                    if (y < p.H && x < p.W) {
                        if (ABL & 4) { if (o[0] == 123.456f) out_b[co] = o[0]; }
                        else if (p.out_p8) *(f32x4*)(out_b + (co >> 3) * ((p.H + 2) * (p.W + 8) * 8) + ((y + 1) * (p.W + 8) + x + 4) * 8 + (co & 7)) = o;      // chunk plane co / 8: the lane's four channels are half of a pixel's 32-byte piece
                        else *(f32x4*)(out_b + ((y + 1) * (p.W + 2) + x + 1) * p.Cout + co) = o;
                    }
                }
        }
        tick(5);                              // epilogue issue
    }
    // the LDS-DMA transfers the last item issued into the free buffers must land before the workgroup returns its LDS
    // (see conv_wino_split_k: another stream's workgroup may start on this CU at once)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if ((ABL & 16) && lane == 0) {
        long long* dbg = p.dbg;   // microbench only
#pragma unroll
        for (int k = 0; k < 6; ++k) dbg[(blockIdx.x * NW + wave) * 6 + k] = tl[k];
    }
}

// Weight transform U = G g G^T, packed as [Cout/32][Cin/16][pos][32 couts][16 floats]; the 16-byte pieces are
// XOR-swizzled by (cout>>2)&3.  ups = 0: G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]] (16 positions);
// ups = 1: G = [[1,1,1],[1,0,0],[0,0,1]] (9 positions, the upsample-fused form above).
__global__ void pack_wino_k(const float* __restrict__ w, float* __restrict__ dst, int Cout, int Cin, int ups,
                            const float* __restrict__ wsc = nullptr) {   // wsc: [Cout][Cin] 1x1 shortcut -> position 9 (ups only)
    constexpr int CH = 16;
    const int npt = ups ? 9 : 16, pw = ups ? 3 : 4;      // transform positions
    const int np = npt + (wsc ? 1 : 0);                  // U blocks per chunk
    const size_t total = (size_t)Cout * Cin * np;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        size_t r = i;
        const int cl = (int)(r % CH); r /= CH;            // stored channel slot inside the chunk
        const int j = r & 31; r >>= 5;
        const int pos = (int)(r % np); r /= np;
        const int nchunks = Cin / CH;
        const int chunk = r % nchunks; r /= nchunks;
        const int n_tile = (int)r;
        const int e = cl & 3, qs = cl >> 2;
        const int qq = qs ^ ((0 - (j >> 2)) & 3);     // XOR mask (0,3,2,1)[(j>>2)&3]: conflict-free ds_read_b128 of a 16-row fragment
        const int co = n_tile * 32 + j, ci = chunk * CH + qq * 4 + e;
        if (pos >= npt) { dst[i] = wsc[(size_t)co * Cin + ci]; continue; }
        const float* g = w + ((size_t)co * Cin + ci) * 9;
        const int pr = pos / pw, pc = pos % pw;
        auto G3 = [&](int row, float g0, float g1, float g2) {
            if (ups) return row == 0 ? g0 + g1 + g2 : (row == 1 ? g0 : g2);
            return row == 0 ? g0 : (row == 1 ? 0.5f * (g0 + g1 + g2) : (row == 2 ? 0.5f * (g0 - g1 + g2) : g2));
        };
        float rowv[3];   // (G g)[pr][kx]
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) rowv[kx] = G3(pr, g[0 * 3 + kx], g[1 * 3 + kx], g[2 * 3 + kx]);
        dst[i] = G3(pc, rowv[0], rowv[1], rowv[2]);
    }
}
