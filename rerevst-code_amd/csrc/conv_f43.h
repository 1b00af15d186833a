// conv_f43.h — Winograd F(4x4,3x3) 3x3 convolution on the fp32 matrix cores: 36 multiplies per 4x4 outputs instead of
// 144 (the F(2x2,3x3) kernel of conv_wino_split.h needs 64), for the same-resolution 3x3 layers with Cin, Cout >= 64
// (vgg19.features convs, test/style_network_global.py:271-281; ResidualBlock.conv2 :104,119-122) when a launch carries
// enough 32x32-pixel work items for its coarser rounds to win (rerevst_hip.hip: use_f43).
//
//   Y = A^T [ sum_c (G g_c G^T) .* (B^T d_c B) ] A,   6x6 patches four pixels apart, interpolation points 0, +-3/4, +-3/2, inf
//
// Exact in real arithmetic.  In fp32 its rounding error is dominated by the accumulation over the input channels in the transform
// domain; the balanced points (matrices below) cut it 2.4x against the textbook 0, +-1, +-2, inf: mean error 1.4x that of
// F(2x2,3x3) at the BASELINE sizes, worst pre-clamp error / bound over 32 inputs x 3 well-conditioned weight sets <= 0.44
// (F(2x2,3x3): <= 0.44; profiles/r05_parity_margin.txt, r05_fullsize_margin.txt).
//
// Why this form fits the machine where "36 positions x 32 couts x 16 channels" does not:
//  * one wave per SIMD (4 waves, 512 registers each): a wave owns 16 tiles (8 x 32 output pixels) x 32 output channels
//    x all 36 positions = 288 accumulator registers (AGPRs + a few VGPRs), so the input transform is amortised over
//    two 16-cout blocks: ONE packed VALU op per MFMA;
//  * 8-channel chunks: v_mfma_f32_16x16x4_f32 twice per (position, cout block); the transformed patch of a lane is
//    36 x 2 floats (its tile, channel pair 2q, 2q+1) and IS the matrix operand — it never touches LDS.  It is the A
//    operand (U the B operand), so the accumulators are D[tile][cout]: a lane group of 16 ends with the 32 consecutive
//    output channels of a pixel and the epilogue stores whole 128-byte lines (round 4; with U as the A operand every
//    lane stored its own 16 bytes and the stores cost 14-23 % of the kernel);
//  * the four waves of a workgroup share one U block (36 x 32 couts x 8 channels = 36 KB) and one 34 x 34 halo
//    (40 KB), both double buffered by buffer_load ... lds: 154 KB of LDS, one workgroup per CU;
//  * a work item = 32 x 32 output pixels x 32 output channels; persistent workgroups walk the items as one stream
//    (the last two chunks of an item request the next item's first tiles, as in conv_wino_k);
//  * 8-channel chunks mean 32-byte pieces of an NHWC pixel: the halo of a chunk is 2 312 of them, 32 cache lines per 1 KB LDS-DMA
//    request, and the L2 -> LDS path charges per line — so between two launches of this kernel the activations travel
//    channel-chunk-major (template parameter LAY below, round 6: +12 %).
//
// LDS images (conflict free in every lane group the LDS serves in one cycle):
//  raw : [halo row y 0..33][x & 3][x >> 2 (0..8)][32 B = 4 slots of one channel pair]; slot = pair ^ 2*((y>>2)&1):
//        a ds_read_b64 of patch piece (dy, dx) by the 16 tiles (2 x 8) x 2 pairs of a 32-lane group covers 256 distinct bytes
//  U   : [position 0..35][cout row 0..15][4 slots of 16 B = {block 0, block 1} x one channel pair]; slot = pair ^ (-(row>>2) & 3):
//        ONE ds_read_b128 per position brings a lane the fragments of both cout blocks (six LDS instructions per position
//        batch instead of twelve ds_read_b64 — every LDS instruction costs the MFMA stream its issue slot); the four
//        16-lane groups of a ds_read_b128 ({0-3, 12-15, 20-27}, ...) hold all 16 rows once, rows r, r+4, r+8, r+12 share 16
//        banks and land in four different slots
//
// F43_ABL (default 0; tools/f43_bench.hip builds one binary per value): microbenchmark switches — 1 no LDS-DMA after the
// first stage, 2 no K-loop barriers, 4 no stores, 8 no input transform, 16 per-phase clock64 timeline into p.dbg, 32 no
// epilogue, 64 halo requests over contiguous memory (what the scattered 32-byte pieces cost).  (The halo access pattern and the stores of a channel-chunk-major tensor were measured the same way — ABL 128 / 256 / 512 in commit f074ee2, profiles/r06_f43_layout.txt — and became the LAY template parameter below; ABL 1024 in commit 034c0ff priced conv1_1 folded into this kernel's loader: 23 % slower than the pair, profiles/r06_thin_fusion.txt.)  The library is compiled with 0: every hook is a discarded constexpr branch.
#pragma once
#include "conv_wino.h"

#ifndef F43_ABL
#define F43_ABL 0
#endif
#ifndef F43_SPAN
#define F43_SPAN 28      // a chunk's 19 LDS-DMA requests are spread over its first F43_SPAN position steps (tools/f43_bench.hip sweeps it)
#endif
// Measured-and-lost variants of this kernel (where the LDS-DMA requests go: F43_DMA 0-5, 7, 8; F43_TAIL, F43_RD2, F43_AS, F43_UMID,
// F43_U1; six transform lines in lockstep) are in the git history up to 0a3fa2e and in profiles/r05_f43_timeline.txt; what
// ships: one request every 28/19 position steps, the six transform multipliers in SGPR pairs, three lines in lockstep.

typedef float f32x2 __attribute__((ext_vector_type(2)));

struct F43Geo {
    static constexpr int NT = 256;                              // 4 waves, one per SIMD
    static constexpr int NPOS = 36;
    static constexpr int RAW_ROW_BYTES = 4 * 9 * 32;            // 1152: 4 column phases x 9 column groups x 32 B
    static constexpr int RAW_PIECES = 34 * 4 * 9 * 2;           // 16-byte pieces (2448; 2312 of them are halo pixels)
    static constexpr int RAW_IT = (RAW_PIECES + NT - 1) / NT;   // 10 LDS-DMA instructions per thread
    static constexpr int RAW_BYTES = RAW_IT * NT * 16;          // 40960
    static constexpr int U_BYTES = NPOS * 32 * 8 * 4;           // 36864 per (cout slab, chunk)
    static constexpr int U_IT = U_BYTES / 16 / NT;              // 9
    static constexpr int SMEM = 2 * RAW_BYTES + 2 * U_BYTES + WINO_PAR_BYTES;     // 157 696 B
};

// ---- packed fp32 helpers (v_pk_* through inline asm: the compiler has no packed form for constants / subtractions)
__device__ __forceinline__ f32x2 p2sub(const f32x2 a, const f32x2 b) {
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
template <int IMM>
__device__ __forceinline__ f32x2 lds_rd64(unsigned addr) {
    static_assert(IMM >= 0 && IMM < 65536, "ds_read offset is a 16-bit unsigned immediate");
    f32x2 r;
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "i"(IMM));
    return r;
}

// The 288 accumulators of a wave exceed the 256 AGPRs, and the compiler keeps all MFMA results of a kernel in ONE
// register file: the MFMAs are therefore issued by inline asm with the accumulator's file pinned per position
// (positions 0..27 in AGPRs, 28..35 in VGPRs).  Hazards the compiler no longer sees are covered by construction:
// dependent MFMAs on one accumulator are one independent MFMA apart, operands come from counted s_waitcnt or a chunk
// earlier, and the epilogue reads the accumulators behind a barrier and explicit s_nops.
template <bool AG>
__device__ __forceinline__ void mfma_acc(f32x4& acc, const float a, const float b) {
    if constexpr (AG) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
template <bool AG>
__device__ __forceinline__ void mfma_zero(f32x4& acc, const float a, const float b) {      // acc = a (x) b
    if constexpr (AG) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, 0" : "=a"(acc) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, 0" : "=v"(acc) : "v"(a), "v"(b));
}

// ---- the transform matrices.  Interpolation points 0, +-a, +-b, inf with a = 3/4, b = 3/2 instead of the textbook
// 0, +-1, +-2, inf: in fp32 the error of F(4x4,3x3) is dominated by the accumulation over the input channels in the
// transform domain, whose values are larger than the outputs by the norms of the transforms; points balanced around 1
// (A^T entries 0.42 .. 3.4 instead of 1 .. 8, B^T entries <= 2.8 instead of 5) cut that error 2.4x (tools/f43_points.py:
// 3.0e-6 against 7.1e-6 relative, 256 channels; F(2x2,3x3) 7.2e-7, direct fp32 2.2e-7).  All entries are dyadic
// rationals, so the algorithm stays exact in real arithmetic with exact fp32 constants.  With A2 = a^2, B2 = b^2,
// P = a^2 b^2, S = a^2 + b^2:
//   B^T d:  t0 = P d0 - S d2 + d4          t1, t2 = (d4 - B2 d2) +- a (d3 - B2 d1)
//           t5 = P d1 - S d3 + d5          t3, t4 = (d4 - A2 d2) +- b (d3 - A2 d1)
//   A^T m:  y0 = m0 + (m1 + m2) + (m3 + m4)            y1 = a (m1 - m2) + b (m3 - m4)
//           y2 = A2 (m1 + m2) + B2 (m3 + m4)           y3 = a^3 (m1 - m2) + b^3 (m3 - m4) + m5
//   G (pack_f43_k, in double): row j = (1, p_j, p_j^2) / prod_{l != j} (p_j - p_l) for the five finite points, (0, 0, 1) for inf.
// None of the multipliers is an inline constant of the ISA, so each comes from an SGPR pair loaded by an s_mov_b64 with
// a literal INSIDE the asm statement that uses it (op_sel_hi:[1,0,1] gives both packed halves the low dword; SALU is
// free next to the vector pipe, and no constant occupies a register across statements: the kernel has none to spare).
constexpr float F43_A = 0.75f, F43_B = 1.5f;
constexpr float F43_A2 = F43_A * F43_A, F43_B2 = F43_B * F43_B, F43_P = F43_A2 * F43_B2, F43_S = F43_A2 + F43_B2;
constexpr float F43_A3 = F43_A2 * F43_A, F43_B3 = F43_B2 * F43_B;
constexpr unsigned f43_bits(float f) { return __builtin_bit_cast(unsigned, f); }

// The 1-D input transform B^T (6 -> 6) of a line d0..d5 as twelve ops:
//   0: A = d4 - B2 d2     1: B = d3 - B2 d1     2: C = d4 - A2 d2     3: F = d3 - A2 d1
//   4: g = P d0 + d4      5: d0 = g - S d2      6: g = P d1 + d5      7: d5 = g - S d3
//   8: d1 = A + a B       9: d2 = A - a B      10: d3 = C + b F      11: d4 = C - b F
// with the six multipliers in SGPR pairs loaded ONCE per transform (no SALU -> VALU wait states inside the op stream;
// tools/pkfma_bench.hip: 5.1 instead of 5.8 clocks per packed op): k = {B2, A2, P, S, A, B}.
struct F43Tmp { f32x2 a, b, c, f, g; };
struct F43K { unsigned long long b2, a2, p, s, a, b; };
__device__ __forceinline__ F43K f43_load_k() {
    F43K k;
    asm volatile("s_mov_b64 %0, %6\n\ts_mov_b64 %1, %7\n\ts_mov_b64 %2, %8\n\ts_mov_b64 %3, %9\n\ts_mov_b64 %4, %10\n\ts_mov_b64 %5, %11"
                 : "=s"(k.b2), "=s"(k.a2), "=s"(k.p), "=s"(k.s), "=s"(k.a), "=s"(k.b)
                 : "i"(f43_bits(F43_B2)), "i"(f43_bits(F43_A2)), "i"(f43_bits(F43_P)), "i"(f43_bits(F43_S)), "i"(f43_bits(F43_A)), "i"(f43_bits(F43_B)));
    return k;
}
// THREE lines in lockstep: an op's result is needed three ops (~15 clocks) later — still beyond the ~9 clocks a dependent
// packed op waits — and the temporaries of a call are 30 registers (six lines in lockstep: 60).  LB = first line of the call.
template <bool NEG>
__device__ __forceinline__ void fmak3s(f32x2& r0, f32x2& r1, f32x2& r2, const f32x2 x0, const f32x2 x1, const f32x2 x2,
                                       const f32x2 c0, const f32x2 c1, const f32x2 c2, const unsigned long long ks) {
    if constexpr (NEG)
        asm volatile("v_pk_fma_f32 %0, %4, %3, %7 op_sel_hi:[1,0,1] neg_lo:[0,1,0] neg_hi:[0,1,0]\n\tv_pk_fma_f32 %1, %5, %3, %8 op_sel_hi:[1,0,1] neg_lo:[0,1,0] neg_hi:[0,1,0]\n\t"
                     "v_pk_fma_f32 %2, %6, %3, %9 op_sel_hi:[1,0,1] neg_lo:[0,1,0] neg_hi:[0,1,0]"
                     : "=&v"(r0), "=&v"(r1), "=&v"(r2) : "s"(ks), "v"(x0), "v"(x1), "v"(x2), "v"(c0), "v"(c1), "v"(c2));
    else
        asm volatile("v_pk_fma_f32 %0, %4, %3, %7 op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 %1, %5, %3, %8 op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 %2, %6, %3, %9 op_sel_hi:[1,0,1]"
                     : "=&v"(r0), "=&v"(r1), "=&v"(r2) : "s"(ks), "v"(x0), "v"(x1), "v"(x2), "v"(c0), "v"(c1), "v"(c2));
}
template <int LB, class LineFn>
__device__ __forceinline__ void f43_in3s(LineFn&& L, const F43K& k) {
    F43Tmp t[3];
    using N0 = std::integral_constant<int, LB>; using N1 = std::integral_constant<int, LB + 1>; using N2 = std::integral_constant<int, LB + 2>;
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>; using I4 = std::integral_constant<int, 4>; using I5 = std::integral_constant<int, 5>;
#define F43_D(J) L(N0{}, I##J{}), L(N1{}, I##J{}), L(N2{}, I##J{})
#define F43_T(M) t[0].M, t[1].M, t[2].M
    fmak3s<true>(F43_T(a), F43_D(2), F43_D(4), k.b2);
    fmak3s<true>(F43_T(b), F43_D(1), F43_D(3), k.b2);
    fmak3s<true>(F43_T(c), F43_D(2), F43_D(4), k.a2);
    fmak3s<true>(F43_T(f), F43_D(1), F43_D(3), k.a2);
    fmak3s<false>(F43_T(g), F43_D(0), F43_D(4), k.p);
    fmak3s<true>(F43_D(0), F43_D(2), F43_T(g), k.s);
    fmak3s<false>(F43_T(g), F43_D(1), F43_D(5), k.p);
    fmak3s<true>(F43_D(5), F43_D(3), F43_T(g), k.s);
    fmak3s<false>(F43_D(1), F43_T(b), F43_T(a), k.a);
    fmak3s<true>(F43_D(2), F43_T(b), F43_T(a), k.a);
    fmak3s<false>(F43_D(3), F43_T(f), F43_T(c), k.b);
    fmak3s<true>(F43_D(4), F43_T(f), F43_T(c), k.b);
#undef F43_D
#undef F43_T
}
// 1-D output transform A^T (6 -> 4) on a channel pair (see the matrices above): twelve packed ops and six constant loads in
// ONE asm statement (the compiler pads every asm boundary with wait states it cannot prove unnecessary), ordered so that
// no op needs the result of the three before it.
__device__ __forceinline__ void f43_out(const f32x2 m0, const f32x2 m1, const f32x2 m2, const f32x2 m3, const f32x2 m4, const f32x2 m5,
                                        f32x2& y0, f32x2& y1, f32x2& y2, f32x2& y3) {
    f32x2 s1, d1, s2, d2;
    unsigned long long ks;
    asm volatile(
        "v_pk_add_f32 %[s1], %[m1], %[m2]\n\t"
        "v_pk_add_f32 %[d1], %[m1], %[m2] neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %[s2], %[m3], %[m4]\n\t"
        "v_pk_add_f32 %[d2], %[m3], %[m4] neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %[y0], %[m0], %[s1]\n\t"
        "s_mov_b64 %[k], %[ka]\n\t"
        "v_pk_mul_f32 %[y1], %[d1], %[k] op_sel_hi:[1,0]\n\t"
        "s_mov_b64 %[k], %[ka2]\n\t"
        "v_pk_mul_f32 %[y2], %[s1], %[k] op_sel_hi:[1,0]\n\t"
        "s_mov_b64 %[k], %[ka3]\n\t"
        "v_pk_fma_f32 %[y3], %[d1], %[k], %[m5] op_sel_hi:[1,0,1]\n\t"
        "v_pk_add_f32 %[y0], %[y0], %[s2]\n\t"
        "s_mov_b64 %[k], %[kb]\n\t"
        "v_pk_fma_f32 %[y1], %[d2], %[k], %[y1] op_sel_hi:[1,0,1]\n\t"
        "s_mov_b64 %[k], %[kb2]\n\t"
        "v_pk_fma_f32 %[y2], %[s2], %[k], %[y2] op_sel_hi:[1,0,1]\n\t"
        "s_mov_b64 %[k], %[kb3]\n\t"
        "v_pk_fma_f32 %[y3], %[d2], %[k], %[y3] op_sel_hi:[1,0,1]"
        : [y0] "=&v"(y0), [y1] "=&v"(y1), [y2] "=&v"(y2), [y3] "=&v"(y3), [s1] "=&v"(s1), [d1] "=&v"(d1), [s2] "=&v"(s2), [d2] "=&v"(d2), [k] "=&s"(ks)
        : [m0] "v"(m0), [m1] "v"(m1), [m2] "v"(m2), [m3] "v"(m3), [m4] "v"(m4), [m5] "v"(m5),
          [ka] "i"(f43_bits(F43_A)), [ka2] "i"(f43_bits(F43_A2)), [ka3] "i"(f43_bits(F43_A3)),
          [kb] "i"(f43_bits(F43_B)), [kb2] "i"(f43_bits(F43_B2)), [kb3] "i"(f43_bits(F43_B3)));
}

// LAY: tensor layouts.  0: input and output are ring-layout NHWC.  Bit 0: the INPUT is channel-chunk-major ("P8":
// [B][Cin/8][Hi+2][Wi+2][8] — image b, chunk k is an 8-channel ring-layout image of its own, zero ring included); bit 1: the
// OUTPUT is.  Round 6: a chunk's halo out of an NHWC tensor is 2 312 separate 32-byte pieces one pixel stride apart — every
// 1 KB LDS-DMA request touches 32 cache lines — and that, not the byte count, is what the L2 -> LDS path charges for: the
// same bytes from contiguous memory run the kernel 12-19 % faster (profiles/r06_f43_layout.txt).  Out of a P8 plane a halo
// row is 1 088 contiguous bytes.  The data that lands in LDS is the same byte for byte (also past the image's right edge
// and last rows), so the results of the two layouts are bit-identical; the host picks P8 for the tensors between two
// conv_f43_k launches (rerevst_hip.hip: run_encoder) and keeps NHWC wherever another kernel reads or writes.
// A P8 plane's rows are pitched W + 8 pixels with pixel x at stored column x + 4 (P8_PITCH / P8_COL0; NHWC: W + 2, x + 1): 4 pixels
// are one 128-byte line, so tiles and their rows start on line boundaries — stores that start 32 bytes into a line run at half the
// write bandwidth (tools/store_align_bench.hip: 3.0 against 5.8 TB/s).  Columns 0 .. 3 and W + 4 .. W + 7 stay zero: the conv padding.
// Past the right edge of an image (W not a multiple of 32) a tile reads the zero columns and then the next row, as it does in NHWC with
// other values there: in-image outputs agree with the NHWC chain to the last bit where every level is a multiple of 32 wide, and to
// rounding noise of the discarded columns elsewhere.
#ifndef P8_PAD
#define P8_PAD 6              /* extra pixels of pitch: a P8 twin is a ring-layout tensor of width W + 6 */
#define P8_COL0 4             /* stored column of pixel x = 0 */
#endif
template <int EPI, int LAY = 0>
__global__ __launch_bounds__(256, 1) void conv_f43_k(const ConvP p) {
    static_assert(!(EPI & E_RES), "same-resolution residuals are not needed by the layers this kernel serves");
    constexpr bool INP8 = (LAY & 1) != 0, OUTP8 = (LAY & 2) != 0;
    const int WP = p.Wi + 2 + (INP8 ? P8_PAD : 0);      // input pitch in pixels
    using G = F43Geo;
    constexpr int ABL = F43_ABL;      // microbenchmark switches; 0 in the library
    constexpr int RAW_BYTES = G::RAW_BYTES, U_BYTES = G::U_BYTES, NT = G::NT, NPOS = G::NPOS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, t = lane & 15, q = lane >> 4;
    const int tr = t >> 3, tc = t & 7;           // the wave's 16 tiles: 2 rows x 8 columns of 4x4 outputs = 8 x 32 pixels
    const int nchunks = p.Cin >> 3;              // 8-channel chunks; even and >= 4 (Cin a multiple of 16, >= 32)
    const int n_ntiles = p.Cout >> 5;

    // ---- work items (32 x 32 output pixels x 32 couts): the same XCD-aware incremental walk as conv_wino_k
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
    const size_t img_floats = (size_t)(p.Hi + 2) * WP * p.Cin;
    auto in_of = [&](const Item& a) {      // the tile's halo origin: stored row 32 ty, stored column 32 tx (+ P8_COL0 - 1 in a P8 plane)
        return p.in + (size_t)a.b * img_floats + (size_t)(((a.ty + p.ty0) * 32) * WP + (a.tx + p.tx0) * 32 + (INP8 ? P8_COL0 - 1 : 0)) * (INP8 ? 8 : p.Cin);
    };
    // floats from one 8-channel chunk of a pixel to the next: 8 inside an NHWC pixel, one plane in P8 (added to the descriptor's
    // BASE there, so that `lim` below — the bytes to the end of the tile's own plane — bounds every chunk alike)
    const size_t plane_floats = (size_t)(p.Hi + 2) * WP * 8;
    // bytes from the item's tile origin to the end of ITS image (ring included): LDS-DMA lanes beyond get zeros, so a
    // tile that overruns the image's last rows never sees the next image (a frame's arithmetic is the same in any batch)
    auto lim_of = [&](const Item& a) {
        const long rows_left = (long)(p.Hi + 2) - (long)(a.ty + p.ty0) * 32;
        const long n = (rows_left * WP - (long)(a.tx + p.tx0) * 32 - (INP8 ? P8_COL0 - 1 : 0)) * (INP8 ? 8 : p.Cin) * 4;
        return (int)(n > 0x7fffffffL ? 0x7fffffffL : n);
    };
    auto w_of = [&](const Item& a) { return p.wpk + (size_t)a.nt * nchunks * (U_BYTES / 4); };
    int asrc[G::RAW_IT];
#pragma unroll
    for (int it = 0; it < G::RAW_IT; ++it) {
        int e = it * NT + tid;
        if (e >= G::RAW_PIECES) e = 0;
        const int half = e & 1, xd = (e >> 1) % 9, ph = ((e >> 1) / 9) & 3, y = (e >> 1) / 36;
        const int x = 4 * xd + ph, par = (y >> 2) & 1;
        asrc[it] = ((y * WP + x) * (INP8 ? 8 : p.Cin) + 4 * (half ^ par)) * 4;
        if (ABL & 64) asrc[it] = (it * NT + tid) * 16;      // microbench only (wrong data): the halo requests lane-linear over 40 contiguous KB instead of 32-byte pieces one pixel stride apart
    }
    bool have = cur.b < p.B, have_nxt = false;
    const float* in_t = in_of(cur);
    const float* w_t = w_of(cur);
    int lim_t = lim_of(cur);
    const float* in_n = in_t;
    const float* w_n = w_t;
    int lim_n = lim_t;
    auto stage_u = [&](int chunk) {
        char* udst = smem + 2 * RAW_BYTES + (chunk & 1) * U_BYTES;
#pragma unroll
        for (int it = 0; it < G::U_IT; ++it) bufld16(w_t, udst + (it * NT + wave * 64) * 16, tid * 16, chunk * U_BYTES + it * NT * 16);
    };
    auto stage_raw = [&](int chunk) {
        char* rdst = smem + (chunk & 1) * RAW_BYTES;
        const rsrc_t rs = make_rsrc(INP8 ? in_t + chunk * plane_floats : in_t, lim_t);
#pragma unroll
        for (int it = 0; it < G::RAW_IT; ++it) bufld16_rs(rs, rdst + (it * NT + wave * 64) * 16, asrc[it], INP8 ? 0 : chunk * 32);
    };
    char* const par = smem + 2 * RAW_BYTES + 2 * U_BYTES;
    // img: the item's image.  Per-image state (ConvP::par_bstride != 0, the grouped multi-style decoder): image b reads its
    // saved statistics / style affine at n1 / n2 / sty + b * par_bstride; bias and weights are shared by the layers this kernel serves.
    auto stage_params = [&](int ntile, int img) {
        if (wave < 2) {
            const int e = tid;
            const int row = e >> 3, col = (e & 7) * 4;
            const float* src = p.bias;
            int off = ntile * 32 + col;
            const int pb = img * p.par_bstride;
            if (row >= 1 && row <= 4) { src = (EPI & E_NORM1) ? p.n1 : p.bias; off += (EPI & E_NORM1) ? (row - 1) * p.Cout + pb : 0; }
            if (row >= 5 && row <= 8) { src = (EPI & E_NORM2) ? p.n2 : p.bias; off += (EPI & E_NORM2) ? (row - 5) * p.Cout + pb : 0; }
            if (row >= 9) { src = (EPI & E_NORM2) ? p.sty : p.bias; off += (EPI & E_NORM2) ? (row - 9) * p.Cout + pb : 0; }
            if (row > 10) { src = p.bias; off = ntile * 32; }
            glds16(src + off, par + wave * 1024);
        }
    };

    // ---- LDS read addresses.  Patch piece (dy, dx) of tile (tr, tc): halo row 8 wave + 4 tr + dy, column 4 tc + dx.
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const int y0 = 8 * wave + 4 * tr;
    // rows dy 0..3 have (y>>2)&1 == tr&1 (8 wave is even), rows dy 4,5 the opposite parity
    const unsigned rawA = lds0 + y0 * G::RAW_ROW_BYTES + tc * 32 + ((q ^ (2 * (tr & 1))) << 3);
    const unsigned rawB = lds0 + y0 * G::RAW_ROW_BYTES + tc * 32 + ((q ^ (2 * ((tr & 1) ^ 1))) << 3);
    const unsigned offU = lds0 + 2 * RAW_BYTES + t * 64 + ((q ^ ((0 - (t >> 2)) & 3)) << 4);      // cout row t of both blocks, channel pair q: one 16-byte slot

    const unsigned lane_off = (unsigned)(((4 * (q >> 1)) * (p.W + 2) + 16 * (q & 1)) * p.Cout + 2 * t) * 4u;      // epilogue stores, see there
    // Positions whose accumulators live in AGPRs: 28 x 2 blocks x 4 = 224 of the 256; positions 28..35 sit in VGPRs, and the
    // 32 free AGPRs take what the register allocator cannot keep in VGPRs outside the K loops (v_accvgpr moves instead of
    // scratch reloads, whose s_waitcnt vmcnt(0) would serialise the LDS-DMA stream)
#ifndef F43_NAG
#define F43_NAG 28
#endif
    constexpr int NAG = F43_NAG;
    f32x4 accA[NAG][2], accV[NPOS - NAG][2];
    int par_ntile = -1, par_img = -1;
    long long tl[6] = {0, 0, 0, 0, 0, 0}, tl_t = 0;      // ABL & 16 (microbench): cycles per phase, summed over items
    auto tick = [&](int k) { if (ABL & 16) { const long long n = clock64(); tl[k] += n - tl_t; tl_t = n; } };
    f32x2 v[NPOS];                // transformed patch B^T d B of the chunk in flight: V[r][k] at index k*6 + r (raw piece (dy, dx) at dx*6 + dy)
    f32x4 ur[2][6];               // U fragments of two position batches: [ring slot][r] = {block 0: channels 2q, 2q+1; block 1: the same}
    // the 2-D input transform of the patch in d: column passes (line dx = d[6 dx + 0..5]), then row passes (line r = d[r], d[6 + r], .., d[30 + r])
    auto full_transform = [&](f32x2 (&d)[NPOS]) {
        const F43K k = f43_load_k();
        auto col = [&](auto nc, auto jc) -> f32x2& { return d[decltype(nc)::value * 6 + decltype(jc)::value]; };
        auto row = [&](auto nc, auto jc) -> f32x2& { return d[decltype(jc)::value * 6 + decltype(nc)::value]; };
        f43_in3s<0>(col, k);
        f43_in3s<3>(col, k);
        f43_in3s<0>(row, k);
        f43_in3s<3>(row, k);
    };
    // patch piece (dy, dx) of the raw tile in raw buffer `RB` (byte offset) -> v[dx*6 + dy]
    auto read_patch_col = [&](auto rbc, auto dxc) {
        constexpr int RB = decltype(rbc)::value, dx = decltype(dxc)::value;
        static_for([&](auto dyc) {
            constexpr int dy = decltype(dyc)::value;
            constexpr int off = RB + dy * G::RAW_ROW_BYTES + (dx & 3) * 288 + (dx >> 2) * 32;
            v[dx * 6 + dy] = lds_rd64<off>(dy < 4 ? rawA : rawB);
        }, std::make_integer_sequence<int, 6>{});
    };
    // U fragments of position batch b (= transform column k = b: positions r*6 + b, r = 0..5) from the U buffer at `ub`
    auto read_u_batch = [&](unsigned ub, auto bc, auto slotc) {
        constexpr int b = decltype(bc)::value, sl = decltype(slotc)::value;
        static_for([&](auto rc) {
            constexpr int r = decltype(rc)::value;
            ur[sl][r] = lds_rd128<(r * 6 + b) * 1024>(ub);
        }, std::make_integer_sequence<int, 6>{});
    };

    // One chunk.  On gfx950 the f32 MFMA runs on the vector ALU: NOTHING overlaps it inside a wave — every interruption of
    // an MFMA run costs ~10 cycles plus ~4 per instruction (tools/mfma_filler_bench) — so with one wave per SIMD the
    // chunk is SIX uninterrupted runs of 24 MFMAs (position batch b = transform column b, both cout blocks, 8 channels)
    // and the LDS reads / LDS-DMA requests / input transform are packed into the gaps between them:
    //   gap b (before run b): wait for U batch b; request U batch b+1; patch column b-1 of the NEXT chunk into the V
    //                         registers run b-1 has just consumed (V is single buffered); the chunk's LDS-DMA (gaps 1..3)
    //   tail               : patch column 5; wait; K-loop barrier; request U batch 0 of the next chunk; the whole input
    //                         transform (144 packed ops, three independent lines in lockstep) covers that latency.
    // Every instruction of the loop is a volatile asm: the order written is the order issued.
    auto chunk_body = [&](int c, auto par_c, auto first_c, bool last) {
        constexpr int PAR = decltype(par_c)::value;
        constexpr bool FIRST = decltype(first_c)::value;
        // chunk c requests U(c+1) -> U buffer (c+1)&1 and raw(c+2) -> raw buffer c&1; past the item's end the same
        // slots carry the NEXT item's U(0), raw(0), raw(1) (nchunks is even)
        const bool own_u = c + 1 < nchunks, own_r = c + 2 < nchunks;
        const rsrc_t rs_u = make_rsrc(own_u ? w_t : w_n);
        const int rchunk = own_r ? c + 2 : c + 2 - nchunks;
        const rsrc_t rs_r = make_rsrc(INP8 ? (own_r ? in_t : in_n) + rchunk * plane_floats : (own_r ? in_t : in_n), own_r ? lim_t : lim_n);
        const int usoff = own_u ? (c + 1) * U_BYTES : 0;
        const int rsoff = INP8 ? 0 : rchunk * 32;
        char* const udst = smem + 2 * RAW_BYTES + (1 - PAR) * U_BYTES;
        char* const rdst = smem + PAR * RAW_BYTES;
        const unsigned ub = offU + PAR * U_BYTES;          // U buffer c&1
        const unsigned ubn = offU + (1 - PAR) * U_BYTES;   // U buffer (c+1)&1
        using RBt = std::integral_constant<int, (1 - PAR) * RAW_BYTES>;      // raw buffer (c+1)&1: the next chunk's patch
        static_for([&](auto bc) {
            constexpr int b = decltype(bc)::value;
            // ---- gap b: U batch b is needed now; younger LDS reads = the patch column requested in run b-1's mini gap
            if constexpr (b <= 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            else if (last) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
            if constexpr (b < 5) read_u_batch(ub, std::integral_constant<int, b + 1>{}, std::integral_constant<int, (b + 1) & 1>{});
            // ---- run b: 24 MFMAs, with ONE mini gap in the middle for the patch column (issued in the gap, behind the U reads, it
            // measured 1-2 % slower: profiles/r04_f43_store_study.txt) that also carries the chunk's LDS-DMA requests — wave w issues all of its 19 in
            // run w: the four waves of the CU share one address unit, requests issued at the same time queue behind each other
            static_for([&](auto rc) {
                constexpr int r = decltype(rc)::value, pos = r * 6 + b;
                constexpr int step = b * 6 + r;                  // issue order of the 36 position steps of a chunk
                auto dma_req = [&](auto nc) {                    // request n of the chunk: 0..9 the halo, 10..18 the U block
                    constexpr int n = decltype(nc)::value;
                    if constexpr (n < G::RAW_IT) bufld16_rs(rs_r, rdst + (n * NT + wave * 64) * 16, asrc[n], rsoff);
                    else if constexpr (n < G::RAW_IT + G::U_IT) bufld16_rs(rs_u, udst + ((n - G::RAW_IT) * NT + wave * 64) * 16, tid * 16, usoff + (n - G::RAW_IT) * NT * 16);
                };
                {      // request n at step n * 28 / 19: the chunk's 19 requests evenly over its first 28 position steps (the CU's L2 -> LDS
                       // path sustains 13-20 B/clock and the kernel needs 76 KB per ~7 000-clock chunk: a request issued into a full
                       // queue stalls the wave, and with one wave per SIMD that is lost MFMA time; profiles/r05_f43_timeline.txt)
                    constexpr int SPAN = F43_SPAN, NREQ = G::RAW_IT + G::U_IT;
                    constexpr int n = (step * NREQ + SPAN - 1) / SPAN;
                    if constexpr (n < NREQ && (n * SPAN) / NREQ == step) { if (!(ABL & 1)) dma_req(std::integral_constant<int, n>{}); }
                }
                if constexpr (r == 3) {
                    if constexpr (b >= 1) { if (!last) read_patch_col(RBt{}, std::integral_constant<int, b - 1>{}); }
                }
                constexpr bool AG = pos < NAG;
                f32x4 (&ac)[2] = *[&]() -> f32x4 (*)[2] { if constexpr (AG) return &accA[pos]; else return &accV[pos - NAG]; }();
                const f32x2 vv = v[b * 6 + r];
                const f32x4 uu = ur[b & 1][r];
                // A operand = the patch (D row = tile), B operand = U (D column = cout row): see the epilogue for why
                if constexpr (FIRST) { mfma_zero<AG>(ac[0], vv[0], uu[0]); mfma_zero<AG>(ac[1], vv[0], uu[2]); }
                else { mfma_acc<AG>(ac[0], vv[0], uu[0]); mfma_acc<AG>(ac[1], vv[0], uu[2]); }
                mfma_acc<AG>(ac[0], vv[1], uu[1]);
                mfma_acc<AG>(ac[1], vv[1], uu[3]);
            }, std::make_integer_sequence<int, 6>{});
        }, std::make_integer_sequence<int, 6>{});
        // ---- tail: the last patch column, the barrier, the next chunk's first U batch, the transform.  The last chunk of
        // an item only runs the barrier: the next item's patch is read behind the output transform (which needs the
        // registers), see the item loop.
        if (!last) read_patch_col(RBt{}, std::integral_constant<int, 5>{});
        tick(1);
        if (!(ABL & 2)) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        tick(3);
        if (!last) {
            read_u_batch(ubn, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
            if (!(ABL & 8)) full_transform(v);
        }
        tick(5);
    };

    // first tiles of an item whose predecessor did not request them (the workgroup's first item)
    auto next_patch = [&]() {        // the item's raw(0) patch + first U batch -> registers, then V(0)
        static_for([&](auto dxc) { read_patch_col(std::integral_constant<int, 0>{}, dxc); }, std::make_integer_sequence<int, 6>{});
        read_u_batch(offU, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
        // every wave has read raw(0) before any wave's chunk 0 requests raw(2) into the same buffer
        if (!(ABL & 2)) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (!(ABL & 8)) full_transform(v);
    };
    if (have) {
        stage_raw(0);
        stage_u(0);
        stage_raw(1);
        stage_params(cur.nt, cur.b);
        par_ntile = cur.nt; par_img = cur.b;
        __syncthreads();
        next_patch();
        __syncthreads();      // round-3 fix, as in the library kernels: raw(0) is read by every wave before the first chunk's LDS-DMA reuses its buffer
    }
    if (ABL & 16) tl_t = clock64();
    while (have) {
        const int e_y0 = (cur.ty + p.ty0) * 32, e_x0 = (cur.tx + p.tx0) * 32, e_b = cur.b, e_ntile = cur.nt;
        nxt = advance(cur);
        have_nxt = nxt.b < p.B;
        in_n = have_nxt ? in_of(nxt) : in_t;
        w_n = have_nxt ? w_of(nxt) : w_t;
        lim_n = have_nxt ? lim_of(nxt) : lim_t;
        if (par_ntile != e_ntile || (p.par_bstride != 0 && par_img != e_b)) {
            __syncthreads();
            stage_params(e_ntile, e_b);
            par_ntile = e_ntile; par_img = e_b;
        }
        tick(0);                                  // item setup
        chunk_body(0, std::integral_constant<int, 0>{}, std::true_type{}, false);
        chunk_body(1, std::integral_constant<int, 1>{}, std::false_type{}, false);
        for (int c = 2; c < nchunks; c += 2) {
            chunk_body(c, std::integral_constant<int, 0>{}, std::false_type{}, false);
            chunk_body(c + 1, std::integral_constant<int, 1>{}, std::false_type{}, c + 2 == nchunks);
        }
        cur = nxt; have = have_nxt; in_t = in_n; w_t = w_n; lim_t = lim_n;
        asm volatile("s_nop 15\n\ts_nop 7");       // the last MFMAs' results before any VALU / v_accvgpr_read touches them
        auto ACC = [&](int i, int nb) -> f32x4 { return i < NAG ? accA[i][nb] : accV[i - NAG][nb]; };
        if (ABL & 32) {                           // microbench only: no epilogue at all (keeps the accumulators alive)
            f32x4 s = ACC(0, 0);
#pragma unroll
            for (int i = 0; i < NPOS; ++i)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) if (i || nb) s += ACC(i, nb);
            if (s[0] + s[1] + s[2] + s[3] == 123.456f) p.out[tid] = s[0];
            if (have) next_patch();
            continue;
        }
        // ---- output transform Y = A^T M A (rows of M = acc[r*6 + k]) + fused epilogue, all in registers.
        // The MFMAs ran with the patch as the A operand and U as the B operand, so a lane holds D[tile 4q + e][cout row t]:
        // FOUR tiles (e = 0..3: row q>>1, columns 4 (q&1) + e of the wave's 2 x 8 tiles) for TWO output channels (cout row t
        // of both blocks = channels 2t, 2t+1 of the slab, pack_f43_k).  The 16 lanes of a group therefore hold the 32
        // consecutive channels of one pixel: every store instruction writes four whole 128-byte lines (one per lane group)
        // instead of 64 separate 16-byte pieces, and the half-resolution residual is read the same way.  The 2 x 2 max pool
        // (E_POOL: a tile holds four whole pooling windows) and the residual (E_RES_UPS: a tile lies over 2 x 2
        // low-resolution pixels) stay inside the lane.
        const int mr = q >> 1, mc0 = 4 * (q & 1);
        const int yb = e_y0 + 8 * wave + 4 * mr, xb0 = e_x0 + 4 * mc0;
        // Stores: wave-uniform 64-bit base (SGPRs: image, item origin, the wave's rows, pixel (i, 4e + j)) + ONE
        // loop-invariant 32-bit lane offset (the lane's first tile inside the wave's 8 x 32 pixels and its channel pair): no
        // per-store address arithmetic in vector registers (precomputed addresses would be held across the whole K loop)
        constexpr bool POOL = (EPI & E_POOL) != 0;
        const int Ho = POOL ? (p.H >> 1) : p.H, Wo = POOL ? (p.W >> 1) : p.W;
        constexpr int OP = OUTP8 ? P8_PAD : 0;      // output pitch: Wo + 2 + OP pixels
        char* const sb = OUTP8 ? (char*)(p.out + (size_t)e_b * (size_t)(Ho + 2) * (Wo + 2 + OP) * p.Cout + (size_t)e_ntile * 4 * (size_t)(Ho + 2) * (Wo + 2 + OP) * 8 +
                                         ((size_t)((POOL ? (e_y0 >> 1) + 4 * wave : e_y0 + 8 * wave) + 1) * (Wo + 2 + OP) + (POOL ? (e_x0 >> 1) : e_x0) + P8_COL0) * 8)
                               : (char*)(p.out + (size_t)e_b * (size_t)(Ho + 2) * (Wo + 2) * p.Cout +
                                 ((size_t)((POOL ? (e_y0 >> 1) + 4 * wave : e_y0 + 8 * wave) + 1) * (Wo + 2) + (POOL ? (e_x0 >> 1) : e_x0) + 1) * p.Cout + e_ntile * 32);
        int rowb = (Wo + 2) * p.Cout * 4, pixb = p.Cout * 4;
        unsigned st_off = POOL ? (unsigned)(((2 * mr) * (Wo + 2) + 2 * mc0) * p.Cout + 2 * t) * 4u : lane_off;
        if constexpr (OUTP8) {      // [C/8][Ho+2][Wo+2][8]: a lane's channel pair (2t, 2t+1 of the slab) is 8 bytes of the 32-byte piece of chunk 4 ntile + t/4; pixels 32 bytes apart
            const unsigned plane = (unsigned)(Ho + 2) * (Wo + 2 + OP) * 32u;
            rowb = (Wo + 2 + OP) * 32; pixb = 32;
            st_off = (POOL ? (unsigned)((2 * mr) * (Wo + 2 + OP) + 2 * mc0) : (unsigned)((4 * (q >> 1)) * (p.W + 2 + OP) + 16 * (q & 1))) * 32u + (unsigned)(t >> 2) * plane + (unsigned)(t & 3) * 8u;
        }
        const bool interior = e_y0 + 32 <= p.H && e_x0 + 32 <= p.W;       // wave-uniform: no per-pixel masks inside the image
        // the per-channel parameters of the lane's two channels: read from LDS once per item
        const char* const pl = par + 8 * t;
        const f32x2 bias = *(const f32x2*)(pl);
        f32x2 n1m, n1r, n1lo, n1hi, n2m, n2r, n2lo, n2hi, smean, sstd;
        if constexpr ((EPI & E_NORM1) != 0) { n1m = *(const f32x2*)(pl + 128); n1r = *(const f32x2*)(pl + 256); n1lo = *(const f32x2*)(pl + 384); n1hi = *(const f32x2*)(pl + 512); }
        if constexpr ((EPI & E_NORM2) != 0) {
            n2m = *(const f32x2*)(pl + 640); n2r = *(const f32x2*)(pl + 768); n2lo = *(const f32x2*)(pl + 896); n2hi = *(const f32x2*)(pl + 1024);
            smean = *(const f32x2*)(pl + 1152); sstd = *(const f32x2*)(pl + 1280);
        }
        // E_RES_UPS: the 2 x 2 low-resolution residual pixels under each of the four tiles, all requested before the first
        // store (the counter of outstanding vector-memory operations is in order: a load behind a store waits for the store)
        f32x2 rres[4][2][2];
        if constexpr ((EPI & E_RES_UPS) != 0) {
            const float* const res_b = p.res + (size_t)e_b * (size_t)(p.Hr + 2) * (p.Wr + 2) * p.Cout + e_ntile * 32 + 2 * t;
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b2 = 0; b2 < 2; ++b2) {
                        // pixels outside the image read the tensor's first pixel instead (valid memory; their outputs are never stored)
                        const bool in = (yb + 2 * a < p.H) && (xb0 + 4 * e + 2 * b2 < p.W);
                        const int pix = in ? (((yb >> 1) + a + 1) * (p.Wr + 2) + ((xb0 + 4 * e) >> 1) + b2 + 1) : 0;
                        rres[e][a][b2] = *(const f32x2*)(res_b + (size_t)pix * p.Cout);
                    }
        }
        float lrk = 0.2f;
        asm volatile("" : "+s"(lrk));             // opaque: a literal would be scalarised into two v_mul_f32
        const f32x2 lrk2 = {lrk, lrk};
#pragma unroll
        for (int e = 0; e < 4; ++e) {             // tile 4q + e, channels 2t, 2t+1 as one packed pair
            auto PR = [&](int i) -> f32x2 { return f32x2{ACC(i, 0)[e], ACC(i, 1)[e]}; };
            f32x2 T[6][4];      // T[r][j] = sum_k M[r][k] A^T[j][k]
#pragma unroll
            for (int r = 0; r < 6; ++r) {
                f43_out(PR(r * 6 + 0), PR(r * 6 + 1), PR(r * 6 + 2), PR(r * 6 + 3), PR(r * 6 + 4), PR(r * 6 + 5), T[r][0], T[r][1], T[r][2], T[r][3]);
                __builtin_amdgcn_sched_barrier(0);      // accumulators are copied out of the AGPRs row by row, not all up front
            }
            f32x2 pool_prev[2];      // E_POOL: the even column's row-pair maxima, waiting for the odd column
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                // The four pixels (i = 0..3) of column j go through the epilogue STAGE by stage: four independent values per
                // stage (one wave per SIMD: a pixel-by-pixel chain would wait ~9 cycles on every dependent packed op).
                f32x2 o[4];
                f43_out(T[0][j], T[1][j], T[2][j], T[3][j], T[4][j], T[5][j], o[0], o[1], o[2], o[3]);
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = o[i] + bias;
                if (EPI & E_RELU) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) o[i] = f32x2{fmaxf(o[i][0], 0.f), fmaxf(o[i][1], 0.f)};
                }
                if (EPI & E_LRELU) {             // LeakyReLU(0.2): v >= 0 ? v : 0.2 v == max(v, 0.2 v)
#pragma unroll
                    for (int i = 0; i < 4; ++i) { const f32x2 s2 = o[i] * lrk2; o[i] = f32x2{fmaxf(o[i][0], s2[0]), fmaxf(o[i][1], s2[1])}; }
                }
                auto norm_clamp = [&](const f32x2 m, const f32x2 r, const f32x2 lo, const f32x2 hi) {      // InstanceNorm.forward with saved statistics: (x - mean) * rstd, clamped
#pragma unroll
                    for (int i = 0; i < 4; ++i) o[i] = p2sub(o[i], m) * r;
#pragma unroll
                    for (int i = 0; i < 4; ++i) o[i] = f32x2{med3(o[i][0], lo[0], hi[0]), med3(o[i][1], lo[1], hi[1])};
                };
                if constexpr ((EPI & E_NORM1) != 0) norm_clamp(n1m, n1r, n1lo, n1hi);
                if constexpr ((EPI & E_RES_UPS) != 0) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) o[i] = o[i] + rres[e][i >> 1][j >> 1];
                }
                if constexpr ((EPI & E_NORM2) != 0) {
                    norm_clamp(n2m, n2r, n2lo, n2hi);
#pragma unroll
                    for (int i = 0; i < 4; ++i) o[i] = __builtin_elementwise_fma(o[i], sstd, smean);
                }
                if constexpr (POOL) {
                    f32x2 m2[2];
#pragma unroll
                    for (int a = 0; a < 2; ++a) m2[a] = f32x2{fmaxf(o[2 * a][0], o[2 * a + 1][0]), fmaxf(o[2 * a][1], o[2 * a + 1][1])};
                    if ((j & 1) == 0) { pool_prev[0] = m2[0]; pool_prev[1] = m2[1]; continue; }
#pragma unroll
                    for (int a = 0; a < 2; ++a) m2[a] = f32x2{fmaxf(m2[a][0], pool_prev[a][0]), fmaxf(m2[a][1], pool_prev[a][1])};
                    char* const dst = sb + (2 * e + (j >> 1)) * pixb + st_off;
                    if (ABL & 4) { if (m2[0][0] == 123.456f) *(float*)dst = m2[0][0] + m2[1][0]; }
                    else if (interior) {                                     // wave-uniform: the whole item lies inside the image
                        *(f32x2*)(dst) = m2[0]; *(f32x2*)(dst + rowb) = m2[1];
                    } else {
#pragma unroll
                        for (int a = 0; a < 2; ++a)
                            if ((yb >> 1) + a < Ho && ((xb0 + 4 * e) >> 1) + (j >> 1) < Wo) *(f32x2*)(dst + a * rowb) = m2[a];
                    }
                } else {
                    char* const dst = sb + (4 * e + j) * pixb + st_off;
                    if (ABL & 4) { if (o[0][0] == 123.456f) *(float*)dst = o[0][0] + o[1][0] + o[2][0] + o[3][0]; }
                    else if (interior) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) *(f32x2*)(dst + i * rowb) = o[i];
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            if (yb + i < p.H && xb0 + 4 * e + j < p.W) *(f32x2*)(dst + i * rowb) = o[i];
                    }
                }
            }
        }
        tick(4);                                  // epilogue issue
        if (have) next_patch();                   // the next item's raw(0), U(0) landed with the last chunk's barrier
        tick(5);
    }
    if ((ABL & 16) && lane == 0) {
#pragma unroll
        for (int k = 0; k < 6; ++k) p.dbg[(blockIdx.x * 4 + wave) * 6 + k] = tl[k];
    }
}

// Weight transform U = G g G^T for F(4x4,3x3), packed [Cout/32][Cin/8][position 36][cout row 16][slot 4][block 2][2 floats] with
// the 16-byte slots of a row XOR-swizzled by -(row>>2) & 3 (a lane-linear LDS-DMA copy lands as the conflict-free image):
//   G row j = (1, p_j, p_j^2) / prod_{l != j} (p_j - p_l) for the finite points 0, +a, -a, +b, -b (a = 3/4, b = 3/2), (0, 0, 1) for inf
__global__ void pack_f43_k(const float* __restrict__ w, float* __restrict__ dst, int Cout, int Cin) {
    const size_t total = (size_t)Cout * Cin * 36;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        size_t r = i;
        const int fl = (int)(r & 3); r >>= 2;             // float inside the 16-byte slot: block fl >> 1, channel fl & 1 of the pair
        const int slot = (int)(r & 3); r >>= 2;
        const int row = (int)(r & 15); r >>= 4;
        const int pos = (int)(r % 36); r /= 36;
        const int nchunks = Cin / 8;
        const int chunk = (int)(r % nchunks); r /= nchunks;
        const int n_tile = (int)r;
        const int pair = slot ^ ((0 - (row >> 2)) & 3);
        // cout row r of block nb = channel 2 r + nb of the slab: a lane's two accumulator blocks are adjacent channels
        const int co = n_tile * 32 + 2 * row + (fl >> 1), ci = chunk * 8 + 2 * pair + (fl & 1);
        const float* g = w + ((size_t)co * Cin + ci) * 9;
        const int pr = pos / 6, pc = pos % 6;
        // U = G g G^T in double, rounded once.  G row j = (1, p_j, p_j^2) / prod_{l != j} (p_j - p_l) for the finite points
        // 0, +a, -a, +b, -b (in this order: the positions of the transforms above), (0, 0, 1) for the point at infinity.
        auto G3 = [](int rw, double g0, double g1, double g2) -> double {
            const double A = (double)F43_A, B = (double)F43_B;
            const double pt[5] = {0.0, A, -A, B, -B};
            if (rw == 5) return g2;
            double N = 1.0;
            for (int l = 0; l < 5; ++l) if (l != rw) N *= pt[rw] - pt[l];
            return (g0 + pt[rw] * g1 + pt[rw] * pt[rw] * g2) / N;
        };
        double rowv[3];   // (G g)[pr][kx]
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) rowv[kx] = G3(pr, g[0 * 3 + kx], g[1 * 3 + kx], g[2 * 3 + kx]);
        dst[i] = (float)G3(pc, rowv[0], rowv[1], rowv[2]);
    }
}
