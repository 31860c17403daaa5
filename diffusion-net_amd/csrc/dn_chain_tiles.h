// dn_chain_tiles.h -- device-side building blocks shared by the chained row kernels (dn_chain.hip: block forward, dn_chain_bwd.hip: block
// backward): the transposed 16x16x32 f16 MFMA step, the two-term fp16 split of eight values into operand fragments, the LDS-DMA request of
// the weight-piece ring and the piece-times-fragment product macros.  See dn_chain.hip for the decomposition.
#pragma once
#include "dn_common.h"

typedef float dn_f32x4 __attribute__((ext_vector_type(4)));

// One 16x16x32 f16 MFMA step (fp32 accumulate): lane l supplies A[i = l & 15][k = 8 (l >> 4) + j] and B[k = 8 (l >> 4) + j][col = l & 15],
// j < 8, packed in a uint4; accumulator register r of lane l is D[4 (l >> 4) + r][l & 15].
__device__ __forceinline__ dn_f32x4 dn_mfma16_f16(uint4 a, uint4 b, dn_f32x4 c) {
#ifdef DN_EMULATE
    return dnemu_mfma_f32_16x16x32_f16(a, b, c);
#else
    typedef _Float16 dn_f16x8 __attribute__((ext_vector_type(8)));
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(dn_f16x8, a), __builtin_bit_cast(dn_f16x8, b), c, 0, 0, 0);
#endif
}

// eight values (slots 0..3 = a, 4..7 = b) -> one fragment per plane
__device__ __forceinline__ void ch_split8(const float (&a)[4], const float (&b)[4], float s, uint4& hi, uint4& lo) {
    dn_split2_pair(a[0], a[1], s, hi.x, lo.x);
    dn_split2_pair(a[2], a[3], s, hi.y, lo.y);
    dn_split2_pair(b[0], b[1], s, hi.z, lo.z);
    dn_split2_pair(b[2], b[3], s, hi.w, lo.w);
}
__device__ __forceinline__ void ch_split8(const float4& a, const float4& b, float s, uint4& hi, uint4& lo) {
    dn_split2_pair(a.x, a.y, s, hi.x, lo.x);
    dn_split2_pair(a.z, a.w, s, hi.y, lo.y);
    dn_split2_pair(b.x, b.y, s, hi.z, lo.z);
    dn_split2_pair(b.z, b.w, s, hi.w, lo.w);
}
// tanh for the gradient-feature epilogue: (1 - t) / (1 + t), t = exp(-2 |x|), through the hardware exp2 / rcp with one Newton step on the
// quotient; below |x| = 0.25, where 1 - t cancels, the odd Taylor polynomial to x^9.  Measured against double: <= 1.3e-7 absolute
// (|tanh| <= 1: about one ulp of the result's range); libm's tanhf costs ~44 instructions and several branches per element.
__device__ __forceinline__ float ch_tanh(float x) {
    const float ax = fabsf(x);
    const float x2 = x * x;
    const float p = x * (1.f + x2 * (-0.33333334f + x2 * (0.13333334f + x2 * (-0.053968254f + x2 * 0.021869488f))));
#ifdef DN_EMULATE
    const float t = exp2f(-2.885390082f * ax);
    const float r = 1.f / (1.f + t);
#else
    const float t = __builtin_amdgcn_exp2f(-2.885390082f * ax);
    float r = __builtin_amdgcn_rcpf(1.f + t);
    r = r * (2.f - (1.f + t) * r);
#endif
    const float big = copysignf((1.f - t) * r, x);
    return ax < 0.25f ? p : big;
}
// a b + c d with the roundings spelled out: one product rounded, the other fused into the sum.  Left to the compiler's contraction the choice of
// WHICH product is fused follows the surrounding code (a restructure of the kernel flipped it and moved the tanh features by one ulp in a
// quarter of the elements -- 4.6e-6 -> 6.3e-6 from fp64 on the trained-checkpoint golden after four blocks); written out, it stays put.
#ifndef DN_CH_DOT_ORDER
#define DN_CH_DOT_ORDER 0
#endif
__device__ __forceinline__ float ch_dot2(float a, float b, float c, float d) {
#ifdef DN_EMULATE
    return a * b + c * d;
#else
    return DN_CH_DOT_ORDER ? __builtin_fmaf(c, d, __fmul_rn(a, b)) : __builtin_fmaf(a, b, __fmul_rn(c, d));
#endif
}
// 1 / s for a power of two s = 2^k, -126 <= k <= 126 (what dn_pow2_scale returns but for its two clamped extremes): exponent arithmetic, exact
__device__ __forceinline__ float ch_pow2_inv(float s) {
    const unsigned e = (__float_as_uint(s) >> 23) & 0xffu;
    return (e >= 1u && e <= 253u) ? __uint_as_float((254u - e) << 23) : 1.f / s;
}
// a wave-uniform value that was computed on the vector unit (loaded words, scales): one copy in a scalar register instead of a vector register
__device__ __forceinline__ float ch_uniform(float v) {
#ifdef DN_EMULATE
    return v;
#else
    return __uint_as_float((unsigned)__builtin_amdgcn_readfirstlane((int)__float_as_uint(v)));
#endif
}
__device__ __forceinline__ int ch_uniform_i(int v) {
#ifdef DN_EMULATE
    return v;
#else
    return __builtin_amdgcn_readfirstlane(v);
#endif
}
// Streaming global accesses of the chained kernels (dn_common.h): every activation tile is written exactly once per launch and consumed by
// a LATER kernel.  Build with -DDN_CH_STREAM=0 for plain stores (A/B).
#ifndef DN_CH_STREAM
#define DN_CH_STREAM 1
#endif
__device__ __forceinline__ void ch_st4(float* p, const float4 v) {
#if DN_CH_STREAM
    dn_st4_stream(p, v);
#else
    *reinterpret_cast<float4*>(p) = v;
#endif
}
// ---- whole-line stores of accumulator-layout tiles.  Lane (m, q) of a wave holds channels 16 nt + 4 q .. + 3 of row m: a store of tile nt
// writes 64 contiguous bytes per row -- half a 128-byte line; the other half comes with the store of tile nt + 1.  As streaming stores
// those halves reach memory separately often enough to cost 19-27 % more written bytes than the arrays hold (WRITE_SIZE of the chained
// kernels: 772 / 719 MB for 649 / 568 MB; plain stores are exact but displace the gathered rows from the L2 and measure slower,
// profiles/r06_line_stores.txt).  Here the two tiles are exchanged between rows m and m + 8 (DPP row_ror:8: one v_mov per register) so
// that ONE instruction writes tiles nt, nt + 1 of rows 0..7 and the next one of rows 8..15: every row segment written by an
// instruction is a whole line.  base: the array; row: this lane's row; v_end: rows below it exist.
#ifndef DN_CH_LINES
#define DN_CH_LINES 1
#endif
__device__ __forceinline__ float ch_rot8(float v) {      // the value lane (m + 8) % 16 of the same 16-lane row holds
#ifdef DN_EMULATE
    const int l_ = (int)(threadIdx.x & 63);
    return __shfl(v, (l_ & 48) | ((l_ + 8) & 15), 64);
#else
    return __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), 0x128, 0xf, 0xf, false));   // row_ror:8
#endif
}
// (tile: a callable nt -> float4, evaluated pair by pair: two tiles and their rotated copies are alive at a time)
// LINES = false: the plain form, one 64-byte segment per row and instruction (kernels at the register limit: the exchange costs ~16 registers)
template <int NT, bool LINES = true, typename F>
__device__ __forceinline__ void ch_st_tiles(float* base, const int C, const long long row, const int v_end, const int m, const int q, F&& tile) {
  if constexpr (LINES && DN_CH_LINES != 0) {
    static_assert(NT % 2 == 0, "tile pairs");
    const bool hi = m >= 8;
    const long long r0 = hi ? row - 8 : row, r1 = hi ? row : row + 8;      // rows this lane writes in the first / second instruction of a pair
    float* p0 = base + r0 * C + 4 * q + (hi ? 16 : 0);
    float* p1 = base + r1 * C + 4 * q + (hi ? 16 : 0);
    const bool ok0 = r0 < v_end, ok1 = r1 < v_end;
#pragma unroll
    for (int np = 0; np < NT; np += 2) {
        const float4 a = tile(np), b = tile(np + 1);
        const float4 ra = make_float4(ch_rot8(a.x), ch_rot8(a.y), ch_rot8(a.z), ch_rot8(a.w));
        const float4 rb = make_float4(ch_rot8(b.x), ch_rot8(b.y), ch_rot8(b.z), ch_rot8(b.w));
        if (ok0) ch_st4(p0 + 16 * np, hi ? rb : a);
        if (ok1) ch_st4(p1 + 16 * np, hi ? b : ra);
    }
  } else {
    if (row < v_end) {
        float* o = base + row * C + 4 * q;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) ch_st4(o + 16 * nt, tile(nt));
    }
  }
}
// (the LOADS of the backward's single-use tiles are plain since round 6: a 64-byte request per row is half a line, the other half comes with the next
// tile's request, and a streaming line is often gone by then -- FETCH_SIZE 387 -> 375 k KiB, block backward 851-858 -> 842-846 us; -DDN_CH_STREAM_LD=1: A/B)
#ifndef DN_CH_STREAM_LD
#define DN_CH_STREAM_LD 0
#endif
__device__ __forceinline__ float4 ch_ld4(const float* p) {
#if DN_CH_STREAM_LD
    return dn_ld4_stream(p);
#else
    return *reinterpret_cast<const float4*>(p);
#endif
}
// int shuffle (the emulator's shuffles carry floats: row indices of its test meshes are exact in fp32)
__device__ __forceinline__ int ch_shfl_i(int v, int src_lane) {
#ifdef DN_EMULATE
    return (int)__shfl((float)v, src_lane, 64);
#else
    return __shfl(v, src_lane, 64);
#endif
}
// Orders this wave's LDS traffic: what its lanes wrote before is what its lanes read after (LDS operations of a wave execute in order;
// this keeps the compiler from moving them across).  No hardware barrier: the waves of a workgroup are at different places here.
__device__ __forceinline__ void ch_wave_sync() {
#ifdef DN_EMULATE
    dnemu::wave_barrier();
#else
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#endif
}
// Maximum over the wave, the same value in every lane.  Data-parallel-primitive moves inside the rows of 16 lanes, row broadcasts across them,
// one v_readlane: ~10 dependent VALU instructions.  The shuffle form (six ds_bpermute round trips through the LDS crossbar, each behind an
// lgkmcnt(0)) cost ~700 cycles per reduction, three to five times per pass of a wave (profiles/r05_chain_trace_hh.txt).
__device__ __forceinline__ float ch_wave_max(float m) {
#ifdef DN_EMULATE
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { const float o = __shfl_xor(m, d, 64); m = o > m ? o : m; }
    return m;
#else
    int v = (int)__float_as_uint(m);
#define CH_DPP_MAX_(ctrl_, rows_)                                                                        \
    do {                                                                                                 \
        const int t_ = __builtin_amdgcn_update_dpp(v, v, (ctrl_), (rows_), 0xf, false);                  \
        const float a_ = __uint_as_float((unsigned)t_), b_ = __uint_as_float((unsigned)v);               \
        v = (int)__float_as_uint(a_ > b_ ? a_ : b_);                                                     \
    } while (0)
    CH_DPP_MAX_(0xB1, 0xf);     // quad_perm [1, 0, 3, 2]
    CH_DPP_MAX_(0x4E, 0xf);     // quad_perm [2, 3, 0, 1]
    CH_DPP_MAX_(0x141, 0xf);    // row_half_mirror
    CH_DPP_MAX_(0x140, 0xf);    // row_mirror: every lane of a row holds the row's maximum
    CH_DPP_MAX_(0x142, 0xa);    // row_bcast15 into rows 1, 3
    CH_DPP_MAX_(0x143, 0xc);    // row_bcast31 into rows 2, 3: lane 63 holds the wave's maximum
#undef CH_DPP_MAX_
    return __uint_as_float((unsigned)__builtin_amdgcn_readlane(v, 63));
#endif
}

#ifndef DN_CH_RING
#define DN_CH_RING 4      // LDS slots of the piece stream; DN_CH_RING - 1 pieces are requested ahead of the one being multiplied
#endif
#ifndef DN_CH_GCHUNK
#define DN_CH_GCHUNK 4    // pattern entries gathered per step (all their row pieces in flight together)
#endif
// One LDS-DMA request: 16 bytes per lane, global -> LDS, lane l's data lands at lds_byte + 16 l (lds_byte wave-uniform).  Inline asm on
// purpose: the compiler's waitcnt bookkeeping does not see it, so it neither drains the request queue at a barrier nor at the next use of
// an ordinary load while requests are in flight; the kernel counts them itself (CH_WAIT_PIECES).  Loads return in order, so a compiler-made
// vmcnt(n) for one of its own loads stays correct with these requests in the queue (it may wait for some of them too: conservative).
__device__ __forceinline__ void ch_dma16(const uint4* gsrc, uint4* lds_generic, unsigned lds_byte) {
#ifdef DN_EMULATE
    (void)lds_byte;
    lds_generic[threadIdx.x & 63] = *gsrc;
#else
    (void)lds_generic;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_byte) : "memory");
#endif
}


// ---- piece x fragment products (inside a kernel that defines NT, lane and ws_ = the ring slot being read) --------------------------
// (CH_PF, a constexpr of the kernel that expands the macros: how many tile pairs ahead of their MFMAs the weight fragments are read from LDS)
// acc[h][nt] += W(piece) * frag[h]   (hi*lo, lo*hi, hi*hi: smallest terms first); two output tiles at a time, so that an MFMA and the
// next one into the same accumulator are four issues apart
#define CH_WLOAD(W_, np_) do { W_[0] = ws_[(np_) * 64 + lane]; W_[1] = ws_[(NT + (np_)) * 64 + lane];                       \
                           W_[2] = ws_[((np_) + 1) * 64 + lane]; W_[3] = ws_[(NT + (np_) + 1) * 64 + lane]; } while (0)
// (HH = 16-row halves per wave, a template parameter of the kernel that expands the macro: with HH = 1 the second half's MFMAs are discarded)
#define CH_MMA2(ACC, FH0, FL0, FH1, FL1)                                                                                \
do {                                                                                                                \
    uint4 wq_[CH_PF + 1][4];   /* weight fragments of two output tiles (hi a, lo a, hi b, lo b), fetched CH_PF pairs ahead of their MFMAs */ \
    _Pragma("unroll") for (int pf_ = 0; pf_ < CH_PF; ++pf_) if (2 * pf_ < NT) CH_WLOAD(wq_[pf_], 2 * pf_);                                                                                            \
    _Pragma("unroll") for (int np_ = 0; np_ < NT; np_ += 2) {                                                       \
        const int cb_ = (np_ >> 1) % (CH_PF + 1);                                                                   \
        if (np_ + 2 * CH_PF < NT) CH_WLOAD(wq_[((np_ >> 1) + CH_PF) % (CH_PF + 1)], np_ + 2 * CH_PF);                                                          \
        ACC[0][np_] = dn_mfma16_f16(wq_[cb_][0], FL0, ACC[0][np_]);                                                 \
        if constexpr (HH > 1) ACC[HH - 1][np_] = dn_mfma16_f16(wq_[cb_][0], FL1, ACC[HH - 1][np_]);                 \
        ACC[0][np_ + 1] = dn_mfma16_f16(wq_[cb_][2], FL0, ACC[0][np_ + 1]);                                         \
        if constexpr (HH > 1) ACC[HH - 1][np_ + 1] = dn_mfma16_f16(wq_[cb_][2], FL1, ACC[HH - 1][np_ + 1]);         \
        ACC[0][np_] = dn_mfma16_f16(wq_[cb_][1], FH0, ACC[0][np_]);                                                 \
        if constexpr (HH > 1) ACC[HH - 1][np_] = dn_mfma16_f16(wq_[cb_][1], FH1, ACC[HH - 1][np_]);                 \
        ACC[0][np_ + 1] = dn_mfma16_f16(wq_[cb_][3], FH0, ACC[0][np_ + 1]);                                         \
        if constexpr (HH > 1) ACC[HH - 1][np_ + 1] = dn_mfma16_f16(wq_[cb_][3], FH1, ACC[HH - 1][np_ + 1]);         \
        ACC[0][np_] = dn_mfma16_f16(wq_[cb_][0], FH0, ACC[0][np_]);                                                 \
        if constexpr (HH > 1) ACC[HH - 1][np_] = dn_mfma16_f16(wq_[cb_][0], FH1, ACC[HH - 1][np_]);                 \
        ACC[0][np_ + 1] = dn_mfma16_f16(wq_[cb_][2], FH0, ACC[0][np_ + 1]);                                         \
        if constexpr (HH > 1) ACC[HH - 1][np_ + 1] = dn_mfma16_f16(wq_[cb_][2], FH1, ACC[HH - 1][np_ + 1]);         \
    }                                                                                                               \
} while (0)

// one product, ACCROW[nt] += W(piece) * frag, weight fragments fetched one tile pair ahead
#define CH_MMA1(ACCROW, FH, FL)                                                                                         \
do {                                                                                                                \
    uint4 wq_[CH_PF + 1][4];                                                                                        \
    _Pragma("unroll") for (int pf_ = 0; pf_ < CH_PF; ++pf_) if (2 * pf_ < NT) CH_WLOAD(wq_[pf_], 2 * pf_);                                                                                            \
    _Pragma("unroll") for (int np_ = 0; np_ < NT; np_ += 2) {                                                       \
        const int cb_ = (np_ >> 1) % (CH_PF + 1);                                                                   \
        if (np_ + 2 * CH_PF < NT) CH_WLOAD(wq_[((np_ >> 1) + CH_PF) % (CH_PF + 1)], np_ + 2 * CH_PF);                                                          \
        ACCROW[np_] = dn_mfma16_f16(wq_[cb_][0], FL, ACCROW[np_]);                                                  \
        ACCROW[np_ + 1] = dn_mfma16_f16(wq_[cb_][2], FL, ACCROW[np_ + 1]);                                          \
        ACCROW[np_] = dn_mfma16_f16(wq_[cb_][1], FH, ACCROW[np_]);                                                  \
        ACCROW[np_ + 1] = dn_mfma16_f16(wq_[cb_][3], FH, ACCROW[np_ + 1]);                                          \
        ACCROW[np_] = dn_mfma16_f16(wq_[cb_][0], FH, ACCROW[np_]);                                                  \
        ACCROW[np_ + 1] = dn_mfma16_f16(wq_[cb_][2], FH, ACCROW[np_ + 1]);                                          \
    }                                                                                                               \
} while (0)

// two products sharing the weights, ACC[0] += W frag0 and ACC[1] += W frag1, with the weight fragments fetched one tile pair ahead (the
// gradient-feature stage of the one-wave-per-SIMD form: nobody else covers the LDS latency there, and the registers are not the limit)
#define CH_MMA2_PAIR(ACC, FH0, FL0, FH1, FL1)                                                                           \
do {                                                                                                                \
    uint4 wq_[CH_PF + 1][4];                                                                                        \
    _Pragma("unroll") for (int pf_ = 0; pf_ < CH_PF; ++pf_) if (2 * pf_ < NT) CH_WLOAD(wq_[pf_], 2 * pf_);                                                                                            \
    _Pragma("unroll") for (int np_ = 0; np_ < NT; np_ += 2) {                                                       \
        const int cb_ = (np_ >> 1) % (CH_PF + 1);                                                                   \
        if (np_ + 2 * CH_PF < NT) CH_WLOAD(wq_[((np_ >> 1) + CH_PF) % (CH_PF + 1)], np_ + 2 * CH_PF);                                                          \
        ACC[0][np_] = dn_mfma16_f16(wq_[cb_][0], FL0, ACC[0][np_]);                                                 \
        ACC[1][np_] = dn_mfma16_f16(wq_[cb_][0], FL1, ACC[1][np_]);                                                 \
        ACC[0][np_ + 1] = dn_mfma16_f16(wq_[cb_][2], FL0, ACC[0][np_ + 1]);                                         \
        ACC[1][np_ + 1] = dn_mfma16_f16(wq_[cb_][2], FL1, ACC[1][np_ + 1]);                                         \
        ACC[0][np_] = dn_mfma16_f16(wq_[cb_][1], FH0, ACC[0][np_]);                                                 \
        ACC[1][np_] = dn_mfma16_f16(wq_[cb_][1], FH1, ACC[1][np_]);                                                 \
        ACC[0][np_ + 1] = dn_mfma16_f16(wq_[cb_][3], FH0, ACC[0][np_ + 1]);                                         \
        ACC[1][np_ + 1] = dn_mfma16_f16(wq_[cb_][3], FH1, ACC[1][np_ + 1]);                                         \
        ACC[0][np_] = dn_mfma16_f16(wq_[cb_][0], FH0, ACC[0][np_]);                                                 \
        ACC[1][np_] = dn_mfma16_f16(wq_[cb_][0], FH1, ACC[1][np_]);                                                 \
        ACC[0][np_ + 1] = dn_mfma16_f16(wq_[cb_][2], FH0, ACC[0][np_ + 1]);                                         \
        ACC[1][np_ + 1] = dn_mfma16_f16(wq_[cb_][2], FH1, ACC[1][np_ + 1]);                                         \
    }                                                                                                               \
} while (0)

// the same product with the weight fragments fetched tile pair by tile pair (16 registers less): for the gradient-feature stage, whose
// live set (gx, gy, both accumulators, the other half's tanh features) is the kernel's peak
#define CH_MMA2_LEAN(ACC, FH0, FL0, FH1, FL1)                                                                           \
_Pragma("unroll") for (int np_ = 0; np_ < NT; np_ += 2) {                                                           \
    uint4 w_[4];                                                                                                    \
    CH_WLOAD(w_, np_);                                                                                              \
    ACC[0][np_] = dn_mfma16_f16(w_[0], FL0, ACC[0][np_]);                                                           \
    ACC[1][np_] = dn_mfma16_f16(w_[0], FL1, ACC[1][np_]);                                                           \
    ACC[0][np_ + 1] = dn_mfma16_f16(w_[2], FL0, ACC[0][np_ + 1]);                                                   \
    ACC[1][np_ + 1] = dn_mfma16_f16(w_[2], FL1, ACC[1][np_ + 1]);                                                   \
    ACC[0][np_] = dn_mfma16_f16(w_[1], FH0, ACC[0][np_]);                                                           \
    ACC[1][np_] = dn_mfma16_f16(w_[1], FH1, ACC[1][np_]);                                                           \
    ACC[0][np_ + 1] = dn_mfma16_f16(w_[3], FH0, ACC[0][np_ + 1]);                                                   \
    ACC[1][np_ + 1] = dn_mfma16_f16(w_[3], FH1, ACC[1][np_ + 1]);                                                   \
    ACC[0][np_] = dn_mfma16_f16(w_[0], FH0, ACC[0][np_]);                                                           \
    ACC[1][np_] = dn_mfma16_f16(w_[0], FH1, ACC[1][np_]);                                                           \
    ACC[0][np_ + 1] = dn_mfma16_f16(w_[2], FH0, ACC[0][np_ + 1]);                                                   \
    ACC[1][np_ + 1] = dn_mfma16_f16(w_[2], FH1, ACC[1][np_ + 1]);                                                   \
}

// three products sharing the piece: ACC[op][nt] += W(piece) * FR[op] (FR[op][0] = hi, [1] = lo plane of the operand fragment), op = xd, gx, gy of
// the spectral-gradient stage.  One weight-fragment read from LDS feeds three MFMAs; consecutive MFMAs go to different accumulators (the same
// accumulator comes round every sixth issue).  Lean form (the stage holds 24 operand fragments and three accumulator rows: the register peak).
#define CH_MMA3(ACC, FR)                                                                                                \
_Pragma("unroll") for (int np_ = 0; np_ < NT; np_ += 2) {                                                           \
    uint4 w_[4];                                                                                                    \
    CH_WLOAD(w_, np_);                                                                                              \
    _Pragma("unroll") for (int op_ = 0; op_ < 3; ++op_) {                                                           \
        ACC[op_][np_] = dn_mfma16_f16(w_[0], FR[op_][1], ACC[op_][np_]);                                            \
        ACC[op_][np_ + 1] = dn_mfma16_f16(w_[2], FR[op_][1], ACC[op_][np_ + 1]);                                    \
    }                                                                                                               \
    _Pragma("unroll") for (int op_ = 0; op_ < 3; ++op_) {                                                           \
        ACC[op_][np_] = dn_mfma16_f16(w_[1], FR[op_][0], ACC[op_][np_]);                                            \
        ACC[op_][np_ + 1] = dn_mfma16_f16(w_[3], FR[op_][0], ACC[op_][np_ + 1]);                                    \
    }                                                                                                               \
    _Pragma("unroll") for (int op_ = 0; op_ < 3; ++op_) {                                                           \
        ACC[op_][np_] = dn_mfma16_f16(w_[0], FR[op_][0], ACC[op_][np_]);                                            \
        ACC[op_][np_ + 1] = dn_mfma16_f16(w_[2], FR[op_][0], ACC[op_][np_ + 1]);                                    \
    }                                                                                                               \
}
