// dn_common.h -- shared definitions for the gfx950 (MI355X / CDNA4) DiffusionNet kernels.
//
// The kernels are written for wave64 and the gfx950 matrix pipe: split-bf16 MFMA (v_mfma_f32_32x32x16_bf16 on a 3-term
// split of fp32 operands, fp32 accumulation) on the aligned 128-wide paths, exact-f32 MFMA (v_mfma_f32_32x32x2_f32)
// everywhere else.  The only concession to portability is the DN_EMULATE switch, which lets tests/emu compile the same
// sources for the host to check index logic; the product build never defines it.
#pragma once
#include <stdint.h>
#include <stddef.h>

#ifdef DN_EMULATE
#include "hipemu.h"
#define DN_LAUNCH(kernel, grid, block, smem, stream, ...) \
    dnemu::launch((grid), (block), (smem), [=]() { kernel(__VA_ARGS__); })
#define DN_DYN_SMEM(name) char* name = dnemu::dyn_smem()
#define DN_RESTRICT
#define DN_WAVES_PER_EU(n)
#define DN_MIN_WAVES_PER_EU(n)
#define DN_SETPRIO(n) do {} while (0)
#else
#include <hip/hip_runtime.h>
#define DN_LAUNCH(kernel, grid, block, smem, stream, ...) \
    hipLaunchKernelGGL(kernel, (grid), (block), (smem), (stream), __VA_ARGS__)
#define DN_DYN_SMEM(name) extern __shared__ __attribute__((aligned(16))) char name[]
#define DN_RESTRICT __restrict__
// occupancy the kernel is designed for: stops the scheduler from trading instruction order for registers it cannot use
#define DN_WAVES_PER_EU(n) __attribute__((amdgpu_waves_per_eu(n, n)))
#define DN_MIN_WAVES_PER_EU(n) __attribute__((amdgpu_waves_per_eu(n)))
#define DN_SETPRIO(n) __builtin_amdgcn_s_setprio(n)
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));

// One 32x32x2 exact-f32 MFMA step: lane l supplies A[i=l&31][k=l>>5] and B[k=l>>5][j=l&31];
// accumulator register r of lane l is D[(r&3)+8*(r>>2)+4*(l>>5)][l&31].
__device__ __forceinline__ f32x16 dn_mfma(float a, float b, f32x16 c) {
#ifdef DN_EMULATE
    return dnemu_mfma_f32_32x32x2f32(a, b, c);
#else
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
#endif
}
// One 32x32x16 bf16 MFMA step (fp32 accumulate): lane l supplies eight consecutive-k bf16 of row i=l&31 (A) / column j=l&31
// (B), k = 8*(l>>5) .. +7, packed in a uint4.
__device__ __forceinline__ f32x16 dn_mfma_bf16(uint4 a, uint4 b, f32x16 c) {
#ifdef DN_EMULATE
    return dnemu_mfma_f32_32x32x16_bf16(a, b, c);
#else
    typedef __bf16 dn_bf16x8 __attribute__((ext_vector_type(8)));
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(dn_bf16x8, a), __builtin_bit_cast(dn_bf16x8, b), c, 0, 0, 0);
#endif
}
// One 32x32x16 f16 MFMA step (fp32 accumulate); operand layout as the bf16 form.
__device__ __forceinline__ f32x16 dn_mfma_f16(uint4 a, uint4 b, f32x16 c) {
#ifdef DN_EMULATE
    return dnemu_mfma_f32_32x32x16_f16(a, b, c);
#else
    typedef _Float16 dn_f16x8 __attribute__((ext_vector_type(8)));
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(dn_f16x8, a), __builtin_bit_cast(dn_f16x8, b), c, 0, 0, 0);
#endif
}
// gfx950 LDS transpose read: lane l of a 16-lane group gets column l&15 of the 4x16 16-bit matrix whose row j is named by
// the addresses of lanes 4j..4j+3 of the group (each an 8-byte chunk).  Turns k-major LDS tiles into MFMA operands.
__device__ __forceinline__ uint2 dn_lds_tr16(const unsigned char* p) {
#ifdef DN_EMULATE
    return dnemu_ds_read_tr16_b64(p);
#else
    typedef short dn_v4s __attribute__((ext_vector_type(4)));
    const dn_v4s v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) dn_v4s*)p);
    return __builtin_bit_cast(uint2, v);
#endif
}
// 3-term bf16 split of an fp32 value, x = hi + mid + lo up to 2^-24 relative (round-to-nearest-even at every step; the
// residuals are exact in fp32).  With the six largest cross products of two split operands accumulated in fp32 the result
// is as accurate as an fp32 FMA chain (measured rel-L2 1.6e-7 vs 2.0e-7 against fp64, profiles/r01_exp_bf16x3.txt) at
// 16/6 = 2.7x the f32-MFMA rate.
// The split works on PAIRS: gfx950's v_cvt_pk_bf16_f32 rounds and packs two values in one instruction; 11 VALU instructions
// per pair (the software-rounding formulation took ~27 and made the staging phase, not the MFMAs, the critical path).
__device__ __forceinline__ unsigned dn_bf16_bits(float x, float& back) {   // software RNE (emulator and odd call sites)
    unsigned u = __float_as_uint(x);
    u += 0x7fffu + ((u >> 16) & 1u);
    u >>= 16;
    back = __uint_as_float(u << 16);
    return u;
}
// packed dwords: low half = bf16 of x, high half = bf16 of y
__device__ __forceinline__ void dn_split3_pair(float x, float y, unsigned& hi, unsigned& mid, unsigned& lo) {
#if defined(DN_EMULATE)
    float bx, by;
    unsigned hx = dn_bf16_bits(x, bx), hy = dn_bf16_bits(y, by);
    hi = hx | (hy << 16);
    const float rx = x - bx, ry = y - by;
    hx = dn_bf16_bits(rx, bx); hy = dn_bf16_bits(ry, by);
    mid = hx | (hy << 16);
    hx = dn_bf16_bits(rx - bx, bx); hy = dn_bf16_bits(ry - by, by);
    lo = hx | (hy << 16);
#else
    // v_cvt_pk_bf16_f32 rounds (RNE) and packs two values; the residuals are plain v_sub_f32 -- no packed-f32 VALU ops
    // anywhere in this library (see dn_f4_mul for why)
    typedef float dn_f2 __attribute__((ext_vector_type(2)));
    typedef __bf16 dn_bf2 __attribute__((ext_vector_type(2)));
    hi = __builtin_bit_cast(unsigned, __builtin_convertvector(dn_f2{x, y}, dn_bf2));
    const float rx = x - __uint_as_float(hi << 16), ry = y - __uint_as_float(hi & 0xffff0000u);
    mid = __builtin_bit_cast(unsigned, __builtin_convertvector(dn_f2{rx, ry}, dn_bf2));
    const float tx = rx - __uint_as_float(mid << 16), ty = ry - __uint_as_float(mid & 0xffff0000u);
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(dn_f2{tx, ty}, dn_bf2));
#endif
}
// a float4 (four consecutive k) -> one 8-byte chunk per plane
__device__ __forceinline__ void dn_split3_f4(float4 v, uint2& hi, uint2& mid, uint2& lo) {
    dn_split3_pair(v.x, v.y, hi.x, mid.x, lo.x);
    dn_split3_pair(v.z, v.w, hi.y, mid.y, lo.y);
}
// 2-term fp16 split, x * s = hi + lo up to 2^-22 relative, for |x * s| < 65504 (RNE; the residual is exact in fp32; s is a power of two
// chosen from the operand's largest magnitude, so x * s is exact as well).  gfx950: v_cvt_pk_f16_f32 rounds and packs two values, the
// residual is one v_fma_mix_f32 (fp16 operand read in place): 6 VALU instructions per pair against 11 for the 3-term bf16 split --
// and three cross products (hi*lo, lo*hi, hi*hi) on v_mfma_f32_32x32x16_f16 instead of six, two LDS planes instead of three.
// Elements below 2^-14 / s lose their low term to fp16's subnormal spacing: an absolute error of 2^-25 / s, i.e. 2^-40 of the operand's
// largest magnitude -- far below fp32's own 2^-24 in every norm-wise measure (the tolerance of the parity tests is norm-wise).
__device__ __forceinline__ void dn_split2_pair(float x, float y, float s, unsigned& hi, unsigned& lo) {
    typedef float dn_f2 __attribute__((ext_vector_type(2)));
    typedef _Float16 dn_h2 __attribute__((ext_vector_type(2)));
    const float xs = x * s, ys = y * s;
    const dn_h2 h = __builtin_convertvector(dn_f2{xs, ys}, dn_h2);
    hi = __builtin_bit_cast(unsigned, h);
    const dn_f2 hb = __builtin_convertvector(h, dn_f2);
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(dn_f2{xs - hb.x, ys - hb.y}, dn_h2));
}
// NP = 3: bf16 planes (hi, mid, lo), s ignored;  NP = 2: fp16 planes (hi, lo) of v * s
template <int NP>
__device__ __forceinline__ void dn_split_f4(float4 v, float s, uint2 (&pl)[NP]) {
    if constexpr (NP == 3) dn_split3_f4(v, pl[0], pl[1], pl[2]);
    else { dn_split2_pair(v.x, v.y, s, pl[0].x, pl[1].x); dn_split2_pair(v.z, v.w, s, pl[0].y, pl[1].y); }
}
template <int NP>
__device__ __forceinline__ void dn_split_pair(float x, float y, float s, unsigned (&w)[NP]) {
    if constexpr (NP == 3) dn_split3_pair(x, y, w[0], w[1], w[2]);
    else dn_split2_pair(x, y, s, w[0], w[1]);
}
// power of two that puts `amax` into [2^14, 2^15) (1 for amax = 0; inf / nan operands propagate through the products on their own)
__device__ __forceinline__ float dn_pow2_scale(float amax) {
    const unsigned e = (__float_as_uint(amax) >> 23) & 0xffu;
    int f = 127 + 14 + 127 - (int)e;
    f = f > 254 ? 254 : (f < 1 ? 1 : f);
    return e == 0 ? 1.f : __uint_as_float((unsigned)f << 23);
}
// LDS bf16 plane: R rows x 32 k, 64 bytes per row = four 16-byte slots, slot' = slot ^ ((row>>2)&3): a wave's ds_read_b128 of
// one slot for 32 consecutive rows hits all sixteen 16-byte positions of the 256-byte bank row once per 16-lane group.
__device__ __forceinline__ int dn_plane_off(int row, int slot) { return row * 64 + ((slot ^ ((row >> 2) & 3)) << 4); }

// Dropout keep bits of the four consecutive columns 4*c4 .. 4*c4+3 of row `grow` (p = 1/2): bits 28..31 of a 32-bit
// integer hash (two multiply-xorshift rounds, "lowbias32") of the group index keyed by the 64-bit seed.  One hash per float4;
// every kernel path derives the same bit for the same (seed, row, column), so the mask does not depend on tiling.
__device__ __forceinline__ unsigned dn_keep_bits(unsigned long long seed, long long grow, int c4, int groups_per_row) {
    unsigned x = (unsigned)(grow * groups_per_row + c4) ^ (unsigned)seed;
    x += (unsigned)(seed >> 32) * 0x9E3779B9u;
    x ^= x >> 16; x *= 0x7feb352du;
    x ^= x >> 15; x *= 0x846ca68bu;
    x ^= x >> 16;
    return x >> 28;
}
// the same bits as a 4-byte keep mask (byte e != 0 <=> column 4*c4+e kept), the format of an explicit uint8 mask load
__device__ __forceinline__ unsigned dn_keep_bytes(unsigned bits) {
    return (bits & 1u) | ((bits & 2u) << 7) | ((bits & 4u) << 14) | ((bits & 8u) << 21);
}

__device__ __forceinline__ int dn_acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// A unit of row work: rows [row0,row0+nrows) of the concatenated vertex axis, all belonging to
// mesh `mesh` (tiles/chunks never straddle meshes).  Built on the host once per mesh batch.
struct DnTile {
    int row0, nrows, mesh, aux;
};

__device__ __forceinline__ float4 dn_f4_zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
// Streaming (nontemporal) 16-byte accesses for data a kernel touches once and a LATER kernel consumes: they should not displace what the L2
// is needed for inside the kernel (gathered neighbour rows, weight pieces).  The hand-written copy gains 6.2 vs 5.2-5.9 TB/s from them.
typedef float dn_v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void dn_st4_stream(float* p, const float4 v) {
#ifndef DN_EMULATE
    dn_v4f t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
    __builtin_nontemporal_store(t, reinterpret_cast<dn_v4f*>(p));
#else
    *reinterpret_cast<float4*>(p) = v;
#endif
}
__device__ __forceinline__ float4 dn_ld4_stream(const float* p) {
#ifndef DN_EMULATE
    const dn_v4f t = __builtin_nontemporal_load(reinterpret_cast<const dn_v4f*>(p));
    return make_float4(t.x, t.y, t.z, t.w);
#else
    return *reinterpret_cast<const float4*>(p);
#endif
}
// streaming (nontemporal) 16-byte accesses: the hand-written copy runs at 6.2-6.4 TB/s with them against 5.2-5.9 TB/s without (tools/kbench copyk);
// in the row GEMM they moved the block by +-2 % either way (round 3, tools/experiments/rowgemm_ws_knobs/) and are not used there.
typedef float dn_vf4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void dn_store_f4_nt(float* p, const float4& v) {
#ifdef DN_EMULATE
    *reinterpret_cast<float4*>(p) = v;
#else
    __builtin_nontemporal_store(dn_vf4{v.x, v.y, v.z, v.w}, reinterpret_cast<dn_vf4*>(p));
#endif
}
__device__ __forceinline__ float4 dn_load_f4_nt(const float* p) {
#ifdef DN_EMULATE
    return *reinterpret_cast<const float4*>(p);
#else
    const dn_vf4 v = __builtin_nontemporal_load(reinterpret_cast<const dn_vf4*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
#endif
}
// L1-bypassing ("sc1") 8-byte loads: agent-scope relaxed atomics.  What a workgroup reads of another workgroup's write-through (sc1)
// stores inside ONE launch (dn_diffuse.hip): MI355X_MICROARCH.md, inter-workgroup visibility -- "sc1 loads may replace the acquire only
// when the producer stored sc1".  Compiler-tracked (no inline asm: an asm load's destination may be spilled before the data arrives).
__device__ __forceinline__ float2 dn_ld2_coherent(const float* p) {
#ifdef DN_EMULATE
    return *reinterpret_cast<const float2*>(p);
#else
    const unsigned long long u = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_float2(__uint_as_float((unsigned)u), __uint_as_float((unsigned)(u >> 32)));
#endif
}
__device__ __forceinline__ float4 dn_ld4_coherent(const float* p) {
    const float2 lo = dn_ld2_coherent(p), hi = dn_ld2_coherent(p + 2);
    return make_float4(lo.x, lo.y, hi.x, hi.y);
}
// Plain multiplies on purpose.  An explicit two-element vector multiply here (v_pk_mul_f32 with a broadcast operand in src1,
// `op_sel_hi:[1,0]`) produced INTERMITTENTLY wrong low halves on gfx950 / ROCm 7.2: one stale bf16 pair in a few launches per
// thousand at small sizes, in every launch at the benchmark size -- a whole output column of a tile off by ~1e-1 relative, found
// by tools/determinism_stress.py, invisible to the 2e-4 gradient tolerance.  -DDN_F4_PACKED restores that form for re-testing;
// the packed forms the compiler's SLP pass emits from the code below were bitwise stable in the same stress runs.
__device__ __forceinline__ float4 dn_f4_mul(float4 a, float4 b) {
#if defined(DN_F4_PACKED) && !defined(DN_EMULATE)   // experiment switch: explicit two-element vector multiplies
    typedef float dn_f2 __attribute__((ext_vector_type(2)));
    const dn_f2 lo = dn_f2{a.x, a.y} * dn_f2{b.x, b.y}, hi = dn_f2{a.z, a.w} * dn_f2{b.z, b.w};
    return make_float4(lo.x, lo.y, hi.x, hi.y);
#else
    return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w);
#endif
}
__device__ __forceinline__ float4 dn_f4_scale(float4 a, float s) {
#if defined(DN_F4_PACKED) && !defined(DN_EMULATE)
    return dn_f4_mul(a, make_float4(s, s, s, s));
#else
    return make_float4(a.x * s, a.y * s, a.z * s, a.w * s);
#endif
}
__device__ __forceinline__ float dn_f4_get(const float4& v, int t) {
    return t == 0 ? v.x : (t == 1 ? v.y : (t == 2 ? v.z : v.w));
}

// LDS "COLK" tile: R rows x 32 floats (one 32-wide slice of the contraction axis per row),
// stored row-major with the eight 16-byte slots of each 128-byte row XOR-swizzled by
// (row>>1)&7.  A wave's ds_read_b128 of slot q for 32 consecutive rows then touches all
// sixteen 16-byte positions of the 256-byte bank row once per 16-lane group (conflict-free),
// and a row's eight slots written by eight consecutive lanes stay inside one 128-byte row.
__device__ __forceinline__ int dn_colk_off(int row, int slot) { return row * 32 + ((slot ^ ((row >> 1) & 7)) << 2); }

// ---------------------------------------------------------------------------------------
// row-tile GEMM   out[r, n] = epilogue( sum_s sum_k A_s[r, k] * B_s(k, n) )    (dn_rowgemm.hip, dn_rowgemm_persist.hip)
// ---------------------------------------------------------------------------------------
#define DN_TM 128      // rows per workgroup tile
#define DN_KB 32       // contraction slice staged per step

// Magnitude bound of an operand of the split-fp16 engine: max(c, *p[0], *p[1], *p[2]) * (*mul), null pointers skipped.  The words are
// written by the kernels that produced the tensors (atomic max over |value|); `mul` covers elementwise products (a*b <= max|a| max|b|).
struct DnAmax {
    const float* p[3];
    const float* mul;
    float c;
};
// The words are read with device-scope atomic loads (they are written by atomics of the kernels launched before; belt and braces --
// the one run-to-run flip seen during bring-up was a host-side bug: a word zeroed concurrently with the store that set it).
__device__ __forceinline__ float dn_amax_word(const float* p) {
#ifdef DN_EMULATE
    return *p;
#else
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}
__device__ __forceinline__ float dn_amax_eval(const DnAmax& a) {
    float m = a.c;
#pragma unroll
    for (int i = 0; i < 3; ++i) if (a.p[i]) { const float v = dn_amax_word(a.p[i]); m = v > m ? v : m; }
    if (a.mul) m *= dn_amax_word(a.mul);
    return m;
}
// atomic max of a non-negative float through its bit pattern (monotonic for x >= 0; NaN never raises the word: it propagates through the
// data itself); one atomic per wave
// CHECK: read the word first and skip the atomic unless this wave would raise it -- for kernels with tens of thousands of short waves,
// whose atomics on one address would serialise in its L2 channel.  Without it the atomic is posted (no return value, the wave does not
// wait): right for long-lived waves (a few thousand commits per launch).
template <bool CHECK = false>
__device__ __forceinline__ void dn_amax_commit(float* word, float m) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { const float o = __shfl_xor(m, d, 64); m = o > m ? o : m; }
    if ((threadIdx.x & 63) == 0 && m > 0.f) {
        if (!CHECK || m > *reinterpret_cast<volatile float*>(word)) atomicMax(reinterpret_cast<unsigned*>(word), __float_as_uint(m));
    }
}
// One commit per WORKGROUP: every wave leaves its maximum in LDS and the last one to arrive (LDS counter) issues the single check-first
// atomic.  For small batches all waves of a launch are resident together and all see the word at its start value: a 7k-vertex launch of
// the wave-specialised row GEMM then queues 440 atomics on one address, 3.5 us behind an otherwise finished kernel.  lds2: two words
// (maximum, arrivals) zeroed before a barrier that every committing wave has passed.
__device__ __forceinline__ void dn_amax_commit_group(float* word, float m, unsigned* lds2, int nwaves) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { const float o = __shfl_xor(m, d, 64); m = o > m ? o : m; }
    if ((threadIdx.x & 63) == 0) {
        atomicMax(&lds2[0], __float_as_uint(m));
        if (atomicAdd(&lds2[1], 1u) == (unsigned)(nwaves - 1)) {
            const float mm = __uint_as_float(*reinterpret_cast<volatile unsigned*>(&lds2[0]));
            if (mm > 0.f && mm > *reinterpret_cast<volatile float*>(word)) atomicMax(reinterpret_cast<unsigned*>(word), __float_as_uint(mm));
        }
    }
}
// per-lane form for kernels whose lanes do not all reach the end together: skip the atomic unless this lane would raise the word
__device__ __forceinline__ void dn_amax_commit_lane(float* word, float m) {
    if (m > *reinterpret_cast<volatile float*>(word)) atomicMax(reinterpret_cast<unsigned*>(word), __float_as_uint(m));
}
__device__ __forceinline__ float dn_f4_amax(float m, const float4& v) {
    const float a = fabsf(v.x) > fabsf(v.y) ? fabsf(v.x) : fabsf(v.y), b = fabsf(v.z) > fabsf(v.w) ? fabsf(v.z) : fabsf(v.w);
    const float c = a > b ? a : b;
    return c > m ? c : m;
}

struct RgSeg {           // one column segment of the (virtually concatenated) A operand
    const float* p;      // [rows, ld] row-major
    const float* q;      // optional elementwise factor, same shape/ld (A = p*q), else null
    int ld, w;
};
struct RgArgs {
    const DnTile* tiles;
    RgSeg a[3];
    int nseg;
    // B operand per output o (0/1) and segment s: sign[o][s] * Bmat[o][s]
    const float* b[2][3];
    float bsign[2][3];
    int ldb;
    int b_colk;            // 0: B[k*ldb + n] ("NN");  1: B[n*ldb + k] (nn.Linear weight, "NT")
    long long b_mesh_stride;   // elements between consecutive meshes' B (0 = shared)
    int N;
    int aligned;           // 1: every width % 32 == 0, ld % 4 == 0, pointers 16-byte aligned, N % 4 == 0
    // epilogue
    int mode;
    float* o0; float* o1; float* o2; int ldo;
    const float* bias;
    const float* r0; const float* r1; const float* r2; int ldr;
    const float* rowv;
    const uint8_t* mask;
    unsigned long long rng_seed;   // != 0 with mask == null: Bernoulli(1/2) keep bits drawn in the epilogue (dn_keep_bits)
    const unsigned long long* rng_seed_dev;   // optional device word added to rng_seed by the kernel (a captured graph advances it per replay)
    float scale;
    int acct_rows;         // rows covered by the launch (= v_total of the mesh batch)
    // split-fp16 engine (f16 != 0): A (all segments) and B are scaled by powers of two derived from device words holding (an upper bound
    // of) their largest magnitudes -- written by the kernels that produced them -- and the result is scaled back exactly.  o_amax: optional
    // device word that receives max |o0| over the launch (atomic max on the bit pattern of a non-negative float; zeroed by the caller).
    int f16;
    DnAmax a_amax, b_amax;
    float* o_amax;
};
enum {
    DN_EPI_STORE = 0,        // o0 = acc (+bias)
    DN_EPI_BIAS_RELU = 1,    // o0 = relu(acc+bias) * (mask ? mask*scale : 1)
    DN_EPI_BIAS_RESID = 2,   // o0 = acc + bias + r0
    DN_EPI_GRADFEAT = 3,     // o0 = tanh(r0*acc0 + r1*acc1); o1 = acc0; o2 = acc1 (if non-null)
    DN_EPI_MUL_DFAC = 4,     // o0 = acc * (r0 > 0 ? scale : 0)
    DN_EPI_ADD = 5,          // o0 = acc + r0
    DN_EPI_DTANH = 6,        // o0 = acc * (1 - r0^2)
    DN_EPI_GRADFEAT_BWD = 7, // o0 = acc0 + r0*r1 ; o1 = acc1 + r0*r2
    DN_EPI_MASS_ADD = 8,     // o0 = (r0 ? r0 : 0) + rowv[row]*acc
};
#define DN_ERR_BAD_MODE 1

// ---------------------------------------------------------------------------------------
// split-V "TN" GEMM   partial[chunk][m, n] = sum_{r in chunk} A[r, m] * B[r, n]   (dn_tngemm.hip)
// ---------------------------------------------------------------------------------------
struct TnSeg {
    const float* p;
    const float* q;     // optional elementwise factor
    int ld, w;
};
struct TnArgs {
    const DnTile* chunks;
    TnSeg a[2]; int na; int M;
    TnSeg b[3]; int nb; int N;
    const float* b_rowscale;    // optional per-row factor on B (lumped mass)
    float* partial;             // [nchunks][M][N]
    float* colsum;              // optional [nchunks][M]: column sums of A over the chunk
    int aligned;                // widths % 4 == 0, ld % 4 == 0, 16-byte aligned pointers
    int group;                  // consecutive chunks accumulated by one workgroup (1 for per-mesh outputs)
    int nchunks;                // filled by the launcher
    int acct_rows;              // host-side accounting only
    int f16;                    // split-fp16 engine (aligned split path only); operand magnitude bounds as in RgArgs
    DnAmax a_amax, b_amax;
    int lin_nb, lin_ny, lin_nz; // filled by the launcher: > 0 = one-dimensional launch in the XCD-aware tile order (tngemm_x3_kernel)
};
// Allow more than 64 KiB of dynamic LDS for one kernel instantiation, once per DEVICE (function attributes are per device;
// `done` is the instantiation's own 64-bit device bitmap).  Not thread-safe beyond "setting it twice is harmless".
#ifndef DN_EMULATE
static inline int dn_lds_opt_in(const void* fn, size_t smem, unsigned long long* done) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) dev = 0;
    if (!((*done >> dev) & 1ull)) {
        const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return (int)e;   // reported by the launcher instead of surfacing later as an opaque launch failure
        *done |= 1ull << dev;
    }
    return 0;
}
#endif
// CUs of the CURRENT device (one workgroup per CU in the persistent kernels)
static inline int dn_num_cus() {
#ifdef DN_EMULATE
    return 3;
#else
    static int cus[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) dev = 0;
    if (cus[dev] == 0) {
        int v = 0;
        cus[dev] = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
    }
    return cus[dev];
#endif
}
// number of partial results a tngemm launch over `nchunks` chunks with grouping `group` writes
static inline int dn_tn_npartial(int nchunks, int group) { return (nchunks + (group < 1 ? 1 : group) - 1) / (group < 1 ? 1 : group); }
// grouping for sums over ALL rows (weight gradients): about two workgroups per CU
#ifndef DN_TN_TARGET_PARTIALS
#define DN_TN_TARGET_PARTIALS 512   // about two (lock-step) workgroups per CU (a wave-specialised one-per-CU kernel was measured and rejected: tools/experiments/tngemm_ws/)
#endif
static inline int dn_tn_global_group(int nchunks) { int g = (nchunks + DN_TN_TARGET_PARTIALS - 1) / DN_TN_TARGET_PARTIALS; return g < 1 ? 1 : g; }
// the same for an M x N result of several 128 x 128 output tiles: every (partial, tile) pair is a workgroup, so the partial count --
// and with it the partial-result traffic, 2 x 4 M N bytes each -- shrinks with the tile count (never below 128 partials; never
// more partials than dn_tn_global_group(nchunks) gives, which is what the workspace is sized for)
static inline int dn_tn_global_group_mn(int nchunks, int M, int N) {
    const int tiles = ((M + 127) / 128) * ((N + 127) / 128);
    const int g0 = dn_tn_global_group(nchunks);
    int target = DN_TN_TARGET_PARTIALS / (tiles < 1 ? 1 : tiles);
    if (target < 128) target = 128;
    int g = (nchunks + target - 1) / target;
    return g < g0 ? g0 : g;
}

// ---------------------------------------------------------------------------------------
// CSR gather (dn_sparse.hip)
// ---------------------------------------------------------------------------------------
struct SpArgs {
    const int* rowptr;
    const int* col;
    const float* va;   // values (null = 1.0)
    const float* vb;   // second value array sharing the pattern (modes FWD2/BWD2)
    const float* x1;
    const float* x2;
    const float* add;  // optional addend (ld = ldo)
    float* o1;
    float* o2;
    int nrows, C, ldx, ldo, mode;
    float div;         // DN_SP_ONE: result divided by this (exact mean of n gathered rows)
    long long acct_nnz; // host-side accounting only
    float* o_amax;     // optional device word: (a bound of) max |o1|, |o2| over the launch for split-fp16 consumers.  With op_norm and
    const float* op_norm; const float* in_amax;   // in_amax it is STORED as *in_amax * *op_norm (||G||_inf max|x|); else measured (zeroed by the caller)
};
enum { DN_SP_FWD2 = 0, DN_SP_BWD2 = 1, DN_SP_ONE = 2 };

// ---------------------------------------------------------------------------------------
// chained row pipeline of the block forward (dn_chain.hip): CSR gather -> gradient features -> MiniMLP in one launch
// ---------------------------------------------------------------------------------------
#define DN_CH_LAYERS 4   // MiniMLP depth the chained kernel takes (the default net has 3); deeper nets use the unfused path

struct ChainPrepPiece {
    const float* W;      // [C, ld] row-major (nn.Linear layout: W[out][in])
    const float* W2;     // optional second matrix of the same shape sharing the magnitude word (A_re / A_im)
    float* amax;         // device word that RECEIVES the largest magnitude of the matrix (every piece of a matrix stores the same value);
                         // its power-of-two scale puts the weights into [2^14, 2^15)
    int ld, col0;        // the piece holds columns col0 .. col0 + 31 (in the permuted order)
    int transposed, row0;   // transposed (backward products d_in = d_out W): the piece's rows are W's COLUMNS row0 .. row0 + C - 1 and its
                            // contraction slots W's ROWS col0 .. col0 + 31: element (r, slot) = W[(col0 + slot) * ld + row0 + r]
};
#define DN_CH_MAX_PIECES 72   // C = 256 with rotations and a three-layer MiniMLP: 16 + 24 + 16 + 16
struct ChainPrepArgs {
    ChainPrepPiece pc[DN_CH_MAX_PIECES];
    uint4* out;          // [npieces][2 * (C / 16) * 64]
    int npieces;
    // start-of-call bookkeeping done by the workgroup behind the last piece (what dn_launch_amax_init does for the unfused path):
    // word ranges zeroed, one word copied
    float* zero[4]; int zero_n[4]; int nzero; const float* copy_src; float* copy_dst;
    float* clamp_p; int clamp_n; float clamp_min;       // as AmaxInit's
    void zero_range(float* p, int n) { if (p && n > 0 && nzero < 4) { zero[nzero] = p; zero_n[nzero] = n; ++nzero; } }
};
struct ChainArgs {
    // gradient operators (shared CSR pattern, two value arrays) and the dense inputs
    const int* rowptr; const int* col; const float* vx; const float* vy;
    const float* x; const float* xd;
    int V;
    int with_grad, with_rot, n_mlp;
    // weight pieces: the n_gf gradient-feature pieces (T-major: A_re T, A_im T; the kernel streams them once per 16-row half), layer 0
    // ([g | x | xd] segments, T-major inside a segment), layer 1, ...
    const uint4* wp;
    int n_pieces, n_gf;
    const float* wa_amax;                 // magnitude word of A_re / A_im (joint)
    const float* w_amax[DN_CH_LAYERS];    // of W_j
    const float* bias[DN_CH_LAYERS];
    const uint8_t* mask[DN_CH_LAYERS];    // explicit keep-mask applied to the OUTPUT of layer j ([V, C] bytes), or null
    unsigned long long seed[DN_CH_LAYERS];   // != 0 with mask == null: keep bits drawn in the epilogue (dn_keep_bits)
    const unsigned long long* seed_dev;
    // outputs (null: not saved)
    float* gx; float* gy; float* g; float* bre; float* bim;
    float* h[DN_CH_LAYERS];
    float* out;
    // magnitudes
    const float* x_amax; const float* xd_amax; const float* grad_norm;
    float* g_amax;                        // receives the bound ||G||_inf max|xd| (the backward's split-fp16 products scale by it)
    float* h_amax[DN_CH_LAYERS];          // accumulate max |h_j|
    float* out_amax;                      // accumulates max |out|
    int units;                            // workgroup passes: ceil(V / (16 halves-per-wave waves-per-workgroup))
    // ---- spectral-gradient form (dn_spectral.hip; kernel template parameter KE = k_eig / 32 > 0): xd, gx, gy are NOT read / gathered but
    // computed in the kernel as [Phi | G_X Phi | G_Y Phi][rows] * ys[mesh] from the packed operands of the batch (sg_pack) and the scaled
    // spectrum's pieces (ysp); xd / rowptr / col / vx / vy / xd_amax / grad_norm above are unused
    const uint4* sg_pack;                 // [groups][3][KE][2][64] fragment-ordered fp16 (hi, lo) planes, 16 rows per group, 4 groups per unit
    const DnTile* sg_units;               // [sg_n_units] runs of <= dn_sg_unit_rows(K) rows of one mesh; unit u owns that many / 16 consecutive groups
    int sg_n_units, sg_unit_rows, sg_n_mesh;
    const float* sg_amax;                 // [n_mesh][4]: largest magnitudes of Phi, G_X Phi, G_Y Phi per mesh (their power-of-two scales), largest row 2-norm of Phi
    const uint4* ysp;                     // [n_mesh][KE][piece] the scaled spectrum as transposed weight pieces (dn_launch_spec_pieces)
    const float* ys_amax;                 // [2 n_mesh] largest magnitude of every mesh's scaled spectrum; behind them the largest column 2-norms
    float* xd_out;                        // [V, C] receives xd (saved for the backward), or null
    float* xd_amax_out;                   // accumulates max |xd| (g_amax accumulates max |gx|, |gy| in this form)
};
// backward of the same stages: d_out -> d(pre-activations) of every layer -> [d_x | d_xd | d_dots] -> d_gx, d_gy (dn_chain_bwd.hip)
struct ChainBwdArgs {
    const float* d_out;                   // [V, C] gradient of the block output
    const float* h[DN_CH_LAYERS];         // saved post-ReLU(+dropout) hidden activations h_j, j < n_mlp - 1
    const float* g; const float* gx; const float* gy; const float* bre; const float* bim;   // saved gradient-feature tensors (with_grad)
    int V;
    int with_grad, with_rot, n_mlp;
    float dscale[DN_CH_LAYERS];           // dropout scale of h_j (2 with dropout, else 1): d(pre-act j) = (d_a W_{j+1}) * (h_j > 0 ? dscale : 0)
    const uint4* wp; int n_pieces;        // transposed weight pieces in stream order: W_{n-1}, ..., W_1, W_0 segments [x | xd | g], then
                                          // [A_re T, A_im T] x 2 (the two 16-row halves of the gradient-feature stage)
    const float* wa_amax; const float* w_amax[DN_CH_LAYERS];
    const float* d_out_amax;              // magnitude word of d_out
    // outputs
    float* d_a[DN_CH_LAYERS];             // d_a[j]: d(pre-activation of layer j), j < n_mlp - 1  (the weight-gradient products read them)
    float* d_xacc;                        // d_out + d_a0 W_0[:, x segment]   (residual + x branch)
    float* d_xd;                          // d_a0 W_0[:, xd segment]
    float* d_dots;                        // (d_a0 W_0[:, g segment]) * (1 - g^2)
    float* d_gx; float* d_gy;             // d_dots * Bre + (d_dots gx) A_re + (d_dots gy) A_im ;  d_dots * Bim - (d_dots gx) A_im + (d_dots gy) A_re
    int units;
};
int dn_chain_bwd_pieces(int C, int with_grad, int with_rot, int n_mlp);
int dn_launch_chain_bwd(int npieces, const ChainBwdArgs& a, int C, hipStream_t stream, int hh = 2);   // hh: 16-row halves per wave (the piece list repeats the gradient-feature pieces hh times)
int dn_chain_pieces(int C, int with_grad, int with_rot, int n_mlp);
size_t dn_chain_ws_bytes(int C, int with_grad, int with_rot, int n_mlp);
bool dn_chain_eligible(int C, int n_mlp, const int* widths, int with_grad, long long g_nnz, int V, int backward = 0);   // (the backward kernel exists at C = 64, 128)
int dn_launch_chain_prep(const ChainPrepArgs& pa, int npieces, int C, hipStream_t stream);
int dn_launch_chain_fwd(int npieces, const ChainArgs& a, int C, hipStream_t stream, int hh = 2);
// ---- spectral-gradient operands (dn_spectral.hip): the gradient apply re-associated, gx = G_X (Phi ys) = (G_X Phi) ys.  Built once per
// mesh batch: G_X Phi, G_Y Phi by a CSR gather accumulated in fp64, then [Phi | G_X Phi | G_Y Phi] split into fp16 (hi, lo) planes in the
// operand-fragment order of the chained forward kernel, 16 rows per group, every mesh padded to whole 64-row units.
int dn_sg_unit_rows(int K);                                         // rows per unit: 64 (K <= 128), 128 (K = 256) = one workgroup pass of the consuming kernel
bool dn_chain_sg_eligible(int C, int K, int with_grad, int hh);     // shapes chain_fwd_kernel<C, NW, HH, K / 32> is instantiated for
int dn_sg_units_host(const int* sizes, int n_mesh, int K, DnTile* out);    // out may be null: returns the unit count
size_t dn_sg_pack_elems(int n_units, int K);                        // uint4 elements of the packed operand
int dn_launch_sg_pack(const DnTile* units, int n_units, int n_mesh, int K, const float* evecs, const int* rowptr, const int* col, const float* vx,
                      const float* vy, float* gpx, float* gpy, float* amax4, uint4* out, hipStream_t stream);
// ys [n_mesh, K, C] -> transposed weight pieces [n_mesh][K / 32][2 (C / 16) 64] uint4 (fp16 hi / lo planes scaled by the power of two of the
// mesh's largest |ys|); ys_amax[2 n_mesh] receives those magnitudes and, behind them, every mesh's largest column 2-norm of ys
int dn_launch_spec_pieces(const float* ys, int n_mesh, int K, int C, uint4* out, float* ys_amax, hipStream_t stream);
// C = K = 256: xd, gx, gy = [Phi | G_X Phi | G_Y Phi] ys as a launch of its own (the fields of ChainArgs it reads: sg_*, ysp, ys_amax, xd_out, gx, gy,
// xd_amax_out, g_amax); the chained forward then runs in its MODE 2 form (dn_chain.hip)
int dn_launch_spectral_apply(const ChainArgs& a, int C, hipStream_t stream);

// ---------------------------------------------------------------------------------------
// one-launch learned-time diffusion, forward and backward (dn_diffuse.hip; K = C = 128)
// ---------------------------------------------------------------------------------------
#define DN_DF_MAX_GROUPS 4
#define DN_DF_MAX_SCHED 12
enum { DN_DF_OP_P1 = 1, DN_DF_OP_R = 2, DN_DF_OP_P3 = 3 };
enum { DN_DF_FLAG_DEFER = 1,       // an arrival is posted from inside the NEXT projection loop (its stores drain under that loop's loads)
       DN_DF_FLAG_SOLO_R = 2,      // tests: half of the workgroups pretend their poll for the partials ran out
       DN_DF_FLAG_SOLO_P3 = 4 };   // tests: the other half pretend their poll for the scaled spectrum ran out
struct DfLaunch {
    const DnTile* plan;            // device: [n_groups * n_wg], dn_diffuse_plan_host()
    int n_wg, n_groups, n_mesh;
    const float* evecs; const float* x; const float* mass; const float* evals; const float* time;
    float* xs;                     // forward: receives the unscaled spectrum (may be null); backward: the forward's spectrum (read)
    float* out; const float* add;  // backward: out = add + mass * (...) (add may be null)
    float* dt_part;                // backward: [dn_diffuse_dt_rows()][128]
    float* out_amax;               // optional: max |out| is accumulated into it (atomic max)
    void* ws;                      // dn_diffuse_ws_bytes()
    int bwd, order, flags;
    int split;                     // bit i: kernel boundary after schedule step i (0: one launch)
    long long acct_rows;           // host-side accounting only
};
size_t dn_diffuse_ws_bytes(int n_wg, int n_groups, int n_mesh);
int dn_diffuse_dt_rows(int n_wg, int n_groups);
int dn_diffuse_schedule(int G, int order, int* sched);
int dn_diffuse_plan_host(const int* sizes, int n_mesh, int n_wg, int n_groups, DnTile* plan);
int dn_launch_diffuse(const DfLaunch& L, hipStream_t stream);
int dn_launch_backproject(const DnTile* plan, int n_wg, const float* evecs, const float* ys, float* out, const float* add, const float* mass,
                          float* out_amax, double acct_rows, hipStream_t stream, int f16 = 0, const DnAmax* a_amax = nullptr, const DnAmax* b_amax = nullptr);
// dn_backproject_wide.hip: the forward back-projection at K = C = 256 (3-term engine, the spectrum streamed through an LDS-DMA ring)
bool dn_backproject_wide_ok(int K, int C, int n_tiles);
size_t dn_backproject_wide_ws_floats(int n_mesh, int K, int C);
int dn_launch_backproject_wide(const DnTile* tiles, int n_tiles, int n_mesh, const float* evecs, const float* ys, float* ws, float* out, float* out_amax,
                               int K, int C, double acct_rows, hipStream_t stream);
int dn_opt_chain_nw(void);   // option "chain_nw" (dn_api.hip): development override of the chained kernels' workgroup width

// ---------------------------------------------------------------------------------------
// small reductions / pointwise kernels (dn_pointwise.hip)
// ---------------------------------------------------------------------------------------
int dn_launch_spec_fwd(const float* partial, const int* mesh_chunk_off, const float* evals, const float* time,
                       float* xs, float* ys, int n_mesh, int K, int C, hipStream_t stream, float* ys_amax = nullptr);
// the spectral step of the diffusion backward in one launch (partials -> scaled spectrum + d_t contributions [dn_spec_bwd_dt_rows][C])
int dn_spec_bwd_dt_rows(int n_mesh, int K);
bool dn_spec_bwd_fused_ok(const float* partial, const float* time, const float* xs, const float* dys, const float* dt_part, int C);
int dn_launch_spec_bwd_fused(const float* partial, const int* mesh_chunk_off, const float* evals, const float* time, const float* xs, float* dys,
                             float* dt_part, int n_mesh, int K, int C, hipStream_t stream, float* dys_amax);
int dn_launch_spec_bwd(float* dys_inplace, const float* evals, const float* time, const float* xs, float* dt_part,
                       int n_mesh, int K, int C, hipStream_t stream, float* dys_amax = nullptr);
// max |x| of up to DN_AMAX_MAX_JOBS buffers in one launch (job j -> *dst[j], accumulated by atomic max: the caller zeroes the words);
// two buffers may share a word (e.g. A_re and A_im)
#define DN_AMAX_MAX_JOBS 12
struct AmaxJobs { const float* src[DN_AMAX_MAX_JOBS]; long long n[DN_AMAX_MAX_JOBS]; float* dst[DN_AMAX_MAX_JOBS]; int count;
    void push(const float* s, long long len, float* d) { if (s && d && len > 0 && count < DN_AMAX_MAX_JOBS) { src[count] = s; n[count] = len; dst[count] = d; ++count; } } };
int dn_launch_amax(const AmaxJobs& jobs, hipStream_t stream);
// one-launch start of a block call: stored maxima of small tensors (jobs with the same destination must be adjacent), zeroing of up to
// four word ranges, one word copy
struct AmaxInit { AmaxJobs jobs; int same[DN_AMAX_MAX_JOBS]; float* zero[4]; int zero_n[4]; int nzero; const float* copy_src; float* copy_dst;
    float* clamp_p; int clamp_n; float clamp_min;       // clamp_p[0..clamp_n) raised to >= clamp_min in place (the diffusion times, layers.py:48-49)
    void zero_range(float* p, int n) { if (p && n > 0 && nzero < 4) { zero[nzero] = p; zero_n[nzero] = n; ++nzero; } } };
int dn_launch_amax_init(const AmaxInit& a, hipStream_t stream);
int dn_launch_reduce(const float* partial, float* out, int n, long long stride, long long len, hipStream_t stream);
// out[s][i] = sum_{ch in [seg_off[s], seg_off[s+1])} partial[ch][i]  (seg_off == nullptr: one segment [0,n))
int dn_launch_seg_reduce(const float* partial, const int* seg_off, int nseg, int n, float* out, long long len, hipStream_t stream);
int dn_launch_combine_dA(const float* P, float* dA_re, float* dA_im, int C, hipStream_t stream);
int dn_launch_reduce_split(const float* partial, int n, float* o0, float* o1, long long half, hipStream_t stream);
// several whole-range sums in ONE launch (the weight / bias / rotation gradients of a block's backward are reduced together once
// their partials are all written): job j sums src[k][0..len) over k < n; element i goes to o0[i] (i < half) or o1[i - half].
#define DN_MR_MAX_JOBS 20
struct MrJob { const float* src; float* o0; float* o1; long long half, len; int n; };
struct MrJobs {
    MrJob j[DN_MR_MAX_JOBS];
    int count;
    bool push(const float* src, int n, long long len, float* o0, float* o1 = nullptr, long long half = -1) {
        if (count >= DN_MR_MAX_JOBS || len % 4 != 0 || ((uintptr_t)src & 15) != 0 || (o1 && half % 4 != 0)) return false;
        j[count++] = MrJob{src, o0, o1, o1 ? half : len, len, n};
        return true;
    }
};
int dn_launch_multi_reduce(const MrJobs& jobs, hipStream_t stream);
int dn_launch_thin_tn(const float* X, int M, const float* Y, int N, long long rows, int nm_major, int db_is_sx, float* dW, float* db,
                      float* ws_p, float* ws_s, int nblk, hipStream_t stream);
// dn_tn_da.hip: dA_re / dA_im partials of all four quadrants of [dd*gx | dd*gy]^T [gx | gy] from ONE pass over the three arrays (C = 128)
struct DaArgs {
    const float* dd; const float* gx; const float* gy;
    float* partial;                 // [gridDim.x][2][128][128]
    long long V;
    int rows_per_wg;                // multiple of 16
    int f16;                        // split-fp16 engine: A = dd * g bounded by a_amax (product of the two words), B = g by b_amax
    DnAmax a_amax, b_amax;
};
int dn_launch_tn_da(const float* dd, const float* gx, const float* gy, long long V, float* partial, int nwg, hipStream_t stream,
                    const float* dd_amax = nullptr, const float* g_amax = nullptr);
int dn_launch_mass_mean_fwd(const DnTile* meshrows, const float* mass, const float* x, float* out, float* msum,
                            int n_mesh, int C, hipStream_t stream);
int dn_launch_mass_mean_bwd(const DnTile* tiles, int ntiles, const float* mass, const float* msum, const float* dout,
                            float* dx, int C, hipStream_t stream);
int dn_launch_spmm(const SpArgs& s, hipStream_t stream);
// out[r, n] = (bias ? bias[n] : 0) + sum_{k<K} x[r,k] * (w_kn ? W[k*N+n] : W[n*K+k]),  K <= 32  (VALU, bandwidth-bound)
int dn_launch_smallk_rows(const float* x, int K, const float* W, int w_kn, const float* bias, int N, float* out,
                          long long rows, hipStream_t stream, float* o_amax = nullptr);
// partial-free small-N vertex contraction: out[m, n] = sum_r A[r,m] * B[r,n], N <= 16, via per-block partials in ws
int dn_launch_smalln_tn(const float* A, int M, const float* B, int N, long long rows, float* out, float* ws, int nblk,
                        hipStream_t stream);
// head / loss kernels (dn_head.hip)
struct HeadArgs {
    const float* x; int ldx;               // [n_src, C] logits (or log-probabilities when !lsm)
    const int* rowptr; const int* col;     // gather pattern of the outputs (null: output i = row i)
    float inv_div;                         // 1 / entries per output (mean)
    const long long* labels;               // [n_out] or null
    float smoothing;
    int n_out, n_src, C, lsm;
    float* logp;                           // [n_out, C] or null
    float* partial;                        // [2 * gridDim.x]: loss sums, valid counts
    // backward
    const int* t_rowptr; const int* t_col; // transposed gather (null: identity)
    const float* d_logp;                   // [n_out, C] or null
    const float* g_loss;                   // device scalar or null
    const float* count;                    // device scalar: valid rows of the forward
    float* d_x;                            // [n_src, C]
};
int dn_launch_head_fwd(const HeadArgs& a, int nb, float* loss, float* count, hipStream_t stream);
int dn_launch_head_bwd(const HeadArgs& a, hipStream_t stream);
int dn_launch_dtanh(const float* dg, const float* g, float* out, long long n, hipStream_t stream);
int dn_launch_reduce_pair(const float* pa, float* oa, long long la, const float* pb, float* ob, long long lb, int n, hipStream_t stream);
int dn_launch_hks(const float* evals, const float* evecs, const float* scales, int B, int V, int K, int S, long long scale_stride,
                  float* out, hipStream_t stream);
// COO -> CSR + CSR of the transpose (dn_pack.hip).  rows == nullptr: entry j belongs to row j / row_div (gather patterns)
size_t dn_pack_ws_bytes(long long nnz, int n_cols);
int dn_launch_coo_to_csr(const long long* rows, int row_div, const long long* cols, const float* vx, const float* vy, long long nnz, int n_rows,
                         int n_cols, int* rowptr, int* col32, int* t_rowptr, int* t_col, float* t_vx, float* t_vy, int* status, int* ws,
                         hipStream_t stream);
int dn_launch_checksum(const void* data, long long nwords, unsigned long long salt, unsigned long long* acc, hipStream_t stream);
#define DN_CK_MAX_JOBS 16
int dn_launch_checksum_multi(int n, const void* const* data, const long long* nwords, const unsigned long long* salts, unsigned long long* acc,
                             hipStream_t stream);
// launchers (host), defined in the .hip files; all return hipError_t as int
int dn_launch_rowgemm(const RgArgs& g, int ntiles, int nout, hipStream_t stream);
int dn_launch_tngemm(const TnArgs& g, int nchunks, hipStream_t stream);
int dn_launch_tngemm_multi(const TnArgs* gs, const int* nchunks, int count, hipStream_t stream);   // several products, one launch

// ---------------------------------------------------------------------------------------
// development build -DDN_CLK_TRACE (make variant TAG=clk EXTRA=-DDN_CLK_TRACE): thread 0 of the first 256 workgroups of a kernel stamps
// s_memtime (shader clock) and s_memrealtime (constant 100 MHz) at entry and exit; the ratio of the two differences IS the shader clock the
// kernel ran at (the "1.44 GHz under the 3-term MFMA stream" of round 5 was inferred from one trace of another kernel).
// One buffer and one reader per translation unit that uses it (tools/kbench --clk reads them).
// ---------------------------------------------------------------------------------------
#if defined(DN_CLK_TRACE) && !defined(DN_EMULATE)
#define DN_CLK_DECLARE(name_)                                                                                          \
    __device__ unsigned long long dn_clk_buf_##name_[256 * 4];                                                         \
    extern "C" int dn_debug_clk_read_##name_(unsigned long long* out, int n) {                                         \
        return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(dn_clk_buf_##name_), sizeof(unsigned long long) * (size_t)(n < 1024 ? n : 1024)); }
#define DN_CLK_STAMP(name_, i_)                                                                                        \
    do {                                                                                                               \
        if (threadIdx.x == 0 && blockIdx.x < 256 && blockIdx.y == 0 && blockIdx.z == 0) {                              \
            dn_clk_buf_##name_[blockIdx.x * 4 + 2 * (i_)] = __builtin_amdgcn_s_memtime();                              \
            dn_clk_buf_##name_[blockIdx.x * 4 + 2 * (i_) + 1] = __builtin_amdgcn_s_memrealtime();                      \
        }                                                                                                              \
    } while (0)
#else
#define DN_CLK_DECLARE(name_)
#define DN_CLK_STAMP(name_, i_) do {} while (0)
#endif

// ---------------------------------------------------------------------------------------
// opt-in per-kernel timing (bench.py's roofline leg): when enabled, every launch is bracketed by
// hipEvents on its own stream and summed per kernel family.  Off by default; compiled out of the
// emulator build.
// ---------------------------------------------------------------------------------------
enum { DN_K_ROWGEMM = 0, DN_K_ROWGEMM_DUAL = 1, DN_K_TNGEMM = 2, DN_K_SPMM = 3, DN_K_SMALL = 4, DN_K_CHAIN = 5, DN_K_CHAIN_BWD = 6, DN_K_DIFFUSE = 7, DN_K_TN_MULTI = 8, DN_K_TN_DA = 9, DN_K_BACKPROJECT = 10, DN_K_SPECTRAL = 11, DN_K_COUNT = 12 };   // one kind per KERNEL (rocprof name), except the small-kernel bucket
#ifdef DN_EMULATE
static inline void dn_prof_begin(int, hipStream_t) {}
static inline void dn_prof_end(int, hipStream_t, double, double) {}
#else
void dn_prof_begin(int kind, hipStream_t stream);
void dn_prof_end(int kind, hipStream_t stream, double flops, double bytes);
#endif
