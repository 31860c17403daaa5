// dn_direct_tiles.h -- "direct" row product for K = 128 (the back-projection phase of the one-launch diffusion operator, dn_diffuse.hip): out[r, 0..127] = sum_k A[r, k] B[k, n] with the long operand A read straight
// from global memory into MFMA fragments (no LDS round trip, no barrier) and the 128 x 128 operand B split once per workgroup into
// fragment-ordered bf16 planes resident in LDS.  Developed and measured as tools/experiments/rowgemm_direct (see its README).
//   * unit = 16 rows x 128 columns per wave, eight 16x16 accumulators, v_mfma_f32_16x16x32_bf16 with the operands swapped
//     (D^T = B^T A^T): a lane's four registers of a tile are four consecutive output columns of one row -> float4 stores;
//   * A fragment of a lane = eight k-consecutive floats of one row = two float4 loads; k is permuted inside a 32-float line (the
//     same permutation on both operands): lane group g of a k32 step holds floats {4g..4g+3} and {16+4g..+3} of the line, so each
//     load instruction covers 64 contiguous bytes per row;
//   * B fragments: ring of two groups (tile pairs), a plane's registers refilled right after its last use in a group (product order
//     hi*lo | mid*mid, hi*mid | lo*hi, mid*hi, hi*hi retires B's lo plane after 2 MFMAs, mid after 6, hi after 12);
//   * A registers: two sets (units i, i+1 of the wave), a pair refilled with the same step two units ahead as soon as it is split.
#pragma once
#include "dn_gemm_tiles.h"

#define DN_RD_WAVES 8
#define DN_RD_ROWS 16
#define DN_RD_LDS_B (4 * 8 * 3 * 1024)   // [k32 step][16-column tile][plane][lane] x 16 B

typedef float f32x4 __attribute__((ext_vector_type(4)));
// One 16x16x32 bf16 MFMA step (fp32 accumulate): lane l supplies eight consecutive-k bf16 of row l&15 of the first operand and
// of column l&15 of the second, k = 8*(l>>4) .. +7; accumulator register r of lane l is D[4*(l>>4) + r][l&15].
__device__ __forceinline__ f32x4 dn_mfma_bf16_16(uint4 a, uint4 b, f32x4 c) {
#ifdef DN_EMULATE
    return dnemu_mfma_f32_16x16x32_bf16(a, b, c);
#else
    typedef __bf16 dn_bf16x8 __attribute__((ext_vector_type(8)));
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(dn_bf16x8, a), __builtin_bit_cast(dn_bf16x8, b), c, 0, 0, 0);
#endif
}
// the same step on fp16 operands (2-term split-fp16 engine)
__device__ __forceinline__ f32x4 dn_mfma_f16_16(uint4 a, uint4 b, f32x4 c) {
#ifdef DN_EMULATE
    return dnemu_mfma_f32_16x16x32_f16(a, b, c);
#else
    typedef _Float16 dn_f16x8 __attribute__((ext_vector_type(8)));
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(dn_f16x8, a), __builtin_bit_cast(dn_f16x8, b), c, 0, 0, 0);
#endif
}
#ifdef DN_EMULATE
#define DN_SCHED_FENCE() do {} while (0)
#define DN_UNIFORM(x) (x)
#else
#define DN_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#define DN_UNIFORM(x) __builtin_amdgcn_readfirstlane(x)   // wave-uniform value the compiler cannot prove uniform -> SGPR
#endif

// eight consecutive-k values -> NP planes of eight 16-bit terms each.  NP = 3: bf16 (hi, mid, lo), sc ignored; NP = 2: fp16 (hi, lo) of v * sc
template <int NP>
__device__ __forceinline__ void rd_split8(const float4& u, const float4& v, uint4 (&pl)[NP], float sc) {
    unsigned w[4][NP];
    dn_split_pair<NP>(u.x, u.y, sc, w[0]);
    dn_split_pair<NP>(u.z, u.w, sc, w[1]);
    dn_split_pair<NP>(v.x, v.y, sc, w[2]);
    dn_split_pair<NP>(v.z, v.w, sc, w[3]);
#pragma unroll
    for (int p = 0; p < NP; ++p) pl[p] = make_uint4(w[0][p], w[1][p], w[2][p], w[3][p]);
}
// first physical k of slot group h (0: slots 0-3, 1: slots 4-7) of lane group lg (0..3) in k32 step s
__device__ __forceinline__ int rd_k0(int s, int lg, int h) { return 32 * s + 16 * h + 4 * lg; }

// Split a 128 x 128 operand B[k][n] (row-major, n contiguous, leading dimension ldb) into the fragment-ordered planes: a thread
// fetches an 8 (k) x 4 (n) block as eight float4 along n -- the eight k of four items (four consecutive lanes).  NTHR threads.
// COH: B was written by other workgroups of the SAME launch (write-through stores): L1-bypassing loads.
template <int NTHR, bool COH = false, int NP = 3>
__device__ __forceinline__ void rd_stage_b_nn(const float* bp, int ldb, unsigned char* sB, int tid, float sc = 1.f) {
#pragma unroll
    for (int h = 0; h < 512 / NTHR; ++h) {                     // 4 steps x 4 lane groups x 32 column quads = 512 blocks
        const int b = tid + NTHR * h;
        const int nq = b & 31, lg = (b >> 5) & 3, s = b >> 7;
        const float* cp = bp + 4 * nq;
        float4 r[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float* p0 = cp + (long long)(rd_k0(s, lg, 0) + j) * ldb;
            const float* p1 = cp + (long long)(rd_k0(s, lg, 1) + j) * ldb;
            r[j] = COH ? dn_ld4_coherent(p0) : *reinterpret_cast<const float4*>(p0);
            r[4 + j] = COH ? dn_ld4_coherent(p1) : *reinterpret_cast<const float4*>(p1);
        }
        const int t = nq >> 2;                                   // 16-column tile of columns 4 nq .. 4 nq + 3
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float4 u = make_float4(dn_f4_get(r[0], c), dn_f4_get(r[1], c), dn_f4_get(r[2], c), dn_f4_get(r[3], c));
            const float4 v = make_float4(dn_f4_get(r[4], c), dn_f4_get(r[5], c), dn_f4_get(r[6], c), dn_f4_get(r[7], c));
            uint4 pl[NP];
            rd_split8<NP>(u, v, pl, sc);
            const int lane = 16 * lg + ((4 * nq + c) & 15);
            unsigned char* dst = sB + (((s * 8 + t) * NP) * 64 + lane) * 16;
#pragma unroll
            for (int p = 0; p < NP; ++p) *reinterpret_cast<uint4*>(dst + 1024 * p) = pl[p];
        }
    }
}

struct RdUnit { int row0, nrows; };   // up to 16 consecutive rows: one wave's unit of work
// unit j of the contiguous rows [rs, re): pure arithmetic; past the end the last unit is returned (a harmless prefetch target)
__device__ __forceinline__ RdUnit rd_unit(int rs, int re, int j) {
    const int nu = (re - rs + DN_RD_ROWS - 1) / DN_RD_ROWS;
    j = j < nu ? j : nu - 1;
    RdUnit r;
    r.row0 = rs + DN_RD_ROWS * j;
    r.nrows = re - r.row0 < DN_RD_ROWS ? re - r.row0 : DN_RD_ROWS;
    return r;
}
// this lane's row of the unit (clamped to the unit's first row past its end: feeds outputs that are never stored)
__device__ __forceinline__ const float* rd_row_ptr(const float* ap, int ald, const RdUnit& un, int li, int lg) {
    return ap + (long long)(un.row0 + (li < un.nrows ? li : 0)) * ald + 4 * lg;
}
// B fragments of plane p of group G = 4 s + pr of a unit (k32 step s, column tiles 2 pr and 2 pr + 1): two ds_read_b128
template <int NP>
__device__ __forceinline__ void rd_read_plane(const unsigned char* sB, int lane, int G, int p, uint4 (&F)[NP][2]) {
    const int s = (G & 15) >> 2, pr = G & 3;
#pragma unroll
    for (int e = 0; e < 2; ++e)
        F[p][e] = *reinterpret_cast<const uint4*>(sB + (((s * 8 + 2 * pr + e) * NP + p) * 64 + lane) * 16);
}

// One unit.  MODE: DN_EPI_STORE (out = acc) or DN_EPI_MASS_ADD (out = r0 + rowv[row] * acc, r0 optional) through pt_piece_store.
// X: this unit's A registers, Y: the next unit's (their first pair is split here, for the next call); np_x / np_y: this lane's row of
// the units two ahead of X's / Y's owner.  `a`: planes of the current k32 step (carried across units), F: the B-fragment ring.
template <int MODE, int NP>
__device__ __forceinline__ void rd_unit_body(const RgArgs& g, const unsigned char* sB, const RdUnit& cur, const float* np_x,
                                             const float* np_y, int n0, int lane, float4 (&X)[8], float4 (&Y)[8], uint4 (&a)[NP],
                                             uint4 (&F)[2][NP][2], float sa, float so, float& om) {
    const int li = lane & 15, lg = lane >> 4;
    // auxiliary operands of the epilogue: requested first (vmcnt is an in-order counter: waiting for them later must not drain the
    // prefetch loads issued during the MFMA steps)
    const bool row_ok = li < cur.nrows;
    const long long grow = cur.row0 + (row_ok ? li : 0);
    const int col0 = n0 + 4 * lg;                              // this lane's columns: col0 + 16 t .. +3
    PtPiece P[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        P[t].off = grow * g.ldo + col0 + 16 * t;
        P[t].ok = row_ok;
        P[t].a0 = dn_f4_zero();
        if (MODE == DN_EPI_MASS_ADD && g.r0 != nullptr) P[t].a0 = *reinterpret_cast<const float4*>(g.r0 + grow * g.ldr + col0 + 16 * t);
        if (MODE == DN_EPI_MASS_ADD) P[t].rs = g.rowv[grow];
    }
    f32x4 acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[t][r] = 0.f;
    DN_SCHED_FENCE();
    // (A plane, B plane) per product, smallest terms first.  NP = 3: B's lo (2) is used by product 0 only, mid (1) by 1-2, hi (0) by 3-5;
    // NP = 2 (split-fp16): hi*lo, lo*hi, hi*hi -- B's lo (1) by product 0, hi (0) by 1-2.  A plane's registers are refilled right after its last use.
    constexpr int NPROD = NP == 3 ? 6 : 3;
    constexpr int PA[6] = {0, 1, NP == 3 ? 0 : 0, 2, 1, 0}, PB[6] = {NP == 3 ? 2 : 1, NP == 3 ? 1 : 0, NP == 3 ? 1 : 0, 0, 0, 0};
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        uint4 an[NP];
#pragma unroll
        for (int pr = 0; pr < 4; ++pr) {
            const int G = 4 * s + pr;
            uint4 (&Fg)[NP][2] = F[G & 1];
#define RD_MMA(p_)                                                                                                      \
    _Pragma("unroll") for (int e = 0; e < 2; ++e) {                                                                     \
        if constexpr (NP == 3) acc[2 * pr + e] = dn_mfma_bf16_16(Fg[PB[p_]][e], a[PA[p_]], acc[2 * pr + e]);            \
        else acc[2 * pr + e] = dn_mfma_f16_16(Fg[PB[p_]][e], a[PA[p_]], acc[2 * pr + e]);   /* operands swapped: D[n][row] */ \
    }
            RD_MMA(0)
            DN_SCHED_FENCE();
            rd_read_plane<NP>(sB, lane, G + 2, NP - 1, Fg);
            if (pr == 2) {                                     // next step's planes, and the consumed registers' refill
                if (s < 3) {
                    rd_split8<NP>(X[2 * s + 2], X[2 * s + 3], an, sa);
                    X[2 * s + 2] = *reinterpret_cast<const float4*>(np_x + 32 * (s + 1));
                    X[2 * s + 3] = *reinterpret_cast<const float4*>(np_x + 32 * (s + 1) + 16);
                } else {
                    rd_split8<NP>(Y[0], Y[1], an, sa);
                    Y[0] = *reinterpret_cast<const float4*>(np_y);
                    Y[1] = *reinterpret_cast<const float4*>(np_y + 16);
                }
            }
            RD_MMA(1)
            RD_MMA(2)
            DN_SCHED_FENCE();
            if constexpr (NP == 3) {
                rd_read_plane<NP>(sB, lane, G + 2, 1, Fg);
                RD_MMA(3)
                RD_MMA(4)
                RD_MMA(5)
                DN_SCHED_FENCE();
            }
            rd_read_plane<NP>(sB, lane, G + 2, 0, Fg);
#undef RD_MMA
        }
#pragma unroll
        for (int p = 0; p < NP; ++p) a[p] = an[p];
    }
    (void)NPROD;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        P[t].v = make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]);
        if (NP == 2) P[t].v = dn_f4_scale(P[t].v, so);                 // exact power-of-two rescale of the split-fp16 product
        om = dn_f4_amax(om, pt_piece_store<MODE, false>(g, P[t]));     // om: running max |stored value| of this lane
    }
}

// All 16-row units of the contiguous rows [rs, re) for one wave of an eight-wave workgroup, in two parts: rd_rows_begin requests the
// wave's first two units (no dependence on B: a kernel may issue it BEFORE it stages the B planes), rd_rows_run does the rest (B planes
// staged in sB).  Returns this lane's max |stored value|.
struct RdStart { float4 A0[8], A1[8]; RdUnit c0, c1; int j, nu; };
__device__ __forceinline__ void rd_rows_begin(const float* ap, int ald, int rs, int re, int wave, int lane, RdStart& S) {
    const int li = lane & 15, lg = lane >> 4;
    S.nu = (re - rs + DN_RD_ROWS - 1) / DN_RD_ROWS;
    S.j = wave;
    if (S.j >= S.nu) return;
    S.c0 = rd_unit(rs, re, S.j);
    S.c1 = rd_unit(rs, re, S.j + DN_RD_WAVES);
    const float* p0 = rd_row_ptr(ap, ald, S.c0, li, lg);
    const float* p1 = rd_row_ptr(ap, ald, S.c1, li, lg);
#pragma unroll
    for (int i = 0; i < 8; ++i) S.A0[i] = *reinterpret_cast<const float4*>(p0 + 16 * i);
#pragma unroll
    for (int i = 0; i < 8; ++i) S.A1[i] = *reinterpret_cast<const float4*>(p1 + 16 * i);
}
template <int MODE, int NP = 3>
__device__ __forceinline__ float rd_rows_run(const RgArgs& g, const unsigned char* sB, const float* ap, int ald, int rs, int re, int n0,
                                             int lane, RdStart& S, float sa = 1.f, float so = 1.f) {
    const int li = lane & 15, lg = lane >> 4;
    const int nu = S.nu;
    int j = S.j;
    float om = 0.f;
    if (j >= nu) return om;
    float4 (&A0)[8] = S.A0;
    float4 (&A1)[8] = S.A1;
    RdUnit c0 = S.c0, c1 = S.c1;
    uint4 a[NP], F[2][NP][2];
    rd_split8<NP>(A0[0], A0[1], a, sa);
    {
        const float* p2 = rd_row_ptr(ap, ald, rd_unit(rs, re, j + 2 * DN_RD_WAVES), li, lg);
        A0[0] = *reinterpret_cast<const float4*>(p2);
        A0[1] = *reinterpret_cast<const float4*>(p2 + 16);
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        rd_read_plane<NP>(sB, lane, 0, p, F[0]);
        rd_read_plane<NP>(sB, lane, 1, p, F[1]);
    }
    for (; j < nu; j += 2 * DN_RD_WAVES) {
        {
            const RdUnit n2 = rd_unit(rs, re, j + 2 * DN_RD_WAVES), n3 = rd_unit(rs, re, j + 3 * DN_RD_WAVES);
            rd_unit_body<MODE, NP>(g, sB, c0, rd_row_ptr(ap, ald, n2, li, lg), rd_row_ptr(ap, ald, n3, li, lg), n0, lane, A0, A1, a, F, sa, so, om);
            c0 = n2;
        }
        if (j + DN_RD_WAVES < nu) {                            // wave-uniform
            const RdUnit n3 = rd_unit(rs, re, j + 3 * DN_RD_WAVES), n4 = rd_unit(rs, re, j + 4 * DN_RD_WAVES);
            rd_unit_body<MODE, NP>(g, sB, c1, rd_row_ptr(ap, ald, n3, li, lg), rd_row_ptr(ap, ald, n4, li, lg), n0, lane, A1, A0, a, F, sa, so, om);
            c1 = n3;
        }
    }
    return om;
}
template <int MODE>
__device__ __forceinline__ float rd_run_rows(const RgArgs& g, const unsigned char* sB, const float* ap, int ald, int rs, int re, int n0,
                                             int wave, int lane) {
    RdStart S;
    rd_rows_begin(ap, ald, rs, re, wave, lane, S);
    return rd_rows_run<MODE>(g, sB, ap, ald, rs, re, n0, lane, S);
}
