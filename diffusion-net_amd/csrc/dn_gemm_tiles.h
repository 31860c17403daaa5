// dn_gemm_tiles.h -- device-side building blocks shared by the row-GEMM kernels (dn_rowgemm.hip, dn_rowgemm_persist.hip):
// the compile-time epilogue of one accumulator tile, the branch-free slice loaders, the exact-f32 LDS staging / MFMA step, and
// the split-bf16 ("x3") staging halves (split -> planes, planes -> LDS) and operand fetch / MFMA halves.
#pragma once
#include "dn_common.h"


// =======================================================================================
// rowgemm
// =======================================================================================
// Epilogue of one 32x32 accumulator tile.  MODE is a compile-time constant, all auxiliary operands of the 16
// elements a lane owns are fetched first (from clamped, always-valid addresses -> no branches between the
// loads), then combined and stored under the validity predicate.
// dropout seed of this launch: the host part plus the optional device word (a captured graph advances it per replay).  Read once per
// kernel, only by the instances that can draw a mask; the by-value argument block itself stays untouched (modifying it makes the
// compiler materialise a private copy of all of RgArgs: measured 51 -> 72 us on the C->C product).
__device__ __forceinline__ unsigned long long rg_seed(const RgArgs& g) {
    unsigned long long s = g.rng_seed;
    if (s && g.rng_seed_dev) s += *g.rng_seed_dev;
    return s;
}

template <int MODE, int NOUT>
__device__ __forceinline__ void rg_epilogue_tile(const RgArgs& g, unsigned long long seed, int row_base, int rows_valid, int col, bool col_ok,
                                                 int lane, const f32x16& a0, const f32x16& a1) {
    bool ok[16];
    long long io[16], ir[16];
    const int cc = col_ok ? col : 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int rl = dn_acc_row(r, lane);
        ok[r] = col_ok && rl < rows_valid;
        const long long rr = row_base + (ok[r] ? rl : 0);
        io[r] = rr * g.ldo + cc;
        ir[r] = rr * g.ldr + cc;
    }
    float v0[16], v1[16], v2[16], res0[16], res1[16];
    constexpr bool need_r0 = MODE == DN_EPI_BIAS_RESID || MODE == DN_EPI_GRADFEAT || MODE == DN_EPI_MUL_DFAC ||
                             MODE == DN_EPI_ADD || MODE == DN_EPI_DTANH || MODE == DN_EPI_GRADFEAT_BWD ||
                             MODE == DN_EPI_MASS_ADD;
    constexpr bool need_r1 = MODE == DN_EPI_GRADFEAT || MODE == DN_EPI_GRADFEAT_BWD;
    constexpr bool need_r2 = MODE == DN_EPI_GRADFEAT_BWD;
    const bool has_r0 = g.r0 != nullptr;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        v0[r] = (need_r0 && has_r0) ? g.r0[ir[r]] : 0.f;
        v1[r] = need_r1 ? g.r1[ir[r]] : 0.f;
        v2[r] = need_r2 ? g.r2[ir[r]] : 0.f;
    }
    float bias = 0.f;
    if (MODE == DN_EPI_STORE || MODE == DN_EPI_BIAS_RELU || MODE == DN_EPI_BIAS_RESID) bias = g.bias ? g.bias[cc] : 0.f;
    if (MODE == DN_EPI_BIAS_RELU) {
        if (g.mask) {
#pragma unroll
            for (int r = 0; r < 16; ++r) v1[r] = g.mask[ir[r]] ? g.scale : 0.f;
        } else if (seed) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long long rr = row_base + (ok[r] ? dn_acc_row(r, lane) : 0);
                v1[r] = ((dn_keep_bits(seed, rr, cc >> 2, (g.N + 3) >> 2) >> (cc & 3)) & 1u) ? g.scale : 0.f;
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) v1[r] = 1.f;
        }
    }
    if (MODE == DN_EPI_MASS_ADD) {
#pragma unroll
        for (int r = 0; r < 16; ++r) v1[r] = g.rowv[row_base + (ok[r] ? dn_acc_row(r, lane) : 0)];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float x0 = a0[r], x1 = NOUT == 2 ? a1[r] : 0.f;
        float y0 = 0.f, y1 = 0.f;
        if (MODE == DN_EPI_STORE) y0 = x0 + bias;
        else if (MODE == DN_EPI_BIAS_RELU) { float h = x0 + bias; y0 = (h > 0.f ? h : 0.f) * v1[r]; }
        else if (MODE == DN_EPI_BIAS_RESID) y0 = (x0 + bias) + v0[r];
        else if (MODE == DN_EPI_GRADFEAT) y0 = tanhf(v0[r] * x0 + v1[r] * x1);
        else if (MODE == DN_EPI_MUL_DFAC) y0 = v0[r] > 0.f ? x0 * g.scale : 0.f;
        else if (MODE == DN_EPI_ADD) y0 = x0 + v0[r];
        else if (MODE == DN_EPI_DTANH) y0 = x0 * (1.f - v0[r] * v0[r]);
        else if (MODE == DN_EPI_GRADFEAT_BWD) { y0 = x0 + v0[r] * v1[r]; y1 = x1 + v0[r] * v2[r]; }
        else if (MODE == DN_EPI_MASS_ADD) y0 = v0[r] + v1[r] * x0;
        res0[r] = y0;
        res1[r] = y1;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r)
        if (ok[r]) g.o0[io[r]] = res0[r];
    if (g.o_amax) {   // the split-fp16 consumers of o0 need its largest magnitude
        float m = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { const float a = fabsf(res0[r]); m = (ok[r] && a > m) ? a : m; }
        dn_amax_commit<true>(g.o_amax, m);
    }
    if (MODE == DN_EPI_GRADFEAT_BWD) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (ok[r]) g.o1[io[r]] = res1[r];
    }
    if (MODE == DN_EPI_GRADFEAT && g.o1) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (ok[r]) { g.o1[io[r]] = a0[r]; g.o2[io[r]] = a1[r]; }
    }
}

// ---- staging helpers -------------------------------------------------------------------------------------
// ALIGNED fast path: no branch sits between the loads (row / column guards are applied by clamping the address to
// a valid one and zeroing the value afterwards), so every global_load of a slice is in flight at once and the
// wait lands at the LDS store after the MFMAs of the previous slice.
// A staged slice in registers: raw loaded values plus the factors applied when it is written to LDS.  Nothing
// here is *used* before the MFMAs of the previous slice have been issued, so the loads stay in flight under them.
template <int NOUT, int A_IT, int B_IT>
struct RgRegs {
    float4 a[A_IT];
    float4 q[A_IT];          // optional elementwise factor of A (valid when has_q)
    float am[A_IT];          // row guard as 0/1 factor
    float4 b[NOUT][B_IT];
    float bm[NOUT][B_IT];    // column guard * sign
};

// PAIRK (bf16x3 path, row-contraction B, TN = 128): a thread fetches rows 2p and 2p+1 (p = tid & 15) of a 4-column group
// q4 = (tid >> 4) + (NTHR / 16) * h, h < B_IT / 2, so that it can write packed (k, k+1) bf16 pairs of the transposed B planes.
template <int TN, int NTHR, int NOUT, bool ALIGNED, bool BCOLK, bool HASQ, int A_IT, int B_IT, bool PAIRK = false>
__device__ __forceinline__ void rg_load(const RgArgs& g, const DnTile& tile, int n0, int seg, int koff, int tid,
                                        RgRegs<NOUT, A_IT, B_IT>& R) {   // tile.row0/nrows may describe a sub-tile
    const RgSeg sg = g.a[seg];
    if (ALIGNED) {
        long long off[A_IT];
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const int idx = tid + i * NTHR;
            const int row = idx >> 3, q = idx & 7;
            const bool rok = row < tile.nrows;
            R.am[i] = rok ? 1.f : 0.f;
            off[i] = (long long)(tile.row0 + (rok ? row : 0)) * sg.ld + koff + 4 * q;
            R.a[i] = *reinterpret_cast<const float4*>(sg.p + off[i]);
        }
        if (HASQ) {   // compile-time; loads only, no use
#pragma unroll
            for (int i = 0; i < A_IT; ++i) R.q[i] = *reinterpret_cast<const float4*>(sg.q + off[i]);
        }
#pragma unroll
        for (int o = 0; o < NOUT; ++o) {
            const float* bp = g.b[o][seg] + (long long)tile.mesh * g.b_mesh_stride;
            const float sgn = g.bsign[o][seg];
#pragma unroll
            for (int i = 0; i < B_IT; ++i) {
                const int idx = tid + i * NTHR;
                long long boff;
                bool nok;
                if (BCOLK) {
                    const int nrow = idx >> 3, q = idx & 7;
                    nok = n0 + nrow < g.N;
                    boff = (long long)(nok ? n0 + nrow : 0) * g.ldb + koff + 4 * q;
                } else {
                    const int krow = PAIRK ? 2 * (tid & 15) + (i & 1) : idx / (TN / 4);
                    const int q4 = PAIRK ? (tid >> 4) + (NTHR / 16) * (i >> 1) : idx % (TN / 4);
                    nok = n0 + 4 * q4 < g.N;
                    boff = (long long)(koff + krow) * g.ldb + (nok ? n0 + 4 * q4 : 0);
                }
                R.b[o][i] = *reinterpret_cast<const float4*>(bp + boff);
                R.bm[o][i] = nok ? sgn : 0.f;
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const int idx = tid + i * NTHR;
            const int row = idx >> 3, q = idx & 7;
            const long long base = (long long)(tile.row0 + row) * sg.ld + koff + 4 * q;
            float e[4] = {0.f, 0.f, 0.f, 0.f};
            if (row < tile.nrows) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if (koff + 4 * q + c < sg.w) {
                        e[c] = sg.p[base + c];
                        if (sg.q) e[c] *= sg.q[base + c];
                    }
                }
            }
            R.a[i] = make_float4(e[0], e[1], e[2], e[3]);
            R.am[i] = 1.f;
            if (HASQ) R.q[i] = make_float4(1.f, 1.f, 1.f, 1.f);   // already folded in above
        }
#pragma unroll
        for (int o = 0; o < NOUT; ++o) {
            const float* bp = g.b[o][seg] + (long long)tile.mesh * g.b_mesh_stride;
#pragma unroll
            for (int i = 0; i < B_IT; ++i) {
                const int idx = tid + i * NTHR;
                float e[4] = {0.f, 0.f, 0.f, 0.f};
                if (BCOLK) {
                    const int nrow = idx >> 3, q = idx & 7;
                    const long long base = (long long)(n0 + nrow) * g.ldb + koff + 4 * q;
                    if (n0 + nrow < g.N) {
#pragma unroll
                        for (int c = 0; c < 4; ++c)
                            if (koff + 4 * q + c < sg.w) e[c] = bp[base + c];
                    }
                } else {
                    const int krow = idx / (TN / 4), q4 = idx % (TN / 4);
                    const long long base = (long long)(koff + krow) * g.ldb + n0 + 4 * q4;
                    if (koff + krow < sg.w) {
#pragma unroll
                        for (int c = 0; c < 4; ++c)
                            if (n0 + 4 * q4 + c < g.N) e[c] = bp[base + c];
                    }
                }
                R.b[o][i] = make_float4(e[0], e[1], e[2], e[3]);
                R.bm[o][i] = g.bsign[o][seg];
            }
        }
    }
}

template <int TN, int NTHR, int NOUT, bool BCOLK, bool HASQ, int A_IT, int B_IT>
__device__ __forceinline__ void rg_store(float* sA, float* sB, int tid, const RgRegs<NOUT, A_IT, B_IT>& R) {
    constexpr int SB = DN_KB * TN;
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const int idx = tid + i * NTHR;
        float4 v = dn_f4_scale(R.a[i], R.am[i]);
        if (HASQ) v = dn_f4_mul(v, R.q[i]);
        *reinterpret_cast<float4*>(&sA[dn_colk_off(idx >> 3, idx & 7)]) = v;
    }
#pragma unroll
    for (int o = 0; o < NOUT; ++o)
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            const int idx = tid + i * NTHR;
            const float4 v = dn_f4_scale(R.b[o][i], R.bm[o][i]);
            if (BCOLK)
                *reinterpret_cast<float4*>(&sB[o * SB + dn_colk_off(idx >> 3, idx & 7)]) = v;
            else
                *reinterpret_cast<float4*>(&sB[o * SB + 4 * idx]) = v;
        }
}

template <int TN, int MT, int NT, int NOUT, bool BCOLK>
__device__ __forceinline__ void rg_compute(const float* sA, const float* sB, int arow0, int bcol0, int li, int ls,
                                           f32x16 (&acc)[NOUT][MT][NT]) {
    constexpr int SB = DN_KB * TN;
#pragma unroll
    for (int kg = 0; kg < 4; ++kg) {
        float4 af[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
            af[mt] = *reinterpret_cast<const float4*>(&sA[dn_colk_off(arow0 + mt * 32 + li, 2 * kg + ls)]);
        float bv[NOUT][NT][4];
        if (BCOLK) {
#pragma unroll
            for (int o = 0; o < NOUT; ++o)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const float4 t4 = *reinterpret_cast<const float4*>(&sB[o * SB + dn_colk_off(bcol0 + nt * 32 + li, 2 * kg + ls)]);
                    bv[o][nt][0] = t4.x; bv[o][nt][1] = t4.y; bv[o][nt][2] = t4.z; bv[o][nt][3] = t4.w;
                }
        } else {
#pragma unroll
            for (int o = 0; o < NOUT; ++o)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        bv[o][nt][t] = sB[o * SB + (8 * kg + 4 * ls + t) * TN + bcol0 + nt * 32 + li];
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int o = 0; o < NOUT; ++o) {
                        acc[o][mt][nt] = dn_mfma(dn_f4_get(af[mt], t), bv[o][nt][t], acc[o][mt][nt]);
                    }
    }
}

// ---- split-bf16 ("x3") staging and MFMA for the persistent and the two-output kernels: both operands live in LDS as three bf16 planes
//      (hi, mid, lo) of [rows][32 k]; six cross products per k16 step replace sixteen f32 MFMA k2 steps.
// The staging is written as two halves so that a kernel can put the MFMAs of the current slice between them: rg_split_x3
// is pure VALU on the prefetched registers, rg_put_x3 only writes LDS (the compiler must keep LDS writes behind earlier LDS
// reads of the other buffer -- it cannot prove they do not alias -- so the reads are issued first, see rg_frag_x3).
template <int NOUT, int A_IT, int B_IT, int NP = 3>
struct X3Planes {             // NP planes: 3 = split-bf16 (hi, mid, lo), 2 = split-fp16 (hi, lo) of the pre-scaled operand
    uint2 a[A_IT][NP];
    uint2 b[NOUT][B_IT][NP];  // BCOLK: one 8-byte chunk per float4; PAIRK: dword e of column group h is (i = 2h + (e >> 1), .x/.y = e & 1)
};

template <int NOUT, bool BCOLK, bool HASQ, int A_IT, int B_IT, bool DO_A = true, bool DO_B = true, int NP = 3>
__device__ __forceinline__ void rg_split_x3(const RgRegs<NOUT, A_IT, B_IT>& R, X3Planes<NOUT, A_IT, B_IT, NP>& P, float sa = 1.f, float sb = 1.f) {
    if (DO_A) {
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            // no row mask here: a row past the unit's end (its address was clamped) only feeds its own, never stored, output row
            float4 v = R.a[i];
            if (HASQ) v = dn_f4_mul(v, R.q[i]);
            dn_split_f4<NP>(v, sa, P.a[i]);
        }
    }
    if (!DO_B) return;
#pragma unroll
    for (int o = 0; o < NOUT; ++o) {
        if (BCOLK) {
#pragma unroll
            for (int i = 0; i < B_IT; ++i) {
                // the factor carries the sign of a two-output product; with one output it is the column mask only, and
                // a column past N (clamped address) only feeds its own, never stored, output column
                const float4 v = NOUT == 2 ? dn_f4_scale(R.b[o][i], R.bm[o][i]) : R.b[o][i];
                dn_split_f4<NP>(v, sb, P.b[o][i]);
            }
        } else {   // PAIRK: R.b[o][2h] = row 2p, R.b[o][2h+1] = row 2p+1 of column group q4(h) -> packed (k, k+1) dwords per column
            static_assert(BCOLK || B_IT % 2 == 0, "pair mapping needs two rows per thread and column group");
#pragma unroll
            for (int h = 0; h < B_IT / 2; ++h) {
                const float4 v0 = NOUT == 2 ? dn_f4_scale(R.b[o][2 * h], R.bm[o][2 * h]) : R.b[o][2 * h];
                const float4 v1 = NOUT == 2 ? dn_f4_scale(R.b[o][2 * h + 1], R.bm[o][2 * h + 1]) : R.b[o][2 * h + 1];
                unsigned wx[NP], wy[NP], wz[NP], ww[NP];
                dn_split_pair<NP>(v0.x, v1.x, sb, wx);
                dn_split_pair<NP>(v0.y, v1.y, sb, wy);
                dn_split_pair<NP>(v0.z, v1.z, sb, wz);
                dn_split_pair<NP>(v0.w, v1.w, sb, ww);
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    P.b[o][2 * h][p].x = wx[p]; P.b[o][2 * h][p].y = wy[p];
                    P.b[o][2 * h + 1][p].x = wz[p]; P.b[o][2 * h + 1][p].y = ww[p];
                }
            }
        }
    }
}

template <int NTHR, int NOUT, bool BCOLK, int A_IT, int B_IT, int NP = 3, bool DO_A = true, bool DO_B = true>
__device__ __forceinline__ void rg_put_x3(unsigned char* sA, unsigned char* sB, int tid, const X3Planes<NOUT, A_IT, B_IT, NP>& P) {
    constexpr int PL = DN_TM * 64;    // bytes per A plane (128 rows x 32 bf16)
    constexpr int PLB = 128 * 64;     // bytes per B plane (128 output columns); output o uses planes [3o, 3o+3)
#pragma unroll
    for (int i = 0; i < (DO_A ? A_IT : 0); ++i) {
        const int idx = tid + i * NTHR;
        const int row = idx >> 3, q = idx & 7;
        const int off = dn_plane_off(row, q >> 1) + (q & 1) * 8;
#pragma unroll
        for (int p = 0; p < NP; ++p) *reinterpret_cast<uint2*>(sA + p * PL + off) = P.a[i][p];
    }
#pragma unroll
    for (int o = 0; o < (DO_B ? NOUT : 0); ++o) {
        unsigned char* sBo = sB + o * 3 * PLB;
        if (BCOLK) {
#pragma unroll
            for (int i = 0; i < B_IT; ++i) {
                const int idx = tid + i * NTHR;
                const int nrow = idx >> 3, q = idx & 7;
                const int off = dn_plane_off(nrow, q >> 1) + (q & 1) * 8;
#pragma unroll
                for (int p = 0; p < NP; ++p) *reinterpret_cast<uint2*>(sBo + p * PLB + off) = P.b[o][i][p];
            }
        } else {
            const int pr = tid & 15;
#pragma unroll
            for (int h = 0; h < B_IT / 2; ++h) {
                const int q4 = (tid >> 4) + (NTHR / 16) * h;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int off = dn_plane_off(4 * q4 + e, pr >> 2) + (pr & 3) * 4;
#pragma unroll
                    for (int p = 0; p < NP; ++p) {
                        const uint2 w = P.b[o][2 * h + (e >> 1)][p];
                        *reinterpret_cast<unsigned*>(sBo + p * PLB + off) = (e & 1) ? w.y : w.x;
                    }
                }
            }
        }
    }
}

template <int NTHR, int NOUT, bool BCOLK, bool HASQ, int A_IT, int B_IT, int NP = 3>
__device__ __forceinline__ void rg_store_x3(unsigned char* sA, unsigned char* sB, int tid, const RgRegs<NOUT, A_IT, B_IT>& R,
                                            float sa = 1.f, float sb = 1.f) {
    X3Planes<NOUT, A_IT, B_IT, NP> P;
    rg_split_x3<NOUT, BCOLK, HASQ, A_IT, B_IT, true, true, NP>(R, P, sa, sb);
    rg_put_x3<NTHR, NOUT, BCOLK, A_IT, B_IT, NP>(sA, sB, tid, P);
}

// MFMA operands of one 32-wide slice (two k16 steps; lane group lg owns k = 16 s + 8 lg .. +7), read in one burst
template <int MT, int NT, int NOUT, int NP = 3>
struct X3Frags {
    uint4 a[2][NP][MT];
    uint4 b[2][NOUT][NP][NT];
};

template <int MT, int NT, int NOUT, int NP = 3>
__device__ __forceinline__ void rg_frag_x3(const unsigned char* sA, const unsigned char* sB, int arow0, int bcol0, int li, int lg,
                                           int s, X3Frags<MT, NT, NOUT, NP>& F) {
    constexpr int PL = DN_TM * 64, PLB = 128 * 64;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
            F.a[s][p][mt] = *reinterpret_cast<const uint4*>(sA + p * PL + dn_plane_off(arow0 + mt * 32 + li, 2 * s + lg));
#pragma unroll
        for (int o = 0; o < NOUT; ++o)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                F.b[s][o][p][nt] = *reinterpret_cast<const uint4*>(sB + (o * 3 + p) * PLB + dn_plane_off(bcol0 + nt * 32 + li, 2 * s + lg));
    }
}

// The cross products of one k16 step, smallest terms first (NP = 3, split-bf16: mid*mid, hi*lo, lo*hi, hi*mid, mid*hi, hi*hi; NP = 2,
// split-fp16: hi*lo, lo*hi, hi*hi).  Product-major issue order: consecutive MFMAs go to DIFFERENT accumulators (an MFMA on the
// accumulator of the previous one waits out its full latency), while every accumulator still receives its products in the same
// order -- bitwise the same sums as an accumulator-major loop.
template <int MT, int NT, int NOUT, int NP = 3>
__device__ __forceinline__ void rg_mma_x3(const X3Frags<MT, NT, NOUT, NP>& F, int s, f32x16 (&acc)[NOUT][MT][NT]) {
    constexpr int NPROD = NP == 3 ? 6 : 3;
    constexpr int PA[6] = {NP == 3 ? 1 : 0, NP == 3 ? 0 : 1, NP == 3 ? 2 : 0, 0, 1, 0};
    constexpr int PB[6] = {1, NP == 3 ? 2 : 0, 0, 1, 0, 0};
#pragma unroll
    for (int p = 0; p < NPROD; ++p)
#pragma unroll
        for (int o = 0; o < NOUT; ++o)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    if constexpr (NP == 3) acc[o][mt][nt] = dn_mfma_bf16(F.a[s][PA[p]][mt], F.b[s][o][PB[p]][nt], acc[o][mt][nt]);
                    else acc[o][mt][nt] = dn_mfma_f16(F.a[s][PA[p]][mt], F.b[s][o][PB[p]][nt], acc[o][mt][nt]);
                }
}

template <int MT, int NT, int NOUT, int NP = 3>
__device__ __forceinline__ void rg_compute_x3(const unsigned char* sA, const unsigned char* sB, int arow0, int bcol0, int li,
                                              int lg, f32x16 (&acc)[NOUT][MT][NT]) {
    X3Frags<MT, NT, NOUT, NP> F;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        rg_frag_x3<MT, NT, NOUT, NP>(sA, sB, arow0, bcol0, li, lg, s, F);
        rg_mma_x3<MT, NT, NOUT, NP>(F, s, acc);
    }
}

// ---- float4 "piece" epilogue shared by the persistent kernels (parked accumulators) and the direct kernel (accumulator registers):
//      FLAG: STORE -> a bias vector is added; BIAS_RELU -> a dropout keep-mask is applied; unused otherwise.
struct PtPiece {
    float4 v, a0, bias;
    uint32_t mk;
    float rs;
    long long off;
    bool ok;
};

// phase 2 (after the MFMAs): epilogue maths + one coalesced float4 store
template <int MODE, bool FLAG>
__device__ __forceinline__ float4 pt_piece_store(const RgArgs& g, const PtPiece& P) {
    constexpr bool need_bias = (MODE == DN_EPI_STORE && FLAG) || MODE == DN_EPI_BIAS_RELU || MODE == DN_EPI_BIAS_RESID;
    float x[4] = {P.v.x, P.v.y, P.v.z, P.v.w};
    if (need_bias) { x[0] += P.bias.x; x[1] += P.bias.y; x[2] += P.bias.z; x[3] += P.bias.w; }
    const float r[4] = {P.a0.x, P.a0.y, P.a0.z, P.a0.w};
    float y[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (MODE == DN_EPI_STORE) y[e] = x[e];
        else if (MODE == DN_EPI_BIAS_RELU) {
            const float h = x[e] > 0.f ? x[e] : 0.f;
            y[e] = FLAG ? (((P.mk >> (8 * e)) & 0xffu) ? h * g.scale : 0.f) : h;
        }
        else if (MODE == DN_EPI_BIAS_RESID) y[e] = x[e] + r[e];
        else if (MODE == DN_EPI_MUL_DFAC) y[e] = r[e] > 0.f ? x[e] * g.scale : 0.f;
        else if (MODE == DN_EPI_ADD) y[e] = x[e] + r[e];
        else if (MODE == DN_EPI_DTANH) y[e] = x[e] * (1.f - r[e] * r[e]);
        else if (MODE == DN_EPI_MASS_ADD) y[e] = r[e] + P.rs * x[e];
        else y[e] = x[e];
    }
    const float4 yv = make_float4(y[0], y[1], y[2], y[3]);
    if (P.ok) *reinterpret_cast<float4*>(g.o0 + P.off) = yv;
    return P.ok ? yv : dn_f4_zero();     // what was stored (zeros for a dead piece): the caller may track max |o0|
}

