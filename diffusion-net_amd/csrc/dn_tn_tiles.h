// dn_tn_tiles.h -- device-side building blocks of the split-bf16 "TN" product (contraction over the vertex axis): staging of a
// 32-row step into k-major bf16 planes and the transpose-read MFMA step.  Shared by tngemm_x3_kernel (dn_tngemm.hip) and the
// fused diffusion kernel (dn_diffusion_fused.hip).
#pragma once
#include "dn_common.h"

enum { DN_TN_PLAIN = 0, DN_TN_ROWSCALE = 1, DN_TN_COLSUM = 2, DN_TN_QA = 3 };

#define DN_TX_THREADS 512
#define DN_TX_ROWB 320                      // bytes per LDS plane row (128 bf16 + 32 B pad)
#define DN_TX_PLANE (DN_KB * DN_TX_ROWB)    // bytes per plane (10 KiB)

struct TxRegs {
    float4 a[2], b[2], qa[2];
    float ma[2], mb[2];
};

template <int FLAVOR>
__device__ __forceinline__ void tx_load(const TnArgs& g, const DnTile& ch, int step, int kr0, bool a_ok, bool b_ok,
                                        const float* ap, const float* aq, int ald, const float* bp, int bld, TxRegs& R) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int kr = step * DN_KB + kr0 + 16 * i;
        const bool kok = kr < ch.nrows;
        const long long row = (long long)ch.row0 + (kok ? kr : 0);
        R.ma[i] = (kok && a_ok) ? 1.f : 0.f;
        R.mb[i] = (kok && b_ok) ? 1.f : 0.f;
        R.a[i] = *reinterpret_cast<const float4*>(ap + row * ald);
        R.b[i] = *reinterpret_cast<const float4*>(bp + row * bld);
        if (FLAVOR == DN_TN_QA) R.qa[i] = *reinterpret_cast<const float4*>(aq + row * ald);
        if (FLAVOR == DN_TN_ROWSCALE) R.qa[i].x = g.b_rowscale[row];
    }
}

template <int NP = 3>
__device__ __forceinline__ void tx_put(unsigned char* planes, int off, float4 v, float s = 1.f) {
    uint2 pl[NP];
    dn_split_f4<NP>(v, s, pl);
#pragma unroll
    for (int p = 0; p < NP; ++p) *reinterpret_cast<uint2*>(planes + p * DN_TX_PLANE + off) = pl[p];
}

template <int FLAVOR, int NP = 3>
__device__ __forceinline__ void tx_store(unsigned char* sA, unsigned char* sB, int kr0, int q, const TxRegs& R, float4& csum,
                                         float sa = 1.f, float sb = 1.f) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        float4 va = dn_f4_scale(R.a[i], R.ma[i]);
        const float4 vb = dn_f4_scale(R.b[i], FLAVOR == DN_TN_ROWSCALE ? R.mb[i] * R.qa[i].x : R.mb[i]);
        if (FLAVOR == DN_TN_QA) va = dn_f4_mul(va, R.qa[i]);
        if (FLAVOR == DN_TN_COLSUM) { csum.x += va.x; csum.y += va.y; csum.z += va.z; csum.w += va.w; }
        const int off = (kr0 + 16 * i) * DN_TX_ROWB + q * 8;
        tx_put<NP>(sA, off, va, sa);
        tx_put<NP>(sB, off, vb, sb);
    }
}

// one MFMA operand (8 consecutive k of a column) from a k-major plane: two transpose reads of 4 rows each
__device__ __forceinline__ uint4 tx_frag(const unsigned char* p) {
    const uint2 lo = dn_lds_tr16(p), hi = dn_lds_tr16(p + 4 * DN_TX_ROWB);
    return make_uint4(lo.x, lo.y, hi.x, hi.y);
}

template <int NP = 3>
__device__ __forceinline__ void tx_compute(const unsigned char* sA, const unsigned char* sB, int wr, int wc, int lane,
                                           f32x16 (&acc)[2]) {
    // lane -> chunk it names inside its 16-lane group: row (c/4) of the 4-row block, columns 4*(c%4)..+3 of the 16-column half
    const int g = lane >> 4, c = lane & 15;
    const int lane_off = (8 * (g >> 1) + (c >> 2)) * DN_TX_ROWB + (16 * (g & 1) + 4 * (c & 3)) * 2;
#pragma unroll
    for (int s = 0; s < 2; ++s) {   // two k16 steps per 32-row step
        uint4 a[NP][2], b[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int base = p * DN_TX_PLANE + s * 16 * DN_TX_ROWB + lane_off;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) a[p][mt] = tx_frag(sA + base + (wr * 64 + mt * 32) * 2);
            b[p] = tx_frag(sB + base + (wc * 32) * 2);
        }
        // product-major: consecutive MFMAs alternate between the two accumulators (same per-accumulator order, same sums)
        // NP = 3: mid*mid, hi*lo, lo*hi, hi*mid, mid*hi, hi*hi;  NP = 2 (split-fp16): hi*lo, lo*hi, hi*hi
        constexpr int NPROD = NP == 3 ? 6 : 3;
        constexpr int PA[6] = {NP == 3 ? 1 : 0, NP == 3 ? 0 : 1, NP == 3 ? 2 : 0, 0, 1, 0};
        constexpr int PB[6] = {1, NP == 3 ? 2 : 0, 0, 1, 0, 0};
#pragma unroll
        for (int p = 0; p < NPROD; ++p)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                if constexpr (NP == 3) acc[mt] = dn_mfma_bf16(a[PA[p]][mt], b[PB[p]], acc[mt]);
                else acc[mt] = dn_mfma_f16(a[PA[p]][mt], b[PB[p]], acc[mt]);
            }
    }
}

