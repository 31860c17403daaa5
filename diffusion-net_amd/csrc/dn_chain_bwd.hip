// dn_chain_bwd.hip -- the chained row pipeline of DiffusionNetBlock's BACKWARD (autograd of layers.py:217-239): from the gradient of the
// block output through the MiniMLP, the tanh features and the complex-linear gradient features down to d_gx / d_gy, ONE launch, the
// per-row gradients never leave the registers between stages.
//
//   Replaces, per block backward: the five input-gradient row GEMMs of the MiniMLP (d_a W_j with the ReLU' / dropout / residual / 1 - tanh^2
//   epilogues) and the two-output gradient-feature backward product -- six launches -- by one kernel that reads d_out and each saved
//   activation once and writes each gradient tensor once: d(pre-activations) d_a[j] (the weight-gradient products dW_j = d_a[j]^T h_{j-1}
//   stay separate split-V launches and read them), d_x (residual + x branch), d_xd, d_dots, d_gx, d_gy.
//
// Same decomposition as the forward (dn_chain.hip): a wave owns 32 rows as two 16-row halves on v_mfma_f32_16x16x32_f16, products
// transposed -- here D[k][m] = sum_n W[n][k] d_a[m][n]: the weight pieces hold W with rows and columns exchanged (chain_prep_kernel,
// `transposed`), the contraction runs over the layer's OUTPUT channels in the permuted slot order, and a product's accumulator layout is
// again the next product's operand layout.  Weight pieces stream through the same LDS-DMA ring.
// Arithmetic: two-term fp16 split (3 MFMAs per product), operands scaled by powers of two: d_out by its magnitude word, every later
// operand by the largest magnitude of the wave's own 32 x C tile.
#include "dn_chain_tiles.h"
#include <stdlib.h>

DN_CLK_DECLARE(chain_bwd)
template <int C, int NW, int HH>
__global__ __launch_bounds__(64 * NW) DN_WAVES_PER_EU(2) void chain_bwd_kernel(ChainBwdArgs a) {
    DN_CLK_STAMP(chain_bwd, 0);
    constexpr int NT = C / 16;
    constexpr int NK = C / 32;
    constexpr int NTHR = 64 * NW;
    constexpr int PIECE = 2 * NT * 64;
    constexpr int LPT = PIECE / NTHR;
    constexpr int RING = DN_CH_RING;
    constexpr int CH_PF = 1;              // (weight-fragment prefetch distance of the product macros, in tile pairs)
    static_assert(PIECE % NTHR == 0, "piece staging");
    static_assert(RING >= 2 && RING <= 8 && (RING - 1) * LPT < 60, "ring depth vs the vmcnt range");

    DN_DYN_SMEM(smem_raw);
    uint4* ring = reinterpret_cast<uint4*>(smem_raw);
#ifdef DN_EMULATE
    const unsigned lds0 = 0;
#else
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem_raw;
#endif
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = lane & 15, q = lane >> 4;

    const int GX = gridDim.x >> 3;
    const int per_x = (a.units + 7) >> 3;
    const int xcd = blockIdx.x & 7, slot0 = blockIdx.x >> 3;
    int npass = 0;
    if (slot0 < per_x) {
        int hi_local = a.units - xcd * per_x;
        hi_local = hi_local > per_x ? per_x : hi_local;
        if (slot0 < hi_local) npass = (hi_local - slot0 + GX - 1) / GX;
    }
    if (npass == 0) return;

    const float s_do = ch_uniform(dn_pow2_scale(dn_amax_word(a.d_out_amax)));
    float sw_inv[DN_CH_LAYERS];
#pragma unroll
    for (int j = 0; j < DN_CH_LAYERS; ++j) sw_inv[j] = j < a.n_mlp ? ch_uniform(ch_pow2_inv(dn_pow2_scale(dn_amax_word(a.w_amax[j])))) : 1.f;
    const float swa_inv = a.with_grad ? ch_uniform(ch_pow2_inv(dn_pow2_scale(dn_amax_word(a.wa_amax)))) : 1.f;

    // ---- the piece stream (see dn_chain.hip)
    const uint4* src_piece = a.wp;
    const uint4* const src_end = a.wp + (size_t)a.n_pieces * PIECE;
#ifdef DN_EMULATE
    const int wave_u = wave;
#else
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
#endif
    int rq = 0;
    auto issue = [&]() {
#pragma unroll
        for (int i = 0; i < LPT; ++i) {
            const int e0 = rq * PIECE + i * NTHR + wave_u * 64;
            ch_dma16(src_piece + i * NTHR + tid, ring + e0, lds0 + 16u * (unsigned)e0);
        }
        src_piece = src_piece + PIECE == src_end ? a.wp : src_piece + PIECE;
        rq = rq + 1 == RING ? 0 : rq + 1;
    };
#ifdef DN_EMULATE
#define CH_WAIT_PIECES(n) do {} while (0)
#define CH_BARRIER() __syncthreads()
#else
#define CH_WAIT_PIECES(n) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"((n) * LPT) : "memory")
#define CH_BARRIER() do { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)
#endif
    int gp = 0;
#pragma unroll
    for (int i = 0; i < RING - 1; ++i) issue();
    CH_WAIT_PIECES(RING - 2);
    CH_BARRIER();
#define CH_PIECE_BEGIN()                                                                                                \
    const uint4* ws_ = ring + (gp % RING) * PIECE;                                                                     \
    issue()
#define CH_PIECE_END() do { CH_WAIT_PIECES(RING - 2); CH_BARRIER(); ++gp; } while (0)
#define CH_ZERO(ACC)                                                                                                    \
    _Pragma("unroll") for (int h_ = 0; h_ < HH; ++h_)                                                                   \
        _Pragma("unroll") for (int n_ = 0; n_ < NT; ++n_) ACC[h_][n_] = dn_f32x4{0.f, 0.f, 0.f, 0.f}
    // accumulator tiles (now holding fp32 values) -> operand fragments of the next product, split with scale S_
#define CH_PACK(ACC, S_, FH, FL)                                                                                        \
    _Pragma("unroll") for (int h_ = 0; h_ < HH; ++h_)                                                                   \
        _Pragma("unroll") for (int T_ = 0; T_ < NK; ++T_) {                                                             \
            const float va_[4] = {ACC[h_][2 * T_][0], ACC[h_][2 * T_][1], ACC[h_][2 * T_][2], ACC[h_][2 * T_][3]};       \
            const float vb_[4] = {ACC[h_][2 * T_ + 1][0], ACC[h_][2 * T_ + 1][1], ACC[h_][2 * T_ + 1][2], ACC[h_][2 * T_ + 1][3]}; \
            ch_split8(va_, vb_, S_, FH[h_][T_], FL[h_][T_]);                                                            \
        }

    for (int pass = 0; pass < npass; ++pass) {
        const int unit = xcd * per_x + slot0 + pass * GX;
        const int rb = unit * (16 * HH * NW) + 16 * HH * wave;
        int rowh[HH]; bool liveh[HH]; int rch[HH];
#pragma unroll
        for (int hh = 0; hh < HH; ++hh) {
            rowh[hh] = rb + 16 * hh + m;
            liveh[hh] = rowh[hh] < a.V;
            rch[hh] = liveh[hh] ? rowh[hh] : a.V - 1;
        }
        // ---- d_out -> operand fragments
        uint4 fh[HH][NK], fl[HH][NK];
        {
            float4 v[HH][NT];
#pragma unroll
            for (int hh = 0; hh < HH; ++hh) {
                const float* p = a.d_out + (long long)rch[hh] * C + 4 * q;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) v[hh][nt] = *reinterpret_cast<const float4*>(p + 16 * nt);   // (d_out is read again for the residual)
            }
#pragma unroll
            for (int hh = 0; hh < HH; ++hh)
#pragma unroll
                for (int T = 0; T < NK; ++T) ch_split8(v[hh][2 * T], v[hh][2 * T + 1], s_do, fh[hh][T], fl[hh][T]);
        }
        float s_act = s_do;
        dn_f32x4 acc[HH][NT];
        // ---- hidden layers, last to first: d_a[j-1] = (d_a[j] W_j) * relu'(h_{j-1}) * dropout scale   (h > 0 <=> kept and active)
#pragma unroll 1
        for (int j = a.n_mlp - 1; j >= 1; --j) {
            CH_ZERO(acc);
#pragma unroll
            for (int T = 0; T < NK; ++T) {
                CH_PIECE_BEGIN();
                CH_MMA2(acc, fh[0][T], fl[0][T], fh[HH - 1][T], fl[HH - 1][T]);
                CH_PIECE_END();
            }
            const float so = ch_pow2_inv(s_act) * (j == 1 ? sw_inv[1] : (j == 2 ? sw_inv[2] : sw_inv[3]));
            const float* hp = a.h[j - 1];
            const float ds = j == 1 ? a.dscale[0] : (j == 2 ? a.dscale[1] : a.dscale[2]);
            float4 hv[HH][NT];
#pragma unroll
            for (int hh = 0; hh < HH; ++hh) {
                const float* p = hp + (long long)rch[hh] * C + 4 * q;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) hv[hh][nt] = *reinterpret_cast<const float4*>(p + 16 * nt);   // (d_out is read again for the residual)
            }
            float wm = 0.f;
#pragma unroll
            for (int hh = 0; hh < HH; ++hh)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const float hq[4] = {hv[hh][nt].x, hv[hh][nt].y, hv[hh][nt].z, hv[hh][nt].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float t = hq[e] > 0.f ? acc[hh][nt][e] * so * ds : 0.f;
                        acc[hh][nt][e] = t;
                        wm = fabsf(t) > wm ? fabsf(t) : wm;
                    }
                }
            float* dj = a.d_a[j - 1];
#pragma unroll
            for (int hh = 0; hh < HH; ++hh)
                ch_st_tiles<NT>(dj, C, rowh[hh], a.V, m, q, [&](const int nt) { return make_float4(acc[hh][nt][0], acc[hh][nt][1], acc[hh][nt][2], acc[hh][nt][3]); });
            s_act = ch_uniform(dn_pow2_scale(ch_wave_max(wm)));
            CH_PACK(acc, s_act, fh, fl);
        }
        // ---- layer 0: d_a[0] W_0 split into its column groups [x | xd | g]   (fh / fl hold d_a[0])
        const float so0 = ch_pow2_inv(s_act) * sw_inv[0];
        {   // x group + residual: d_xacc = d_out + d_a0 W_0[:, :C]
            CH_ZERO(acc);
#pragma unroll
            for (int T = 0; T < NK; ++T) {
                CH_PIECE_BEGIN();
                CH_MMA2(acc, fh[0][T], fl[0][T], fh[HH - 1][T], fl[HH - 1][T]);
                CH_PIECE_END();
            }
            float4 r4[HH][NT];
#pragma unroll
            for (int hh = 0; hh < HH; ++hh) {
                const float* p = a.d_out + (long long)rch[hh] * C + 4 * q;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) r4[hh][nt] = *reinterpret_cast<const float4*>(p + 16 * nt);
            }
#pragma unroll
            for (int hh = 0; hh < HH; ++hh)
                ch_st_tiles<NT>(a.d_xacc, C, rowh[hh], a.V, m, q, [&](const int nt) {
                    return make_float4(acc[hh][nt][0] * so0 + r4[hh][nt].x, acc[hh][nt][1] * so0 + r4[hh][nt].y,
                                       acc[hh][nt][2] * so0 + r4[hh][nt].z, acc[hh][nt][3] * so0 + r4[hh][nt].w); });
        }
        {   // xd group
            CH_ZERO(acc);
#pragma unroll
            for (int T = 0; T < NK; ++T) {
                CH_PIECE_BEGIN();
                CH_MMA2(acc, fh[0][T], fl[0][T], fh[HH - 1][T], fl[HH - 1][T]);
                CH_PIECE_END();
            }
#pragma unroll
            for (int hh = 0; hh < HH; ++hh)
                ch_st_tiles<NT>(a.d_xd, C, rowh[hh], a.V, m, q, [&](const int nt) {
                    return make_float4(acc[hh][nt][0] * so0, acc[hh][nt][1] * so0, acc[hh][nt][2] * so0, acc[hh][nt][3] * so0); });
        }
        if (a.with_grad) {
            // g group: d_dots = (d_a0 W_0[:, 2C:]) * (1 - g^2)   (kept in the accumulator registers for the gradient-feature stage)
            CH_ZERO(acc);
#pragma unroll
            for (int T = 0; T < NK; ++T) {
                CH_PIECE_BEGIN();
                CH_MMA2(acc, fh[0][T], fl[0][T], fh[HH - 1][T], fl[HH - 1][T]);
                CH_PIECE_END();
            }
            {
                float4 gq[HH][NT];
#pragma unroll
                for (int hh = 0; hh < HH; ++hh) {
                    const float* p = a.g + (long long)rch[hh] * C + 4 * q;
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) gq[hh][nt] = ch_ld4(p + 16 * nt);
                }
#pragma unroll
                for (int hh = 0; hh < HH; ++hh)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        acc[hh][nt][0] = acc[hh][nt][0] * so0 * (1.f - gq[hh][nt].x * gq[hh][nt].x);
                        acc[hh][nt][1] = acc[hh][nt][1] * so0 * (1.f - gq[hh][nt].y * gq[hh][nt].y);
                        acc[hh][nt][2] = acc[hh][nt][2] * so0 * (1.f - gq[hh][nt].z * gq[hh][nt].z);
                        acc[hh][nt][3] = acc[hh][nt][3] * so0 * (1.f - gq[hh][nt].w * gq[hh][nt].w);
                    }
#pragma unroll
                for (int hh = 0; hh < HH; ++hh)
                    ch_st_tiles<NT>(a.d_dots, C, rowh[hh], a.V, m, q, [&](const int nt) { return make_float4(acc[hh][nt][0], acc[hh][nt][1], acc[hh][nt][2], acc[hh][nt][3]); });
            }
            // ---- gradient features backward, one 16-row half at a time:
            //      d_gx = d_dots * Bre + (d_dots gx) A_re + (d_dots gy) A_im ;  d_gy = d_dots * Bim - (d_dots gx) A_im + (d_dots gy) A_re
            //      (without rotations: d_gx = d_dots * Bre + (d_dots gx) A, d_gy = d_dots * Bim + (d_dots gy) A)
            auto half = [&](const int hh) __attribute__((always_inline)) {      // (hh is a literal in each of the two copies below: no row is selected or moved)
                const int row = hh ? rowh[HH - 1] : rowh[0];
                const int rc = hh ? rch[HH - 1] : rch[0];
                const bool live = hh ? liveh[HH - 1] : liveh[0];
                float dd[NT][4];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int e = 0; e < 4; ++e) dd[nt][e] = hh ? acc[HH - 1][nt][e] : acc[0][nt][e];
                float u[NT][4], v[NT][4];
                float wm = 0.f;
                {
                    const float* px = a.gx + (long long)rc * C + 4 * q;
                    const float* py = a.gy + (long long)rc * C + 4 * q;
                    float4 tx[NT], ty[NT];
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) { tx[nt] = ch_ld4(px + 16 * nt); ty[nt] = ch_ld4(py + 16 * nt); }
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        u[nt][0] = dd[nt][0] * tx[nt].x; u[nt][1] = dd[nt][1] * tx[nt].y; u[nt][2] = dd[nt][2] * tx[nt].z; u[nt][3] = dd[nt][3] * tx[nt].w;
                        v[nt][0] = dd[nt][0] * ty[nt].x; v[nt][1] = dd[nt][1] * ty[nt].y; v[nt][2] = dd[nt][2] * ty[nt].z; v[nt][3] = dd[nt][3] * ty[nt].w;
#pragma unroll
                        for (int e = 0; e < 4; ++e) { wm = fabsf(u[nt][e]) > wm ? fabsf(u[nt][e]) : wm; wm = fabsf(v[nt][e]) > wm ? fabsf(v[nt][e]) : wm; }
                    }
                }
                const float s_uv = ch_uniform(dn_pow2_scale(ch_wave_max(wm)));
                dn_f32x4 ag[2][NT];       // [0] = d_gx, [1] = d_gy of this half
#pragma unroll
                for (int o = 0; o < 2; ++o)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) ag[o][nt] = dn_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int T = 0; T < NK; ++T) {
                    uint4 uh, ul, vh, vl;
                    ch_split8(u[2 * T], u[2 * T + 1], s_uv, uh, ul);
                    ch_split8(v[2 * T], v[2 * T + 1], s_uv, vh, vl);
                    {   // A_re (or A): d_gx += u A, d_gy += v A
                        CH_PIECE_BEGIN();
                        CH_MMA2_LEAN(ag, uh, ul, vh, vl);
                        CH_PIECE_END();
                    }
                    if (a.with_rot) {   // A_im: d_gx += v A_im, d_gy -= u A_im
                        const uint4 nuh = make_uint4(uh.x ^ 0x80008000u, uh.y ^ 0x80008000u, uh.z ^ 0x80008000u, uh.w ^ 0x80008000u);
                        const uint4 nul = make_uint4(ul.x ^ 0x80008000u, ul.y ^ 0x80008000u, ul.z ^ 0x80008000u, ul.w ^ 0x80008000u);
                        CH_PIECE_BEGIN();
                        CH_MMA2_LEAN(ag, vh, vl, nuh, nul);
                        CH_PIECE_END();
                    }
                }
                const float sog = ch_pow2_inv(s_uv) * swa_inv;
                {
                    const float* pr = a.bre + (long long)rc * C + 4 * q;
                    const float* pi = a.bim + (long long)rc * C + 4 * q;
                    float4 br[NT], bi[NT];
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) { br[nt] = ch_ld4(pr + 16 * nt); bi[nt] = ch_ld4(pi + 16 * nt); }
                    ch_st_tiles<NT>(a.d_gx, C, row, a.V, m, q, [&](const int nt) {
                        return make_float4(ag[0][nt][0] * sog + dd[nt][0] * br[nt].x, ag[0][nt][1] * sog + dd[nt][1] * br[nt].y,
                                           ag[0][nt][2] * sog + dd[nt][2] * br[nt].z, ag[0][nt][3] * sog + dd[nt][3] * br[nt].w); });
                    ch_st_tiles<NT>(a.d_gy, C, row, a.V, m, q, [&](const int nt) {
                        return make_float4(ag[1][nt][0] * sog + dd[nt][0] * bi[nt].x, ag[1][nt][1] * sog + dd[nt][1] * bi[nt].y,
                                           ag[1][nt][2] * sog + dd[nt][2] * bi[nt].z, ag[1][nt][3] * sog + dd[nt][3] * bi[nt].w); });
                }
            };
            half(0);
            if constexpr (HH > 1) half(HH - 1);
        }
    }
#ifndef DN_EMULATE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the requests that ran past the end of the stream
#endif
    DN_CLK_STAMP(chain_bwd, 1);
#undef CH_PACK
#undef CH_ZERO
#undef CH_PIECE_END
#undef CH_PIECE_BEGIN
#undef CH_BARRIER
#undef CH_WAIT_PIECES
}

// ---------------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------------
int dn_chain_bwd_pieces(int C, int with_grad, int with_rot, int n_mlp) {
    const int NK = C / 32;
    return (n_mlp - 1) * NK + (with_grad ? 3 : 2) * NK + (with_grad ? 2 * NK * (with_rot ? 2 : 1) : 0);
}

template <int C, int NW, int HH>
static int chain_bwd_launch_nw(ChainBwdArgs a, hipStream_t stream) {
    a.units = (a.V + 16 * HH * NW - 1) / (16 * HH * NW);
    int g = (8 / NW) * dn_num_cus();
    if (g > a.units) g = a.units;
    g = (g + 7) / 8 * 8;
    const size_t smem = (size_t)DN_CH_RING * (2 * (C / 16) * 64) * sizeof(uint4);
#ifndef DN_EMULATE
    static unsigned long long lds_opt_in = 0;   // per-device bitmap
    { const int oe_ = dn_lds_opt_in(reinterpret_cast<const void*>(&chain_bwd_kernel<C, NW, HH>), smem, &lds_opt_in); if (oe_) return oe_; }
#endif
    DN_LAUNCH((chain_bwd_kernel<C, NW, HH>), dim3(g, 1, 1), dim3(64 * NW, 1, 1), smem, stream, a);
    return (int)hipGetLastError();
}
template <int C>
static int chain_bwd_launch(int npieces, const ChainBwdArgs& a_in, hipStream_t stream, int hh) {
    const int nw_env = dn_opt_chain_nw();   // (development override; see dn_chain.hip for the choice of NW and HH)
    int nw = nw_env;
    if (nw != 1 && nw != 2 && nw != 4) {
        const int half = dn_num_cus() / 2;
        nw = 4;
        while (nw > 1 && (a_in.V + 16 * hh * nw - 1) / (16 * hh * nw) < half) nw >>= 1;
    }
    ChainBwdArgs a = a_in;
    a.n_pieces = npieces;
    if (hh == 1) return nw == 4 ? chain_bwd_launch_nw<C, 4, 1>(a, stream) : (nw == 2 ? chain_bwd_launch_nw<C, 2, 1>(a, stream) : chain_bwd_launch_nw<C, 1, 1>(a, stream));
    switch (nw) {
        case 2: return chain_bwd_launch_nw<C, 2, 2>(a, stream);
        case 1: return chain_bwd_launch_nw<C, 1, 2>(a, stream);
        default: return chain_bwd_launch_nw<C, 4, 2>(a, stream);
    }
}

int dn_launch_chain_bwd(int npieces, const ChainBwdArgs& a, int C, hipStream_t stream, int hh) {
    if (npieces > DN_CH_MAX_PIECES || npieces <= 0 || (hh != 1 && hh != 2)) return 1;
    dn_prof_begin(DN_K_CHAIN_BWD, stream);
    int err;
    if (C == 128) err = chain_bwd_launch<128>(npieces, a, stream, hh);
    else if (C == 64) err = chain_bwd_launch<64>(npieces, a, stream, hh);
    else err = 1;
    {
        // algorithmic traffic: d_out read once (+ once more for the residual: L2), every saved activation read once, every gradient written once
        const double VC = 4.0 * (double)a.V * C;
        const double nr = 1.0 + (a.n_mlp - 1) + (a.with_grad ? 5.0 : 0.0);
        const double nw = (a.n_mlp - 1) + 2.0 + (a.with_grad ? 3.0 : 0.0);
        const double prod = (a.n_mlp - 1) + (a.with_grad ? 3.0 : 2.0) + (a.with_grad ? (a.with_rot ? 4.0 : 2.0) : 0.0);
        dn_prof_end(DN_K_CHAIN_BWD, stream, 2.0 * (double)a.V * C * C * prod, VC * (nr + nw));
    }
    return err;
}
