// dn_tn_da.hip -- gradients of the gradient-rotation matrices (layers.py:117-130; SURVEY 3.4 dA_re, dA_im), C = 128:
//
//     dA_re = (dd*gx)^T gx + (dd*gy)^T gy          dA_im = (dd*gy)^T gx - (dd*gx)^T gy          (sums over ALL vertices)
//
// i.e. the four 128 x 128 quadrants of [dd*gx | dd*gy]^T [gx | gy].  The generic split-V kernel (dn_tngemm.hip) runs the four
// quadrants as four workgroups that each stream three of the arrays again (12 array passes, 2.4x the algorithmic bytes at the
// HBM counters, 141 us).  Here ONE workgroup owns a contiguous row range and computes the whole 256 x 256 tile set from ONE pass
// over dd, gx, gy: every 16-row step is loaded once (3 float4 per thread), turned into the four operands, split into bf16 planes
// once, and feeds 8 x 48 MFMAs; the quadrants are combined in LDS before the partial is written, so a partial is 2 x 128 x 128
// floats and the separate combine launch is gone.
//
// Layout: 512 threads = 8 waves as 2 (operand-A halves: dd*gx | dd*gy) x 4 (64-column strips of [gx | gy]); a wave owns a
// 128 x 64 output tile = 8 accumulator tiles (128 registers, 2 waves per SIMD).  LDS: two step buffers of 6 planes
// (A hi/mid/lo, B hi/mid/lo), each plane 16 rows x 256 bf16 with rows padded to 576 B (== 64 mod 256: the transpose reads of a
// half-wave touch 4 rows x 64 B = all 64 banks once); one barrier per step, the staging of step s+1 shares the iteration with the
// MFMAs of step s.  Register prefetch ring of two steps (the loads of step s+3 are issued while s is multiplied).
#include "dn_common.h"
#include <string.h>

#define DN_DA_THREADS 512
#define DN_DA_KS 16                              // rows per step (one MFMA k-step)
#define DN_DA_ROWB 576                           // bytes per plane row: 256 bf16 + 64 B pad
#define DN_DA_PLANE (DN_DA_KS * DN_DA_ROWB)      // 9216
#define DN_DA_BUF (6 * DN_DA_PLANE)              // 55296: one step, A planes then B planes
#define DN_DA_LDS (128 * 1024)                   // two step buffers (108 KiB) / the 4 x 32 KiB quadrant exchange of the epilogue

struct DaRegs {
    float4 dd, gx, gy;
    float live;
};

__device__ __forceinline__ void da_load(const DaArgs& g, long long r_beg, long long r_end, int step, int kr, int q, DaRegs& R) {
    const long long row = r_beg + (long long)step * DN_DA_KS + kr;
    const bool ok = row < r_end;
    const long long off = (ok ? row : r_beg) * 128 + 4 * q;      // always a valid address; dead rows are zeroed by `live`
    R.live = ok ? 1.f : 0.f;
    R.dd = *reinterpret_cast<const float4*>(g.dd + off);
    R.gx = *reinterpret_cast<const float4*>(g.gx + off);
    R.gy = *reinterpret_cast<const float4*>(g.gy + off);
}

// NP planes: 3 = split-bf16 (hi, mid, lo); 2 = split-fp16 (hi, lo) of v * s
template <int NP>
__device__ __forceinline__ void da_put(unsigned char* planes, int off, float4 v, float s) {
    uint2 pl[NP];
    dn_split_f4<NP>(v, s, pl);
#pragma unroll
    for (int p = 0; p < NP; ++p) *reinterpret_cast<uint2*>(planes + p * DN_DA_PLANE + off) = pl[p];
}

// one 16-row step -> the A planes ([dd*gx | dd*gy]) and the B planes ([gx | gy]) of a step buffer
template <int NP>
__device__ __forceinline__ void da_store(unsigned char* buf, int kr, int q, const DaRegs& R, float sa, float sb) {
    const float4 bx = dn_f4_scale(R.gx, R.live), by = dn_f4_scale(R.gy, R.live);
    const float4 ax = dn_f4_mul(R.dd, bx), ay = dn_f4_mul(R.dd, by);
    const int off = kr * DN_DA_ROWB + q * 8;
    da_put<NP>(buf, off, ax, sa);
    da_put<NP>(buf, off + 256, ay, sa);
    da_put<NP>(buf + 3 * DN_DA_PLANE, off, bx, sb);
    da_put<NP>(buf + 3 * DN_DA_PLANE, off + 256, by, sb);
}

__device__ __forceinline__ uint4 da_frag(const unsigned char* p) {   // 8 consecutive k of one column: two transpose reads of 4 rows
    const uint2 lo = dn_lds_tr16(p), hi = dn_lds_tr16(p + 4 * DN_DA_ROWB);
    return make_uint4(lo.x, lo.y, hi.x, hi.y);
}

template <int NP>
__device__ __forceinline__ void da_compute(const unsigned char* buf, int wr, int wc, int lane, f32x16 (&acc)[4][2]) {
    const int g = lane >> 4, c = lane & 15;
    const int lane_off = (8 * (g >> 1) + (c >> 2)) * DN_DA_ROWB + (16 * (g & 1) + 4 * (c & 3)) * 2;
    const unsigned char* sA = buf + lane_off + wr * 256;                       // this wave's 128 operand-A columns
    const unsigned char* sB = buf + 3 * DN_DA_PLANE + lane_off + wc * 128;     // its 64 operand-B columns
    uint4 b[NP][2];
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) b[p][nt] = da_frag(sB + p * DN_DA_PLANE + nt * 64);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        uint4 a[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) a[p] = da_frag(sA + p * DN_DA_PLANE + mt * 64);
        // same per-accumulator order of the products as the generic kernel (NP = 3: mid*mid, hi*lo, lo*hi, hi*mid, mid*hi, hi*hi;
        // NP = 2, split-fp16: hi*lo, lo*hi, hi*hi)
        constexpr int NPROD = NP == 3 ? 6 : 3;
        constexpr int PA[6] = {NP == 3 ? 1 : 0, NP == 3 ? 0 : 1, NP == 3 ? 2 : 0, 0, 1, 0};
        constexpr int PB[6] = {1, NP == 3 ? 2 : 0, 0, 1, 0, 0};
#pragma unroll
        for (int p = 0; p < NPROD; ++p)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                if constexpr (NP == 3) acc[mt][nt] = dn_mfma_bf16(a[PA[p]], b[PB[p]][nt], acc[mt][nt]);
                else acc[mt][nt] = dn_mfma_f16(a[PA[p]], b[PB[p]][nt], acc[mt][nt]);
            }
    }
}

DN_CLK_DECLARE(tn_da)
template <int NP>
__global__ __launch_bounds__(DN_DA_THREADS) DN_WAVES_PER_EU(2) void tngemm_da_kernel(DaArgs g) {
    DN_CLK_STAMP(tn_da, 0);
    float sa = 1.f, sb = 1.f, so = 1.f;
    if constexpr (NP == 2) {   // split-fp16: A = dd * g (bound: the product of the two magnitudes), B = g
        sa = dn_pow2_scale(dn_amax_eval(g.a_amax));
        sb = dn_pow2_scale(dn_amax_eval(g.b_amax));
        so = (1.f / sa) * (1.f / sb);
    }
    DN_DYN_SMEM(smem_raw);
    unsigned char* smem = reinterpret_cast<unsigned char*>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 2, wc = wave & 3;
    const int kr = tid >> 5, q = tid & 31;               // staging: row kr of the step, columns 4q..4q+3 of each array

    const long long r_beg = (long long)blockIdx.x * g.rows_per_wg;
    const long long r_end = (r_beg + g.rows_per_wg < g.V) ? r_beg + g.rows_per_wg : g.V;
    const int nsteps = r_end > r_beg ? (int)((r_end - r_beg + DN_DA_KS - 1) / DN_DA_KS) : 0;

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (nsteps > 0) {
        DaRegs R0, R1;
        da_load(g, r_beg, r_end, 0, kr, q, R0);
        da_load(g, r_beg, r_end, 1, kr, q, R1);           // past-the-end steps load a valid row and stage zeros
        da_store<NP>(smem, kr, q, R0, sa, sb);
        da_load(g, r_beg, r_end, 2, kr, q, R0);
        __syncthreads();
        // steps in pairs so that the ring set is a compile-time choice: buffer s&1 holds step s; R1 carries odd steps, R0 even ones
        for (int s = 0; s < nsteps; s += 2) {
            da_store<NP>(smem + DN_DA_BUF, kr, q, R1, sa, sb);            // step s+1
            da_load(g, r_beg, r_end, s + 3, kr, q, R1);
            da_compute<NP>(smem, wr, wc, lane, acc);                      // step s
            __syncthreads();
            if (s + 1 < nsteps) {
                da_store<NP>(smem, kr, q, R0, sa, sb);                    // step s+2
                da_load(g, r_beg, r_end, s + 4, kr, q, R0);
                da_compute<NP>(smem + DN_DA_BUF, wr, wc, lane, acc);      // step s+1
                __syncthreads();
            }
        }
    }
    __syncthreads();   // (also orders the last reads of the step buffers before the exchange overwrites them)

    // ---- epilogue: dA_re = Qxx + Qyy, dA_im = Qyx - Qxy.  The waves of the [gy] strips (wc >= 2) hand their tiles over in LDS
    // (fragment layout: 64 consecutive floats per register index -> conflict-free both ways); the [gx] waves add / subtract.
    float* ex = reinterpret_cast<float*>(smem);
    if (wc >= 2) {
        float* mine = ex + (size_t)(wr * 2 + (wc - 2)) * 8 * 16 * 64;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) mine[((mt * 2 + nt) * 16 + r) * 64 + lane] = acc[mt][nt][r];
    }
    __syncthreads();
    if (wc < 2) {
        const float* theirs = ex + (size_t)((1 - wr) * 2 + wc) * 8 * 16 * 64;
        const float sgn = wr == 0 ? 1.f : -1.f;          // wr 0: Qxx + Qyy -> dA_re;  wr 1: Qyx - Qxy -> dA_im
        float* out = g.partial + ((long long)blockIdx.x * 2 + wr) * 128 * 128;
        const int li = lane & 31;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int o = mt * 32 + dn_acc_row(r, lane), c = wc * 64 + nt * 32 + li;
                    const float v = acc[mt][nt][r] + sgn * theirs[((mt * 2 + nt) * 16 + r) * 64 + lane];
                    out[o * 128 + c] = NP == 2 ? v * so : v;
                }
    }
    DN_CLK_STAMP(tn_da, 1);
}

// partial: [nwg][2][128][128] floats (dA_re part, dA_im part of every workgroup's row range)
int dn_launch_tn_da(const float* dd, const float* gx, const float* gy, long long V, float* partial, int nwg, hipStream_t stream,
                    const float* dd_amax, const float* g_amax) {
    if (V <= 0 || nwg <= 0) return DN_ERR_BAD_MODE;
    DaArgs g;
    memset(&g, 0, sizeof(g));
    g.dd = dd; g.gx = gx; g.gy = gy; g.partial = partial; g.V = V;
    g.f16 = (dd_amax && g_amax) ? 1 : 0;
    g.a_amax.p[0] = dd_amax; g.a_amax.mul = g_amax; g.b_amax.p[0] = g_amax;
    const long long per = (V + nwg - 1) / nwg;
    g.rows_per_wg = (int)((per + DN_DA_KS - 1) / DN_DA_KS * DN_DA_KS);
#ifndef DN_EMULATE
    static unsigned long long lds_opt_in = 0;   // per-device bitmap
    { const int oe_ = dn_lds_opt_in(reinterpret_cast<const void*>(&tngemm_da_kernel<3>), DN_DA_LDS, &lds_opt_in); if (oe_) return oe_; }
    static unsigned long long lds_opt_in2 = 0;
    { const int oe_ = dn_lds_opt_in(reinterpret_cast<const void*>(&tngemm_da_kernel<2>), DN_DA_LDS, &lds_opt_in2); if (oe_) return oe_; }
#endif
    dn_prof_begin(DN_K_TN_DA, stream);
    if (g.f16) DN_LAUNCH(tngemm_da_kernel<2>, dim3(nwg, 1, 1), dim3(DN_DA_THREADS, 1, 1), DN_DA_LDS, stream, g);
    else DN_LAUNCH(tngemm_da_kernel<3>, dim3(nwg, 1, 1), dim3(DN_DA_THREADS, 1, 1), DN_DA_LDS, stream, g);
    // four 128 x 128 products over V rows; three arrays read once, the partials written
    dn_prof_end(DN_K_TN_DA, stream, 8.0 * (double)V * 128.0 * 128.0, 4.0 * (3.0 * (double)V * 128.0 + 2.0 * nwg * 128.0 * 128.0));
    return (int)hipGetLastError();
}
