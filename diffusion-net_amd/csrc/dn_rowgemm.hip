// dn_rowgemm.hip -- row GEMM  out[r, n] = epi( sum_k A[r, k] * B(k, n) )  over 128-row vertex tiles: the lock-step kernel
// (every wave stages, multiplies and stores; all shapes, one or two outputs) and the dispatcher of the whole family.
//   A is the long operand ([V, .] activations or the eigenbasis), B a small matrix (weights or a per-mesh spectrum).
//   Replaces: geometry.from_basis (geometry.py:598), every nn.Linear of the block (layers.py:122-126, :236), their
//   input-gradients in backward, and the element-wise body of SpatialGradientFeatures.
// The persistent / wave-specialised kernels of the same product live in dn_rowgemm_persist.hip.
#include "dn_gemm_tiles.h"

// implemented in dn_rowgemm_persist.hip: launches a persistent kernel if the product is eligible (returns true) ...
bool dn_rowgemm_try_persistent(const RgArgs& g, int ntiles, int nout, hipStream_t stream, int* err);

#ifndef DN_RG_X3
#define DN_RG_X3 1   // -DDN_RG_X3=0: exact-f32 MFMA in the two-output kernels
#endif
#if defined(DN_DEBUG_SCALES) && !defined(DN_EMULATE)   // development build: the operand scales every workgroup of the two-output split-fp16 kernel read
__device__ float dn_dbg_scales[2 * 8192];
extern "C" int dn_debug_scales_read(float* out, int n) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(dn_dbg_scales), (size_t)(n < 2 * 8192 ? n : 2 * 8192) * sizeof(float)); }
#endif
#define RG_STORE(buf)                                                                                                          \
    do {                                                                                                                       \
        if constexpr (X3) rg_store_x3<NTHR, NOUT, BCOLK, HASQ, A_IT, B_IT, NP>(reinterpret_cast<unsigned char*>(buf),           \
                                                                          reinterpret_cast<unsigned char*>((buf) + SA), tid, R, sa, sb); \
        else rg_store<TN, NTHR, NOUT, BCOLK, HASQ, A_IT, B_IT>((buf), (buf) + SA, tid, R);                                      \
    } while (0)
#define RG_COMPUTE(buf)                                                                                                        \
    do {                                                                                                                       \
        if constexpr (X3) rg_compute_x3<MT, NT, NOUT, NP>(reinterpret_cast<const unsigned char*>(buf),                         \
                                                      reinterpret_cast<const unsigned char*>((buf) + SA), wr * MT * 32,        \
                                                      wc * NT * 32, li, ls, acc);                                              \
        else rg_compute<TN, MT, NT, NOUT, BCOLK>((buf), (buf) + SA, wr * MT * 32, wc * NT * 32, li, ls, acc);                  \
    } while (0)
#ifndef DN_RG2_VEC_EPI
#define DN_RG2_VEC_EPI 1   // parked float4 epilogue of the two-output split-bf16 kernel (0: per-element dword epilogue)
#endif
#ifndef DN_RG2_EARLY_EPI
#define DN_RG2_EARLY_EPI(MODE) 1   // epilogue operands fetched under the last two slices (measured: fwd 201 -> 186 us, bwd pair 360 -> 348 us; 0: after them)
#endif
constexpr bool rg_is_x3(int TN, int NTHR, int NOUT, bool ALIGNED) { return DN_RG_X3 && ALIGNED && NOUT == 2 && TN == 128 && NTHR == 512; }
// NP: planes of the split engine on the two-output path (3 = split-bf16; 2 = split-fp16 with the operand scales of RgArgs.a_amax / b_amax)
template <int TN, int WR, int WC, int NOUT, int MODE, bool ALIGNED, bool BCOLK, int NP = 3>
__global__ __launch_bounds__(WR* WC * 64) DN_MIN_WAVES_PER_EU(1)
void rowgemm_kernel(RgArgs g) {
    const unsigned long long seed = (MODE == DN_EPI_BIAS_RELU) ? rg_seed(g) : 0ull;
    float sa = 1.f, sb = 1.f, so = 1.f;
    if constexpr (NP == 2) {
        sa = dn_pow2_scale(dn_amax_eval(g.a_amax));
        sb = dn_pow2_scale(dn_amax_eval(g.b_amax));
        so = (1.f / sa) * (1.f / sb);
#if defined(DN_DEBUG_SCALES) && !defined(DN_EMULATE)
        if (threadIdx.x == 0 && blockIdx.x < 8192 && blockIdx.y == 0) { dn_dbg_scales[2 * blockIdx.x] = sa; dn_dbg_scales[2 * blockIdx.x + 1] = sb; }
#endif
    }

    constexpr int NTHR = WR * WC * 64;
    constexpr int MT = DN_TM / (32 * WR);
    constexpr int NT = TN / (32 * WC);
    constexpr int A_IT = DN_TM * 8 / NTHR;
    constexpr int B_IT = DN_KB * TN / 4 / NTHR;
    // the two-output (gradient feature) products run on split-bf16 MFMA: three bf16 planes per operand tile
    constexpr bool X3 = DN_RG_X3 && ALIGNED && NOUT == 2 && TN == 128 && NTHR == 512;
    constexpr int SA = X3 ? (DN_TM * 64 * 3) / 4 : DN_TM * DN_KB;                     // floats of one A slice
    constexpr int SBUF = SA + NOUT * (X3 ? (128 * 64 * 3) / 4 : DN_KB * TN);          // one (A,B) slice buffer; two in LDS
    constexpr bool HASQ = (MODE == DN_EPI_GRADFEAT_BWD);   // the only op whose A operand is an elementwise product
    constexpr bool PAIRK = X3 && !BCOLK;
    static_assert(MT >= 1 && NT >= 1 && A_IT >= 1 && B_IT >= 1, "bad tile config");

    DN_DYN_SMEM(smem_raw);
    float* smem = reinterpret_cast<float*>(smem_raw);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave / WC, wc = wave % WC;
    const int li = lane & 31, ls = lane >> 5;
    const DnTile tile = g.tiles[blockIdx.x];
    const int n0 = blockIdx.y * TN;
    // a wave whose whole sub-tile lies outside the tile's rows / the output's columns has nothing to store
    const bool wave_active = (wr * MT * 32 < tile.nrows) && (n0 + wc * NT * 32 < g.N);

    f32x16 acc[NOUT][MT][NT];
#pragma unroll
    for (int o = 0; o < NOUT; ++o)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[o][mt][nt][r] = 0.f;

    RgRegs<NOUT, A_IT, B_IT> R;
    int nslices = 0;
    for (int s = 0; s < g.nseg; ++s) nslices += (g.a[s].w + DN_KB - 1) / DN_KB;

    // Two-output split-bf16 configuration: the epilogue's elementwise operands (float4 pieces, see below) are fetched while the last
    // two slices are still being multiplied -- with one workgroup per CU nothing else would cover that HBM round trip.
    constexpr bool VEPI = X3 && DN_RG2_VEC_EPI;
    constexpr int NPC = VEPI ? 128 * 128 / 4 / NTHR : 1;   // 8 pieces per thread
    float4 er0[NPC], er1[NPC], er2[NPC];
    long long eoff[NPC];
    bool eok[NPC];
    bool vec_ok = false;
    if constexpr (VEPI)
        vec_ok = (((uintptr_t)g.o0 | (uintptr_t)g.o1 | (uintptr_t)g.o2 | (uintptr_t)g.r0 | (uintptr_t)g.r1 | (uintptr_t)g.r2) & 15) == 0 &&
                 g.ldo % 4 == 0 && g.ldr % 4 == 0 && g.N % 4 == 0;
    auto epi_fetch = [&]() {
        if constexpr (VEPI) {
            if (vec_ok) {
#pragma unroll
                for (int k = 0; k < NPC; ++k) {
                    const int idx = tid + k * NTHR;
                    const int row = idx >> 5, c4 = idx & 31;
                    const int col = n0 + 4 * c4;
                    eok[k] = row < tile.nrows && col < g.N;
                    const long long grow = tile.row0 + (eok[k] ? row : 0);
                    const int ccol = eok[k] ? col : 0;
                    eoff[k] = grow * g.ldo + ccol;
                    const long long roff = grow * g.ldr + ccol;
                    er0[k] = *reinterpret_cast<const float4*>(g.r0 + roff);
                    er1[k] = *reinterpret_cast<const float4*>(g.r1 + roff);
                }
            }
        }
    };

    // Software pipeline over 32-wide slices of the contraction axis, two LDS buffers, ONE barrier per slice:
    //   iteration sl:  regs(slice sl+1) -> LDS[other] ; global loads of slice sl+2 -> regs ; MFMAs on LDS[cur] ; barrier
    // The steady-state body has no branch, so the LDS writes and the global loads can be scheduled under the MFMAs.
    int seg = 0, koff = 0;
    rg_load<TN, NTHR, NOUT, ALIGNED, BCOLK, HASQ, A_IT, B_IT, PAIRK>(g, tile, n0, seg, koff, tid, R);
    RG_STORE(smem);
    if (nslices > 1) {
        koff += DN_KB;
        if (koff >= g.a[seg].w) { koff = 0; ++seg; }
        rg_load<TN, NTHR, NOUT, ALIGNED, BCOLK, HASQ, A_IT, B_IT, PAIRK>(g, tile, n0, seg, koff, tid, R);
    }
    __syncthreads();
    int sl = 0;
    for (; sl + 2 < nslices; ++sl) {
        float* cur = smem + (sl & 1) * SBUF;
        float* nxt = smem + ((sl & 1) ^ 1) * SBUF;
        RG_STORE(nxt);
        koff += DN_KB;
        if (koff >= g.a[seg].w) { koff = 0; ++seg; }
        rg_load<TN, NTHR, NOUT, ALIGNED, BCOLK, HASQ, A_IT, B_IT, PAIRK>(g, tile, n0, seg, koff, tid, R);
        RG_COMPUTE(cur);
        __syncthreads();
    }
    if (sl + 1 < nslices) {   // second-to-last slice: stage the last one, nothing left to load
        float* cur = smem + (sl & 1) * SBUF;
        float* nxt = smem + ((sl & 1) ^ 1) * SBUF;
        RG_STORE(nxt);
        if (DN_RG2_EARLY_EPI(MODE)) epi_fetch();   // (the slice loads have all been consumed: nothing younger is waited on before the epilogue)
        RG_COMPUTE(cur);
        __syncthreads();
        ++sl;
    } else {
        if (DN_RG2_EARLY_EPI(MODE)) epi_fetch();
    }
    {
        float* cur = smem + (sl & 1) * SBUF;
        RG_COMPUTE(cur);
    }
    if (!DN_RG2_EARLY_EPI(MODE)) epi_fetch();

    // ---------------- epilogue ----------------
    if constexpr (NP == 2) {   // split-fp16: exact power-of-two rescale of the products
#pragma unroll
        for (int o = 0; o < NOUT; ++o)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[o][mt][nt][r] *= so;
    }
    if constexpr (VEPI) {
        // The slice buffers are dead now, so both 128 x 128 accumulator tiles are parked in them (2 x 64 KiB of the 144 KiB) and the
        // epilogue runs on float4 pieces with coalesced 16-byte loads (issued above) and stores -- 8 pieces x (2-3 loads + 2-3
        // stores) per thread instead of 32 elements x (2-3 dword loads + 2-3 dword stores).
        if (vec_ok) {
            float* sE0 = smem;
            float* sE1 = smem + 128 * 128;
            if (MODE == DN_EPI_GRADFEAT_BWD) {   // the third operand only now (register budget); the parking below covers part of its latency
#pragma unroll
                for (int k = 0; k < NPC; ++k) {
                    const int idx = tid + k * NTHR;
                    const long long grow = tile.row0 + (eok[k] ? (idx >> 5) : 0);
                    er2[k] = *reinterpret_cast<const float4*>(g.r2 + grow * g.ldr + (eok[k] ? n0 + 4 * (idx & 31) : 0));
                }
            }
            __syncthreads();   // every wave is done with the slice buffers
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int e = ((wr * MT + mt) * 32 + dn_acc_row(r, lane)) * 128 + wc * 32 + li;
                    sE0[e] = acc[0][mt][0][r];
                    sE1[e] = acc[NOUT - 1][mt][0][r];
                }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < NPC; ++k) {
                const int idx = tid + k * NTHR;
                const int row = idx >> 5, c4 = idx & 31;
                const float4 a0 = *reinterpret_cast<const float4*>(&sE0[row * 128 + 4 * c4]);
                const float4 a1 = *reinterpret_cast<const float4*>(&sE1[row * 128 + 4 * c4]);
                if (MODE == DN_EPI_GRADFEAT) {
                    const float4 y = make_float4(tanhf(er0[k].x * a0.x + er1[k].x * a1.x), tanhf(er0[k].y * a0.y + er1[k].y * a1.y),
                                                 tanhf(er0[k].z * a0.z + er1[k].z * a1.z), tanhf(er0[k].w * a0.w + er1[k].w * a1.w));
                    if (eok[k]) {
                        *reinterpret_cast<float4*>(g.o0 + eoff[k]) = y;
                        if (g.o1) {
                            *reinterpret_cast<float4*>(g.o1 + eoff[k]) = a0;
                            *reinterpret_cast<float4*>(g.o2 + eoff[k]) = a1;
                        }
                    }
                } else {
                    const float4 y0 = make_float4(a0.x + er0[k].x * er1[k].x, a0.y + er0[k].y * er1[k].y, a0.z + er0[k].z * er1[k].z, a0.w + er0[k].w * er1[k].w);
                    const float4 y1 = make_float4(a1.x + er0[k].x * er2[k].x, a1.y + er0[k].y * er2[k].y, a1.z + er0[k].z * er2[k].z, a1.w + er0[k].w * er2[k].w);
                    if (eok[k]) {
                        *reinterpret_cast<float4*>(g.o0 + eoff[k]) = y0;
                        *reinterpret_cast<float4*>(g.o1 + eoff[k]) = y1;
                    }
                }
            }
            return;
        }
    }
    if (wave_active) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int rbase = (wr * MT + mt) * 32;
                const int col = n0 + (wc * NT + nt) * 32 + li;
                if (rbase < tile.nrows)
                    rg_epilogue_tile<MODE, NOUT>(g, seed, tile.row0 + rbase, tile.nrows - rbase, col, col < g.N, lane,
                                                 acc[0][mt][nt], acc[NOUT - 1][mt][nt]);
            }
    }
}

#undef RG_STORE
#undef RG_COMPUTE

template <int TN, int WR, int WC, int NOUT, int MODE, bool ALIGNED, bool BCOLK, int NP = 3>
static int rg_launch(const RgArgs& g, int ntiles, hipStream_t stream) {
    const int ncol = (g.N + TN - 1) / TN;
    constexpr bool X3 = DN_RG_X3 && ALIGNED && NOUT == 2 && TN == 128 && WR * WC == 8;
    const size_t smem = X3 ? (size_t)2 * (DN_TM * 64 * 3 + NOUT * 128 * 64 * 3)
                           : (size_t)2 * (DN_TM * DN_KB + NOUT * DN_KB * TN) * sizeof(float);
#ifndef DN_EMULATE
    static unsigned long long lds_opt_in = 0;   // per-device bitmap
    { const int oe_ = dn_lds_opt_in(reinterpret_cast<const void*>(&rowgemm_kernel<TN, WR, WC, NOUT, MODE, ALIGNED, BCOLK, NP>), smem, &lds_opt_in); if (oe_) return oe_; }
#endif
    DN_LAUNCH((rowgemm_kernel<TN, WR, WC, NOUT, MODE, ALIGNED, BCOLK, NP>), dim3(ntiles, ncol, 1), dim3(WR * WC * 64, 1, 1), smem,
              stream, g);
    return (int)hipGetLastError();
}

// tile width by output width on the aligned path; the general (odd-size) path always uses the 128-wide tile
template <int NOUT, int MODE, bool BCOLK>
static int rg_dispatch_width(const RgArgs& g, int ntiles, hipStream_t stream) {
    constexpr int WC128 = NOUT == 1 ? 2 : 4;
    if (!g.aligned) return rg_launch<128, 2, WC128, NOUT, MODE, false, BCOLK>(g, ntiles, stream);
    if (g.N <= 32) return rg_launch<32, 4, 1, NOUT, MODE, true, BCOLK>(g, ntiles, stream);
    if (g.N <= 64) return rg_launch<64, 2, 2, NOUT, MODE, true, BCOLK>(g, ntiles, stream);
    if (NOUT == 2 && g.f16) return rg_launch<128, 2, WC128, NOUT, MODE, true, BCOLK, NOUT == 2 ? 2 : 3>(g, ntiles, stream);
    return rg_launch<128, 2, WC128, NOUT, MODE, true, BCOLK>(g, ntiles, stream);
}
int dn_launch_rowgemm(const RgArgs& g, int ntiles, int nout, hipStream_t stream) {
    if (ntiles <= 0 || g.N <= 0 || g.nseg <= 0) return 0;
    int ktot = 0;
    for (int s = 0; s < g.nseg; ++s) ktot += g.a[s].w;
    const double rows = g.acct_rows;
    const double flops = 2.0 * rows * ktot * g.N * nout;
    const double bytes = 4.0 * (rows * ktot + rows * (double)g.N * nout + (double)ktot * g.N * nout);
    const int kind = nout == 1 ? DN_K_ROWGEMM : DN_K_ROWGEMM_DUAL;
    dn_prof_begin(kind, stream);
    int err = DN_ERR_BAD_MODE;
    const bool ck = g.b_colk != 0;
    if (dn_rowgemm_try_persistent(g, ntiles, nout, stream, &err)) {
        dn_prof_end(kind, stream, flops, bytes);
        return err;
    }
    if (nout == 1) {
        switch (g.mode) {
            case DN_EPI_STORE:
                err = ck ? rg_dispatch_width<1, DN_EPI_STORE, true>(g, ntiles, stream)
                         : rg_dispatch_width<1, DN_EPI_STORE, false>(g, ntiles, stream);
                break;
            case DN_EPI_BIAS_RELU: if (ck) err = rg_dispatch_width<1, DN_EPI_BIAS_RELU, true>(g, ntiles, stream); break;
            case DN_EPI_BIAS_RESID: if (ck) err = rg_dispatch_width<1, DN_EPI_BIAS_RESID, true>(g, ntiles, stream); break;
            case DN_EPI_MUL_DFAC: if (!ck) err = rg_dispatch_width<1, DN_EPI_MUL_DFAC, false>(g, ntiles, stream); break;
            case DN_EPI_ADD: if (!ck) err = rg_dispatch_width<1, DN_EPI_ADD, false>(g, ntiles, stream); break;
            case DN_EPI_DTANH: if (!ck) err = rg_dispatch_width<1, DN_EPI_DTANH, false>(g, ntiles, stream); break;
            case DN_EPI_MASS_ADD: if (!ck) err = rg_dispatch_width<1, DN_EPI_MASS_ADD, false>(g, ntiles, stream); break;
            default: break;
        }
    } else {
        switch (g.mode) {
            case DN_EPI_GRADFEAT: if (ck) err = rg_dispatch_width<2, DN_EPI_GRADFEAT, true>(g, ntiles, stream); break;
            case DN_EPI_GRADFEAT_BWD: if (!ck) err = rg_dispatch_width<2, DN_EPI_GRADFEAT_BWD, false>(g, ntiles, stream); break;
            default: break;
        }
    }
    dn_prof_end(kind, stream, flops, bytes);
    return err;
}

