// dn_chain.hip -- the chained row pipeline of DiffusionNetBlock.forward (layers.py:217-239): CSR gradient gather -> complex-linear
// gradient features + tanh -> MiniMLP (all layers) + residual, ONE launch, activations never leave the registers between stages.
//
//   Replaces, per block forward: spmm_kernel (gradX/gradY apply), the two-output gradient-feature row GEMM, and the three MiniMLP row
//   GEMMs -- five launches and ~18 passes over [V, C] arrays -- by one kernel that reads xd (gathered) and x once and writes, in
//   training, each saved activation once (gx, gy, bre, bim, g, h_j, out); in inference only `out`.
//
// Decomposition (gfx950, wave64): a wave owns 32 consecutive vertex rows for the whole chain, as two 16-row halves on
// v_mfma_f32_16x16x32_f16.  The products are computed TRANSPOSED, D[n][m] = sum_k W[n][k] act[m][k]: the weights are the MFMA's A
// operand (rows = output channels), the activations its B operand (columns = vertices).  The accumulator layout of that instruction
// -- lane (m = l & 15, q = l >> 4) holds output channels 16 nt + 4 q + r, r < 4, of vertex m -- is, channel for channel, what the SAME
// lane must supply as the B operand of the next layer once the contraction index is enumerated in the order
//        slot j of k-group q in 32-channel step T  <->  channel 32 T + 16 (j >> 2) + 4 q + (j & 3),
// so a layer's output becomes the next layer's operand with no data movement: bias / ReLU / dropout / tanh in registers, a two-term
// fp16 split (dn_split2_pair, as the split-fp16 row GEMMs), done.  The weights are permuted to that order once per call by
// chain_prep_kernel, which also splits them into fp16 (hi, lo) planes laid out as the LDS image of a "piece" (one matrix, one
// 32-channel contraction step: [plane][nt][lane] x 16 bytes, conflict-free ds_read_b128).  Pieces stream through a two-slot LDS ring,
// one barrier per piece; every wave of the workgroup consumes every piece (the weights are the only thing the waves share).
// The gradient features of the two 16-row halves are computed one after the other (the stage that needs gx, gy AND both accumulators
// in registers); their pieces are therefore listed twice in the stream.
//
// Arithmetic: the split-fp16 engine of dn_rowgemm_persist.hip (three MFMAs per product term pair: hi*lo, lo*hi, hi*hi; fp32
// accumulation; operands scaled by powers of two and the result scaled back exactly).  Operand scales: x, xd from the producers'
// magnitude words, gx / gy from the bound ||G||_inf max|xd| (as the unfused path), tanh features by 1, hidden activations by the
// largest magnitude of the wave's own 32 x C tile (a wave reduces it with shuffles -- tighter than the per-tensor word of the
// unfused path, and available without a grid-wide dependency).
#include "dn_common.h"
#include <stdlib.h>

#include "dn_chain_tiles.h"

#ifndef DN_CH_GCR
#define DN_CH_GCR 1      // (1: 393.9 us block forward at 16 x 10k, 2: 408.1, 3: 431.5 -- registers, not bytes in flight; profiles/r05_rcg_ab.txt)
#endif
#ifndef DN_CH_L0_EXACT
#define DN_CH_L0_EXACT 1  // layer 0's x / xd row requests behind the piece request and counted exactly in the end-of-piece wait (0: the round-4 order; A/B)
#endif
#ifndef DN_CH_PF_WIDE
#define DN_CH_PF_WIDE 2
#endif
#ifndef DN_CH_NXR_WIDE
#define DN_CH_NXR_WIDE 5  // x / xd operand rows of layer 0 in flight, in pieces + 1, in the C = 256 form (3 at C <= 128)
#endif
#ifndef DN_CH_GCR_WIDE
#define DN_CH_GCR_WIDE 2 // the same in the C = 256 form (one wave per SIMD: 16 KiB in flight per wave)
#endif
#ifndef DN_CH_LINES_WIDE
#define DN_CH_LINES_WIDE 1   // C = 256: the result rows as whole-line stores too (dn_chain_tiles.h, ch_st_tiles)
#endif
#ifndef DN_CH_ROLL
#define DN_CH_ROLL 1     // rolling row requests in the row-contiguous gather: 1 = in the C = 256 form (latency-bound there: one wave per SIMD), 2 = everywhere
                         // (C = 128: 28 spilled registers in the benchmark's kernel), 0 = all rows of a step requested, then all summed
#endif
#ifndef DN_CH_SG_PF
#define DN_CH_SG_PF 0    // contraction steps of the spectral stage whose operand fragments are requested a pass ahead (24 registers each: 1 / 2 / 3 / 4 steps carried cost 10 / 42 / 101 / 140 spilled registers at C = 128 wherever they are requested)
#endif
#ifndef DN_CH_SG_PFPOINT
#define DN_CH_SG_PFPOINT 1
#endif
#ifndef DN_CH_SG_MAXP
#define DN_CH_SG_MAXP 64 // passes per workgroup of the spectral-gradient form (their descriptors live in LDS; the launcher checks)
#endif
#ifndef DN_CH_RCG
#define DN_CH_RCG 1      // row-contiguous gather in the one-half-per-wave form of the chained forward (0: operand-layout gather everywhere; A/B)
#endif

template <int C>
__global__ __launch_bounds__(1024) void chain_prep_kernel(ChainPrepArgs a) {
    constexpr int NT = C / 16;
    constexpr int NTHR = 1024;            // one workgroup per piece: what it costs is latency, so many threads with few steps each
    __shared__ float red[NTHR];
    const int tid = threadIdx.x;
    if ((int)blockIdx.x >= a.npieces) {          // start-of-call bookkeeping
        for (int r = 0; r < a.nzero; ++r)
            for (int i = tid; i < a.zero_n[r]; i += NTHR) a.zero[r][i] = 0.f;
        if (tid == 0 && a.copy_src && a.copy_dst) *a.copy_dst = *a.copy_src;
        for (int i = tid; i < a.clamp_n; i += NTHR) { const float v = a.clamp_p[i]; a.clamp_p[i] = v < a.clamp_min ? a.clamp_min : v; }   // (NaN stays NaN, as torch.clamp)
        return;
    }
    const ChainPrepPiece pc = a.pc[blockIdx.x];
    uint4* out = a.out + (size_t)blockIdx.x * (2 * NT * 64);
    // this thread's eight elements of the piece are requested FIRST (they do not depend on the matrix's magnitude, only their split does): their
    // latency passes under the magnitude sweep below (round 6: 6.1 -> ~5 us per launch, eight launches per training step)
    static_assert(NT * 64 <= NTHR, "one piece element group per thread");
    float va[4] = {0.f, 0.f, 0.f, 0.f}, vb[4] = {0.f, 0.f, 0.f, 0.f};
    if (tid < NT * 64) {
        const int nt = tid >> 6, lane = tid & 63;
        const int n = 16 * nt + (lane & 15), q = lane >> 4;
        if (pc.transposed) {     // rows of the piece = columns of W (backward products)
            const float* src = pc.W + (long long)(pc.col0 + 4 * q) * pc.ld + pc.row0 + n;
#pragma unroll
            for (int t = 0; t < 4; ++t) { va[t] = src[(long long)t * pc.ld]; vb[t] = src[(long long)(16 + t) * pc.ld]; }
        } else {
            const float* src = pc.W + (long long)n * pc.ld + pc.col0 + 4 * q;
#pragma unroll
            for (int t = 0; t < 4; ++t) { va[t] = src[t]; vb[t] = src[16 + t]; }
        }
    }
    // largest magnitude of the whole matrix (every piece's workgroup measures it: a few tens of KB out of L2)
    float mx = 0.f;
    {
        const int n4 = C * pc.ld / 4;   // (ld is a multiple of 4 and the rows 16-byte aligned on this path)
        // four loads in flight per thread and step (one workgroup: latency, not bandwidth -- a load per step made this kernel 25 us, 256 threads
        // with eight in flight 12 us)
        for (int i0 = tid; i0 < n4; i0 += 4 * NTHR) {
            float4 v[4], w[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + NTHR * u < n4 ? i0 + NTHR * u : i0;
                v[u] = *reinterpret_cast<const float4*>(pc.W + 4 * (long long)i);
                w[u] = pc.W2 ? *reinterpret_cast<const float4*>(pc.W2 + 4 * (long long)i) : dn_f4_zero();
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) { mx = dn_f4_amax(mx, v[u]); mx = dn_f4_amax(mx, w[u]); }
        }
    }
    // wave maxima (DPP), then the sixteen waves' words through LDS: two barriers instead of ten
    mx = ch_wave_max(mx);
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();
    mx = 0.f;
#pragma unroll
    for (int w = 0; w < NTHR / 64; ++w) mx = red[w] > mx ? red[w] : mx;
    if (tid == 0 && pc.amax) *pc.amax = mx;
    const float s = dn_pow2_scale(mx);
    if (tid < NT * 64) {
        uint4 hi, lo;
        ch_split8(va, vb, s, hi, lo);
        out[tid] = hi;
        out[NT * 64 + tid] = lo;
    }
}

#if defined(DN_CH_TRACE) && !defined(DN_EMULATE)   // development build (make variant EXTRA=-DDN_CH_TRACE=<workgroup>): s_memtime stamps of wave 0
__device__ unsigned long long dn_ch_trace_buf[512];
extern "C" int dn_debug_ch_trace_read(unsigned long long* out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(dn_ch_trace_buf), (size_t)(n < 512 ? n : 512) * sizeof(unsigned long long));
}
#define CH_TR()                                                                                                         \
    do {                                                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                                              \
        if (blockIdx.x == (DN_CH_TRACE) && threadIdx.x == 0 && trn < 512) dn_ch_trace_buf[trn] = __builtin_amdgcn_s_memtime(); \
        ++trn;                                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                              \
    } while (0)
#else
#define CH_TR() do {} while (0)
#endif

DN_CLK_DECLARE(chain_fwd)
// KE > 0: the spectral-gradient form (dn_spectral.hip) for k_eig = 32 KE -- no CSR gather, no xd read: the pass starts with the three products
// [Phi | G_X Phi | G_Y Phi][rows] * ys[mesh] (operand fragments streamed pre-split from the packed batch operand, the scaled spectrum's pieces
// through the same LDS ring as the weights) and xd, gx, gy are born in the accumulator layout the following stages consume.
// MODE 2 (C = K = 256 only): the spectral-gradient form there is TWO launches -- spectral_apply_kernel (dn_spectral.hip: xd, gx, gy of every row
// written to memory) and this kernel reading gx, gy where the gather form gathers them (KE = 0: plain piece stream, plain row mapping).  Fused
// into this kernel the spectral products' 192 accumulators + operand buffers sat inside a kernel whose other stages are at the register limit:
// values spilled THERE were reloaded in the products' piece loop, and a scratch reload is followed by a compiler-made s_waitcnt vmcnt(0) that
// drains the ring and the operand requests in flight (8-9 k cycles per piece instead of ~3 k; profiles/r06_sg256.txt).
template <int C, int NW, int HH, int KE = 0, int MODE = 0>
__global__ __launch_bounds__(64 * NW) DN_WAVES_PER_EU(C >= 256 ? 1 : 2) void chain_fwd_kernel(ChainArgs a) {
    constexpr bool SG = KE > 0;
    constexpr bool PRE = MODE == 2;       // gx, gy come from memory (written by spectral_apply_kernel)
    static_assert(MODE == 0 || (MODE == 2 && C >= 256 && KE == 0), "two-launch spectral form: C = 256");
    static_assert(!SG || (HH == 1 && C < 256 && KE <= 4), "fused spectral-gradient form: C <= 128, one 16-row half per wave");

    DN_CLK_STAMP(chain_fwd, 0);
    // (the spectral-gradient forms exist with gradient features only: a compile-time fact there -- as a run-time flag it keeps the MiniMLP
    // accumulators of the no-gradient-features path alive across the spectral phase: 128 of the 256 accumulator registers at C = 256)
    const bool with_grad_ = SG ? true : (a.with_grad != 0);
    constexpr int NT = C / 16;            // 16-channel output tiles
    constexpr int NK = C / 32;            // 32-channel contraction steps (= pieces per matrix)
    constexpr int NTHR = 64 * NW;
    constexpr int PIECE = 2 * NT * 64;    // uint4 per piece
    constexpr int LPT = PIECE / NTHR;     // DMA requests per thread and piece
    constexpr bool G0 = C >= 256;         // the one-wave-per-SIMD form of C = 256 (see the gradient-feature stage)
    constexpr int RING = DN_CH_RING;      // (C = 256: 32 KiB pieces; 128 KiB of ring + 32 KiB of gather slices are the CU's 160 KiB)
    constexpr int GCH = DN_CH_GCHUNK;
    constexpr int CH_PF = G0 ? DN_CH_PF_WIDE : 1;      // weight-fragment prefetch distance in tile pairs (one wave per SIMD: nobody else covers the LDS latency)
    static_assert(PIECE % NTHR == 0, "piece staging");
    static_assert(RING >= 2 && RING <= 8 && (RING - 1) * LPT < 60 && (RING - 2) * LPT + 6 * HH < 64, "ring depth vs the vmcnt range");


    DN_DYN_SMEM(smem_raw);
    uint4* ring = reinterpret_cast<uint4*>(smem_raw);                      // RING slots of PIECE uint4
    float* sbias = reinterpret_cast<float*>(ring + RING * PIECE);          // [DN_CH_LAYERS][C]  (G0: not staged, the epilogues read the biases from memory)
    // row-contiguous gather (RCG; at C = 128 in the one-half-per-wave form only: it needs the registers the second half's features occupy
    // otherwise.  The C = 256 form runs one wave per SIMD with 512 registers and takes it with both halves)
    constexpr bool RCG = !SG && (HH == 1 || C >= 256) && DN_CH_RCG != 0;
    constexpr int GC = G0 ? 128 : C;      // channels per gather sweep (G0: the row in two column halves -- sums, request buffers and the slice are
                                          // those of C = 128, the half gathered first is in its final registers while the second one runs)
    constexpr int NSW = C / GC;           // sweeps
    constexpr int GNT = GC / 16;          // 16-channel tiles per sweep
    constexpr int LPR = GC / 4;           // gather: lanes that cover one row (segment) of a sweep (16 bytes each)
    constexpr int RPI = 64 / LPR;         // rows one gather instruction covers
    constexpr int NI = 16 / RPI;          // row groups of a 16-row half
    constexpr int SROWS = G0 ? 16 : 4;    // rows per transposition slice (G0: the whole half at once -- every lane reads its row unconditionally, and
                                          // gx / gy become registers one after the other instead of as 128 conditionally merged ones)
    constexpr int GCR = G0 ? DN_CH_GCR_WIDE : DN_CH_GCR;        // pattern entries per gather step of the row-contiguous form (NI x GCR KiB in flight per wave)
    static_assert(SROWS % RPI == 0 && 16 % SROWS == 0, "gather slices");
    float4* slice = reinterpret_cast<float4*>(sbias + (G0 ? 0 : DN_CH_LAYERS * C)) + (threadIdx.x >> 6) * (SROWS * LPR);   // wave-private, SROWS rows (RCG)
    // SG: the workgroup's passes -- {first row, end of the mesh's rows, mesh, first 16-row group} and the three result scales -- staged once in the
    // space of the gather slices: a per-pass descriptor read from memory compiles to a vector load behind an s_waitcnt vmcnt(0), which drains
    // the piece ring and the operand requests in flight (seen as 8-16 k cycles at the top of every pass)
    int4* pinfo = reinterpret_cast<int4*>(sbias + (G0 ? 0 : DN_CH_LAYERS * C));          // [DN_CH_SG_MAXP]
    float4* pscale = reinterpret_cast<float4*>(pinfo + DN_CH_SG_MAXP);                   // [DN_CH_SG_MAXP]
#ifdef DN_EMULATE
    const unsigned lds0 = 0;
#else
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem_raw;
#endif
#if defined(DN_CH_TRACE) && !defined(DN_EMULATE)
    int trn = 0;
#endif

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = lane & 15, q = lane >> 4;
    const int rsub = lane / LPR, cc = lane % LPR;

    // XCD-contiguous unit ranges: workgroup b runs on XCD b % 8 and walks units of the b % 8-th eighth of the row axis, so that the
    // ~7 neighbour rows a row gathers are mostly rows the same L2 has just served
    // (SG: a.units counts 16 NW-row sub-units of the batch's 64-row units -- rows of ONE mesh, padded: the pass's spectrum is that mesh's)
    const int GX = gridDim.x >> 3;        // workgroups per XCD (the host launches a multiple of 8)
    const int per_x = (a.units + 7) >> 3;
    const int xcd = blockIdx.x & 7, slot0 = blockIdx.x >> 3;
    int npass = 0;
    if (slot0 < per_x) {
        int hi_local = a.units - xcd * per_x;       // units of this XCD's range that exist
        hi_local = hi_local > per_x ? per_x : hi_local;
        if (slot0 < hi_local) npass = (hi_local - slot0 + GX - 1) / GX;
    }
    if (npass == 0) return;

    // ---- operand scales known up front
    // (SG: xd, gx, gy are produced by this kernel -- their scales are per pass, from the wave's own 16 rows, as the hidden activations')
    const float x_mag = SG ? dn_amax_word(a.x_amax) : 0.f;
    float s_in = SG ? 1.f : ch_uniform(dn_pow2_scale(fmaxf(fmaxf(dn_amax_word(a.x_amax), dn_amax_word(a.xd_amax)), with_grad_ ? 1.f : 0.f)));   // [x | xd | g]
    float s_gf = 1.f, so_gf = 1.f;
    const float swa_inv = (SG && with_grad_) ? ch_uniform(ch_pow2_inv(dn_pow2_scale(dn_amax_word(a.wa_amax)))) : 1.f;
    float xdmax = 0.f, gmax = 0.f;        // SG: largest |xd|, |gx|, |gy| this wave produced
    if (!SG && with_grad_) {
        const float gb = PRE ? dn_amax_word(a.g_amax) : dn_amax_word(a.xd_amax) * dn_amax_word(a.grad_norm);      // (PRE: max |gx|, |gy| as measured by the spectral launch)
        if (!PRE && blockIdx.x == 0 && tid == 0 && a.g_amax) atomicMax(reinterpret_cast<unsigned*>(a.g_amax), __float_as_uint(gb));
        s_gf = ch_uniform(dn_pow2_scale(gb));
        so_gf = ch_uniform(ch_pow2_inv(s_gf) * ch_pow2_inv(dn_pow2_scale(dn_amax_word(a.wa_amax))));
    }
    float sw_inv[DN_CH_LAYERS];           // 1 / (power-of-two scale the weights of layer j were split with)
#pragma unroll
    for (int j = 0; j < DN_CH_LAYERS; ++j) sw_inv[j] = j < a.n_mlp ? ch_uniform(ch_pow2_inv(dn_pow2_scale(dn_amax_word(a.w_amax[j])))) : 1.f;
    unsigned long long seed_add = 0ull;
    if (a.seed_dev) seed_add = *a.seed_dev;
    if constexpr (!G0)
        for (int i = tid; i < a.n_mlp * C; i += NTHR) sbias[i] = a.bias[i / C][i % C];   // (published by the first barrier below)

    // ---- the piece stream: LDS-DMA fills slot p % RING with piece p, RING - 1 pieces ahead of the one being multiplied.  Every wave
    //      requests its share of a piece, waits for its share of the NEXT piece at the end of a piece (counted: the younger requests stay in
    //      flight) and the barrier there publishes it -- and tells everybody that the slot read in this piece may be overwritten.
    // The stream of a pass: the gradient-feature pieces once per 16-row half (the stage runs half by half), then every other piece once;
    // position sq of that sequence fetches piece sq (first round of the gradient-feature pieces), sq - n_gf (second round) or
    // sq - (HH - 1) n_gf (the layers).  It wraps at the end of a pass: the requests run ahead into the next one.
    // SG: the pass's KE spectrum pieces come first, from the pass's mesh (the requests run ahead into the next pass: imesh follows the stream)
    const int n_gf = with_grad_ ? a.n_gf : 0;
    // (SG: the spectrum's pieces come first, once per half -- the spectral phase runs both halves back to back)
    const int n_seq = a.n_pieces + (HH - 1) * n_gf + HH * KE;
    int sq = 0;                           // position the NEXT request fetches
    const int SUB = SG ? a.sg_unit_rows / (16 * HH * NW) : 1;  // workgroup passes per unit of the packed operands (64 / 128 rows)
    const int GPU_ = SG ? a.sg_unit_rows / 16 : 1;             // 16-row groups per unit
    auto unit_of = [&](int pass_) { return xcd * per_x + slot0 + pass_ * GX; };
    int imesh = 0, mesh_nx = 0;           // SG: mesh of the pass whose pieces are being requested / of the pass after the one being multiplied
    if constexpr (SG) {
        for (int pp = tid; pp < npass; pp += NTHR) {
            const int un = unit_of(pp);
            const DnTile tl = a.sg_units[un / SUB];
            const int sub = un % SUB;
            pinfo[pp] = int4{tl.row0 + 16 * HH * NW * sub, tl.row0 + tl.nrows, tl.mesh, GPU_ * (un / SUB) + HH * NW * sub};
            const float4 am = *reinterpret_cast<const float4*>(a.sg_amax + 4 * tl.mesh);
            const float ys_inv = ch_pow2_inv(dn_pow2_scale(a.ys_amax[tl.mesh]));
            // .w: a bound of |xd| over the mesh -- (largest row 2-norm of Phi) x (largest column 2-norm of ys), Cauchy-Schwarz -- for the forms whose
            // layer-0 scale must be fixed before xd exists (C = 256)
            pscale[pp] = make_float4(ys_inv * ch_pow2_inv(dn_pow2_scale(am.x)), ys_inv * ch_pow2_inv(dn_pow2_scale(am.y)),
                                     ys_inv * ch_pow2_inv(dn_pow2_scale(am.z)), am.w * a.ys_amax[a.sg_n_mesh + tl.mesh]);
        }
        imesh = ch_uniform_i(a.sg_units[unit_of(0) / SUB].mesh);      // (pass 0's pieces are requested before the table is published)
        mesh_nx = imesh;
    }
#ifdef DN_EMULATE
    const int wave_u = wave;
#else
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
#endif
    int rq = 0;                           // slot the next request fills
    auto issue = [&]() {
        const uint4* src_piece;
        if (SG && sq < HH * KE) {
            src_piece = a.ysp + ((size_t)imesh * KE + (sq >= KE ? sq - KE : sq)) * PIECE;       // (HH <= 2)
        } else {
            const int sw_ = sq - HH * KE;
            const int pidx = sw_ < HH * n_gf ? (sw_ >= n_gf ? sw_ - n_gf : sw_) : sw_ - (HH - 1) * n_gf;
            src_piece = a.wp + (size_t)pidx * PIECE;
        }
#pragma unroll
        for (int i = 0; i < LPT; ++i) {
            const int e0 = rq * PIECE + i * NTHR + wave_u * 64;
            ch_dma16(src_piece + i * NTHR + tid, ring + e0, lds0 + 16u * (unsigned)e0);
        }
        if (sq + 1 == n_seq) { sq = 0; if constexpr (SG) imesh = mesh_nx; }      // (mesh_nx: of the pass after the one being multiplied, or the last one again)
        else ++sq;
        rq = rq + 1 == RING ? 0 : rq + 1;
    };
#ifdef DN_EMULATE
#define CH_WAIT_PIECES(n) do {} while (0)
#define CH_WAIT_OPS(n) do {} while (0)
#define CH_BARRIER() __syncthreads()
#else
    // wait until at most n * LPT of this wave's vector-memory operations are outstanding (loads return in order: everything older than
    // the n youngest pieces has landed), and for every LDS read issued so far (the slot just read may be overwritten after the barrier)
#define CH_WAIT_PIECES(n) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"((n) * LPT) : "memory")
#define CH_WAIT_OPS(n) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(n) : "memory")      // the same, counted in operations
#define CH_BARRIER() do { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)
#endif
    int gp = 0;
#pragma unroll
    for (int i = 0; i < RING - 1; ++i) issue();
    CH_WAIT_PIECES(RING - 2);
    CH_BARRIER();
    // start of a piece: request the piece RING - 1 ahead (into the slot everybody finished reading at the last barrier); ws_ = slot to read
#define CH_PIECE_BEGIN()                                                                                                \
    const uint4* ws_ = ring + (gp % RING) * PIECE;                                                                     \
    issue()
#define CH_PIECE_END() do { CH_WAIT_PIECES(RING - 2); CH_BARRIER(); ++gp; } while (0)
    float hmax[DN_CH_LAYERS];
#pragma unroll
    for (int j = 0; j < DN_CH_LAYERS; ++j) hmax[j] = 0.f;
    float omax = 0.f;
    // SG: the operand fragments of a pass -- [Phi | G_X Phi | G_Y Phi] of this wave's 16 rows, pre-split fp16 (hi, lo), this lane's eight
    // contraction slots of step T as one uint4 per plane (every request of the wave is 1 KiB contiguous) -- are requested a pass AHEAD, inside
    // the previous pass's layer 0, and carried in registers to the top of their pass
    [[maybe_unused]] uint4 fr[SG ? KE : 1][3][2];
    // (all KE steps carried across the pass cost 140 spilled registers -- the fragments went to scratch as they arrived; the first PFK steps are
    // carried, the rest are requested at the top of their pass and arrive under the first steps' products)
    constexpr int PFK = SG ? (DN_CH_SG_PF < KE ? DN_CH_SG_PF : KE) : 0;
    auto load_frags = [&](const int grp, const int t0, const int t1) {
        if constexpr (SG) {
            const uint4* fp = a.sg_pack + (size_t)grp * (3 * KE * 128) + lane;
#pragma unroll
            for (int T = 0; T < KE; ++T)
#pragma unroll
                for (int op = 0; op < 3; ++op)
                    if (T >= t0 && T < t1) {
                        fr[T][op][0] = fp[(size_t)(op * KE + T) * 128];
                        fr[T][op][1] = fp[(size_t)(op * KE + T) * 128 + 64];
                    }
        }
    };
    if constexpr (SG) load_frags(ch_uniform_i(pinfo[0].w) + wave, 0, PFK);

    for (int pass = 0; pass < npass; ++pass) {
        CH_TR();
        const int unit = unit_of(pass);
        int rb = unit * (16 * HH * NW) + 16 * HH * wave;     // first of this wave's 32 rows (row * C fits 32 bits for every batch the library takes)
        int row_end = a.V;
        [[maybe_unused]] float u_xd = 1.f, u_gx = 1.f, u_gy = 1.f;
        [[maybe_unused]] int grp_nx = 0, grp_cur = 0;
        if constexpr (SG) {
            // this pass's rows (of one mesh; past the mesh's end the packed operands hold zeros) and the next pass's mesh and group, which the piece
            // requests and the operand requests that run ahead need
            const int4 pi_ = pinfo[pass];
            const float4 ps_ = pscale[pass];
            const int4 pn_ = pinfo[pass + 1 < npass ? pass + 1 : pass];
            rb = ch_uniform_i(pi_.x) + 16 * HH * wave;
            row_end = ch_uniform_i(pi_.y);
            mesh_nx = ch_uniform_i(pn_.z);
            grp_nx = ch_uniform_i(pn_.w) + HH * wave;
            grp_cur = ch_uniform_i(pi_.w) + HH * wave;
            u_xd = ch_uniform(ps_.x); u_gx = ch_uniform(ps_.y); u_gy = ch_uniform(ps_.z);
        }
        int rowh[HH]; bool liveh[HH]; int rch[HH];
        int begh[HH], endh[HH];
#pragma unroll
        for (int hh = 0; hh < HH; ++hh) {
            rowh[hh] = rb + 16 * hh + m;
            liveh[hh] = rowh[hh] < row_end;
            rch[hh] = liveh[hh] ? rowh[hh] : row_end - 1;      // dead rows repeat the last row (computed, never stored)
            if (!SG && !PRE && with_grad_) { begh[hh] = a.rowptr[rch[hh]]; endh[hh] = a.rowptr[rch[hh] + 1]; }
        }

        // =================================================== gradient features, one 16-row half at a time
        // G0 (the C = 256 form): the layer-0 product of a half's tanh features follows them at once, inside the half loop -- their operand
        // fragments then live for eight pieces instead of across the other half's whole stage (64 registers of a stage that holds gx, gy and
        // four accumulators; the general registers of a wave end at 256 whatever the SIMD has free, the accumulators live in the other 256)
        dn_f32x4 acc[HH][NT];               // MiniMLP accumulators of the two halves
        // G0: an accumulator row STARTS at its layer's bias in the units of the product (b / so, exact: so is a power of two) -- one batch of
        // bias requests per product, one latency, and epilogues that are register arithmetic only (the biases have no LDS there)
        auto bias_init = [&](dn_f32x4 (&row)[NT], const float* bp, const float sc) {
            float4 bq[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bq[nt] = *reinterpret_cast<const float4*>(bp + 16 * nt);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) row[nt] = dn_f32x4{bq[nt].x * sc, bq[nt].y * sc, bq[nt].z * sc, bq[nt].w * sc};
        };
        const float sc0 = s_in * ch_pow2_inv(sw_inv[0]);      // 1 / (output scale of layer 0)
        if constexpr (G0) {
            if (!with_grad_) {       // (with gradient features each half's row starts where its g-segment product does)
#pragma unroll
                for (int hh = 0; hh < HH; ++hh) bias_init(acc[hh], a.bias[0] + 4 * q, sc0);
            }
        } else {
#pragma unroll
            for (int hh = 0; hh < HH; ++hh)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[hh][nt] = dn_f32x4{0.f, 0.f, 0.f, 0.f};
        }
        uint4 gfh[G0 ? 1 : HH][NK], gfl[G0 ? 1 : HH][NK];     // tanh features (of the two halves) as operand fragments (hi / lo planes) for layer 0
        [[maybe_unused]] float xdv[(SG && !G0) ? NT : 1][4];   // SG, C <= 128: this wave's xd rows (accumulator layout), layer 0's third operand segment
        if (with_grad_) {
            auto half = [&](const int hh) __attribute__((always_inline)) {
                const long long row = hh ? rowh[HH - 1] : rowh[0];
                const bool live = hh ? liveh[HH - 1] : liveh[0];
                // ---- CSR gather of the row: gx = sum_j vx_j xd[col_j], gy likewise (entry order, fmaf: bit for bit spmm_kernel's sums)
                float gxv[NT][4], gyv[NT][4];
                [[maybe_unused]] int beg = 0, end = 0, nmax = 0;
                if constexpr (!SG && !PRE) {
                    beg = hh ? begh[HH - 1] : begh[0]; end = hh ? endh[HH - 1] : endh[0];
                    nmax = (int)ch_wave_max((float)(end - beg));
                }
                if constexpr (PRE) {
                    // ---- two-launch spectral form: gx, gy of these rows were computed by spectral_apply_kernel in this very layout
                    const long long rc_ = hh ? rch[HH - 1] : rch[0];
                    const float* px = a.gx + rc_ * C + 4 * q;
                    const float* py = a.gy + rc_ * C + 4 * q;
                    float4 tx[NT], ty[NT];
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) { tx[nt] = *reinterpret_cast<const float4*>(px + 16 * nt); ty[nt] = *reinterpret_cast<const float4*>(py + 16 * nt); }
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        gxv[nt][0] = tx[nt].x; gxv[nt][1] = tx[nt].y; gxv[nt][2] = tx[nt].z; gxv[nt][3] = tx[nt].w;
                        gyv[nt][0] = ty[nt].x; gyv[nt][1] = ty[nt].y; gyv[nt][2] = ty[nt].z; gyv[nt][3] = ty[nt].w;
                    }
                } else if constexpr (SG && !G0) {
                    // ---- [xd | gx | gy] = [Phi | G_X Phi | G_Y Phi][16 rows] ys[mesh]: the operand fragments arrive pre-split (fp16 hi / lo, this
                    // lane's eight contraction slots of step T as one uint4 per plane: every request of the wave is 1 KiB contiguous), all of a pass's
                    // requested up front; the spectrum's pieces come through the ring.  One piece read from LDS feeds all three products.
                    // (fr: steps 0 .. PFK - 1 requested a pass ahead, in front of the previous pass's hidden layers -- before the loop for the first pass)
                    load_frags(grp_cur, PFK, KE);
                    CH_TR();
                    dn_f32x4 sa[3][NT];
#pragma unroll
                    for (int op = 0; op < 3; ++op)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) sa[op][nt] = dn_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int T = 0; T < KE; ++T) {
                        CH_PIECE_BEGIN();
                        CH_MMA3(sa, fr[T]);
                        CH_PIECE_END();
                    }
                    CH_TR();
                    float wx = 0.f, wg = 0.f;
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float vd = sa[0][nt][e] * u_xd, vx_ = sa[1][nt][e] * u_gx, vy_ = sa[2][nt][e] * u_gy;
                            xdv[nt][e] = vd; gxv[nt][e] = vx_; gyv[nt][e] = vy_;
                            wx = fabsf(vd) > wx ? fabsf(vd) : wx;
                            wg = fabsf(vx_) > wg ? fabsf(vx_) : wg;
                            wg = fabsf(vy_) > wg ? fabsf(vy_) : wg;
                        }
                    wx = ch_wave_max(wx); wg = ch_wave_max(wg);
                    xdmax = wx > xdmax ? wx : xdmax; gmax = wg > gmax ? wg : gmax;
                    s_in = ch_uniform(dn_pow2_scale(fmaxf(fmaxf(x_mag, wx), 1.f)));        // [g | x | xd] of this pass
                    s_gf = ch_uniform(dn_pow2_scale(wg));
                    so_gf = ch_uniform(ch_pow2_inv(s_gf) * swa_inv);
                    // (whole-line stores, dn_chain_tiles.h: every lane takes part -- a lane also writes for row m +- 8 -- so no per-lane guard here)
                    if (a.xd_out) ch_st_tiles<NT, HH == 1>(a.xd_out, C, row, row_end, m, q, [&](const int nt) { return make_float4(xdv[nt][0], xdv[nt][1], xdv[nt][2], xdv[nt][3]); });
                    if (a.gx) {
                        ch_st_tiles<NT, HH == 1>(a.gx, C, row, row_end, m, q, [&](const int nt) { return make_float4(gxv[nt][0], gxv[nt][1], gxv[nt][2], gxv[nt][3]); });
                        ch_st_tiles<NT, HH == 1>(a.gy, C, row, row_end, m, q, [&](const int nt) { return make_float4(gyv[nt][0], gyv[nt][1], gyv[nt][2], gyv[nt][3]); });
                    }
                } else if constexpr (!RCG) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int e = 0; e < 4; ++e) { gxv[nt][e] = 0.f; gyv[nt][e] = 0.f; }
                    int cj[GCH]; float wx[GCH], wy[GCH];
                    auto entries = [&](int j0) {      // pattern entries j0 .. j0 + GCH - 1 of the row (past its end: entry 0 with weight 0)
#pragma unroll
                        for (int u = 0; u < GCH; ++u) {
                            const int idx = beg + j0 + u;
                            const bool in = idx < end;
                            const int ii = in ? idx : 0;
                            cj[u] = a.col[ii];
                            wx[u] = in ? a.vx[ii] : 0.f;
                            wy[u] = in ? a.vy[ii] : 0.f;
                        }
                    };
                    entries(0);
                    for (int j0 = 0; j0 < nmax; j0 += GCH) {
                        float4 v[GCH][NT];
                        float cx[GCH], cy[GCH];
#pragma unroll
                        for (int u = 0; u < GCH; ++u) {
                            const float* src = a.xd + (long long)cj[u] * C + 4 * q;
                            cx[u] = wx[u]; cy[u] = wy[u];
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt) v[u][nt] = *reinterpret_cast<const float4*>(src + 16 * nt);
                        }
                        entries(j0 + GCH);            // the next step's entries travel under this step's row pieces
#pragma unroll
                        for (int u = 0; u < GCH; ++u)
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt) {
                                gxv[nt][0] = fmaf(cx[u], v[u][nt].x, gxv[nt][0]); gyv[nt][0] = fmaf(cy[u], v[u][nt].x, gyv[nt][0]);
                                gxv[nt][1] = fmaf(cx[u], v[u][nt].y, gxv[nt][1]); gyv[nt][1] = fmaf(cy[u], v[u][nt].y, gyv[nt][1]);
                                gxv[nt][2] = fmaf(cx[u], v[u][nt].z, gxv[nt][2]); gyv[nt][2] = fmaf(cy[u], v[u][nt].z, gyv[nt][2]);
                                gxv[nt][3] = fmaf(cx[u], v[u][nt].w, gxv[nt][3]); gyv[nt][3] = fmaf(cy[u], v[u][nt].w, gyv[nt][3]);
                            }
                    }
                    if (a.gx) {
                        ch_st_tiles<NT, HH == 1>(a.gx, C, row, row_end, m, q, [&](const int nt) { return make_float4(gxv[nt][0], gxv[nt][1], gxv[nt][2], gxv[nt][3]); });
                        ch_st_tiles<NT, HH == 1>(a.gy, C, row, row_end, m, q, [&](const int nt) { return make_float4(gyv[nt][0], gyv[nt][1], gyv[nt][2], gyv[nt][3]); });
                    }
                } else {
                    // The loads run in the ROW-CONTIGUOUS lane mapping (LPR lanes x 16 bytes = one row, RPI rows per instruction): the texture path
                    // serves ~13 B/clk/CU when every lane of a quarter-wave sits on a different row -- the operand layout -- and twice that when a
                    // quarter-wave reads 256 contiguous bytes (tools/experiments/gather_rate).  The sums -- the same fmaf chains in entry order,
                    // bit for bit -- are then transposed into the operand layout (lane (m, q): row m, channels 16 nt + 4 q ..) through a
                    // wave-private 4-row LDS slice.  All NI row groups of a step are in flight together (NI x GCH requests of 1 KiB per wave: the
                    // form that did not fit the register file with two halves per wave, round 4).
#pragma unroll
                    for (int sw = 0; sw < NSW; ++sw) {
                        float4 bx[NI], by[NI];            // sums of row RPI * i + rsub of the half, channels 4 cc .. 4 cc + 3
#pragma unroll
                        for (int i = 0; i < NI; ++i) { bx[i] = make_float4(0.f, 0.f, 0.f, 0.f); by[i] = make_float4(0.f, 0.f, 0.f, 0.f); }
                        auto entries = [&](int j0, int (&cjv)[GCR], float (&wxv)[GCR], float (&wyv)[GCR]) {   // entries j0 .. j0 + GCR - 1 of this lane's row (past its end: entry 0, weight 0)
#pragma unroll
                            for (int u = 0; u < GCR; ++u) {
                                const int idx = beg + j0 + u;
                                const bool in = idx < end;
                                const int ii = in ? idx : 0;
                                cjv[u] = a.col[ii];
                                wxv[u] = in ? a.vx[ii] : 0.f;
                                wyv[u] = in ? a.vy[ii] : 0.f;
                            }
                        };
                        auto request = [&](const int (&cjv)[GCR], float4 (&v)[NI][GCR]) {
#pragma unroll
                            for (int i = 0; i < NI; ++i) {
                                const int sl = RPI * i + rsub;          // lane (m = that row, q = 0) holds the row's entries
#pragma unroll
                                for (int u = 0; u < GCR; ++u) {
                                    const int c = ch_shfl_i(cjv[u], sl);
                                    v[i][u] = *reinterpret_cast<const float4*>(a.xd + (long long)c * C + GC * sw + 4 * cc);
                                }
                            }
                        };
                        auto add = [&](const float (&wxv)[GCR], const float (&wyv)[GCR], const float4 (&v)[NI][GCR]) {
#pragma unroll
                            for (int i = 0; i < NI; ++i) {
                                const int sl = RPI * i + rsub;
#pragma unroll
                                for (int u = 0; u < GCR; ++u) {
                                    const float cx = __shfl(wxv[u], sl, 64), cy = __shfl(wyv[u], sl, 64);
                                    const float4 t = v[i][u];
                                    bx[i].x = fmaf(cx, t.x, bx[i].x); by[i].x = fmaf(cy, t.x, by[i].x);
                                    bx[i].y = fmaf(cx, t.y, bx[i].y); by[i].y = fmaf(cy, t.y, by[i].y);
                                    bx[i].z = fmaf(cx, t.z, bx[i].z); by[i].z = fmaf(cy, t.z, by[i].z);
                                    bx[i].w = fmaf(cx, t.w, bx[i].w); by[i].w = fmaf(cy, t.w, by[i].w);
                                }
                            }
                        };
                        // (two row buffers -- the rows of step k + 1 requested before the sums of step k are formed -- were measured: 85 spilled
                        // registers, block forward 471 vs 410 us; profiles/r05_rcg_ab.txt)
                        if constexpr (DN_CH_ROLL == 2 || (DN_CH_ROLL == 1 && G0)) {
                            // Rolling form, same registers: row group i of step k is summed as soon as ITS request has landed (requests return in
                            // order: the other NI - 1 stay in flight) and its buffer is re-requested for step k + 1 at once -- NI requests in flight
                            // all the time instead of a sawtooth between NI and none.  The pattern entries run two steps ahead.
                            int cj[GCR], cjn[GCR]; float wx[GCR], wy[GCR], wxn[GCR], wyn[GCR];
                            float4 v[NI][GCR];
                            entries(0, cj, wx, wy);
                            entries(GCR, cjn, wxn, wyn);
                            request(cj, v);
                            for (int j0 = 0; j0 < nmax; j0 += GCR) {
                                int cj2[GCR]; float wx2[GCR], wy2[GCR];
                                entries(j0 + 2 * GCR, cj2, wx2, wy2);
                                const bool more = j0 + GCR < nmax;      // (wave-uniform)
#pragma unroll
                                for (int i = 0; i < NI; ++i) {
                                    const int sl = RPI * i + rsub;
#pragma unroll
                                    for (int u = 0; u < GCR; ++u) {
                                        const float cx = __shfl(wx[u], sl, 64), cy = __shfl(wy[u], sl, 64);
                                        const float4 t = v[i][u];
                                        bx[i].x = fmaf(cx, t.x, bx[i].x); by[i].x = fmaf(cy, t.x, by[i].x);
                                        bx[i].y = fmaf(cx, t.y, bx[i].y); by[i].y = fmaf(cy, t.y, by[i].y);
                                        bx[i].z = fmaf(cx, t.z, bx[i].z); by[i].z = fmaf(cy, t.z, by[i].z);
                                        bx[i].w = fmaf(cx, t.w, bx[i].w); by[i].w = fmaf(cy, t.w, by[i].w);
                                    }
                                    if (more) {
#pragma unroll
                                        for (int u = 0; u < GCR; ++u) {
                                            const int c = ch_shfl_i(cjn[u], sl);
                                            v[i][u] = *reinterpret_cast<const float4*>(a.xd + (long long)c * C + GC * sw + 4 * cc);
                                        }
                                    }
                                }
#pragma unroll
                                for (int u = 0; u < GCR; ++u) { cj[u] = cjn[u]; wx[u] = wxn[u]; wy[u] = wyn[u]; cjn[u] = cj2[u]; wxn[u] = wx2[u]; wyn[u] = wy2[u]; }
                            }
                        } else {
                            int cj[GCR], cjn[GCR]; float wx[GCR], wy[GCR], wxn[GCR], wyn[GCR];
                            entries(0, cjn, wxn, wyn);
                            for (int j0 = 0; j0 < nmax; j0 += GCR) {
#pragma unroll
                                for (int u = 0; u < GCR; ++u) { cj[u] = cjn[u]; wx[u] = wxn[u]; wy[u] = wyn[u]; }
                                float4 v[NI][GCR];
                                request(cj, v);
                                entries(j0 + GCR, cjn, wxn, wyn);       // the next step's entries travel under this step's rows
                                add(wx, wy, v);
                            }
                        }
                        if (a.gx) {                       // saved for the backward: whole rows per half-wave
#pragma unroll
                            for (int i = 0; i < NI; ++i) {
                                const int rr = rb + 16 * hh + RPI * i + rsub;
                                if (rr < a.V) {
                                    ch_st4(a.gx + (long long)rr * C + GC * sw + 4 * cc, bx[i]);
                                    ch_st4(a.gy + (long long)rr * C + GC * sw + 4 * cc, by[i]);
                                }
                            }
                        }
                        // row-contiguous -> operand layout, SROWS rows at a time (chunk c of slice row r sits at c ^ 4 r -- c ^ r in the 16-row slice: the
                        // 16 lanes of a read land on 16 different 16-byte bank groups)
                        constexpr int SWZ = SROWS == 16 ? 1 : 4;
#pragma unroll
                        for (int t = 0; t < 2; ++t)
#pragma unroll
                            for (int sq = 0; sq < 16 / SROWS; ++sq) {
#pragma unroll
                                for (int k = 0; k < SROWS / RPI; ++k) {
                                    const int r = RPI * k + rsub;
                                    slice[r * LPR + (cc ^ (SWZ * r))] = t ? by[sq * (SROWS / RPI) + k] : bx[sq * (SROWS / RPI) + k];
                                }
                                ch_wave_sync();
                                if (SROWS == 16 || (m / SROWS) == sq) {
                                    const int r = m % SROWS;
#pragma unroll
                                    for (int nt = 0; nt < GNT; ++nt) {
                                        const float4 f = slice[r * LPR + ((4 * nt + q) ^ (SWZ * r))];
                                        if (t) { gyv[GNT * sw + nt][0] = f.x; gyv[GNT * sw + nt][1] = f.y; gyv[GNT * sw + nt][2] = f.z; gyv[GNT * sw + nt][3] = f.w; }
                                        else   { gxv[GNT * sw + nt][0] = f.x; gxv[GNT * sw + nt][1] = f.y; gxv[GNT * sw + nt][2] = f.z; gxv[GNT * sw + nt][3] = f.w; }
                                    }
                                }
                                ch_wave_sync();
                            }
                    }
                    }
                CH_TR();
                // ---- Bre = gx A_re^T - gy A_im^T, Bim = gy A_re^T + gx A_im^T  (layers.py:122-123; without rotations Bre = gx A^T, Bim = gy A^T)
                dn_f32x4 ga[2][NT];       // [0] = Bre, [1] = Bim of this half
#pragma unroll
                for (int o = 0; o < 2; ++o)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) ga[o][nt] = dn_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int T = 0; T < NK; ++T) {
                    uint4 fxh, fxl, fyh, fyl;
                    ch_split8(gxv[2 * T], gxv[2 * T + 1], s_gf, fxh, fxl);
                    ch_split8(gyv[2 * T], gyv[2 * T + 1], s_gf, fyh, fyl);
                    {   // A_re (or A): Bre += A gx, Bim += A gy
                        CH_PIECE_BEGIN();
                        if constexpr (C >= 256) CH_MMA2_PAIR(ga, fxh, fxl, fyh, fyl); else CH_MMA2_LEAN(ga, fxh, fxl, fyh, fyl);
                        CH_PIECE_END();
                    }
                    if (a.with_rot) {   // A_im: Bre -= A_im gy, Bim += A_im gx
                        const uint4 nyh = make_uint4(fyh.x ^ 0x80008000u, fyh.y ^ 0x80008000u, fyh.z ^ 0x80008000u, fyh.w ^ 0x80008000u);
                        const uint4 nyl = make_uint4(fyl.x ^ 0x80008000u, fyl.y ^ 0x80008000u, fyl.z ^ 0x80008000u, fyl.w ^ 0x80008000u);
                        CH_PIECE_BEGIN();
                        if constexpr (C >= 256) CH_MMA2_PAIR(ga, nyh, nyl, fxh, fxl); else CH_MMA2_LEAN(ga, nyh, nyl, fxh, fxl);
                        CH_PIECE_END();
                    }
                }
                CH_TR();
                // ---- g = tanh(gx * Bre + gy * Bim)   (layers.py:128-130), saved tensors, operand fragments for layer 0
                if constexpr (G0) {
                    // tile pair by tile pair, every value stored where it is produced: the scaled accumulators (general registers once they are
                    // multiplied) never exist all at once
#pragma unroll
                    for (int T = 0; T < NK; ++T) {
                        float gv2[2][4];
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            const int nt = 2 * T + u;
                            float br[4], bi[4];
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                br[e] = ga[0][nt][e] * so_gf;
                                bi[e] = ga[1][nt][e] * so_gf;
                                gv2[u][e] = ch_tanh(ch_dot2(gxv[nt][e], br[e], gyv[nt][e], bi[e]));
                            }
                            if (live && a.g) ch_st4(a.g + row * C + 4 * q + 16 * nt, make_float4(gv2[u][0], gv2[u][1], gv2[u][2], gv2[u][3]));
                            if (live && a.bre) {
                                ch_st4(a.bre + row * C + 4 * q + 16 * nt, make_float4(br[0], br[1], br[2], br[3]));
                                ch_st4(a.bim + row * C + 4 * q + 16 * nt, make_float4(bi[0], bi[1], bi[2], bi[3]));
                            }
                        }
                        ch_split8(gv2[0], gv2[1], s_in, gfh[0][T], gfl[0][T]);
                    }
                } else {
                    float gv[NT][4];
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            ga[0][nt][e] *= so_gf;                   // Bre, Bim (the accumulator registers keep them for the stores below)
                            ga[1][nt][e] *= so_gf;
                            gv[nt][e] = ch_tanh(ch_dot2(gxv[nt][e], ga[0][nt][e], gyv[nt][e], ga[1][nt][e]));
                        }
                    if (a.g) ch_st_tiles<NT, HH == 1>(a.g, C, row, row_end, m, q, [&](const int nt) { return make_float4(gv[nt][0], gv[nt][1], gv[nt][2], gv[nt][3]); });
                    if (a.bre) {
                        ch_st_tiles<NT, HH == 1>(a.bre, C, row, row_end, m, q, [&](const int nt) { return make_float4(ga[0][nt][0], ga[0][nt][1], ga[0][nt][2], ga[0][nt][3]); });
                        ch_st_tiles<NT, HH == 1>(a.bim, C, row, row_end, m, q, [&](const int nt) { return make_float4(ga[1][nt][0], ga[1][nt][1], ga[1][nt][2], ga[1][nt][3]); });
                    }
#pragma unroll
                    for (int T = 0; T < NK; ++T) ch_split8(gv[2 * T], gv[2 * T + 1], s_in, gfh[hh][T], gfl[hh][T]);
                }
                if constexpr (G0) {       // layer 0, g segment, this half (hh is a literal here: the G0 form calls this body once per half)
                    bias_init(acc[hh], a.bias[0] + 4 * q, sc0);
#pragma unroll
                    for (int T = 0; T < NK; ++T) {
                        CH_PIECE_BEGIN();
                        CH_MMA1(acc[hh], gfh[0][T], gfl[0][T]);
                        CH_PIECE_END();
                    }
                }
                CH_TR();
            };
            half(0);                  // both halves spelled out (hh a literal in each copy): every accumulator row and fragment set has its own registers
            if constexpr (HH > 1) half(HH - 1);
        }

        // =================================================== MiniMLP layer 0 on [g | x | xd] (the tanh features first: their fragments die here)
        {
            // operands of the 2 NK pieces of the x and xd segments, fetched two pieces ahead (requesting all of them up front in the
            // one-half form -- the registers would allow it -- measured no gain: 410 vs 400 us block forward, profiles/r05_rcg_ab.txt)
            constexpr int NXR = G0 ? DN_CH_NXR_WIDE : 3;   // (the requests share the in-order return queue with the piece stream: a row still on its way from HBM holds the pieces behind it)
            constexpr int NFETCH = (SG && !G0) ? NK : 2 * NK;       // pieces whose operand rows come from memory (SG, C <= 128: the xd segment is in registers)
            static_assert(NXR - 1 <= NFETCH, "operand prefetch depth");
            float4 nx[NXR][HH][2];
            auto fetch = [&](int pi, float4 (&d)[HH][2]) {
                const float* p = (pi < NK ? a.x : a.xd) + 32 * (pi < NK ? pi : pi - NK) + 4 * q;
#pragma unroll
                for (int hh = 0; hh < HH; ++hh) {
                    d[hh][0] = *reinterpret_cast<const float4*>(p + (long long)rch[hh] * C);
                    d[hh][1] = *reinterpret_cast<const float4*>(p + (long long)rch[hh] * C + 16);
                }
            };
#pragma unroll
            for (int pi = 0; pi < NXR - 1; ++pi) fetch(pi, nx[pi]);
            if constexpr (!G0) {
                if (with_grad_) {
#pragma unroll
                    for (int T = 0; T < NK; ++T) {
                        CH_PIECE_BEGIN();
                        CH_MMA2(acc, gfh[0][T], gfl[0][T], gfh[G0 ? 0 : HH - 1][T], gfl[G0 ? 0 : HH - 1][T]);
                        CH_PIECE_END();
                    }
                }
            }
#pragma unroll
            for (int pi = 0; pi < 2 * NK; ++pi) {
                uint4 fh[HH], fl[HH];
#pragma unroll
                for (int hh = 0; hh < HH; ++hh) {
                    if (SG && !G0 && pi >= NK) ch_split8(xdv[(2 * (pi - NK)) % NT], xdv[(2 * (pi - NK) + 1) % NT], s_in, fh[hh], fl[hh]);
                    else ch_split8(nx[pi % NXR][hh][0], nx[pi % NXR][hh][1], s_in, fh[hh], fl[hh]);
                }
                if constexpr (DN_CH_L0_EXACT == 0) {
                    if (pi + NXR - 1 < NFETCH) fetch(pi + NXR - 1, nx[(pi + NXR - 1) % NXR]);
                    CH_PIECE_BEGIN();
                    CH_MMA2(acc, fh[0], fl[0], fh[HH - 1], fl[HH - 1]);
                    CH_PIECE_END();
                } else {
                    // The row requests go out BEHIND the piece request of their piece, and the wait at the end of the piece counts them: the
                    // operations younger than the piece it needs are two piece requests and the row requests of this piece and the two before
                    // it (2 HH each) -- none of them has to land.  (With the rows requested first and a wait of two pieces flat, every piece
                    // of this stage waited for a row request one piece old: HBM latency against a piece time of a fraction of it.  The
                    // compiler's own wait before the split above still asks for everything but the youngest row requests -- pieces requested
                    // one piece ago included --: that one is not ours to count.)
                    CH_PIECE_BEGIN();
                    const bool f0 = pi + NXR - 1 < NFETCH;
                    if (f0) fetch(pi + NXR - 1, nx[(pi + NXR - 1) % NXR]);
                    CH_MMA2(acc, fh[0], fl[0], fh[HH - 1], fl[HH - 1]);
                    // row requests younger than the piece waited for: issued in pieces pi - 2, pi - 1, pi (the first two pieces of the stage
                    // count none: what precedes them differs by configuration, and fewer is the safe side)
                    const int kf = pi < 2 ? 0 : (int)f0 + (int)(pi - 1 + NXR - 1 < NFETCH) + (int)(pi - 2 + NXR - 1 < NFETCH);
                    if (kf == 3) CH_WAIT_OPS((RING - 2) * LPT + 3 * 2 * HH);
                    else if (kf == 2) CH_WAIT_OPS((RING - 2) * LPT + 2 * 2 * HH);
                    else if (kf == 1) CH_WAIT_OPS((RING - 2) * LPT + 1 * 2 * HH);
                    else CH_WAIT_OPS((RING - 2) * LPT);
                    CH_BARRIER();
                    ++gp;
                }
            }
        }
        CH_TR();
        // =================================================== hidden layers: h_j = dropout(relu(acc + b_j)) -> operand fragments -> next product
        //      (layers.py:143-160: the dropout in front of linear layer j + 1 is applied where h_j is produced)
        // SG: the next pass's operand fragments go out here -- layer 0's operands are dead, the hidden layers hold two 32-register sets -- and
        // have the rest of the pass to arrive (requested inside layer 0 they cost 140 spilled registers)
        const bool pf = SG && pass + 1 < npass;
        if constexpr (SG && DN_CH_SG_PFPOINT == 0) { if (pf) load_frags(grp_nx, 0, PFK); }
        float s_act = s_in;               // scale the operand of the product just finished was split with
        uint4 hfh[HH][NK], hfl[HH][NK];     // hidden activations as operand fragments
#pragma unroll 1
        for (int j = 0; j + 1 < a.n_mlp; ++j) {
            const float so = ch_pow2_inv(s_act) * (j == 0 ? sw_inv[0] : (j == 1 ? sw_inv[1] : sw_inv[2]));
            const float* bj = G0 ? (j == 0 ? a.bias[0] : (j == 1 ? a.bias[1] : a.bias[2])) + 4 * q : sbias + j * C + 4 * q;
            const uint8_t* mk = a.mask[j];
            const unsigned long long sd = a.seed[j] ? a.seed[j] + seed_add : 0ull;
            const float dscale = (mk || sd) ? 2.f : 1.f;
            float* hj = a.h[j];
            float wm = 0.f;
            if constexpr (G0) {
                // two sweeps over the accumulators -- the largest magnitude first (the split's scale), then tile pair by tile pair: value,
                // store, operand fragment -- so that h_j never exists as 128 general registers next to its 128 registers of fragments
                auto tile = [&](const int hh, const int nt, float (&t)[4]) {
                    unsigned kb = 0x01010101u;
                    if (mk) kb = *reinterpret_cast<const uint32_t*>(mk + (long long)rch[hh] * C + 16 * nt + 4 * q);
                    else if (sd) kb = dn_keep_bytes(dn_keep_bits(sd, rch[hh], 4 * nt + q, C / 4));
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float v = acc[hh][nt][e] * so;               // (the bias is in the accumulator)
                        v = v > 0.f ? v : 0.f;
                        t[e] = ((kb >> (8 * e)) & 0xffu) ? v * dscale : 0.f;
                    }
                };
#pragma unroll
                for (int hh = 0; hh < HH; ++hh)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        float t[4];
                        tile(hh, nt, t);
#pragma unroll
                        for (int e = 0; e < 4; ++e) wm = t[e] > wm ? t[e] : wm;
                    }
                wm = ch_wave_max(wm);
#pragma unroll
                for (int jj = 0; jj < DN_CH_LAYERS; ++jj) hmax[jj] = (jj == j && wm > hmax[jj]) ? wm : hmax[jj];
                s_act = ch_uniform(dn_pow2_scale(wm));
#pragma unroll
                for (int hh = 0; hh < HH; ++hh)
#pragma unroll
                    for (int T = 0; T < NK; ++T) {
                        float va[4], vb[4];
                        tile(hh, 2 * T, va);
                        tile(hh, 2 * T + 1, vb);
                        if (hj && liveh[hh]) {
                            float* oh = hj + (long long)rowh[hh] * C + 4 * q + 32 * T;
                            ch_st4(oh, make_float4(va[0], va[1], va[2], va[3]));
                            ch_st4(oh + 16, make_float4(vb[0], vb[1], vb[2], vb[3]));
                        }
                        ch_split8(va, vb, s_act, hfh[hh][T], hfl[hh][T]);
                    }
                {   // the next product's accumulators start at its bias
                    const float scn = s_act * ch_pow2_inv(j == 0 ? sw_inv[1] : (j == 1 ? sw_inv[2] : sw_inv[3]));
                    const float* bn = (j == 0 ? a.bias[1] : (j == 1 ? a.bias[2] : a.bias[3])) + 4 * q;
#pragma unroll
                    for (int hh = 0; hh < HH; ++hh) bias_init(acc[hh], bn, scn);
                }
            } else {
#pragma unroll
                for (int hh = 0; hh < HH; ++hh)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const float4 b4 = *reinterpret_cast<const float4*>(bj + 16 * nt);
                        const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
                        unsigned kb = 0x01010101u;
                        if (mk) kb = *reinterpret_cast<const uint32_t*>(mk + (long long)rch[hh] * C + 16 * nt + 4 * q);
                        else if (sd) kb = dn_keep_bytes(dn_keep_bits(sd, rch[hh], 4 * nt + q, C / 4));
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float t = acc[hh][nt][e] * so + bb[e];
                            t = t > 0.f ? t : 0.f;
                            t = ((kb >> (8 * e)) & 0xffu) ? t * dscale : 0.f;
                            acc[hh][nt][e] = t;                       // (the accumulator registers now hold h_j)
                            wm = t > wm ? t : wm;
                        }
                    }
                if (hj) {
#pragma unroll
                    for (int hh = 0; hh < HH; ++hh)
                        ch_st_tiles<NT, HH == 1>(hj, C, rowh[hh], row_end, m, q, [&](const int nt) { return make_float4(acc[hh][nt][0], acc[hh][nt][1], acc[hh][nt][2], acc[hh][nt][3]); });
                }
                wm = ch_wave_max(wm);
#pragma unroll
                for (int jj = 0; jj < DN_CH_LAYERS; ++jj) hmax[jj] = (jj == j && wm > hmax[jj]) ? wm : hmax[jj];   // (no dynamic register index)
                s_act = ch_uniform(dn_pow2_scale(wm));
#pragma unroll
                for (int hh = 0; hh < HH; ++hh)
#pragma unroll
                    for (int T = 0; T < NK; ++T) {
                        const float va[4] = {acc[hh][2 * T][0], acc[hh][2 * T][1], acc[hh][2 * T][2], acc[hh][2 * T][3]};
                        const float vb[4] = {acc[hh][2 * T + 1][0], acc[hh][2 * T + 1][1], acc[hh][2 * T + 1][2], acc[hh][2 * T + 1][3]};
                        ch_split8(va, vb, s_act, hfh[hh][T], hfl[hh][T]);
                    }
            }
            CH_TR();
            // ---- product of layer j + 1
            if constexpr (!G0) {
#pragma unroll
                for (int hh = 0; hh < HH; ++hh)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[hh][nt] = dn_f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int T = 0; T < NK; ++T) {
                CH_PIECE_BEGIN();
                CH_MMA2(acc, hfh[0][T], hfl[0][T], hfh[HH - 1][T], hfl[HH - 1][T]);
                // (SG: the 6 KE fragment requests issued in front of this loop are younger than the piece waited for -- DMA(gp + 1), out RING - 2
                // pieces ago -- during its first RING - 2 pieces: they may stay in flight; loads only are counted)
                if (SG && DN_CH_SG_PFPOINT == 0 && T < RING - 2 && j == 0 && pf) { CH_WAIT_OPS((RING - 2) * LPT + 6 * PFK); CH_BARRIER(); ++gp; }
                else CH_PIECE_END();
            }
            CH_TR();
        }
        // =================================================== last layer: out = acc + b + x   (layers.py:236-239)
        {
            // SG, DN_CH_SG_PFPOINT = 1: the next pass's operand fragments go out in front of the last epilogue (the hidden fragments are dead)
            if constexpr (SG && DN_CH_SG_PFPOINT == 1) { if (pf) load_frags(grp_nx, 0, PFK); }
            const int jl = a.n_mlp - 1;
            const float so = ch_pow2_inv(s_act) * (jl == 1 ? sw_inv[1] : (jl == 2 ? sw_inv[2] : sw_inv[3]));
            const float* bj = G0 ? (jl == 1 ? a.bias[1] : (jl == 2 ? a.bias[2] : a.bias[3])) + 4 * q : sbias + jl * C + 4 * q;
            if constexpr (G0) {
                // one half at a time: its x row requested whole (16 requests, one latency), result formed and stored (the bias is in the
                // accumulator)
#pragma unroll
                for (int hh = 0; hh < HH; ++hh) {
                    const float* px = a.x + (long long)rch[hh] * C + 4 * q;
                    float* oo = a.out + (long long)rowh[hh] * C + 4 * q;
                    float4 r[NT];
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) r[nt] = *reinterpret_cast<const float4*>(px + 16 * nt);
#if DN_CH_LINES_WIDE
                    (void)oo;
                    ch_st_tiles<NT, true>(a.out, C, rowh[hh], row_end, m, q, [&](const int nt) {
                        float4 y;
                        y.x = acc[hh][nt][0] * so + r[nt].x;
                        y.y = acc[hh][nt][1] * so + r[nt].y;
                        y.z = acc[hh][nt][2] * so + r[nt].z;
                        y.w = acc[hh][nt][3] * so + r[nt].w;
                        omax = dn_f4_amax(omax, y);
                        return y; });
#else
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        float4 y;
                        y.x = acc[hh][nt][0] * so + r[nt].x;
                        y.y = acc[hh][nt][1] * so + r[nt].y;
                        y.z = acc[hh][nt][2] * so + r[nt].z;
                        y.w = acc[hh][nt][3] * so + r[nt].w;
                        omax = dn_f4_amax(omax, y);
                        if (liveh[hh]) ch_st4(oo + 16 * nt, y);
                    }
#endif
                }
            } else {
                float4 r4[HH][NT];
#pragma unroll
                for (int hh = 0; hh < HH; ++hh) {
                    const float* px = a.x + (long long)rch[hh] * C + 4 * q;
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) r4[hh][nt] = *reinterpret_cast<const float4*>(px + 16 * nt);   // (x: second read, out of L2)
                }
#pragma unroll
                for (int hh = 0; hh < HH; ++hh)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const float4 b4 = *reinterpret_cast<const float4*>(bj + 16 * nt);
                        float4 y;
                        y.x = (acc[hh][nt][0] * so + b4.x) + r4[hh][nt].x;
                        y.y = (acc[hh][nt][1] * so + b4.y) + r4[hh][nt].y;
                        y.z = (acc[hh][nt][2] * so + b4.z) + r4[hh][nt].z;
                        y.w = (acc[hh][nt][3] * so + b4.w) + r4[hh][nt].w;
                        r4[hh][nt] = y;
                        omax = dn_f4_amax(omax, y);
                    }
#pragma unroll
                for (int hh = 0; hh < HH; ++hh)
                    ch_st_tiles<NT, HH == 1>(a.out, C, rowh[hh], row_end, m, q, [&](const int nt) { return r4[hh][nt]; });
            }
            CH_TR();
        }
    }
#undef CH_PIECE_END
#undef CH_PIECE_BEGIN
#ifndef DN_EMULATE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the requests that ran past the end of the stream
#endif
#undef CH_BARRIER
#undef CH_WAIT_OPS
#undef CH_WAIT_PIECES
    // ---- magnitude words of what was produced: one check-first atomic per WORKGROUP and word.  The persistent workgroups finish within a few
    //      microseconds of each other, most of them read the words before anybody has raised them, and atomics on one cache line retire at
    //      ~8 ns each: one per wave (2048 x 3) cost 13 us of a 345 us kernel, one per workgroup 5-6 us (measured against a build that leaves the words unwritten).  The waves of a workgroup arrive here together (same pass count,
    //      a barrier per piece); the piece ring is idle by now and lends its first words.
    {
        constexpr int NWORD = DN_CH_LAYERS + 3;                 // hidden layers, out, and (SG) xd, gx / gy
        float* wmax = reinterpret_cast<float*>(ring);          // [NW][NWORD]
        float vals[NWORD];
#pragma unroll
        for (int j = 0; j < DN_CH_LAYERS; ++j) vals[j] = ch_wave_max(hmax[j]);
        vals[DN_CH_LAYERS] = ch_wave_max(omax);
        vals[DN_CH_LAYERS + 1] = xdmax;                         // (already wave-uniform)
        vals[DN_CH_LAYERS + 2] = gmax;
        __syncthreads();          // every wave has waited for its last DMA requests (above): nothing lands in the ring any more
        if (lane == 0) {
#pragma unroll
            for (int j = 0; j < NWORD; ++j) wmax[wave * NWORD + j] = vals[j];
        }
        __syncthreads();
        if (tid < NWORD) {
            float mm = 0.f;
            for (int w = 0; w < NW; ++w) { const float t = wmax[w * NWORD + tid]; mm = t > mm ? t : mm; }
            float* word = tid == DN_CH_LAYERS ? a.out_amax : (tid < a.n_mlp - 1 ? a.h_amax[tid] : nullptr);
            if (tid == DN_CH_LAYERS + 1) word = SG ? a.xd_amax_out : nullptr;
            if (tid == DN_CH_LAYERS + 2) word = SG ? a.g_amax : nullptr;
            if (word && mm > 0.f && mm > *reinterpret_cast<volatile float*>(word)) atomicMax(reinterpret_cast<unsigned*>(word), __float_as_uint(mm));
        }
    }
    DN_CLK_STAMP(chain_fwd, 1);
}

// ---------------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------------
int dn_chain_pieces(int C, int with_grad, int with_rot, int n_mlp) {
    const int NK = C / 32;
    return (with_grad ? NK * (with_rot ? 2 : 1) : 0) + (with_grad ? 3 : 2) * NK + (n_mlp - 1) * NK;
}
size_t dn_chain_ws_bytes(int C, int with_grad, int with_rot, int n_mlp) {
    return (size_t)dn_chain_pieces(C, with_grad, with_rot, n_mlp) * (2 * (C / 16) * 64) * sizeof(uint4);
}
bool dn_chain_sg_eligible(int C, int K, int with_grad, int hh) {
    return with_grad && (((C == 128 || C == 64) && K == 128 && hh == 1) || (C == 256 && K == 256 && hh == 2));
}
bool dn_chain_eligible(int C, int n_mlp, const int* widths, int with_grad, long long g_nnz, int V, int backward) {
    if (C != 128 && C != 64 && !(C == 256 && !backward)) return false;
    if (n_mlp < 2 || n_mlp > DN_CH_LAYERS) return false;
    for (int j = 1; j <= n_mlp; ++j) if (widths[j] != C) return false;
    if (with_grad && g_nnz <= 0) return false;
    return V > 0;
}

template <int C, int NW, int HH, int KE = 0, int MODE = 0>
static int chain_launch_nw(ChainArgs a, hipStream_t stream) {
    a.units = KE > 0 ? a.sg_n_units * (a.sg_unit_rows / (16 * HH * NW)) : (a.V + 16 * HH * NW - 1) / (16 * HH * NW);
    // eight waves per CU (256 registers per lane each): two 4-wave workgroups or one 8-wave workgroup.  C = 256: one 4-wave workgroup per CU,
    // one wave per SIMD with the whole register file (the gradient-feature stage alone holds gx, gy and both accumulators: 256 registers)
    int g = (C >= 256 ? 1 : 8 / NW) * dn_num_cus();
    if (g > a.units) g = a.units;
    g = (g + 7) / 8 * 8;
    const size_t smem = C >= 256 ? (size_t)DN_CH_RING * (2 * (C / 16) * 64) * sizeof(uint4) + (KE > 0 ? (size_t)DN_CH_SG_MAXP * 32 : (size_t)NW * 16 * 128 * sizeof(float))    // piece ring + one 16-row, 128-channel slice per wave (KE > 0: the pass table)
                                 : (size_t)DN_CH_RING * (2 * (C / 16) * 64) * sizeof(uint4) + (size_t)DN_CH_LAYERS * C * sizeof(float) +
                                       (KE > 0 ? (size_t)DN_CH_SG_MAXP * 32 : (size_t)NW * 4 * C * sizeof(float));      // piece ring + biases + one 4-row gather slice per wave (KE > 0: the pass table)
    if (KE > 0 && ((a.units + 7) / 8 + g / 8 - 1) / (g / 8) > DN_CH_SG_MAXP) return 1;      // (never for a batch the dispatch sends here: <= 262144 rows on >= 8 workgroups)
#ifndef DN_EMULATE
    static unsigned long long lds_opt_in = 0;   // per-device bitmap
    { const int oe_ = dn_lds_opt_in(reinterpret_cast<const void*>(&chain_fwd_kernel<C, NW, HH, KE, MODE>), smem, &lds_opt_in); if (oe_) return oe_; }
#endif
    DN_LAUNCH((chain_fwd_kernel<C, NW, HH, KE, MODE>), dim3(g, 1, 1), dim3(64 * NW, 1, 1), smem, stream, a);
    return (int)hipGetLastError();
}
template <int C>
static int chain_launch(int npieces, const ChainArgs& a_in, hipStream_t stream, int hh) {
    // Waves per workgroup: every workgroup streams the whole weight set once per 16 * HH * NW rows, so large batches take four (two
    // workgroups per CU share its matrix pipes out of phase); only batches that would leave most CUs without a workgroup are cut finer
    // (measured, block forward at 7k / 20k / 160k vertices: NW = 1: 102 / 179 / 658 us, 2: 104 / 140 / 486, 4: 113 / 129 / 388).
    // HH = 16-row halves per wave: 2 for batches that fill the device (a weight fragment read from LDS feeds two MFMAs); 1 for small ones,
    // where a wave's serial chain of 18 C x C products over 32 rows IS the kernel's duration (one 7k-vertex mesh: 219 waves on 1024 SIMDs):
    // twice the waves, half the chain each.
    const int nw_env = dn_opt_chain_nw();   // (development override)
    int nw = nw_env;
    if (nw != 1 && nw != 2 && nw != 4 && nw != 8) {
        const int half = dn_num_cus() / 2;
        nw = 4;
        while (nw > 1 && (a_in.V + 16 * hh * nw - 1) / (16 * hh * nw) < half) nw >>= 1;
    }
    if (hh == 1 && nw > 4) nw = 4;
    ChainArgs a = a_in;
    a.n_pieces = npieces;
    if constexpr (C >= 256) {      // (one wave shape: BASELINE config 4 is a 200k-vertex mesh)
        if (!a.sg_pack) return chain_launch_nw<C, 4, 2>(a, stream);
        // spectral-gradient form at C = K = 256: spectral_apply_kernel (xd, gx, gy -> memory, their magnitudes -> xd_amax_out / g_amax), then the
        // chain reading them (plain rows, plain piece stream)
        const int e1 = dn_launch_spectral_apply(a, C, stream);
        if (e1) return e1;
        ChainArgs b = a;
        b.sg_pack = nullptr; b.sg_units = nullptr; b.sg_n_units = 0; b.ysp = nullptr;
        b.xd_amax = a.xd_amax_out;
        return chain_launch_nw<C, 4, 2, 0, 2>(b, stream);
    }
    else {
    if (a.sg_pack) {       // spectral-gradient form (dn_chain_sg_eligible: hh == 1, k_eig == 128)
        if (hh != 1) return 1;
        return nw == 4 ? chain_launch_nw<C, 4, 1, 4>(a, stream) : (nw == 2 ? chain_launch_nw<C, 2, 1, 4>(a, stream) : chain_launch_nw<C, 1, 1, 4>(a, stream));
    }
    if (hh == 1) return nw == 4 ? chain_launch_nw<C, 4, 1>(a, stream) : (nw == 2 ? chain_launch_nw<C, 2, 1>(a, stream) : chain_launch_nw<C, 1, 1>(a, stream));
    switch (nw) {
        case 8: return chain_launch_nw<C, 8, 2>(a, stream);
        case 2: return chain_launch_nw<C, 2, 2>(a, stream);
        case 1: return chain_launch_nw<C, 1, 2>(a, stream);
        default: return chain_launch_nw<C, 4, 2>(a, stream);
    }
    }
}

int dn_launch_chain_prep(const ChainPrepArgs& pa, int npieces, int C, hipStream_t stream) {
    if (npieces > DN_CH_MAX_PIECES || npieces <= 0 || (C != 128 && C != 256 && C != 64)) return 1;      // (before the timing bracket opens)
    ChainPrepArgs pb = pa;
    pb.npieces = npieces;
    dn_prof_begin(DN_K_SMALL, stream);
    if (C == 128) DN_LAUNCH((chain_prep_kernel<128>), dim3(npieces + 1, 1, 1), dim3(1024, 1, 1), 0, stream, pb);
    else if (C == 256) DN_LAUNCH((chain_prep_kernel<256>), dim3(npieces + 1, 1, 1), dim3(1024, 1, 1), 0, stream, pb);
    else if (C == 64) DN_LAUNCH((chain_prep_kernel<64>), dim3(npieces + 1, 1, 1), dim3(1024, 1, 1), 0, stream, pb);
    else return 1;
    dn_prof_end(DN_K_SMALL, stream, 0.0, 0.0);
    return (int)hipGetLastError();
}
int dn_launch_chain_fwd(int npieces, const ChainArgs& a, int C, hipStream_t stream, int hh) {
    if (npieces > DN_CH_MAX_PIECES || (hh != 1 && hh != 2)) return 1;
    if (a.sg_pack && C >= 256 && (!a.xd_out || a.xd != a.xd_out || !a.gx || !a.gy)) return 1;      // (the C = 256 spectral form reads xd, gx, gy back from the buffers it writes)
    dn_prof_begin(DN_K_CHAIN, stream);
    int err;
    if (C == 128) err = chain_launch<128>(npieces, a, stream, hh);
    else if (C == 256) err = chain_launch<256>(npieces, a, stream, hh);
    else if (C == 64) err = chain_launch<64>(npieces, a, stream, hh);
    else err = 1;
    {
        // algorithmic traffic: xd gathered once + x read once (+ once more for the residual: L2), every saved tensor written once
        // (spectral-gradient form: the three packed operands [V, 128] instead of xd, which becomes one more saved tensor; three more products)
        const double VC = 4.0 * (double)a.V * C;
        double nw = 1.0;                                     // out
        if (a.gx) nw += 2.0; if (a.bre) nw += 2.0; if (a.g) nw += 1.0;
        if (a.sg_pack && a.xd_out) nw += 1.0;
        for (int j = 0; j < DN_CH_LAYERS; ++j) if (j < a.n_mlp - 1 && a.h[j]) nw += 1.0;
        const double prod = (a.with_grad ? (a.with_rot ? 4.0 : 2.0) : 0.0) + (a.with_grad ? 3.0 : 2.0) + (a.n_mlp - 1);
        const double Ksg = a.sg_pack ? (C >= 256 ? 256.0 : 128.0) : 0.0;
        const double rd = a.sg_pack ? VC + 3.0 * 4.0 * (double)a.V * Ksg + (C >= 256 ? VC : 0.0) : 2.0 * VC;      // (C = 256: xd written and read back)
        dn_prof_end(DN_K_CHAIN, stream, 2.0 * (double)a.V * C * C * prod + 3.0 * 2.0 * (double)a.V * Ksg * C, rd + VC * nw + (a.sg_pack && C >= 256 && !a.gx ? VC : 0.0));
    }
    return err;
}
