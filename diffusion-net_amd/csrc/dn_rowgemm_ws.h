// dn_rowgemm_ws.h -- the wave-specialised persistent row GEMM (kernel template + launcher); included by dn_rowgemm_persist.hip, which
// chooses between it and the lock-step kernel.  Split out of that file in round 4 (no code change).
#pragma once
#include "dn_gemm_tiles.h"

// ---- wave-specialised persistent row GEMM (split-bf16, one output, >= 4 slices) --------------------------------------------
// Measured on the lock-step kernel above (linear C->C, 158k rows): the compute side alone (no global traffic) takes 36 us,
// the memory side alone (no MFMA / split / LDS reads) 35 us, the two together 52-56 us -- every wave ran the same phase at
// the same time and sat in the memory instructions it issued.  Here the eight waves of a workgroup have fixed roles:
//   waves 0-3 (one per SIMD): LDS fragment reads + MFMAs of a 64x64 sub-tile each, and parking the finished unit in LDS;
//   waves 4-11              : global prefetch of the next slice, split into bf16 planes, LDS writes, and the deferred
//                             epilogue of the parked unit (LDS read, auxiliary operands, float4 stores).
// One barrier per slice hands the slice buffer over.  A parked unit must be streamed out before the next one is parked at
// the end of the following unit's last slice, hence PPI = ceil(NP / (nsl - 1)) pieces per loader thread and slice.
#define DN_WS_LW 8                            // loader waves per workgroup (measured: 4 made the loaders the pole); all of them stream the parked unit out
#define DN_WS_LTHR (64 * DN_WS_LW)             // loader threads
#define DN_WS_PTHR DN_WS_LTHR
#define DN_WS_NP (128 * 128 / 4 / DN_WS_PTHR)   // float4 pieces per piece thread and unit

struct WsAux {
    float4 a0;
    uint32_t mk;
    float rs;
    long long off;
    int lds;      // float index of the piece in the parked unit
    bool ok;
};

// issue the auxiliary loads of one deferred piece (nothing here is used before the next slice iteration)
template <int MODE, bool FLAG, bool XMASK = false>
__device__ __forceinline__ void ws_aux_load(const RgArgs& g, unsigned long long seed, int piece, int lt, int row0, int nrows, int n0, WsAux& A) {
    const bool live = piece < DN_WS_NP;
    const int idx = lt + (live ? piece : 0) * DN_WS_PTHR;
    const int row = idx >> 5, c4 = idx & 31;
    const int col = n0 + 4 * c4;
    A.ok = live && row < nrows && col < g.N;
    A.lds = row * 128 + 4 * c4;
    const long long grow = row0 + (A.ok ? row : 0);
    const int ccol = A.ok ? col : 0;
    A.off = grow * g.ldo + ccol;
    const long long roff = grow * g.ldr + ccol;
    constexpr bool need_r0 = MODE == DN_EPI_BIAS_RESID || MODE == DN_EPI_MUL_DFAC || MODE == DN_EPI_ADD ||
                             MODE == DN_EPI_DTANH || MODE == DN_EPI_MASS_ADD;
    if (need_r0) A.a0 = *reinterpret_cast<const float4*>(g.r0 + roff);
    // drawn bits or (XMASK, compile time: the parity tests' explicit uint8 masks) a 4-byte mask load.  The round-2 form "mask ? load :
    // hash" compiled to a branch per piece with s_waitcnt vmcnt(0) at every join -- three full drains of the memory pipeline per slice.
    if (MODE == DN_EPI_BIAS_RELU && FLAG) {
        if constexpr (XMASK) A.mk = *reinterpret_cast<const uint32_t*>(g.mask + roff);
        else A.mk = dn_keep_bytes(dn_keep_bits(seed, grow, ccol >> 2, (g.N + 3) >> 2));
    }
    if (MODE == DN_EPI_MASS_ADD) A.rs = g.rowv[grow];
}

template <int MODE, bool FLAG>
__device__ __forceinline__ void ws_piece_out(const RgArgs& g, const float4& v, const float4& bias, const WsAux& A, float so, float& om) {
    PtPiece P;
    P.v = v;
    if (so != 1.f) P.v = dn_f4_scale(P.v, so);     // split-fp16 engine: exact power-of-two rescale of the product
    P.a0 = A.a0; P.bias = bias; P.mk = A.mk; P.rs = A.rs; P.off = A.off; P.ok = A.ok;
    om = dn_f4_amax(om, pt_piece_store<MODE, FLAG>(g, P));
}

// the cross products of one k16 step, product-major: consecutive MFMAs go to different accumulators.
// NP = 3 (split-bf16): mid*mid, hi*lo, lo*hi, hi*mid, mid*hi, hi*hi;  NP = 2 (split-fp16): hi*lo, lo*hi, hi*hi -- smallest terms first
template <int NP>
__device__ __forceinline__ void ws_mma(const X3Frags<2, 2, 1, NP>& F, int s, f32x16 (&acc)[1][2][2]) {
    constexpr int NPROD = NP == 3 ? 6 : 3;
    constexpr int PA[6] = {NP == 3 ? 1 : 0, NP == 3 ? 0 : 1, NP == 3 ? 2 : 0, 0, 1, 0};
    constexpr int PB[6] = {1, NP == 3 ? 2 : 0, 0, 1, 0, 0};
#pragma unroll
    for (int p = 0; p < NPROD; ++p)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                if constexpr (NP == 3) acc[0][mt][nt] = dn_mfma_bf16(F.a[s][PA[p]][mt], F.b[s][0][PB[p]][nt], acc[0][mt][nt]);
                else acc[0][mt][nt] = dn_mfma_f16(F.a[s][PA[p]][mt], F.b[s][0][PB[p]][nt], acc[0][mt][nt]);
            }
}

// loader-side fetch of one slice with every descriptor already in registers (no kernel-argument or tile-table loads on the
// path to the global loads: a dependent scalar load costs a few hundred cycles, and the lock-step kernel paid four per slice)
template <bool BCOLK, int A_IT, int B_IT, bool LOAD_A = true, bool LOAD_B = true>
__device__ __forceinline__ void ws_load(const float* ap, int ald, const float* bp, int ldb, int N, int row0, int nrows, int n0,
                                        int koff, int lt, RgRegs<1, A_IT, B_IT>& R) {
    constexpr int LTHR = DN_WS_LTHR;
    if (LOAD_A) {
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const int idx = lt + i * LTHR;
            const int row = idx >> 3, q = idx & 7;
            const long long off = (long long)(row0 + (row < nrows ? row : 0)) * ald + koff + 4 * q;
            R.a[i] = *reinterpret_cast<const float4*>(ap + off);
        }
    }
    if (!LOAD_B) return;
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
        const int idx = lt + i * LTHR;
        long long boff;
        if (BCOLK) {
            const int nrow = idx >> 3, q = idx & 7;
            boff = (long long)(n0 + nrow < N ? n0 + nrow : 0) * ldb + koff + 4 * q;
        } else {
            const int krow = 2 * (lt & 15) + (i & 1);
            const int q4 = (lt >> 4) + (LTHR / 16) * (i >> 1);
            boff = (long long)(koff + krow) * ldb + (n0 + 4 * q4 < N ? n0 + 4 * q4 : 0);
        }
        R.b[0][i] = *reinterpret_cast<const float4*>(bp + boff);
    }
}

// BC ("B cached"): products with ONE 128-wide segment (4 slices) and the same B for every unit (nn.Linear weights): every loader
// thread stages the same B elements of slice s for every unit, so it splits them once, before the loop, and keeps the 4 x 12
// plane dwords in registers -- the per-slice B work shrinks from 2 loads + 44 VALU + 6 LDS writes to the 6 LDS writes.
template <int MODE, bool BCOLK, bool FLAG, int PPI, bool BC, int NP, bool XMASK>
__global__ __launch_bounds__(256 + DN_WS_LTHR) DN_WAVES_PER_EU(3) void rowgemm_ws_kernel(RgArgs g, int ntiles) {
    const unsigned long long seed = (MODE == DN_EPI_BIAS_RELU && FLAG) ? rg_seed(g) : 0ull;

    constexpr int TN = 128, NOUT = 1, LTHR = DN_WS_LTHR;
    constexpr int A_IT = DN_TM * 8 / LTHR;            // 4 float4 of the A slice per loader thread
    constexpr int B_IT = DN_KB * TN / 4 / LTHR;       // 4 float4 of the B slice
    constexpr int SA = (DN_TM * 64 * 3) / 4;          // floats of the A planes of one slice (24 KiB)
    constexpr int SBUF = SA + (128 * 64 * 3) / 4;     // one (A,B) slice buffer (48 KiB); two in LDS + the parked unit (64 KiB)
    constexpr bool PAIRK = !BCOLK;
    constexpr bool HASQ = false;
    constexpr bool need_bias = (MODE == DN_EPI_STORE && FLAG) || MODE == DN_EPI_BIAS_RELU || MODE == DN_EPI_BIAS_RESID;

    DN_DYN_SMEM(smem_raw);
    float* smem = reinterpret_cast<float*>(smem_raw);
    float* sE = smem + 2 * SBUF;
    constexpr int WS_AMAX_LDS = DN_TM * 64 * 2 / 4;   // float index of two spare words: the third A plane of stage 0, unused by the 2-term engine
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int G = gridDim.x;
    const int n0 = blockIdx.y * TN;
    int nsl = 0;
    for (int s = 0; s < g.nseg; ++s) nsl += g.a[s].w / DN_KB;   // host guarantees nsl >= 4 and whole slices
    const int my_units = ((int)blockIdx.x < ntiles) ? (ntiles - (int)blockIdx.x + G - 1) / G : 0;
    const int T = my_units * nsl;
    if (T == 0) return;

    // Roles: waves 0-3 (one per SIMD) multiply, waves 4-11 load.
    const bool is_mfma = wave < 4;
    const int mw = wave, lw = wave - 4;
    if (is_mfma) {
        // ------------------------------------------------ MFMA waves ------------------------------------------------
        const int wr = mw >> 1, wc = mw & 1;
        const int li = lane & 31, lg = lane >> 5;
        f32x16 acc[1][2][2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[0][mt][nt][r] = 0.f;
        int cs = 0;
        __syncthreads();   // slice 0 staged
        for (int j = 0; j < T; ++j) {
            const unsigned char* cA = reinterpret_cast<const unsigned char*>(smem + (j & 1) * SBUF);
            const unsigned char* cB = cA + SA * 4;
            X3Frags<2, 2, 1, NP> F;
            rg_frag_x3<2, 2, 1, NP>(cA, cB, wr * 64, wc * 64, li, lg, 0, F);
            rg_frag_x3<2, 2, 1, NP>(cA, cB, wr * 64, wc * 64, li, lg, 1, F);
            ws_mma<NP>(F, 0, acc);
            ws_mma<NP>(F, 1, acc);
            if (++cs == nsl) {   // unit complete: park it (fragment layout -> row-major) for the loader waves to stream out
                cs = 0;
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            sE[(wr * 64 + mt * 32 + dn_acc_row(r, lane)) * 128 + wc * 64 + nt * 32 + li] = acc[0][mt][nt][r];
                            acc[0][mt][nt][r] = 0.f;
                        }
            }
            __syncthreads();
        }
        return;
    }

    // ---------------------------------------------------- loader waves ----------------------------------------------------
    const int lt = lw * 64 + lane;
    if (NP == 2 && lt < 2) reinterpret_cast<unsigned*>(smem + WS_AMAX_LDS)[lt] = 0u;   // workgroup-level magnitude commit (before the first barrier)
    // split-fp16: operand scales (powers of two from the producers' amax words) and the exact inverse of their product
    float sa = 1.f, sb = 1.f, so = 1.f, om = 0.f;   // om: running max |o0| of this thread's pieces
    if constexpr (NP == 2) {
        sa = dn_pow2_scale(dn_amax_eval(g.a_amax));
        sb = dn_pow2_scale(dn_amax_eval(g.b_amax));
        so = (1.f / sa) * (1.f / sb);
    }
    float4 bias = dn_f4_zero();
    {
        const int col = n0 + 4 * (lt & 31);
        if (need_bias) bias = *reinterpret_cast<const float4*>(g.bias + (col < g.N ? col : 0));
    }
    RgRegs<NOUT, A_IT, B_IT> R0;
    // segment descriptors in registers (nseg <= 3); the B operand of segment s starts koff = 0 again
    const float* sp0 = g.a[0].p; const float* sp1 = g.a[1].p; const float* sp2 = g.a[2].p;
    const int sl0 = g.a[0].ld, sl1 = g.a[1].ld, sl2 = g.a[2].ld;
    const int sw0 = g.a[0].w, sw1 = g.a[1].w, sw2 = g.a[2].w;
    const float* sb0 = g.b[0][0]; const float* sb1 = g.b[0][1]; const float* sb2 = g.b[0][2];
    const int nseg = g.nseg, ldb = g.ldb, Ncols = g.N;
    const long long bms = g.b_mesh_stride;
    // Tile descriptors are fetched per lane (every lane the same address) through a pointer the compiler cannot prove uniform: for a
    // uniform address it emits a vector load + v_readfirstlane, i.e. an s_waitcnt vmcnt(0) right behind the load -- a full drain of
    // the memory pipeline (slice prefetch included) in every iteration (seen in the ISA of round 2's kernel).  As per-lane values
    // they are waited for where they are used: one iteration later.  (An explicit s_load through inline asm is not an option: the
    // compiler copies the destination registers at the loop back-edge before the load has returned -- tried, wrong results.)
    const DnTile* tl = g.tiles;
#ifndef DN_EMULATE
    { int vz_; asm volatile("v_mov_b32 %0, 0" : "=v"(vz_)); tl += vz_; }
#endif
    // load cursor; the next unit's tile descriptor is fetched one unit ahead
    int lu = blockIdx.x, lseg = 0, lkoff = 0;
    DnTile ltile = tl[lu];
    DnTile ltile_next = tl[lu + G < ntiles ? lu + G : lu];
    // mirror of the compute cursor (which unit is parked when) and the parked unit being streamed out
    int cu = blockIdx.x, cs = 0;
    DnTile ctile = ltile, ctile_next = ltile_next;
    int p_row0 = ctile.row0, p_nrows = 0, p_next = DN_WS_NP;   // p_next >= NP: nothing pending
    WsAux AX[PPI];
#pragma unroll
    for (int k = 0; k < PPI; ++k) ws_aux_load<MODE, FLAG, XMASK>(g, seed, DN_WS_NP, lt, p_row0, p_nrows, n0, AX[k]);   // dead pieces

// one step of the load cursor without control flow or memory access on the path; past the last slice it stays put
#define WS_ADVANCE(commit)                                                                                              \
    do {                                                                                                                \
        const int cw_ = lseg == 0 ? sw0 : (lseg == 1 ? sw1 : sw2);                                                      \
        int nk_ = lkoff + DN_KB, ns_ = lseg, nu_ = lu;                                                                  \
        const bool se_ = nk_ >= cw_;                                                                                    \
        nk_ = se_ ? 0 : nk_;                                                                                            \
        ns_ = se_ ? ns_ + 1 : ns_;                                                                                      \
        const bool ue_ = ns_ >= nseg;                                                                                   \
        ns_ = ue_ ? 0 : ns_;                                                                                            \
        nu_ = ue_ ? nu_ + G : nu_;                                                                                      \
        const bool ok_ = (commit) && nu_ < ntiles;                                                                      \
        const bool sw_ = ok_ && ue_;                                                                                    \
        lkoff = ok_ ? nk_ : lkoff; lseg = ok_ ? ns_ : lseg; lu = ok_ ? nu_ : lu;                                        \
        ltile.row0 = sw_ ? ltile_next.row0 : ltile.row0; ltile.nrows = sw_ ? ltile_next.nrows : ltile.nrows;            \
        ltile.mesh = sw_ ? ltile_next.mesh : ltile.mesh;                                                                \
        ltile_next = tl[lu + G < ntiles ? lu + G : lu];   /* consumed at the next unit switch at the earliest */         \
    } while (0)
#define WS_LOAD(RS)                                                                                                     \
    ws_load<BCOLK, A_IT, B_IT, true, !BC>(lseg == 0 ? sp0 : (lseg == 1 ? sp1 : sp2), lseg == 0 ? sl0 : (lseg == 1 ? sl1 : sl2), \
                               (lseg == 0 ? sb0 : (lseg == 1 ? sb1 : sb2)) + (long long)ltile.mesh * bms, ldb, Ncols,    \
                               ltile.row0, ltile.nrows, n0, lkoff, lt, RS)
#define WS_STAGE(buf, RS, SIDX)                                                                                         \
    do {                                                                                                                \
        X3Planes<NOUT, A_IT, B_IT, NP> PLN;                                                                             \
        rg_split_x3<NOUT, BCOLK, HASQ, A_IT, B_IT, true, !BC, NP>(RS, PLN, sa, sb);                                     \
        if constexpr (BC) {                                                                                    \
            _Pragma("unroll") for (int i_ = 0; i_ < B_IT; ++i_)                                                         \
                _Pragma("unroll") for (int p_ = 0; p_ < NP; ++p_) PLN.b[0][i_][p_] = Bc[SIDX][i_][p_];                  \
        }                                                                                                               \
        rg_put_x3<LTHR, NOUT, BCOLK, A_IT, B_IT, NP, true, true>(reinterpret_cast<unsigned char*>(buf),                \
                                                 reinterpret_cast<unsigned char*>((buf) + SA), lt, PLN);                \
    } while (0)

// Order inside an iteration: stage -> deferred pieces (their operands were requested an iteration ago) -> operands of the
// next iteration's pieces -> slice prefetch.  (Measured: a second register set / fetching two slices ahead, and requesting
// the piece operands before the prefetch, were both slower -- 60/48/135 us vs 53/51/127 us for the NN, C->C, 3C->C products.)
#define WS_SPLIT(RS, SIDX, PLN)                                                                                         \
    do {                                                                                                                \
        rg_split_x3<NOUT, BCOLK, HASQ, A_IT, B_IT, true, !BC, NP>(RS, PLN, sa, sb);                                     \
        if constexpr (BC) {                                                                                    \
            _Pragma("unroll") for (int i_ = 0; i_ < B_IT; ++i_)                                                         \
                _Pragma("unroll") for (int p_ = 0; p_ < NP; ++p_) PLN.b[0][i_][p_] = Bc[SIDX][i_][p_];                  \
        }                                                                                                               \
    } while (0)
#define WS_PUT(buf, PLN)                                                                                                \
    rg_put_x3<LTHR, NOUT, BCOLK, A_IT, B_IT, NP, true, true>(reinterpret_cast<unsigned char*>(buf), reinterpret_cast<unsigned char*>((buf) + SA), lt, PLN)
#define WS_PIECES()                                                                                                     \
    do {                                                                                                                \
        {                                                                                                               \
            float4 pv_[PPI];           /* all LDS reads of the parked unit first, then the maths and the stores */       \
            _Pragma("unroll") for (int k = 0; k < PPI; ++k) pv_[k] = *reinterpret_cast<const float4*>(&sE[AX[k].lds]);  \
            _Pragma("unroll") for (int k = 0; k < PPI; ++k) ws_piece_out<MODE, FLAG>(g, pv_[k], bias, AX[k], so, om);       \
            p_next = (p_next + PPI < DN_WS_NP) ? p_next + PPI : DN_WS_NP;                                               \
            {   /* the MFMA waves park unit cu at the end of the iteration that multiplies its last slice */            \
                const bool park = ++cs == nsl;                                                                          \
                p_row0 = park ? ctile.row0 : p_row0; p_nrows = park ? ctile.nrows : p_nrows;                            \
                p_next = park ? 0 : p_next;                                                                      \
                cs = park ? 0 : cs;                                                                                     \
                cu = park ? cu + G : cu;                                                                                \
                ctile.row0 = park ? ctile_next.row0 : ctile.row0; ctile.nrows = park ? ctile_next.nrows : ctile.nrows;  \
                const int cn = cu + G < ntiles ? cu + G : ntiles - 1;                                                   \
                ctile_next = tl[cn];   /* consumed at the next park at the earliest */                                  \
            }                                                                                                           \
            _Pragma("unroll") for (int k = 0; k < PPI; ++k)                                                             \
                ws_aux_load<MODE, FLAG, XMASK>(g, seed, p_next + k, lt, p_row0, p_nrows, n0, AX[k]);                                 \
        }                                                                                                               \
    } while (0)

// Order inside an iteration (round 3): wait for slice j+1 -> split it into plane registers -> the registers it came in are free: request
// slice j+2 NOW -> LDS writes of slice j+1 -> deferred pieces -> operands of the next pieces -> barrier.  (The round-2 order requested
// at the END of the iteration: its s_memtime timeline, profiles/r03_ws_trace_*.txt, showed 1300-2100 of a loader's ~4600-5200 cycles per
// slice spent waiting for that request.  A second register set requesting two slices ahead, and the pieces ahead of the request, were
// measured and rejected: tools/experiments/rowgemm_ws_knobs/.)
#define WS_ITER(j, RS, SIDX)                                                                                            \
    do {                                                                                                                \
        float* nxt = smem + (((j) & 1) ^ 1) * SBUF;                                                                     \
        X3Planes<NOUT, A_IT, B_IT, NP> PLN;                                                                             \
        WS_SPLIT(RS, SIDX, PLN);       /* slice j+1 (the last iteration stages a stale copy nobody reads) */            \
        WS_ADVANCE((j) + 2 < T);                                                                                        \
        WS_LOAD(RS);                   /* slice j+2 */                                                                  \
        WS_PUT(nxt, PLN);                                                                                               \
        WS_PIECES();                                                                                                    \
        __syncthreads();                                                                                                \
    } while (0)

    uint2 Bc[BC ? 4 : 1][B_IT][NP];
    if constexpr (BC) {   // split the whole B strip of this workgroup once (4 slices of the one segment)
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            RgRegs<NOUT, A_IT, B_IT> Rb;
            ws_load<BCOLK, A_IT, B_IT, false, true>(sp0, sl0, sb0, ldb, Ncols, 0, 0, n0, DN_KB * s4, lt, Rb);
            X3Planes<NOUT, A_IT, B_IT, NP> Pb;
            rg_split_x3<NOUT, BCOLK, HASQ, A_IT, B_IT, false, true, NP>(Rb, Pb, sa, sb);
#pragma unroll
            for (int i = 0; i < B_IT; ++i)
#pragma unroll
                for (int p3 = 0; p3 < NP; ++p3) Bc[s4][i][p3] = Pb.b[0][i][p3];
        }
    }
    WS_LOAD(R0);
    WS_STAGE(smem, R0, 0);
    WS_ADVANCE(T > 1);
    WS_LOAD(R0);                       // slice 1
    __syncthreads();                   // slice 0 staged
    if constexpr (BC) {                // T is a multiple of 4: iteration j stages slice (j + 1) % 4 of its unit
        for (int j = 0; j < T; j += 4) {
            WS_ITER(j, R0, 1);
            WS_ITER(j + 1, R0, 2);
            WS_ITER(j + 2, R0, 3);
            WS_ITER(j + 3, R0, 0);
        }
    } else {
        for (int j = 0; j < T; ++j) WS_ITER(j, R0, 0);
    }
#undef WS_ITER
#undef WS_PIECES
#undef WS_PUT
#undef WS_SPLIT
#undef WS_STAGE
#undef WS_LOAD
#undef WS_ADVANCE
    // flush the last parked unit
    for (; p_next < DN_WS_NP; ++p_next) {
        WsAux A1;
        ws_aux_load<MODE, FLAG, XMASK>(g, seed, p_next, lt, p_row0, p_nrows, n0, A1);
        ws_piece_out<MODE, FLAG>(g, *reinterpret_cast<const float4*>(&sE[A1.lds]), bias, A1, so, om);
    }
    if (g.o_amax) {
        if constexpr (NP == 2) dn_amax_commit_group(g.o_amax, om, reinterpret_cast<unsigned*>(smem + WS_AMAX_LDS), DN_WS_LW);
        else dn_amax_commit<true>(g.o_amax, om);
    }
}

template <int MODE, bool BCOLK, bool FLAG, int PPI, bool BC, int NP, bool XMASK>
static int ws_launch_x(const RgArgs& g, int ntiles, hipStream_t stream) {
    const size_t smem = (size_t)(2 * (DN_TM * 64 * 3 + 128 * 64 * 3) + 128 * 128 * 4);   // 160 KiB
#ifndef DN_EMULATE
    static unsigned long long lds_opt_in = 0;   // per-device bitmap
    { const int oe_ = dn_lds_opt_in(reinterpret_cast<const void*>(&rowgemm_ws_kernel<MODE, BCOLK, FLAG, PPI, BC, NP, XMASK>), smem, &lds_opt_in); if (oe_) return oe_; }
#endif
    int gx = dn_num_cus();
    if (gx > ntiles) gx = ntiles;
    DN_LAUNCH((rowgemm_ws_kernel<MODE, BCOLK, FLAG, PPI, BC, NP, XMASK>), dim3(gx, (g.N + 127) / 128, 1), dim3(256 + DN_WS_LTHR, 1, 1), smem, stream, g, ntiles);
    return (int)hipGetLastError();
}

template <int MODE, bool BCOLK, bool FLAG, int PPI, bool BC, int NP = 3>
static int ws_launch(const RgArgs& g, int ntiles, hipStream_t stream) {
    if constexpr (MODE == DN_EPI_BIAS_RELU && FLAG) {
        if (g.mask) return ws_launch_x<MODE, BCOLK, FLAG, PPI, BC, NP, true>(g, ntiles, stream);
    }
    return ws_launch_x<MODE, BCOLK, FLAG, PPI, BC, NP, false>(g, ntiles, stream);
}

