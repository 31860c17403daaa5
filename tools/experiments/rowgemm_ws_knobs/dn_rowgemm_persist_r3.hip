// dn_rowgemm_persist.hip -- persistent forms of the row GEMM (see dn_rowgemm.hip for the product): the lock-step persistent
// kernel, the wave-specialised kernel (4 MFMA waves + 8 loader/epilogue waves; the default for one-output products with
// >= 4 slices) and the wave-specialised two-output kernel (parity-green, measured slower, off).
#include "dn_gemm_tiles.h"

#ifndef DN_RG_X3
#define DN_RG_X3 1   // keep in step with dn_rowgemm.hip (two-output split-bf16 path)
#endif


// =======================================================================================
// persistent single-output rowgemm (the heavy N >= 128 products)
//
// One 8-wave workgroup per CU walks its tiles (tile = blockIdx.x, += gridDim.x ...).  The slice pipeline runs
// ACROSS tile boundaries (the loads of the next tile's first slices are in flight under the current tile's last
// MFMAs), and a finished tile's accumulators are parked in a 64 KiB LDS staging area from which every thread
// streams float4 rows to HBM -- with the epilogue maths and coalesced float4 auxiliary loads -- while the MFMAs of
// the following tile run.  Only the very first prologue and the very last flush of a workgroup are exposed.
// LDS: 2 x 32 KiB slice buffers + 64 KiB staging = 128 KiB.
// =======================================================================================
#ifndef DN_PT_MAX_SLICES
#define DN_PT_MAX_SLICES 32  // up to K = 1024 (the 3C -> C MLP layer: 12 slices at C = 128, 24 at C = 256; the lock-step exact-f32 fallback took 829 us per launch at C = 256: cfg4 18.9 -> 20.3 M vertices/s)
#endif
// Work-unit geometry.  Measured on MI355X (K = N = 128 product, 158k rows): 128-row units with 8 waves and one
// workgroup per CU: 68 us; 64-row units with 4 waves and two workgroups per CU (2 x 80 KiB LDS): 71-75 us (twice the
// B-operand staging per MFMA); non-persistent kernel: 75-80 us.
#ifndef DN_PT_ROWS
#define DN_PT_ROWS 128
#endif
#define DN_PT_THREADS (4 * DN_PT_ROWS)   // 64x32 outputs per wave
#define DN_PT_NP (DN_PT_ROWS * 128 / 4 / DN_PT_THREADS)   // float4 pieces per thread per unit (8)

// FLAG: STORE -> a bias vector is added; BIAS_RELU -> a dropout keep-mask is applied; unused otherwise.
// phase 1 of a deferred piece: ISSUE the LDS read of the parked accumulators and the global loads of the auxiliary
// operands.  Branch-free, and nothing loaded here is used before the MFMA block (a use would put a vmcnt wait --
// which also waits for the slice prefetch issued just before -- in front of the MFMAs).
template <int MODE, bool FLAG>
__device__ __forceinline__ void pt_piece_load(const RgArgs& g, unsigned long long seed, const float* sE, int piece, int tid, int row0, int nrows,
                                              int n0, PtPiece& P) {
    const bool live = piece < DN_PT_NP;
    const int idx = tid + (live ? piece : 0) * DN_PT_THREADS;
    const int row = idx >> 5, c4 = idx & 31;
    const int col = n0 + 4 * c4;
    P.ok = live && row < nrows && col < g.N;
    const long long grow = row0 + (P.ok ? row : 0);
    const int ccol = P.ok ? col : 0;
    P.off = grow * g.ldo + ccol;
    const long long roff = grow * g.ldr + ccol;
    P.v = *reinterpret_cast<const float4*>(&sE[row * 128 + 4 * c4]);
    constexpr bool need_r0 = MODE == DN_EPI_BIAS_RESID || MODE == DN_EPI_MUL_DFAC || MODE == DN_EPI_ADD ||
                             MODE == DN_EPI_DTANH || MODE == DN_EPI_MASS_ADD;
    constexpr bool need_bias = (MODE == DN_EPI_STORE && FLAG) || MODE == DN_EPI_BIAS_RELU || MODE == DN_EPI_BIAS_RESID;
    if (need_r0) P.a0 = *reinterpret_cast<const float4*>(g.r0 + roff);
    if (need_bias) P.bias = *reinterpret_cast<const float4*>(g.bias + ccol);
    if (MODE == DN_EPI_BIAS_RELU && FLAG) {   // explicit mask or drawn bits, without a branch: the load goes to a valid address either way
        const uint32_t ld = *reinterpret_cast<const uint32_t*>(g.mask ? g.mask + roff : reinterpret_cast<const uint8_t*>(g.bias));
        P.mk = g.mask ? ld : dn_keep_bytes(dn_keep_bits(seed, grow, ccol >> 2, (g.N + 3) >> 2));
    }
    if (MODE == DN_EPI_MASS_ADD) P.rs = g.rowv[grow];
}

template <int MODE, bool BCOLK, bool FLAG, bool X3>
__global__ __launch_bounds__(DN_PT_THREADS) DN_WAVES_PER_EU(2) void rowgemm_persist_kernel(RgArgs g, int ntiles) {
    const unsigned long long seed = (MODE == DN_EPI_BIAS_RELU && FLAG) ? rg_seed(g) : 0ull;

    constexpr int TN = 128, WR = DN_PT_ROWS / 64, WC = 4, NOUT = 1, NTHR = DN_PT_THREADS, TMU = DN_PT_ROWS;
    constexpr int UPT = DN_TM / TMU;                 // work units per 128-row tile (1 or 2)
    constexpr int MT = TMU / (32 * WR);              // 2
    constexpr int NT = TN / (32 * WC);               // 1
    constexpr int A_IT = TMU * 8 / NTHR;             // 2
    constexpr int B_IT = DN_KB * TN / 4 / NTHR;      // 2 (128-row units) or 4
    // one (A,B) slice buffer, in floats: f32 tiles (A 4 B/elem + B 4 B/elem) or three bf16 planes each (6 B/elem)
    constexpr int SA = X3 ? (TMU * 64 * 3) / 4 : TMU * DN_KB;
    constexpr int SBUF = SA + (X3 ? (128 * 64 * 3) / 4 : DN_KB * TN);
    constexpr bool HASQ = false;
    constexpr bool PAIRK = X3 && !BCOLK;
    constexpr int PPI = 4;                           // deferred pieces per slice iteration: 8 pieces over 2 iterations
    static_assert(!X3 || (TMU == 128 && NTHR == 512), "the bf16x3 staging is written for 128-row units and 512 threads");
    static_assert(UPT * TMU == DN_TM && (UPT == 1 || UPT == 2), "a work unit is a whole or half a row tile");

    DN_DYN_SMEM(smem_raw);
    float* smem = reinterpret_cast<float*>(smem_raw);
    float* sE = smem + 2 * SBUF;                     // [TMU][128] parked accumulators

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave / WC, wc = wave % WC;
    const int li = lane & 31, ls = lane >> 5;
    const int G = gridDim.x;
    const int n0 = blockIdx.y * TN;
    const int nunits = UPT * ntiles;                 // unit u = rows [TMU*(u%UPT), +TMU) of tile u/UPT

    int nsl = 0;
    for (int s = 0; s < g.nseg; ++s) nsl += (g.a[s].w + DN_KB - 1) / DN_KB;   // host guarantees nsl >= 3
    const int my_units = ((int)blockIdx.x < nunits) ? (nunits - (int)blockIdx.x + G - 1) / G : 0;
    const int T = my_units * nsl;                    // slices this workgroup processes
    if (T == 0) return;

    f32x16 acc[NOUT][MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][mt][0][r] = 0.f;

    // unit -> (sub-)tile descriptor; an empty second half (tile shorter than 64 rows) yields nrows = 0 (fully masked)
    auto unit_tile = [&](int u) {
        DnTile t = g.tiles[u / UPT];
        const int h = (u % UPT) * TMU;
        int n = t.nrows - h;
        n = n < 0 ? 0 : (n > TMU ? TMU : n);
        t.row0 += (n > 0 ? h : 0);
        t.nrows = n;
        return t;
    };

    // 1 register sets form the prefetch ring: slice s travels in set s % DEPTH and is loaded DEPTH iterations
    // before it is written to LDS, so DEPTH x 16 KiB of reads per CU are in flight (one slice ahead covers only ~1 us of HBM
    // latency once the split-bf16 MFMAs made an iteration that short).
    RgRegs<NOUT, A_IT, B_IT> R0;
    float om = 0.f;   // running max |o0| of this thread's pieces (committed to g.o_amax at the end)
    // load cursor (runs ahead of the compute cursor, across unit boundaries)
    int lu = blockIdx.x, lseg = 0, lkoff = 0;
    DnTile ltile = unit_tile(lu);
    // compute cursor
    int cu = blockIdx.x, cs = 0;
    DnTile ctile = ltile;
    // parked unit being streamed out
    int p_row0 = ctile.row0, p_nrows = 0, p_next = DN_PT_NP;   // p_next >= NP: nothing pending

#define PT_ADVANCE()                                                                                                    \
    do {                                                                                                                \
        lkoff += DN_KB;                                                                                                 \
        if (lkoff >= g.a[lseg].w) { lkoff = 0; ++lseg; if (lseg >= g.nseg) { lseg = 0; lu += G; if (lu < nunits) ltile = unit_tile(lu); } } \
    } while (0)
// the same step without control flow (the split-bf16 iteration must stay ONE basic block: at a join the compiler
// falls back to s_waitcnt vmcnt(0), which would wait for the prefetch it has just issued); past the last slice the
// cursor stays put and the final slice is simply fetched again
#define PT_ADVANCE_SEL(commit)                                                                                          \
    do {                                                                                                                \
        int nk_ = lkoff + DN_KB, ns_ = lseg, nu_ = lu;                                                                  \
        const bool se_ = nk_ >= g.a[lseg].w;                                                                            \
        nk_ = se_ ? 0 : nk_;                                                                                            \
        ns_ = se_ ? ns_ + 1 : ns_;                                                                                      \
        const bool ue_ = ns_ >= g.nseg;                                                                                 \
        ns_ = ue_ ? 0 : ns_;                                                                                            \
        nu_ = ue_ ? nu_ + G : nu_;                                                                                      \
        const bool ok_ = (commit) && nu_ < nunits;                                                                      \
        lkoff = ok_ ? nk_ : lkoff; lseg = ok_ ? ns_ : lseg; lu = ok_ ? nu_ : lu;                                        \
        ltile = unit_tile(lu);                                                                                          \
    } while (0)
#define PT_LOAD(RS) rg_load<TN, NTHR, NOUT, true, BCOLK, HASQ, A_IT, B_IT, PAIRK>(g, ltile, n0, lseg, lkoff, tid, RS)
#define PT_STORE(buf, RS)                                                                                               \
    do {                                                                                                                \
        if constexpr (X3) rg_store_x3<NTHR, NOUT, BCOLK, HASQ, A_IT, B_IT>(reinterpret_cast<unsigned char*>(buf),        \
                                                                          reinterpret_cast<unsigned char*>((buf) + SA), tid, RS); \
        else rg_store<TN, NTHR, NOUT, BCOLK, HASQ, A_IT, B_IT>((buf), (buf) + SA, tid, RS);                             \
    } while (0)
// One pipeline iteration on slice j; RS is the ring set of slice j+1 (staged now) and of slice j+1+DEPTH (loaded now).
// Split-bf16 order: every LDS read of slice j is issued first, then the pure-VALU split of slice j+1 and the MFMAs of slice j
// (independent instruction streams the scheduler can interleave), then the LDS writes of slice j+1.  With the writes first
// the MFMAs had to wait behind them (may-alias LDS), and a slice took ~5.8k cycles of which the matrix pipe was busy 1.5k.
#define PT_ITER(j, RS)                                                                                                  \
    do {                                                                                                                \
        float* cur = smem + ((j) & 1) * SBUF;                                                                           \
        float* nxt = smem + (((j) & 1) ^ 1) * SBUF;                                                                     \
        PtPiece P[PPI];                                                                                                 \
        const bool pending = p_next < DN_PT_NP;                                                                         \
        if constexpr (X3) {                                                                                             \
            const unsigned char* cA = reinterpret_cast<const unsigned char*>(cur);                                      \
            const unsigned char* cB = reinterpret_cast<const unsigned char*>(cur + SA);                                 \
            X3Frags<MT, NT, NOUT> F;                                                                                    \
            X3Planes<NOUT, A_IT, B_IT> PLN;                                                                             \
            /* no branches in here except around the prefetch: pieces past the end are dead (ok = false), and the   */ \
            /* last iteration stages stale registers into a buffer nobody reads                                      */ \
            rg_frag_x3<MT, NT, NOUT>(cA, cB, wr * MT * 32, wc * NT * 32, li, ls, 0, F);                                 \
            /* every auxiliary load of this iteration's pieces goes out BEFORE the prefetch: loads retire in order,   */ \
            /* so a piece that waited on a younger load would wait for the prefetch too                              */ \
            _Pragma("unroll") for (int k = 0; k < PPI; ++k)                                                             \
                pt_piece_load<MODE, FLAG>(g, seed, sE, p_next + k, tid, p_row0, p_nrows, n0, P[k]);                           \
            rg_split_x3<NOUT, BCOLK, HASQ, A_IT, B_IT>(RS, PLN);                                                        \
            rg_mma_x3<MT, NT, NOUT>(F, 0, acc);                                                                         \
            PT_ADVANCE_SEL((j) + 1 + 1 < T);                                                                  \
            PT_LOAD(RS);                                                                                                \
            rg_frag_x3<MT, NT, NOUT>(cA, cB, wr * MT * 32, wc * NT * 32, li, ls, 1, F);                                 \
            rg_mma_x3<MT, NT, NOUT>(F, 1, acc);                                                                         \
            rg_put_x3<NTHR, NOUT, BCOLK, A_IT, B_IT>(reinterpret_cast<unsigned char*>(nxt),                             \
                                                     reinterpret_cast<unsigned char*>(nxt + SA), tid, PLN);             \
            _Pragma("unroll") for (int k = 0; k < PPI; ++k) om = dn_f4_amax(om, pt_piece_store<MODE, FLAG>(g, P[k]));   \
            p_next = (p_next + PPI < DN_PT_NP) ? p_next + PPI : DN_PT_NP;                                               \
        } else {                                                                                                        \
            if ((j) + 1 < T) PT_STORE(nxt, RS);                                                                         \
            if ((j) + 1 + 1 < T) { PT_ADVANCE(); PT_LOAD(RS); }                                               \
            if (pending) {                                                                                              \
                _Pragma("unroll") for (int k = 0; k < PPI; ++k)                                                         \
                    pt_piece_load<MODE, FLAG>(g, seed, sE, p_next + k, tid, p_row0, p_nrows, n0, P[k]);                       \
            }                                                                                                           \
            rg_compute<TN, MT, NT, NOUT, BCOLK>(cur, cur + SA, wr * MT * 32, wc * NT * 32, li, ls, acc);                \
            if (pending) {                                                                                              \
                _Pragma("unroll") for (int k = 0; k < PPI; ++k) om = dn_f4_amax(om, pt_piece_store<MODE, FLAG>(g, P[k])); \
                p_next += PPI;                                                                                          \
            }                                                                                                           \
        }                                                                                                               \
        if (++cs == nsl) { /* unit complete: park the accumulators (fragment layout -> row-major, conflict-free) */     \
            _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                                           \
                _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                        \
                    sE[((wr * MT + mt) * 32 + dn_acc_row(r, lane)) * 128 + wc * 32 + li] = acc[0][mt][0][r];            \
                    acc[0][mt][0][r] = 0.f;                                                                             \
                }                                                                                                       \
            p_row0 = ctile.row0; p_nrows = ctile.nrows; p_next = 0;                                       \
            cs = 0;                                                                                                     \
            cu += G;                                                                                                    \
            if (cu < nunits) ctile = unit_tile(cu);                                                                     \
        }                                                                                                               \
        __syncthreads(); /* slice buffer hand-off + visibility of the parked unit */                                    \
    } while (0)

    // prologue: slice 0 -> LDS, slices 1..DEPTH -> ring sets (slice % DEPTH)
    PT_LOAD(R0);
    PT_STORE(smem, R0);
    if (T > 1) { PT_ADVANCE(); PT_LOAD(R0); }
    __syncthreads();
    for (int j = 0; j < T; ++j) PT_ITER(j, R0);
#undef PT_ITER
#undef PT_STORE
#undef PT_LOAD
#undef PT_ADVANCE
    // flush what is still parked (the last unit)
    for (; p_next < DN_PT_NP; ++p_next) {
        PtPiece P1;
        pt_piece_load<MODE, FLAG>(g, seed, sE, p_next, tid, p_row0, p_nrows, n0, P1);
        om = dn_f4_amax(om, pt_piece_store<MODE, FLAG>(g, P1));
    }
    if (g.o_amax) dn_amax_commit<true>(g.o_amax, om);
}

#ifndef DN_PT_X3
#define DN_PT_X3 (DN_PT_ROWS == 128)   // split-bf16 MFMA in the persistent kernel (build with -DDN_PT_X3=0 for exact-f32 MFMA)
#endif
// ---- wave-specialised persistent row GEMM (split-bf16, one output, >= 4 slices) --------------------------------------------
// Measured on the lock-step kernel above (linear C->C, 158k rows): the compute side alone (no global traffic) takes 36 us,
// the memory side alone (no MFMA / split / LDS reads) 35 us, the two together 52-56 us -- every wave ran the same phase at
// the same time and sat in the memory instructions it issued.  Here the eight waves of a workgroup have fixed roles:
//   waves 0-3 (one per SIMD): LDS fragment reads + MFMAs of a 64x64 sub-tile each, and parking the finished unit in LDS;
//   waves 4-11              : global prefetch of the next slice, split into bf16 planes, LDS writes, and the deferred
//                             epilogue of the parked unit (LDS read, auxiliary operands, float4 stores).
// One barrier per slice hands the slice buffer over.  A parked unit must be streamed out before the next one is parked at
// the end of the following unit's last slice, hence PPI = ceil(NP / (nsl - 1)) pieces per loader thread and slice.
#ifndef DN_PT_WS
#define DN_PT_WS 1
#endif
// DN_WS_KO (development, timing only -- results are wrong): knock out one phase of the loader / MFMA loops to see what bounds the period:
// 1 = no deferred pieces (nothing is stored), 2 = no LDS writes, 4 = no slice requests after the prologue, 8 = no MFMAs, 16 = no LDS fragment reads
#ifndef DN_WS_KO
#define DN_WS_KO 0
#endif
#ifndef DN_WS_DIRECT
#define DN_WS_DIRECT 0   // plain-store products (no epilogue operand): the MFMA waves store the finished tile straight from their accumulators
#endif                   // (dword stores, 128 B per half-wave) instead of parking it in LDS for the loaders to stream out
#ifndef DN_WS_PAIR
#define DN_WS_PAIR 0     // with DN_WS_DIRECT on the 2-term engine: TWO 32-wide slices per barrier (the parked tile's 64 KiB hold the second pair of
#endif                   // sub-stages): half the barriers and per-iteration fixed costs per byte
#ifndef DN_WS_BRES
#define DN_WS_BRES 0   // B-cached products on the 2-term engine: the four split B slices live in LDS for the whole kernel (the spare 2 x 32 KiB of the
#endif                 // two slice buffers) instead of being re-written from registers every slice
#ifndef DN_WS_GROUP_COMMIT
#define DN_WS_GROUP_COMMIT 1   // one magnitude commit per workgroup (0: one per loader wave)
#endif
#ifndef DN_WS_LW
#define DN_WS_LW 8                            // loader waves per workgroup (4 or 8); measured: 4 loader waves made the loaders the pole
#endif
#ifndef DN_WS_LOADER_PRIO
#define DN_WS_LOADER_PRIO 0   // s_setprio of the loader waves (the MFMA waves are the older ones and win arbitration at equal priority)
#endif
#define DN_WS_LTHR (64 * DN_WS_LW)             // loader threads
#ifndef DN_WS_BCACHE
#define DN_WS_BCACHE 1   // keep the split B strip in loader registers when B is the same for every unit (see rowgemm_ws_kernel)
#endif
#ifndef DN_WS_PW
#define DN_WS_PW DN_WS_LW                     // loader waves that also stream the parked unit out; measured: 4 or 2 (the oldest) instead of all 8 is slower (C->C 47-52 / 59 us vs 46-49)
#endif
#define DN_WS_PTHR (64 * DN_WS_PW)
#define DN_WS_NP (128 * 128 / 4 / DN_WS_PTHR)   // float4 pieces per piece thread and unit

#if defined(DN_WS_TRACE) && !defined(DN_EMULATE)   // development build only (make EXTRA=-DDN_WS_TRACE=<block>): s_memtime stamps of one workgroup's
__device__ unsigned long long dn_ws_trace_buf[12 * 256];   // waves (12 x 256 stamps), read by tools/kbench --trace
extern "C" int dn_debug_rd_trace_read(unsigned long long* out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(dn_ws_trace_buf), (size_t)(n < 12 * 256 ? n : 12 * 256) * sizeof(unsigned long long));
}
#define WS_TR_DECL int trn = 0
#define WS_TR()                                                                                                         \
    do {                                                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                                              \
        if (blockIdx.x == (DN_WS_TRACE) && blockIdx.y == 0 && (threadIdx.x & 63) == 0 && trn < 256)                     \
            dn_ws_trace_buf[(threadIdx.x >> 6) * 256 + trn] = __builtin_amdgcn_s_memtime();                            \
        ++trn;                                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                              \
    } while (0)
#define WS_TR_WAITV() __builtin_amdgcn_s_waitcnt(0x0F70)   /* vmcnt(0), other counters untouched */
#else
#define WS_TR_DECL
#define WS_TR() do {} while (0)
#define WS_TR_WAITV() do {} while (0)
#endif

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
            R.a[i] = (DN_WS_NT & 2) ? dn_load_f4_nt(ap + off) : *reinterpret_cast<const float4*>(ap + off);
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
__global__ __launch_bounds__(256 + DN_WS_LTHR) DN_WAVES_PER_EU(DN_WS_LW == 4 ? 2 : 3) void rowgemm_ws_kernel(RgArgs g, int ntiles) {
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
    constexpr bool BRES = BC && NP == 2 && DN_WS_BRES != 0;
    constexpr bool DIRECT = DN_WS_DIRECT != 0 && MODE == DN_EPI_STORE && !FLAG;
    constexpr bool PAIR = DIRECT && NP == 2 && !BRES && DN_WS_PAIR != 0;   // (the host sends products with nsl % 4 == 0 here: T / 2 is even)
    // PAIR: sub-stage u of stage s at byte offset (2 s + u) * 32 KiB: A planes, B planes at + 16 KiB
#define WS_PAIR_A(s, u) (reinterpret_cast<unsigned char*>(smem) + (2 * (s) + (u)) * 32768)
    // BRES: slice s of B lives at byte offset (s >> 1) * SBUF * 4 + 16 KiB + (s & 1) * 16 KiB (behind the two A planes of either slice buffer)
#define WS_BRES_PTR(s) (reinterpret_cast<unsigned char*>(smem) + ((s) >> 1) * (SBUF * 4) + 16384 + ((s) & 1) * 16384)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int G = gridDim.x;
    const int n0 = blockIdx.y * TN;
    int nsl = 0;
    for (int s = 0; s < g.nseg; ++s) nsl += g.a[s].w / DN_KB;   // host guarantees nsl >= 4 and whole slices
    const int my_units = ((int)blockIdx.x < ntiles) ? (ntiles - (int)blockIdx.x + G - 1) / G : 0;
    const int T = my_units * nsl;
    if (T == 0) return;

    // Roles.  Default: waves 0-3 (one per SIMD) multiply, waves 4-11 load.  DN_WS_SPLIT_SIMD=1 instead puts the four MFMA
    // waves on two SIMDs (a workgroup's waves are dealt to the SIMDs cyclically) and leaves the other two to the loaders.
#ifndef DN_WS_SPLIT_SIMD
#define DN_WS_SPLIT_SIMD 0
#endif
#if DN_WS_SPLIT_SIMD   // MFMA waves 0,1,4,5 (SIMDs 0 and 1, two each), loaders 2,3,6,7 (SIMDs 2, 3) and 8..11 (one per SIMD)
    const bool is_mfma = wave < 8 && (wave & 3) < 2;
    const int mw = (wave & 1) + 2 * (wave >> 2);
    const int lw = wave >= 8 ? wave - 4 : (wave & 1) + 2 * (wave >> 2);
#else
    const bool is_mfma = wave < 4;
    const int mw = wave, lw = wave - 4;
#endif
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
        int ucur = blockIdx.x;          // (DIRECT) the unit being multiplied
        float so_m = 1.f, om_m = 0.f;
        if constexpr (DIRECT && NP == 2) so_m = (1.f / dn_pow2_scale(dn_amax_eval(g.a_amax))) * (1.f / dn_pow2_scale(dn_amax_eval(g.b_amax)));
        WS_TR_DECL;
        auto store_unit = [&]() {   // (DIRECT) the finished unit straight from the accumulators: lane = column, 32 lanes = 128 contiguous bytes
            const DnTile t = g.tiles[ucur];
            ucur += G;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    const int col = n0 + wc * 64 + nt * 32 + li;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = wr * 64 + mt * 32 + dn_acc_row(r, lane);
                        const float v = acc[0][mt][nt][r] * so_m;
                        if (row < t.nrows && col < g.N) {
                            g.o0[(long long)(t.row0 + row) * g.ldo + col] = v;
                            om_m = fabsf(v) > om_m ? fabsf(v) : om_m;
                        }
                        acc[0][mt][nt][r] = 0.f;
                    }
                }
        };
        __syncthreads();   // slice 0 staged
        if constexpr (PAIR) {
            for (int jp = 0; jp < T / 2; ++jp) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const unsigned char* cA = WS_PAIR_A(jp & 1, u);
                    const unsigned char* cB = cA + 16384;
                    X3Frags<2, 2, 1, NP> F;
                    rg_frag_x3<2, 2, 1, NP>(cA, cB, wr * 64, wc * 64, li, lg, 0, F);
                    rg_frag_x3<2, 2, 1, NP>(cA, cB, wr * 64, wc * 64, li, lg, 1, F);
                    ws_mma<NP>(F, 0, acc);
                    ws_mma<NP>(F, 1, acc);
                }
                cs += 2;
                if (cs == nsl) { cs = 0; store_unit(); }
                __syncthreads();
            }
            if (g.o_amax) dn_amax_commit<true>(g.o_amax, om_m);
            return;
        }
#if DN_WS_KO & 16
        X3Frags<2, 2, 1, NP> F;
#endif
        for (int j = 0; j < T; ++j) {
            const unsigned char* cA = reinterpret_cast<const unsigned char*>(smem + (j & 1) * SBUF);
            const unsigned char* cB = BRES ? WS_BRES_PTR(j & 3) : cA + SA * 4;
#if !(DN_WS_KO & 16)
            X3Frags<2, 2, 1, NP> F;
#endif
            WS_TR();   // m0: iteration start
            if ((DN_WS_KO & 16) == 0 || j == 0) {
                rg_frag_x3<2, 2, 1, NP>(cA, cB, wr * 64, wc * 64, li, lg, 0, F);
                rg_frag_x3<2, 2, 1, NP>(cA, cB, wr * 64, wc * 64, li, lg, 1, F);
            }
            if (!(DN_WS_KO & 8)) {
                ws_mma<NP>(F, 0, acc);
                ws_mma<NP>(F, 1, acc);
            }
            WS_TR();   // m1: reads + MFMAs issued
            if (DIRECT && ++cs == nsl) {   // unit complete
                cs = 0;
                store_unit();
            }
            if (!DIRECT && ++cs == nsl) {   // unit complete: park it (fragment layout -> row-major) for the loader waves to stream out
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
            WS_TR();   // m2: before the barrier
            __syncthreads();
        }
        if (DIRECT && g.o_amax) dn_amax_commit<true>(g.o_amax, om_m);
        return;
    }

    // ---------------------------------------------------- loader waves ----------------------------------------------------
    DN_SETPRIO(DN_WS_LOADER_PRIO);
    const int lt = lw * 64 + lane;
    if (NP == 2 && !BRES && !PAIR && lt < 2) reinterpret_cast<unsigned*>(smem + WS_AMAX_LDS)[lt] = 0u;   // workgroup-level magnitude commit (before the first barrier)
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
    const bool piece_wave = (DN_WS_PW == DN_WS_LW) ? true : lt < DN_WS_PTHR;
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
        if constexpr (BC && !BRES) {                                                                                    \
            _Pragma("unroll") for (int i_ = 0; i_ < B_IT; ++i_)                                                         \
                _Pragma("unroll") for (int p_ = 0; p_ < NP; ++p_) PLN.b[0][i_][p_] = Bc[SIDX][i_][p_];                  \
        }                                                                                                               \
        rg_put_x3<LTHR, NOUT, BCOLK, A_IT, B_IT, NP, true, !BRES>(reinterpret_cast<unsigned char*>(buf),                \
                                                 reinterpret_cast<unsigned char*>((buf) + SA), lt, PLN);                \
    } while (0)

// Order inside an iteration: stage -> deferred pieces (their operands were requested an iteration ago) -> operands of the
// next iteration's pieces -> slice prefetch.  (Measured: a second register set / fetching two slices ahead, and requesting
// the piece operands before the prefetch, were both slower -- 60/48/135 us vs 53/51/127 us for the NN, C->C, 3C->C products.)
#define WS_SPLIT(RS, SIDX, PLN)                                                                                         \
    do {                                                                                                                \
        rg_split_x3<NOUT, BCOLK, HASQ, A_IT, B_IT, true, !BC, NP>(RS, PLN, sa, sb);                                     \
        if constexpr (BC && !BRES) {                                                                                    \
            _Pragma("unroll") for (int i_ = 0; i_ < B_IT; ++i_)                                                         \
                _Pragma("unroll") for (int p_ = 0; p_ < NP; ++p_) PLN.b[0][i_][p_] = Bc[SIDX][i_][p_];                  \
        }                                                                                                               \
    } while (0)
#define WS_PUT(buf, PLN)                                                                                                \
    rg_put_x3<LTHR, NOUT, BCOLK, A_IT, B_IT, NP, true, !BRES>(reinterpret_cast<unsigned char*>(buf), reinterpret_cast<unsigned char*>((buf) + SA), lt, PLN)
#define WS_PIECES()                                                                                                     \
    do {                                                                                                                \
        if (piece_wave) {              /* wave-uniform: only the first DN_WS_PW loader waves stream the parked unit out */ \
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

#ifndef DN_WS_EARLY
#define DN_WS_EARLY 1
#endif
// DN_WS_DEPTH = 2: two register sets, slice s travels in set s & 1 and is requested two iterations before it is split (three slices of
// reads per loader lane in flight instead of two).  The 16 KiB A slice per workgroup and ~1.5 slices in flight are 6 MB of HBM reads in
// flight on 256 CUs -- at ~2 us loaded latency that is the 3.2 TB/s the kernel runs at.
#ifndef DN_WS_DEPTH
#define DN_WS_DEPTH 1
#endif
// Order inside an iteration (DN_WS_EARLY, round 3): wait for slice j+1 -> split it into plane registers -> the registers it came in are
// free: request slice j+2 NOW -> LDS writes of slice j+1 -> deferred pieces -> operands of the next pieces -> barrier.  The s_memtime
// timeline of the round-2 order (request at the end of the iteration, profiles/r03_ws_trace_*.txt) showed 1300-2100 of a loader's
// ~4600-5200 cycles per slice spent waiting for that request: it had only the barrier to fly in; now it has most of an iteration.
// DN_WS_EARLY=0 keeps the round-2 order: stage -> pieces -> piece operands -> request.
// DN_WS_ORDER 1: pieces before the slice request (with DN_WS_DEPTH = 2 the request is two iterations ahead anyway, and vmcnt retires in
// order: piece operands requested AFTER a slice force that slice to have arrived when they are consumed)
#ifndef DN_WS_ORDER
#define DN_WS_ORDER 0
#endif
#if DN_WS_EARLY
#define WS_ITER(j, RS, SIDX)                                                                                            \
    do {                                                                                                                \
        float* nxt = smem + (((j) & 1) ^ 1) * SBUF;                                                                     \
        X3Planes<NOUT, A_IT, B_IT, NP> PLN;                                                                             \
        WS_TR();                       /* l0: iteration start */                                                        \
        WS_TR_WAITV(); WS_TR();        /* l1: the prefetched slice has arrived (explicit wait in the trace build only) */ \
        WS_SPLIT(RS, SIDX, PLN);       /* slice j+1 (the last iteration stages a stale copy nobody reads) */            \
        WS_TR();                       /* l2: split */                                                                  \
        if (DN_WS_ORDER == 1 && !(DN_WS_KO & 1) && !DIRECT) WS_PIECES();                                                           \
        WS_ADVANCE((j) + 1 + DN_WS_DEPTH < T);                                                                          \
        if (!(DN_WS_KO & 4)) WS_LOAD(RS);   /* slice j+1+DEPTH */                                                       \
        WS_TR();                       /* l3: prefetch issued */                                                        \
        if (!(DN_WS_KO & 2)) WS_PUT(nxt, PLN);                                                                          \
        WS_TR();                       /* l4: LDS writes issued */                                                      \
        if (DN_WS_ORDER == 0 && !(DN_WS_KO & 1) && !DIRECT) WS_PIECES();                                                           \
        WS_TR();                       /* l5: pieces out + next pieces' operands requested */                           \
        __syncthreads();                                                                                                \
    } while (0)
#else
#define WS_ITER(j, RS, SIDX)                                                                                            \
    do {                                                                                                                \
        float* nxt = smem + (((j) & 1) ^ 1) * SBUF;                                                                     \
        X3Planes<NOUT, A_IT, B_IT, NP> PLN;                                                                             \
        WS_TR();                       /* l0: iteration start */                                                        \
        WS_TR_WAITV(); WS_TR();        /* l1: the prefetched slice has arrived (trace build only) */                    \
        WS_SPLIT(RS, SIDX, PLN);       /* slice j+1 (the last iteration stages a stale copy nobody reads) */            \
        WS_PUT(nxt, PLN);                                                                                               \
        WS_TR();                       /* l2: staged */                                                                 \
        WS_PIECES();                                                                                                    \
        WS_TR();                       /* l3: pieces out + next pieces' operands requested */                           \
        WS_ADVANCE((j) + 1 + DN_WS_DEPTH < T);                                                                          \
        WS_LOAD(RS);                   /* slice j+1+DEPTH */                                                            \
        WS_TR();                       /* l4: prefetch issued */                                                        \
        __syncthreads();                                                                                                \
    } while (0)
#endif

    WS_TR_DECL;
    uint2 Bc[(BC && !BRES) ? 4 : 1][B_IT][NP];
    if constexpr (BC) {   // split the whole B strip of this workgroup once (4 slices of the one segment)
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            RgRegs<NOUT, A_IT, B_IT> Rb;
            ws_load<BCOLK, A_IT, B_IT, false, true>(sp0, sl0, sb0, ldb, Ncols, 0, 0, n0, DN_KB * s4, lt, Rb);
            X3Planes<NOUT, A_IT, B_IT, NP> Pb;
            rg_split_x3<NOUT, BCOLK, HASQ, A_IT, B_IT, false, true, NP>(Rb, Pb, sa, sb);
            if constexpr (BRES) {   // resident in LDS for the whole kernel (the first barrier below publishes it)
                rg_put_x3<LTHR, NOUT, BCOLK, A_IT, B_IT, NP, false, true>(nullptr, WS_BRES_PTR(s4), lt, Pb);
            } else {
#pragma unroll
                for (int i = 0; i < B_IT; ++i)
#pragma unroll
                    for (int p3 = 0; p3 < NP; ++p3) Bc[s4][i][p3] = Pb.b[0][i][p3];
            }
        }
    }
    if constexpr (PAIR) {
        // two slices per barrier: pair p = slices (2p, 2p + 1) lives in stage p & 1; two register sets, requested one pair ahead
        RgRegs<NOUT, A_IT, B_IT> Ra, Rc;
#define WS_PPUT(stg, sub, PL_) rg_put_x3<LTHR, NOUT, BCOLK, A_IT, B_IT, NP>(WS_PAIR_A(stg, sub), WS_PAIR_A(stg, sub) + 16384, lt, PL_)
#define WS_PAIR_ITER(jp, STG, S0, S1)                                                                                   \
        do {                                                                                                            \
            X3Planes<NOUT, A_IT, B_IT, NP> P0, P1;                                                                      \
            WS_SPLIT(Ra, S0, P0);                                                                                       \
            WS_SPLIT(Rc, S1, P1);                                                                                       \
            WS_ADVANCE(2 * (jp) + 4 < T);                                                                               \
            WS_LOAD(Ra);                                                                                                \
            WS_ADVANCE(2 * (jp) + 5 < T);                                                                               \
            WS_LOAD(Rc);                                                                                                \
            WS_PPUT(STG, 0, P0);                                                                                        \
            WS_PPUT(STG, 1, P1);                                                                                        \
            __syncthreads();                                                                                            \
        } while (0)
        WS_LOAD(Ra);
        WS_ADVANCE(T > 1);
        WS_LOAD(Rc);
        {
            X3Planes<NOUT, A_IT, B_IT, NP> P0, P1;
            WS_SPLIT(Ra, 0, P0);
            WS_SPLIT(Rc, 1, P1);
            WS_ADVANCE(T > 2);
            WS_LOAD(Ra);                   // slice 2
            WS_ADVANCE(T > 3);
            WS_LOAD(Rc);                   // slice 3
            WS_PPUT(0, 0, P0);
            WS_PPUT(0, 1, P1);
        }
        __syncthreads();                   // pair 0 staged
        for (int jp = 0; jp < T / 2; jp += 2) {     // iteration jp stages pair jp + 1 (slices 2, 3 of a unit), jp + 1 the next unit's slices 0, 1
            WS_PAIR_ITER(jp, 1, 2, 3);
            WS_PAIR_ITER(jp + 1, 0, 0, 1);
        }
#undef WS_PAIR_ITER
#undef WS_PPUT
        return;
    }
    WS_LOAD(R0);
    WS_STAGE(smem, R0, 0);
    WS_ADVANCE(T > 1);
#if DN_WS_DEPTH == 2
    RgRegs<NOUT, A_IT, B_IT> R1;
    WS_LOAD(R1);                       // slice 1
    WS_ADVANCE(T > 2);
    WS_LOAD(R0);                       // slice 2
    __syncthreads();                   // slice 0 staged
    if constexpr (BC) {                // T is a multiple of 4: iteration j stages slice (j + 1) % 4 of its unit (it travels in set (j + 1) & 1)
        for (int j = 0; j < T; j += 4) {
            WS_ITER(j, R1, 1);
            WS_ITER(j + 1, R0, 2);
            WS_ITER(j + 2, R1, 3);
            WS_ITER(j + 3, R0, 0);
        }
    } else {
        int j = 0;
        for (; j + 1 < T; j += 2) {
            WS_ITER(j, R1, 0);
            WS_ITER(j + 1, R0, 0);
        }
        if (j < T) WS_ITER(j, R1, 0);
    }
#else
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
#endif
#undef WS_PAIR_A
#undef WS_BRES_PTR
#undef WS_ITER
#undef WS_PIECES
#undef WS_PUT
#undef WS_SPLIT
#undef WS_STAGE
#undef WS_LOAD
#undef WS_ADVANCE
    // flush the last parked unit
    if (!piece_wave || DIRECT) return;
    for (; p_next < DN_WS_NP; ++p_next) {
        WsAux A1;
        ws_aux_load<MODE, FLAG, XMASK>(g, seed, p_next, lt, p_row0, p_nrows, n0, A1);
        ws_piece_out<MODE, FLAG>(g, *reinterpret_cast<const float4*>(&sE[A1.lds]), bias, A1, so, om);
    }
    if (g.o_amax) {
        if constexpr (NP == 2 && !BRES && DN_WS_PW == DN_WS_LW && DN_WS_GROUP_COMMIT) dn_amax_commit_group(g.o_amax, om, reinterpret_cast<unsigned*>(smem + WS_AMAX_LDS), DN_WS_LW);
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

template <int MODE, bool BCOLK, bool FLAG>
static int pt_launch(const RgArgs& g, int ntiles, hipStream_t stream) {
    constexpr bool X3 = DN_PT_X3 != 0;
    if (X3 && DN_PT_WS && DN_PT_ROWS == 128) {
        int nsl = 0;
        for (int s = 0; s < g.nseg; ++s) nsl += g.a[s].w / DN_KB;
        // PPI = ceil(NP / (nsl - 1)) for the two slice counts that matter (K = 128: 4 slices, K = 384: 12)
        if (g.f16) {   // split-fp16 engine (the caller supplies the operand magnitudes)
            if (nsl >= 9) return ws_launch<MODE, BCOLK, FLAG, (DN_WS_NP + 7) / 8, false, 2>(g, ntiles, stream);
            if (DN_WS_BCACHE && nsl == 4 && g.nseg == 1 && g.b_mesh_stride == 0)
                return ws_launch<MODE, BCOLK, FLAG, (DN_WS_NP + 2) / 3, true, 2>(g, ntiles, stream);
            if (nsl >= 4) return ws_launch<MODE, BCOLK, FLAG, (DN_WS_NP + 2) / 3, false, 2>(g, ntiles, stream);
        }
        if (nsl >= 9) return ws_launch<MODE, BCOLK, FLAG, (DN_WS_NP + 7) / 8, false>(g, ntiles, stream);
        if (DN_WS_BCACHE && nsl == 4 && g.nseg == 1 && g.b_mesh_stride == 0)
            return ws_launch<MODE, BCOLK, FLAG, (DN_WS_NP + 2) / 3, true>(g, ntiles, stream);
        if (nsl >= 4) return ws_launch<MODE, BCOLK, FLAG, (DN_WS_NP + 2) / 3, false>(g, ntiles, stream);
    }
    // slice buffers (2x) + parked accumulators: 128 KiB with f32 tiles, exactly 160 KiB with bf16x3 planes
    const size_t smem = X3 ? (size_t)(2 * (DN_PT_ROWS * 64 * 3 + 128 * 64 * 3) + DN_PT_ROWS * 128 * 4)
                           : (size_t)(2 * (DN_PT_ROWS * DN_KB + DN_KB * 128) + DN_PT_ROWS * 128) * sizeof(float);
#ifndef DN_EMULATE
    static unsigned long long lds_opt_in = 0;   // per-device bitmap
    { const int oe_ = dn_lds_opt_in(reinterpret_cast<const void*>(&rowgemm_persist_kernel<MODE, BCOLK, FLAG, X3>), smem, &lds_opt_in); if (oe_) return oe_; }
#endif
    const int upt = DN_TM / DN_PT_ROWS;
    int gx = upt * dn_num_cus();   // one 128-row workgroup per CU (two 64-row ones)
    if (gx > upt * ntiles) gx = upt * ntiles;
    DN_LAUNCH((rowgemm_persist_kernel<MODE, BCOLK, FLAG, X3>), dim3(gx, (g.N + 127) / 128, 1), dim3(DN_PT_THREADS, 1, 1), smem,
              stream, g, ntiles);
    return (int)hipGetLastError();
}

// eligibility of the persistent path: aligned operands, wide output, 3..8 slices, float4-able epilogue operands
static bool pt_eligible(const RgArgs& g, int nout) {
    if (nout != 1 || !g.aligned || g.N < 128 || g.N % 4 != 0 || g.ldo % 4 != 0 || g.ldr % 4 != 0) return false;
    int nsl = 0;
    for (int s = 0; s < g.nseg; ++s) { nsl += g.a[s].w / DN_KB; if (g.a[s].q) return false; }
    if (nsl < 3 || nsl > DN_PT_MAX_SLICES) return false;
    auto al = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
    if (!al(g.o0) || !al(g.r0) || !al(g.bias) || ((uintptr_t)g.mask & 3) != 0) return false;
    switch (g.mode) {
        case DN_EPI_STORE: return true;
        case DN_EPI_BIAS_RELU: return g.bias != nullptr && g.b_colk;
        case DN_EPI_BIAS_RESID: return g.bias != nullptr && g.r0 != nullptr && g.b_colk;
        case DN_EPI_MUL_DFAC: case DN_EPI_ADD: case DN_EPI_DTANH: case DN_EPI_MASS_ADD: return g.r0 != nullptr && !g.b_colk;
        default: return false;
    }
}

static int pt_dispatch(const RgArgs& g, int ntiles, hipStream_t stream) {
    const bool ck = g.b_colk != 0;
    switch (g.mode) {
        case DN_EPI_STORE:
            if (g.bias) return ck ? pt_launch<DN_EPI_STORE, true, true>(g, ntiles, stream) : pt_launch<DN_EPI_STORE, false, true>(g, ntiles, stream);
            return ck ? pt_launch<DN_EPI_STORE, true, false>(g, ntiles, stream) : pt_launch<DN_EPI_STORE, false, false>(g, ntiles, stream);
        case DN_EPI_BIAS_RELU:
            return (g.mask || g.rng_seed) ? pt_launch<DN_EPI_BIAS_RELU, true, true>(g, ntiles, stream) : pt_launch<DN_EPI_BIAS_RELU, true, false>(g, ntiles, stream);
        case DN_EPI_BIAS_RESID: return pt_launch<DN_EPI_BIAS_RESID, true, false>(g, ntiles, stream);
        case DN_EPI_MUL_DFAC: return pt_launch<DN_EPI_MUL_DFAC, false, false>(g, ntiles, stream);
        case DN_EPI_ADD: return pt_launch<DN_EPI_ADD, false, false>(g, ntiles, stream);
        case DN_EPI_DTANH: return pt_launch<DN_EPI_DTANH, false, false>(g, ntiles, stream);
        case DN_EPI_MASS_ADD: return pt_launch<DN_EPI_MASS_ADD, false, false>(g, ntiles, stream);
        default: return DN_ERR_BAD_MODE;
    }
}

bool dn_rowgemm_try_persistent(const RgArgs& g, int ntiles, int nout, hipStream_t stream, int* err) {
    const bool ck = g.b_colk != 0;
    if (pt_eligible(g, nout)) {
        *err = pt_dispatch(g, ntiles, stream);
        return true;
    }
    return false;
}
