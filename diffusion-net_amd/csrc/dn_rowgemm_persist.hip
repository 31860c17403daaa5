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
#define DN_PT_MAX_SLICES 12  // up to K = 384 (the 3C -> C MLP layer); measured 177 -> 155 us with the bf16x3 path
#endif
#ifndef DN_PT_DEPTH
#define DN_PT_DEPTH 1   // register sets in the prefetch ring (slice s+DEPTH is loaded once slice s has been split)
#endif
#if defined(DN_PT_ABLATE_OUT)   // development ablation: parked units are never streamed out
#define DN_PT_OUT_START DN_PT_NP
#else
#define DN_PT_OUT_START 0
#endif
#if defined(DN_PT_ABLATE_LOADS)
#define DN_PT_SKIP_LOADS 1
#else
#define DN_PT_SKIP_LOADS 0
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
__device__ __forceinline__ void pt_piece_load(const RgArgs& g, const float* sE, int piece, int tid, int row0, int nrows,
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
        P.mk = g.mask ? ld : dn_keep_bytes(dn_keep_bits(g.rng_seed, grow, ccol >> 2, (g.N + 3) >> 2));
    }
    if (MODE == DN_EPI_MASS_ADD) P.rs = g.rowv[grow];
}

#if defined(DN_PT_TRACE)   // development build only: per-phase timestamps of workgroup 0 (waves 0 and 7), tools/trace_pt.py
__device__ unsigned long long dn_pt_trace_buf[2 * 4096];
#define PT_T()                                                                                                          \
    do {                                                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                                              \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime();                                                     \
        if (blockIdx.x == 0 && blockIdx.y == 0 && (tid == 0 || tid == 448) && trn < 4096)                              \
            dn_pt_trace_buf[(tid ? 4096 : 0) + trn] = t_;                                                              \
        ++trn;                                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                              \
    } while (0)
extern "C" int dn_debug_trace_read(unsigned long long* out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(dn_pt_trace_buf), sizeof(unsigned long long) * n);
}
#ifndef DN_WS_TRACE_TID
#define DN_WS_TRACE_TID 256   // first lane of the traced loader wave
#endif
#define WS_T(who)                                                                                                       \
    do {                                                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                                              \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime();                                                     \
        if (blockIdx.x == 0 && blockIdx.y == 0 && tid == (who) && trn < 4096) dn_pt_trace_buf[((who) ? 4096 : 0) + trn] = t_; \
        ++trn;                                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                              \
    } while (0)
#else
#define PT_T() do {} while (0)
#define WS_T(who) do {} while (0)
#endif
template <int MODE, bool BCOLK, bool FLAG, bool X3>
__global__ __launch_bounds__(DN_PT_THREADS) DN_WAVES_PER_EU(2) void rowgemm_persist_kernel(RgArgs g, int ntiles) {

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

    // DN_PT_DEPTH register sets form the prefetch ring: slice s travels in set s % DEPTH and is loaded DEPTH iterations
    // before it is written to LDS, so DEPTH x 16 KiB of reads per CU are in flight (one slice ahead covers only ~1 us of HBM
    // latency once the split-bf16 MFMAs made an iteration that short).
#if DN_PT_DEPTH == 3
    RgRegs<NOUT, A_IT, B_IT> R0, R1, R2;
#elif DN_PT_DEPTH == 2
    RgRegs<NOUT, A_IT, B_IT> R0, R1;
#else
    RgRegs<NOUT, A_IT, B_IT> R0;
#endif
    // load cursor (runs ahead of the compute cursor, across unit boundaries)
    int lu = blockIdx.x, lseg = 0, lkoff = 0;
    DnTile ltile = unit_tile(lu);
    // compute cursor
    int cu = blockIdx.x, cs = 0;
    DnTile ctile = ltile;
    // parked unit being streamed out
    int p_row0 = ctile.row0, p_nrows = 0, p_next = DN_PT_NP;   // p_next >= NP: nothing pending
#if DN_PT_SKIP_LOADS
    bool pt_first_load = true;
#endif
#if defined(DN_PT_TRACE)
    int trn = 0;
#endif

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
#if DN_PT_SKIP_LOADS   // development ablation: only the prologue fetches
#define PT_LOAD(RS) do { if (pt_first_load) rg_load<TN, NTHR, NOUT, true, BCOLK, HASQ, A_IT, B_IT, PAIRK>(g, ltile, n0, lseg, lkoff, tid, RS); pt_first_load = false; } while (0)
#else
#define PT_LOAD(RS) rg_load<TN, NTHR, NOUT, true, BCOLK, HASQ, A_IT, B_IT, PAIRK>(g, ltile, n0, lseg, lkoff, tid, RS)
#endif
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
        PT_T();                                                                                                         \
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
                pt_piece_load<MODE, FLAG>(g, sE, p_next + k, tid, p_row0, p_nrows, n0, P[k]);                           \
            PT_T();                                                                                                     \
            rg_split_x3<NOUT, BCOLK, HASQ, A_IT, B_IT>(RS, PLN);                                                        \
            rg_mma_x3<MT, NT, NOUT>(F, 0, acc);                                                                         \
            PT_T();                                                                                                     \
            PT_ADVANCE_SEL((j) + 1 + DN_PT_DEPTH < T);                                                                  \
            PT_LOAD(RS);                                                                                                \
            rg_frag_x3<MT, NT, NOUT>(cA, cB, wr * MT * 32, wc * NT * 32, li, ls, 1, F);                                 \
            PT_T();                                                                                                     \
            rg_mma_x3<MT, NT, NOUT>(F, 1, acc);                                                                         \
            PT_T();                                                                                                     \
            rg_put_x3<NTHR, NOUT, BCOLK, A_IT, B_IT>(reinterpret_cast<unsigned char*>(nxt),                             \
                                                     reinterpret_cast<unsigned char*>(nxt + SA), tid, PLN);             \
            _Pragma("unroll") for (int k = 0; k < PPI; ++k) pt_piece_store<MODE, FLAG>(g, P[k]);                        \
            p_next = (p_next + PPI < DN_PT_NP) ? p_next + PPI : DN_PT_NP;                                               \
            PT_T();                                                                                                     \
        } else {                                                                                                        \
            if ((j) + 1 < T) PT_STORE(nxt, RS);                                                                         \
            if ((j) + 1 + DN_PT_DEPTH < T) { PT_ADVANCE(); PT_LOAD(RS); }                                               \
            if (pending) {                                                                                              \
                _Pragma("unroll") for (int k = 0; k < PPI; ++k)                                                         \
                    pt_piece_load<MODE, FLAG>(g, sE, p_next + k, tid, p_row0, p_nrows, n0, P[k]);                       \
            }                                                                                                           \
            rg_compute<TN, MT, NT, NOUT, BCOLK>(cur, cur + SA, wr * MT * 32, wc * NT * 32, li, ls, acc);                \
            if (pending) {                                                                                              \
                _Pragma("unroll") for (int k = 0; k < PPI; ++k) pt_piece_store<MODE, FLAG>(g, P[k]);                    \
                p_next += PPI;                                                                                          \
            }                                                                                                           \
        }                                                                                                               \
        if (++cs == nsl) { /* unit complete: park the accumulators (fragment layout -> row-major, conflict-free) */     \
            _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                                           \
                _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                        \
                    sE[((wr * MT + mt) * 32 + dn_acc_row(r, lane)) * 128 + wc * 32 + li] = acc[0][mt][0][r];            \
                    acc[0][mt][0][r] = 0.f;                                                                             \
                }                                                                                                       \
            p_row0 = ctile.row0; p_nrows = ctile.nrows; p_next = DN_PT_OUT_START;                                       \
            cs = 0;                                                                                                     \
            cu += G;                                                                                                    \
            if (cu < nunits) ctile = unit_tile(cu);                                                                     \
        }                                                                                                               \
        PT_T();                                                                                                         \
        __syncthreads(); /* slice buffer hand-off + visibility of the parked unit */                                    \
    } while (0)

    // prologue: slice 0 -> LDS, slices 1..DEPTH -> ring sets (slice % DEPTH)
    PT_LOAD(R0);
    PT_STORE(smem, R0);
#if DN_PT_DEPTH == 3
    if (T > 1) { PT_ADVANCE(); PT_LOAD(R1); }
    if (T > 2) { PT_ADVANCE(); PT_LOAD(R2); }
    if (T > 3) { PT_ADVANCE(); PT_LOAD(R0); }
    __syncthreads();
    for (int j = 0; j < T; j += 3) {
        PT_ITER(j, R1);
        if (j + 1 < T) PT_ITER(j + 1, R2);
        if (j + 2 < T) PT_ITER(j + 2, R0);
    }
#elif DN_PT_DEPTH == 2
    if (T > 1) { PT_ADVANCE(); PT_LOAD(R1); }
    if (T > 2) { PT_ADVANCE(); PT_LOAD(R0); }
    __syncthreads();
    for (int j = 0; j < T; j += 2) {
        PT_ITER(j, R1);
        if (j + 1 < T) PT_ITER(j + 1, R0);
    }
#else
    if (T > 1) { PT_ADVANCE(); PT_LOAD(R0); }
    __syncthreads();
    for (int j = 0; j < T; ++j) PT_ITER(j, R0);
#endif
#undef PT_ITER
#undef PT_STORE
#undef PT_LOAD
#undef PT_ADVANCE
    // flush what is still parked (the last unit)
#if defined(DN_PT_ABLATE_OUT)
    p_next = DN_PT_NP - 1;
#endif
    for (; p_next < DN_PT_NP; ++p_next) {
        PtPiece P1;
        pt_piece_load<MODE, FLAG>(g, sE, p_next, tid, p_row0, p_nrows, n0, P1);
        pt_piece_store<MODE, FLAG>(g, P1);
    }
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
#ifndef DN_WS_LW
#define DN_WS_LW 8                            // loader waves per workgroup (4 or 8); measured: 4 loader waves made the loaders the pole
#endif
#ifndef DN_WS_DEPTH
#define DN_WS_DEPTH 1                         // register sets of the loader's prefetch ring
#endif
#ifndef DN_WS_LOADER_PRIO
#define DN_WS_LOADER_PRIO 0   // s_setprio of the loader waves (the MFMA waves are the older ones and win arbitration at equal priority)
#endif
#ifndef DN_WS_SPLIT_SIMD
#define DN_WS_SPLIT_SIMD 0   // 1: MFMA waves on two SIMDs, loaders on the other two -- measured slower (58 vs 51 us, C->C product)
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

struct WsAux {
    float4 a0;
    uint32_t mk;
    float rs;
    long long off;
    int lds;      // float index of the piece in the parked unit
    bool ok;
};

// issue the auxiliary loads of one deferred piece (nothing here is used before the next slice iteration)
template <int MODE, bool FLAG>
__device__ __forceinline__ void ws_aux_load(const RgArgs& g, int piece, int lt, int row0, int nrows, int n0, WsAux& A) {
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
    if (MODE == DN_EPI_BIAS_RELU && FLAG) {   // explicit mask or drawn bits (see pt_piece_load)
        const uint32_t ld = *reinterpret_cast<const uint32_t*>(g.mask ? g.mask + roff : reinterpret_cast<const uint8_t*>(g.bias));
        A.mk = g.mask ? ld : dn_keep_bytes(dn_keep_bits(g.rng_seed, grow, ccol >> 2, (g.N + 3) >> 2));
    }
    if (MODE == DN_EPI_MASS_ADD) A.rs = g.rowv[grow];
}

template <int MODE, bool FLAG>
__device__ __forceinline__ void ws_piece_out(const RgArgs& g, const float* sE, const float4& bias, const WsAux& A) {
    PtPiece P;
    P.v = *reinterpret_cast<const float4*>(&sE[A.lds]);
    P.a0 = A.a0; P.bias = bias; P.mk = A.mk; P.rs = A.rs; P.off = A.off; P.ok = A.ok;
    pt_piece_store<MODE, FLAG>(g, P);
}

// six cross products of one k16 step, product-major: consecutive MFMAs go to different accumulators
__device__ __forceinline__ void ws_mma(const X3Frags<2, 2, 1>& F, int s, f32x16 (&acc)[1][2][2]) {
    constexpr int PA[6] = {1, 0, 2, 0, 1, 0}, PB[6] = {1, 2, 0, 1, 0, 0};   // mid*mid, hi*lo, lo*hi, hi*mid, mid*hi, hi*hi
#pragma unroll
    for (int p = 0; p < 6; ++p)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
#if defined(DN_X3_ABLATE_MFMA)
                acc[0][mt][nt][0] += __uint_as_float((F.a[s][PA[p]][mt].x & F.b[s][0][PB[p]][nt].x) & 0x3f800000u);
#else
                acc[0][mt][nt] = dn_mfma_bf16(F.a[s][PA[p]][mt], F.b[s][0][PB[p]][nt], acc[0][mt][nt]);
#endif
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
template <int MODE, bool BCOLK, bool FLAG, int PPI, bool BC>
__global__ __launch_bounds__(256 + DN_WS_LTHR) DN_WAVES_PER_EU(DN_WS_LW == 4 ? 2 : 3) void rowgemm_ws_kernel(RgArgs g, int ntiles) {

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

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int G = gridDim.x;
    const int n0 = blockIdx.y * TN;
    int nsl = 0;
    for (int s = 0; s < g.nseg; ++s) nsl += g.a[s].w / DN_KB;   // host guarantees nsl >= 4 and whole slices
    const int my_units = ((int)blockIdx.x < ntiles) ? (ntiles - (int)blockIdx.x + G - 1) / G : 0;
    const int T = my_units * nsl;
    if (T == 0) return;
#if defined(DN_PT_TRACE)
    int trn = 0;
#endif

    // Roles.  Default: waves 0-3 (one per SIMD) multiply, waves 4-11 load.  DN_WS_SPLIT_SIMD=1 instead puts the four MFMA
    // waves on two SIMDs (a workgroup's waves are dealt to the SIMDs cyclically) and leaves the other two to the loaders.
#if DN_WS_SPLIT_SIMD
    const bool is_mfma = wave < 8 && (wave & 3) < 2;
    const int mw = wave < 4 ? wave : wave - 2;          // 0, 1, 4, 5 -> 0..3
    const int lw = wave < 4 ? wave - 2 : wave - 4;      // 2, 3, 6, 7, 8..11 -> 0..7
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
        __syncthreads();   // slice 0 staged
        for (int j = 0; j < T; ++j) {
            const unsigned char* cA = reinterpret_cast<const unsigned char*>(smem + (j & 1) * SBUF);
            const unsigned char* cB = cA + SA * 4;
            X3Frags<2, 2, 1> F;
            WS_T(0);
            rg_frag_x3<2, 2, 1>(cA, cB, wr * 64, wc * 64, li, lg, 0, F);
            rg_frag_x3<2, 2, 1>(cA, cB, wr * 64, wc * 64, li, lg, 1, F);
            ws_mma(F, 0, acc);
            WS_T(0);
            ws_mma(F, 1, acc);
            WS_T(0);
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
            WS_T(0);
            __syncthreads();
        }
        return;
    }

    // ---------------------------------------------------- loader waves ----------------------------------------------------
    DN_SETPRIO(DN_WS_LOADER_PRIO);
    const int lt = lw * 64 + lane;
    float4 bias = dn_f4_zero();
    {
        const int col = n0 + 4 * (lt & 31);
        if (need_bias) bias = *reinterpret_cast<const float4*>(g.bias + (col < g.N ? col : 0));
    }
    RgRegs<NOUT, A_IT, B_IT> R0;
#if DN_WS_DEPTH == 2
    RgRegs<NOUT, A_IT, B_IT> R1;
#endif
    // segment descriptors in registers (nseg <= 3); the B operand of segment s starts koff = 0 again
    const float* sp0 = g.a[0].p; const float* sp1 = g.a[1].p; const float* sp2 = g.a[2].p;
    const int sl0 = g.a[0].ld, sl1 = g.a[1].ld, sl2 = g.a[2].ld;
    const int sw0 = g.a[0].w, sw1 = g.a[1].w, sw2 = g.a[2].w;
    const float* sb0 = g.b[0][0]; const float* sb1 = g.b[0][1]; const float* sb2 = g.b[0][2];
    const int nseg = g.nseg, ldb = g.ldb, Ncols = g.N;
    const long long bms = g.b_mesh_stride;
    // load cursor; the next unit's tile descriptor is fetched one unit ahead
    int lu = blockIdx.x, lseg = 0, lkoff = 0;
    DnTile ltile = g.tiles[lu];
    DnTile ltile_next = g.tiles[lu + G < ntiles ? lu + G : lu];
    // mirror of the compute cursor (which unit is parked when) and the parked unit being streamed out
    int cu = blockIdx.x, cs = 0;
    DnTile ctile = ltile, ctile_next = ltile_next;
    int p_row0 = ctile.row0, p_nrows = 0, p_next = DN_WS_NP;   // p_next >= NP: nothing pending
    const bool piece_wave = lt < DN_WS_PTHR;
    WsAux AX[PPI];
#pragma unroll
    for (int k = 0; k < PPI; ++k) ws_aux_load<MODE, FLAG>(g, DN_WS_NP, lt, p_row0, p_nrows, n0, AX[k]);   // dead pieces

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
        ltile_next = g.tiles[lu + G < ntiles ? lu + G : lu];   /* consumed at the next unit switch at the earliest */    \
    } while (0)
#define WS_LOAD(RS)                                                                                                     \
    ws_load<BCOLK, A_IT, B_IT, true, !BC>(lseg == 0 ? sp0 : (lseg == 1 ? sp1 : sp2), lseg == 0 ? sl0 : (lseg == 1 ? sl1 : sl2), \
                               (lseg == 0 ? sb0 : (lseg == 1 ? sb1 : sb2)) + (long long)ltile.mesh * bms, ldb, Ncols,    \
                               ltile.row0, ltile.nrows, n0, lkoff, lt, RS)
#define WS_STAGE(buf, RS, SIDX)                                                                                         \
    do {                                                                                                                \
        X3Planes<NOUT, A_IT, B_IT> PLN;                                                                                 \
        rg_split_x3<NOUT, BCOLK, HASQ, A_IT, B_IT, true, !BC>(RS, PLN);                                                 \
        if constexpr (BC) {                                                                                             \
            _Pragma("unroll") for (int i_ = 0; i_ < B_IT; ++i_)                                                         \
                _Pragma("unroll") for (int p_ = 0; p_ < 3; ++p_) PLN.b[0][i_][p_] = Bc[SIDX][i_][p_];                   \
        }                                                                                                               \
        rg_put_x3<LTHR, NOUT, BCOLK, A_IT, B_IT>(reinterpret_cast<unsigned char*>(buf),                                 \
                                                 reinterpret_cast<unsigned char*>((buf) + SA), lt, PLN);                \
    } while (0)

// Order inside an iteration: stage -> deferred pieces (their operands were requested an iteration ago) -> operands of the
// next iteration's pieces -> slice prefetch.  (Measured: a second register set / fetching two slices ahead, and requesting
// the piece operands before the prefetch, were both slower -- 60/48/135 us vs 53/51/127 us for the NN, C->C, 3C->C products.)
#define WS_ITER(j, RS, SIDX)                                                                                            \
    do {                                                                                                                \
        float* nxt = smem + (((j) & 1) ^ 1) * SBUF;                                                                     \
        WS_T(DN_WS_TRACE_TID);                                                                                                      \
        WS_STAGE(nxt, RS, SIDX);       /* slice j+1 (the last iteration stages a stale copy nobody reads) */            \
        WS_T(DN_WS_TRACE_TID);                                                                                                      \
        if (piece_wave) {              /* wave-uniform: only the first DN_WS_PW loader waves stream the parked unit out */ \
            _Pragma("unroll") for (int k = 0; k < PPI; ++k) ws_piece_out<MODE, FLAG>(g, sE, bias, AX[k]);               \
            WS_T(DN_WS_TRACE_TID);                                                                                      \
            p_next = (p_next + PPI < DN_WS_NP) ? p_next + PPI : DN_WS_NP;                                               \
            {   /* the MFMA waves park unit cu at the end of the iteration that multiplies its last slice */            \
                const bool park = ++cs == nsl;                                                                          \
                p_row0 = park ? ctile.row0 : p_row0; p_nrows = park ? ctile.nrows : p_nrows;                            \
                p_next = park ? (DN_PT_OUT_START ? DN_WS_NP : 0) : p_next;                                              \
                cs = park ? 0 : cs;                                                                                     \
                cu = park ? cu + G : cu;                                                                                \
                ctile.row0 = park ? ctile_next.row0 : ctile.row0; ctile.nrows = park ? ctile_next.nrows : ctile.nrows;  \
                const int cn = cu + G < ntiles ? cu + G : ntiles - 1;                                                   \
                ctile_next = g.tiles[cn];   /* consumed at the next park at the earliest */                             \
            }                                                                                                           \
            _Pragma("unroll") for (int k = 0; k < PPI; ++k)                                                             \
                ws_aux_load<MODE, FLAG>(g, p_next + k, lt, p_row0, p_nrows, n0, AX[k]);                                 \
        }                                                                                                               \
        WS_ADVANCE((j) + 1 + DN_WS_DEPTH < T);                                                                          \
        WS_LOAD(RS);                   /* slice j+1+DEPTH */                                                            \
        WS_T(DN_WS_TRACE_TID);                                                                                          \
        __syncthreads();                                                                                                \
    } while (0)

    uint2 Bc[BC ? 4 : 1][B_IT][3];
    if constexpr (BC) {   // split the whole B strip of this workgroup once (4 slices of the one segment)
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            RgRegs<NOUT, A_IT, B_IT> Rb;
            ws_load<BCOLK, A_IT, B_IT, false, true>(sp0, sl0, sb0, ldb, Ncols, 0, 0, n0, DN_KB * s4, lt, Rb);
            X3Planes<NOUT, A_IT, B_IT> Pb;
            rg_split_x3<NOUT, BCOLK, HASQ, A_IT, B_IT, false, true>(Rb, Pb);
#pragma unroll
            for (int i = 0; i < B_IT; ++i)
#pragma unroll
                for (int p3 = 0; p3 < 3; ++p3) Bc[s4][i][p3] = Pb.b[0][i][p3];
        }
    }
    WS_LOAD(R0);
    WS_STAGE(smem, R0, 0);
#if DN_WS_DEPTH == 2
    static_assert(!BC, "the cached-B form is written for the one-set prefetch");
    WS_ADVANCE(T > 1);
    WS_LOAD(R1);                       // slice 1
    WS_ADVANCE(T > 2);
    WS_LOAD(R0);                       // slice 2
    __syncthreads();                   // slice 0 staged
    for (int j = 0; j < T; j += 2) {
        WS_ITER(j, R1, 0);
        if (j + 1 < T) WS_ITER(j + 1, R0, 0);
    }
#else
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
#endif
#undef WS_ITER
#undef WS_STAGE
#undef WS_LOAD
#undef WS_ADVANCE
    // flush the last parked unit
#if defined(DN_PT_ABLATE_OUT)
    p_next = DN_WS_NP;
#endif
    if (!piece_wave) return;
    for (; p_next < DN_WS_NP; ++p_next) {
        WsAux A1;
        ws_aux_load<MODE, FLAG>(g, p_next, lt, p_row0, p_nrows, n0, A1);
        ws_piece_out<MODE, FLAG>(g, sE, bias, A1);
    }
}

template <int MODE, bool BCOLK, bool FLAG, int PPI, bool BC>
static int ws_launch(const RgArgs& g, int ntiles, hipStream_t stream) {
    const size_t smem = (size_t)(2 * (DN_TM * 64 * 3 + 128 * 64 * 3) + 128 * 128 * 4);   // 160 KiB
#ifndef DN_EMULATE
    static unsigned long long lds_opt_in = 0;   // per-device bitmap
    dn_lds_opt_in(reinterpret_cast<const void*>(&rowgemm_ws_kernel<MODE, BCOLK, FLAG, PPI, BC>), smem, &lds_opt_in);
#endif
    int gx = dn_num_cus();
    if (gx > ntiles) gx = ntiles;
    DN_LAUNCH((rowgemm_ws_kernel<MODE, BCOLK, FLAG, PPI, BC>), dim3(gx, (g.N + 127) / 128, 1), dim3(256 + DN_WS_LTHR, 1, 1), smem, stream, g, ntiles);
    return (int)hipGetLastError();
}

// ---- wave-specialised TWO-output row GEMM (the gradient-feature products, split-bf16) --------------------------------------
// Same roles as rowgemm_ws_kernel.  A work unit is 128 rows x 64 output columns of BOTH outputs, so that two slice buffers
// (A planes 24 KiB + 2 x 12 KiB B planes each) and the two parked 128 x 64 accumulator tiles (2 x 32 KiB) fill the 160 KiB of
// LDS exactly; the column halves of a row tile are consecutive units of the same workgroup (the second pass over the A rows is
// an L2 / MALL hit).  The epilogue runs on parked float4 pieces: the lock-step kernel it replaces did it with per-element
// dword loads and stores after its main loop, with one workgroup per CU and nothing to overlap them with.
#ifndef DN_RG_WS2
#define DN_RG_WS2 0   // measured on MI355X: 228 us vs 187 us for the lock-step two-output kernel (gradient features, 158k rows) -> off
#endif
#define DN_WS2_NP (128 * 64 / 4 / 512)   // float4 pieces (of each output) per loader thread and unit (4)

struct Ws2Aux {
    float4 r0, r1, r2;
    long long off;
    int lds;
    bool ok;
};

template <int MODE>
__device__ __forceinline__ void ws2_aux_load(const RgArgs& g, int piece, int lt, int row0, int nrows, int n0, Ws2Aux& A) {
    const bool live = piece < DN_WS2_NP;
    const int idx = lt + (live ? piece : 0) * 512;
    const int row = idx >> 4, c4 = idx & 15;
    const int col = n0 + 4 * c4;
    A.ok = live && row < nrows && col < g.N;
    A.lds = row * 64 + 4 * c4;
    const long long grow = row0 + (A.ok ? row : 0);
    const int ccol = A.ok ? col : 0;
    A.off = grow * g.ldo + ccol;
    const long long roff = grow * g.ldr + ccol;
    A.r0 = *reinterpret_cast<const float4*>(g.r0 + roff);
    A.r1 = *reinterpret_cast<const float4*>(g.r1 + roff);
    if (MODE == DN_EPI_GRADFEAT_BWD) A.r2 = *reinterpret_cast<const float4*>(g.r2 + roff);
}

template <int MODE>
__device__ __forceinline__ void ws2_piece_out(const RgArgs& g, const float* sE0, const float* sE1, const Ws2Aux& A) {
    const float4 a0 = *reinterpret_cast<const float4*>(&sE0[A.lds]);
    const float4 a1 = *reinterpret_cast<const float4*>(&sE1[A.lds]);
    if (MODE == DN_EPI_GRADFEAT) {   // o0 = tanh(r0*acc0 + r1*acc1); o1 = acc0; o2 = acc1 (if wanted)
        const float4 y = make_float4(tanhf(A.r0.x * a0.x + A.r1.x * a1.x), tanhf(A.r0.y * a0.y + A.r1.y * a1.y),
                                     tanhf(A.r0.z * a0.z + A.r1.z * a1.z), tanhf(A.r0.w * a0.w + A.r1.w * a1.w));
        if (A.ok) {
            *reinterpret_cast<float4*>(g.o0 + A.off) = y;
            if (g.o1) {
                *reinterpret_cast<float4*>(g.o1 + A.off) = a0;
                *reinterpret_cast<float4*>(g.o2 + A.off) = a1;
            }
        }
    } else {                         // o0 = acc0 + r0*r1 ; o1 = acc1 + r0*r2
        const float4 y0 = make_float4(a0.x + A.r0.x * A.r1.x, a0.y + A.r0.y * A.r1.y, a0.z + A.r0.z * A.r1.z, a0.w + A.r0.w * A.r1.w);
        const float4 y1 = make_float4(a1.x + A.r0.x * A.r2.x, a1.y + A.r0.y * A.r2.y, a1.z + A.r0.z * A.r2.z, a1.w + A.r0.w * A.r2.w);
        if (A.ok) {
            *reinterpret_cast<float4*>(g.o0 + A.off) = y0;
            *reinterpret_cast<float4*>(g.o1 + A.off) = y1;
        }
    }
}

template <int MODE, bool BCOLK>
__global__ __launch_bounds__(768) DN_WAVES_PER_EU(3) void rowgemm_ws2_kernel(RgArgs g, int ntiles) {
    constexpr bool HASQ = MODE == DN_EPI_GRADFEAT_BWD;   // its A operand is an elementwise product
    constexpr int PLA = 128 * 64, PLB = 64 * 64;         // bytes of one A / B plane of a slice
    constexpr int SA_B = 3 * PLA, SBUF_B = SA_B + 2 * 3 * PLB;   // 24 KiB + 24 KiB per slice buffer

    DN_DYN_SMEM(smem_raw);
    unsigned char* smem = reinterpret_cast<unsigned char*>(smem_raw);
    float* sE0 = reinterpret_cast<float*>(smem + 2 * SBUF_B);   // parked accumulators [128][64] of output 0 ...
    float* sE1 = sE0 + 128 * 64;                                 // ... and of output 1

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int G = gridDim.x;
    const int NH = (g.N + 63) / 64;
    int nsl = 0;
    for (int s = 0; s < g.nseg; ++s) nsl += g.a[s].w / DN_KB;   // host guarantees nsl >= 5 and whole slices
    const int my_tiles = ((int)blockIdx.x < ntiles) ? (ntiles - (int)blockIdx.x + G - 1) / G : 0;
    const int T = my_tiles * NH * nsl;
    if (T == 0) return;

    if (wave < 4) {
        // ------------------------------------------------ MFMA waves ------------------------------------------------
        const int wr = wave >> 1, wc = wave & 1;   // 64 rows x 32 columns of both outputs
        const int li = lane & 31, lg = lane >> 5;
        f32x16 acc[2][2];
#pragma unroll
        for (int o = 0; o < 2; ++o)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[o][mt][r] = 0.f;
        int cs = 0;
        __syncthreads();   // slice 0 staged
        for (int j = 0; j < T; ++j) {
            const unsigned char* cA = smem + (j & 1) * SBUF_B;
            const unsigned char* cB = cA + SA_B;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                uint4 a[3][2], b[2][3];
#pragma unroll
                for (int p = 0; p < 3; ++p) {
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
                        a[p][mt] = *reinterpret_cast<const uint4*>(cA + p * PLA + dn_plane_off(wr * 64 + mt * 32 + li, 2 * s + lg));
#pragma unroll
                    for (int o = 0; o < 2; ++o)
                        b[o][p] = *reinterpret_cast<const uint4*>(cB + (o * 3 + p) * PLB + dn_plane_off(wc * 32 + li, 2 * s + lg));
                }
                constexpr int PA[6] = {1, 0, 2, 0, 1, 0}, PB[6] = {1, 2, 0, 1, 0, 0};   // smallest terms first
#pragma unroll
                for (int p = 0; p < 6; ++p)
#pragma unroll
                    for (int o = 0; o < 2; ++o)
#pragma unroll
                        for (int mt = 0; mt < 2; ++mt) acc[o][mt] = dn_mfma_bf16(a[PA[p]][mt], b[o][PB[p]], acc[o][mt]);
            }
            if (++cs == nsl) {   // unit complete: park both accumulator tiles
                cs = 0;
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int e = (wr * 64 + mt * 32 + dn_acc_row(r, lane)) * 64 + wc * 32 + li;
                        sE0[e] = acc[0][mt][r]; sE1[e] = acc[1][mt][r];
                        acc[0][mt][r] = 0.f; acc[1][mt][r] = 0.f;
                    }
            }
            __syncthreads();
        }
        return;
    }

    // ---------------------------------------------------- loader waves ----------------------------------------------------
    DN_SETPRIO(DN_WS_LOADER_PRIO);
    const int lt = tid - 256;
    const int ob = lt >> 8, l8 = lt & 255;   // each loader thread stages the B operand of ONE output
    // segment descriptors in registers (two segments in both users; a third is carried for generality)
    const float* sp0 = g.a[0].p; const float* sp1 = g.a[1].p; const float* sp2 = g.a[2].p;
    const float* sq0 = g.a[0].q; const float* sq1 = g.a[1].q; const float* sq2 = g.a[2].q;
    const int sl0 = g.a[0].ld, sl1 = g.a[1].ld, sl2 = g.a[2].ld;
    const int sw0 = g.a[0].w, sw1 = g.a[1].w, sw2 = g.a[2].w;
    const float* sb0 = g.b[ob][0]; const float* sb1 = g.b[ob][1]; const float* sb2 = g.b[ob][2];
    const float sg0 = g.bsign[ob][0], sg1 = g.bsign[ob][1], sg2 = g.bsign[ob][2];
    const int nseg = g.nseg, ldb = g.ldb, Ncols = g.N;
    const long long bms = g.b_mesh_stride;
    // load cursor: tile, column half, segment, k offset; the next tile's descriptor is fetched one tile ahead
    int lti = blockIdx.x, lh = 0, lseg = 0, lkoff = 0;
    DnTile ltile = g.tiles[lti];
    DnTile ltile_next = g.tiles[lti + G < ntiles ? lti + G : lti];
    // mirror of the compute cursor and the parked unit being streamed out
    int cti = blockIdx.x, ch = 0, cs = 0;
    DnTile ctile = ltile, ctile_next = ltile_next;
    int p_row0 = ctile.row0, p_nrows = 0, p_n0 = 0, p_next = DN_WS2_NP;
    Ws2Aux AX;
    ws2_aux_load<MODE>(g, DN_WS2_NP, lt, p_row0, p_nrows, p_n0, AX);   // dead piece

    float4 Ra[2], Rq[2], Rb[2];
    float Rsg = 0.f;

#define WS2_ADVANCE(commit)                                                                                             \
    do {                                                                                                                \
        const int cw_ = lseg == 0 ? sw0 : (lseg == 1 ? sw1 : sw2);                                                      \
        int nk_ = lkoff + DN_KB, ns_ = lseg, nh_ = lh, nt_ = lti;                                                       \
        const bool se_ = nk_ >= cw_;                                                                                    \
        nk_ = se_ ? 0 : nk_;                                                                                            \
        ns_ = se_ ? ns_ + 1 : ns_;                                                                                      \
        const bool ue_ = ns_ >= nseg;                                                                                   \
        ns_ = ue_ ? 0 : ns_;                                                                                            \
        nh_ = ue_ ? nh_ + 1 : nh_;                                                                                      \
        const bool te_ = nh_ >= NH;                                                                                     \
        nh_ = te_ ? 0 : nh_;                                                                                            \
        nt_ = te_ ? nt_ + G : nt_;                                                                                      \
        const bool ok_ = (commit) && nt_ < ntiles;                                                                      \
        const bool sw_ = ok_ && te_;                                                                                    \
        lkoff = ok_ ? nk_ : lkoff; lseg = ok_ ? ns_ : lseg; lh = ok_ ? nh_ : lh; lti = ok_ ? nt_ : lti;                 \
        ltile.row0 = sw_ ? ltile_next.row0 : ltile.row0; ltile.nrows = sw_ ? ltile_next.nrows : ltile.nrows;            \
        ltile.mesh = sw_ ? ltile_next.mesh : ltile.mesh;                                                                \
        ltile_next = g.tiles[lti + G < ntiles ? lti + G : lti];                                                         \
    } while (0)
#define WS2_LOAD()                                                                                                      \
    do {                                                                                                                \
        const float* ap_ = lseg == 0 ? sp0 : (lseg == 1 ? sp1 : sp2);                                                   \
        const float* aq_ = lseg == 0 ? sq0 : (lseg == 1 ? sq1 : sq2);                                                   \
        const int ald_ = lseg == 0 ? sl0 : (lseg == 1 ? sl1 : sl2);                                                     \
        const float* bp_ = (lseg == 0 ? sb0 : (lseg == 1 ? sb1 : sb2)) + (long long)ltile.mesh * bms;                   \
        Rsg = lseg == 0 ? sg0 : (lseg == 1 ? sg1 : sg2);                                                                \
        const int n0_ = lh * 64;                                                                                        \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                                 \
            const int idx = lt + i * 512;                                                                               \
            const int row = idx >> 3, q = idx & 7;                                                                      \
            const long long off = (long long)(ltile.row0 + (row < ltile.nrows ? row : 0)) * ald_ + lkoff + 4 * q;       \
            Ra[i] = *reinterpret_cast<const float4*>(ap_ + off);                                                        \
            if (HASQ) Rq[i] = *reinterpret_cast<const float4*>(aq_ + off);                                              \
        }                                                                                                               \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                                 \
            long long boff;                                                                                             \
            if (BCOLK) {                                                                                                \
                const int idx = l8 + i * 256;                                                                           \
                const int nrow = idx >> 3, q = idx & 7;                                                                 \
                boff = (long long)(n0_ + nrow < Ncols ? n0_ + nrow : 0) * ldb + lkoff + 4 * q;                          \
            } else {                                                                                                    \
                const int pr = l8 & 15, q4 = l8 >> 4;                                                                   \
                boff = (long long)(lkoff + 2 * pr + i) * ldb + (n0_ + 4 * q4 < Ncols ? n0_ + 4 * q4 : 0);               \
            }                                                                                                           \
            Rb[i] = *reinterpret_cast<const float4*>(bp_ + boff);                                                       \
        }                                                                                                               \
    } while (0)
#define WS2_STAGE(buf)                                                                                                  \
    do {                                                                                                                \
        unsigned char* sA_ = (buf);                                                                                     \
        unsigned char* sB_ = (buf) + SA_B + ob * 3 * PLB;                                                               \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                                 \
            const int idx = lt + i * 512;                                                                               \
            const int row = idx >> 3, q = idx & 7;                                                                      \
            float4 v = Ra[i];                                                                                           \
            if (HASQ) v = dn_f4_mul(v, Rq[i]);                                                                          \
            uint2 h_, m_, l_;                                                                                           \
            dn_split3_f4(v, h_, m_, l_);                                                                                \
            const int off = dn_plane_off(row, q >> 1) + (q & 1) * 8;                                                    \
            *reinterpret_cast<uint2*>(sA_ + off) = h_;                                                                  \
            *reinterpret_cast<uint2*>(sA_ + PLA + off) = m_;                                                            \
            *reinterpret_cast<uint2*>(sA_ + 2 * PLA + off) = l_;                                                        \
        }                                                                                                               \
        if (BCOLK) {                                                                                                    \
            _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                             \
                const int idx = l8 + i * 256;                                                                           \
                const int nrow = idx >> 3, q = idx & 7;                                                                 \
                uint2 h_, m_, l_;                                                                                       \
                dn_split3_f4(dn_f4_scale(Rb[i], Rsg), h_, m_, l_);                                                      \
                const int off = dn_plane_off(nrow, q >> 1) + (q & 1) * 8;                                               \
                *reinterpret_cast<uint2*>(sB_ + off) = h_;                                                              \
                *reinterpret_cast<uint2*>(sB_ + PLB + off) = m_;                                                        \
                *reinterpret_cast<uint2*>(sB_ + 2 * PLB + off) = l_;                                                    \
            }                                                                                                           \
        } else {   /* rows 2p and 2p+1 of column group q4 -> packed (k, k+1) dwords of the transposed planes */          \
            const int pr = l8 & 15, q4 = l8 >> 4;                                                                       \
            const float4 v0 = dn_f4_scale(Rb[0], Rsg), v1 = dn_f4_scale(Rb[1], Rsg);                                    \
            const float e0[4] = {v0.x, v0.y, v0.z, v0.w}, e1[4] = {v1.x, v1.y, v1.z, v1.w};                             \
            _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                             \
                unsigned h_, m_, l_;                                                                                    \
                dn_split3_pair(e0[e], e1[e], h_, m_, l_);                                                               \
                const int off = dn_plane_off(4 * q4 + e, pr >> 2) + (pr & 3) * 4;                                       \
                *reinterpret_cast<unsigned*>(sB_ + off) = h_;                                                           \
                *reinterpret_cast<unsigned*>(sB_ + PLB + off) = m_;                                                     \
                *reinterpret_cast<unsigned*>(sB_ + 2 * PLB + off) = l_;                                                 \
            }                                                                                                           \
        }                                                                                                               \
    } while (0)

    WS2_LOAD();
    WS2_STAGE(smem);
    WS2_ADVANCE(T > 1);
    WS2_LOAD();                        // slice 1
    __syncthreads();                   // slice 0 staged
    for (int j = 0; j < T; ++j) {
        WS2_STAGE(smem + ((j & 1) ^ 1) * SBUF_B);   // slice j+1 (the last iteration stages a stale copy nobody reads)
        WS2_ADVANCE(j + 2 < T);
        WS2_LOAD();                    // slice j+2
        ws2_piece_out<MODE>(g, sE0, sE1, AX);   // its operands were requested an iteration ago
        p_next = (p_next + 1 < DN_WS2_NP) ? p_next + 1 : DN_WS2_NP;
        {   // the MFMA waves park the unit whose last slice they multiply in this iteration (selects only)
            const bool park = ++cs == nsl;
            p_row0 = park ? ctile.row0 : p_row0; p_nrows = park ? ctile.nrows : p_nrows; p_n0 = park ? ch * 64 : p_n0;
            p_next = park ? 0 : p_next;
            cs = park ? 0 : cs;
            const int nh = park ? ch + 1 : ch;
            const bool te = nh >= NH;
            ch = te ? 0 : nh;
            cti = te ? cti + G : cti;
            ctile.row0 = te ? ctile_next.row0 : ctile.row0; ctile.nrows = te ? ctile_next.nrows : ctile.nrows;
            const int cn = cti + G < ntiles ? cti + G : ntiles - 1;
            ctile_next = g.tiles[cn];
        }
        ws2_aux_load<MODE>(g, p_next, lt, p_row0, p_nrows, p_n0, AX);
        __syncthreads();
    }
#undef WS2_STAGE
#undef WS2_LOAD
#undef WS2_ADVANCE
    for (; p_next < DN_WS2_NP; ++p_next) {   // flush the last parked unit
        Ws2Aux A1;
        ws2_aux_load<MODE>(g, p_next, lt, p_row0, p_nrows, p_n0, A1);
        ws2_piece_out<MODE>(g, sE0, sE1, A1);
    }
}

template <int MODE, bool BCOLK>
static int ws2_launch(const RgArgs& g, int ntiles, hipStream_t stream) {
    const size_t smem = (size_t)(2 * (3 * 128 * 64 + 6 * 64 * 64) + 2 * 128 * 64 * 4);   // 160 KiB
#ifndef DN_EMULATE
    static unsigned long long lds_opt_in = 0;   // per-device bitmap
    dn_lds_opt_in(reinterpret_cast<const void*>(&rowgemm_ws2_kernel<MODE, BCOLK>), smem, &lds_opt_in);
#endif
    int gx = dn_num_cus();
    if (gx > ntiles) gx = ntiles;
    DN_LAUNCH((rowgemm_ws2_kernel<MODE, BCOLK>), dim3(gx, 1, 1), dim3(768, 1, 1), smem, stream, g, ntiles);
    return (int)hipGetLastError();
}

// eligibility of the two-output wave-specialised path
static bool ws2_eligible(const RgArgs& g, int nout) {
    if (!DN_RG_WS2 || !DN_RG_X3 || nout != 2 || !g.aligned || g.N < 64 || g.N % 4 != 0 || g.ldo % 4 != 0 || g.ldr % 4 != 0) return false;
    if (g.mode != DN_EPI_GRADFEAT && g.mode != DN_EPI_GRADFEAT_BWD) return false;
    int nsl = 0;
    for (int s = 0; s < g.nseg; ++s) {
        nsl += g.a[s].w / DN_KB;
        if ((g.a[s].q != nullptr) != (g.mode == DN_EPI_GRADFEAT_BWD)) return false;
    }
    if (nsl < 5) return false;   // one deferred piece per slice must drain a parked unit: (nsl - 1) >= DN_WS2_NP
    auto al = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
    if (!g.r0 || !g.r1 || !al(g.o0) || !al(g.o1) || !al(g.o2) || !al(g.r0) || !al(g.r1) || !al(g.r2)) return false;
    if (g.mode == DN_EPI_GRADFEAT_BWD && (!g.r2 || !g.o1)) return false;
    if ((g.o1 == nullptr) != (g.o2 == nullptr) && g.mode == DN_EPI_GRADFEAT) return false;
    return true;
}

template <int MODE, bool BCOLK, bool FLAG>
static int pt_launch(const RgArgs& g, int ntiles, hipStream_t stream) {
    constexpr bool X3 = DN_PT_X3 != 0;
    if (X3 && DN_PT_WS && DN_PT_ROWS == 128) {
        int nsl = 0;
        for (int s = 0; s < g.nseg; ++s) nsl += g.a[s].w / DN_KB;
        // PPI = ceil(NP / (nsl - 1)) for the two slice counts that matter (K = 128: 4 slices, K = 384: 12)
        if (nsl >= 9) return ws_launch<MODE, BCOLK, FLAG, (DN_WS_NP + 7) / 8, false>(g, ntiles, stream);
        if (DN_WS_BCACHE && DN_WS_DEPTH == 1 && nsl == 4 && g.nseg == 1 && g.b_mesh_stride == 0)
            return ws_launch<MODE, BCOLK, FLAG, (DN_WS_NP + 2) / 3, true>(g, ntiles, stream);
        if (nsl >= 4) return ws_launch<MODE, BCOLK, FLAG, (DN_WS_NP + 2) / 3, false>(g, ntiles, stream);
    }
    // slice buffers (2x) + parked accumulators: 128 KiB with f32 tiles, exactly 160 KiB with bf16x3 planes
    const size_t smem = X3 ? (size_t)(2 * (DN_PT_ROWS * 64 * 3 + 128 * 64 * 3) + DN_PT_ROWS * 128 * 4)
                           : (size_t)(2 * (DN_PT_ROWS * DN_KB + DN_KB * 128) + DN_PT_ROWS * 128) * sizeof(float);
#ifndef DN_EMULATE
    static unsigned long long lds_opt_in = 0;   // per-device bitmap
    dn_lds_opt_in(reinterpret_cast<const void*>(&rowgemm_persist_kernel<MODE, BCOLK, FLAG, X3>), smem, &lds_opt_in);
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
    if (ws2_eligible(g, nout)) {
        if (g.mode == DN_EPI_GRADFEAT) *err = ck ? ws2_launch<DN_EPI_GRADFEAT, true>(g, ntiles, stream) : ws2_launch<DN_EPI_GRADFEAT, false>(g, ntiles, stream);
        else *err = ck ? ws2_launch<DN_EPI_GRADFEAT_BWD, true>(g, ntiles, stream) : ws2_launch<DN_EPI_GRADFEAT_BWD, false>(g, ntiles, stream);
        return true;
    }
    return false;
}
