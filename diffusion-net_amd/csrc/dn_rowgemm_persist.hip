// dn_rowgemm_persist.hip -- persistent forms of the row GEMM (see dn_rowgemm.hip for the product): the lock-step persistent
// kernel (3-slice products) and the wave-specialised kernel (4 MFMA waves + 8 loader/epilogue waves; every one-output product with
// >= 4 slices).  The build knobs of rounds 1-3 (knock-outs, direct stores, slice pairs, resident B, prefetch depth, wave placement,
// priorities, phase traces) were measured and rejected one by one; their source lives in tools/experiments/rowgemm_ws_knobs/.
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
// Work-unit geometry: 128-row units, 8 waves, one workgroup per CU, split-bf16 planes.  (Measured on MI355X, K = N = 128 product, 158k rows:
// 68 us; 64-row units with 4 waves and two workgroups per CU: 71-75 us -- twice the B-operand staging per MFMA; non-persistent: 75-80 us.)
#define DN_PT_ROWS 128
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

template <int MODE, bool BCOLK, bool FLAG>
__global__ __launch_bounds__(DN_PT_THREADS) DN_WAVES_PER_EU(2) void rowgemm_persist_kernel(RgArgs g, int ntiles) {
    const unsigned long long seed = (MODE == DN_EPI_BIAS_RELU && FLAG) ? rg_seed(g) : 0ull;

    constexpr int TN = 128, WR = 2, WC = 4, NOUT = 1, NTHR = DN_PT_THREADS, TMU = DN_PT_ROWS;
    constexpr int MT = TMU / (32 * WR);              // 2
    constexpr int NT = TN / (32 * WC);               // 1
    constexpr int A_IT = TMU * 8 / NTHR;             // 2
    constexpr int B_IT = DN_KB * TN / 4 / NTHR;      // 2
    constexpr int SA = (TMU * 64 * 3) / 4;           // one (A,B) slice buffer, in floats: three bf16 planes each (6 B/elem)
    constexpr int SBUF = SA + (128 * 64 * 3) / 4;
    constexpr bool HASQ = false;
    constexpr bool PAIRK = !BCOLK;
    constexpr int PPI = 4;                           // deferred pieces per slice iteration: 8 pieces over 2 iterations
    static_assert(TMU == DN_TM && NTHR == 512, "the bf16x3 staging is written for 128-row units and 512 threads");

    DN_DYN_SMEM(smem_raw);
    float* smem = reinterpret_cast<float*>(smem_raw);
    float* sE = smem + 2 * SBUF;                     // [TMU][128] parked accumulators

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave / WC, wc = wave % WC;
    const int li = lane & 31, ls = lane >> 5;
    const int G = gridDim.x;
    const int n0 = blockIdx.y * TN;
    const int nunits = ntiles;

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

    auto unit_tile = [&](int u) { return g.tiles[u]; };

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
#define PT_STORE(buf, RS) rg_store_x3<NTHR, NOUT, BCOLK, HASQ, A_IT, B_IT>(reinterpret_cast<unsigned char*>(buf), reinterpret_cast<unsigned char*>((buf) + SA), tid, RS)
// One pipeline iteration on slice j; RS is the ring set of slice j+1 (staged now) and of slice j+1+DEPTH (loaded now).
// Split-bf16 order: every LDS read of slice j is issued first, then the pure-VALU split of slice j+1 and the MFMAs of slice j
// (independent instruction streams the scheduler can interleave), then the LDS writes of slice j+1.  With the writes first
// the MFMAs had to wait behind them (may-alias LDS), and a slice took ~5.8k cycles of which the matrix pipe was busy 1.5k.
#define PT_ITER(j, RS)                                                                                                  \
    do {                                                                                                                \
        float* cur = smem + ((j) & 1) * SBUF;                                                                           \
        float* nxt = smem + (((j) & 1) ^ 1) * SBUF;                                                                     \
        PtPiece P[PPI];                                                                                                 \
        {                                                                                                               \
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

#include "dn_rowgemm_ws.h"   // wave-specialised persistent row GEMM (every one-output product with >= 4 slices): rowgemm_ws_kernel, ws_launch

template <int MODE, bool BCOLK, bool FLAG>
static int pt_launch(const RgArgs& g, int ntiles, hipStream_t stream) {
    {
        int nsl = 0;
        for (int s = 0; s < g.nseg; ++s) nsl += g.a[s].w / DN_KB;
        // PPI = ceil(NP / (nsl - 1)) for the two slice counts that matter (K = 128: 4 slices, K = 384: 12)
        if (g.f16) {   // split-fp16 engine (the caller supplies the operand magnitudes)
            if (nsl >= 9) return ws_launch<MODE, BCOLK, FLAG, (DN_WS_NP + 7) / 8, false, 2>(g, ntiles, stream);
            if (nsl == 4 && g.nseg == 1 && g.b_mesh_stride == 0)   // (constant-B 4-slice products: the split B strip stays in loader registers)
                return ws_launch<MODE, BCOLK, FLAG, (DN_WS_NP + 2) / 3, true, 2>(g, ntiles, stream);
            if (nsl >= 4) return ws_launch<MODE, BCOLK, FLAG, (DN_WS_NP + 2) / 3, false, 2>(g, ntiles, stream);
        }
        if (nsl >= 9) return ws_launch<MODE, BCOLK, FLAG, (DN_WS_NP + 7) / 8, false>(g, ntiles, stream);
        if (nsl == 4 && g.nseg == 1 && g.b_mesh_stride == 0)
            return ws_launch<MODE, BCOLK, FLAG, (DN_WS_NP + 2) / 3, true>(g, ntiles, stream);
        if (nsl >= 4) return ws_launch<MODE, BCOLK, FLAG, (DN_WS_NP + 2) / 3, false>(g, ntiles, stream);
    }
    // 3-slice products: the lock-step kernel; slice buffers (2x) + parked accumulators: exactly 160 KiB with bf16x3 planes
    const size_t smem = (size_t)(2 * (DN_PT_ROWS * 64 * 3 + 128 * 64 * 3) + DN_PT_ROWS * 128 * 4);
#ifndef DN_EMULATE
    static unsigned long long lds_opt_in = 0;   // per-device bitmap
    { const int oe_ = dn_lds_opt_in(reinterpret_cast<const void*>(&rowgemm_persist_kernel<MODE, BCOLK, FLAG>), smem, &lds_opt_in); if (oe_) return oe_; }
#endif
    int gx = dn_num_cus();   // one workgroup per CU
    if (gx > ntiles) gx = ntiles;
    DN_LAUNCH((rowgemm_persist_kernel<MODE, BCOLK, FLAG>), dim3(gx, (g.N + 127) / 128, 1), dim3(DN_PT_THREADS, 1, 1), smem,
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
