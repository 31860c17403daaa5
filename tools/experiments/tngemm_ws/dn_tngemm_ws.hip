// dn_tngemm_ws.hip -- wave-specialised form of the split-V "TN" product (round 3)
//     partial[wg][m, n] = sum over the rows r of the workgroup's chunks of A[r, m] * B[r, n]            (see dn_tngemm.hip)
// for aligned operands: to_basis (geometry.py:582-583), the weight gradients dW = dY^T X and their bias column sums.
//
// The lock-step kernel (tngemm_x3_kernel: 8 waves that all stage, then all multiply; one 60 KiB step buffer, two barriers per
// 32-row step, two workgroups per CU) sits at 3.0 TB/s of a read-only stream (53 us for 162 MB).  Same cure as for the row products:
// 12 waves with fixed roles --
//   waves 0-3 (one per SIMD): transpose-read the k-major planes of step s and multiply a 64 x 64 sub-tile each (2 x 2 accumulators);
//   waves 4-11: fetch the rows of step s+2, split step s+1 into planes and write them to the other step buffer --
// two step buffers (2 x 60 KiB), ONE barrier per step, the request for step s+2 issued as soon as the registers of step s+1 have been
// split (it then has a whole step to fly), chunk descriptors as per-lane values (a uniform load would drag an s_waitcnt vmcnt(0)
// behind it).  One workgroup per CU, so a launch writes half the partial results of the lock-step form (the host groups chunks for
// ~one workgroup per CU): the fixed-order reductions that follow read half as much.
// Summation order per output element: rows ascending inside a workgroup (as the lock-step kernel), so results are bitwise those of
// the lock-step kernel whenever the grouping is the same.
#include "dn_tn_tiles.h"

#define DN_TW_LTHR 512                       // loader threads (8 waves)
#define DN_TW_THREADS (256 + DN_TW_LTHR)
#define DN_TW_BUF (6 * DN_TX_PLANE)          // bytes of one step buffer: A planes then B planes (3 planes each; split-fp16 uses 2 + 2)

template <int NP>
__device__ __forceinline__ void tw_mma(const unsigned char* sA, const unsigned char* sB, int wr, int wc, int lane, f32x16 (&acc)[2][2]) {
    const int g = lane >> 4, c = lane & 15;
    const int lane_off = (8 * (g >> 1) + (c >> 2)) * DN_TX_ROWB + (16 * (g & 1) + 4 * (c & 3)) * 2;
#pragma unroll
    for (int s = 0; s < 2; ++s) {   // two k16 steps per 32-row step
        uint4 a[NP][2], b[NP][2];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int base = p * DN_TX_PLANE + s * 16 * DN_TX_ROWB + lane_off;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                a[p][t] = tx_frag(sA + base + (wr * 64 + t * 32) * 2);
                b[p][t] = tx_frag(sB + base + (wc * 64 + t * 32) * 2);
            }
        }
        constexpr int NPROD = NP == 3 ? 6 : 3;
        constexpr int PA[6] = {NP == 3 ? 1 : 0, NP == 3 ? 0 : 1, NP == 3 ? 2 : 0, 0, 1, 0};
        constexpr int PB[6] = {1, NP == 3 ? 2 : 0, 0, 1, 0, 0};
#pragma unroll
        for (int p = 0; p < NPROD; ++p)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    if constexpr (NP == 3) acc[mt][nt] = dn_mfma_bf16(a[PA[p]][mt], b[PB[p]][nt], acc[mt][nt]);
                    else acc[mt][nt] = dn_mfma_f16(a[PA[p]][mt], b[PB[p]][nt], acc[mt][nt]);
                }
    }
}

template <int FLAVOR, int NP>
__global__ __launch_bounds__(DN_TW_THREADS) DN_WAVES_PER_EU(3) void tngemm_ws_kernel(TnArgs g) {
    DN_DYN_SMEM(smem_raw);
    unsigned char* smem = reinterpret_cast<unsigned char*>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = blockIdx.y * 128, m0 = blockIdx.z * 128;
    const int c_beg = blockIdx.x * g.group;
    const int c_end = (c_beg + g.group < g.nchunks) ? c_beg + g.group : g.nchunks;
    int T = 0;                                    // 32-row steps of this workgroup
    for (int ci = c_beg; ci < c_end; ++ci) T += (g.chunks[ci].nrows + DN_KB - 1) / DN_KB;
    const bool do_colsum = FLAVOR == DN_TN_COLSUM && blockIdx.y == 0;

    if (wave < 4) {
        // ------------------------------------------------ MFMA waves ------------------------------------------------
        const int wr = wave >> 1, wc = wave & 1;
        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        __syncthreads();                          // step 0 staged
        for (int st = 0; st < T; ++st) {
            const unsigned char* cur = smem + (st & 1) * DN_TW_BUF;
            tw_mma<NP>(cur, cur + 3 * DN_TX_PLANE, wr, wc, lane, acc);
            __syncthreads();
        }
        float so = 1.f;
        if constexpr (NP == 2) so = (1.f / dn_pow2_scale(dn_amax_eval(g.a_amax))) * (1.f / dn_pow2_scale(dn_amax_eval(g.b_amax)));
        float* out = g.partial + (long long)blockIdx.x * g.M * g.N;
        const int li = lane & 31;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int n = n0 + wc * 64 + j * 32 + li;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + wr * 64 + i * 32 + dn_acc_row(r, lane);
                    if (m < g.M && n < g.N) out[(long long)m * g.N + n] = NP == 2 ? acc[i][j][r] * so : acc[i][j][r];
                }
            }
        if (FLAVOR == DN_TN_COLSUM) { __syncthreads(); __syncthreads(); }   // (the loaders' column-sum exchange)
        return;
    }

    // ---------------------------------------------------- loader waves ----------------------------------------------------
    const int lt = tid - 256;
    const int q = lt & 31, kr0 = lt >> 5;              // this thread stages column group q of rows kr0, kr0 + 16 of a step
    float sa = 1.f, sb = 1.f;
    if constexpr (NP == 2) { sa = dn_pow2_scale(dn_amax_eval(g.a_amax)); sb = dn_pow2_scale(dn_amax_eval(g.b_amax)); }
    const int acol = m0 + 4 * q, bcol = n0 + 4 * q;
    const float* ap = g.a[0].p; const float* aq = g.a[0].q ? g.a[0].q : g.a[0].p; int ald = g.a[0].ld;
    const float* bp = g.b[0].p; int bld = g.b[0].ld;
    const bool a_ok = acol < g.M, b_ok = bcol < g.N;
    {
        int c = acol;
        for (int i = 0; i < g.na; ++i) {
            if (a_ok && c >= 0 && c < g.a[i].w) { ap = g.a[i].p + c; aq = (g.a[i].q ? g.a[i].q : g.a[i].p) + c; ald = g.a[i].ld; }
            c -= g.a[i].w;
        }
        c = bcol;
        for (int i = 0; i < g.nb; ++i) {
            if (b_ok && c >= 0 && c < g.b[i].w) { bp = g.b[i].p + c; bld = g.b[i].ld; }
            c -= g.b[i].w;
        }
    }
    // chunk descriptors per lane (every lane the same address, through a pointer the compiler cannot prove uniform)
    const DnTile* cl = g.chunks;
#ifndef DN_EMULATE
    { int vz_; asm volatile("v_mov_b32 %0, 0" : "=v"(vz_)); cl += vz_; }
#endif
    int lci = c_beg, lstep = 0;                        // load cursor: (chunk, step inside the chunk)
    DnTile lch = cl[lci];
    DnTile lch_next = cl[lci + 1 < c_end ? lci + 1 : lci];
    float4 csum = dn_f4_zero();
    TxRegs R;
    uint2 PA_[2][NP], PB_[2][NP];

#define TW_LOAD() tx_load<FLAVOR>(g, lch, lstep, kr0, a_ok, b_ok, ap, aq, ald, bp, bld, R)
// one step of the load cursor without control flow on the path; past the last step it stays put
#define TW_ADVANCE(commit)                                                                                              \
    do {                                                                                                                \
        const int ns_ = (lch.nrows + DN_KB - 1) / DN_KB;                                                                \
        const bool ce_ = lstep + 1 >= ns_;                                                                              \
        const bool ok_ = (commit);                                                                                      \
        const bool sw_ = ok_ && ce_;                                                                                    \
        lstep = ok_ ? (ce_ ? 0 : lstep + 1) : lstep;                                                                    \
        lci = sw_ ? lci + 1 : lci;                                                                                      \
        lch.row0 = sw_ ? lch_next.row0 : lch.row0; lch.nrows = sw_ ? lch_next.nrows : lch.nrows;                        \
        lch_next = cl[lci + 1 < c_end ? lci + 1 : lci];                                                                 \
    } while (0)
#define TW_SPLIT()                                                                                                      \
    do {                                                                                                                \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                                 \
            float4 va = dn_f4_scale(R.a[i], R.ma[i]);                                                                   \
            const float4 vb = dn_f4_scale(R.b[i], FLAVOR == DN_TN_ROWSCALE ? R.mb[i] * R.qa[i].x : R.mb[i]);            \
            if (FLAVOR == DN_TN_QA) va = dn_f4_mul(va, R.qa[i]);                                                        \
            if (FLAVOR == DN_TN_COLSUM && cs_on) { csum.x += va.x; csum.y += va.y; csum.z += va.z; csum.w += va.w; }    \
            dn_split_f4<NP>(va, sa, PA_[i]);                                                                            \
            dn_split_f4<NP>(vb, sb, PB_[i]);                                                                            \
        }                                                                                                               \
    } while (0)
#define TW_PUT(buf)                                                                                                     \
    do {                                                                                                                \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                                 \
            const int off = (kr0 + 16 * i) * DN_TX_ROWB + q * 8;                                                        \
            _Pragma("unroll") for (int p = 0; p < NP; ++p) {                                                            \
                *reinterpret_cast<uint2*>((buf) + p * DN_TX_PLANE + off) = PA_[i][p];                                   \
                *reinterpret_cast<uint2*>((buf) + (3 + p) * DN_TX_PLANE + off) = PB_[i][p];                             \
            }                                                                                                           \
        }                                                                                                               \
    } while (0)

    bool cs_on = true;              // the step being split is a real one (the last iteration re-splits the final step: must not count twice)
    TW_LOAD();                      // step 0
    TW_SPLIT();
    TW_ADVANCE(T > 1);
    TW_LOAD();                      // step 1
    TW_PUT(smem);
    __syncthreads();                // step 0 staged
    for (int st = 0; st < T; ++st) {
        unsigned char* nxt = smem + ((st & 1) ^ 1) * DN_TW_BUF;
        cs_on = st + 1 < T;
        TW_SPLIT();                 // step st+1 (the last iteration stages a stale copy nobody reads)
        TW_ADVANCE(st + 2 < T);
        TW_LOAD();                  // step st+2
        TW_PUT(nxt);
        __syncthreads();
    }
#undef TW_PUT
#undef TW_SPLIT
#undef TW_ADVANCE
#undef TW_LOAD
    if (FLAVOR == DN_TN_COLSUM) {   // 16 row lanes x 32 column groups -> [16][128] floats in LDS, summed by 128 threads
        float* red = reinterpret_cast<float*>(smem);
        __syncthreads();
        if (do_colsum) *reinterpret_cast<float4*>(&red[kr0 * 128 + 4 * q]) = csum;
        __syncthreads();
        if (do_colsum && lt < 128 && m0 + lt < g.M) {
            float sum = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) sum += red[k * 128 + lt];
            g.colsum[(long long)blockIdx.x * g.M + m0 + lt] = sum;
        }
    }
}

template <int FLAVOR, int NP>
static int tw_launch_np(const TnArgs& g, dim3 grid, hipStream_t stream) {
    const size_t smem = (size_t)2 * DN_TW_BUF;   // 120 KiB
#ifndef DN_EMULATE
    static unsigned long long lds_opt_in = 0;   // per-device bitmap
    { const int oe_ = dn_lds_opt_in(reinterpret_cast<const void*>(&tngemm_ws_kernel<FLAVOR, NP>), smem, &lds_opt_in); if (oe_) return oe_; }
#endif
    DN_LAUNCH((tngemm_ws_kernel<FLAVOR, NP>), grid, dim3(DN_TW_THREADS, 1, 1), smem, stream, g);
    return (int)hipGetLastError();
}

// returns false if this build does not use the wave-specialised form
bool dn_tngemm_try_ws(const TnArgs& g, int flavor, dim3 grid, hipStream_t stream, int* err) {
    if (!DN_TN_WS) return false;
#define TW_GO(F) (g.f16 ? tw_launch_np<F, 2>(g, grid, stream) : tw_launch_np<F, 3>(g, grid, stream))
    switch (flavor) {
        case DN_TN_QA: *err = TW_GO(DN_TN_QA); break;
        case DN_TN_COLSUM: *err = TW_GO(DN_TN_COLSUM); break;
        case DN_TN_ROWSCALE: *err = TW_GO(DN_TN_ROWSCALE); break;
        default: *err = TW_GO(DN_TN_PLAIN); break;
    }
#undef TW_GO
    return true;
}
