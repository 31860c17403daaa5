// dn_tngemm.hip -- split-V "TN" GEMM  partial[chunk][m, n] = sum_{r in chunk} A[r, m] * B[r, n]: the contraction runs over the
// vertex axis, split into chunks; partials are reduced afterwards in fixed order (deterministic, no float atomics).
// Replaces: geometry.to_basis (geometry.py:582-583) and every weight gradient dW = dY^T X of backward.
// Two kernels: exact-f32 MFMA (any shape) and split-bf16 MFMA with k-major planes read through ds_read_b64_tr_b16.
#include "dn_tn_tiles.h"
#include <string.h>

// =======================================================================================
// tngemm
// =======================================================================================
#define DN_TO 128  // output tile edge (both m and n)

// value of the virtually concatenated operand at (row, col); col is resolved to its segment
__device__ __forceinline__ float tn_elem(const TnSeg* s, int ns, long long row, int col) {
    int c = col;
    for (int i = 0; i < ns; ++i) {
        if (c < s[i].w) {
            const long long off = row * s[i].ld + c;
            float v = s[i].p[off];
            if (s[i].q) v *= s[i].q[off];
            return v;
        }
        c -= s[i].w;
    }
    return 0.f;
}


struct TnRegs {
    float4 a[4], b[4], qa[4];
    float ma[4], mb[4];
};

// raw loads of one 32-row step (nothing is used here, so the loads stay in flight under the MFMAs)
template <bool ALIGNED, int FLAVOR>
__device__ __forceinline__ void tn_load(const TnArgs& g, const DnTile& ch, int step, int kr0, int acol, int bcol, bool a_ok,
                                        bool b_ok, const float* ap, const float* aq, int ald, const float* bp, int bld,
                                        TnRegs& R) {
    if (ALIGNED) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int kr = step * DN_KB + kr0 + 8 * i;
            const bool kok = kr < ch.nrows;
            const long long row = (long long)ch.row0 + (kok ? kr : 0);
            R.ma[i] = (kok && a_ok) ? 1.f : 0.f;
            R.mb[i] = (kok && b_ok) ? 1.f : 0.f;
            R.a[i] = *reinterpret_cast<const float4*>(ap + row * ald);
            R.b[i] = *reinterpret_cast<const float4*>(bp + row * bld);
            if (FLAVOR == DN_TN_QA) R.qa[i] = *reinterpret_cast<const float4*>(aq + row * ald);
            if (FLAVOR == DN_TN_ROWSCALE) R.qa[i].x = g.b_rowscale[row];
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int kr = step * DN_KB + kr0 + 8 * i;
            const long long row = (long long)ch.row0 + kr;
            float e[4] = {0.f, 0.f, 0.f, 0.f}, f[4] = {0.f, 0.f, 0.f, 0.f};
            if (kr < ch.nrows) {
                const float rs = g.b_rowscale ? g.b_rowscale[row] : 1.f;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if (acol + c < g.M) e[c] = tn_elem(g.a, g.na, row, acol + c);
                    if (bcol + c < g.N) f[c] = tn_elem(g.b, g.nb, row, bcol + c) * rs;
                }
            }
            R.a[i] = make_float4(e[0], e[1], e[2], e[3]);
            R.b[i] = make_float4(f[0], f[1], f[2], f[3]);
            R.ma[i] = 1.f;
            R.mb[i] = 1.f;
            if (FLAVOR == DN_TN_QA) R.qa[i] = make_float4(1.f, 1.f, 1.f, 1.f);
            if (FLAVOR == DN_TN_ROWSCALE) R.qa[i].x = 1.f;
        }
    }
}

template <int FLAVOR>
__device__ __forceinline__ void tn_store(float* sA, float* sB, int kr0, int q, const TnRegs& R, float4& csum) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float4 va = dn_f4_scale(R.a[i], R.ma[i]);
        float4 vb = dn_f4_scale(R.b[i], FLAVOR == DN_TN_ROWSCALE ? R.mb[i] * R.qa[i].x : R.mb[i]);
        if (FLAVOR == DN_TN_QA) va = dn_f4_mul(va, R.qa[i]);
        if (FLAVOR == DN_TN_COLSUM) { csum.x += va.x; csum.y += va.y; csum.z += va.z; csum.w += va.w; }
        const int kr = kr0 + 8 * i;
        *reinterpret_cast<float4*>(&sA[kr * DN_TO + 4 * q]) = va;
        *reinterpret_cast<float4*>(&sB[kr * DN_TO + 4 * q]) = vb;
    }
}

__device__ __forceinline__ void tn_compute(const float* sA, const float* sB, int wr, int wc, int li, int ls, f32x16 (&acc)[2][2]) {
#pragma unroll
    for (int kg = 0; kg < 4; ++kg) {
        float af[2][4], bf[2][4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int kr = 8 * kg + 4 * ls + t;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                af[i][t] = sA[kr * DN_TO + (wr * 2 + i) * 32 + li];
                bf[i][t] = sB[kr * DN_TO + (wc * 2 + i) * 32 + li];
            }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = dn_mfma(af[i][t], bf[j][t], acc[i][j]);
    }
}

template <bool ALIGNED, int FLAVOR>
__global__ __launch_bounds__(256) void tngemm_kernel(TnArgs g) {
    constexpr int SBUF = 2 * DN_KB * DN_TO;   // floats of one (A,B) step buffer; two buffers in LDS
    DN_DYN_SMEM(smem_raw);
    float* smem = reinterpret_cast<float*>(smem_raw);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int li = lane & 31, ls = lane >> 5;
    const int n0 = blockIdx.y * DN_TO, m0 = blockIdx.z * DN_TO;
    const bool do_colsum = FLAVOR == DN_TN_COLSUM && blockIdx.y == 0;
    const bool wave_active = (m0 + wr * 64 < g.M) && (n0 + wc * 64 < g.N);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // each thread always stages the same 4-column group (q) of both operands; on the aligned path its segment is
    // resolved once, out-of-range groups read a valid (clamped) address and are zeroed by the 0/1 factor
    const int q = tid & 31, kr0 = tid >> 5;
    const int acol = m0 + 4 * q, bcol = n0 + 4 * q;
    const float* ap = g.a[0].p; const float* aq = g.a[0].q ? g.a[0].q : g.a[0].p; int ald = g.a[0].ld;
    const float* bp = g.b[0].p; int bld = g.b[0].ld;
    const bool a_ok = acol < g.M, b_ok = bcol < g.N;
    if (ALIGNED) {
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
    float4 csum = dn_f4_zero();
    TnRegs R;

    const int c_beg = blockIdx.x * g.group;
    const int c_end = (c_beg + g.group < g.nchunks) ? c_beg + g.group : g.nchunks;
    for (int ci = c_beg; ci < c_end; ++ci) {
        const DnTile ch = g.chunks[ci];
        const int nsteps = (ch.nrows + DN_KB - 1) / DN_KB;
        // pipeline: regs(step+1) -> LDS[other]; loads(step+2) -> regs; MFMAs on LDS[cur]; ONE barrier per step
        tn_load<ALIGNED, FLAVOR>(g, ch, 0, kr0, acol, bcol, a_ok, b_ok, ap, aq, ald, bp, bld, R);
        tn_store<FLAVOR>(smem, smem + DN_KB * DN_TO, kr0, q, R, csum);
        if (nsteps > 1) tn_load<ALIGNED, FLAVOR>(g, ch, 1, kr0, acol, bcol, a_ok, b_ok, ap, aq, ald, bp, bld, R);
        __syncthreads();
        int st = 0;
        for (; st + 2 < nsteps; ++st) {
            float* cur = smem + (st & 1) * SBUF;
            float* nxt = smem + ((st & 1) ^ 1) * SBUF;
            tn_store<FLAVOR>(nxt, nxt + DN_KB * DN_TO, kr0, q, R, csum);
            tn_load<ALIGNED, FLAVOR>(g, ch, st + 2, kr0, acol, bcol, a_ok, b_ok, ap, aq, ald, bp, bld, R);
            tn_compute(cur, cur + DN_KB * DN_TO, wr, wc, li, ls, acc);
            __syncthreads();
        }
        if (st + 1 < nsteps) {
            float* cur = smem + (st & 1) * SBUF;
            float* nxt = smem + ((st & 1) ^ 1) * SBUF;
            tn_store<FLAVOR>(nxt, nxt + DN_KB * DN_TO, kr0, q, R, csum);
            tn_compute(cur, cur + DN_KB * DN_TO, wr, wc, li, ls, acc);
            __syncthreads();
            ++st;
        }
        {
            float* cur = smem + (st & 1) * SBUF;
            tn_compute(cur, cur + DN_KB * DN_TO, wr, wc, li, ls, acc);
            __syncthreads();   // the next chunk's prologue overwrites buffer 0
        }
    }
    // partial tile out
    float* out = g.partial + (long long)blockIdx.x * g.M * g.N;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (!wave_active) continue;
            const int n = n0 + (wc * 2 + j) * 32 + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + (wr * 2 + i) * 32 + dn_acc_row(r, lane);
                if (m < g.M && n < g.N) out[(long long)m * g.N + n] = acc[i][j][r];
            }
        }
    if (do_colsum) {   // uniform per block
        *reinterpret_cast<float4*>(&smem[kr0 * DN_TO + 4 * q]) = csum;
        __syncthreads();
        if (tid < DN_TO && m0 + tid < g.M) {
            float sum = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) sum += smem[k * DN_TO + tid];
            g.colsum[(long long)blockIdx.x * g.M + m0 + tid] = sum;
        }
    }
}

template <bool ALIGNED, int FLAVOR>
static int tn_launch(const TnArgs& g, dim3 grid, hipStream_t stream) {
    const size_t smem = (size_t)2 * 2 * DN_KB * DN_TO * sizeof(float);
    DN_LAUNCH((tngemm_kernel<ALIGNED, FLAVOR>), grid, dim3(256, 1, 1), smem, stream, g);
    return (int)hipGetLastError();
}

// =======================================================================================
// tngemm on split-bf16 MFMA (aligned operands).  The operands are k-major (the contraction runs over the rows r of
// A[r, m] and B[r, n]), so their bf16 planes are stored as loaded -- [32 r][128 cols], 320-byte rows: four consecutive
// rows start 16 banks apart -- and turned into MFMA operands by the LDS transpose read.  8 waves (64 x 32 outputs each),
// two 60 KiB step buffers.
// =======================================================================================
// (staging / fragment / MFMA helpers of this kernel: dn_tn_tiles.h, shared with the fused diffusion kernel)
// single 60 KiB step buffer, two workgroups per CU (measured 60.1 vs 62.2 us against a double-buffered one-workgroup form)
// body of the kernel for workgroup (bx, by, bz) of problem g (the plain kernel passes its blockIdx; the multi-problem kernel the indices of
// the workgroup inside its problem)
template <int FLAVOR, int NP>
__device__ __forceinline__ void tn_x3_body(const TnArgs& g, const int bx, const int by, const int bz) {
    float sa = 1.f, sb = 1.f, so = 1.f;
    if constexpr (NP == 2) {   // split-fp16: operand scales from the producers' amax words, exact inverse on the way out
        sa = dn_pow2_scale(dn_amax_eval(g.a_amax));
        sb = dn_pow2_scale(dn_amax_eval(g.b_amax));
        so = (1.f / sa) * (1.f / sb);
    }

    constexpr int SBUF = 6 * DN_TX_PLANE;   // bytes of one (A,B) step buffer (3 planes each); two buffers in LDS
    DN_DYN_SMEM(smem_raw);
    unsigned char* smem = reinterpret_cast<unsigned char*>(smem_raw);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 2, wc = wave & 3;           // 2 x 4 waves, 64 x 32 outputs each
    const int li = lane & 31;
    const int n0 = by * DN_TO, m0 = bz * DN_TO;
    const bool do_colsum = FLAVOR == DN_TN_COLSUM && by == 0;
    const bool wave_active = (m0 + wr * 64 < g.M) && (n0 + wc * 32 < g.N);

    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    const int q = tid & 31, kr0 = tid >> 5;            // this thread stages column group q of rows kr0, kr0 + 16
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
    float4 csum = dn_f4_zero();
    TxRegs R;

    const int c_beg = bx * g.group;
    const int c_end = (c_beg + g.group < g.nchunks) ? c_beg + g.group : g.nchunks;
    for (int ci = c_beg; ci < c_end; ++ci) {
        const DnTile ch = g.chunks[ci];
        const int nsteps = (ch.nrows + DN_KB - 1) / DN_KB;
        // single 60 KiB step buffer, two barriers per step: two workgroups (16 waves) share a CU and cover each other's
        // staging phases and HBM latency
        tx_load<FLAVOR>(g, ch, 0, kr0, a_ok, b_ok, ap, aq, ald, bp, bld, R);
        tx_store<FLAVOR, NP>(smem, smem + 3 * DN_TX_PLANE, kr0, q, R, csum, sa, sb);
        __syncthreads();
        for (int st = 0; st < nsteps; ++st) {
            if (st + 1 < nsteps) tx_load<FLAVOR>(g, ch, st + 1, kr0, a_ok, b_ok, ap, aq, ald, bp, bld, R);
            tx_compute<NP>(smem, smem + 3 * DN_TX_PLANE, wr, wc, lane, acc);
            __syncthreads();
            if (st + 1 < nsteps) {
                tx_store<FLAVOR, NP>(smem, smem + 3 * DN_TX_PLANE, kr0, q, R, csum, sa, sb);
                __syncthreads();
            }
        }
    }
    float* out = g.partial + (long long)bx * g.M * g.N;
    if (wave_active) {
        const int n = n0 + wc * 32 + li;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + (wr * 2 + i) * 32 + dn_acc_row(r, lane);
                if (m < g.M && n < g.N) out[(long long)m * g.N + n] = NP == 2 ? acc[i][r] * so : acc[i][r];
            }
    }
    if (do_colsum) {   // uniform per block: 16 row lanes x 32 column groups -> [16][128] floats in LDS
        float* red = reinterpret_cast<float*>(smem);
        *reinterpret_cast<float4*>(&red[kr0 * DN_TO + 4 * q]) = csum;
        __syncthreads();
        if (tid < DN_TO && m0 + tid < g.M) {
            float sum = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) sum += red[k * DN_TO + tid];
            g.colsum[(long long)bx * g.M + m0 + tid] = sum;
        }
    }
}

#ifndef DN_TN_XCD
#define DN_TN_XCD 1   // products with several output tiles per chunk block (K or C > 128): the tiles of a chunk block run on ONE XCD, back to back (0: the
#endif                // three-dimensional grid, chunk-major: a chunk's tiles a whole sweep apart; A/B)
template <int FLAVOR, int NP>
__global__ __launch_bounds__(DN_TX_THREADS, 4) void tngemm_x3_kernel(TnArgs g) {
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (g.lin_ny > 0) {
        // The tiles of a chunk block multiply the same rows of both operands (at K = C = 256: four tiles, every row of Phi and of x read by two
        // of them).  Chunk-major the second read comes a whole sweep later -- from memory once the operands exceed the Infinity Cache (0.41 GB for
        // one 200k-vertex mesh).  Here workgroups 8 s + x, s = nt c + t, take tile t of chunk block 8 c + x: round-robin dispatch (blockIdx & 7)
        // sends a chunk block's tiles to one XCD one after the other and its rows are fetched into that XCD's L2 once.
        const int nt = g.lin_ny * g.lin_nz, nb = g.lin_nb, local = (int)blockIdx.x, nfull = nb & ~7;
        int tile;
        if (local < nfull * nt) { const int seq = local >> 3; bx = 8 * (seq / nt) + (local & 7); tile = seq % nt; }
        else { const int t = local - nfull * nt, rem = nb - nfull; bx = nfull + t % rem; tile = t / rem; }
        by = tile % g.lin_ny; bz = tile / g.lin_ny;
    }
    tn_x3_body<FLAVOR, NP>(g, bx, by, bz);
}
// several independent products in ONE launch (the three weight-gradient products of a block's backward: one ramp and one tail instead of
// three, and a single mesh's handful of chunks share the device): workgroup blockIdx.x belongs to the problem whose range it falls into
#define DN_TN_MULTI 3
struct TnMulti { TnArgs p[DN_TN_MULTI]; int first[DN_TN_MULTI + 1]; int nblk[DN_TN_MULTI]; int ny[DN_TN_MULTI]; int count; };
DN_CLK_DECLARE(tn_multi)
template <int FLAVOR, int NP>
__global__ __launch_bounds__(DN_TX_THREADS, 4) void tngemm_x3_multi_kernel(TnMulti mm) {
    DN_CLK_STAMP(tn_multi, 0);
    int pi = 0;
#pragma unroll
    for (int i = 1; i < DN_TN_MULTI; ++i) pi = (i < mm.count && (int)blockIdx.x >= mm.first[i]) ? i : pi;
    const int local = (int)blockIdx.x - mm.first[pi];
    // (selection instead of dynamic indexing of the by-value argument block: an indexed copy would land in scratch)
    if (pi == 0) tn_x3_body<FLAVOR, NP>(mm.p[0], local % mm.nblk[0], (local / mm.nblk[0]) % mm.ny[0], local / (mm.nblk[0] * mm.ny[0]));
    else if (pi == 1) tn_x3_body<FLAVOR, NP>(mm.p[1], local % mm.nblk[1], (local / mm.nblk[1]) % mm.ny[1], local / (mm.nblk[1] * mm.ny[1]));
    else tn_x3_body<FLAVOR, NP>(mm.p[2], local % mm.nblk[2], (local / mm.nblk[2]) % mm.ny[2], local / (mm.nblk[2] * mm.ny[2]));
    DN_CLK_STAMP(tn_multi, 1);
}

template <int FLAVOR, int NP>
static int tx_launch_np(const TnArgs& g, dim3 grid, hipStream_t stream) {
    const size_t smem = (size_t)6 * DN_TX_PLANE;   // 60 KiB (the split-fp16 form uses four of the six planes)
#ifndef DN_EMULATE
    static unsigned long long lds_opt_in = 0;   // per-device bitmap
    { const int oe_ = dn_lds_opt_in(reinterpret_cast<const void*>(&tngemm_x3_kernel<FLAVOR, NP>), smem, &lds_opt_in); if (oe_) return oe_; }
#endif
    DN_LAUNCH((tngemm_x3_kernel<FLAVOR, NP>), grid, dim3(DN_TX_THREADS, 1, 1), smem, stream, g);
    return (int)hipGetLastError();
}
template <int FLAVOR>
static int tx_launch(const TnArgs& g, dim3 grid, hipStream_t stream) {
    return g.f16 ? tx_launch_np<FLAVOR, 2>(g, grid, stream) : tx_launch_np<FLAVOR, 3>(g, grid, stream);
}

#ifndef DN_TN_X3
#define DN_TN_X3 1   // -DDN_TN_X3=0: exact-f32 MFMA in the split-V kernels
#endif


// returns the number of partials written (= gridDim.x) through *npartial
int dn_launch_tngemm(const TnArgs& g_in, int nchunks, hipStream_t stream) {
    if (nchunks <= 0 || g_in.M <= 0 || g_in.N <= 0) return 0;
    TnArgs g = g_in;
    g.nchunks = nchunks;
    if (g.group < 1) g.group = 1;
    const int nblk = (nchunks + g.group - 1) / g.group;
    dim3 grid(nblk, (g.N + DN_TO - 1) / DN_TO, (g.M + DN_TO - 1) / DN_TO);
    // flavour: which optional operand treatment the launch needs (at most one is ever combined by the callers)
    bool has_qa = false;
    for (int i = 0; i < g.na; ++i) has_qa = has_qa || g.a[i].q != nullptr;
    for (int i = 0; i < g.nb; ++i) if (g.b[i].q) return DN_ERR_BAD_MODE;
    const int flavor = has_qa ? DN_TN_QA : (g.colsum ? DN_TN_COLSUM : (g.b_rowscale ? DN_TN_ROWSCALE : DN_TN_PLAIN));
    if ((has_qa && (g.colsum || g.b_rowscale)) || (g.colsum && g.b_rowscale)) return DN_ERR_BAD_MODE;
    const double rows = g.acct_rows;
    dn_prof_begin(DN_K_TNGEMM, stream);
    int err;
    if (g.aligned && DN_TN_X3) {
        g.lin_nb = g.lin_ny = g.lin_nz = 0;
        if (DN_TN_XCD && grid.y * grid.z > 1) {
            g.lin_nb = nblk; g.lin_ny = (int)grid.y; g.lin_nz = (int)grid.z;
            grid = dim3(nblk * grid.y * grid.z, 1, 1);
        }
        switch (flavor) {
            case DN_TN_QA: err = tx_launch<DN_TN_QA>(g, grid, stream); break;
            case DN_TN_COLSUM: err = tx_launch<DN_TN_COLSUM>(g, grid, stream); break;
            case DN_TN_ROWSCALE: err = tx_launch<DN_TN_ROWSCALE>(g, grid, stream); break;
            default: err = tx_launch<DN_TN_PLAIN>(g, grid, stream); break;
        }
    } else if (g.aligned) {
        switch (flavor) {
            case DN_TN_QA: err = tn_launch<true, DN_TN_QA>(g, grid, stream); break;
            case DN_TN_COLSUM: err = tn_launch<true, DN_TN_COLSUM>(g, grid, stream); break;
            case DN_TN_ROWSCALE: err = tn_launch<true, DN_TN_ROWSCALE>(g, grid, stream); break;
            default: err = tn_launch<true, DN_TN_PLAIN>(g, grid, stream); break;
        }
    } else {
        switch (flavor) {
            case DN_TN_QA: err = tn_launch<false, DN_TN_QA>(g, grid, stream); break;
            case DN_TN_COLSUM: err = tn_launch<false, DN_TN_COLSUM>(g, grid, stream); break;
            case DN_TN_ROWSCALE: err = tn_launch<false, DN_TN_ROWSCALE>(g, grid, stream); break;
            default: err = tn_launch<false, DN_TN_PLAIN>(g, grid, stream); break;
        }
    }
    dn_prof_end(DN_K_TNGEMM, stream, 2.0 * rows * g.M * g.N,
                4.0 * (rows * (g.M + g.N) + (double)nblk * g.M * g.N));
    return err;
}

// Up to DN_TN_MULTI aligned, bias-column-sum ("COLSUM") split-bf16 products in one launch; anything else falls back to separate launches.
int dn_launch_tngemm_multi(const TnArgs* gs, const int* nchunks, int count, hipStream_t stream) {
    if (count <= 0) return 0;
    bool ok = count <= DN_TN_MULTI && DN_TN_X3 != 0;
    for (int i = 0; i < count && ok; ++i) {
        const TnArgs& g = gs[i];
        bool has_qa = false;
        for (int k = 0; k < g.na; ++k) has_qa = has_qa || g.a[k].q != nullptr;
        ok = g.aligned && !g.f16 && !has_qa && g.colsum && !g.b_rowscale && nchunks[i] > 0 && g.M > 0 && g.N > 0;
    }
    if (!ok || count == 1) {
        for (int i = 0; i < count; ++i) { const int e = dn_launch_tngemm(gs[i], nchunks[i], stream); if (e) return e; }
        return 0;
    }
    TnMulti mm;
    memset(&mm, 0, sizeof(mm));
    mm.count = count;
    double flops = 0.0, bytes = 0.0;
    int total = 0;
    for (int i = 0; i < count; ++i) {
        mm.p[i] = gs[i];
        mm.p[i].nchunks = nchunks[i];
        if (mm.p[i].group < 1) mm.p[i].group = 1;
        mm.nblk[i] = (nchunks[i] + mm.p[i].group - 1) / mm.p[i].group;
        mm.ny[i] = (gs[i].N + DN_TO - 1) / DN_TO;
        const int nz = (gs[i].M + DN_TO - 1) / DN_TO;
        mm.first[i] = total;
        total += mm.nblk[i] * mm.ny[i] * nz;
        const double rows = gs[i].acct_rows;
        flops += 2.0 * rows * gs[i].M * gs[i].N;
        bytes += 4.0 * (rows * (gs[i].M + gs[i].N) + (double)mm.nblk[i] * gs[i].M * gs[i].N);
    }
    mm.first[count] = total;
    const size_t smem = (size_t)6 * DN_TX_PLANE;
#ifndef DN_EMULATE
    static unsigned long long lds_opt_in = 0;   // per-device bitmap
    { const int oe_ = dn_lds_opt_in(reinterpret_cast<const void*>(&tngemm_x3_multi_kernel<DN_TN_COLSUM, 3>), smem, &lds_opt_in); if (oe_) return oe_; }
#endif
    dn_prof_begin(DN_K_TN_MULTI, stream);
    DN_LAUNCH((tngemm_x3_multi_kernel<DN_TN_COLSUM, 3>), dim3(total, 1, 1), dim3(DN_TX_THREADS, 1, 1), smem, stream, mm);
    dn_prof_end(DN_K_TN_MULTI, stream, flops, bytes);
    return (int)hipGetLastError();
}
