// dn_sparse.hip -- CSR gather kernels for the spatial-gradient operators (gfx950).
//
// Replaces the per-batch-item sparse COO `torch.mm(gradX[b], x)` / `torch.mm(gradY[b], x)` pair of
// layers.py:217-223 (and their autograd transposes) and the gather+mean output remaps of
// layers.py:379-391.  gradX and gradY share one CSR pattern (the reference builds both from one
// complex matrix, geometry.py:381-382), so one pass over the pattern produces both products and
// the dense operand row is fetched once.  Rows are split over 16-byte channel groups: a row of
// C=128 floats is handled by 32 consecutive lanes, each gathering one float4 per non-zero; column
// indices / values are wave-broadcast loads.  HBM/L2-bound (about 2 flop per byte).
#include "dn_common.h"
#ifndef DN_SP_XCD
#define DN_SP_XCD 1   // XCD-contiguous row blocks (block b runs on XCD b % 8): 76 vs 81 us on the bench batch (3 rounds)
#endif
#ifndef DN_SP_CHUNK
#ifndef DN_SP_COOP
#ifdef DN_EMULATE
#define DN_SP_COOP 0   // (the fiber emulator's shuffles are whole-wave barriers; row groups of a wave leave the entry loop at different times)
#else
#define DN_SP_COOP 1   // pattern entries of a row fetched once per row group and shared by shuffles (0: every lane loads them)
#endif
#endif
#define DN_SP_CHUNK 8   // non-zeros gathered per branch-free step (a mesh vertex has ~7 gradient entries)
#endif

// MODE is a compile-time copy of s.mode: with the mode tested at run time every unrolled gather step carried its own scalar branches
// AMAX: track max |output| and commit it to s.o_amax (measured: +25 % on the forward gather -- the waves are short-lived and every
// commit is a round trip to L2; the fused block therefore uses the analytic bound max|G x| <= ||G||_inf max|x| instead, stored by one
// thread: s.op_norm / s.in_amax).  The plain instantiation is the round-2 kernel unchanged (a grid-stride form of it was 50 % slower
// on the transposed gather).
template <int VEC, int MODE, bool AMAX>
__global__ __launch_bounds__(256) void spmm_kernel(SpArgs s, int tpr) {
    if (!AMAX && s.o_amax && blockIdx.x == 0 && threadIdx.x == 0)     // (atomic: the word is read with device-scope atomic loads)
        atomicMax(reinterpret_cast<unsigned*>(s.o_amax), __float_as_uint(dn_amax_word(s.in_amax) * dn_amax_word(s.op_norm)));

    const int tid = threadIdx.x;
    const int rl = tid / tpr, cg = tid % tpr;
    const int rows_per_block = 256 / tpr;
#if DN_SP_XCD   // every XCD walks a contiguous range of row blocks: the ~7 neighbour rows a row gathers mostly live in its L2
    const int per_xcd = (gridDim.x + 7) >> 3;
    const int rb = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    const int row = rb * rows_per_block + rl;
#else
    const int row = blockIdx.x * rows_per_block + rl;
#endif
    // cooperative index loads need every lane of the row's group alive in one wave (the row test below is uniform per group)
    // (and every lane of the group must run every column pass: C a multiple of the group's span)
    const bool coop = DN_SP_COOP && tpr >= DN_SP_CHUNK && tpr <= 64 && s.C % (tpr * VEC) == 0;
    if (!AMAX && row >= s.nrows) return;
    const bool live = row < s.nrows;      // AMAX: every lane stays for the wave-wide reduction at the end; dead rows gather nothing
    const int beg = live ? s.rowptr[row] : 0, end = live ? s.rowptr[row + 1] : 0;
    float om = 0.f;
    for (int c = cg * VEC; c < s.C; c += tpr * VEC) {
        float a1[VEC], a2[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) { a1[e] = 0.f; a2[e] = 0.f; }
        // DN_SP_CHUNK non-zeros per step, no branches inside: indices past the row's end are clamped to its last entry and
        // get weight 0, so all index loads and then all row gathers of a step are in flight together
        for (int j0 = beg; j0 < end; j0 += DN_SP_CHUNK) {
            long long src[DN_SP_CHUNK];
            float wa[DN_SP_CHUNK], wb[DN_SP_CHUNK];
            if (coop) {
                // The lanes of a row read the SAME DN_SP_CHUNK pattern entries: lane u of the row fetches entry u (one index load and
                // one or two value loads per wave instead of 3 x DN_SP_CHUNK broadcast loads) and the row shares them by shuffles.
                const bool mine = cg < DN_SP_CHUNK && j0 + cg < end;
                const int jm = mine ? j0 + cg : end - 1;
                const int colm = s.col[jm];
                const float vam = mine ? (s.va ? s.va[jm] : 1.f) : 0.f;
                const float vbm = (MODE != DN_SP_ONE && mine) ? s.vb[jm] : 0.f;
                const int gbase = (tid & 63) - cg;
#pragma unroll
                for (int u = 0; u < DN_SP_CHUNK; ++u) {
                    src[u] = (long long)__shfl(colm, gbase + u, 64) * s.ldx + c;
                    wa[u] = __shfl(vam, gbase + u, 64);
                    wb[u] = MODE != DN_SP_ONE ? __shfl(vbm, gbase + u, 64) : 0.f;
                }
            } else {
#pragma unroll
                for (int u = 0; u < DN_SP_CHUNK; ++u) {
                    const bool in = j0 + u < end;
                    const int j = in ? j0 + u : end - 1;
                    src[u] = (long long)s.col[j] * s.ldx + c;
                    wa[u] = in ? (s.va ? s.va[j] : 1.f) : 0.f;
                    wb[u] = (MODE != DN_SP_ONE && in) ? s.vb[j] : 0.f;
                }
            }
            float xv[DN_SP_CHUNK][VEC], yv[DN_SP_CHUNK][VEC];
#pragma unroll
            for (int u = 0; u < DN_SP_CHUNK; ++u) {
                if (VEC == 4) {
                    const float4 t = *reinterpret_cast<const float4*>(s.x1 + src[u]);
                    xv[u][0] = t.x; xv[u][1 % VEC] = t.y; xv[u][2 % VEC] = t.z; xv[u][3 % VEC] = t.w;
                } else {
                    xv[u][0] = s.x1[src[u]];
                }
                if (MODE == DN_SP_BWD2) {
                    if (VEC == 4) {
                        const float4 t = *reinterpret_cast<const float4*>(s.x2 + src[u]);
                        yv[u][0] = t.x; yv[u][1 % VEC] = t.y; yv[u][2 % VEC] = t.z; yv[u][3 % VEC] = t.w;
                    } else {
                        yv[u][0] = s.x2[src[u]];
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < DN_SP_CHUNK; ++u) {
                if (MODE == DN_SP_FWD2) {
#pragma unroll
                    for (int e = 0; e < VEC; ++e) { a1[e] = fmaf(wa[u], xv[u][e], a1[e]); a2[e] = fmaf(wb[u], xv[u][e], a2[e]); }
                } else if (MODE == DN_SP_BWD2) {
#pragma unroll
                    for (int e = 0; e < VEC; ++e) a1[e] = fmaf(wb[u], yv[u][e], fmaf(wa[u], xv[u][e], a1[e]));
                } else {
#pragma unroll
                    for (int e = 0; e < VEC; ++e) a1[e] = fmaf(wa[u], xv[u][e], a1[e]);
                }
            }
        }
        if (AMAX && !live) continue;
        const long long dst = (long long)row * s.ldo + c;
        if (MODE == DN_SP_ONE) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) a1[e] = a1[e] / s.div;
        }
        if (s.add) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) a1[e] += s.add[dst + e];
        }
        if (AMAX) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                om = fabsf(a1[e]) > om ? fabsf(a1[e]) : om;
                if (MODE == DN_SP_FWD2) om = fabsf(a2[e]) > om ? fabsf(a2[e]) : om;
            }
        }
        if (VEC == 4) {
            *reinterpret_cast<float4*>(s.o1 + dst) = make_float4(a1[0], a1[1 % VEC], a1[2 % VEC], a1[3 % VEC]);
            if (MODE == DN_SP_FWD2)
                *reinterpret_cast<float4*>(s.o2 + dst) = make_float4(a2[0], a2[1 % VEC], a2[2 % VEC], a2[3 % VEC]);
        } else {
            s.o1[dst] = a1[0];
            if (MODE == DN_SP_FWD2) s.o2[dst] = a2[0];
        }
    }
    if (AMAX) dn_amax_commit<true>(s.o_amax, om);
}

static int pow2_at_least(int v) {
    int p = 1;
    while (p < v) p <<= 1;
    return p;
}

int dn_launch_spmm(const SpArgs& s, hipStream_t stream) {
    if (s.nrows <= 0 || s.C <= 0) return 0;
    const bool vec = (s.C % 4 == 0) && (s.ldx % 4 == 0) && (s.ldo % 4 == 0) &&
                     ((uintptr_t)s.x1 % 16 == 0) && ((uintptr_t)s.o1 % 16 == 0) &&
                     (!s.x2 || (uintptr_t)s.x2 % 16 == 0) && (!s.o2 || (uintptr_t)s.o2 % 16 == 0) &&
                     (!s.add || (uintptr_t)s.add % 16 == 0);
    int tpr = pow2_at_least(vec ? (s.C + 3) / 4 : s.C);
    if (tpr > 256) tpr = 256;
    const int rpb = 256 / tpr;
    dim3 grid(DN_SP_XCD ? (((s.nrows + rpb - 1) / rpb) + 7) / 8 * 8 : (s.nrows + rpb - 1) / rpb, 1, 1);
    dn_prof_begin(DN_K_SPMM, stream);
    const bool amax = s.o_amax && !(s.op_norm && s.in_amax);
#define DN_SP_GO(V, M) do { if (amax) DN_LAUNCH((spmm_kernel<V, M, true>), grid, dim3(256, 1, 1), 0, stream, s, tpr); \
                            else DN_LAUNCH((spmm_kernel<V, M, false>), grid, dim3(256, 1, 1), 0, stream, s, tpr); } while (0)
    if (vec) {
        if (s.mode == DN_SP_FWD2) DN_SP_GO(4, DN_SP_FWD2); else if (s.mode == DN_SP_BWD2) DN_SP_GO(4, DN_SP_BWD2); else DN_SP_GO(4, DN_SP_ONE);
    } else {
        if (s.mode == DN_SP_FWD2) DN_SP_GO(1, DN_SP_FWD2); else if (s.mode == DN_SP_BWD2) DN_SP_GO(1, DN_SP_BWD2); else DN_SP_GO(1, DN_SP_ONE);
    }
#undef DN_SP_GO
    {
        const double nnz = (double)s.acct_nnz, nout = s.mode == DN_SP_FWD2 ? 2.0 : 1.0, nin = s.mode == DN_SP_BWD2 ? 2.0 : 1.0;
        dn_prof_end(DN_K_SPMM, stream, 2.0 * nnz * s.C * (s.mode == DN_SP_ONE ? 1.0 : 2.0),
                    4.0 * ((double)s.nrows + 1 + nnz * (s.mode == DN_SP_ONE ? 1.0 : 3.0) +
                           (double)s.nrows * s.C * (nin + nout + (s.add ? 1.0 : 0.0))));
    }
    return (int)hipGetLastError();
}
