// dn_rowgemm_direct.hip -- "direct" row GEMM for the K = 128 products (from_basis at K_eig = 128, every C -> C linear of the
// 128-wide net and their input gradients): out[r, n] = epi(sum_k A[r, k] * B(k, n)), n-tile of 128 columns per blockIdx.y.
//
// Why another kernel: in the slice-pipelined kernels (dn_rowgemm_persist.hip) every 32-wide slice of the LONG operand A makes
// a round trip global -> registers -> split -> LDS -> registers behind a workgroup barrier, and the SQ counters of round 1 show
// the waves parked on that barrier half of the time.  Here A never touches LDS:
//   * a wave owns a unit of 16 rows x all 128 output columns (eight 16x16 accumulators, v_mfma_f32_16x16x32_bf16).  The
//     fragment of the long operand that a lane feeds to an MFMA is eight k-consecutive floats of ONE row, i.e. two float4
//     global loads: the wave loads its own fragments straight from HBM (64 B contiguous per row and instruction, both
//     instructions of a k32 step together cover whole 128-B lines), splits them into the three bf16 terms in registers
//     (44 VALU per 48 MFMAs: inside the MFMA shadow) and multiplies;
//   * the SHORT operand B (a weight matrix, or the 128 x 128 spectrum of the mesh the rows belong to) is split ONCE per
//     workgroup and mesh into bf16 planes laid out in fragment order in LDS (96 KiB): every B fragment is one conflict-free
//     ds_read_b128 at a lane-linear address -- no transpose reads, no per-slice staging, and NO barrier in the main loop
//     (only around a B re-stage when the workgroup's row range crosses into another mesh);
//   * the MFMA is issued with the operands swapped (D^T = B^T A^T): a lane's four accumulator registers of a tile are then four
//     CONSECUTIVE output columns of one row, so the epilogue runs on float4 pieces straight from the accumulators (coalesced
//     16-byte auxiliary loads and stores, one dropout hash per piece) -- no parking of the result in LDS;
//   * eight independent waves per CU (two per SIMD: one's global waits and epilogue sit under the other's MFMAs), each with
//     its next two units (16 KiB) of A in flight -> 128 KiB of reads in flight per CU;
//   * the k index inside a 32-float line is permuted (the same permutation on both operands, so the sums run over the same set
//     of products): lane group g of a k32 step holds floats {4g .. 4g+3} and {16 + 4g .. +3} of the line.
// vmcnt is an in-order counter: the auxiliary operands of a unit's epilogue are requested at the TOP of the unit, before the
// prefetch loads issued during its MFMA steps, so that waiting for them never drains the prefetch.
// Products per accumulator and k32 step: the six largest cross terms of the 3-term split, smallest first (as everywhere).
#include "dn_gemm_tiles.h"

#ifndef DN_RD
#define DN_RD 1
#endif
#define DN_RD_WAVES 8
#define DN_RD_THREADS (64 * DN_RD_WAVES)
#define DN_RD_ROWS 16                  // rows of a wave's unit
#define DN_RD_LDS_B (4 * 8 * 3 * 1024)   // [k32 step][16-column tile][plane][lane] x 16 B
#define DN_RD_LDS (DN_RD_LDS_B + 512)    // + the 128 bias values of the column tile (read by ds_read: does not touch vmcnt)

#if defined(DN_RD_TRACE) && !defined(DN_EMULATE)   // development build only: s_memtime stamps of workgroup DN_RD_TRACE, all 8 waves (tools/kbench --trace)
__device__ unsigned long long dn_rd_trace_buf[8 * 64];
extern "C" int dn_debug_rd_trace_read(unsigned long long* out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(dn_rd_trace_buf), sizeof(unsigned long long) * n);
}
#define RD_T()                                                                                                          \
    do {                                                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                                              \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime();                                                     \
        if (blockIdx.x == (DN_RD_TRACE) && blockIdx.y == 0 && (threadIdx.x & 63) == 0 && trn < 64)                      \
            dn_rd_trace_buf[(threadIdx.x >> 6) * 64 + trn] = t_;                                                        \
        ++trn;                                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                              \
    } while (0)
#else
#define RD_T() do {} while (0)
#endif

typedef float f32x4 __attribute__((ext_vector_type(4)));
// One 16x16x32 bf16 MFMA step (fp32 accumulate): lane l supplies eight consecutive-k bf16 of row l&15 of the first operand and
// of column l&15 of the second, k = 8*(l>>4) .. +7; accumulator register r of lane l is D[4*(l>>4) + r][l&15].
__device__ __forceinline__ f32x4 dn_mfma_bf16_16(uint4 a, uint4 b, f32x4 c) {
#ifdef DN_EMULATE
    return dnemu_mfma_f32_16x16x32_bf16(a, b, c);
#else
    typedef __bf16 dn_bf16x8 __attribute__((ext_vector_type(8)));
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(dn_bf16x8, a), __builtin_bit_cast(dn_bf16x8, b), c, 0, 0, 0);
#endif
}

__device__ __forceinline__ void rd_split8(const float4& u, const float4& v, uint4& hi, uint4& mid, uint4& lo) {
    dn_split3_pair(u.x, u.y, hi.x, mid.x, lo.x);
    dn_split3_pair(u.z, u.w, hi.y, mid.y, lo.y);
    dn_split3_pair(v.x, v.y, hi.z, mid.z, lo.z);
    dn_split3_pair(v.z, v.w, hi.w, mid.w, lo.w);
}

// first physical k of slot group h (0: slots 0-3, 1: slots 4-7) of lane group lg (0..3) in k32 step s
__device__ __forceinline__ int rd_k0(int s, int lg, int h) { return 32 * s + 16 * h + 4 * lg; }

// Split the 128 x 128 B operand of `mesh` into fragment-ordered bf16 planes (all threads of the workgroup).
template <bool BCOLK>
__device__ __forceinline__ void rd_stage_b(const RgArgs& g, int mesh, int n0, unsigned char* sB, int tid) {
    const float* bp = g.b[0][0] + (long long)mesh * g.b_mesh_stride;
#pragma unroll
    for (int it = 0; it < (4 * 8 * 64) / DN_RD_THREADS; ++it) {
        const int item = tid + it * DN_RD_THREADS;
        const int lane = item & 63, st = item >> 6, s = st >> 3, t = st & 7;
        const int lg = lane >> 4;
        const int n = n0 + 16 * t + (lane & 15);               // N % 128 == 0: always a valid column
        float4 u, v;
        if (BCOLK) {
            u = *reinterpret_cast<const float4*>(bp + (long long)n * g.ldb + rd_k0(s, lg, 0));
            v = *reinterpret_cast<const float4*>(bp + (long long)n * g.ldb + rd_k0(s, lg, 1));
        } else {
            const long long ld = g.ldb;
            const float* c0 = bp + rd_k0(s, lg, 0) * ld + n;
            const float* c1 = bp + rd_k0(s, lg, 1) * ld + n;
            u = make_float4(c0[0], c0[ld], c0[2 * ld], c0[3 * ld]);
            v = make_float4(c1[0], c1[ld], c1[2 * ld], c1[3 * ld]);
        }
        uint4 hi, mid, lo;
        rd_split8(u, v, hi, mid, lo);
        unsigned char* dst = sB + ((st * 3) * 64 + lane) * 16;
        *reinterpret_cast<uint4*>(dst) = hi;
        *reinterpret_cast<uint4*>(dst + 1024) = mid;
        *reinterpret_cast<uint4*>(dst + 2048) = lo;
    }
}

struct RdUnit { int row0, nrows; };   // up to 16 consecutive rows: one wave's unit of work

// Unit j of a run of contiguous rows [rs, re): pure arithmetic -- no table lookups (= no loads, no waits) in the main loop.
// Past the run's end the last unit is returned (callers use that as a harmless prefetch target).
__device__ __forceinline__ RdUnit rd_unit(int rs, int re, int j) {
    const int nu = (re - rs + DN_RD_ROWS - 1) / DN_RD_ROWS;
    j = j < nu ? j : nu - 1;
    RdUnit r;
    r.row0 = rs + DN_RD_ROWS * j;
    r.nrows = re - r.row0 < DN_RD_ROWS ? re - r.row0 : DN_RD_ROWS;
    return r;
}
// this lane's row of the unit (clamped to the unit's first row past its end: feeds outputs that are never stored)
__device__ __forceinline__ const float* rd_row_ptr(const float* ap, int ald, const RdUnit& un, int li, int lg) {
    return ap + (long long)(un.row0 + (li < un.nrows ? li : 0)) * ald + 4 * lg;
}

// B fragments of plane p of group G = 4 s + pr of a unit (k32 step s, column tiles 2 pr and 2 pr + 1): two ds_read_b128
__device__ __forceinline__ void rd_read_plane(const unsigned char* sB, int lane, int G, int p, uint4 (&F)[3][2]) {
#if defined(DN_RD_ABL_LDSR)   // development ablation: the fragments primed before the loop are multiplied again and again
    return;
#endif
    const int s = (G & 15) >> 2, pr = G & 3;
#pragma unroll
    for (int e = 0; e < 2; ++e)
        F[p][e] = *reinterpret_cast<const uint4*>(sB + (((s * 8 + 2 * pr + e) * 3 + p) * 64 + lane) * 16);
}

__device__ __forceinline__ f32x4 rd_mma(const uint4& b, const uint4& a, const f32x4& c) {   // operands swapped: D[n][row], see the header
#if defined(DN_RD_ABL_MFMA)   // development ablation: operands stay live, no matrix work
    f32x4 r = c;
    r[0] += __uint_as_float((b.x & a.x) & 0x3f800000u);
    return r;
#else
    return dn_mfma_bf16_16(b, a, c);
#endif
}

// One unit (16 rows x 128 columns, 16 groups of 12 MFMAs).  Software pipeline, all of it carried ACROSS units:
//   * B fragments: a ring of two groups (48 registers).  A plane's registers are refilled with the same plane of group G+2
//     (wrapping into the next unit) right after its LAST use in group G: the product order (hi*lo | mid*mid, hi*mid | lo*hi,
//     mid*hi, hi*hi) retires B's lo plane after 2 MFMAs, mid after 6, hi after 12, and needs them in that order again, so every
//     read has ~18 MFMAs (288 cycles) to land.  (Read-then-multiply per half step: the older wave of a SIMD ran at read latency +
//     MFMA time, 29 cycles per MFMA, the younger one starved; one whole group ahead: 25 cycles -- timelines in profiles/.)
//   * A planes: `a` holds the split of the current k32 step; the next step's is computed next to the MFMAs of tile pair 2, the
//     first step of the NEXT unit (register set Y) next to the last step's;
//   * whoever splits a register pair reloads it with the same step of the unit two ahead of the pair's owner (np_x / np_y).
template <int MODE, bool FLAG>
__device__ __forceinline__ void rd_unit_body(const RgArgs& g, const unsigned char* sB, const RdUnit& cur, const float* np_x,
                                             const float* np_y, int n0, int lane, float4 (&X)[8], float4 (&Y)[8], uint4 (&a)[3],
                                             uint4 (&F)[2][3][2], int& trn) {
    const int li = lane & 15, lg = lane >> 4;
    (void)trn;
    RD_T();
    constexpr bool need_r0 = MODE == DN_EPI_BIAS_RESID || MODE == DN_EPI_MUL_DFAC || MODE == DN_EPI_ADD || MODE == DN_EPI_DTANH ||
                             MODE == DN_EPI_MASS_ADD;
    constexpr bool need_bias = (MODE == DN_EPI_STORE && FLAG) || MODE == DN_EPI_BIAS_RELU || MODE == DN_EPI_BIAS_RESID;
    // ---- auxiliary operands of the epilogue: requested first (see the header)
    const bool row_ok = li < cur.nrows;
    const long long grow = cur.row0 + (row_ok ? li : 0);
    const int col0 = n0 + 4 * lg;                              // this lane's columns: col0 + 16 t .. +3
    PtPiece P[8];
    const bool has_r0 = g.r0 != nullptr;                       // MASS_ADD's addend is optional
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const long long roff = grow * g.ldr + col0 + 16 * t;
        P[t].off = grow * g.ldo + col0 + 16 * t;
#if defined(DN_RD_ABL_NOSTORE)   // development ablation: nothing is written
        P[t].ok = row_ok && g.ldo < 0;
#else
        P[t].ok = row_ok;
#endif
        P[t].a0 = dn_f4_zero();
        if (need_r0 && (MODE != DN_EPI_MASS_ADD || has_r0)) P[t].a0 = *reinterpret_cast<const float4*>(g.r0 + roff);
        if (MODE == DN_EPI_BIAS_RELU && FLAG) {                // explicit mask or drawn bits (see pt_piece_load)
            const uint32_t ld = *reinterpret_cast<const uint32_t*>(g.mask ? g.mask + roff : reinterpret_cast<const uint8_t*>(g.bias));
            P[t].mk = g.mask ? ld : dn_keep_bytes(dn_keep_bits(g.rng_seed, grow, (col0 + 16 * t) >> 2, (g.N + 3) >> 2));
        }
        if (MODE == DN_EPI_MASS_ADD) P[t].rs = g.rowv[grow];
    }
    f32x4 acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[t][r] = 0.f;
    DN_SCHED_FENCE();
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        uint4 an[3];
#pragma unroll
        for (int pr = 0; pr < 4; ++pr) {
            const int G = 4 * s + pr;
            uint4 (&Fg)[3][2] = F[G & 1];
            // (A plane, B plane) per product; B's lo (2) is used by product 0 only, mid (1) by 1-2, hi (0) by 3-5
            constexpr int PA[6] = {0, 1, 0, 2, 1, 0}, PB[6] = {2, 1, 1, 0, 0, 0};
#define RD_MMA(p_)                                                                                                      \
    _Pragma("unroll") for (int e = 0; e < 2; ++e)                                                                       \
        acc[2 * pr + e] = rd_mma(Fg[PB[p_]][e], a[PA[p_]], acc[2 * pr + e]);
            RD_MMA(0)
            DN_SCHED_FENCE();
            rd_read_plane(sB, lane, G + 2, 2, Fg);
            if (pr == 2) {                                     // next step's planes, and the consumed registers' refill
                if (s < 3) {
                    rd_split8(X[2 * s + 2], X[2 * s + 3], an[0], an[1], an[2]);
#if !defined(DN_RD_ABL_NOLOAD)   // development ablation: the prologue's fragments are multiplied again and again
                    X[2 * s + 2] = *reinterpret_cast<const float4*>(np_x + 32 * (s + 1));
                    X[2 * s + 3] = *reinterpret_cast<const float4*>(np_x + 32 * (s + 1) + 16);
#endif
                } else {
                    rd_split8(Y[0], Y[1], an[0], an[1], an[2]);
#if !defined(DN_RD_ABL_NOLOAD)
                    Y[0] = *reinterpret_cast<const float4*>(np_y);
                    Y[1] = *reinterpret_cast<const float4*>(np_y + 16);
#endif
                }
            }
            RD_MMA(1)
            RD_MMA(2)
            DN_SCHED_FENCE();
            rd_read_plane(sB, lane, G + 2, 1, Fg);
            RD_MMA(3)
            RD_MMA(4)
            RD_MMA(5)
            DN_SCHED_FENCE();
            rd_read_plane(sB, lane, G + 2, 0, Fg);
#undef RD_MMA
        }
#pragma unroll
        for (int p = 0; p < 3; ++p) a[p] = an[p];
    }
    RD_T();
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        P[t].v = make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]);
        if (need_bias) P[t].bias = *reinterpret_cast<const float4*>(sB + DN_RD_LDS_B + (4 * lg + 16 * t) * 4);
        pt_piece_store<MODE, FLAG>(g, P[t]);
    }
    RD_T();
}

template <int MODE, bool BCOLK, bool FLAG>
__global__ __launch_bounds__(DN_RD_THREADS) DN_WAVES_PER_EU(2) void rowgemm_rd_kernel(RgArgs g, int ntiles) {
    DN_DYN_SMEM(smem_raw);
    unsigned char* sB = reinterpret_cast<unsigned char*>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = DN_UNIFORM(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    const int n0 = blockIdx.y * 128;
    const int G = gridDim.x;
    // workgroup b owns the contiguous tile range [t_beg, t_end)
    const int t_beg = (int)((long long)blockIdx.x * ntiles / G), t_end = (int)((long long)(blockIdx.x + 1) * ntiles / G);
    const float* ap = g.a[0].p;
    const int ald = g.a[0].ld;
    const bool shared_b = g.b_mesh_stride == 0;
    int trn = 0;
    RD_T();

    if (tid < 128) reinterpret_cast<float*>(sB + DN_RD_LDS_B)[tid] = g.bias ? g.bias[n0 + tid] : 0.f;   // visible after the first run's barriers
    int t0 = t_beg;
    while (t0 < t_end) {
        // A run: consecutive tiles whose rows are contiguous and that share one B (one mesh, or any mesh when B is a weight
        // matrix).  The descriptors are fetched eight at a time (ONE memory round trip per window, not one per tile).
        int mesh = 0, rs = 0, re = 0, t1 = t0;
        bool open = true;
        while (open && t1 < t_end) {
            DnTile d[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) d[i] = g.tiles[t1 + i < t_end ? t1 + i : t_end - 1];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (open && t1 < t_end) {
                    if (t1 == t0) { mesh = d[i].mesh; rs = d[i].row0; re = rs + d[i].nrows; ++t1; }
                    else if (d[i].row0 == re && (shared_b || d[i].mesh == mesh)) { re += d[i].nrows; ++t1; }
                    else open = false;
                }
            }
        }
        const int nu = (re - rs + DN_RD_ROWS - 1) / DN_RD_ROWS;   // 16-row units of the run; wave w takes units w, w + 8, ...
        // two register sets: the wave's i-th unit travels in set i & 1 and is fetched two units ahead
        int j = wave;
        float4 A0[8], A1[8];
        RdUnit c0 = rd_unit(rs, re, j), c1 = rd_unit(rs, re, j + DN_RD_WAVES);
        {   // both in flight under the B staging (a wave without units fetches the run's last unit: harmless)
            const float* p0 = rd_row_ptr(ap, ald, c0, li, lg);
            const float* p1 = rd_row_ptr(ap, ald, c1, li, lg);
#pragma unroll
            for (int i = 0; i < 8; ++i) A0[i] = *reinterpret_cast<const float4*>(p0 + 16 * i);
#pragma unroll
            for (int i = 0; i < 8; ++i) A1[i] = *reinterpret_cast<const float4*>(p1 + 16 * i);
        }
        RD_T();
        __syncthreads();                                       // nobody still reads the previous run's planes
        RD_T();
        rd_stage_b<BCOLK>(g, mesh, n0, sB, tid);
        __syncthreads();
        RD_T();
        if (j < nu) {                                          // prime the pipeline: planes of the first step, fragments of group 0
            uint4 a[3], F[2][3][2];
            rd_split8(A0[0], A0[1], a[0], a[1], a[2]);
            {
                const float* p2 = rd_row_ptr(ap, ald, rd_unit(rs, re, j + 2 * DN_RD_WAVES), li, lg);
                A0[0] = *reinterpret_cast<const float4*>(p2);
                A0[1] = *reinterpret_cast<const float4*>(p2 + 16);
            }
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                rd_read_plane(sB, lane, 0, p, F[0]);
                rd_read_plane(sB, lane, 1, p, F[1]);
            }
            for (; j < nu; j += 2 * DN_RD_WAVES) {
                {
                    const RdUnit n2 = rd_unit(rs, re, j + 2 * DN_RD_WAVES), n3 = rd_unit(rs, re, j + 3 * DN_RD_WAVES);
                    rd_unit_body<MODE, FLAG>(g, sB, c0, rd_row_ptr(ap, ald, n2, li, lg), rd_row_ptr(ap, ald, n3, li, lg), n0, lane, A0, A1,
                                             a, F, trn);
                    c0 = n2;
                }
                if (j + DN_RD_WAVES < nu) {                    // wave-uniform
                    const RdUnit n3 = rd_unit(rs, re, j + 3 * DN_RD_WAVES), n4 = rd_unit(rs, re, j + 4 * DN_RD_WAVES);
                    rd_unit_body<MODE, FLAG>(g, sB, c1, rd_row_ptr(ap, ald, n3, li, lg), rd_row_ptr(ap, ald, n4, li, lg), n0, lane, A1, A0,
                                             a, F, trn);
                    c1 = n3;
                }
            }
        }
        t0 = t1;
    }
}

template <int MODE, bool BCOLK, bool FLAG>
static int rd_launch(const RgArgs& g, int ntiles, hipStream_t stream) {
#ifndef DN_EMULATE
    static unsigned long long lds_opt_in = 0;   // per-device bitmap
    dn_lds_opt_in(reinterpret_cast<const void*>(&rowgemm_rd_kernel<MODE, BCOLK, FLAG>), DN_RD_LDS, &lds_opt_in);
#endif
    int gx = dn_num_cus();
    if (gx > ntiles) gx = ntiles;
    DN_LAUNCH((rowgemm_rd_kernel<MODE, BCOLK, FLAG>), dim3(gx, (g.N + 127) / 128, 1), dim3(DN_RD_THREADS, 1, 1), DN_RD_LDS, stream, g, ntiles);
    return (int)hipGetLastError();
}

// one output, one unscaled 128-wide A segment, whole 128-column output tiles, 16-byte aligned rows
static bool rd_eligible(const RgArgs& g, int nout) {
    if (!DN_RD || nout != 1 || !g.aligned || g.nseg != 1 || g.a[0].w != 128 || g.a[0].q || g.N < 128 || g.N % 128 != 0) return false;
    auto al = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
    if (g.ldo % 4 != 0 || g.ldr % 4 != 0 || !al(g.o0) || !al(g.r0) || ((uintptr_t)g.mask & 3) != 0) return false;
    switch (g.mode) {
        case DN_EPI_STORE: return true;
        case DN_EPI_BIAS_RELU: return g.bias != nullptr && g.b_colk;
        case DN_EPI_BIAS_RESID: return g.bias != nullptr && g.r0 != nullptr && g.b_colk;
        case DN_EPI_MUL_DFAC: case DN_EPI_ADD: case DN_EPI_DTANH: return g.r0 != nullptr && !g.b_colk;
        case DN_EPI_MASS_ADD: return !g.b_colk;
        default: return false;
    }
}

bool dn_rowgemm_try_direct(const RgArgs& g, int ntiles, int nout, hipStream_t stream, int* err) {
    if (!rd_eligible(g, nout)) return false;
    const bool ck = g.b_colk != 0;
    switch (g.mode) {
        case DN_EPI_STORE:
            if (g.bias) *err = ck ? rd_launch<DN_EPI_STORE, true, true>(g, ntiles, stream) : rd_launch<DN_EPI_STORE, false, true>(g, ntiles, stream);
            else *err = ck ? rd_launch<DN_EPI_STORE, true, false>(g, ntiles, stream) : rd_launch<DN_EPI_STORE, false, false>(g, ntiles, stream);
            break;
        case DN_EPI_BIAS_RELU:
            *err = (g.mask || g.rng_seed) ? rd_launch<DN_EPI_BIAS_RELU, true, true>(g, ntiles, stream) : rd_launch<DN_EPI_BIAS_RELU, true, false>(g, ntiles, stream);
            break;
        case DN_EPI_BIAS_RESID: *err = rd_launch<DN_EPI_BIAS_RESID, true, false>(g, ntiles, stream); break;
        case DN_EPI_MUL_DFAC: *err = rd_launch<DN_EPI_MUL_DFAC, false, false>(g, ntiles, stream); break;
        case DN_EPI_ADD: *err = rd_launch<DN_EPI_ADD, false, false>(g, ntiles, stream); break;
        case DN_EPI_DTANH: *err = rd_launch<DN_EPI_DTANH, false, false>(g, ntiles, stream); break;
        case DN_EPI_MASS_ADD: *err = rd_launch<DN_EPI_MASS_ADD, false, false>(g, ntiles, stream); break;
        default: return false;
    }
    return true;
}
