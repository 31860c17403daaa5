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
    const DnTileRO tiles_ro = DN_TILES_RO(g.tiles);
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
            for (int i = 0; i < 8; ++i) {
                const int ti = t1 + i < t_end ? t1 + i : t_end - 1;
                d[i].row0 = tiles_ro[ti].row0; d[i].nrows = tiles_ro[ti].nrows; d[i].mesh = tiles_ro[ti].mesh;
            }
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

// =====================================================================================================================
// 32-row form: ONE wave per SIMD (four waves, up to 512 registers each), v_mfma_f32_32x32x16_bf16, everything software-pipelined
// inside the wave.  Measured on the 16-row form above (two waves per SIMD, SQ counters in profiles/): the matrix pipe was busy 35 %
// of the time and the waves sat issue-stalled 54 % of theirs -- twice the instructions per flop (16-cycle MFMAs, one B-fragment
// read per two of them) contending for one SIMD's issue port.  Here a B fragment feeds 32-cycle MFMAs, and with a single wave
// per SIMD nothing competes for the port; what the second wave used to hide is hidden by distance instead:
//   * A: one register set (64), a pair is refilled with the same step of the NEXT unit right after it has been split;
//   * B fragments: ring of two groups, refilled plane by plane after a plane's last use (as above);
//   * epilogue: the finished unit stays in its accumulators (two sets, ping-pong) and is streamed out piece by piece (16 float4
//     pieces, one per MFMA group) under the NEXT unit's MFMAs; an auxiliary operand's register is refilled with the current
//     unit's piece as soon as the parked unit's piece has used it.
// Every global load is therefore a whole unit (~3 us) old when it is waited for, and never younger than a load that is still
// needed later (vmcnt is an in-order counter).
// =====================================================================================================================
#ifndef DN_RD32
#define DN_RD32 1
#endif
#define DN_RD32_LDS_B (8 * 4 * 3 * 1024)   // [k16 step][32-column tile][plane][lane] x 16 B
#define DN_RD32_LDS (DN_RD32_LDS_B + 512)

__device__ __forceinline__ int rd32_k0(int s, int lg, int h) { return 16 * s + 8 * h + 4 * lg; }

// B staging in two halves, so that the loads can be issued as early as the operand is known (for a weight matrix: first thing in the
// kernel, concurrently with the tile lookups) and waited for only when everything else of the prologue is in flight.
struct Rd32BRaw { float4 r[16]; };
// BCOLK (B = W[n][k], k contiguous): item = (k16 step, column tile, lane), its eight k are two float4 of row n.
// NN (B[k][n], n contiguous): a thread fetches an 8 (k) x 4 (n) block as eight float4 along n -- the eight k of FOUR items (four
// consecutive lanes) -- instead of 32 strided dwords; 16 loads and 16 address pairs per thread either way.
template <bool BCOLK>
__device__ __forceinline__ void rd32_load_b(const float* bp, int ldb, int n0, int tid, Rd32BRaw& R) {
    if (BCOLK) {
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int item = tid + it * 256;
            const int lane = item & 63, st = item >> 6, s = st >> 2, t = st & 3;
            const int lg = lane >> 5;
            const float* rp = bp + (long long)(n0 + 32 * t + (lane & 31)) * ldb;   // N % 128 == 0: always a valid column
            R.r[2 * it] = *reinterpret_cast<const float4*>(rp + rd32_k0(s, lg, 0));
            R.r[2 * it + 1] = *reinterpret_cast<const float4*>(rp + rd32_k0(s, lg, 1));
        }
    } else {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int b = tid + 256 * h;
            const int nq = b & 31, lg = (b >> 5) & 1, s = b >> 6;
            const float* cp = bp + n0 + 4 * nq;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                R.r[8 * h + j] = *reinterpret_cast<const float4*>(cp + (long long)(rd32_k0(s, lg, 0) + j) * ldb);
                R.r[8 * h + 4 + j] = *reinterpret_cast<const float4*>(cp + (long long)(rd32_k0(s, lg, 1) + j) * ldb);
            }
        }
    }
}
template <bool BCOLK>
__device__ __forceinline__ void rd32_put_b(const Rd32BRaw& R, unsigned char* sB, int tid) {
    if (BCOLK) {
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int item = tid + it * 256;
            const int lane = item & 63, st = item >> 6;
            uint4 hi, mid, lo;
            rd_split8(R.r[2 * it], R.r[2 * it + 1], hi, mid, lo);
            unsigned char* dst = sB + ((st * 3) * 64 + lane) * 16;
            *reinterpret_cast<uint4*>(dst) = hi;
            *reinterpret_cast<uint4*>(dst + 1024) = mid;
            *reinterpret_cast<uint4*>(dst + 2048) = lo;
        }
    } else {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int b = tid + 256 * h;
            const int nq = b & 31, lg = (b >> 5) & 1, s = b >> 6;
            const int t = nq >> 3;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float4 u = make_float4(dn_f4_get(R.r[8 * h], c), dn_f4_get(R.r[8 * h + 1], c), dn_f4_get(R.r[8 * h + 2], c), dn_f4_get(R.r[8 * h + 3], c));
                const float4 v = make_float4(dn_f4_get(R.r[8 * h + 4], c), dn_f4_get(R.r[8 * h + 5], c), dn_f4_get(R.r[8 * h + 6], c), dn_f4_get(R.r[8 * h + 7], c));
                uint4 hi, mid, lo;
                rd_split8(u, v, hi, mid, lo);
                const int lane = 32 * lg + ((4 * nq + c) & 31);
                unsigned char* dst = sB + (((s * 4 + t) * 3) * 64 + lane) * 16;
                *reinterpret_cast<uint4*>(dst) = hi;
                *reinterpret_cast<uint4*>(dst + 1024) = mid;
                *reinterpret_cast<uint4*>(dst + 2048) = lo;
            }
        }
    }
}

__device__ __forceinline__ RdUnit rd32_unit(int rs, int re, int j) {
    const int nu = (re - rs + 31) / 32;
    j = j < nu ? j : nu - 1;
    RdUnit r;
    r.row0 = rs + 32 * j;
    r.nrows = re - r.row0 < 32 ? re - r.row0 : 32;
    return r;
}
__device__ __forceinline__ const float* rd32_row_ptr(const float* ap, int ald, const RdUnit& un, int li, int lg) {
    return ap + (long long)(un.row0 + (li < un.nrows ? li : 0)) * ald + 4 * lg;
}
// plane p of the B fragments of group G = 2 s + pr (k16 step s, column tiles 2 pr and 2 pr + 1): two ds_read_b128
__device__ __forceinline__ void rd32_read_plane(const unsigned char* sB, int lane, int G, int p, uint4 (&F)[3][2]) {
    const int s = (G & 15) >> 1, pr = G & 1;
#pragma unroll
    for (int e = 0; e < 2; ++e)
        F[p][e] = *reinterpret_cast<const uint4*>(sB + (((s * 4 + 2 * pr + e) * 3 + p) * 64 + lane) * 16);
}

// auxiliary operands of piece i (column tile i >> 2, quarter i & 3) of the unit whose row this lane holds is `grow`
template <int MODE, bool FLAG>
__device__ __forceinline__ void rd32_aux_load(const RgArgs& g, long long grow, int col, float4& aux, uint32_t& mk) {
    constexpr bool need_r0 = MODE == DN_EPI_BIAS_RESID || MODE == DN_EPI_MUL_DFAC || MODE == DN_EPI_ADD || MODE == DN_EPI_DTANH ||
                             MODE == DN_EPI_MASS_ADD;
    const long long roff = grow * g.ldr + col;
    if (need_r0 && (MODE != DN_EPI_MASS_ADD || g.r0 != nullptr)) aux = *reinterpret_cast<const float4*>(g.r0 + roff);
    if (MODE == DN_EPI_BIAS_RELU && FLAG && g.mask) mk = *reinterpret_cast<const uint32_t*>(g.mask + roff);
}
// epilogue of one float4 piece straight from the accumulators
template <int MODE, bool FLAG>
__device__ __forceinline__ void rd32_piece_out(const RgArgs& g, const unsigned char* sB, const f32x16& accT, int q, long long grow, bool ok,
                                               int col, const float4& aux, uint32_t mk, float rs) {
    constexpr bool need_bias = (MODE == DN_EPI_STORE && FLAG) || MODE == DN_EPI_BIAS_RELU || MODE == DN_EPI_BIAS_RESID;
    PtPiece P;
    P.v = make_float4(accT[4 * q], accT[4 * q + 1], accT[4 * q + 2], accT[4 * q + 3]);
    P.a0 = aux;
    P.bias = dn_f4_zero();
    if (need_bias) P.bias = *reinterpret_cast<const float4*>(sB + DN_RD32_LDS_B + (col & 127) * 4);
    P.mk = mk;
    if (MODE == DN_EPI_BIAS_RELU && FLAG && !g.mask) P.mk = dn_keep_bytes(dn_keep_bits(g.rng_seed, grow, col >> 2, (g.N + 3) >> 2));
    P.rs = rs;
    P.off = grow * g.ldo + col;
#if defined(DN_RD_ABL_NOSTORE)
    P.ok = ok && g.ldo < 0;
#else
    P.ok = ok;
#endif
    pt_piece_store<MODE, FLAG>(g, P);
}

__device__ __forceinline__ f32x16 rd32_mma(const uint4& b, const uint4& a, const f32x16& c) {   // operands swapped: D[n][row]
    return dn_mfma_bf16(b, a, c);
}

template <int MODE, bool FLAG>
__device__ __forceinline__ void rd32_unit_body(const RgArgs& g, const unsigned char* sB, int n0, int lane, const RdUnit& prev,
                                               const RdUnit& cur, const float* np1, const float* np2, float4 (&A)[16],
                                               float4 (&AUX)[8], uint32_t (&MK)[8], float& rs, uint4 (&a)[3], uint4 (&F)[2][3][2],
                                               f32x16 (&acc)[4], const f32x16 (&accP)[4], int& trn) {
    const int li = lane & 31, lg = lane >> 5;
    (void)trn;
    RD_T();
    const bool ok_prev = li < prev.nrows, ok_cur = li < cur.nrows;
    const long long grow_prev = prev.row0 + (ok_prev ? li : 0), grow_cur = cur.row0 + (ok_cur ? li : 0);
    float rs_new = 0.f;
    if (MODE == DN_EPI_MASS_ADD) rs_new = g.rowv[grow_cur];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    DN_SCHED_FENCE();
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        uint4 an[3];
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
            const int G = 2 * s + pr;
            uint4 (&Fg)[3][2] = F[pr];
            // (A plane, B plane) per product; B's lo (2) is used by product 0 only, mid (1) by 1-2, hi (0) by 3-5
            constexpr int PA[6] = {0, 1, 0, 2, 1, 0}, PB[6] = {2, 1, 1, 0, 0, 0};
#define RD_MMA(p_)                                                                                                      \
    _Pragma("unroll") for (int e = 0; e < 2; ++e)                                                                       \
        acc[2 * pr + e] = rd32_mma(Fg[PB[p_]][e], a[PA[p_]], acc[2 * pr + e]);
            RD_MMA(0)
            DN_SCHED_FENCE();
            rd32_read_plane(sB, lane, G + 2, 2, Fg);
            if (pr == 0) {                                     // next step's planes, and the consumed registers' refill
                if (s < 7) {
                    rd_split8(A[2 * s + 2], A[2 * s + 3], an[0], an[1], an[2]);
                    A[2 * s + 2] = *reinterpret_cast<const float4*>(np1 + 8 * (2 * s + 2));
                    A[2 * s + 3] = *reinterpret_cast<const float4*>(np1 + 8 * (2 * s + 3));
                } else {                                       // first step of the next unit (loaded during the previous unit)
                    rd_split8(A[0], A[1], an[0], an[1], an[2]);
                    A[0] = *reinterpret_cast<const float4*>(np2);
                    A[1] = *reinterpret_cast<const float4*>(np2 + 8);
                }
            }
            {   // piece G of the parked unit goes out; its auxiliary register is refilled for the piece eight groups (~1.5 us) ahead
                const int col = n0 + 32 * (G >> 2) + 8 * (G & 3) + 4 * lg;
                rd32_piece_out<MODE, FLAG>(g, sB, accP[G >> 2], G & 3, grow_prev, ok_prev, col, AUX[G & 7], MK[G & 7], rs);
                const int Gn = (G + 8) & 15;                   // pieces 8-15 of the parked unit, then 0-7 of the current one
                const int coln = n0 + 32 * (Gn >> 2) + 8 * (Gn & 3) + 4 * lg;
                rd32_aux_load<MODE, FLAG>(g, G < 8 ? grow_prev : grow_cur, coln, AUX[G & 7], MK[G & 7]);
            }
            RD_MMA(1)
            RD_MMA(2)
            DN_SCHED_FENCE();
            rd32_read_plane(sB, lane, G + 2, 1, Fg);
            RD_MMA(3)
            RD_MMA(4)
            RD_MMA(5)
            DN_SCHED_FENCE();
            rd32_read_plane(sB, lane, G + 2, 0, Fg);
#undef RD_MMA
        }
#pragma unroll
        for (int p = 0; p < 3; ++p) a[p] = an[p];
    }
    rs = rs_new;
    RD_T();
}

// stream out the last unit of a run (nothing left to overlap it with)
template <int MODE, bool FLAG>
__device__ __forceinline__ void rd32_flush(const RgArgs& g, const unsigned char* sB, int n0, int lane, const RdUnit& prev,
                                           const float4 (&AUX)[8], const uint32_t (&MK)[8], float rs, const f32x16 (&accP)[4]) {
    const int li = lane & 31, lg = lane >> 5;
    const bool ok_prev = li < prev.nrows;
    const long long grow_prev = prev.row0 + (ok_prev ? li : 0);
    float4 AX[8];                                              // pieces 8-15: not requested yet
    uint32_t MX[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        AX[i] = dn_f4_zero(); MX[i] = 0u;
        rd32_aux_load<MODE, FLAG>(g, grow_prev, n0 + 32 * ((i + 8) >> 2) + 8 * (i & 3) + 4 * lg, AX[i], MX[i]);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int col = n0 + 32 * (i >> 2) + 8 * (i & 3) + 4 * lg;
        rd32_piece_out<MODE, FLAG>(g, sB, accP[i >> 2], i & 3, grow_prev, ok_prev, col, i < 8 ? AUX[i] : AX[i - 8], i < 8 ? MK[i] : MX[i - 8], rs);
    }
}

template <int MODE, bool BCOLK, bool FLAG>
__global__ __launch_bounds__(256) DN_WAVES_PER_EU(1) void rowgemm_rd32_kernel(
    // the arguments the streams need to start: <= 16 dwords, delivered in SGPRs at wave launch (kernel-argument preload, built with
    // -mllvm -amdgpu-kernarg-preload-count=16); everything else of RgArgs is first needed microseconds later
    const float* ap, int ald, const float* b_base, int ldb_, long long b_mesh_stride_, int V, int G, int n_mesh_k_, RgArgs g) {
    DN_DYN_SMEM(smem_raw);
    unsigned char* sB = reinterpret_cast<unsigned char*>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = DN_UNIFORM(tid >> 6);
    const int li = lane & 31, lg = lane >> 5;
    // every kernel argument the launch will ever read, requested in ONE round trip (see DN_RESIDENT)
    i32x16 mo0, mo1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { mo0[i] = g.mesh_off[i]; mo1[i] = g.mesh_off[16 + i]; }
    const int mo32 = g.mesh_off[32];
    DN_RESIDENT(g.N, g.o0, g.ldo, g.bias, g.r0, g.ldr, g.rowv, g.mask, g.rng_seed, g.scale, mo0, mo1, mo32);
    const int n0 = blockIdx.y * 128;
    // Everything the prologue needs is a kernel argument: the workgroup's rows are arithmetic on v_total, the mesh boundaries
    // (per-mesh B only) come from g.mesh_off.  No table lookup stands between the launch and the first loads: a dependent
    // scalar lookup chain cost ~3 us of a 40 us launch (timeline in profiles/).
    const int U = (V + 31) >> 5;                               // 32-row slots; workgroup b owns slots [ub, ue)
    const int ub = (int)((long long)blockIdx.x * U / G), ue = (int)((long long)(blockIdx.x + 1) * U / G);
    const int R0 = 32 * ub, R1 = 32 * ue < V ? 32 * ue : V;
    const bool shared_b = b_mesh_stride_ == 0;
    int trn = 0;
    RD_T();
    if (R0 >= R1) return;

    // mesh boundaries from the SGPR copies (static indices only: a dynamic index would be another kernel-argument round trip)
    auto mesh_off_at = [&](int idx) {
        int v = mo32;
#pragma unroll
        for (int i = 0; i < 16; ++i) { v = idx == i ? mo0[i] : v; v = idx == 16 + i ? mo1[i] : v; }
        return v;
    };
    int m = 0;                                                 // mesh of row R0
    if (!shared_b) {
#pragma unroll
        for (int i = 1; i < 16; ++i) m += (i < n_mesh_k_ && mo0[i] <= R0) ? 1 : 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) m += (16 + i < n_mesh_k_ && mo1[i] <= R0) ? 1 : 0;
    }
    float bias_v = 0.f;
    bool first_run = true;
    int cur_row = R0;
    while (cur_row < R1) {
        // a run: the workgroup's rows inside one mesh (all of its rows when B is a weight matrix)
        const int mesh = m, rs_ = cur_row;
        int re = R1;
        if (!shared_b) { const int me = mesh_off_at(m + 1); re = me < R1 ? me : R1; }
        Rd32BRaw BR;
        rd32_load_b<BCOLK>(b_base + (long long)(shared_b ? 0 : mesh) * b_mesh_stride_, ldb_, n0, tid, BR);   // first thing in flight
        const int nu = (re - rs_ + 31) / 32;                    // 32-row units of the run; wave w takes units w, w + 4, ...
        int j = wave;
        float4 A[16];
        {
            const float* p0 = rd32_row_ptr(ap, ald, rd32_unit(rs_, re, j), li, lg);   // in flight under the B staging
#pragma unroll
            for (int i = 0; i < 16; ++i) A[i] = *reinterpret_cast<const float4*>(p0 + 8 * i);
        }
        if (first_run && tid < 128 && g.bias) bias_v = g.bias[n0 + tid];   // after the streams' first requests
        RD_T();
        __syncthreads();                                       // nobody still reads the previous run's planes
        RD_T();
        rd32_put_b<BCOLK>(BR, sB, tid);
        if (first_run && tid < 128) reinterpret_cast<float*>(sB + DN_RD32_LDS_B)[tid] = bias_v;
        first_run = false;
        __syncthreads();
        RD_T();
        if (j < nu) {
            uint4 a[3], F[2][3][2];
            float4 AUX[8];
            uint32_t MK[8];
            float rs = 0.f;
            f32x16 acc0[4], acc1[4];
#pragma unroll
            for (int i = 0; i < 8; ++i) { AUX[i] = dn_f4_zero(); MK[i] = 0u; }
            rd_split8(A[0], A[1], a[0], a[1], a[2]);
            {
                const float* p1 = rd32_row_ptr(ap, ald, rd32_unit(rs_, re, j + 4), li, lg);
                A[0] = *reinterpret_cast<const float4*>(p1);
                A[1] = *reinterpret_cast<const float4*>(p1 + 8);
            }
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                rd32_read_plane(sB, lane, 0, p, F[0]);
                rd32_read_plane(sB, lane, 1, p, F[1]);
            }
            RdUnit prev;
            prev.row0 = rs_; prev.nrows = 0;                    // nothing parked yet
            bool last_in_1 = false;
            for (; j < nu; j += 8) {
                {
                    const RdUnit cur = rd32_unit(rs_, re, j);
                    rd32_unit_body<MODE, FLAG>(g, sB, n0, lane, prev, cur, rd32_row_ptr(ap, ald, rd32_unit(rs_, re, j + 4), li, lg),
                                               rd32_row_ptr(ap, ald, rd32_unit(rs_, re, j + 8), li, lg), A, AUX, MK, rs, a, F, acc0, acc1, trn);
                    prev = cur;
                    last_in_1 = false;
                }
                if (j + 4 < nu) {                              // wave-uniform
                    const RdUnit cur = rd32_unit(rs_, re, j + 4);
                    rd32_unit_body<MODE, FLAG>(g, sB, n0, lane, prev, cur, rd32_row_ptr(ap, ald, rd32_unit(rs_, re, j + 8), li, lg),
                                               rd32_row_ptr(ap, ald, rd32_unit(rs_, re, j + 12), li, lg), A, AUX, MK, rs, a, F, acc1, acc0, trn);
                    prev = cur;
                    last_in_1 = true;
                }
            }
            if (last_in_1) rd32_flush<MODE, FLAG>(g, sB, n0, lane, prev, AUX, MK, rs, acc1);
            else rd32_flush<MODE, FLAG>(g, sB, n0, lane, prev, AUX, MK, rs, acc0);
            RD_T();
        }
        cur_row = re;
        ++m;
    }
}

template <int MODE, bool BCOLK, bool FLAG>
static int rd_launch(const RgArgs& g, int ntiles, hipStream_t stream) {
#ifndef DN_EMULATE
    static unsigned long long lds_opt_in = 0;   // per-device bitmap
#if DN_RD32
    dn_lds_opt_in(reinterpret_cast<const void*>(&rowgemm_rd32_kernel<MODE, BCOLK, FLAG>), DN_RD32_LDS, &lds_opt_in);
#else
    dn_lds_opt_in(reinterpret_cast<const void*>(&rowgemm_rd_kernel<MODE, BCOLK, FLAG>), DN_RD_LDS, &lds_opt_in);
#endif
#endif
    int gx = dn_num_cus();
#if DN_RD32
    {
        const int slots = (g.acct_rows + 127) / 128;           // at least 128 rows per workgroup
        if (gx > slots) gx = slots < 1 ? 1 : slots;
    }
    DN_LAUNCH((rowgemm_rd32_kernel<MODE, BCOLK, FLAG>), dim3(gx, (g.N + 127) / 128, 1), dim3(256, 1, 1), DN_RD32_LDS, stream, g.a[0].p, g.a[0].ld,
              g.b[0][0], g.ldb, g.b_mesh_stride, g.acct_rows, gx, g.n_mesh_k, g);
#else
    if (gx > ntiles) gx = ntiles;
    DN_LAUNCH((rowgemm_rd_kernel<MODE, BCOLK, FLAG>), dim3(gx, (g.N + 127) / 128, 1), dim3(DN_RD_THREADS, 1, 1), DN_RD_LDS, stream, g, ntiles);
#endif
    return (int)hipGetLastError();
}

// one output, one unscaled 128-wide A segment, whole 128-column output tiles, 16-byte aligned rows
#if defined(DN_DEV_ENV)   // development build only: DN_RD_MODES=<bit mask of epilogue modes that may take the direct kernel>
#include <stdlib.h>
static int rd_dev_mask() { static int m = -2; if (m == -2) { const char* e = getenv("DN_RD_MODES"); m = e ? atoi(e) : -1; } return m; }
#endif
static bool rd_eligible(const RgArgs& g, int nout) {
#if defined(DN_DEV_ENV)
    if (rd_dev_mask() >= 0 && !((rd_dev_mask() >> g.mode) & 1)) return false;
#endif
    if (!DN_RD || nout != 1 || !g.aligned || g.nseg != 1 || g.a[0].w != 128 || g.a[0].q || g.N < 128 || g.N % 128 != 0) return false;
#if DN_RD32
    if (g.acct_rows <= 0 || (g.b_mesh_stride != 0 && g.n_mesh_k <= 0)) return false;   // per-mesh B needs the mesh boundaries as kernel arguments
#endif
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
