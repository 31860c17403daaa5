// dn_backproject_wide.hip -- the forward back-projection x_diffuse = Phi ys (layers.py:66 -> geometry.py:590-598, from_basis) at
// K = C = 256, BASELINE config 4's shape, as a streaming launch of its own on the 3-term split-bf16 engine (its output is what the CSR
// gradient operators difference: the engine of every forward back-projection, DESIGN.md "Engines").
//
// Why a kernel of its own: at K = C = 128 the spectrum of a mesh is split once into LDS (96 KiB of planes, backproject_kernel in
// dn_diffuse.hip); at 256 x 256 the planes are 384 KiB, and the wave-specialised row GEMM that took the shape instead walks two 128-column
// tiles per row unit (Phi read twice, both operands split by its loader waves for every tile): 221 us for 0.41 GB on one 200k-vertex mesh.
// Here the spectrum is split ONCE per mesh and call (spec_pieces3_kernel: K / 32 "pieces" of [plane][16-column tile][lane] uint4, 48 KiB each,
// the layout of the chained kernels' weight pieces with three bf16 planes) and streamed through a three-slot LDS-DMA ring by one 8-wave
// workgroup per CU; every wave owns 16 rows of a 128-row tile of the batch (dn_mesh_batch_t.tiles: a tile never crosses a mesh), holds their
// 16 x 256 outputs in 64 accumulators, reads its rows of Phi straight from memory (two float4 per lane and k32 step, requested a whole
// pass ahead: the registers of a step are refilled for the next tile right after they are split) and multiplies against the piece in
// the ring: 6 MFMAs (hi*lo, lo*hi, mid*mid, hi*mid, mid*hi, hi*hi -- smallest terms first) per 16-column tile and step.
// Bounds (one 200k-vertex mesh): HBM 0.41 GB (~80 us at the practical rate); matrix pipe 12.5 k row groups x 768 MFMAs x 16 cycles over
// 1024 SIMDs = ~75 us at 2.1 GHz, in 7 rounds of 8 waves per CU where 6.1 would do (a tile is the unit the ring synchronises).
// Measured (profiles/r06_bw256.txt): 160 us against the row GEMM's 221 -- half the matrix peak, the level of every 3-term kernel here.
// (Streaming stores of the result measured level: diffusion 391-408 vs 392-401 us, block inference 1411-1416 vs 1393-1415 us -- plain stores kept.)
#include "dn_chain_tiles.h"
#include "dn_direct_tiles.h"

#ifndef DN_BW_G
#define DN_BW_G 1        // 16-row groups per wave.  2 (four waves per workgroup, one per SIMD, accumulators in AGPRs, every spectrum fragment read from LDS
#endif                   // feeds two row groups: half the LDS traffic) measured the same on MI355X (458-472 vs 459-469 us for the whole operator): not LDS-bound

#define DN_BW_RING 3
#define DN_BW_MAXP 128   // tiles per workgroup (their descriptors live in LDS; the launcher checks)

struct BwArgs {
    const DnTile* tiles; int n_tiles;
    const float* evecs; const uint4* ysp; float* out; float* out_amax;
};

// ys [n_mesh][K][C] (row-major) -> pieces [n_mesh][K / 32][3 planes][C / 16][64 lanes] uint4: lane (n, q) of tile nt holds the eight k
// 32 T + 4 q .. + 3, 32 T + 16 + 4 q .. + 3 of column 16 nt + n (the k order of the row fragments read by the kernel below).  grid (K / 32, n_mesh)
__global__ __launch_bounds__(1024) void spec_pieces3_kernel(const float* ys, int K, int C, uint4* out) {
    const int T = blockIdx.x, mesh = blockIdx.y, tid = threadIdx.x;
    const int NT = C / 16, KE = K / 32;
    const float* y = ys + (size_t)mesh * K * C;
    uint4* o = out + ((size_t)mesh * KE + T) * (3 * NT * 64);
    for (int e = tid; e < NT * 64; e += 1024) {
        const int nt = e >> 6, lane = e & 63;
        const int n = 16 * nt + (lane & 15), q = lane >> 4;
        const float* src = y + (long long)(32 * T + 4 * q) * C + n;
        const float4 u = make_float4(src[0], src[C], src[2 * (long long)C], src[3 * (long long)C]);
        const float* src2 = src + 16 * (long long)C;
        const float4 v = make_float4(src2[0], src2[C], src2[2 * (long long)C], src2[3 * (long long)C]);
        uint4 pl[3];
        rd_split8<3>(u, v, pl, 1.f);
#pragma unroll
        for (int p = 0; p < 3; ++p) o[p * NT * 64 + e] = pl[p];
    }
}

template <int K, int C, int G>
__global__ __launch_bounds__(64 * (8 / G)) DN_WAVES_PER_EU(2 / G) void backproject_wide_kernel(BwArgs a) {
    constexpr int NT = C / 16, KE = K / 32, NP = 3;
    constexpr int PIECE = NP * NT * 64;                 // uint4 per piece
    constexpr int NWAVES = 8 / G;                       // 128-row tiles: G 16-row groups per wave
    constexpr int NTHR = 64 * NWAVES;
    constexpr int LPT = PIECE / NTHR;                   // DMA requests per thread and piece
    constexpr int RING = DN_BW_RING;
    static_assert(PIECE % NTHR == 0 && NT % 2 == 0 && (G == 1 || G == 2), "piece staging");
    static_assert((RING - 2) * LPT + 4 * G + G * NT < 64, "counted waits: vmcnt has six bits");
    DN_DYN_SMEM(smem_raw);
    uint4* ring = reinterpret_cast<uint4*>(smem_raw);
    int4* pinfo = reinterpret_cast<int4*>(ring + RING * PIECE);                 // [DN_BW_MAXP] {first row, rows, mesh, -}
#ifdef DN_EMULATE
    const unsigned lds0 = 0;
#else
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem_raw;
#endif
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = lane & 15, q = lane >> 4;

    // tiles of this workgroup: XCD-contiguous ranges, as in the chained kernels
    const int units = a.n_tiles;
    const int GX = gridDim.x >> 3;
    const int per_x = (units + 7) >> 3;
    const int xcd = blockIdx.x & 7, slot0 = blockIdx.x >> 3;
    int npass = 0;
    if (slot0 < per_x) {
        int hi_local = units - xcd * per_x;
        hi_local = hi_local > per_x ? per_x : hi_local;
        if (slot0 < hi_local) npass = (hi_local - slot0 + GX - 1) / GX;
    }
    if (npass == 0) return;
    auto unit_of = [&](int pass_) { return xcd * per_x + slot0 + pass_ * GX; };
    for (int pp = tid; pp < npass; pp += NTHR) {
        const DnTile tl = a.tiles[unit_of(pp)];
        pinfo[pp] = int4{tl.row0, tl.nrows, tl.mesh, 0};
    }
    const DnTile t0 = a.tiles[unit_of(0)];
    int imesh = ch_uniform_i(t0.mesh);                  // mesh of the pass whose pieces are being requested
    int mesh_nx = imesh;

    // ---- the piece stream: KE pieces per pass, requested RING - 1 ahead by every wave
    int sq = 0, rq = 0, gp = 0;
#ifdef DN_EMULATE
    const int wave_u = wave;
#else
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
#endif
    auto issue = [&]() {
        const uint4* src_piece = a.ysp + ((size_t)imesh * KE + sq) * PIECE;
#pragma unroll
        for (int i = 0; i < LPT; ++i) {
            const int e0 = rq * PIECE + i * NTHR + wave_u * 64;
            ch_dma16(src_piece + i * NTHR + tid, ring + e0, lds0 + 16u * (unsigned)e0);
        }
        if (sq + 1 == KE) { sq = 0; imesh = mesh_nx; } else ++sq;
        rq = rq + 1 == RING ? 0 : rq + 1;
    };
#ifdef DN_EMULATE
#define BW_WAIT(n) do {} while (0)
#define BW_BARRIER() __syncthreads()
#else
#define BW_WAIT(n) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(n) : "memory")
#define BW_BARRIER() do { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)
#endif
    // this lane's row of a tile (past the tile's end: its first row -- feeds outputs that are never stored)
    auto row_ptr = [&](int row0, int nrows, int g) {
        const int li = 16 * (G * wave + g) + m;
        return a.evecs + (long long)(row0 + (li < nrows ? li : 0)) * K + 4 * q;
    };
    float4 af[G][KE][2];                                // the wave's rows of Phi: step T = floats 32 T + 4 q .. + 3 and 32 T + 16 + 4 q .. + 3 of the lane's row
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const float* ap = row_ptr(t0.row0, t0.nrows, g);
#pragma unroll
        for (int T = 0; T < KE; ++T) {
            af[g][T][0] = *reinterpret_cast<const float4*>(ap + 32 * T);
            af[g][T][1] = *reinterpret_cast<const float4*>(ap + 32 * T + 16);
        }
    }
#pragma unroll
    for (int i = 0; i < RING - 1; ++i) issue();
    // a compiler-visible full wait (not the inline-asm counted one): the waits the compiler makes for the row registers inside the loop are
    // the minimum over the loop's entries, and entering with the prologue's loads pending gave vmcnt(14) at the first step of every tile --
    // a drain of the stores just issued and of the ring requests.  Once per workgroup: the first piece is needed here anyway.
#ifndef DN_EMULATE
    __builtin_amdgcn_s_waitcnt(0x0070);                 // vmcnt(0) lgkmcnt(0)
#endif
    BW_BARRIER();                                       // (publishes the pass table as well)
    float wmax = 0.f;
    int stored = 0;                                     // store instructions this wave issued at the end of the previous tile (wave-uniform: a group without rows issues none)
    for (int pass = 0; pass < npass; ++pass) {
        const int4 pi_ = pinfo[pass];
        const int4 pn_ = pinfo[pass + 1 < npass ? pass + 1 : pass];
        const int row0 = ch_uniform_i(pi_.x), nrows = ch_uniform_i(pi_.y);
        mesh_nx = ch_uniform_i(pn_.z);
        const float* ap_nx[G];
#pragma unroll
        for (int g = 0; g < G; ++g) ap_nx[g] = row_ptr(ch_uniform_i(pn_.x), ch_uniform_i(pn_.y), g);
        dn_f32x4 acc[G][NT];
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[g][nt] = dn_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int T = 0; T < KE; ++T) {
            const uint4* ws_ = ring + (gp % RING) * PIECE;
            issue();
            uint4 pl[G][NP];
#pragma unroll
            for (int g = 0; g < G; ++g) {
                rd_split8<NP>(af[g][T][0], af[g][T][1], pl[g], 1.f);
                // the registers just split take the same step of the next tile -- unconditionally (after the last tile: its own rows again, unused):
                // behind a branch the compiler's own waits for these registers count the path without the loads and drain the ring requests
                // just issued (vmcnt(14 - 2 T) in the listing: a full request latency at every step of the second half of a tile)
                af[g][T][0] = *reinterpret_cast<const float4*>(ap_nx[g] + 32 * T);
                af[g][T][1] = *reinterpret_cast<const float4*>(ap_nx[g] + 32 * T + 16);
            }
            uint4 wq[2][2][NP];                          // spectrum fragments of two column tiles, read one pair ahead of their MFMAs
#define BW_WLOAD(W_, np_) do { _Pragma("unroll") for (int e_ = 0; e_ < 2; ++e_) _Pragma("unroll") for (int p_ = 0; p_ < NP; ++p_) \
                                   W_[e_][p_] = ws_[(p_ * NT + (np_) + e_) * 64 + lane]; } while (0)
            // one product term for the two column tiles and the wave's row groups: 2 G independent accumulators between two MFMAs into the same one
#define BW_TERM(PW, PA) do { _Pragma("unroll") for (int g_ = 0; g_ < G; ++g_) {                                                  \
                                 acc[g_][np] = dn_mfma_bf16_16(wq[cb][0][PW], pl[g_][PA], acc[g_][np]);                            \
                                 acc[g_][np + 1] = dn_mfma_bf16_16(wq[cb][1][PW], pl[g_][PA], acc[g_][np + 1]); } } while (0)
            BW_WLOAD(wq[0], 0);
#pragma unroll
            for (int np = 0; np < NT; np += 2) {
                const int cb = (np >> 1) & 1;
                if (np + 2 < NT) BW_WLOAD(wq[cb ^ 1], np + 2);
                BW_TERM(0, 2); BW_TERM(2, 0); BW_TERM(1, 1); BW_TERM(0, 1); BW_TERM(1, 0); BW_TERM(0, 0);      // smallest terms first
            }
#undef BW_TERM
#undef BW_WLOAD
            // the piece waited for is DMA(gp + 1); younger requests that may stay in flight: the ring's youngest piece and the last two row refills
            // (this step's, and the one issued behind DMA(gp + 1) a step ago); at the first step of a later tile also the previous tile's stores
            if (T == 0 && stored == G * NT) BW_WAIT((RING - 2) * LPT + 4 * G + G * NT);
            else if (G > 1 && T == 0 && stored == NT) BW_WAIT((RING - 2) * LPT + 4 * G + NT);
            else BW_WAIT((RING - 2) * LPT + 4 * G);
            BW_BARRIER();
            ++gp;
        }
        stored = 0;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int r = 16 * (G * wave + g);
            if (r < nrows) {                             // (wave-uniform: the stores below are issued, for the lanes that have a row)
                stored += NT;
                float* out = a.out + (long long)(row0 + r + m) * C + 4 * q;
                if (r + m < nrows) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const float4 v = make_float4(acc[g][nt][0], acc[g][nt][1], acc[g][nt][2], acc[g][nt][3]);
                        wmax = dn_f4_amax(wmax, v);
                        *reinterpret_cast<float4*>(out + 16 * nt) = v;
                    }
                }
            }
        }
    }
#ifndef DN_EMULATE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the requests that ran past the end of the stream
#endif
#undef BW_WAIT
#undef BW_BARRIER
    if (a.out_amax) {                                   // one check-first atomic per workgroup
        float* red = reinterpret_cast<float*>(ring);
        wmax = ch_wave_max(wmax);
        __syncthreads();
        if (lane == 0) red[wave] = wmax;
        __syncthreads();
        if (tid == 0) {
            float mm = 0.f;
            for (int w = 0; w < NWAVES; ++w) mm = red[w] > mm ? red[w] : mm;
            if (mm > 0.f && mm > *reinterpret_cast<volatile float*>(a.out_amax)) atomicMax(reinterpret_cast<unsigned*>(a.out_amax), __float_as_uint(mm));
        }
    }
}

bool dn_backproject_wide_ok(int K, int C, int n_tiles) {
    if (K != 256 || C != 256 || n_tiles <= 0) return false;
    int g = dn_num_cus();
    if (g > n_tiles) g = n_tiles;
    g = (g + 7) / 8 * 8;
    return ((n_tiles + 7) / 8 + g / 8 - 1) / (g / 8) <= DN_BW_MAXP;
}
// floats of workspace for the split spectrum of n_mesh meshes
size_t dn_backproject_wide_ws_floats(int n_mesh, int K, int C) { return (size_t)n_mesh * (K / 32) * (3 * (C / 16) * 64) * 4; }

// out[rows] = evecs[rows] ys[mesh of the rows] for every tile (<= 128 rows of one mesh); ws: dn_backproject_wide_ws_floats() floats.
// out_amax (optional): max |out| is merged into the word (atomic max on the bit pattern).  Returns hipError_t as int, 1 for a shape not taken.
int dn_launch_backproject_wide(const DnTile* tiles, int n_tiles, int n_mesh, const float* evecs, const float* ys, float* ws, float* out, float* out_amax,
                               int K, int C, double acct_rows, hipStream_t stream) {
    if (!dn_backproject_wide_ok(K, C, n_tiles) || !tiles || !evecs || !ys || !ws || !out) return 1;
    dn_prof_begin(DN_K_BACKPROJECT, stream);
    DN_LAUNCH(spec_pieces3_kernel, dim3(K / 32, n_mesh, 1), dim3(1024, 1, 1), 0, stream, ys, K, C, reinterpret_cast<uint4*>(ws));
    BwArgs a;
    a.tiles = tiles; a.n_tiles = n_tiles; a.evecs = evecs; a.ysp = reinterpret_cast<const uint4*>(ws); a.out = out; a.out_amax = out_amax;
    int g = dn_num_cus();
    if (g > n_tiles) g = n_tiles;
    g = (g + 7) / 8 * 8;
    const size_t smem = (size_t)DN_BW_RING * (3 * (256 / 16) * 64) * sizeof(uint4) + (size_t)DN_BW_MAXP * sizeof(int4);
#ifndef DN_EMULATE
    static unsigned long long lds_opt_in = 0;
    { const int oe_ = dn_lds_opt_in(reinterpret_cast<const void*>(&backproject_wide_kernel<256, 256, DN_BW_G>), smem, &lds_opt_in); if (oe_) return oe_; }
#endif
    DN_LAUNCH((backproject_wide_kernel<256, 256, DN_BW_G>), dim3(g, 1, 1), dim3(64 * (8 / DN_BW_G), 1, 1), smem, stream, a);
    dn_prof_end(DN_K_BACKPROJECT, stream, 2.0 * acct_rows * K * C, 4.0 * acct_rows * (K + C));
    return (int)hipGetLastError();
}
