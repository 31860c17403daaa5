// dn_spectral.hip -- the spatial gradient apply of DiffusionNetBlock.forward (layers.py:213-223) re-associated through the eigenbasis.
//
//   The reference computes x_diffuse = Phi ys (geometry.py:586-598, ys = exp(-lambda t) * spectrum) and then gx = G_X x_diffuse,
//   gy = G_Y x_diffuse with two sparse products (layers.py:217-223).  x_diffuse lies in span(Phi) exactly, so
//        gx = (G_X Phi) ys,   gy = (G_Y Phi) ys,   xd = Phi ys
//   are three dense [rows, K] x [K, C] products against the SAME per-mesh spectrum: the latency-bound CSR gather of the chained forward
//   kernel, the back-projection launch in front of it and xd's round trip through memory (inference) become streamed MFMA operands.
//   The differencing G (Phi ys) -- row sums of G are ~0, it amplifies the rounding noise of xd -- happens ONCE per mesh here, accumulated in
//   fp64, instead of in every forward in fp32.
//
//   sg_grad_kernel   G_X Phi, G_Y Phi ([V, K] fp32, fp64 accumulation in entry order) + per-mesh magnitudes of Phi, G_X Phi, G_Y Phi
//   sg_pack_kernel   [Phi | G_X Phi | G_Y Phi] -> fp16 (hi, lo) planes in the operand-fragment order of chain_fwd_kernel (dn_chain.hip):
//                    group g = 16 rows, element ((g 3 + op) KE + T) 128 + plane 64 + lane is the uint4 lane (m, q) feeds the MFMA as B
//                    operand of contraction step T: slot j <-> k = 32 T + 16 (j >> 2) + 4 q + (j & 3) of row (first row of g) + m.
//                    Every mesh is padded to whole units of dn_sg_unit_rows(K) rows -- 64 for K <= 128, 128 for K = 256: the rows of one workgroup
//                    pass of the kernel that consumes them (rows past the mesh's end are zeros): a pass never straddles two meshes.
//   spec_pieces_kernel   the scaled spectrum ys[mesh] ([K, C] fp32) as the transposed weight pieces the chained kernel streams through its
//                    LDS ring (piece T: rows = channels, contraction slots = eigenvectors 32 T .. 32 T + 31), fp16 (hi, lo), one power-of-two
//                    scale per mesh from its largest magnitude.
// Built once per mesh batch (the first two; dn_spectral_pack_f32) / once per block forward (the third, 4-5 us).
#include "dn_common.h"
#include "dn_chain_tiles.h"
#define DN_SA_MAXP 128   // passes per workgroup of spectral_apply_kernel (their descriptors live in LDS; the launcher checks)

// one workgroup per unit (<= 64 / 128 rows of one mesh), waves over its rows, lanes over the eigenvector index
__global__ __launch_bounds__(256) void sg_grad_kernel(const DnTile* units, int K, const float* evecs, const int* rowptr, const int* col, const float* vx,
                                                      const float* vy, float* gpx, float* gpy, float* amax4) {
    const DnTile u = units[blockIdx.x];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float mp = 0.f, mx = 0.f, my = 0.f, mn = 0.f;
    for (int r = wave; r < u.nrows; r += 4) {
        const long long row = (long long)u.row0 + r;
        const int beg = rowptr[row], end = rowptr[row + 1];
        float ss = 0.f;               // sum of squares of the row of Phi (its 2-norm bounds |xd| = |Phi ys| together with the columns' norms of ys)
        for (int k = lane; k < K; k += 64) {
            double ax = 0.0, ay = 0.0;
            for (int e = beg; e < end; ++e) {
                const double phi = (double)evecs[(long long)col[e] * K + k];
                ax += (double)vx[e] * phi;
                ay += (double)vy[e] * phi;
            }
            const float fx = (float)ax, fy = (float)ay;
            gpx[row * K + k] = fx;
            gpy[row * K + k] = fy;
            const float p = fabsf(evecs[row * K + k]);
            ss += p * p;
            mp = p > mp ? p : mp;
            mx = fabsf(fx) > mx ? fabsf(fx) : mx;
            my = fabsf(fy) > my ? fabsf(fy) : my;
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) ss += __shfl_xor(ss, d, 64);
        ss = sqrtf(ss) * 1.0001f;      // (rounded up: it is a bound)
        mn = ss > mn ? ss : mn;
    }
    // (non-finite operators: NaN never raises a word, it propagates through the packed data itself)
    dn_amax_commit<true>(amax4 + 4 * u.mesh + 0, mp);
    dn_amax_commit<true>(amax4 + 4 * u.mesh + 1, mx);
    dn_amax_commit<true>(amax4 + 4 * u.mesh + 2, my);
    dn_amax_commit<true>(amax4 + 4 * u.mesh + 3, mn);
}

template <int KE>
__global__ __launch_bounds__(256) void sg_pack_kernel(const DnTile* units, const float* evecs, const float* gpx, const float* gpy, const float* amax4,
                                                      uint4* out) {
    constexpr int K = 32 * KE;
    constexpr int GPU = (KE > 4 ? 128 : 64) / 16;      // 16-row groups per unit (dn_sg_unit_rows)
    const DnTile u = units[blockIdx.x];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m = lane & 15, q = lane >> 4;
    for (int grp = wave; grp < GPU; grp += 4) {
    const int r = 16 * grp + m;
    const bool valid = r < u.nrows;
    const long long row = (long long)u.row0 + (valid ? r : 0);
    uint4* dst = out + ((size_t)(GPU * blockIdx.x + grp) * 3 * KE) * 128 + lane;
#pragma unroll
    for (int op = 0; op < 3; ++op) {
        const float* src = (op == 0 ? evecs : (op == 1 ? gpx : gpy)) + row * K + 4 * q;
        const float s = dn_pow2_scale(amax4[4 * u.mesh + op]);
#pragma unroll
        for (int T = 0; T < KE; ++T) {
            float4 a = dn_f4_zero(), b = dn_f4_zero();
            if (valid) { a = *reinterpret_cast<const float4*>(src + 32 * T); b = *reinterpret_cast<const float4*>(src + 32 * T + 16); }
            uint4 hi, lo;
            ch_split8(a, b, s, hi, lo);
            dst[(size_t)(op * KE + T) * 128] = hi;
            dst[(size_t)(op * KE + T) * 128 + 64] = lo;
        }
    }
    }
}

// grid (K / 32, n_mesh): every workgroup measures its mesh's whole spectrum (64 KB out of L2 -- the same value in all of them) and writes one piece
//  ys_amax[mesh] = max |ys|, ys_amax[n_mesh + mesh] = the largest 2-norm of a column of ys (C % 4 == 0, C / 4 <= 1024 and a power of two)
__global__ __launch_bounds__(1024) void spec_pieces_kernel(const float* ys, int K, int C, uint4* out, float* ys_amax) {
    constexpr int NTHR = 1024;
    __shared__ float red[NTHR];
    __shared__ float4 sq[NTHR];
    const int tid = threadIdx.x, T = blockIdx.x, mesh = blockIdx.y;
    const int NT = C / 16, KE = K / 32;
    const float* y = ys + (size_t)mesh * K * C;
    float mx = 0.f;
    float4 ssq = dn_f4_zero();         // this thread's column group (tid % (C / 4)) over the rows tid / (C / 4), + NTHR / (C / 4), ...
    const int n4 = K * C / 4;          // (NTHR is a multiple of C / 4: a thread stays on its column group)
    for (int i0 = tid; i0 < n4; i0 += 4 * NTHR) {
        float4 v[4];
#pragma unroll
        for (int uu = 0; uu < 4; ++uu) {
            const bool in = i0 + NTHR * uu < n4;
            v[uu] = in ? *reinterpret_cast<const float4*>(y + 4 * (long long)(i0 + NTHR * uu)) : dn_f4_zero();
        }
#pragma unroll
        for (int uu = 0; uu < 4; ++uu) {
            mx = dn_f4_amax(mx, v[uu]);
            ssq.x += v[uu].x * v[uu].x; ssq.y += v[uu].y * v[uu].y; ssq.z += v[uu].z * v[uu].z; ssq.w += v[uu].w * v[uu].w;
        }
    }
    red[tid] = mx;
    sq[tid] = ssq;
    __syncthreads();
    for (int d = NTHR / 2; d > 0; d >>= 1) {
        if (tid < d) {
            red[tid] = red[tid + d] > red[tid] ? red[tid + d] : red[tid];
            if (d >= C / 4) { sq[tid].x += sq[tid + d].x; sq[tid].y += sq[tid + d].y; sq[tid].z += sq[tid + d].z; sq[tid].w += sq[tid + d].w; }   // (fixed order)
        }
        __syncthreads();
    }
    mx = red[0];
    if (T == 0 && tid == 0) {
        float cn = 0.f;
        for (int g = 0; g < C / 4; ++g) { const float4 t = sq[g]; cn = fmaxf(fmaxf(cn, fmaxf(t.x, t.y)), fmaxf(t.z, t.w)); }
        ys_amax[mesh] = mx;
        ys_amax[gridDim.y + mesh] = sqrtf(cn) * 1.0001f;
    }
    const float s = dn_pow2_scale(mx);
    uint4* o = out + ((size_t)mesh * KE + T) * (2 * NT * 64);
    for (int e = tid; e < NT * 64; e += NTHR) {
        const int nt = e >> 6, lane = e & 63;
        const int n = 16 * nt + (lane & 15), q = lane >> 4;
        const float* src = y + (long long)(32 * T + 4 * q) * C + n;      // piece row n = channel, slots = eigenvectors (the transposed form of chain_prep_kernel)
        float va[4], vb[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) { va[t] = src[(long long)t * C]; vb[t] = src[(long long)(16 + t) * C]; }
        uint4 hi, lo;
        ch_split8(va, vb, s, hi, lo);
        o[e] = hi;
        o[NT * 64 + e] = lo;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// spectral_apply_kernel<C, KE> (C = K = 256): [xd | gx | gy][rows] = [Phi | G_X Phi | G_Y Phi][rows] * ys[mesh] as a streaming launch of its own --
// the first stage of the C = 256 spectral-gradient forward, whose chained kernel has no registers left for it (dn_chain.hip, MODE 2).
//   One persistent 12-wave workgroup per CU; a pass = 64 rows of one mesh = 4 sixteen-row groups x 3 operands: EVERY WAVE OWNS ONE (group, operand)
//   pair -- 64 accumulator registers instead of the 192 a wave that computes all three needs, so three waves share a SIMD and cover each other's
//   latencies.  The spectrum's KE pieces stream through the 4-slot LDS ring of the chained kernels (requested by waves 0-7, consumed by all
//   twelve); a wave's operand fragments (pre-split fp16, 16 KiB contiguous per pass) stream through four 2-step register buffers, refilled six
//   pieces ahead of their use.  Results: scaled back, stored as rows of xd / gx / gy, their largest magnitudes into two device words.
#define DN_SA_WAVES 12
#define DN_SA_DMA_WAVES 8
template <int C, int KE>
__global__ __launch_bounds__(64 * DN_SA_WAVES) DN_WAVES_PER_EU(3) void spectral_apply_kernel(ChainArgs a) {
    constexpr int NT = C / 16;
    constexpr int PIECE = 2 * NT * 64;                  // uint4 per piece
    constexpr int NDMA = 64 * DN_SA_DMA_WAVES;
    constexpr int LPT = PIECE / NDMA;                   // DMA requests per requesting thread and piece
    constexpr int RING = DN_CH_RING;
    constexpr int CH_PF = 1;
    constexpr int HH = 1;                               // (the product macros' half count)
    constexpr int FB = 2, NBUF = 4, NPART = KE / FB;    // operand fragments: four 2-step buffers, one buffer refilled per 2 pieces
    static_assert(PIECE % NDMA == 0 && KE % FB == 0 && NPART == NBUF, "piece staging / fragment buffers");
    DN_DYN_SMEM(smem_raw);
    uint4* ring = reinterpret_cast<uint4*>(smem_raw);
    int4* pinfo = reinterpret_cast<int4*>(ring + RING * PIECE);                 // [DN_SA_MAXP] {first row, end of the mesh's rows, mesh, first group}
    float4* pscale = reinterpret_cast<float4*>(pinfo + DN_SA_MAXP);             // [DN_SA_MAXP] result scales of the three operands
#ifdef DN_EMULATE
    const unsigned lds0 = 0;
#else
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem_raw;
#endif
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = lane & 15, q = lane >> 4;
    const int grp_l = wave / 3, op = wave % 3;          // this wave's 16-row group of the pass and its operand (0 Phi -> xd, 1 -> gx, 2 -> gy)
    const bool dma_wave = wave < DN_SA_DMA_WAVES;

    // passes: 64-row halves of the batch's 128-row units, XCD-contiguous ranges as in the chained kernels
    const int SUB = a.sg_unit_rows / 64, GPU_ = a.sg_unit_rows / 16;
    const int units = a.sg_n_units * SUB;
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
    for (int pp = tid; pp < npass; pp += 64 * DN_SA_WAVES) {
        const int un = unit_of(pp);
        const DnTile tl = a.sg_units[un / SUB];
        const int sub = un % SUB;
        pinfo[pp] = int4{tl.row0 + 64 * sub, tl.row0 + tl.nrows, tl.mesh, GPU_ * (un / SUB) + 4 * sub};
        const float4 am = *reinterpret_cast<const float4*>(a.sg_amax + 4 * tl.mesh);
        const float ys_inv = ch_pow2_inv(dn_pow2_scale(a.ys_amax[tl.mesh]));
        pscale[pp] = make_float4(ys_inv * ch_pow2_inv(dn_pow2_scale(am.x)), ys_inv * ch_pow2_inv(dn_pow2_scale(am.y)),
                                 ys_inv * ch_pow2_inv(dn_pow2_scale(am.z)), 0.f);
    }
    int imesh = ch_uniform_i(a.sg_units[unit_of(0) / SUB].mesh);      // mesh of the pass whose pieces are being requested
    int mesh_nx = imesh;

    // ---- the piece stream (dn_chain.hip): KE pieces per pass, requested RING - 1 ahead by waves 0 .. DN_SA_DMA_WAVES - 1
    int sq = 0, rq = 0, gp = 0;
#ifdef DN_EMULATE
    const int wave_u = wave;
#else
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
#endif
    auto issue = [&]() {
        if (dma_wave) {
            const uint4* src_piece = a.ysp + ((size_t)imesh * KE + sq) * PIECE;
#pragma unroll
            for (int i = 0; i < LPT; ++i) {
                const int e0 = rq * PIECE + i * NDMA + wave_u * 64;
                ch_dma16(src_piece + i * NDMA + tid, ring + e0, lds0 + 16u * (unsigned)e0);
            }
        }
        if (sq + 1 == KE) { sq = 0; imesh = mesh_nx; } else ++sq;
        rq = rq + 1 == RING ? 0 : rq + 1;
    };
#ifdef DN_EMULATE
#define SA_WAIT(n) do {} while (0)
#define SA_WAIT_LDS() do {} while (0)
#define SA_BARRIER() __syncthreads()
#else
#define SA_WAIT(n) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(n) : "memory")
#define SA_WAIT_LDS() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define SA_BARRIER() do { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)
#endif
    // operand fragments of this wave: buffer b holds steps 2 b, 2 b + 1 (hi, lo planes) of the pass it was loaded for
    uint4 fq[NBUF][FB][2];
    auto load_buf = [&](const int grp, const int part, uint4 (&dst)[FB][2]) {
        const uint4* fp = a.sg_pack + ((size_t)grp * 3 + op) * (KE * 128) + lane;
#pragma unroll
        for (int t = 0; t < FB; ++t) {
            dst[t][0] = fp[(size_t)(FB * part + t) * 128];
            dst[t][1] = fp[(size_t)(FB * part + t) * 128 + 64];
        }
    };
#pragma unroll
    for (int i = 0; i < RING - 1; ++i) issue();
    if (dma_wave) SA_WAIT((RING - 2) * LPT); else SA_WAIT_LDS();
    SA_BARRIER();                                     // (publishes the pass table as well)
    {
        const int g0 = ch_uniform_i(pinfo[0].w) + grp_l;
#pragma unroll
        for (int b = 0; b < NBUF; ++b) load_buf(g0, b, fq[b]);
    }
    float wmax = 0.f;
    for (int pass = 0; pass < npass; ++pass) {
        const int4 pi_ = pinfo[pass];
        const float4 ps_ = pscale[pass];
        const int4 pn_ = pinfo[pass + 1 < npass ? pass + 1 : pass];
        const int row = ch_uniform_i(pi_.x) + 16 * grp_l + m;
        const bool live = row < ch_uniform_i(pi_.y);
        mesh_nx = ch_uniform_i(pn_.z);
        const int grp_nx = ch_uniform_i(pn_.w) + grp_l;
        const bool pf = pass + 1 < npass;
        const float u = ch_uniform(op == 0 ? ps_.x : (op == 1 ? ps_.y : ps_.z));
        dn_f32x4 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = dn_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int part = 0; part < NPART; ++part) {
#pragma unroll
            for (int t = 0; t < FB; ++t) {
                const uint4* ws_ = ring + (gp % RING) * PIECE;
                issue();
                CH_MMA1(acc, fq[part][t][0], fq[part][t][1]);
                // the piece waited for is DMA(gp + 1); loads younger than it that may stay in flight: the ring's two youngest pieces and -- while a next
                // pass exists -- the last buffer refill (2 FB requests, issued two pieces ago at the most)
                if (dma_wave) { if (pf) SA_WAIT((RING - 2) * LPT + 2 * FB); else SA_WAIT((RING - 2) * LPT); }
                else SA_WAIT_LDS();                   // (no ring requests of its own: its LDS reads must be done before the slot is overwritten)
                SA_BARRIER();
                ++gp;
            }
            if (pf) load_buf(grp_nx, part, fq[part]);  // the buffer just consumed takes the same steps of the next pass
        }
        float* out = (op == 0 ? a.xd_out : (op == 1 ? a.gx : a.gy)) + (long long)row * C + 4 * q;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const float4 v = make_float4(acc[nt][0] * u, acc[nt][1] * u, acc[nt][2] * u, acc[nt][3] * u);
            wmax = dn_f4_amax(wmax, v);
            if (live) *reinterpret_cast<float4*>(out + 16 * nt) = v;
        }
    }
#ifndef DN_EMULATE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the requests that ran past the end of the stream
#endif
#undef SA_WAIT
#undef SA_WAIT_LDS
#undef SA_BARRIER
    // largest magnitudes: xd (operand 0) and gx / gy (operands 1, 2): one check-first atomic per workgroup and word
    {
        float* red = reinterpret_cast<float*>(ring);
        wmax = ch_wave_max(wmax);
        __syncthreads();
        if (lane == 0) red[wave] = wmax;
        __syncthreads();
        if (tid < 2) {
            float mm = 0.f;
            for (int w = 0; w < DN_SA_WAVES; ++w) if ((w % 3 == 0) == (tid == 0)) mm = red[w] > mm ? red[w] : mm;
            float* word = tid == 0 ? a.xd_amax_out : a.g_amax;
            if (word && mm > 0.f && mm > *reinterpret_cast<volatile float*>(word)) atomicMax(reinterpret_cast<unsigned*>(word), __float_as_uint(mm));
        }
    }
}

int dn_launch_spectral_apply(const ChainArgs& a, int C, hipStream_t stream) {
    if (C != 256 || a.sg_unit_rows != 128 || !a.sg_pack || !a.sg_units || !a.ysp || !a.xd_out || !a.gx || !a.gy || a.sg_n_units <= 0) return 1;
    const int units = a.sg_n_units * 2;
    int g = dn_num_cus();
    if (g > units) g = units;
    g = (g + 7) / 8 * 8;
    if (((units + 7) / 8 + g / 8 - 1) / (g / 8) > DN_SA_MAXP) return 1;
    const size_t smem = (size_t)DN_CH_RING * (2 * (256 / 16) * 64) * sizeof(uint4) + (size_t)DN_SA_MAXP * 32;
#ifndef DN_EMULATE
    static unsigned long long lds_opt_in = 0;
    { const int oe_ = dn_lds_opt_in(reinterpret_cast<const void*>(&spectral_apply_kernel<256, 8>), smem, &lds_opt_in); if (oe_) return oe_; }
#endif
    dn_prof_begin(DN_K_SPECTRAL, stream);
    DN_LAUNCH((spectral_apply_kernel<256, 8>), dim3(g, 1, 1), dim3(64 * DN_SA_WAVES, 1, 1), smem, stream, a);
    // algorithmic traffic: the three packed operands read, xd / gx / gy written
    dn_prof_end(DN_K_SPECTRAL, stream, 3.0 * 2.0 * (double)a.V * 256.0 * 256.0, 3.0 * 4.0 * (double)a.V * 256.0 * 2.0);
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------------
int dn_sg_unit_rows(int K) { return K > 128 ? 128 : 64; }
int dn_sg_units_host(const int* sizes, int n_mesh, int K, DnTile* out) {
    const int UR = dn_sg_unit_rows(K);
    int n = 0;
    long long row0 = 0;
    for (int i = 0; i < n_mesh; ++i) {
        for (int r = 0; r < sizes[i]; r += UR) {
            if (out) out[n] = DnTile{(int)(row0 + r), sizes[i] - r < UR ? sizes[i] - r : UR, i, 0};
            ++n;
        }
        row0 += sizes[i];
    }
    return n;
}
size_t dn_sg_pack_elems(int n_units, int K) { return (size_t)n_units * (dn_sg_unit_rows(K) / 16) * 3 * (K / 32) * 128; }

int dn_launch_sg_pack(const DnTile* units, int n_units, int n_mesh, int K, const float* evecs, const int* rowptr, const int* col, const float* vx,
                      const float* vy, float* gpx, float* gpy, float* amax4, uint4* out, hipStream_t stream) {
    if (n_units <= 0) return 0;
    if (K % 32 != 0 || K <= 0 || K > 256) return 1;
    const int e0 = (int)hipMemsetAsync(amax4, 0, (size_t)n_mesh * 4 * sizeof(float), stream);
    if (e0) return e0;
    DN_LAUNCH(sg_grad_kernel, dim3(n_units, 1, 1), dim3(256, 1, 1), 0, stream, units, K, evecs, rowptr, col, vx, vy, gpx, gpy, amax4);
    switch (K / 32) {
#define DN_SG_CASE(ke) case ke: DN_LAUNCH(sg_pack_kernel<ke>, dim3(n_units, 1, 1), dim3(256, 1, 1), 0, stream, units, evecs, gpx, gpy, amax4, out); break
        DN_SG_CASE(1); DN_SG_CASE(2); DN_SG_CASE(3); DN_SG_CASE(4); DN_SG_CASE(5); DN_SG_CASE(6); DN_SG_CASE(7); DN_SG_CASE(8);
#undef DN_SG_CASE
        default: return 1;
    }
    return (int)hipGetLastError();
}

int dn_launch_spec_pieces(const float* ys, int n_mesh, int K, int C, uint4* out, float* ys_amax, hipStream_t stream) {
    if (n_mesh <= 0) return 0;
    if (K % 32 != 0 || C % 16 != 0 || (((uintptr_t)ys | (uintptr_t)out) & 15) != 0 || (C / 4) > 1024 || ((C / 4) & (C / 4 - 1)) != 0) return 1;
    dn_prof_begin(DN_K_SMALL, stream);
    DN_LAUNCH(spec_pieces_kernel, dim3(K / 32, n_mesh, 1), dim3(1024, 1, 1), 0, stream, ys, K, C, out, ys_amax);
    dn_prof_end(DN_K_SMALL, stream, 0.0, 0.0);
    return (int)hipGetLastError();
}
