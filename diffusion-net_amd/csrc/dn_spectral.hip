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
