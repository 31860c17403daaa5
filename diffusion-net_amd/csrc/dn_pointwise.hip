// dn_pointwise.hip -- the small fixed-order reductions and pointwise kernels around the two
// contraction engines (gfx950).  All of them are tiny, HBM/L2-bound passes:
//   spec_fwd  : sum split-V partials of Phi^T(Mx), scale by exp(-lambda*t)      (layers.py:62-64)
//   spec_bwd  : sum partials of Phi^T(d_xd), scale, reduce d/d(diffusion_time)  (autograd of :62-64)
//   reduce    : fixed-order sum of weight/bias-gradient partials
//   reduce_dA : same, plus the complex-structured recombination for A_re / A_im (layers.py:122-123)
//   mass_mean : mass-weighted global mean pooling and its gradient              (layers.py:397)
// (the loss / head kernels live in dn_head.hip)
#include "dn_common.h"

// dys: [n_mesh,K,C] = evecs^T d_xd (already reduced over chunks).  In place: dys <- exp(-lambda t) * dys (the
// spectrum handed to from_basis), and d_t partial per (mesh, channel).  block = 32 channels x 8 k-lanes.
__global__ __launch_bounds__(256) void spec_bwd_kernel(float* dys, const float* evals, const float* time, const float* xs,
                                                       float* dt_part, int K, int C, float* dys_amax) {
    __shared__ float red[8][32];
    const int m = blockIdx.y;
    const int cl = threadIdx.x & 31, kl = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    const long long KC = (long long)K * C;
    float dt = 0.f, amax = 0.f;
    if (c < C) {
        const float t = time[c];
        for (int k0 = kl; k0 < K; k0 += 32) {     // four k per step with their loads in flight together (64 workgroups in all: latency-bound)
            float d[4], lam[4], x[4];
            long long i[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = k0 + 8 * u;
                const bool ok = k < K;
                i[u] = m * KC + (long long)(ok ? k : kl) * C + c;
                d[u] = dys[i[u]];
                x[u] = xs[i[u]];
                lam[u] = ok ? evals[m * K + k] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (k0 + 8 * u < K) {
                    const float coef = expf(-lam[u] * t);
                    dys[i[u]] = coef * d[u];
                    amax = fabsf(coef * d[u]) > amax ? fabsf(coef * d[u]) : amax;
                    dt -= lam[u] * d[u] * coef * x[u];     // same order as k ascending per lane: bitwise the same sum
                }
            }
        }
    }
    red[kl][cl] = dt;
    __syncthreads();
    if (kl == 0 && c < C) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) s += red[j][cl];
        dt_part[m * C + c] = s;
    }
    if (dys_amax) dn_amax_commit(dys_amax, amax);    // the scaled spectrum feeds the split-fp16 from_basis product
}

int dn_launch_spec_bwd(float* dys, const float* evals, const float* time, const float* xs, float* dt_part, int n_mesh, int K,
                       int C, hipStream_t stream, float* dys_amax) {
    if (n_mesh <= 0 || K <= 0 || C <= 0) return 0;
    dim3 grid((C + 31) / 32, n_mesh, 1);
    dn_prof_begin(DN_K_SMALL, stream);
    DN_LAUNCH(spec_bwd_kernel, grid, dim3(256, 1, 1), 0, stream, dys, evals, time, xs, dt_part, K, C, dys_amax);
    dn_prof_end(DN_K_SMALL, stream, 0.0, 0.0);
    return (int)hipGetLastError();
}

// The three launches of the diffusion backward's spectral step in one: per-mesh sum of the split-V partials of evecs^T d_xd (chunks in
// ascending order), scaling by exp(-lambda t) (the spectrum handed to from_basis), and the d_t contributions summed over the 8 eigenvalues a
// workgroup owns.  block = 32 channel quads x 8 eigenvalues; grid = (ceil(C/128) * ceil(K/8), n_mesh).
// dt_part: [n_mesh * ceil(K/8)][C] -- its fixed-order sum over the rows (dn_spec_bwd_dt_rows) is d_t; the caller hands it to the reduction
// launch it issues anyway.
__global__ __launch_bounds__(256) void spec_bwd_fused_kernel(const float* partial, const int* mco, const float* evals, const float* time,
                                                             const float* xs, float* dys, float* dt_part, int K, int C, int kgroups, float* dys_amax) {
    __shared__ __attribute__((aligned(16))) float red[8][8][128];          // [eigenvalue of the group][chunk lane][channel]; reused for the d_t sums
    const int m = blockIdx.y;
    const int cb = blockIdx.x / kgroups, kg = blockIdx.x % kgroups;
    const int cl = threadIdx.x & 31, kl = threadIdx.x >> 5;
    const int c = cb * 128 + 4 * cl;
    const long long KC = (long long)K * C;
    const int beg = mco[m], end = mco[m + 1];
    // phase 1: lane kl sums chunks beg + kl, beg + kl + 8, ... for each of the group's 8 eigenvalues; a pass of 4 chunks x 8 eigenvalues has
    // its 32 loads in flight together (one latency per pass: a mesh has ~32 chunks)
    float4 a[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = dn_f4_zero();
    if (c < C) {
        for (int ch0 = beg + kl; ch0 < end; ch0 += 32) {
            float4 v[4][8];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int ch = ch0 + 8 * u;
                const bool cok = ch < end;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int k = kg * 8 + j;
                    v[u][j] = (cok && k < K) ? *reinterpret_cast<const float4*>(partial + (long long)ch * KC + (long long)k * C + c) : dn_f4_zero();
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < 8; ++j) { a[j].x += v[u][j].x; a[j].y += v[u][j].y; a[j].z += v[u][j].z; a[j].w += v[u][j].w; }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) *reinterpret_cast<float4*>(&red[j][kl][4 * cl]) = a[j];
    __syncthreads();
    // phase 2: thread (cl, kl) finishes eigenvalue kg * 8 + kl: the 8 lane sums in order, exp(-lambda t), d_t contribution
    const int k = kg * 8 + kl;
    float dt[4] = {0.f, 0.f, 0.f, 0.f}, amax = 0.f;
    if (c < C && k < K) {
        float d[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int l = 0; l < 8; ++l) {
            const float4 t = *reinterpret_cast<const float4*>(&red[kl][l][4 * cl]);
            d[0] += t.x; d[1] += t.y; d[2] += t.z; d[3] += t.w;
        }
        const float lam = evals[m * K + k];
        const long long o = m * KC + (long long)k * C + c;
        const float4 t4 = *reinterpret_cast<const float4*>(time + c);
        const float4 x4 = *reinterpret_cast<const float4*>(xs + o);
        const float tt[4] = {t4.x, t4.y, t4.z, t4.w}, xx[4] = {x4.x, x4.y, x4.z, x4.w};
        float y[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float coef = expf(-lam * tt[e]);
            y[e] = coef * d[e];
            dt[e] = -(lam * d[e] * coef * xx[e]);
            amax = fabsf(y[e]) > amax ? fabsf(y[e]) : amax;
        }
        *reinterpret_cast<float4*>(dys + o) = make_float4(y[0], y[1], y[2], y[3]);
    }
    __shared__ float wave_max[4];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { const float o = __shfl_xor(amax, d, 64); amax = o > amax ? o : amax; }
    __syncthreads();                          // everybody has read its lane sums: the array is reused
    *reinterpret_cast<float4*>(&red[0][kl][4 * cl]) = make_float4(dt[0], dt[1], dt[2], dt[3]);
    if ((threadIdx.x & 63) == 0) wave_max[threadIdx.x >> 6] = amax;
    __syncthreads();
    if (kl == 0 && c < C) {
        float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float4 t = *reinterpret_cast<const float4*>(&red[0][j][4 * cl]);
            s[0] += t.x; s[1] += t.y; s[2] += t.z; s[3] += t.w;
        }
        *reinterpret_cast<float4*>(dt_part + ((long long)m * kgroups + kg) * C + c) = make_float4(s[0], s[1], s[2], s[3]);
    }
    // one check-first atomic per workgroup (a thousand posted atomics on one word serialise in its L2 channel: 8 us behind a 10 us kernel)
    if (dys_amax && threadIdx.x == 0) {
        const float mm = fmaxf(fmaxf(wave_max[0], wave_max[1]), fmaxf(wave_max[2], wave_max[3]));
        if (mm > 0.f && mm > *reinterpret_cast<volatile float*>(dys_amax)) atomicMax(reinterpret_cast<unsigned*>(dys_amax), __float_as_uint(mm));
    }
}

int dn_spec_bwd_dt_rows(int n_mesh, int K) { return n_mesh * ((K + 7) / 8); }
bool dn_spec_bwd_fused_ok(const float* partial, const float* time, const float* xs, const float* dys, const float* dt_part, int C) {
    return C % 4 == 0 && (((uintptr_t)partial | (uintptr_t)time | (uintptr_t)xs | (uintptr_t)dys | (uintptr_t)dt_part) & 15) == 0;
}
int dn_launch_spec_bwd_fused(const float* partial, const int* mesh_chunk_off, const float* evals, const float* time, const float* xs, float* dys,
                             float* dt_part, int n_mesh, int K, int C, hipStream_t stream, float* dys_amax) {
    if (n_mesh <= 0 || K <= 0 || C <= 0) return 0;
    const int kgroups = (K + 7) / 8;
    dim3 grid((unsigned)(((C + 127) / 128) * kgroups), n_mesh, 1);
    dn_prof_begin(DN_K_SMALL, stream);
    DN_LAUNCH(spec_bwd_fused_kernel, grid, dim3(256, 1, 1), 0, stream, partial, mesh_chunk_off, evals, time, xs, dys, dt_part, K, C, kgroups, dys_amax);
    dn_prof_end(DN_K_SMALL, stream, 0.0, 0.0);
    return (int)hipGetLastError();
}

// Fixed-order segmented sum of split-V partials: out[s][i] = sum_{ch in segment s} partial[ch][i].
// block = 32 column groups x 8 chunk lanes: lane kl sums chunks beg+kl, beg+kl+8, ... (several loads in flight),
// the 8 lane sums are then combined in order through LDS -> bitwise reproducible, bandwidth-bound.
struct SegStore {   // plain epilogue: out[s][i + e] = sum
    float* out; long long len;
    __device__ __forceinline__ float operator()(int s, long long i, int e, float t) const { out[(long long)s * len + i + e] = t; return t; }
};
// returns max |epilogue value| over what this thread wrote (0 for the threads that write nothing)
template <int VEC, class EPI>
__device__ __forceinline__ float seg_reduce_body(const float* partial, int s, int beg, int end, long long len, const EPI& epi) {
    __shared__ float red[8][32 * VEC];
    const int cl = threadIdx.x & 31, kl = threadIdx.x >> 5;
    const long long i = ((long long)blockIdx.x * 32 + cl) * VEC;
    float a[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) a[e] = 0.f;
    if (i < len) {
        int ch = beg + kl;
        for (; ch + 24 < end; ch += 32) {
            float v[4][VEC];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float* src = partial + (long long)(ch + 8 * u) * len + i;
                if (VEC == 4) {
                    const float4 t = *reinterpret_cast<const float4*>(src);
                    v[u][0] = t.x; v[u][1 % VEC] = t.y; v[u][2 % VEC] = t.z; v[u][3 % VEC] = t.w;
                } else {
                    v[u][0] = src[0];
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int e = 0; e < VEC; ++e) a[e] += v[u][e];
        }
        for (; ch < end; ch += 8) {
            const float* src = partial + (long long)ch * len + i;
            if (VEC == 4) {
                const float4 t = *reinterpret_cast<const float4*>(src);
                a[0] += t.x; a[1 % VEC] += t.y; a[2 % VEC] += t.z; a[3 % VEC] += t.w;
            } else {
                a[0] += src[0];
            }
        }
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) red[kl][cl * VEC + e] = a[e];
    __syncthreads();
    float om = 0.f;
    if (kl == 0 && i < len) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            float t = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) t += red[j][cl * VEC + e];
            const float w = fabsf(epi(s, i, e, t));
            om = w > om ? w : om;
        }
    }
    return om;
}

template <int VEC>
__global__ __launch_bounds__(256) void seg_reduce_kernel(const float* partial, const int* seg_off, int n, float* out,
                                                         long long len) {
    const int s = blockIdx.y;
    seg_reduce_body<VEC>(partial, s, seg_off ? seg_off[s] : 0, seg_off ? seg_off[s + 1] : n, len, SegStore{out, len});
}

// spec_fwd: the per-mesh sum of the to_basis partials (same parallel fixed-order reduction) with the spectral scaling as its
// epilogue: xs[m] = sum, ys[m][k][c] = exp(-lambda_mk t_c) * sum  (layers.py:62-64).  Either output may be null.
struct SpecEpi {
    const float* evals; const float* time; float* xs; float* ys; int K, C;
    float* ys_amax;   // optional: max |ys| (the scaled spectrum is the B operand of the split-fp16 from_basis product)
    __device__ __forceinline__ float operator()(int m, long long i, int e, float t) const {
        const long long o = (long long)m * K * C + i + e;
        if (xs) xs[o] = t;
        float y = 0.f;
        if (ys) {
            const int k = (int)((i + e) / C), c = (int)((i + e) % C);
            y = time ? expf(-evals[m * K + k] * time[c]) * t : t;
            ys[o] = y;
        }
        return y;
    }
};
template <int VEC>
__global__ __launch_bounds__(256) void spec_fwd_kernel(const float* partial, const int* mco, SpecEpi epi) {
    const int m = blockIdx.y;
    const float om = seg_reduce_body<VEC>(partial, m, mco[m], mco[m + 1], (long long)epi.K * epi.C, epi);
    if (epi.ys_amax) dn_amax_commit<true>(epi.ys_amax, om);
}

int dn_launch_spec_fwd(const float* partial, const int* mesh_chunk_off, const float* evals, const float* time,
                       float* xs, float* ys, int n_mesh, int K, int C, hipStream_t stream, float* ys_amax) {
    if (n_mesh <= 0 || K <= 0 || C <= 0) return 0;
    const long long len = (long long)K * C;
    const bool vec = (len % 4 == 0) && ((uintptr_t)partial % 16 == 0);
    const SpecEpi epi{evals, time, xs, ys, K, C, ys_amax};
    dn_prof_begin(DN_K_SMALL, stream);
    if (vec) {
        DN_LAUNCH(spec_fwd_kernel<4>, dim3((unsigned)((len / 4 + 31) / 32), n_mesh, 1), dim3(256, 1, 1), 0, stream, partial, mesh_chunk_off, epi);
    } else {
        DN_LAUNCH(spec_fwd_kernel<1>, dim3((unsigned)((len + 31) / 32), n_mesh, 1), dim3(256, 1, 1), 0, stream, partial, mesh_chunk_off, epi);
    }
    dn_prof_end(DN_K_SMALL, stream, 0.0, 0.0);
    return (int)hipGetLastError();
}

// two independent whole-range sums in one launch (a weight gradient, float4 path, and its bias gradient): blockIdx.y picks the job
__global__ __launch_bounds__(256) void seg_reduce_pair_kernel(const float* pa, float* oa, long long la, const float* pb, float* ob,
                                                              long long lb, int n) {
    if (blockIdx.y == 0) seg_reduce_body<4>(pa, 0, 0, n, la, SegStore{oa, la});
    else seg_reduce_body<1>(pb, 0, 0, n, lb, SegStore{ob, lb});
}

int dn_launch_seg_reduce(const float* partial, const int* seg_off, int nseg, int n, float* out, long long len,
                         hipStream_t stream) {
    if (len <= 0 || nseg <= 0) return 0;
    const bool vec = (len % 4 == 0) && ((uintptr_t)partial % 16 == 0) && ((uintptr_t)out % 16 == 0);
    dn_prof_begin(DN_K_SMALL, stream);
    if (vec) {
        dim3 grid((unsigned)((len / 4 + 31) / 32), nseg, 1);
        DN_LAUNCH(seg_reduce_kernel<4>, grid, dim3(256, 1, 1), 0, stream, partial, seg_off, n, out, len);
    } else {
        dim3 grid((unsigned)((len + 31) / 32), nseg, 1);
        DN_LAUNCH(seg_reduce_kernel<1>, grid, dim3(256, 1, 1), 0, stream, partial, seg_off, n, out, len);
    }
    dn_prof_end(DN_K_SMALL, stream, 0.0, 4.0 * (double)len * ((double)(seg_off ? 0 : n) + 1.0));
    return (int)hipGetLastError();
}

int dn_launch_reduce_pair(const float* pa, float* oa, long long la, const float* pb, float* ob, long long lb, int n,
                          hipStream_t stream) {
    if (n <= 0 || la <= 0 || lb <= 0) return DN_ERR_BAD_MODE;
    const bool vec = (la % 4 == 0) && ((uintptr_t)pa % 16 == 0) && ((uintptr_t)oa % 16 == 0);
    if (!vec) {   // odd sizes: two plain launches
        int e = dn_launch_seg_reduce(pa, nullptr, 1, n, oa, la, stream);
        return e ? e : dn_launch_seg_reduce(pb, nullptr, 1, n, ob, lb, stream);
    }
    const long long ba = (la / 4 + 31) / 32, bb = (lb + 31) / 32;
    dn_prof_begin(DN_K_SMALL, stream);
    DN_LAUNCH(seg_reduce_pair_kernel, dim3((unsigned)(ba > bb ? ba : bb), 2, 1), dim3(256, 1, 1), 0, stream, pa, oa, la, pb, ob, lb, n);
    dn_prof_end(DN_K_SMALL, stream, 0.0, 4.0 * (double)(la + lb) * ((double)n + 1.0));
    return (int)hipGetLastError();
}

// whole-range sum of [n][2 * half] partials whose two halves go to different arrays (dA_re | dA_im, dn_tn_da.hip)
struct SegSplitStore {
    float* o0; float* o1; long long half;
    __device__ __forceinline__ float operator()(int, long long i, int e, float t) const {
        const long long j = i + e;
        if (j < half) o0[j] = t; else o1[j - half] = t;
        return t;
    }
};
__global__ __launch_bounds__(256) void seg_reduce_split_kernel(const float* partial, int n, float* o0, float* o1, long long half) {
    seg_reduce_body<4>(partial, 0, 0, n, 2 * half, SegSplitStore{o0, o1, half});
}
int dn_launch_reduce_split(const float* partial, int n, float* o0, float* o1, long long half, hipStream_t stream) {
    if (n <= 0 || half <= 0 || half % 4 != 0 || (uintptr_t)partial % 16 != 0) return DN_ERR_BAD_MODE;
    dn_prof_begin(DN_K_SMALL, stream);
    DN_LAUNCH(seg_reduce_split_kernel, dim3((unsigned)((2 * half / 4 + 31) / 32), 1, 1), dim3(256, 1, 1), 0, stream, partial, n, o0, o1, half);
    dn_prof_end(DN_K_SMALL, stream, 0.0, 8.0 * (double)half * ((double)n + 1.0));
    return (int)hipGetLastError();
}

__global__ __launch_bounds__(256) void multi_reduce_kernel(MrJobs J) {
    const MrJob& j = J.j[blockIdx.y];
    if ((long long)blockIdx.x * 128 >= j.len) return;          // (uniform per block)
    seg_reduce_body<4>(j.src, 0, 0, j.n, j.len, SegSplitStore{j.o0, j.o1, j.half});
}
int dn_launch_multi_reduce(const MrJobs& jobs, hipStream_t stream) {
    if (jobs.count <= 0) return 0;
    long long nb = 0;
    double bytes = 0.0;
    for (int i = 0; i < jobs.count; ++i) {
        const long long b = (jobs.j[i].len / 4 + 31) / 32;
        nb = b > nb ? b : nb;
        bytes += 4.0 * (double)jobs.j[i].len * ((double)jobs.j[i].n + 1.0);
    }
    dn_prof_begin(DN_K_SMALL, stream);
    DN_LAUNCH(multi_reduce_kernel, dim3((unsigned)nb, jobs.count, 1), dim3(256, 1, 1), 0, stream, jobs);
    dn_prof_end(DN_K_SMALL, stream, 0.0, bytes);
    return (int)hipGetLastError();
}

int dn_launch_reduce(const float* partial, float* out, int n, long long stride, long long len, hipStream_t stream) {
    if (stride != len) return DN_ERR_BAD_MODE;
    return dn_launch_seg_reduce(partial, nullptr, 1, n, out, len, stream);
}

// P: [2C][2C] = [dBre|dBim]^T [gx|gy] summed over all rows;  dA_re = P00 + P11,  dA_im = P10 - P01
// (dA_im null: gradient rotations off, single matrix A, dA = P00 + P11 -> dA_re)
__global__ __launch_bounds__(256) void combine_dA_kernel(const float* P, float* dA_re, float* dA_im, int C) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= C * C) return;
    const int o = i / C, c = i % C;
    const long long W = 2LL * C;
    dA_re[i] = P[o * W + c] + P[(C + o) * W + C + c];
    if (dA_im) dA_im[i] = P[(C + o) * W + c] - P[o * W + C + c];
}

int dn_launch_combine_dA(const float* P, float* dA_re, float* dA_im, int C, hipStream_t stream) {
    if (C <= 0) return 0;
    dim3 grid((C * C + 255) / 256, 1, 1);
    DN_LAUNCH(combine_dA_kernel, grid, dim3(256, 1, 1), 0, stream, P, dA_re, dA_im, C);
    return (int)hipGetLastError();
}

// meshrows[m] = {row0, nrows, m, 0}.  block = 64 channels x 4 row lanes.
__global__ __launch_bounds__(256) void mass_mean_fwd_kernel(const DnTile* meshrows, const float* mass, const float* x,
                                                            float* out, float* msum, int C, int CL) {
    // CL channel lanes x (256 / CL) row lanes; every row lane keeps four independent partial sums (four rows in flight), all
    // combined in a fixed order: bitwise reproducible
    __shared__ float red[8][64];
    __shared__ float redm[8][64];
    const DnTile mr = meshrows[blockIdx.y];
    const int RL = 256 / CL;
    const int cl = threadIdx.x % CL, rl = threadIdx.x / CL;
    const int c = blockIdx.x * CL + cl;
    const bool cok = c < C;
    float s[4] = {0.f, 0.f, 0.f, 0.f}, ms[4] = {0.f, 0.f, 0.f, 0.f};
    for (int r0 = rl; r0 < mr.nrows; r0 += 4 * RL) {
        float mv[4], xv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r = r0 + u * RL;
            const bool ok = r < mr.nrows;
            const long long row = mr.row0 + (ok ? r : r0);
            mv[u] = ok ? mass[row] : 0.f;
            xv[u] = (ok && cok) ? x[row * C + c] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) { ms[u] += mv[u]; s[u] = fmaf(mv[u], xv[u], s[u]); }
    }
    red[rl][cl] = (s[0] + s[1]) + (s[2] + s[3]);
    redm[rl][cl] = (ms[0] + ms[1]) + (ms[2] + ms[3]);
    __syncthreads();
    if (rl == 0) {
        float st = 0.f, mt = 0.f;
        for (int k = 0; k < RL; ++k) { st += red[k][cl]; mt += redm[k][cl]; }
        if (cok) out[blockIdx.y * C + c] = st / mt;
        if (blockIdx.x == 0 && cl == 0) msum[blockIdx.y] = mt;
    }
}

int dn_launch_mass_mean_fwd(const DnTile* meshrows, const float* mass, const float* x, float* out, float* msum,
                            int n_mesh, int C, hipStream_t stream) {
    if (n_mesh <= 0 || C <= 0) return 0;
    const int CL = C <= 32 ? 32 : 64;
    dim3 grid((C + CL - 1) / CL, n_mesh, 1);
    DN_LAUNCH(mass_mean_fwd_kernel, grid, dim3(256, 1, 1), 0, stream, meshrows, mass, x, out, msum, C, CL);
    return (int)hipGetLastError();
}

__global__ __launch_bounds__(256) void mass_mean_bwd_kernel(const DnTile* tiles, const float* mass, const float* msum,
                                                            const float* dout, float* dx, int C) {
    const DnTile t = tiles[blockIdx.x];
    const float inv = 1.f / msum[t.mesh];
    for (int i = threadIdx.x; i < t.nrows * C; i += 256) {
        const int r = i / C, c = i % C;
        dx[(long long)(t.row0 + r) * C + c] = dout[t.mesh * C + c] * (mass[t.row0 + r] * inv);
    }
}

int dn_launch_mass_mean_bwd(const DnTile* tiles, int ntiles, const float* mass, const float* msum, const float* dout,
                            float* dx, int C, hipStream_t stream) {
    if (ntiles <= 0 || C <= 0) return 0;
    DN_LAUNCH(mass_mean_bwd_kernel, dim3(ntiles, 1, 1), dim3(256, 1, 1), 0, stream, tiles, mass, msum, dout, dx, C);
    return (int)hipGetLastError();
}

// d(pre-tanh) = d_g * (1 - g^2)
__global__ __launch_bounds__(256) void dtanh_kernel(const float* dg, const float* g, float* out, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float t = g[i];
        out[i] = dg[i] * (1.f - t * t);
    }
}

// max |x| of several buffers in one launch (operand magnitudes for the split-fp16 engine): blockIdx.y = job, grid-stride over the buffer,
// one atomic max per wave.  Streams float4 where the buffer allows it.
__global__ __launch_bounds__(256) void amax_kernel(AmaxJobs jobs) {
    const int j = blockIdx.y;
    const float* x = jobs.src[j];
    const long long n = jobs.n[j];
    float m = 0.f;
    const long long stride = (long long)gridDim.x * 256;
    if ((((uintptr_t)x) & 15) == 0) {
        const long long n4 = n / 4;
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) m = dn_f4_amax(m, *reinterpret_cast<const float4*>(x + 4 * i));
        for (long long i = 4 * n4 + (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) m = fabsf(x[i]) > m ? fabsf(x[i]) : m;
    } else {
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) m = fabsf(x[i]) > m ? fabsf(x[i]) : m;
    }
    dn_amax_commit<true>(jobs.dst[j], m);
}
// Start-of-call bookkeeping of the split-fp16 engine in ONE launch: job j < count: max |src[j]| by a single workgroup (weights: a few
// 10k elements), max-combined in LDS and STORED (jobs sharing a destination are merged by the host into consecutive jobs of one
// workgroup chain -- see `same`); the last workgroup zeroes the words the later kernels of the call accumulate into and forwards one
// word.  Replaces three memsets, a copy and the atomic amax launch (4 x ~6 us of blit kernels per block call).
__global__ __launch_bounds__(1024) void amax_init_kernel(AmaxInit a) {
    __shared__ float red[1024];
    const int j = blockIdx.x, tid = threadIdx.x;
    if (j == a.jobs.count) {
        for (int r = 0; r < a.nzero; ++r)
            for (int i = tid; i < a.zero_n[r]; i += 1024) a.zero[r][i] = 0.f;
        if (tid == 0 && a.copy_src && a.copy_dst) *a.copy_dst = *a.copy_src;
        for (int i = tid; i < a.clamp_n; i += 1024) { const float v = a.clamp_p[i]; a.clamp_p[i] = v < a.clamp_min ? a.clamp_min : v; }   // (NaN stays NaN, as torch.clamp)
        return;
    }
    if (a.same[j]) return;                      // merged into the previous job's workgroup
    float m = 0.f;
    for (int jj = j; jj < a.jobs.count && (jj == j || a.same[jj]); ++jj) {
        const float* x = a.jobs.src[jj];
        const long long n = a.jobs.n[jj];
        if ((((uintptr_t)x) & 15) == 0) {
            const long long n4 = n / 4;
            for (long long i = tid; i < n4; i += 4096) {     // four float4 in flight per thread (one workgroup: latency, not bandwidth)
                float4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { const long long k = i + 1024 * u; v[u] = *reinterpret_cast<const float4*>(x + 4 * (k < n4 ? k : i)); }
#pragma unroll
                for (int u = 0; u < 4; ++u) m = dn_f4_amax(m, v[u]);
            }
            for (long long i = 4 * n4 + tid; i < n; i += 1024) m = fabsf(x[i]) > m ? fabsf(x[i]) : m;
        } else {
            for (long long i = tid; i < n; i += 1024) m = fabsf(x[i]) > m ? fabsf(x[i]) : m;
        }
    }
    red[tid] = m;
    __syncthreads();
    for (int d = 512; d > 0; d >>= 1) {
        if (tid < d) red[tid] = red[tid + d] > red[tid] ? red[tid + d] : red[tid];
        __syncthreads();
    }
    if (tid == 0) *a.jobs.dst[j] = red[0];
}
int dn_launch_amax_init(const AmaxInit& a_in, hipStream_t stream) {
    AmaxInit a = a_in;
    for (int j = 0; j < a.jobs.count; ++j) a.same[j] = (j > 0 && a.jobs.dst[j] == a.jobs.dst[j - 1]) ? 1 : 0;
    dn_prof_begin(DN_K_SMALL, stream);
    DN_LAUNCH(amax_init_kernel, dim3(a.jobs.count + 1, 1, 1), dim3(1024, 1, 1), 0, stream, a);
    dn_prof_end(DN_K_SMALL, stream, 0.0, 0.0);
    return (int)hipGetLastError();
}

int dn_launch_amax(const AmaxJobs& jobs, hipStream_t stream) {
    if (jobs.count <= 0) return 0;
    long long nmax = 0;
    for (int j = 0; j < jobs.count; ++j) nmax = jobs.n[j] > nmax ? jobs.n[j] : nmax;
    long long nb = (nmax / 4 + 256 * 8 - 1) / (256 * 8);
    nb = nb < 1 ? 1 : (nb > 2048 ? 2048 : nb);
    dn_prof_begin(DN_K_SMALL, stream);
    DN_LAUNCH(amax_kernel, dim3((unsigned)nb, jobs.count, 1), dim3(256, 1, 1), 0, stream, jobs);
    dn_prof_end(DN_K_SMALL, stream, 0.0, 0.0);
    return (int)hipGetLastError();
}

int dn_launch_dtanh(const float* dg, const float* g, float* out, long long n, hipStream_t stream) {
    if (n <= 0) return 0;
    long long nb = (n + 255) / 256;
    if (nb > 4096) nb = 4096;
    DN_LAUNCH(dtanh_kernel, dim3((unsigned)nb, 1, 1), dim3(256, 1, 1), 0, stream, dg, g, out, n);
    return (int)hipGetLastError();
}

// ---- thin products with a tiny contraction / output width (first_lin: C_in = 3 or 16; last_lin backward) ----
// A thread owns 4 output columns for all the rows it visits and keeps their K <= 16 weights in registers; per row it
// reads the K inputs (wave-broadcast) and writes one float4.  Bandwidth-bound on the output stream.
template <int KMAX>
__global__ __launch_bounds__(256) void smallk_rows_kernel(const float* x, int K, const float* W, int w_kn, const float* bias,
                                                          int N, float* out, long long rows, float* o_amax) {
    constexpr int UR = KMAX <= 8 ? 4 : 2;       // independent row passes in flight per thread
    const int n4 = (N + 3) / 4;                 // column groups per row
    const int cg = threadIdx.x % n4, rl = threadIdx.x / n4;
    const int rows_per_pass = 256 / n4;         // rows a block covers per pass (threads beyond rows_per_pass*n4 idle)
    const bool idle = rl >= rows_per_pass;      // (stays for the wave-wide reduction of the magnitude word)
    const int n = cg * 4;
    float w[4][KMAX], b4[4], om = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        b4[e] = (bias && n + e < N) ? bias[n + e] : 0.f;
#pragma unroll
        for (int k = 0; k < KMAX; ++k)
            w[e][k] = (k < K && n + e < N) ? (w_kn ? W[(long long)k * N + n + e] : W[(long long)(n + e) * K + k]) : 0.f;
    }
    const bool vec = (N % 4 == 0) && (((uintptr_t)out & 15) == 0);
    const bool kvec = (K == KMAX) && (((uintptr_t)x & 15) == 0);
    const long long stride = (long long)gridDim.x * rows_per_pass;
    for (long long r0 = (long long)blockIdx.x * rows_per_pass + rl; r0 < rows && !idle; r0 += UR * stride) {
        float xv[UR][KMAX];
#pragma unroll
        for (int u = 0; u < UR; ++u) {           // all input loads of the UR passes first (clamped row, no branch)
            const long long r = r0 + u * stride < rows ? r0 + u * stride : r0;
            if (kvec) {            // K == KMAX, 16-byte aligned rows: KMAX / 4 vector loads instead of KMAX dword loads
#pragma unroll
                for (int k = 0; k < KMAX; k += 4) {
                    const float4 t = *reinterpret_cast<const float4*>(x + r * K + k);
                    xv[u][k] = t.x; xv[u][k + 1] = t.y; xv[u][k + 2] = t.z; xv[u][k + 3] = t.w;
                }
            } else {
#pragma unroll
                for (int k = 0; k < KMAX; ++k) xv[u][k] = x[r * K + (k < K ? k : 0)];
            }
        }
#pragma unroll
        for (int u = 0; u < UR; ++u) {
            const long long r = r0 + u * stride;
            float acc[4] = {b4[0], b4[1], b4[2], b4[3]};
#pragma unroll
            for (int k = 0; k < KMAX; ++k)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] = fmaf(xv[u][k], w[e][k], acc[e]);   // w is 0 for k >= K
            if (r < rows) {
#pragma unroll
                for (int e = 0; e < 4; ++e) om = fmaxf(om, fabsf(acc[e]));   // one v_max_f32 with |.| each (a column past N has zero weights and bias: acc = 0)
                if (vec) {
                    *reinterpret_cast<float4*>(out + r * N + n) = make_float4(acc[0], acc[1], acc[2], acc[3]);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (n + e < N) out[r * N + n + e] = acc[e];
                }
            }
        }
    }
    // max |out| for the split-fp16 engine of the consumer (first_lin -> block 0, last_lin's input gradient -> last block): the threads are
    // long-lived (grid-stride), so this is one read of the word per wave at the very end and an atomic only from the few that raise it
    // One commit per workgroup, not per wave: all ~3500 waves of a 7k-vertex launch are resident together, every one of them sees the word
    // still at zero, and 3500 atomics on one address cost 25 us (8 ns each, serialised in one L2 channel) on top of a 7 us kernel.
    if (o_amax) {
        __shared__ float wave_max[4];
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) { const float o = __shfl_xor(om, d, 64); om = o > om ? o : om; }
        if ((threadIdx.x & 63) == 0) wave_max[threadIdx.x >> 6] = om;
        __syncthreads();
        if (threadIdx.x == 0) {
            const float m = fmaxf(fmaxf(wave_max[0], wave_max[1]), fmaxf(wave_max[2], wave_max[3]));
            if (m > *reinterpret_cast<volatile float*>(o_amax)) atomicMax(reinterpret_cast<unsigned*>(o_amax), __float_as_uint(m));
        }
    }
}

int dn_launch_smallk_rows(const float* x, int K, const float* W, int w_kn, const float* bias, int N, float* out,
                          long long rows, hipStream_t stream, float* o_amax) {
    if (rows <= 0 || N <= 0) return 0;
    if (K > 16 || N > 1024) return DN_ERR_BAD_MODE;
    const int n4 = (N + 3) / 4;
    const int rpp = 256 / n4;
    long long nb = (rows + rpp - 1) / rpp;
    // every workgroup ends with one atomic on the magnitude word, and in a kernel this short they all arrive together (the check-first
    // read sees the word unraised): the atomics of one address retire at ~8 ns each -- measured at 160 k rows: 46 / 32 / 28 us with 4096 / 1024 / 512 workgroups
#ifndef DN_SMALLK_MAX_WGS
#define DN_SMALLK_MAX_WGS 512
#endif
    if (nb > DN_SMALLK_MAX_WGS) nb = DN_SMALLK_MAX_WGS;
    dn_prof_begin(DN_K_SMALL, stream);
    if (K <= 4) {
        DN_LAUNCH(smallk_rows_kernel<4>, dim3((unsigned)nb, 1, 1), dim3(256, 1, 1), 0, stream, x, K, W, w_kn, bias, N, out, rows, o_amax);
    } else if (K <= 8) {
        DN_LAUNCH(smallk_rows_kernel<8>, dim3((unsigned)nb, 1, 1), dim3(256, 1, 1), 0, stream, x, K, W, w_kn, bias, N, out, rows, o_amax);
    } else {
        DN_LAUNCH(smallk_rows_kernel<16>, dim3((unsigned)nb, 1, 1), dim3(256, 1, 1), 0, stream, x, K, W, w_kn, bias, N, out, rows, o_amax);
    }
    dn_prof_end(DN_K_SMALL, stream, 2.0 * rows * K * N, 4.0 * rows * (K + N));
    return (int)hipGetLastError();
}

// out[m, n] = sum_r A[r,m] B[r,n] with N <= 16: block b sums rows b, b+nblk, ... for all m (thread = one m, loops M in
// strides of 256), writes ws[b][m][n]; a fixed-order seg_reduce finishes.  Bandwidth-bound stream over A.
__global__ __launch_bounds__(256) void smalln_tn_kernel(const float* A, int M, const float* B, int N, long long rows, float* ws) {
    for (int m = threadIdx.x; m < M; m += 256) {
        float acc[16];
#pragma unroll
        for (int n = 0; n < 16; ++n) acc[n] = 0.f;
        for (long long r = blockIdx.x; r < rows; r += gridDim.x) {
            const float a = A[r * M + m];
#pragma unroll
            for (int n = 0; n < 16; ++n)
                if (n < N) acc[n] = fmaf(a, B[r * N + n], acc[n]);
        }
        for (int n = 0; n < N; ++n) ws[((long long)blockIdx.x * M + m) * N + n] = acc[n];
    }
}

int dn_launch_smalln_tn(const float* A, int M, const float* B, int N, long long rows, float* out, float* ws, int nblk,
                        hipStream_t stream) {
    if (rows <= 0 || M <= 0 || N <= 0 || N > 16) return N > 16 ? DN_ERR_BAD_MODE : 0;
    dn_prof_begin(DN_K_SMALL, stream);
    DN_LAUNCH(smalln_tn_kernel, dim3(nblk, 1, 1), dim3(256, 1, 1), 0, stream, A, M, B, N, rows, ws);
    dn_prof_end(DN_K_SMALL, stream, 2.0 * rows * M * N, 4.0 * rows * (M + N));
    return dn_launch_seg_reduce(ws, nullptr, 1, nblk, out, (long long)M * N, stream);
}

// ---- thin split-V product for the two ends of the network (first_lin: C_in = 3 inputs; last_lin: C_out <= 8 classes):
//      P[m][n] = sum_r X[r,m] Y[r,n]   with X wide (M columns, float4 per thread) and Y thin (N <= 32, eight columns per pass over X),
//      plus the column sums sx[m] = sum_r X[r,m] and sy[n] = sum_r Y[r,n] (whichever of them is the bias gradient) from the SAME pass.
//      first_lin: X = d_out, Y = x      -> dW[o][i] = P[o][i] (m-major store), db = sx
//      last_lin : X = x,     Y = d_out  -> dW[o][c] = P[c][o] (n-major store), db = sy
//      Block = 8 row lanes x 32 column groups, two rows in flight per lane; partials per block, fixed-order reduce afterwards.
#define DN_THIN_NMAX 8
__global__ __launch_bounds__(256) void thin_tn_kernel(const float* X, int M, const float* Y, int N, long long rows, int nm_major,
                                                      float* ws_p, float* ws_sx, float* ws_sy) {
    __shared__ float red[8][128];
    const int rl = threadIdx.x >> 5, c4 = threadIdx.x & 31;
    const long long per = (rows + gridDim.x - 1) / gridDim.x;
    const long long r_beg = (long long)blockIdx.x * per, r_end = (r_beg + per < rows) ? r_beg + per : rows;
    float* wp = ws_p + (long long)blockIdx.x * M * N;
    for (int m0 = 0; m0 < M; m0 += 128)
    for (int n0 = 0; n0 < N; n0 += DN_THIN_NMAX) {       // N > 8 (e.g. 30 classes, 16 hks features): X is streamed once per 8 columns of Y
        const int m = m0 + 4 * c4;
        const bool mok = m < M;
        float acc[DN_THIN_NMAX][4], sx[4] = {0.f, 0.f, 0.f, 0.f}, sy[DN_THIN_NMAX];
#pragma unroll
        for (int n = 0; n < DN_THIN_NMAX; ++n) { sy[n] = 0.f; acc[n][0] = acc[n][1] = acc[n][2] = acc[n][3] = 0.f; }
        for (long long r0 = r_beg + rl; r0 < r_end; r0 += 16) {
            float4 xv[2];
            float yv[2][DN_THIN_NMAX];
#pragma unroll
            for (int u = 0; u < 2; ++u) {          // both rows' loads first
                const long long r = r0 + 8 * u;
                const bool ok = r < r_end;
                const long long rr = ok ? r : r0;
                xv[u] = *reinterpret_cast<const float4*>(X + rr * M + (mok ? m : 0));
                if (!ok || !mok) xv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int n = 0; n < DN_THIN_NMAX; ++n) yv[u][n] = (ok && n0 + n < N) ? Y[rr * N + n0 + n] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                sx[0] += xv[u].x; sx[1] += xv[u].y; sx[2] += xv[u].z; sx[3] += xv[u].w;
#pragma unroll
                for (int n = 0; n < DN_THIN_NMAX; ++n) {
                    sy[n] += yv[u][n];
                    acc[n][0] = fmaf(xv[u].x, yv[u][n], acc[n][0]); acc[n][1] = fmaf(xv[u].y, yv[u][n], acc[n][1]);
                    acc[n][2] = fmaf(xv[u].z, yv[u][n], acc[n][2]); acc[n][3] = fmaf(xv[u].w, yv[u][n], acc[n][3]);
                }
            }
        }
        // fixed-order sum over the 8 row lanes, one quantity at a time through 4 KiB of LDS
#pragma unroll
        for (int q = 0; q <= DN_THIN_NMAX; ++q) {      // q < NMAX: P[.][n0 + q];  q == NMAX: sx
            if (q < DN_THIN_NMAX && n0 + q >= N) continue;
            if (q == DN_THIN_NMAX && (!ws_sx || n0 != 0)) continue;
            __syncthreads();
#pragma unroll
            for (int e = 0; e < 4; ++e) red[rl][4 * c4 + e] = q < DN_THIN_NMAX ? acc[q][e] : sx[e];
            __syncthreads();
            if (threadIdx.x < 128 && m0 + (int)threadIdx.x < M) {
                float t = 0.f;
#pragma unroll
                for (int k = 0; k < 8; ++k) t += red[k][threadIdx.x];
                const int mm = m0 + threadIdx.x;
                if (q < DN_THIN_NMAX) wp[nm_major ? (long long)(n0 + q) * M + mm : (long long)mm * N + n0 + q] = t;
                else ws_sx[(long long)blockIdx.x * M + mm] = t;
            }
        }
        if (ws_sy && m0 == 0) {                        // sy: every column group holds the same sums; group 0 of each row lane reports
            __syncthreads();
            if (c4 == 0)
#pragma unroll
                for (int n = 0; n < DN_THIN_NMAX; ++n) red[rl][n] = sy[n];
            __syncthreads();
            if (threadIdx.x < DN_THIN_NMAX && n0 + (int)threadIdx.x < N) {
                float t = 0.f;
#pragma unroll
                for (int k = 0; k < 8; ++k) t += red[k][threadIdx.x];
                ws_sy[(long long)blockIdx.x * N + n0 + threadIdx.x] = t;
            }
        }
    }
}

// dW (+ db) of a thin linear layer in two launches: the streaming pass above and one fixed-order reduce of both partial sets
int dn_launch_thin_tn(const float* X, int M, const float* Y, int N, long long rows, int nm_major, int db_is_sx, float* dW, float* db,
                      float* ws_p, float* ws_s, int nblk, hipStream_t stream) {
    if (rows <= 0 || M <= 0 || N <= 0) return 0;
    if (N > 4 * DN_THIN_NMAX || M % 4 != 0 || ((uintptr_t)X & 15) != 0 || nblk <= 0) return DN_ERR_BAD_MODE;
    dn_prof_begin(DN_K_SMALL, stream);
    DN_LAUNCH(thin_tn_kernel, dim3(nblk, 1, 1), dim3(256, 1, 1), 0, stream, X, M, Y, N, rows, nm_major,
              ws_p, (db && db_is_sx) ? ws_s : (float*)nullptr, (db && !db_is_sx) ? ws_s : (float*)nullptr);
    dn_prof_end(DN_K_SMALL, stream, 2.0 * rows * M * N, 4.0 * rows * (M + N));
    int err = (int)hipGetLastError();
    if (err) return err;
    if (db) return dn_launch_reduce_pair(ws_p, dW, (long long)M * N, ws_s, db, db_is_sx ? M : N, nblk, stream);
    return dn_launch_seg_reduce(ws_p, nullptr, 1, nblk, dW, (long long)M * N, stream);
}

// ---- heat kernel signature (geometry.py:600-633): out[b][v][s] = sum_k exp(-lambda[b][k] * t[s]) * evecs[b][v][k]^2.
//      One streaming pass over the eigenbasis (the input feature of the "hks" experiments); block = 64 rows x 4 groups of
//      4 scales, 32-wide k chunks staged through LDS (coalesced float4 loads of Phi, exp() once per (k, scale) and block).
__global__ __launch_bounds__(256) void hks_kernel(const float* evals, const float* evecs, const float* scales, int V, int K, int S,
                                                  long long scale_stride, float* out) {
    __shared__ float sphi[64][33];
    __shared__ float scoef[32][17];
    const int b = blockIdx.z, s_base = blockIdx.y * 16;
    const int tid = threadIdx.x, rl = tid >> 2, g4 = tid & 3;
    const long long row0 = (long long)blockIdx.x * 64;
    const float* ev = evals + (long long)b * K;
    const float* ph = evecs + (long long)b * V * K;
    const float* sc = scales + (long long)b * scale_stride;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < K; k0 += 32) {
        for (int i = tid; i < 64 * 32; i += 256) {           // Phi chunk, squared
            const int r = i >> 5, kk = i & 31;
            const long long row = row0 + r;
            const float v = (row < V && k0 + kk < K) ? ph[row * K + k0 + kk] : 0.f;
            sphi[r][kk] = v * v;
        }
        for (int i = tid; i < 32 * 16; i += 256) {           // exp(-lambda t) for the chunk
            const int kk = i >> 4, ss = i & 15;
            scoef[kk][ss] = (k0 + kk < K && s_base + ss < S) ? expf(-ev[k0 + kk] * sc[s_base + ss]) : 0.f;
        }
        __syncthreads();
#pragma unroll 8
        for (int kk = 0; kk < 32; ++kk) {
            const float p2 = sphi[rl][kk];
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] = fmaf(p2, scoef[kk][4 * g4 + e], acc[e]);
        }
        __syncthreads();
    }
    const long long row = row0 + rl;
    if (row < V) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int ss = s_base + 4 * g4 + e;
            if (ss < S) out[((long long)b * V + row) * S + ss] = acc[e];
        }
    }
}

int dn_launch_hks(const float* evals, const float* evecs, const float* scales, int B, int V, int K, int S, long long scale_stride,
                  float* out, hipStream_t stream) {
    if (B <= 0 || V <= 0 || K <= 0 || S <= 0) return 0;
    dn_prof_begin(DN_K_SMALL, stream);
    DN_LAUNCH(hks_kernel, dim3((unsigned)((V + 63) / 64), (unsigned)((S + 15) / 16), (unsigned)B), dim3(256, 1, 1), 0, stream, evals, evecs,
              scales, V, K, S, scale_stride, out);
    dn_prof_end(DN_K_SMALL, stream, 3.0 * B * (double)V * K * S, 4.0 * B * ((double)V * K + (double)V * S));
    return (int)hipGetLastError();
}
