// dn_pointwise.hip -- the small fixed-order reductions and pointwise kernels around the two
// contraction engines (gfx950).  All of them are tiny, HBM/L2-bound passes:
//   spec_fwd  : sum split-V partials of Phi^T(Mx), scale by exp(-lambda*t)      (layers.py:62-64)
//   spec_bwd  : sum partials of Phi^T(d_xd), scale, reduce d/d(diffusion_time)  (autograd of :62-64)
//   reduce    : fixed-order sum of weight/bias-gradient partials
//   reduce_dA : same, plus the complex-structured recombination for A_re / A_im (layers.py:122-123)
//   mass_mean : mass-weighted global mean pooling and its gradient              (layers.py:397)
#include "dn_common.h"

__global__ __launch_bounds__(256) void spec_fwd_kernel(const float* partial, const int* mco, const float* evals,
                                                       const float* time, float* xs, float* ys, int K, int C) {
    const int m = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int KC = K * C;
    if (i >= KC) return;
    const int k = i / C, c = i % C;
    float s = 0.f;
    for (int ch = mco[m]; ch < mco[m + 1]; ++ch) s += partial[(long long)ch * KC + i];
    if (xs) xs[(long long)m * KC + i] = s;
    if (ys) ys[(long long)m * KC + i] = time ? expf(-evals[m * K + k] * time[c]) * s : s;
}

int dn_launch_spec_fwd(const float* partial, const int* mesh_chunk_off, const float* evals, const float* time,
                       float* xs, float* ys, int n_mesh, int K, int C, hipStream_t stream) {
    if (n_mesh <= 0 || K <= 0 || C <= 0) return 0;
    dim3 grid((K * C + 255) / 256, n_mesh, 1);
    DN_LAUNCH(spec_fwd_kernel, grid, dim3(256, 1, 1), 0, stream, partial, mesh_chunk_off, evals, time, xs, ys, K, C);
    return (int)hipGetLastError();
}

// block = 32 channels x 8 k-lanes; d_t partial per (mesh, channel)
__global__ __launch_bounds__(256) void spec_bwd_kernel(const float* partial, const int* mco, const float* evals,
                                                       const float* time, const float* xs, float* dxs, float* dt_part,
                                                       int K, int C) {
    __shared__ float red[8][32];
    const int m = blockIdx.y;
    const int cl = threadIdx.x & 31, kl = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    const int KC = K * C;
    float dt = 0.f;
    if (c < C) {
        const float t = time[c];
        const int ch0 = mco[m], ch1 = mco[m + 1];
        for (int k = kl; k < K; k += 8) {
            const int i = k * C + c;
            float d = 0.f;
            for (int ch = ch0; ch < ch1; ++ch) d += partial[(long long)ch * KC + i];
            const float lam = evals[m * K + k];
            const float coef = expf(-lam * t);
            dxs[(long long)m * KC + i] = coef * d;
            dt -= lam * d * coef * xs[(long long)m * KC + i];
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
}

int dn_launch_spec_bwd(const float* partial, const int* mesh_chunk_off, const float* evals, const float* time,
                       const float* xs, float* dxs, float* dt_part, int n_mesh, int K, int C, hipStream_t stream) {
    if (n_mesh <= 0 || K <= 0 || C <= 0) return 0;
    dim3 grid((C + 31) / 32, n_mesh, 1);
    DN_LAUNCH(spec_bwd_kernel, grid, dim3(256, 1, 1), 0, stream, partial, mesh_chunk_off, evals, time, xs, dxs,
              dt_part, K, C);
    return (int)hipGetLastError();
}

__global__ __launch_bounds__(256) void reduce_kernel(const float* partial, float* out, int n, long long stride,
                                                     long long len) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= len) return;
    float s = 0.f;
    for (int ch = 0; ch < n; ++ch) s += partial[ch * stride + i];
    out[i] = s;
}

int dn_launch_reduce(const float* partial, float* out, int n, long long stride, long long len, hipStream_t stream) {
    if (len <= 0) return 0;
    dim3 grid((unsigned)((len + 255) / 256), 1, 1);
    DN_LAUNCH(reduce_kernel, grid, dim3(256, 1, 1), 0, stream, partial, out, n, stride, len);
    return (int)hipGetLastError();
}

// partial: [n][2C][2C] of [dBre|dBim]^T [gx|gy];  dA_re = P00 + P11,  dA_im = P10 - P01
// (dA_im null: gradient rotations off, single matrix A, dA = P00 + P11 -> dA_re)
__global__ __launch_bounds__(256) void reduce_dA_kernel(const float* partial, float* dA_re, float* dA_im, int n, int C) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= C * C) return;
    const int o = i / C, c = i % C;
    const long long W = 2LL * C, stride = W * W;
    float p00 = 0.f, p11 = 0.f, p01 = 0.f, p10 = 0.f;
    for (int ch = 0; ch < n; ++ch) {
        const float* P = partial + ch * stride;
        p00 += P[o * W + c];
        p11 += P[(C + o) * W + C + c];
        if (dA_im) {
            p01 += P[o * W + C + c];
            p10 += P[(C + o) * W + c];
        }
    }
    dA_re[i] = p00 + p11;
    if (dA_im) dA_im[i] = p10 - p01;
}

int dn_launch_reduce_dA(const float* partial, float* dA_re, float* dA_im, int n, int C, hipStream_t stream) {
    if (C <= 0) return 0;
    dim3 grid((C * C + 255) / 256, 1, 1);
    DN_LAUNCH(reduce_dA_kernel, grid, dim3(256, 1, 1), 0, stream, partial, dA_re, dA_im, n, C);
    return (int)hipGetLastError();
}

// meshrows[m] = {row0, nrows, m, 0}.  block = 64 channels x 4 row lanes.
__global__ __launch_bounds__(256) void mass_mean_fwd_kernel(const DnTile* meshrows, const float* mass, const float* x,
                                                            float* out, float* msum, int C) {
    __shared__ float red[4][64];
    __shared__ float redm[4][64];
    const DnTile mr = meshrows[blockIdx.y];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    float s = 0.f, ms = 0.f;
    for (int r = rl; r < mr.nrows; r += 4) {
        const float mv = mass[mr.row0 + r];
        ms += mv;
        if (c < C) s += mv * x[(long long)(mr.row0 + r) * C + c];
    }
    red[rl][cl] = s;
    redm[rl][cl] = ms;
    __syncthreads();
    if (rl == 0) {
        const float st = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
        const float mt = (redm[0][cl] + redm[1][cl]) + (redm[2][cl] + redm[3][cl]);
        if (c < C) out[blockIdx.y * C + c] = st / mt;
        if (blockIdx.x == 0 && cl == 0) msum[blockIdx.y] = mt;
    }
}

int dn_launch_mass_mean_fwd(const DnTile* meshrows, const float* mass, const float* x, float* out, float* msum,
                            int n_mesh, int C, hipStream_t stream) {
    if (n_mesh <= 0 || C <= 0) return 0;
    dim3 grid((C + 63) / 64, n_mesh, 1);
    DN_LAUNCH(mass_mean_fwd_kernel, grid, dim3(256, 1, 1), 0, stream, meshrows, mass, x, out, msum, C);
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

int dn_launch_dtanh(const float* dg, const float* g, float* out, long long n, hipStream_t stream) {
    if (n <= 0) return 0;
    long long nb = (n + 255) / 256;
    if (nb > 4096) nb = 4096;
    DN_LAUNCH(dtanh_kernel, dim3((unsigned)nb, 1, 1), dim3(256, 1, 1), 0, stream, dg, g, out, n);
    return (int)hipGetLastError();
}
