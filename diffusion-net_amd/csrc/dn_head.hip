// dn_head.hip -- the head of the network on the far side of the path, one pass each way:
//   forward : [gather-mean over the 2/3 vertices of an edge/face (layers.py:379-391)] -> [log_softmax (the scripts' last_activation,
//             human_segmentation_original.py:75)] -> log-probabilities out -> [NLL / label-smoothed log loss against labels
//             (F.nll_loss, human_segmentation_original.py:136; utils.label_smoothing_log_loss, utils.py:18-24), mean over valid rows]
//   backward: d_x[v] = sum over the outputs i that gathered v of (1/div) dz_i, dz_i = dlp_i - softmax_i * sum_c dlp_i[c],
//             dlp_i = d_logp_i (from autograd, optional) + the loss term -g/count * smoothed one-hot(label_i)   -- one kernel,
//             driven by the transposed gather pattern, no intermediate [n_out, C] gradient tensor.
// Every stage in brackets is optional, so the same two kernels are: F.nll_loss alone, the label-smoothing loss alone,
// face-mean + log_softmax (what the unmodified scripts get), and the fully fused face-mean + log_softmax + loss.
// Rows labelled ignore_index = -100 (torch's default) do not contribute and do not count.  Any other label outside [0, C) is an error in
// torch (a device-side assert); here it turns the loss into NaN -- loud, and without a host synchronisation.
// A row is handled by G = 2^k <= 64 lanes (G >= C for C <= 64), lane l owns classes l, l + G, ...
#include "dn_common.h"

#define DN_HEAD_CPL 32  // classes per lane: C <= 2048 (instantiated for 1, 2, 4, 8, 16, 32 classes per lane)
#define DN_HEAD_IGNORE (-100LL)

__device__ __forceinline__ float head_group_sum(float v, int G) {
    for (int m = G >> 1; m > 0; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}
__device__ __forceinline__ float head_group_max(float v, int G) {
    for (int m = G >> 1; m > 0; m >>= 1) { const float o = __shfl_xor(v, m, 64); v = o > v ? o : v; }
    return v;
}

// CPL = classes per lane (1, 2, 4 or 8): with a fixed CPL = 8 the eight-class head spent most of its time on seven dead class slots
template <int CPL>
__global__ __launch_bounds__(256) void head_fwd_kernel(HeadArgs a, int G) {
    __shared__ float red[2][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int rows_per_wave = 64 / G, gl = lane % G, gr = lane / G;
    const int rows_per_block = 4 * rows_per_wave;
    float loss_sum = 0.f, cnt = 0.f;
    for (long long i0 = (long long)blockIdx.x * rows_per_block; i0 < a.n_out; i0 += (long long)gridDim.x * rows_per_block) {
        const long long i = i0 + wave * rows_per_wave + gr;
        const bool live = i < a.n_out;                       // dead rows run along (the shuffles need every lane) on row 0
        const long long ii = live ? i : 0;
        float z[CPL];
        int beg = 0, end = 1;
        if (a.rowptr) { beg = a.rowptr[ii]; end = a.rowptr[ii + 1]; }
#pragma unroll
        for (int k = 0; k < CPL; ++k) {
            const int c = gl + k * G;
            float s = 0.f;
            if (c < a.C) {
                if (a.rowptr) { for (int j = beg; j < end; ++j) s += a.x[(long long)a.col[j] * a.ldx + c]; s *= a.inv_div; }
                else s = a.x[ii * a.ldx + c];
            }
            z[k] = s;
        }
        if (a.lsm) {
            float mx = -3.0e38f;
#pragma unroll
            for (int k = 0; k < CPL; ++k) if (gl + k * G < a.C) mx = z[k] > mx ? z[k] : mx;
            mx = head_group_max(mx, G);
            float se = 0.f;
#pragma unroll
            for (int k = 0; k < CPL; ++k) if (gl + k * G < a.C) se += expf(z[k] - mx);
            se = head_group_sum(se, G);
            const float lse = mx + logf(se);
#pragma unroll
            for (int k = 0; k < CPL; ++k) z[k] -= lse;
        }
        if (a.logp && live) {
#pragma unroll
            for (int k = 0; k < CPL; ++k) if (gl + k * G < a.C) a.logp[i * a.C + gl + k * G] = z[k];
        }
        if (a.labels) {
            const long long t = a.labels[ii];
            const bool valid = live && t >= 0 && t < a.C;
            float at = 0.f, all = 0.f;
#pragma unroll
            for (int k = 0; k < CPL; ++k) if (gl + k * G < a.C) { all += z[k]; if (gl + k * G == t) at += z[k]; }
            at = head_group_sum(at, G);
            all = head_group_sum(all, G);
            if (valid && gl == 0) {
                const float off = a.C > 1 ? a.smoothing / (float)(a.C - 1) : 0.f;
                loss_sum -= (1.f - a.smoothing) * at + off * (all - at);
                cnt += 1.f;
            }
            if (live && !valid && t != DN_HEAD_IGNORE && gl == 0) loss_sum += __uint_as_float(0x7fc00000u);   // not a class, not ignore_index
        }
    }
    if (!a.labels) return;
    // fixed-order block sums: lanes of a wave by shuffles, the four waves through LDS
    loss_sum = head_group_sum(loss_sum, 64);
    cnt = head_group_sum(cnt, 64);
    if (lane == 0) { red[0][wave] = loss_sum; red[1][wave] = cnt; }
    __syncthreads();
    if (tid == 0) {
        a.partial[blockIdx.x] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        a.partial[gridDim.x + blockIdx.x] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
}

// loss = sum(partial loss) / sum(valid counts); count kept for the backward
__global__ __launch_bounds__(256) void head_finish_kernel(const float* partial, int nb, float* loss, float* count) {
    __shared__ float red[2][256];
    float s = 0.f, c = 0.f;
    for (int i = threadIdx.x; i < nb; i += 256) { s += partial[i]; c += partial[nb + i]; }
    red[0][threadIdx.x] = s;
    red[1][threadIdx.x] = c;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) { red[0][threadIdx.x] += red[0][threadIdx.x + w]; red[1][threadIdx.x] += red[1][threadIdx.x + w]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { loss[0] = red[0][0] / red[1][0]; count[0] = red[1][0]; }
}

template <int CPL>
__global__ __launch_bounds__(256) void head_bwd_kernel(HeadArgs a, int G) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int rows_per_wave = 64 / G, gl = lane % G, gr = lane / G;
    const int rows_per_block = 4 * rows_per_wave;
    const float gscale = (a.labels && a.g_loss) ? -a.g_loss[0] / a.count[0] : 0.f;
    const float off = a.C > 1 ? a.smoothing / (float)(a.C - 1) : 0.f;
    for (long long v0 = (long long)blockIdx.x * rows_per_block; v0 < a.n_src; v0 += (long long)gridDim.x * rows_per_block) {
        const long long v = v0 + wave * rows_per_wave + gr;
        const bool live = v < a.n_src;
        const long long vv = live ? v : 0;
        int beg = 0, end = 1;
        if (a.t_rowptr) { beg = a.t_rowptr[vv]; end = a.t_rowptr[vv + 1]; }
        // the lanes of a group walk the same outputs; groups of one wave may have different counts -> pad to the wave's maximum
        int n_it = end - beg;
        for (int m = 32; m >= G && m > 0; m >>= 1) { const int o = __shfl_xor(n_it, m, 64); n_it = o > n_it ? o : n_it; }
        float acc[CPL];
#pragma unroll
        for (int k = 0; k < CPL; ++k) acc[k] = 0.f;
        for (int it = 0; it < n_it; ++it) {
            const bool on = beg + it < end;
            const long long i = a.t_rowptr ? a.t_col[on ? beg + it : beg < end ? beg : 0] : vv;
            long long t = -1;
            if (a.labels) t = a.labels[i];
            const bool valid = t >= 0 && t < a.C;
            float dlp[CPL], sum = 0.f;
#pragma unroll
            for (int k = 0; k < CPL; ++k) {
                const int c = gl + k * G;
                float d = 0.f;
                if (c < a.C) {
                    if (a.d_logp) d = a.d_logp[i * a.C + c];
                    if (valid) d += gscale * (c == t ? 1.f - a.smoothing : off);
                }
                dlp[k] = d;
                sum += d;
            }
            if (a.lsm) sum = head_group_sum(sum, G);
#pragma unroll
            for (int k = 0; k < CPL; ++k) {
                const int c = gl + k * G;
                if (c < a.C && on) {
                    float dz = dlp[k];
                    if (a.lsm) dz -= expf(a.logp[i * a.C + c]) * sum;
                    acc[k] += dz;
                }
            }
        }
        if (live) {
#pragma unroll
            for (int k = 0; k < CPL; ++k) if (gl + k * G < a.C) a.d_x[v * a.C + gl + k * G] = acc[k] * a.inv_div;
        }
    }
}

static int head_group(int C) { int g = 1; while (g < C && g < 64) g <<= 1; return g; }

int dn_launch_head_fwd(const HeadArgs& a, int nb, float* loss, float* count, hipStream_t stream) {
    if (a.n_out <= 0 || a.C <= 0) return 0;
    if (a.C > 64 * DN_HEAD_CPL) return DN_ERR_BAD_MODE;
    const int G = head_group(a.C), rpb = 4 * (64 / G);
    int blocks = (int)(((long long)a.n_out + rpb - 1) / rpb);
    if (blocks > nb) blocks = nb;
    dn_prof_begin(DN_K_SMALL, stream);
    const int cpl = (a.C + G - 1) / G;
    if (cpl <= 1) DN_LAUNCH(head_fwd_kernel<1>, dim3(blocks, 1, 1), dim3(256, 1, 1), 0, stream, a, G);
    else if (cpl <= 2) DN_LAUNCH(head_fwd_kernel<2>, dim3(blocks, 1, 1), dim3(256, 1, 1), 0, stream, a, G);
    else if (cpl <= 4) DN_LAUNCH(head_fwd_kernel<4>, dim3(blocks, 1, 1), dim3(256, 1, 1), 0, stream, a, G);
    else if (cpl <= 8) DN_LAUNCH(head_fwd_kernel<8>, dim3(blocks, 1, 1), dim3(256, 1, 1), 0, stream, a, G);
    else if (cpl <= 16) DN_LAUNCH(head_fwd_kernel<16>, dim3(blocks, 1, 1), dim3(256, 1, 1), 0, stream, a, G);
    else DN_LAUNCH(head_fwd_kernel<DN_HEAD_CPL>, dim3(blocks, 1, 1), dim3(256, 1, 1), 0, stream, a, G);
    if (a.labels) DN_LAUNCH(head_finish_kernel, dim3(1, 1, 1), dim3(256, 1, 1), 0, stream, a.partial, blocks, loss, count);
    dn_prof_end(DN_K_SMALL, stream, 0.0, 4.0 * a.n_out * a.C * (a.logp ? 2.0 : 1.0));
    return (int)hipGetLastError();
}

int dn_launch_head_bwd(const HeadArgs& a, hipStream_t stream) {
    if (a.n_src <= 0 || a.C <= 0) return 0;
    if (a.C > 64 * DN_HEAD_CPL) return DN_ERR_BAD_MODE;
    const int G = head_group(a.C), rpb = 4 * (64 / G);
    long long blocks = ((long long)a.n_src + rpb - 1) / rpb;
    if (blocks > 8192) blocks = 8192;
    dn_prof_begin(DN_K_SMALL, stream);
    const int cpl = (a.C + G - 1) / G;
    if (cpl <= 1) DN_LAUNCH(head_bwd_kernel<1>, dim3((unsigned)blocks, 1, 1), dim3(256, 1, 1), 0, stream, a, G);
    else if (cpl <= 2) DN_LAUNCH(head_bwd_kernel<2>, dim3((unsigned)blocks, 1, 1), dim3(256, 1, 1), 0, stream, a, G);
    else if (cpl <= 4) DN_LAUNCH(head_bwd_kernel<4>, dim3((unsigned)blocks, 1, 1), dim3(256, 1, 1), 0, stream, a, G);
    else if (cpl <= 8) DN_LAUNCH(head_bwd_kernel<8>, dim3((unsigned)blocks, 1, 1), dim3(256, 1, 1), 0, stream, a, G);
    else if (cpl <= 16) DN_LAUNCH(head_bwd_kernel<16>, dim3((unsigned)blocks, 1, 1), dim3(256, 1, 1), 0, stream, a, G);
    else DN_LAUNCH(head_bwd_kernel<DN_HEAD_CPL>, dim3((unsigned)blocks, 1, 1), dim3(256, 1, 1), 0, stream, a, G);
    dn_prof_end(DN_K_SMALL, stream, 0.0, 4.0 * a.n_src * a.C * 3.0);
    return (int)hipGetLastError();
}
