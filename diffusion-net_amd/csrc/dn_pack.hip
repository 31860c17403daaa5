// dn_pack.hip -- device-side packing of the sparse operators: COO (as the reference hands them over, utils.py:55 /
// geometry.py:375-382) -> the int32 CSR of the pattern and the CSR of its transpose, values carried along.
// Replaces the per-pack torch chain (bincount + cumsum + stable argsort + three gathers, each a launch or more, two of them
// with a host synchronisation) -- the cost that made the unmodified training loop host-bound.  Integer atomics only (their
// result does not depend on the order); the order inside a transposed row is fixed by a final per-row sort, so the packed
// operator -- and with it every gradient summed through it -- is bitwise reproducible.
#include "dn_common.h"

// rowptr from non-decreasing row ids without a histogram: entry j opens every row in (row[j-1], row[j]]
__global__ __launch_bounds__(256) void pack_rows_kernel(const long long* rows, int row_div, const long long* cols, long long nnz, int n_rows,
                                                        int n_cols, int* rowptr, int* col32, int* t_cnt, int* status) {
    const long long j = (long long)blockIdx.x * 256 + threadIdx.x;
    if (j >= nnz) {
        if (j == nnz && nnz == 0) for (int r = 0; r <= n_rows; ++r) rowptr[r] = 0;
        return;
    }
    const long long r = rows ? rows[j] : j / row_div;
    const long long rp = j == 0 ? -1 : (rows ? rows[j - 1] : (j - 1) / row_div);
    const long long c = cols[j];
    const bool bad = r < 0 || r >= n_rows || c < 0 || c >= n_cols || r < rp;
    if (bad) { atomicOr(status, 1); return; }
    for (long long q = rp + 1; q <= r; ++q) rowptr[q] = (int)j;
    if (j == nnz - 1) for (long long q = r + 1; q <= n_rows; ++q) rowptr[q] = (int)nnz;
    col32[j] = (int)c;
    atomicAdd(&t_cnt[c + 1], 1);
}

// in-place exclusive scan of cnt[0..n] (cnt[0] = 0 on entry, cnt[i+1] = count of i): one workgroup, 1024 lanes, chunked
__global__ __launch_bounds__(1024) void pack_scan_kernel(int* cnt, int n_plus_1) {
    __shared__ int part[1024];
    __shared__ int carry;
    const int tid = threadIdx.x;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < n_plus_1; base += 1024) {
        const int i = base + tid;
        const int v = i < n_plus_1 ? cnt[i] : 0;
        part[tid] = v;
        __syncthreads();
        for (int d = 1; d < 1024; d <<= 1) {            // Hillis-Steele inclusive scan
            const int t = tid >= d ? part[tid - d] : 0;
            __syncthreads();
            part[tid] += t;
            __syncthreads();
        }
        if (i < n_plus_1) cnt[i] = carry + part[tid];   // inclusive over counts shifted by one = exclusive over entries
        __syncthreads();
        if (tid == 1023) carry += part[1023];
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void pack_scatter_kernel(const int* col32, long long nnz, const int* t_rowptr, int* cursor, int* t_src, const int* status) {
    const long long j = (long long)blockIdx.x * 256 + threadIdx.x;
    if (j >= nnz || *status) return;
    const int c = col32[j];
    t_src[t_rowptr[c] + atomicAdd(&cursor[c], 1)] = (int)j;
}

// one thread per transposed row: entries ascending in source order (= ascending row id), then the payload
__global__ __launch_bounds__(256) void pack_finish_kernel(const long long* rows, int row_div, const float* vx, const float* vy, int n_cols,
                                                          const int* t_rowptr, int* t_src, int* t_col, float* t_vx, float* t_vy, const int* status) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= n_cols || *status) return;
    const int beg = t_rowptr[c], end = t_rowptr[c + 1];
    for (int i = beg + 1; i < end; ++i) {               // insertion sort: a vertex has ~7 entries
        const int key = t_src[i];
        int k = i - 1;
        while (k >= beg && t_src[k] > key) { t_src[k + 1] = t_src[k]; --k; }
        t_src[k + 1] = key;
    }
    for (int i = beg; i < end; ++i) {
        const int j = t_src[i];
        t_col[i] = (int)(rows ? rows[j] : j / row_div);
        if (vx) t_vx[i] = vx[j];
        if (vy) t_vy[i] = vy[j];
    }
}

size_t dn_pack_ws_bytes(long long nnz, int n_cols) {
    return (((size_t)n_cols + 1 + (size_t)nnz) * sizeof(int) + 255) & ~(size_t)255;
}

int dn_launch_coo_to_csr(const long long* rows, int row_div, const long long* cols, const float* vx, const float* vy, long long nnz, int n_rows,
                         int n_cols, int* rowptr, int* col32, int* t_rowptr, int* t_col, float* t_vx, float* t_vy, int* status, int* ws,
                         hipStream_t stream) {
    int* cursor = ws;
    int* t_src = ws + n_cols + 1;
    (void)hipMemsetAsync(t_rowptr, 0, ((size_t)n_cols + 1) * sizeof(int), stream);
    (void)hipMemsetAsync(cursor, 0, ((size_t)n_cols + 1) * sizeof(int), stream);
    (void)hipMemsetAsync(status, 0, sizeof(int), stream);
    const unsigned nb = (unsigned)((nnz + 256) / 256);   // one spare thread for the empty pattern
    DN_LAUNCH(pack_rows_kernel, dim3(nb, 1, 1), dim3(256, 1, 1), 0, stream, rows, row_div, cols, nnz, n_rows, n_cols, rowptr, col32, t_rowptr, status);
    DN_LAUNCH(pack_scan_kernel, dim3(1, 1, 1), dim3(1024, 1, 1), 0, stream, t_rowptr, n_cols + 1);
    if (nnz > 0) DN_LAUNCH(pack_scatter_kernel, dim3(nb, 1, 1), dim3(256, 1, 1), 0, stream, col32, nnz, t_rowptr, cursor, t_src, status);
    DN_LAUNCH(pack_finish_kernel, dim3((unsigned)((n_cols + 255) / 256), 1, 1), dim3(256, 1, 1), 0, stream, rows, row_div, vx, vy, n_cols, t_rowptr,
              t_src, t_col, t_vx, t_vy, status);
    return (int)hipGetLastError();
}

// ---- content checksum of an operand (operator cache, diffusion_net/batch.py) -----------------------------------------------------
// 128 bits = two 64-bit lanes; lane k is the wrapping SUM over the 32-bit words w_i of the buffer of mix64((i << 32 | w_i) ^ key_k).
// A sum of per-position hashes does not depend on the order the words are visited in (integer atomics, reproducible), and any
// change of any word -- or of its position -- changes both lanes with probability 1 - 2^-64 each.  `salt` separates the operands
// that are accumulated into the same 128 bits.
__device__ __forceinline__ unsigned long long dn_mix64(unsigned long long z) {
    z ^= z >> 30; z *= 0xbf58476d1ce4e5b9ull;
    z ^= z >> 27; z *= 0x94d049bb133111ebull;
    z ^= z >> 31;
    return z;
}
// several operands in ONE launch (blockIdx.y = operand): a mesh's content key covers eight buffers, and at the sizes of the reference's own
// experiments a launch costs more than the sum itself
struct CkJobs {
    const unsigned* w[DN_CK_MAX_JOBS];
    long long nwords[DN_CK_MAX_JOBS];
    unsigned long long salt[DN_CK_MAX_JOBS];
};
// Few, fat workgroups: every workgroup ends with two 64-bit atomics on the same cache line, and atomics on one line retire at ~8 ns each --
// with ~8 words per thread a 5 MB key was ~1000 workgroups and 55 us, most of it the atomic queue.  At most DN_CK_WGS workgroups per operand,
// 16-byte loads with four in flight (the per-word terms are the same: the sum does not depend on who adds them).
#define DN_CK_WGS 48
__device__ __forceinline__ void ck_word(unsigned long long i, unsigned w, unsigned long long k0, unsigned long long k1, unsigned long long& s0, unsigned long long& s1) {
    const unsigned long long v = (i << 32) ^ (i >> 32) ^ (unsigned long long)w;
    s0 += dn_mix64(v ^ k0);
    s1 += dn_mix64((v + k1) * 0xff51afd7ed558ccdull);
}
__global__ __launch_bounds__(256) void checksum_multi_kernel(CkJobs jobs, unsigned long long* acc) {
    __shared__ unsigned long long red[2][256];
    const int j = blockIdx.y;
    const unsigned* w = jobs.w[j];
    const long long nwords = jobs.nwords[j];
    if ((long long)blockIdx.x * 256 >= nwords) return;      // (uniform per block; nothing for it in either loop form)
    const unsigned long long salt = jobs.salt[j];
    const unsigned long long k0 = dn_mix64(salt + 0x9E3779B97F4A7C15ull), k1 = dn_mix64(salt ^ 0xD1B54A32D192ED03ull);
    unsigned long long s0 = 0, s1 = 0;
    const long long tid = (long long)blockIdx.x * 256 + threadIdx.x, nthr = (long long)gridDim.x * 256;
    if ((((uintptr_t)w) & 15) == 0) {
        const long long n4 = nwords / 4;
        const uint4* w4 = reinterpret_cast<const uint4*>(w);
        for (long long i0 = tid; i0 < n4; i0 += 4 * nthr) {
            uint4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const long long i = i0 + u * nthr; v[u] = w4[i < n4 ? i : i0]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long long i = i0 + u * nthr;
                if (i < n4) {
                    ck_word(4ull * i, v[u].x, k0, k1, s0, s1); ck_word(4ull * i + 1, v[u].y, k0, k1, s0, s1);
                    ck_word(4ull * i + 2, v[u].z, k0, k1, s0, s1); ck_word(4ull * i + 3, v[u].w, k0, k1, s0, s1);
                }
            }
        }
        for (long long i = 4 * n4 + tid; i < nwords; i += nthr) ck_word((unsigned long long)i, w[i], k0, k1, s0, s1);
    } else {
        for (long long i = tid; i < nwords; i += nthr) ck_word((unsigned long long)i, w[i], k0, k1, s0, s1);
    }
    red[0][threadIdx.x] = s0; red[1][threadIdx.x] = s1;
    __syncthreads();
    for (int d = 128; d > 0; d >>= 1) {
        if ((int)threadIdx.x < d) { red[0][threadIdx.x] += red[0][threadIdx.x + d]; red[1][threadIdx.x] += red[1][threadIdx.x + d]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { atomicAdd(&acc[0], red[0][0]); atomicAdd(&acc[1], red[1][0]); }
}
int dn_launch_checksum_multi(int n, const void* const* data, const long long* nwords, const unsigned long long* salts, unsigned long long* acc,
                             hipStream_t stream) {
    CkJobs jobs;
    long long most = 0;
    int m = 0;
    for (int j = 0; j < n; ++j) {
        if (nwords[j] <= 0) continue;
        jobs.w[m] = (const unsigned*)data[j]; jobs.nwords[m] = nwords[j]; jobs.salt[m] = salts[j];
        most = nwords[j] > most ? nwords[j] : most;
        ++m;
    }
    if (m == 0) return 0;
    long long nb = (most + 256 * 16 - 1) / (256 * 16);     // >= 16 words per thread of the largest operand, few workgroups (see the kernel)
    if (nb > DN_CK_WGS) nb = DN_CK_WGS;
    DN_LAUNCH(checksum_multi_kernel, dim3((unsigned)nb, (unsigned)m, 1), dim3(256, 1, 1), 0, stream, jobs, acc);
    return (int)hipGetLastError();
}
int dn_launch_checksum(const void* data, long long nwords, unsigned long long salt, unsigned long long* acc, hipStream_t stream) {
    const void* d[1] = {data};
    return dn_launch_checksum_multi(1, d, &nwords, &salt, acc, stream);     // (one operand of the same kernel: the same per-word terms)
}
