// exp_bf16x3.hip -- standalone experiment (not part of the library): C[M,N] = A[M,K] * B[N,K]^T with fp32 inputs/outputs,
// computed (1) with exact-f32 MFMA (v_mfma_f32_32x32x2_f32) and (2) with a 3-term bf16 split of both operands and the six
// largest cross products on v_mfma_f32_32x32x16_bf16.  Reports time and error against an fp64 host reference.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 exp_bf16x3.hip -o exp_bf16x3 && ./exp_bf16x3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <cstdint>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define TM 128
#define TN 128
#define KB 32

__device__ __forceinline__ uint16_t bf16_rne(float x, float& back) {
    uint32_t u = __float_as_uint(x);
    u += 0x7fffu + ((u >> 16) & 1u);
    const uint16_t h = (uint16_t)(u >> 16);
    back = __uint_as_float((uint32_t)h << 16);
    return h;
}
__device__ __forceinline__ int acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// ---------------- f32 MFMA baseline: both operands k-contiguous, swizzled COLK tiles, single-buffered ----------------
__device__ __forceinline__ int colk_off(int row, int slot) { return row * 32 + ((slot ^ ((row >> 1) & 7)) << 2); }

__global__ __launch_bounds__(256) void gemm_f32(const float* A, const float* B, float* C, int M, int N, int K) {
    __shared__ float sA[TM * KB];
    __shared__ float sB[TN * KB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 1, wc = wave & 1, li = lane & 31, ls = lane >> 5;
    const long long r0 = (long long)blockIdx.x * TM;
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int k0 = 0; k0 < K; k0 += KB) {
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + i * 256, row = idx >> 3, q = idx & 7;
            float4 va = make_float4(0, 0, 0, 0);
            if (r0 + row < M) va = *reinterpret_cast<const float4*>(A + (r0 + row) * K + k0 + 4 * q);
            *reinterpret_cast<float4*>(&sA[colk_off(row, q)]) = va;
            *reinterpret_cast<float4*>(&sB[colk_off(row, q)]) = *reinterpret_cast<const float4*>(B + (long long)row * K + k0 + 4 * q);
        }
        __syncthreads();
        for (int kg = 0; kg < 4; ++kg) {
            float4 af[2], bf[2];
            for (int i = 0; i < 2; ++i) {
                af[i] = *reinterpret_cast<const float4*>(&sA[colk_off((wr * 2 + i) * 32 + li, 2 * kg + ls)]);
                bf[i] = *reinterpret_cast<const float4*>(&sB[colk_off((wc * 2 + i) * 32 + li, 2 * kg + ls)]);
            }
            const float* ap0 = reinterpret_cast<const float*>(&af[0]); const float* ap1 = reinterpret_cast<const float*>(&af[1]);
            const float* bp0 = reinterpret_cast<const float*>(&bf[0]); const float* bp1 = reinterpret_cast<const float*>(&bf[1]);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ap0[t], bp0[t], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ap0[t], bp1[t], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ap1[t], bp0[t], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ap1[t], bp1[t], acc[1][1], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) {
        const long long row = r0 + (wr * 2 + i) * 32 + acc_row(r, lane);
        if (row < M) C[row * N + (wc * 2 + j) * 32 + li] = acc[i][j][r];
    }
}

// ---------------- bf16 x 3 ----------------
// LDS plane: [128 rows][32 k] bf16 = 64 B per row = 4 slots of 16 B; slot' = slot ^ ((row>>2)&3) -> conflict-free b128 reads.
__device__ __forceinline__ int plane_off_bytes(int row, int slot) { return row * 64 + ((slot ^ ((row >> 2) & 3)) << 4); }

__device__ __forceinline__ void split4(float4 v, uint16_t (&hi)[4], uint16_t (&mid)[4], uint16_t (&lo)[4]) {
    const float x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float b0, b1, b2;
        hi[e] = bf16_rne(x[e], b0);
        const float r1 = x[e] - b0;
        mid[e] = bf16_rne(r1, b1);
        const float r2 = r1 - b1;
        lo[e] = bf16_rne(r2, b2);
    }
}

template <int NTERMS>
__global__ __launch_bounds__(256) void gemm_bf16x3(const float* A, const float* B, float* C, int M, int N, int K) {
    __shared__ __attribute__((aligned(16))) unsigned char sA[3][TM * 64];
    __shared__ __attribute__((aligned(16))) unsigned char sB[3][TN * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 1, wc = wave & 1, li = lane & 31, lg = lane >> 5;
    const long long r0 = (long long)blockIdx.x * TM;
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int k0 = 0; k0 < K; k0 += KB) {
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + i * 256, row = idx >> 3, q = idx & 7;      // q: group of 4 k (8 bytes of bf16)
            float4 va = make_float4(0, 0, 0, 0);
            if (r0 + row < M) va = *reinterpret_cast<const float4*>(A + (r0 + row) * K + k0 + 4 * q);
            const float4 vb = *reinterpret_cast<const float4*>(B + (long long)row * K + k0 + 4 * q);
            uint16_t h[4], m[4], l[4];
            const int off = plane_off_bytes(row, q >> 1) + (q & 1) * 8;
            split4(va, h, m, l);
            *reinterpret_cast<uint2*>(&sA[0][off]) = make_uint2(h[0] | (h[1] << 16), h[2] | (h[3] << 16));
            *reinterpret_cast<uint2*>(&sA[1][off]) = make_uint2(m[0] | (m[1] << 16), m[2] | (m[3] << 16));
            *reinterpret_cast<uint2*>(&sA[2][off]) = make_uint2(l[0] | (l[1] << 16), l[2] | (l[3] << 16));
            split4(vb, h, m, l);
            *reinterpret_cast<uint2*>(&sB[0][off]) = make_uint2(h[0] | (h[1] << 16), h[2] | (h[3] << 16));
            *reinterpret_cast<uint2*>(&sB[1][off]) = make_uint2(m[0] | (m[1] << 16), m[2] | (m[3] << 16));
            *reinterpret_cast<uint2*>(&sB[2][off]) = make_uint2(l[0] | (l[1] << 16), l[2] | (l[3] << 16));
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 2; ++s) {          // two k16 steps per 32-wide slice
            bf16x8 a[3][2], b[3][2];
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    a[p][i] = *reinterpret_cast<const bf16x8*>(&sA[p][plane_off_bytes((wr * 2 + i) * 32 + li, s * 2 + lg)]);
                    b[p][i] = *reinterpret_cast<const bf16x8*>(&sB[p][plane_off_bytes((wc * 2 + i) * 32 + li, s * 2 + lg)]);
                }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    f32x16 c = acc[i][j];
                    // smallest terms first
                    if (NTERMS >= 6) {
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][i], b[1][j], c, 0, 0, 0);   // mid*mid
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[2][j], c, 0, 0, 0);   // hi*lo
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2][i], b[0][j], c, 0, 0, 0);   // lo*hi
                    }
                    if (NTERMS >= 3) {
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[1][j], c, 0, 0, 0);   // hi*mid
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][i], b[0][j], c, 0, 0, 0);   // mid*hi
                    }
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[0][j], c, 0, 0, 0);       // hi*hi
                    acc[i][j] = c;
                }
        }
        __syncthreads();
    }
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) {
        const long long row = r0 + (wr * 2 + i) * 32 + acc_row(r, lane);
        if (row < M) C[row * N + (wc * 2 + j) * 32 + li] = acc[i][j][r];
    }
}

static double check(const std::vector<float>& A, const std::vector<float>& B, const std::vector<float>& C, int M, int N, int K) {
    double num = 0, den = 0, maxabs = 0, maxref = 0;
    for (int s = 0; s < 64; ++s) {
        const long long r = (long long)s * (M / 64) + (s % 7);
        for (int n = 0; n < N; ++n) {
            double ref = 0;
            for (int k = 0; k < K; ++k) ref += (double)A[r * K + k] * (double)B[(long long)n * K + k];
            const double d = C[r * N + n] - ref;
            num += d * d; den += ref * ref;
            if (fabs(d) > maxabs) maxabs = fabs(d);
            if (fabs(ref) > maxref) maxref = fabs(ref);
        }
    }
    printf("   rel-L2 %.3e   max-abs/max-ref %.3e\n", sqrt(num / den), maxabs / maxref);
    return sqrt(num / den);
}

int main() {
    const int M = 158481, N = 128, K = 128;
    std::vector<float> hA((size_t)M * K), hB((size_t)N * K), hC((size_t)M * N);
    srand(1);
    for (auto& v : hA) v = (float)rand() / RAND_MAX * 2.f - 1.f;
    for (auto& v : hB) v = ((float)rand() / RAND_MAX * 2.f - 1.f) * 0.1f;
    float *dA, *dB, *dC;
    hipMalloc(&dA, hA.size() * 4); hipMalloc(&dB, hB.size() * 4); hipMalloc(&dC, hC.size() * 4);
    hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
    const int nblk = (M + TM - 1) / TM;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char* name, auto launch) {
        for (int i = 0; i < 3; ++i) launch();
        hipEventRecord(e0);
        for (int i = 0; i < 20; ++i) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-28s %8.1f us  (%.1f effective fp32 TFLOP/s)\n", name, ms / 20 * 1e3, 2.0 * M * N * K / (ms / 20 * 1e-3) / 1e12);
        hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost);
        check(hA, hB, hC, M, N, K);
    };
    run("f32 MFMA 32x32x2", [&] { hipLaunchKernelGGL(gemm_f32, dim3(nblk), dim3(256), 0, 0, dA, dB, dC, M, N, K); });
    run("bf16 x1 (hi*hi)", [&] { hipLaunchKernelGGL(gemm_bf16x3<1>, dim3(nblk), dim3(256), 0, 0, dA, dB, dC, M, N, K); });
    run("bf16 x2 (3 terms)", [&] { hipLaunchKernelGGL(gemm_bf16x3<3>, dim3(nblk), dim3(256), 0, 0, dA, dB, dC, M, N, K); });
    run("bf16 x3 (6 terms)", [&] { hipLaunchKernelGGL(gemm_bf16x3<6>, dim3(nblk), dim3(256), 0, 0, dA, dB, dC, M, N, K); });
    printf("hipGetLastError: %d\n", (int)hipGetLastError());
    return 0;
}
