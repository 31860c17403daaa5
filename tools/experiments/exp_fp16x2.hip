// exp_fp16x2.hip -- standalone experiment (not part of the library), round-2 candidate: C[M,N] = A[M,K] * B[N,K]^T with
// fp32 inputs/outputs computed with a TWO-term fp16 split of both operands (hi + lo, 22 mantissa bits) and the three largest
// cross products on v_mfma_f32_32x32x16_f16 -- half the MFMAs and two thirds of the LDS planes of the shipped 3-term bf16
// split.  fp16 has 5 exponent bits, so the block-scaled variant multiplies every (row, 32-wide k block) of A and every
// (column, k block) of B by a power of two that puts its maximum at 2^14, accumulates the block in its own accumulator and
// rescales (exactly) when adding it to the running sum.  Reports time and error against an fp64 host reference, next to
// the bf16 x3 numbers of exp_bf16x3.hip.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 exp_fp16x2.hip -o exp_fp16x2 && ./exp_fp16x2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <cstdint>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define TM 128
#define TN 128
#define KB 32

__device__ __forceinline__ int acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }
__device__ __forceinline__ int plane_off_bytes(int row, int slot) { return row * 64 + ((slot ^ ((row >> 2) & 3)) << 4); }

// (x, y) -> packed fp16 hi and lo words (RNE via the hardware convert); hi + lo == value up to 2^-22 relative
__device__ __forceinline__ void split2_pair(float x, float y, unsigned& hi, unsigned& lo) {
    const f16x2 h = __builtin_convertvector(f32x2{x, y}, f16x2);
    hi = __builtin_bit_cast(unsigned, h);
    const f32x2 hb = __builtin_convertvector(h, f32x2);
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{x - hb.x, y - hb.y}, f16x2));
}
// power of two that maps absmax into [2^14, 2^15)  (1 for an all-zero block)
__device__ __forceinline__ float pow2_scale(float absmax) {
    const unsigned e = (__float_as_uint(absmax) >> 23) & 0xffu;          // biased exponent of the max
    int f = 127 + 14 + 127 - (int)e;                                      // exponent field of 2^(14 - unbiased e)
    f = f > 254 ? 254 : f;
    return e == 0 ? 1.f : __uint_as_float((unsigned)f << 23);
}

template <bool SCALED>
__global__ __launch_bounds__(256) void gemm_fp16x2(const float* A, const float* B, float* C, int M, int N, int K) {
    __shared__ __attribute__((aligned(16))) unsigned char sA[2][TM * 64];
    __shared__ __attribute__((aligned(16))) unsigned char sB[2][TN * 64];
    __shared__ float sIA[TM], sIB[TN];                                    // inverse block scales
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 1, wc = wave & 1, li = lane & 31, lg = lane >> 5;
    const long long r0 = (long long)blockIdx.x * TM;
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int k0 = 0; k0 < K; k0 += KB) {
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + i * 256, row = idx >> 3, q = idx & 7;      // 8 consecutive lanes hold one row's 32 k
            float4 va = make_float4(0, 0, 0, 0);
            if (r0 + row < M) va = *reinterpret_cast<const float4*>(A + (r0 + row) * K + k0 + 4 * q);
            float4 vb = *reinterpret_cast<const float4*>(B + (long long)row * K + k0 + 4 * q);
            if (SCALED) {
                float ma = fmaxf(fmaxf(fabsf(va.x), fabsf(va.y)), fmaxf(fabsf(va.z), fabsf(va.w)));
                float mb = fmaxf(fmaxf(fabsf(vb.x), fabsf(vb.y)), fmaxf(fabsf(vb.z), fabsf(vb.w)));
#pragma unroll
                for (int d = 1; d < 8; d <<= 1) { ma = fmaxf(ma, __shfl_xor(ma, d)); mb = fmaxf(mb, __shfl_xor(mb, d)); }
                const float sa = pow2_scale(ma), sb = pow2_scale(mb);
                va.x *= sa; va.y *= sa; va.z *= sa; va.w *= sa;
                vb.x *= sb; vb.y *= sb; vb.z *= sb; vb.w *= sb;
                if (q == 0) { sIA[row] = 1.f / sa; sIB[row] = 1.f / sb; }
            }
            const int off = plane_off_bytes(row, q >> 1) + (q & 1) * 8;
            unsigned h0, l0, h1, l1;
            split2_pair(va.x, va.y, h0, l0); split2_pair(va.z, va.w, h1, l1);
            *reinterpret_cast<uint2*>(&sA[0][off]) = make_uint2(h0, h1);
            *reinterpret_cast<uint2*>(&sA[1][off]) = make_uint2(l0, l1);
            split2_pair(vb.x, vb.y, h0, l0); split2_pair(vb.z, vb.w, h1, l1);
            *reinterpret_cast<uint2*>(&sB[0][off]) = make_uint2(h0, h1);
            *reinterpret_cast<uint2*>(&sB[1][off]) = make_uint2(l0, l1);
        }
        __syncthreads();
        f32x16 blk[2][2];
        if (SCALED) for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) blk[i][j][r] = 0.f;
#pragma unroll
        for (int s = 0; s < 2; ++s) {          // two k16 steps per 32-wide slice
            f16x8 a[2][2], b[2][2];
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    a[p][i] = *reinterpret_cast<const f16x8*>(&sA[p][plane_off_bytes((wr * 2 + i) * 32 + li, s * 2 + lg)]);
                    b[p][i] = *reinterpret_cast<const f16x8*>(&sB[p][plane_off_bytes((wc * 2 + i) * 32 + li, s * 2 + lg)]);
                }
            constexpr int PA[3] = {0, 1, 0}, PB[3] = {1, 0, 0};   // hi*lo, lo*hi, hi*hi -- product-major issue order
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        if (SCALED) blk[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[PA[p]][i], b[PB[p]][j], blk[i][j], 0, 0, 0);
                        else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[PA[p]][i], b[PB[p]][j], acc[i][j], 0, 0, 0);
                    }
        }
        if (SCALED) {                          // exact rescale (powers of two) of the block's sum into the running sum
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const float ib = sIB[(wc * 2 + j) * 32 + li];
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = fmaf(blk[i][j][r], sIA[(wr * 2 + i) * 32 + acc_row(r, lane)] * ib, acc[i][j][r]);
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
    float *dA, *dB, *dC;
    hipMalloc(&dA, hA.size() * 4); hipMalloc(&dB, hB.size() * 4); hipMalloc(&dC, hC.size() * 4);
    const int nblk = (M + TM - 1) / TM;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char* name, auto launch) {
        for (int i = 0; i < 3; ++i) launch();
        hipEventRecord(e0);
        for (int i = 0; i < 20; ++i) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-34s %8.1f us  (%.1f effective fp32 TFLOP/s)\n", name, ms / 20 * 1e3, 2.0 * M * N * K / (ms / 20 * 1e-3) / 1e12);
        hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost);
        check(hA, hB, hC, M, N, K);
    };
    for (int data = 0; data < 2; ++data) {
        srand(1);
        if (data == 0) {   // same data as exp_bf16x3: U(-1,1) x 0.1 U(-1,1)
            for (auto& v : hA) v = (float)rand() / RAND_MAX * 2.f - 1.f;
            for (auto& v : hB) v = ((float)rand() / RAND_MAX * 2.f - 1.f) * 0.1f;
            printf("== data: U(-1,1) x 0.1 U(-1,1)\n");
        } else {           // gradient-like: every row of A scaled by 10^U(-7,0), B large
            for (int r = 0; r < M; ++r) {
                const float s = powf(10.f, -7.f * (float)rand() / RAND_MAX);
                for (int k = 0; k < K; ++k) hA[(size_t)r * K + k] = ((float)rand() / RAND_MAX * 2.f - 1.f) * s;
            }
            for (auto& v : hB) v = ((float)rand() / RAND_MAX * 2.f - 1.f) * 300.f;
            printf("== data: rows of A scaled by 10^U(-7,0), B ~ 300 U(-1,1)\n");
        }
        hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
        run("fp16 x2 (3 terms), unscaled", [&] { hipLaunchKernelGGL(gemm_fp16x2<false>, dim3(nblk), dim3(256), 0, 0, dA, dB, dC, M, N, K); });
        run("fp16 x2 (3 terms), block-scaled", [&] { hipLaunchKernelGGL(gemm_fp16x2<true>, dim3(nblk), dim3(256), 0, 0, dA, dB, dC, M, N, K); });
    }
    printf("hipGetLastError: %d\n", (int)hipGetLastError());
    return 0;
}
