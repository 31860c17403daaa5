// dma_rate.hip -- how fast can a CU pull L2-resident data into LDS with global_load_lds_dwordx4?  (round 4: the chained kernels stream
// 0.58 MB of weight pieces per 128 rows through every CU; is ~15 B/clk/CU the ceiling or the kernel's own doing?)
//   hipcc --offload-arch=gfx950 -O3 dma_rate.hip -o dma_rate && ./dma_rate
// Every workgroup (NW waves) copies the same `src_kb` KiB (L2 / MALL resident after the first pass) into a ring of LDS slots, `depth`
// 16 KiB pieces in flight, `reps` times; reports bytes per clock per CU at 1 and 2 workgroups per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define HC(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

__device__ __forceinline__ void dma16(const uint4* gsrc, unsigned lds_byte) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_byte) : "memory");
}

template <int NW, int DEPTH>
__global__ __launch_bounds__(64 * NW) void k(const uint4* src, int npieces, int reps, unsigned long long* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int PIECE = 1024, NTHR = 64 * NW, LPT = PIECE / NTHR, RING = DEPTH + 1;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    int sp = 0, rq = 0;
    auto issue = [&]() {
#pragma unroll
        for (int i = 0; i < LPT; ++i) dma16(src + (size_t)sp * PIECE + i * NTHR + tid, lds0 + 16u * (unsigned)(rq * PIECE + i * NTHR + wave * 64));
        sp = sp + 1 == npieces ? 0 : sp + 1;
        rq = rq + 1 == RING ? 0 : rq + 1;
    };
    for (int i = 0; i < DEPTH; ++i) issue();
    unsigned acc = 0;
    const int total = npieces * reps;
    for (int p = 0; p < total; ++p) {
        issue();
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(DEPTH * LPT) : "memory");
        __builtin_amdgcn_s_barrier();
        acc += reinterpret_cast<const unsigned*>(smem)[((p % RING) * PIECE * 4 + tid) & 0x3fff];   // touch the landed piece
        asm volatile("" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int NW, int DEPTH>
void run(const uint4* src, int npieces, int wg_per_cu, unsigned long long* sink) {
    int cus = 256;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    const size_t smem = (size_t)(DEPTH + 1) * 16384;
    HC(hipFuncSetAttribute((const void*)k<NW, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    const int reps = 40;
    hipEvent_t e0, e1; HC(hipEventCreate(&e0)); HC(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<NW, DEPTH>), dim3(cus * wg_per_cu), dim3(64 * NW), smem, 0, src, npieces, 2, sink);
    HC(hipEventRecord(e0));
    hipLaunchKernelGGL((k<NW, DEPTH>), dim3(cus * wg_per_cu), dim3(64 * NW), smem, 0, src, npieces, reps, sink);
    HC(hipEventRecord(e1)); HC(hipEventSynchronize(e1));
    float ms = 0; HC(hipEventElapsedTime(&ms, e0, e1));
    const double bytes_per_cu = (double)wg_per_cu * npieces * reps * 16384.0;
    printf("NW=%d depth=%d wg/cu=%d LDS=%zu KB: %.1f us, %.1f GB/s per CU, %.1f B/clk/CU at 2.4 GHz, chip %.2f TB/s\n", NW, DEPTH, wg_per_cu, smem / 1024,
           ms * 1e3, bytes_per_cu / (ms * 1e-3) / 1e9, bytes_per_cu / (ms * 1e-3) / 2.4e9, bytes_per_cu * cus / (ms * 1e-3) / 1e12);
}

int main() {
    const int npieces = 36;                      // 576 KB, as one pass of the chained forward
    std::vector<unsigned> h((size_t)npieces * 4096, 1u);
    uint4* src; HC(hipMalloc(&src, h.size() * 4)); HC(hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    unsigned long long* sink; HC(hipMalloc(&sink, 8));
    run<4, 1>(src, npieces, 1, sink); run<4, 1>(src, npieces, 2, sink);
    run<4, 3>(src, npieces, 1, sink); run<4, 3>(src, npieces, 2, sink);
    run<4, 4>(src, npieces, 2, sink);
    run<8, 3>(src, npieces, 1, sink);
    run<1, 3>(src, npieces, 2, sink); run<2, 3>(src, npieces, 2, sink);
    return 0;
}
