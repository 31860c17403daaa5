// Do vector-memory loads (here: LDS-DMA requests and plain loads that miss to HBM) and YOUNGER stores (to lines hot in L2) retire from
// vmcnt in issue order on gfx950?  The counted waits of the chained kernels / backproject_wide_kernel that leave stores in flight
// ("s_waitcnt vmcnt(n_stores)" = everything older than the stores has landed) are only right if they do.
//   hipcc --offload-arch=gfx950 -O2 vmcnt_order.hip -o vmcnt_order && ./vmcnt_order
// Every wave: one cold 16-byte-per-lane request (a different 1 KiB of a 2 GiB buffer each time), then NS stores to its own hot line,
// then s_waitcnt vmcnt(NS), then it checks that the cold data is there.  A stale read = the stores' acknowledgements let the counter pass
// the load.  Prints the number of stale reads (expected 0) and, as a control, with vmcnt(NS + 1) (must be > 0: the wait is what protects).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define HC(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int NS = 16;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int SLACK>   // MODE 0: LDS-DMA request, 1: plain global load.  SLACK 0: vmcnt(NS); 1: vmcnt(NS + 1) (control)
__global__ __launch_bounds__(64) void probe(const uint4* src, size_t n_kib, uint4* hot, unsigned long long* stale, int iters) {
    __shared__ uint4 lds[64];
    const int lane = threadIdx.x;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
    unsigned long long bad = 0;
    size_t pos = ((size_t)blockIdx.x * 2654435761ull) % n_kib;
    uint4* mine = hot + (size_t)blockIdx.x * 64 * NS + lane;
    for (int it = 0; it < iters; ++it) {
        pos = (pos * 6364136223846793005ull + 1442695040888963407ull) % n_kib;
        const uint4* p = src + pos * 64 + lane;
        lds[lane] = make_uint4(0, 0, 0, 0);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        u32x4 v = {0, 0, 0, 0};
        if (MODE == 0) {
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(p), "s"(lds0) : "memory");
        } else {
            asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(v) : "v"(p) : "memory");
        }
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const u32x4 w = {(unsigned)it, (unsigned)s, (unsigned)lane, 7u};
            asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(mine + 64 * s), "v"(w) : "memory");
        }
        if (SLACK == 0) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NS) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NS + 1) : "memory");
        if (MODE == 0) { const uint4 t = lds[lane]; v.x = t.x; }
        else asm volatile("v_mov_b32 %0, %0" : "+v"(v.x) :: "memory");     // (read the destination register as it is now)
        const unsigned want = (unsigned)((pos * 64 + lane) * 2654435761ull);
        if (v.x != want) ++bad;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    if (bad) atomicAdd(stale, bad);
}
__global__ void fill(uint4* src, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        src[i] = make_uint4((unsigned)(i * 2654435761ull), 1, 2, 3);
}
int main() {
    const size_t n_kib = (size_t)2 << 20;      // 2 Mi x 1 KiB = 2 GiB
    uint4 *src, *hot; unsigned long long* stale;
    HC(hipMalloc(&src, n_kib * 1024)); HC(hipMalloc(&hot, (size_t)2048 * 64 * NS * 16)); HC(hipMalloc(&stale, 8));
    fill<<<4096, 256>>>(src, n_kib * 64); HC(hipDeviceSynchronize());
    auto run = [&](auto kern, const char* name) {
        HC(hipMemset(stale, 0, 8));
        kern<<<2048, 64>>>(src, n_kib, hot, stale, 4000);
        HC(hipDeviceSynchronize());
        unsigned long long h; HC(hipMemcpy(&h, stale, 8, hipMemcpyDeviceToHost));
        printf("%-58s stale reads %llu of %llu\n", name, h, 2048ull * 64 * 4000);
    };
    run(probe<0, 0>, "LDS-DMA request, 16 younger stores, vmcnt(16):");
    run(probe<1, 0>, "plain load,      16 younger stores, vmcnt(16):");
    run(probe<0, 1>, "control: LDS-DMA request, vmcnt(17) (no wait):");
    run(probe<1, 1>, "control: plain load,      vmcnt(17) (no wait):");
    return 0;
}
