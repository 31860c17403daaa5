// tile_io_rate.hip -- do row-tile loads / stores in the MFMA operand layout (lane (m = l & 15, q = l >> 4): row m, bytes 64 nt + 16 q, i.e. 16 rows
// x 64 B per instruction) run slower than row-contiguous ones (2 rows x 512 B per instruction) when the rows ARE consecutive?  (round 4: every
// tile the chained kernels read or write is in the operand layout.)   hipcc --offload-arch=gfx950 -O3 tile_io_rate.hip -o tile_io_rate
// Each wave walks 16-row tiles of NT tensors [V,128] fp32; mode: 0 = store A, 1 = store B, 2 = load A, 3 = load B (loads are summed into a sink).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define HC(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)
constexpr int C = 128;
template <int MODE, int NTEN>
__global__ __launch_bounds__(256) void k(float* base, size_t tstride, int ngroups, float* sink) {
    const int l = threadIdx.x & 63, m = l & 15, q = l >> 4, wave = threadIdx.x >> 6, half = l >> 5, c4 = l & 31;
    const int gx = gridDim.x >> 3, per_x = (ngroups + 7) >> 3;
    float acc = 0.f;
    for (int it = 0;; ++it) {
        const int lg = ((blockIdx.x >> 3) + it * gx) * 4 + wave;
        if (lg >= per_x) break;
        const int g = (blockIdx.x & 7) * per_x + lg;
        if (g >= ngroups) break;
#pragma unroll
        for (int t = 0; t < NTEN; ++t) {
            float* p = base + t * tstride + (size_t)g * 16 * C;
            if (MODE == 0) {
#pragma unroll
                for (int nt = 0; nt < 8; ++nt) *reinterpret_cast<float4*>(p + m * C + 16 * nt + 4 * q) = make_float4(1.f + g, 2.f, 3.f, 4.f + nt);
            } else if (MODE == 1) {
#pragma unroll
                for (int i = 0; i < 8; ++i) *reinterpret_cast<float4*>(p + (2 * i + half) * C + 4 * c4) = make_float4(1.f + g, 2.f, 3.f, 4.f + i);
            } else if (MODE == 2) {
                float4 v[8];
#pragma unroll
                for (int nt = 0; nt < 8; ++nt) v[nt] = *reinterpret_cast<const float4*>(p + m * C + 16 * nt + 4 * q);
#pragma unroll
                for (int nt = 0; nt < 8; ++nt) acc += v[nt].x + v[nt].w;
            } else {
                float4 v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = *reinterpret_cast<const float4*>(p + (2 * i + half) * C + 4 * c4);
#pragma unroll
                for (int i = 0; i < 8; ++i) acc += v[i].x + v[i].w;
            }
        }
    }
    if (acc == 1.2345f) sink[0] = acc;
}
template <int MODE, int NTEN>
void run(const char* name, float* base, size_t tstride, int V, int wg_per_cu, float* sink) {
    int cus = 256; hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    hipEvent_t e0, e1; HC(hipEventCreate(&e0)); HC(hipEventCreate(&e1));
    const int reps = 5;
    hipLaunchKernelGGL((k<MODE, NTEN>), dim3(cus * wg_per_cu), dim3(256), 0, 0, base, tstride, V / 16, sink);
    HC(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k<MODE, NTEN>), dim3(cus * wg_per_cu), dim3(256), 0, 0, base, tstride, V / 16, sink);
    HC(hipEventRecord(e1)); HC(hipEventSynchronize(e1)); HC(hipGetLastError());
    float ms = 0; HC(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    const double bytes = (double)NTEN * V * C * 4;
    printf("%-28s %d tensors wg/cu=%d: %7.1f us  %.2f TB/s  %.1f B/clk/CU\n", name, NTEN, wg_per_cu, ms * 1e3, bytes / (ms * 1e-3) / 1e12, bytes / cus / (ms * 1e-3) / 2.4e9);
}
int main() {
    const int V = 160000; const size_t ts = (size_t)V * C;
    float *base, *sink; HC(hipMalloc(&base, 8 * ts * 4)); HC(hipMemset(base, 0, 8 * ts * 4)); HC(hipMalloc(&sink, 4));
    for (int wg = 1; wg <= 2; ++wg) {
        run<0, 1>("store, operand layout", base, ts, V, wg, sink); run<1, 1>("store, row-contiguous", base, ts, V, wg, sink);
        run<0, 6>("store, operand layout", base, ts, V, wg, sink); run<1, 6>("store, row-contiguous", base, ts, V, wg, sink);
        run<2, 1>("load, operand layout", base, ts, V, wg, sink);  run<3, 1>("load, row-contiguous", base, ts, V, wg, sink);
        run<2, 6>("load, operand layout", base, ts, V, wg, sink);  run<3, 6>("load, row-contiguous", base, ts, V, wg, sink);
    }
    return 0;
}
