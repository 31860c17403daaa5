// gather_rate.hip -- how fast can a CU gather neighbour rows (512 B fp32 rows of a [V,128] matrix, 7 per row, torus pattern like the
// benchmark meshes) and does the lane -> address mapping matter?  (round 4: the chained forward spends 37 % of a pass in its gather.)
//   hipcc --offload-arch=gfx950 -O3 gather_rate.hip -o gather_rate && ./gather_rate
// Variants (all: one wave = 16 rows at a time, 2 workgroups of 4 waves per CU, persistent over 16-row groups, sums gx only):
//   A  : the chain's mapping: lane (m = l & 15, q = l >> 4) loads 16 B at row col[m][j], byte 64 nt + 16 q  (16 rows x 64 B per instruction)
//   B  : row-contiguous mapping: lanes 0-31 / 32-63 each read one whole 512 B row per instruction (2 rows x 512 B per instruction)
//   AD : mapping A through LDS-DMA (global_load_lds_dwordx4) into a wave-private LDS buffer, then ds_read + fma
// ENT = entries in flight per lane before the first use.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define HC(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)
constexpr int C = 128, NT = 8, NE = 8;      // NE entries per row (7 real + 1 padding that points at the row itself)

template <int ENT>
__global__ __launch_bounds__(256) void gather_a(const float* __restrict__ xd, const int* __restrict__ col, const float* __restrict__ w, float* __restrict__ out, int ngroups, int ngroups_x) {
    const int l = threadIdx.x & 63, m = l & 15, q = l >> 4, wave = threadIdx.x >> 6;
    for (int it = 0;; ++it) {
        int g;
        if (ngroups_x) { const int xc = blockIdx.x & 7, sl = blockIdx.x >> 3, gx = gridDim.x >> 3; const int lg = (sl + it * gx) * 4 + wave; if (lg >= ngroups_x) break; g = xc * ngroups_x + lg; if (g >= ngroups) break; }
        else { g = (blockIdx.x + it * gridDim.x) * 4 + wave; if (g >= ngroups) break; }
        const int row = g * 16 + m;
        float acc[NT][4] = {};
        for (int j0 = 0; j0 < NE; j0 += ENT) {
            float4 v[ENT][NT]; float ww[ENT];
#pragma unroll
            for (int u = 0; u < ENT; ++u) {
                const int c = col[row * NE + j0 + u]; ww[u] = w[row * NE + j0 + u];
                const float* src = xd + (size_t)c * C + 4 * q;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) v[u][nt] = *reinterpret_cast<const float4*>(src + 16 * nt);
            }
#pragma unroll
            for (int u = 0; u < ENT; ++u)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    acc[nt][0] = fmaf(ww[u], v[u][nt].x, acc[nt][0]); acc[nt][1] = fmaf(ww[u], v[u][nt].y, acc[nt][1]);
                    acc[nt][2] = fmaf(ww[u], v[u][nt].z, acc[nt][2]); acc[nt][3] = fmaf(ww[u], v[u][nt].w, acc[nt][3]);
                }
        }
        float* o = out + (size_t)row * C + 4 * q;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) *reinterpret_cast<float4*>(o + 16 * nt) = make_float4(acc[nt][0], acc[nt][1], acc[nt][2], acc[nt][3]);
    }
}


// A2: the chain's register layout (lane (m,q) ends up with row m, bytes 64 nt + 16 q) from FULL 128-byte lines: instruction I0 reads rows 0-7 of
// the group (lane (m,q): row m & 7, 16-byte part q + 4 (m >> 3) of the line), I1 rows 8-15 mirrored; each lane keeps the part that is its own
// (v_cndmask) and receives the other from lane m ^ 8 (DPP row_ror:8).  8 rows x 128 B per instruction instead of 16 rows x 64 B.
template <int ENT>
__global__ __launch_bounds__(256) void gather_a2(const float* __restrict__ xd, const int* __restrict__ col, const float* __restrict__ w, float* __restrict__ out, int ngroups, int ngroups_x) {
    const int l = threadIdx.x & 63, m = l & 15, q = l >> 4, wave = threadIdx.x >> 6;
    const bool lo = m < 8;
    for (int it = 0;; ++it) {
        int g;
        if (ngroups_x) { const int xc = blockIdx.x & 7, sl = blockIdx.x >> 3, gx = gridDim.x >> 3; const int lg = (sl + it * gx) * 4 + wave; if (lg >= ngroups_x) break; g = xc * ngroups_x + lg; if (g >= ngroups) break; }
        else { g = (blockIdx.x + it * gridDim.x) * 4 + wave; if (g >= ngroups) break; }
        const int row = g * 16 + m;
        const int row0 = g * 16 + (m & 7), row1 = row0 + 8;      // the rows this lane loads for in I0 / I1
        const int part0 = 4 * (q + (lo ? 0 : 4)), part1 = 4 * (q + (lo ? 4 : 0));   // float offsets inside the 32-float line
        float acc[NT][4] = {};
        for (int j0 = 0; j0 < NE; j0 += ENT) {
            float4 v0[ENT][NT / 2], v1[ENT][NT / 2]; float ww[ENT];
#pragma unroll
            for (int u = 0; u < ENT; ++u) {
                const int c0 = col[row0 * NE + j0 + u], c1 = col[row1 * NE + j0 + u]; ww[u] = w[row * NE + j0 + u];
                const float* s0 = xd + (size_t)c0 * C + part0;
                const float* s1 = xd + (size_t)c1 * C + part1;
#pragma unroll
                for (int L = 0; L < NT / 2; ++L) { v0[u][L] = *reinterpret_cast<const float4*>(s0 + 32 * L); v1[u][L] = *reinterpret_cast<const float4*>(s1 + 32 * L); }
            }
#pragma unroll
            for (int u = 0; u < ENT; ++u)
#pragma unroll
                for (int L = 0; L < NT / 2; ++L) {
                    const float own[4] = {lo ? v0[u][L].x : v1[u][L].x, lo ? v0[u][L].y : v1[u][L].y, lo ? v0[u][L].z : v1[u][L].z, lo ? v0[u][L].w : v1[u][L].w};
                    const float oth[4] = {lo ? v1[u][L].x : v0[u][L].x, lo ? v1[u][L].y : v0[u][L].y, lo ? v1[u][L].z : v0[u][L].z, lo ? v1[u][L].w : v0[u][L].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        acc[2 * L][e] = fmaf(ww[u], own[e], acc[2 * L][e]);
                        const float f = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, oth[e]), 0x128, 0xf, 0xf, false));
                        acc[2 * L + 1][e] = fmaf(ww[u], f, acc[2 * L + 1][e]);
                    }
                }
        }
        float* o = out + (size_t)row * C + 4 * q;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) *reinterpret_cast<float4*>(o + 16 * nt) = make_float4(acc[nt][0], acc[nt][1], acc[nt][2], acc[nt][3]);
    }
}

template <int ROWS_IN_FLIGHT>      // row pairs in flight (each: NE entries x one float4 per lane)
__global__ __launch_bounds__(256) void gather_b(const float* __restrict__ xd, const int* __restrict__ col, const float* __restrict__ w, float* __restrict__ out, int ngroups, int ngroups_x) {
    const int l = threadIdx.x & 63, half = l >> 5, c4 = l & 31, wave = threadIdx.x >> 6;
    for (int it = 0;; ++it) {
        int g;
        if (ngroups_x) { const int xc = blockIdx.x & 7, sl = blockIdx.x >> 3, gx = gridDim.x >> 3; const int lg = (sl + it * gx) * 4 + wave; if (lg >= ngroups_x) break; g = xc * ngroups_x + lg; if (g >= ngroups) break; }
        else { g = (blockIdx.x + it * gridDim.x) * 4 + wave; if (g >= ngroups) break; }
        for (int p0 = 0; p0 < 8; p0 += ROWS_IN_FLIGHT) {
            float4 v[ROWS_IN_FLIGHT][NE]; float ww[ROWS_IN_FLIGHT][NE];
#pragma unroll
            for (int p = 0; p < ROWS_IN_FLIGHT; ++p) {
                const int row = g * 16 + 2 * (p0 + p) + half;
#pragma unroll
                for (int j = 0; j < NE; ++j) {
                    const int c = col[row * NE + j]; ww[p][j] = w[row * NE + j];
                    v[p][j] = *reinterpret_cast<const float4*>(xd + (size_t)c * C + 4 * c4);
                }
            }
#pragma unroll
            for (int p = 0; p < ROWS_IN_FLIGHT; ++p) {
                const int row = g * 16 + 2 * (p0 + p) + half;
                float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int j = 0; j < NE; ++j) { a.x = fmaf(ww[p][j], v[p][j].x, a.x); a.y = fmaf(ww[p][j], v[p][j].y, a.y); a.z = fmaf(ww[p][j], v[p][j].z, a.z); a.w = fmaf(ww[p][j], v[p][j].w, a.w); }
                *reinterpret_cast<float4*>(out + (size_t)row * C + 4 * c4) = a;
            }
        }
    }
}

__device__ __forceinline__ void dma16(const void* gsrc, unsigned lds_byte) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_byte) : "memory");
}

template <int ENT>
__global__ __launch_bounds__(256) void gather_ad(const float* __restrict__ xd, const int* __restrict__ col, const float* __restrict__ w, float* __restrict__ out, int ngroups, int ngroups_x) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int l = threadIdx.x & 63, m = l & 15, q = l >> 4, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem + (unsigned)wave * (ENT * NT * 1024);
    const float4* lbuf = reinterpret_cast<const float4*>(smem + wave * (ENT * NT * 1024));
    for (int it = 0;; ++it) {
        int g;
        if (ngroups_x) { const int xc = blockIdx.x & 7, sl = blockIdx.x >> 3, gx = gridDim.x >> 3; const int lg = (sl + it * gx) * 4 + wave; if (lg >= ngroups_x) break; g = xc * ngroups_x + lg; if (g >= ngroups) break; }
        else { g = (blockIdx.x + it * gridDim.x) * 4 + wave; if (g >= ngroups) break; }
        const int row = g * 16 + m;
        float acc[NT][4] = {};
        for (int j0 = 0; j0 < NE; j0 += ENT) {
            float ww[ENT];
#pragma unroll
            for (int u = 0; u < ENT; ++u) {
                const int c = col[row * NE + j0 + u]; ww[u] = w[row * NE + j0 + u];
                const float* src = xd + (size_t)c * C + 4 * q;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) dma16(src + 16 * nt, lds0 + (unsigned)(u * NT + nt) * 1024u);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int u = 0; u < ENT; ++u)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const float4 v = lbuf[(u * NT + nt) * 64 + l];
                    acc[nt][0] = fmaf(ww[u], v.x, acc[nt][0]); acc[nt][1] = fmaf(ww[u], v.y, acc[nt][1]);
                    acc[nt][2] = fmaf(ww[u], v.z, acc[nt][2]); acc[nt][3] = fmaf(ww[u], v.w, acc[nt][3]);
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        float* o = out + (size_t)row * C + 4 * q;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) *reinterpret_cast<float4*>(o + 16 * nt) = make_float4(acc[nt][0], acc[nt][1], acc[nt][2], acc[nt][3]);
    }
}

static int g_xcd = 0;
template <typename K>
void run(const char* name, K kern, size_t smem, int wg_per_cu, const float* xd, const int* col, const float* w, float* out, int V) {
    int cus = 256; hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    if (smem) HC(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipEvent_t e0, e1; HC(hipEventCreate(&e0)); HC(hipEventCreate(&e1));
    const int ngroups = V / 16, reps = 5, ngroups_x = g_xcd ? (ngroups + 7) / 8 : 0;
    hipLaunchKernelGGL(kern, dim3(cus * wg_per_cu), dim3(256), smem, 0, xd, col, w, out, ngroups, ngroups_x);
    HC(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kern, dim3(cus * wg_per_cu), dim3(256), smem, 0, xd, col, w, out, ngroups, ngroups_x);
    HC(hipEventRecord(e1)); HC(hipEventSynchronize(e1)); HC(hipGetLastError());
    float ms = 0; HC(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    const double gathered = (double)V * NE * C * 4;
    { std::vector<float> ho(4096); HC(hipMemcpy(ho.data(), out + 12345 * 128, 4096 * 4, hipMemcpyDeviceToHost)); double cs = 0; for (int i = 0; i < 4096; ++i) cs += ho[i] * (1 + i % 7); printf("[cs %.3f] ", cs); }
    printf("%-22s wg/cu=%d: %7.1f us, gathered %.1f B/clk/CU (%.2f TB/s chip), [V,C] passes/s equiv %.2f TB/s\n", name, wg_per_cu, ms * 1e3,
           gathered / cus / (ms * 1e-3) / 2.4e9, gathered / (ms * 1e-3) / 1e12, 2.0 * V * C * 4 / (ms * 1e-3) / 1e12);
}

int main(int argc, char** argv) {
    const bool random = argc > 1 && atoi(argv[1]) == 1;
    g_xcd = argc > 2 ? atoi(argv[2]) : 0;
    const int V = argc > 3 ? atoi(argv[3]) : 160000, Vm = V < 10000 ? V : 10000, s = 100;
    const int xrows = argc > 4 ? atoi(argv[4]) : 0;          // > 0: all neighbours folded into the first xrows rows (L2-resident source)
    std::vector<int> col((size_t)V * NE); std::vector<float> w((size_t)V * NE, 0.25f);
    srand(1);
    for (int v = 0; v < V; ++v) {
        const int base = v / Vm * Vm, i = v - base;
        const int d[NE] = {0, 1, -1, s, -s, s + 1, -s - 1, 0};
        for (int j = 0; j < NE; ++j) col[(size_t)v * NE + j] = random ? base + rand() % Vm : base + ((i + d[j]) % Vm + Vm) % Vm;
        if (xrows) for (int j = 0; j < NE; ++j) col[(size_t)v * NE + j] %= xrows;
        w[(size_t)v * NE + 7] = 0.f;
    }
    std::vector<float> h((size_t)V * C, 1.f);
    float *xd, *out, *wd; int* cd;
    HC(hipMalloc(&xd, h.size() * 4)); HC(hipMalloc(&out, h.size() * 4)); HC(hipMalloc(&wd, w.size() * 4)); HC(hipMalloc(&cd, col.size() * 4));
    HC(hipMemcpy(xd, h.data(), h.size() * 4, hipMemcpyHostToDevice)); HC(hipMemcpy(wd, w.data(), w.size() * 4, hipMemcpyHostToDevice));
    HC(hipMemcpy(cd, col.data(), col.size() * 4, hipMemcpyHostToDevice));
    printf("pattern: %s, V = %d, C = %d, %d entries per row, %s\n", random ? "random within the mesh" : "torus (0, +-1, +-100, +-101)", V, C, NE, g_xcd ? "16-row groups contiguous per XCD" : "16-row groups round-robin over workgroups");
    if (xrows) printf("source folded into %d rows (%.1f MB)\n", xrows, xrows * 512e-6);
    for (int wg = 1; wg <= 2; ++wg) {
        run("A  ent=2", gather_a<2>, 0, wg, xd, cd, wd, out, V);
        run("A  ent=4", gather_a<4>, 0, wg, xd, cd, wd, out, V);
        run("A  ent=8", gather_a<8>, 0, wg, xd, cd, wd, out, V);
        run("A2 ent=2 (full lines)", gather_a2<2>, 0, wg, xd, cd, wd, out, V);
        run("A2 ent=4 (full lines)", gather_a2<4>, 0, wg, xd, cd, wd, out, V);
        run("B  pairs=1", gather_b<1>, 0, wg, xd, cd, wd, out, V);
        run("B  pairs=2", gather_b<2>, 0, wg, xd, cd, wd, out, V);
        run("B  pairs=4", gather_b<4>, 0, wg, xd, cd, wd, out, V);
        run("AD ent=2 (LDS-DMA)", gather_ad<2>, 4 * 2 * NT * 1024, wg, xd, cd, wd, out, V);
        run("AD ent=4 (LDS-DMA)", gather_ad<4>, 4 * 4 * NT * 1024, wg, xd, cd, wd, out, V);
    }
    run("A  ent=4", gather_a<4>, 0, 4, xd, cd, wd, out, V);
    run("B  pairs=2", gather_b<2>, 0, 4, xd, cd, wd, out, V);
    run("B  pairs=2", gather_b<2>, 0, 8, xd, cd, wd, out, V);
    return 0;
}
