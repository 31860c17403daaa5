// hipemu.h -- TEST INFRASTRUCTURE ONLY.
//
// A tiny single-threaded "fiber" emulator of the HIP execution model, so that the
// *same* kernel sources under diffusion-net_amd/csrc can be compiled for the host
// (clang++ -DDN_EMULATE) and their index logic (tile maps, LDS swizzles, MFMA
// fragment layouts, bounds guards) checked against the oracle in the CPU-only
// test tier.  It is never linked into, loaded by, or reachable from the product
// library: the product loader only ever opens the gfx950 libdiffnet_hip.so.
//
// Model: one workgroup at a time; every thread of the workgroup is a ucontext
// fiber scheduled round-robin; __syncthreads() and the wave-level collectives
// (MFMA, shuffles) are generation barriers that yield until the group arrives.
// MFMA fragment layout follows cdna_hip_programming.md section 3:
//   v_mfma_f32_32x32x2_f32: A lane l -> A[i=l&31][k=l>>5], B lane l -> B[k=l>>5][j=l&31],
//   C/D reg r of lane l -> row (r&3)+8*(r>>2)+4*(l>>5), col l&31;
//   result = fma(a_k1,b_k1, fma(a_k0,b_k0,c)) (k-ordered fmaf chain).
#pragma once
#include <ucontext.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct alignas(16) float4 { float x, y, z, w; };
struct alignas(8) float2 { float x, y; };
struct alignas(16) int4 { int x, y, z, w; };
struct alignas(16) uint4 { unsigned x, y, z, w; };
struct alignas(8) uint2 { unsigned x, y; };
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }
static inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }
static inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
static inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline float2 make_float2(float x, float y) { return float2{x, y}; }

typedef void* hipStream_t;
typedef int hipError_t;
enum { hipSuccess = 0 };
static inline hipError_t hipGetLastError() { return 0; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return 0; }
enum { hipMemcpyDeviceToDevice = 3 };
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, int, hipStream_t) { memcpy(d, s, n); return 0; }

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ static

namespace dnemu {
struct Idx { unsigned x, y, z; };
inline Idx g_threadIdx, g_blockIdx, g_blockDim, g_gridDim;

struct WaveState {
    int count = 0;
    unsigned gen = 0;
    float a[2][64], b[2][64];
    unsigned short ha[2][64][8], hb[2][64][8];   // bf16 MFMA operands
    const unsigned char* ptr[2][64];             // ds_read_b64_tr_b16 lane addresses
};
struct BlockState {
    std::vector<ucontext_t> ctx;
    std::vector<char*> stacks;
    std::vector<char> done;
    ucontext_t main_ctx;
    int cur = 0, nthreads = 0;
    int bar_count = 0;
    unsigned bar_gen = 0;
    std::vector<WaveState> waves;
    std::vector<char> smem;
    std::function<void()> body;
};
inline BlockState g_blk;
static const size_t kStack = 256 * 1024;

inline void yield() { swapcontext(&g_blk.ctx[g_blk.cur], &g_blk.main_ctx); }
inline void set_ids(int t) {
    g_threadIdx.x = t % g_blockDim.x;
    g_threadIdx.y = (t / g_blockDim.x) % g_blockDim.y;
    g_threadIdx.z = t / (g_blockDim.x * g_blockDim.y);
}
inline void trampoline() {
    g_blk.body();
    g_blk.done[g_blk.cur] = 1;
    swapcontext(&g_blk.ctx[g_blk.cur], &g_blk.main_ctx);
}
inline void block_barrier() {
    unsigned gen = g_blk.bar_gen;
    if (++g_blk.bar_count == g_blk.nthreads) { g_blk.bar_count = 0; g_blk.bar_gen++; return; }
    while (g_blk.bar_gen == gen) yield();
}
inline int cur_lane() { return g_blk.cur & 63; }
inline WaveState& cur_wave() { return g_blk.waves[g_blk.cur >> 6]; }
inline int wave_width() {
    int w = g_blk.cur >> 6;
    int left = g_blk.nthreads - w * 64;
    return left < 64 ? left : 64;
}
// returns the generation index in which this collective happens (for double buffering)
inline unsigned wave_barrier() {
    WaveState& w = cur_wave();
    unsigned gen = w.gen;
    if (++w.count == wave_width()) { w.count = 0; w.gen++; return gen; }
    while (w.gen == gen) yield();
    return gen;
}
inline char* dyn_smem() { return g_blk.smem.data(); }

template <class F>
void launch(dim3 grid, dim3 block, size_t smem_bytes, F body) {
    int nthreads = (int)(block.x * block.y * block.z);
    g_blockDim = {block.x, block.y, block.z};
    g_gridDim = {grid.x, grid.y, grid.z};
    g_blk.nthreads = nthreads;
    g_blk.body = body;
    if ((int)g_blk.stacks.size() < nthreads) {
        size_t old = g_blk.stacks.size();
        g_blk.stacks.resize(nthreads);
        for (size_t i = old; i < (size_t)nthreads; ++i) g_blk.stacks[i] = (char*)malloc(kStack);
    }
    g_blk.ctx.resize(nthreads);
    g_blk.done.assign(nthreads, 0);
    g_blk.waves.assign((nthreads + 63) / 64, WaveState());
    g_blk.smem.assign(smem_bytes + 64, 0);
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx) {
                g_blockIdx = {bx, by, bz};
                g_blk.bar_count = 0;
                for (auto& w : g_blk.waves) { w.count = 0; }
                std::fill(g_blk.done.begin(), g_blk.done.end(), 0);
                for (int t = 0; t < nthreads; ++t) {
                    getcontext(&g_blk.ctx[t]);
                    g_blk.ctx[t].uc_stack.ss_sp = g_blk.stacks[t];
                    g_blk.ctx[t].uc_stack.ss_size = kStack;
                    g_blk.ctx[t].uc_link = &g_blk.main_ctx;
                    makecontext(&g_blk.ctx[t], (void (*)())trampoline, 0);
                }
                int remaining = nthreads;
                long spins = 0;
                while (remaining > 0) {
                    remaining = 0;
                    for (int t = 0; t < nthreads; ++t) {
                        if (g_blk.done[t]) continue;
                        g_blk.cur = t;
                        set_ids(t);
                        swapcontext(&g_blk.main_ctx, &g_blk.ctx[t]);
                        if (!g_blk.done[t]) ++remaining;
                    }
                    if (++spins > 50000000L) { fprintf(stderr, "dnemu: deadlock (barrier mismatch)\n"); abort(); }
                }
            }
}
}  // namespace dnemu

#define threadIdx dnemu::g_threadIdx
#define blockIdx dnemu::g_blockIdx
#define blockDim dnemu::g_blockDim
#define gridDim dnemu::g_gridDim
static inline void __syncthreads() { dnemu::block_barrier(); }
// single-threaded fibers: plain read-modify-write is atomic here
static inline int atomicAdd(int* p, int v) { int o = *p; *p = o + v; return o; }
static inline int atomicOr(int* p, int v) { int o = *p; *p = o | v; return o; }
static inline unsigned atomicMax(unsigned* p, unsigned v) { unsigned o = *p; if (v > o) *p = v; return o; }
static inline unsigned atomicAdd(unsigned* p, unsigned v) { unsigned o = *p; *p = o + v; return o; }
static inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { unsigned long long o = *p; *p = o + v; return o; }

typedef float dnemu_f32x16 __attribute__((ext_vector_type(16)));
static inline dnemu_f32x16 dnemu_mfma_f32_32x32x2f32(float a, float b, dnemu_f32x16 c) {
    dnemu::WaveState& w = dnemu::cur_wave();
    int l = dnemu::cur_lane();
    unsigned slot = w.gen & 1;  // everyone in the wave sees the same gen before arriving
    w.a[slot][l] = a;
    w.b[slot][l] = b;
    dnemu::wave_barrier();
    dnemu_f32x16 d;
    for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        int col = l & 31;
        float v = c[r];
        v = fmaf(w.a[slot][row], w.b[slot][col], v);
        v = fmaf(w.a[slot][row + 32], w.b[slot][col + 32], v);
        d[r] = v;
    }
    return d;
}
// v_mfma_f32_32x32x16_bf16: lane l supplies A[i=l&31][k=8*(l>>5)+j] and B[k=8*(l>>5)+j][col=l&31], j = 0..7 (8 bf16 = 16 bytes);
// C/D layout as the f32 form.  fp32 accumulation (order inside the instruction is not architecturally specified).
static inline dnemu_f32x16 dnemu_mfma_f32_32x32x16_bf16(uint4 a, uint4 b, dnemu_f32x16 c) {
    dnemu::WaveState& w = dnemu::cur_wave();
    int l = dnemu::cur_lane();
    unsigned slot = w.gen & 1;
    const unsigned aw[4] = {a.x, a.y, a.z, a.w}, bw[4] = {b.x, b.y, b.z, b.w};
    for (int j = 0; j < 8; ++j) {
        w.ha[slot][l][j] = (unsigned short)((aw[j >> 1] >> (16 * (j & 1))) & 0xffffu);
        w.hb[slot][l][j] = (unsigned short)((bw[j >> 1] >> (16 * (j & 1))) & 0xffffu);
    }
    dnemu::wave_barrier();
    dnemu_f32x16 d;
    for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        int col = l & 31;
        float v = c[r];
        for (int g = 0; g < 2; ++g)
            for (int j = 0; j < 8; ++j)
                v = fmaf(__uint_as_float((unsigned)w.ha[slot][row + 32 * g][j] << 16),
                         __uint_as_float((unsigned)w.hb[slot][col + 32 * g][j] << 16), v);
        d[r] = v;
    }
    return d;
}
// v_mfma_f32_32x32x16_f16: same layout, fp16 operands
static inline dnemu_f32x16 dnemu_mfma_f32_32x32x16_f16(uint4 a, uint4 b, dnemu_f32x16 c) {
    dnemu::WaveState& w = dnemu::cur_wave();
    int l = dnemu::cur_lane();
    unsigned slot = w.gen & 1;
    const unsigned aw[4] = {a.x, a.y, a.z, a.w}, bw[4] = {b.x, b.y, b.z, b.w};
    for (int j = 0; j < 8; ++j) {
        w.ha[slot][l][j] = (unsigned short)((aw[j >> 1] >> (16 * (j & 1))) & 0xffffu);
        w.hb[slot][l][j] = (unsigned short)((bw[j >> 1] >> (16 * (j & 1))) & 0xffffu);
    }
    dnemu::wave_barrier();
    auto h2f = [](unsigned short u) { _Float16 h; memcpy(&h, &u, 2); return (float)h; };
    dnemu_f32x16 d;
    for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        int col = l & 31;
        float v = c[r];
        for (int g = 0; g < 2; ++g)
            for (int j = 0; j < 8; ++j)
                v = fmaf(h2f(w.ha[slot][row + 32 * g][j]), h2f(w.hb[slot][col + 32 * g][j]), v);
        d[r] = v;
    }
    return d;
}
// v_mfma_f32_16x16x32_bf16: lane l supplies A[i=l&15][k=8*(l>>4)+j] and B[k=8*(l>>4)+j][col=l&15], j = 0..7;
// accumulator register r of lane l is D[4*(l>>4)+r][l&15] (cdna_hip_programming.md section 3).
typedef float dnemu_f32x4 __attribute__((ext_vector_type(4)));
static inline dnemu_f32x4 dnemu_mfma_f32_16x16x32_bf16(uint4 a, uint4 b, dnemu_f32x4 c) {
    dnemu::WaveState& w = dnemu::cur_wave();
    int l = dnemu::cur_lane();
    unsigned slot = w.gen & 1;
    const unsigned aw[4] = {a.x, a.y, a.z, a.w}, bw[4] = {b.x, b.y, b.z, b.w};
    for (int j = 0; j < 8; ++j) {
        w.ha[slot][l][j] = (unsigned short)((aw[j >> 1] >> (16 * (j & 1))) & 0xffffu);
        w.hb[slot][l][j] = (unsigned short)((bw[j >> 1] >> (16 * (j & 1))) & 0xffffu);
    }
    dnemu::wave_barrier();
    dnemu_f32x4 d;
    for (int r = 0; r < 4; ++r) {
        int row = 4 * (l >> 4) + r, col = l & 15;
        float v = c[r];
        for (int g = 0; g < 4; ++g)
            for (int j = 0; j < 8; ++j)
                v = fmaf(__uint_as_float((unsigned)w.ha[slot][row + 16 * g][j] << 16),
                         __uint_as_float((unsigned)w.hb[slot][col + 16 * g][j] << 16), v);
        d[r] = v;
    }
    return d;
}
// v_mfma_f32_16x16x32_f16: the layout of the bf16 form, fp16 operands
static inline dnemu_f32x4 dnemu_mfma_f32_16x16x32_f16(uint4 a, uint4 b, dnemu_f32x4 c) {
    dnemu::WaveState& w = dnemu::cur_wave();
    int l = dnemu::cur_lane();
    unsigned slot = w.gen & 1;
    const unsigned aw[4] = {a.x, a.y, a.z, a.w}, bw[4] = {b.x, b.y, b.z, b.w};
    for (int j = 0; j < 8; ++j) {
        w.ha[slot][l][j] = (unsigned short)((aw[j >> 1] >> (16 * (j & 1))) & 0xffffu);
        w.hb[slot][l][j] = (unsigned short)((bw[j >> 1] >> (16 * (j & 1))) & 0xffffu);
    }
    dnemu::wave_barrier();
    auto h2f = [](unsigned short u) { _Float16 h; memcpy(&h, &u, 2); return (float)h; };
    dnemu_f32x4 d;
    for (int r = 0; r < 4; ++r) {
        int row = 4 * (l >> 4) + r, col = l & 15;
        float v = c[r];
        for (int g = 0; g < 4; ++g)
            for (int j = 0; j < 8; ++j)
                v = fmaf(h2f(w.ha[slot][row + 16 * g][j]), h2f(w.hb[slot][col + 16 * g][j]), v);
        d[r] = v;
    }
    return d;
}
// ds_read_b64_tr_b16 (semantics probed on gfx950, profiles/r01_exp_tr_read.txt): inside every 16-lane group the lanes'
// addresses name sixteen 8-byte chunks = a 4x16 matrix of 16-bit elements (row j = chunks of lanes 4j..4j+3); lane l
// receives column l&15, i.e. element j comes from the address of lane 4j + (l&15)/4 of its group, 16-bit slot (l&15)%4.
static inline uint2 dnemu_ds_read_tr16_b64(const unsigned char* p) {
    dnemu::WaveState& w = dnemu::cur_wave();
    int l = dnemu::cur_lane();
    unsigned slot = w.gen & 1;
    w.ptr[slot][l] = p;
    dnemu::wave_barrier();
    unsigned short e[4];
    const int g0 = l & ~15, c = l & 15;
    for (int j = 0; j < 4; ++j) memcpy(&e[j], w.ptr[slot][g0 + 4 * j + c / 4] + 2 * (c % 4), 2);
    return uint2{(unsigned)e[0] | ((unsigned)e[1] << 16), (unsigned)e[2] | ((unsigned)e[3] << 16)};
}
static inline float dnemu_shfl(float v, int src_lane, bool relative_xor, int width) {
    dnemu::WaveState& w = dnemu::cur_wave();
    int l = dnemu::cur_lane();
    unsigned slot = w.gen & 1;
    w.a[slot][l] = v;
    dnemu::wave_barrier();
    int src = relative_xor ? (l ^ src_lane) : src_lane;
    (void)width;
    if (src < 0 || src >= 64) src = l;
    return w.a[slot][src];
}
static inline float __shfl_xor(float v, int mask, int width = 64) { return dnemu_shfl(v, mask, true, width); }
static inline float __shfl_down(float v, int d, int width = 64) {
    int l = dnemu::cur_lane();
    int src = l + d;
    if ((src / width) != (l / width)) src = l;
    return dnemu_shfl(v, src, false, width);
}
static inline float __shfl(float v, int lane, int width = 64) {
    int l = dnemu::cur_lane();
    return dnemu_shfl(v, (l / width) * width + lane, false, width);
}
