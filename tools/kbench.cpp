// kbench.cpp -- standalone timing + spot-check harness for the C ABI of libdiffnet_hip.so (development aid, GPU only).
// No torch: starts in milliseconds, so one GPU call can compare many library build variants.
//   hipcc --offload-arch=gfx950 -O2 -std=c++17 tools/kbench.cpp -o tools/kbench -ldl
//   tools/kbench [--lib path.so] [--meshes 16] [--verts 10000] [--C 128] [--K 128] [--reps 20] [--ops a,b,..] [--check]
// Ops: to_basis from_basis diffusion diffusion_bwd spmm gradfeat gradfeat_bwd linear linear_relu linear_bwd
//      block_inf block_fwd block_bwd copy copyk (hand-written float4 copy / read / fill kernels on 2 GiB of rotating buffers)
// Every op line: avg us (hipEvents around `reps` back-to-back calls), algorithmic GB/s and, with --check, the worst error of a
// row sample against an fp64 host evaluation of the same formula (relative to the largest reference magnitude).
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <random>
#include <string>
#include <vector>
#include "../include/diffnet_hip.h"

#define HC(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %d (%s) at %s:%d\n", (int)e_, hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)
#define DC(x) do { int e_ = (x); if (e_) { fprintf(stderr, "library error %d at %s:%d: %s\n", e_, __FILE__, __LINE__, #x); exit(3); } } while (0)

// ---- hand-written HBM calibration kernels (hipMemcpy is a blit path, not the yard-stick): float4 grid-stride copy, plain and
//      with nontemporal loads/stores, a read-only sum and a write-only fill; run on rotating buffers of > 1 GiB in total.
typedef float kb_f4 __attribute__((ext_vector_type(4)));   // the nontemporal builtins take native vectors, not HIP's float4 struct
__device__ __forceinline__ float4 kb_nt_load(const float4* p) { const kb_f4 v = __builtin_nontemporal_load(reinterpret_cast<const kb_f4*>(p)); return make_float4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ void kb_nt_store(float4 v, float4* p) { __builtin_nontemporal_store(kb_f4{v.x, v.y, v.z, v.w}, reinterpret_cast<kb_f4*>(p)); }
template <bool NT>
__global__ __launch_bounds__(256) void k_copy_f4(const float4* __restrict__ src, float4* __restrict__ dst, long long n4) {
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        if (NT) { const float4 v = kb_nt_load(src + i); kb_nt_store(v, dst + i); }
        else dst[i] = src[i];
    }
}
// four independent float4 in flight per thread and iteration (a thread owns 4 consecutive 256-float4 stripes)
template <bool NT>
__global__ __launch_bounds__(256) void k_copy_f4x4(const float4* __restrict__ src, float4* __restrict__ dst, long long n4) {
    const long long stride = (long long)gridDim.x * 1024;
    for (long long i = (long long)blockIdx.x * 1024 + threadIdx.x; i < n4; i += stride) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const long long j = i + 256 * u; if (j < n4) v[u] = NT ? kb_nt_load(src + j) : src[j]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) { const long long j = i + 256 * u; if (j < n4) { if (NT) kb_nt_store(v[u], dst + j); else dst[j] = v[u]; } }
    }
}
__global__ __launch_bounds__(256) void k_read_f4(const float4* __restrict__ src, float* __restrict__ sink, long long n4) {
    const long long stride = (long long)gridDim.x * 1024;
    float s = 0.f;
    for (long long i = (long long)blockIdx.x * 1024 + threadIdx.x; i < n4; i += stride) {
#pragma unroll
        for (int u = 0; u < 4; ++u) { const long long j = i + 256 * u; if (j < n4) { const float4 v = src[j]; s += (v.x + v.y) + (v.z + v.w); } }
    }
    if (s == 123.456f) sink[0] = s;
}
__global__ __launch_bounds__(256) void k_fill_f4(float4* __restrict__ dst, long long n4, float val) {
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) dst[i] = make_float4(val, val, val, val);
}

struct Lib {
    void* h;
    template <class F> F sym(const char* n) { void* p = dlsym(h, n); if (!p) { fprintf(stderr, "missing symbol %s\n", n); exit(4); } return (F)p; }
};

template <class T> T* dev(const std::vector<T>& v) {
    T* p = nullptr;
    HC(hipMalloc(&p, std::max<size_t>(v.size(), 1) * sizeof(T)));
    if (!v.empty()) HC(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return p;
}
static float* devz(size_t n) { float* p; HC(hipMalloc(&p, std::max<size_t>(n, 1) * sizeof(float))); HC(hipMemset(p, 0, n * sizeof(float))); return p; }
static std::vector<float> host(const float* d, size_t n) { std::vector<float> v(n); HC(hipMemcpy(v.data(), d, n * sizeof(float), hipMemcpyDeviceToHost)); return v; }

int main(int argc, char** argv) {
    std::string libpath = "diffusion-net_amd/diffusion_net/libdiffnet_hip.so", ops = "all";
    int n_mesh = 16, verts = 10000, C = 128, K = 128, reps = 20, chunk_rows = 0, df_groups = 0;
    bool check = false, trace = false, no_plan = false;
    std::vector<std::pair<std::string, int>> lib_opts;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        auto nxt = [&]() { if (i + 1 >= argc) { fprintf(stderr, "missing value for %s\n", a.c_str()); exit(1); } return std::string(argv[++i]); };
        if (a == "--lib") libpath = nxt(); else if (a == "--meshes") n_mesh = atoi(nxt().c_str()); else if (a == "--verts") verts = atoi(nxt().c_str());
        else if (a == "--C") C = atoi(nxt().c_str()); else if (a == "--K") K = atoi(nxt().c_str()); else if (a == "--reps") reps = atoi(nxt().c_str());
        else if (a == "--ops") ops = nxt(); else if (a == "--check") check = true; else if (a == "--trace") trace = true; else if (a == "--chunk") chunk_rows = atoi(nxt().c_str());
        else if (a == "--groups") df_groups = atoi(nxt().c_str()); else if (a == "--no-plan") no_plan = true;
        else if (a == "--opt") { const std::string kv = nxt(); const size_t eq = kv.find('='); if (eq == std::string::npos) { fprintf(stderr, "--opt name=value\n"); return 1; }
            lib_opts.push_back({kv.substr(0, eq), atoi(kv.substr(eq + 1).c_str())}); }
        else { fprintf(stderr, "unknown argument %s\n", a.c_str()); return 1; }
    }
    auto want = [&](const char* o) { return ops == "all" || ("," + ops + ",").find(std::string(",") + o + ",") != std::string::npos; };
    Lib L{dlopen(libpath.c_str(), RTLD_NOW | RTLD_LOCAL)};
    if (!L.h) { fprintf(stderr, "dlopen %s: %s\n", libpath.c_str(), dlerror()); return 4; }
    auto tile_rows = L.sym<int (*)()>("dn_tile_rows")();
    for (auto& kv : lib_opts) if (L.sym<int (*)(const char*, int)>("dn_set_option")(kv.first.c_str(), kv.second)) { fprintf(stderr, "unknown library option %s\n", kv.first.c_str()); return 1; }

    // ---- synthetic ragged batch
    std::mt19937 rng(1234);
    std::uniform_real_distribution<float> U(0.f, 1.f);
    std::normal_distribution<float> Nrm(0.f, 1.f);
    std::vector<int> sizes(n_mesh);
    for (auto& s : sizes) s = (int)(verts * (0.9 + 0.2 * U(rng)));
    long long V = 0; for (int s : sizes) V += s;
    // chunk table: --chunk R gives fixed R-row chunks; default: as diffusion_net.batch.balanced_chunk_rows -- dn_tn_target_chunks_k(K) chunks of nearly equal size
    std::vector<int> per_mesh(n_mesh, chunk_rows);
    if (!chunk_rows) {
        auto tgt = (int (*)(int))dlsym(L.h, "dn_tn_target_chunks_k");
        const int target = tgt ? tgt(K) : 512;
        std::vector<int> cnt(n_mesh), cap(n_mesh); std::vector<double> quota(n_mesh); int sum = 0;
        for (int m = 0; m < n_mesh; ++m) { quota[m] = (double)target * sizes[m] / V; cap[m] = std::max(1, sizes[m] / 128); cnt[m] = std::min(cap[m], std::max(1, (int)quota[m])); sum += cnt[m]; }
        std::vector<int> order(n_mesh); for (int m = 0; m < n_mesh; ++m) order[m] = m;
        std::sort(order.begin(), order.end(), [&](int a, int b) { return quota[a] - (int)quota[a] > quota[b] - (int)quota[b]; });
        bool prog = true; int rem = target - sum;
        while (rem > 0 && prog) { prog = false; for (int j : order) if (rem > 0 && cnt[j] < cap[j]) { ++cnt[j]; --rem; prog = true; } }
        for (int m = 0; m < n_mesh; ++m) per_mesh[m] = 32 * std::max(1, (sizes[m] + cnt[m] * 32 - 1) / (cnt[m] * 32));
        chunk_rows = per_mesh[0];
    }
    std::vector<dn_tile_t> tiles, chunks, mrows; std::vector<int> mco{0};
    { int row0 = 0; for (int m = 0; m < n_mesh; ++m) { int v = sizes[m]; mrows.push_back({row0, v, m, 0});
        for (int r = 0; r < v; r += tile_rows) tiles.push_back({row0 + r, std::min(tile_rows, v - r), m, 0});
        int i = 0; for (int r = 0; r < v; r += per_mesh[m]) chunks.push_back({row0 + r, std::min(per_mesh[m], v - r), m, i++});
        mco.push_back((int)chunks.size()); row0 += v; } }
    std::vector<float> mass(V), evals((size_t)n_mesh * K), evecs((size_t)V * K);
    for (auto& m : mass) m = (0.5f + U(rng)) * 12.566f / verts;
    for (int m = 0; m < n_mesh; ++m) { std::vector<float> e(K); for (auto& x : e) x = 50.f * U(rng); e[0] = 0.f; std::sort(e.begin(), e.end()); std::copy(e.begin(), e.end(), evals.begin() + (size_t)m * K); }
    { const float s = sqrtf((float)verts / 12.566f); for (auto& x : evecs) x = s * Nrm(rng) * 0.3f; }
    // gradient pattern: ~7 entries per row, neighbours within the mesh and close in index (as a mesh ordering gives), row sums zero
    std::vector<int> rowptr(V + 1, 0), col; std::vector<float> vx, vy;
    { long long r = 0; for (int m = 0; m < n_mesh; ++m) { const int row0 = mrows[m].row0, v = sizes[m];
        for (int i = 0; i < v; ++i, ++r) { int deg = 6 + (int)(U(rng) * 3); std::vector<int> cs{row0 + i};
            for (int d = 0; d < deg - 1; ++d) { int j = i + (int)((U(rng) - 0.5f) * 256); j = std::min(v - 1, std::max(0, j)); cs.push_back(row0 + j); }
            std::sort(cs.begin(), cs.end()); cs.erase(std::unique(cs.begin(), cs.end()), cs.end());
            float sx = 0, sy = 0; size_t b = col.size();
            for (int c : cs) { col.push_back(c); float a = 20.f * Nrm(rng), bb = 20.f * Nrm(rng); vx.push_back(a); vy.push_back(bb); sx += a; sy += bb; }
            vx[b] -= sx; vy[b] -= sy; rowptr[r + 1] = (int)col.size(); } } }
    const long long nnz = col.size();
    std::vector<int> t_rowptr(V + 1, 0), t_col(nnz); std::vector<float> t_vx(nnz), t_vy(nnz);
    { for (long long j = 0; j < nnz; ++j) t_rowptr[col[j] + 1]++; for (long long i = 0; i < V; ++i) t_rowptr[i + 1] += t_rowptr[i];
      std::vector<int> fill(t_rowptr.begin(), t_rowptr.end() - 1);
      for (long long r = 0; r < V; ++r) for (int j = rowptr[r]; j < rowptr[r + 1]; ++j) { int p = fill[col[j]]++; t_col[p] = (int)r; t_vx[p] = vx[j]; t_vy[p] = vy[j]; } }
    dn_mesh_batch_t mb; memset(&mb, 0, sizeof(mb));
    mb.n_mesh = n_mesh; mb.v_total = (int)V; mb.k_eig = K; mb.n_tiles = (int)tiles.size(); mb.n_chunks = (int)chunks.size(); mb.g_nnz = (int)nnz;
    mb.tiles = dev(tiles); mb.chunks = dev(chunks); mb.mesh_chunk_off = dev(mco); mb.mesh_rows = dev(mrows);
    mb.mass = dev(mass); mb.evals = dev(evals); mb.evecs = dev(evecs);
    mb.g_rowptr = dev(rowptr); mb.g_col = dev(col); mb.g_vx = dev(vx); mb.g_vy = dev(vy);
    mb.gt_rowptr = dev(t_rowptr); mb.gt_col = dev(t_col); mb.gt_vx = dev(t_vx); mb.gt_vy = dev(t_vy);
    int df_used = 0;
    if (!no_plan && K == 128) {   // work plan of the one-launch diffusion operator, as diffusion_net.batch.MeshBatch attaches it
        const int nwg = L.sym<int (*)()>("dn_diffusion_plan_wgs")();
        std::vector<dn_tile_t> plan((size_t)DN_DIFFUSION_MAX_GROUPS * nwg);
        df_used = L.sym<int (*)(const int32_t*, int, int, int, dn_tile_t*)>("dn_diffusion_plan")(sizes.data(), n_mesh, nwg, df_groups, plan.data());
        if (df_used > 0) { plan.resize((size_t)df_used * nwg); mb.df_plan = dev(plan); mb.df_n_wg = nwg; mb.df_n_groups = df_used; mb.df_v_total = (int)V; }
    }
    {   // operand magnitudes for the split-fp16 engine, as diffusion_net.batch.MeshBatch provides them
        float am[2] = {0.f, 0.f};
        for (float v : evecs) am[0] = std::max(am[0], fabsf(v));
        for (float v : mass) am[1] = std::max(am[1], fabsf(v));
        float gn = 0.f;
        for (long long r = 0; r < V; ++r) { float sx = 0, sy = 0; for (int j = rowptr[r]; j < rowptr[r + 1]; ++j) { sx += fabsf(vx[j]); sy += fabsf(vy[j]); } gn = std::max(gn, std::max(sx, sy)); }
        const float* d = dev(std::vector<float>{am[0], am[1], gn});
        if (!getenv("KB_NO_AMAX")) { mb.evecs_amax = d; mb.mass_amax = d + 1; mb.grad_norm = d + 2; }
    }

    int sg_units_n = 0;
    if (!getenv("KB_NO_SG") && L.sym<int (*)(int, int)>("dn_spectral_grad_supported")(K, C)) {
        // spectral-gradient operands, as diffusion_net.batch.MeshBatch attaches them (dn_spectral.hip): built once, on the device
        auto f_units = L.sym<int (*)(const int32_t*, int, int, dn_tile_t*)>("dn_spectral_units");
        sg_units_n = f_units(sizes.data(), n_mesh, K, nullptr);
        std::vector<dn_tile_t> units(sg_units_n);
        f_units(sizes.data(), n_mesh, K, units.data());
        const dn_tile_t* d_units = dev(units);
        void* pack; HC(hipMalloc(&pack, L.sym<size_t (*)(int, int)>("dn_spectral_pack_bytes")(sg_units_n, K)));
        float* sgam = devz((size_t)4 * n_mesh);
        const size_t pws = L.sym<size_t (*)(const dn_mesh_batch_t*)>("dn_spectral_pack_workspace_bytes")(&mb);
        void* pw; HC(hipMalloc(&pw, pws));
        const int e = L.sym<int (*)(const dn_mesh_batch_t*, const dn_tile_t*, int, void*, float*, void*, size_t, void*)>("dn_spectral_pack_f32")(
            &mb, d_units, sg_units_n, pack, sgam, pw, pws, nullptr);
        if (e) { fprintf(stderr, "dn_spectral_pack_f32 -> %d\n", e); return 5; }
        HC(hipDeviceSynchronize());
        HC(hipFree(pw));
        mb.sg_pack = pack; mb.sg_units = d_units; mb.sg_amax = sgam; mb.sg_n_units = sg_units_n;
    }

    auto randv = [&](size_t n, float sc) { std::vector<float> v(n); for (auto& x : v) x = sc * Nrm(rng); return v; };
    std::vector<float> hx = randv((size_t)V * C, 1.f), hy = randv((size_t)V * C, 1.f), hW = randv((size_t)C * C, 1.f / sqrtf((float)C)),
                       hW2 = randv((size_t)C * C, 1.f / sqrtf((float)C)), hW3 = randv((size_t)3 * C * C, 1.f / sqrtf(3.f * C)), hb = randv(C, 0.1f),
                       hspec = randv((size_t)n_mesh * K * C, 1.f), htime(C, 0.05f);
    // Inputs / outputs rotate over NROT buffer sets, so that consecutive repetitions never find their operands in the 256 MiB
    // Infinity Cache (one 81 MB pair re-read 20 times does: a copy of it "runs" at 7 TB/s) -- in the network every operand comes
    // from another kernel and ~1.3 GB of activations stream between two uses of the same buffer.
    constexpr int NROT = 4;
    float *xr[NROT], *yr[NROT], *o0r[NROT], *o1r[NROT];
    for (int i = 0; i < NROT; ++i) { xr[i] = dev(hx); yr[i] = dev(hy); o0r[i] = devz((size_t)V * C); o1r[i] = devz((size_t)V * C); }
    float *x = xr[0], *y = yr[0], *W = dev(hW), *W2 = dev(hW2), *W3 = dev(hW3), *b = dev(hb), *spec = dev(hspec), *tm = dev(htime);
    float *o0 = o0r[0], *o1 = o1r[0], *o2 = devz((size_t)V * C), *o3 = devz((size_t)V * C), *o4 = devz((size_t)V * C);
    float *specout = devz((size_t)n_mesh * K * C), *dW = devz((size_t)3 * C * C), *db = devz(C), *dW2 = devz((size_t)C * C), *dt = devz(C);

    const float* kb_dout_amax = nullptr; float* kb_dx_amax = nullptr;
    dn_block_params_t bp; memset(&bp, 0, sizeof(bp));
    bp.C = C; bp.n_mlp = 3; bp.with_grad = 1; bp.with_rot = 1; bp.widths[0] = 3 * C; bp.widths[1] = bp.widths[2] = bp.widths[3] = C;
    bp.time = tm; bp.A_re = W; bp.A_im = W2; bp.W[0] = W3; bp.W[1] = W; bp.W[2] = W2; bp.b[0] = bp.b[1] = bp.b[2] = b; bp.drop_seed = 0x1234567ull;
    dn_block_saved_t sv; memset(&sv, 0, sizeof(sv));
    sv.xs = devz((size_t)n_mesh * K * C); sv.xd = devz((size_t)V * C); sv.gx = devz((size_t)V * C); sv.gy = devz((size_t)V * C); sv.g = devz((size_t)V * C);
    sv.bre = devz((size_t)V * C); sv.bim = devz((size_t)V * C); sv.h[0] = devz((size_t)V * C); sv.h[1] = devz((size_t)V * C);
    sv.amax = devz(DN_BLOCK_AMAX_WORDS);
    {   // block input / incoming gradient magnitudes (x = hx, d_out = hy), as the previous / next block would hand them over
        float ax = 0.f, ay = 0.f;
        for (float v : hx) ax = std::max(ax, fabsf(v));
        for (float v : hy) ay = std::max(ay, fabsf(v));
        float* d = dev(std::vector<float>{ax, ay, 0.f, 0.f});
        if (!getenv("KB_NO_AMAX")) { bp.x_amax = d; bp.out_amax = d + 2; }
        kb_dout_amax = getenv("KB_NO_AMAX") ? nullptr : d + 1; kb_dx_amax = d + 3;
    }
    dn_block_grads_t gr; memset(&gr, 0, sizeof(gr));
    gr.d_out_amax = kb_dout_amax; gr.d_x_amax = kb_dx_amax;
    gr.d_x = o4; gr.d_time = dt; gr.dA_re = devz((size_t)C * C); gr.dA_im = devz((size_t)C * C);
    gr.dW[0] = dW; gr.dW[1] = devz((size_t)C * C); gr.dW[2] = devz((size_t)C * C); gr.db[0] = db; gr.db[1] = devz(C); gr.db[2] = devz(C);

    auto f_ws = [&](const char* n) { return L.sym<size_t (*)(const dn_mesh_batch_t*, int)>(n); };
    size_t wsb = 0;
    wsb = std::max(wsb, f_ws("dn_to_basis_workspace_bytes")(&mb, C));
    wsb = std::max(wsb, f_ws("dn_diffusion_workspace_bytes")(&mb, C));
    wsb = std::max(wsb, f_ws("dn_gradfeat_workspace_bytes")(&mb, C));
    wsb = std::max(wsb, L.sym<size_t (*)(const dn_mesh_batch_t*, int, int)>("dn_linear_workspace_bytes")(&mb, C, C));
    wsb = std::max(wsb, L.sym<size_t (*)(const dn_mesh_batch_t*, const dn_block_params_t*, int)>("dn_block_fwd_workspace_bytes")(&mb, &bp, 0));
    wsb = std::max(wsb, L.sym<size_t (*)(const dn_mesh_batch_t*, const dn_block_params_t*)>("dn_block_bwd_workspace_bytes")(&mb, &bp));
    void* ws; HC(hipMalloc(&ws, wsb));
    hipStream_t st; HC(hipStreamCreate(&st));
    hipEvent_t e0, e1; HC(hipEventCreate(&e0)); HC(hipEventCreate(&e1));
    printf("# lib=%s V=%lld meshes=%d K=%d C=%d tiles=%d chunks=%d (rows %d) nnz=%lld ws=%.0f MB diffusion-plan groups=%d wgs=%d spectral-gradient units=%d\n", libpath.c_str(), V, n_mesh, K, C, mb.n_tiles, mb.n_chunks, chunk_rows, nnz, wsb / 1e6, df_used, mb.df_n_wg, sg_units_n);
    auto df_wg_times = [&]() {   // -DDN_DF_TRACE builds: when did every workgroup of the last backproject_kernel start / end (10 ns ticks, chip-wide clock)
        auto rd = (int (*)(unsigned long long*, int))dlsym(L.h, "dn_debug_df_wg_times_read");
        if (!rd) return;
        std::vector<unsigned long long> tb(512 * 2); rd(tb.data(), 512 * 2);
        const int n = mb.df_n_wg; unsigned long long t0 = ~0ull;
        for (int w = 0; w < n; ++w) if (tb[2 * w]) t0 = std::min(t0, tb[2 * w]);
        std::vector<double> st, en, life;
        for (int w = 0; w < n; ++w) if (tb[2 * w] && tb[2 * w + 1] >= tb[2 * w]) { st.push_back((tb[2 * w] - t0) * 0.01); en.push_back((tb[2 * w + 1] - t0) * 0.01); life.push_back((tb[2 * w + 1] - tb[2 * w]) * 0.01); }
        if (st.empty()) return;
        auto pct = [](std::vector<double> v, double q) { std::sort(v.begin(), v.end()); return v[(size_t)(q * (v.size() - 1))]; };
        printf("  backproject workgroups (us from the first start): start min/median/max %.2f %.2f %.2f | end min/median/max %.2f %.2f %.2f | lifetime min/median/max %.2f %.2f %.2f\n",
               pct(st, 0), pct(st, .5), pct(st, 1), pct(en, 0), pct(en, .5), pct(en, 1), pct(life, 0), pct(life, .5), pct(life, 1));
    };
    auto df_trace = [&]() {   // libraries built with -DDN_DF_TRACE: s_memtime stamps of the first 16 workgroups of the one-launch diffusion kernel
        auto rd = (int (*)(unsigned long long*, int))dlsym(L.h, "dn_debug_df_trace_read");
        if (!rd) return;
        std::vector<unsigned long long> tb(16 * 32); rd(tb.data(), 16 * 32);
        for (int w = 0; w < 16; ++w) { printf("  wg %2d:", w); for (int i = 1; i < 24; ++i) printf(" %lld", tb[w * 32 + i] >= tb[w * 32] ? (long long)(tb[w * 32 + i] - tb[w * 32]) : -1ll); printf("\n"); }
    };

    auto timeit = [&](const char* name, double bytes, double flops, auto fn) {
        for (int i = 0; i < 3; ++i) fn(i);
        HC(hipStreamSynchronize(st));
        HC(hipEventRecord(e0, st));
        for (int i = 0; i < reps; ++i) fn(i + 3);
        HC(hipEventRecord(e1, st));
        HC(hipEventSynchronize(e1));
        float ms; HC(hipEventElapsedTime(&ms, e0, e1));
        const double us = 1e3 * ms / reps;
        printf("%-16s %9.1f us  %7.0f GB/s  %6.1f TF", name, us, bytes / us / 1e3, flops / us / 1e6);
        fflush(stdout);
        return us;
    };
    auto endl_ = [&]() { printf("\n"); fflush(stdout); };
    // sample of rows for the fp64 spot checks
    std::vector<long long> rows_s;
    for (int i = 0; i < 48; ++i) rows_s.push_back((long long)(U(rng) * (V - 1)));
    for (auto& t : tiles) if (t.nrows < tile_rows) { rows_s.push_back(t.row0); rows_s.push_back(t.row0 + t.nrows - 1); }   // ragged tile ends
    rows_s.push_back(0); rows_s.push_back(V - 1);
    auto mesh_of = [&](long long r) { int m = 0; while (r >= mrows[m].row0 + mrows[m].nrows) ++m; return m; };
    auto report = [&](double err, double ref) { printf("  err %.2e", err / std::max(ref, 1e-30)); };

    const double VC = (double)V * C * 4, VK = (double)V * K * 4;
    if (want("copy")) { timeit("copy", 2 * VC, 0, [&](int it) { HC(hipMemcpyAsync(o0r[it % NROT], xr[it % NROT], (size_t)V * C * 4, hipMemcpyDeviceToDevice, st)); }); endl_(); }
    if (want("copyk")) {   // eight rotating 128 MiB source / destination pairs = 2 GiB touched per round trip
        constexpr int NR8 = 8; const long long n4 = (128ll << 20) / 16; const double bytes = 2.0 * (128ll << 20);
        float4 *cs[NR8], *cd[NR8];
        for (int i = 0; i < NR8; ++i) { HC(hipMalloc(&cs[i], 128ll << 20)); HC(hipMalloc(&cd[i], 128ll << 20)); HC(hipMemset(cs[i], 0x3c, 128ll << 20)); HC(hipMemset(cd[i], 0, 128ll << 20)); }
        timeit("copyk_memcpy", bytes, 0, [&](int it) { HC(hipMemcpyAsync(cd[it % NR8], cs[it % NR8], 128ll << 20, hipMemcpyDeviceToDevice, st)); }); endl_();
        for (int nb : {1024, 2048, 4096, 8192, 32768}) {
            char nm[64];
            snprintf(nm, sizeof nm, "copyk_f4_g%d", nb);
            timeit(nm, bytes, 0, [&](int it) { hipLaunchKernelGGL(k_copy_f4<false>, dim3(nb), dim3(256), 0, st, cs[it % NR8], cd[it % NR8], n4); }); endl_();
            snprintf(nm, sizeof nm, "copyk_f4nt_g%d", nb);
            timeit(nm, bytes, 0, [&](int it) { hipLaunchKernelGGL(k_copy_f4<true>, dim3(nb), dim3(256), 0, st, cs[it % NR8], cd[it % NR8], n4); }); endl_();
            snprintf(nm, sizeof nm, "copyk_f4x4_g%d", nb);
            timeit(nm, bytes, 0, [&](int it) { hipLaunchKernelGGL(k_copy_f4x4<false>, dim3(nb), dim3(256), 0, st, cs[it % NR8], cd[it % NR8], n4); }); endl_();
            snprintf(nm, sizeof nm, "copyk_f4x4nt_g%d", nb);
            timeit(nm, bytes, 0, [&](int it) { hipLaunchKernelGGL(k_copy_f4x4<true>, dim3(nb), dim3(256), 0, st, cs[it % NR8], cd[it % NR8], n4); }); endl_();
        }
        for (int nb : {2048, 8192}) {
            char nm[64];
            snprintf(nm, sizeof nm, "readk_f4x4_g%d", nb);
            timeit(nm, bytes / 2, 0, [&](int it) { hipLaunchKernelGGL(k_read_f4, dim3(nb), dim3(256), 0, st, cs[it % NR8], (float*)cd[0], n4); }); endl_();
            snprintf(nm, sizeof nm, "fillk_f4_g%d", nb);
            timeit(nm, bytes / 2, 0, [&](int it) { hipLaunchKernelGGL(k_fill_f4, dim3(nb), dim3(256), 0, st, cd[it % NR8], n4, 1.0f); }); endl_();
        }
        for (int i = 0; i < NR8; ++i) { HC(hipFree(cs[i])); HC(hipFree(cd[i])); }
    }
    if (want("to_basis")) {
        auto f = L.sym<int (*)(const dn_mesh_batch_t*, const float*, int, int, float*, void*, size_t, void*)>("dn_to_basis_f32");
        timeit("to_basis", VC + VK + V * 4.0, 2.0 * V * K * C, [&](int it) { DC(f(&mb, xr[it % NROT], C, 1, specout, ws, wsb, st)); });
        if (check) {
            auto got = host(specout, (size_t)n_mesh * K * C); double err = 0, ref = 0;
            for (int m : {0, n_mesh - 1}) for (int k : {0, 1, 37, K - 1}) for (int c : {0, 5, C - 1}) {
                double s = 0; for (int r = mrows[m].row0; r < mrows[m].row0 + mrows[m].nrows; ++r) s += (double)evecs[(size_t)r * K + k] * ((double)mass[r] * hx[(size_t)r * C + c]);
                err = std::max(err, fabs(s - got[((size_t)m * K + k) * C + c])); ref = std::max(ref, fabs(s)); }
            report(err, ref);
        }
        endl_();
    }
    if (want("from_basis")) {
        auto f = L.sym<int (*)(const dn_mesh_batch_t*, const float*, int, int, float*, void*)>("dn_from_basis_f32");
        timeit("from_basis", VC + VK, 2.0 * V * K * C, [&](int it) { DC(f(&mb, spec, C, 0, o0r[it % NROT], st)); });
        if (check) {
            auto got = host(o0, (size_t)V * C); double err = 0, ref = 0;
            for (long long r : rows_s) { int m = mesh_of(r); for (int c = 0; c < C; ++c) { double s = 0; for (int k = 0; k < K; ++k) s += (double)evecs[(size_t)r * K + k] * hspec[((size_t)m * K + k) * C + c];
                err = std::max(err, fabs(s - got[(size_t)r * C + c])); ref = std::max(ref, fabs(s)); } }
            report(err, ref);
        }
        endl_();
        if (trace) {   // libraries built with -DDN_RD_TRACE=<block>: s_memtime stamps of that workgroup's eight waves
            auto rd = (int (*)(unsigned long long*, int))dlsym(L.h, "dn_debug_rd_trace_read");
            if (rd) {
                DC(f(&mb, spec, C, 0, o0, st)); HC(hipStreamSynchronize(st));
                std::vector<unsigned long long> tb(8 * 64); rd(tb.data(), 8 * 64);
                unsigned long long t0 = ~0ull; for (int w = 0; w < 8; ++w) t0 = std::min(t0, tb[w * 64]);
                for (int w = 0; w < 8; ++w) { printf("  wave %d:", w); for (int i = 0; i < 40 && (i == 0 || tb[w * 64 + i] >= tb[w * 64 + i - 1]) ; ++i) printf(" %llu", tb[w * 64 + i] - t0); printf("\n"); }
            }
        }
    }
    if (want("diffusion")) {
        auto f = L.sym<int (*)(const dn_mesh_batch_t*, const float*, const float*, int, float*, float*, void*, size_t, void*)>("dn_diffusion_fwd_f32");
        double byts = 0; for (int s : sizes) byts += 4.0 * ((double)s * (2 * C + 2 * K + 1) + 2.0 * K * C + K + C);
        double us = timeit("diffusion", byts, 4.0 * V * K * C, [&](int it) { DC(f(&mb, xr[it % NROT], tm, C, sv.xs, o0r[it % NROT], ws, wsb, st)); });
        printf("  frac_hbm_8TBs %.3f", byts / us / 1e3 / 8000.0);
        if (check) {   // meshes 0 and last: fp64 spectrum on the host, then sampled rows of x_diffuse and sampled entries of xs
            auto got = host(o0r[0], (size_t)V * C), gxs = host(sv.xs, (size_t)n_mesh * K * C);
            double err = 0, ref = 0, errs = 0, refs = 0;
            for (int m : {0, n_mesh - 1}) {
                std::vector<double> sp((size_t)K * C, 0.0);
                for (int r = mrows[m].row0; r < mrows[m].row0 + mrows[m].nrows; ++r)
                    for (int k = 0; k < K; ++k) { const double w = (double)evecs[(size_t)r * K + k] * mass[r]; const float* xr = &hx[(size_t)r * C]; double* sk = &sp[(size_t)k * C];
                        for (int c = 0; c < C; ++c) sk[c] += w * xr[c]; }
                for (int k : {0, 5, K - 1}) for (int c : {0, 77, C - 1}) { errs = std::max(errs, fabs(sp[(size_t)k * C + c] - gxs[((size_t)m * K + k) * C + c])); refs = std::max(refs, fabs(sp[(size_t)k * C + c])); }
                for (int k = 0; k < K; ++k) for (int c = 0; c < C; ++c) sp[(size_t)k * C + c] *= exp(-(double)evals[(size_t)m * K + k] * htime[c]);
                for (int i = 0; i < 40; ++i) { const long long r = mrows[m].row0 + (long long)(i * 997 % mrows[m].nrows);
                    for (int c = 0; c < C; ++c) { double s2 = 0; for (int k = 0; k < K; ++k) s2 += (double)evecs[(size_t)r * K + k] * sp[(size_t)k * C + c];
                        err = std::max(err, fabs(s2 - got[(size_t)r * C + c])); ref = std::max(ref, fabs(s2)); } }
                const long long rl = mrows[m].row0 + mrows[m].nrows - 1;   // last row of the mesh (ragged end)
                for (int c = 0; c < C; ++c) { double s2 = 0; for (int k = 0; k < K; ++k) s2 += (double)evecs[(size_t)rl * K + k] * sp[(size_t)k * C + c];
                    err = std::max(err, fabs(s2 - got[(size_t)rl * C + c])); ref = std::max(ref, fabs(s2)); }
            }
            report(errs, refs); report(err, ref);
        }
        endl_();
        if (trace) { DC(f(&mb, xr[0], tm, C, sv.xs, o0r[0], ws, wsb, st)); HC(hipStreamSynchronize(st)); df_trace(); df_wg_times(); }
    }
    if (want("diffusion_bwd")) {   // needs sv.xs of the forward above (run --ops diffusion,diffusion_bwd for the check)
        auto f = L.sym<int (*)(const dn_mesh_batch_t*, const float*, const float*, const float*, int, const float*, float*, float*, void*, size_t, void*)>("dn_diffusion_bwd_f32");
        double byts = 0; for (int s : sizes) byts += 4.0 * ((double)s * (3 * C + 2 * K + 1) + 3.0 * K * C + K + C);
        double us = timeit("diffusion_bwd", byts, 4.0 * V * K * C, [&](int it) { DC(f(&mb, yr[it % NROT], sv.xs, tm, C, xr[it % NROT], o1r[it % NROT], dt, ws, wsb, st)); });
        printf("  frac_hbm_8TBs %.3f", byts / us / 1e3 / 8000.0);
        if (check) {   // meshes 0 and last: d_x = add + mass * (Phi (coef * (Phi^T d_xd))) in fp64 on sampled rows; d_time against the fp64 sum over ALL meshes' sampled channels
            auto got = host(o1r[0], (size_t)V * C), gxs = host(sv.xs, (size_t)n_mesh * K * C), gdt = host(dt, C);
            double err = 0, ref = 0;
            std::vector<double> dts(C, 0.0);
            for (int m = 0; m < n_mesh; ++m) {
                const bool rows_too = (m == 0 || m == n_mesh - 1);
                std::vector<double> sp((size_t)K * C, 0.0);
                for (int r = mrows[m].row0; r < mrows[m].row0 + mrows[m].nrows; ++r)
                    for (int k = 0; k < K; ++k) { const double w = (double)evecs[(size_t)r * K + k]; const float* yr_ = &hy[(size_t)r * C]; double* sk = &sp[(size_t)k * C];
                        for (int c = 0; c < C; ++c) sk[c] += w * yr_[c]; }
                for (int k = 0; k < K; ++k) for (int c = 0; c < C; ++c) { const double lam = evals[(size_t)m * K + k], coef = exp(-lam * htime[c]);
                    dts[c] += -lam * coef * sp[(size_t)k * C + c] * gxs[((size_t)m * K + k) * C + c]; sp[(size_t)k * C + c] *= coef; }
                if (!rows_too) continue;
                for (int i = 0; i < 40; ++i) { const long long r = mrows[m].row0 + (i == 39 ? mrows[m].nrows - 1 : (long long)(i * 997 % mrows[m].nrows));
                    for (int c = 0; c < C; ++c) { double s2 = 0; for (int k = 0; k < K; ++k) s2 += (double)evecs[(size_t)r * K + k] * sp[(size_t)k * C + c];
                        s2 = hx[(size_t)r * C + c] + (double)mass[r] * s2;
                        err = std::max(err, fabs(s2 - got[(size_t)r * C + c])); ref = std::max(ref, fabs(s2)); } }
            }
            double errt = 0, reft = 0;
            for (int c = 0; c < C; ++c) { errt = std::max(errt, fabs(dts[c] - gdt[c])); reft = std::max(reft, fabs(dts[c])); }
            report(err, ref); report(errt, reft);
        }
        endl_();
        if (trace) { DC(f(&mb, yr[0], sv.xs, tm, C, xr[0], o1r[0], dt, ws, wsb, st)); HC(hipStreamSynchronize(st)); df_trace(); }
    }
    if (want("spmm")) {
        auto f = L.sym<int (*)(const dn_mesh_batch_t*, const float*, int, float*, float*, void*)>("dn_grad_apply_fwd_f32");
        timeit("spmm", 4.0 * (V + 3.0 * nnz) + 3 * VC, 4.0 * nnz * C, [&](int it) { DC(f(&mb, xr[it % NROT], C, o1r[it % NROT], o2, st)); });
        endl_();
    }
    if (want("gradfeat")) {
        auto f = L.sym<int (*)(const dn_mesh_batch_t*, const float*, const float*, const float*, const float*, int, float*, float*, float*, void*)>("dn_gradfeat_fwd_f32");
        timeit("gradfeat", 5 * VC, 8.0 * V * C * C, [&](int it) { DC(f(&mb, xr[it % NROT], yr[it % NROT], W, W2, C, o0r[it % NROT], o1r[it % NROT], o2, st)); });
        if (check) {
            auto got = host(o0, (size_t)V * C); double err = 0, ref = 0;
            for (long long r : rows_s) for (int c = 0; c < C; ++c) { double bre = 0, bim = 0;
                for (int k = 0; k < C; ++k) { double gx = hx[(size_t)r * C + k], gy = hy[(size_t)r * C + k]; bre += gx * hW[(size_t)c * C + k] - gy * hW2[(size_t)c * C + k]; bim += gx * hW2[(size_t)c * C + k] + gy * hW[(size_t)c * C + k]; }
                double s = tanh(hx[(size_t)r * C + c] * bre + hy[(size_t)r * C + c] * bim);
                err = std::max(err, fabs(s - got[(size_t)r * C + c])); ref = 1.0; }
            report(err, ref);
        }
        endl_();
    }
    if (want("gradfeat_bwd")) {
        auto f = L.sym<int (*)(const dn_mesh_batch_t*, const float*, const float*, const float*, const float*, const float*, const float*, const float*, const float*, int,
                               float*, float*, float*, float*, void*, size_t, void*)>("dn_gradfeat_bwd_f32");
        timeit("gradfeat_bwd", 9 * VC, 16.0 * V * C * C, [&](int it) { DC(f(&mb, yr[it % NROT], sv.g, xr[it % NROT], yr[it % NROT], o1r[it % NROT], o2, W, W2, C, o3, o4, gr.dA_re, gr.dA_im, ws, wsb, st)); });
        endl_();
    }
    auto lin = L.sym<int (*)(const dn_mesh_batch_t*, const float*, int, const float*, const float*, int, int, const uint8_t*, float*, void*)>("dn_linear_fwd_f32");
    for (int relu = 0; relu < 2; ++relu) {
        if (!want(relu ? "linear_relu" : "linear")) continue;
        timeit(relu ? "linear_relu" : "linear", 2 * VC, 2.0 * V * C * C, [&](int it) { DC(lin(&mb, xr[it % NROT], C, W, b, C, relu, nullptr, o0r[it % NROT], st)); });
        if (check) {
            auto got = host(o0, (size_t)V * C); double err = 0, ref = 0;
            for (long long r : rows_s) for (int c = 0; c < C; ++c) { double s = hb[c]; for (int k = 0; k < C; ++k) s += (double)hx[(size_t)r * C + k] * hW[(size_t)c * C + k];
                if (relu) s = s > 0 ? s : 0; err = std::max(err, fabs(s - got[(size_t)r * C + c])); ref = std::max(ref, fabs(s)); }
            report(err, ref);
        }
        endl_();
    }
    if (want("linear_bwd")) {
        auto f = L.sym<int (*)(const dn_mesh_batch_t*, const float*, const float*, const float*, int, int, float*, float*, float*, void*, size_t, void*)>("dn_linear_bwd_f32");
        timeit("linear_bwd", 5 * VC, 4.0 * V * C * C, [&](int it) { DC(f(&mb, yr[it % NROT], xr[it % NROT], W, C, C, o1r[it % NROT], dW2, db, ws, wsb, st)); });
        if (check) {
            auto got = host(o1, (size_t)V * C); double err = 0, ref = 0;
            for (long long r : rows_s) for (int c = 0; c < C; ++c) { double s = 0; for (int k = 0; k < C; ++k) s += (double)hy[(size_t)r * C + k] * hW[(size_t)k * C + c];
                err = std::max(err, fabs(s - got[(size_t)r * C + c])); ref = std::max(ref, fabs(s)); }
            report(err, ref);
            auto gw = host(dW2, (size_t)C * C); err = ref = 0;
            for (int o : {0, 3, C - 1}) for (int i : {0, 7, C - 1}) { double s = 0; for (long long r = 0; r < V; ++r) s += (double)hy[(size_t)r * C + o] * hx[(size_t)r * C + i];
                err = std::max(err, fabs(s - gw[(size_t)o * C + i])); ref = std::max(ref, fabs(s)); }
            report(err, ref);
        }
        endl_();
    }
    auto blk_f = L.sym<int (*)(const dn_mesh_batch_t*, const dn_block_params_t*, const float*, float*, const dn_block_saved_t*, void*, size_t, void*)>("dn_block_fwd_f32");
    if (want("block_inf")) { dn_block_params_t p2 = bp; p2.drop_seed = 0; timeit("block_inf", 12 * VC, 0, [&](int it) { DC(blk_f(&mb, &p2, xr[it % NROT], o0r[it % NROT], nullptr, ws, wsb, st)); }); endl_(); }
    if (want("block_fwd")) {
        dn_block_saved_t svf = sv;
        if (getenv("KB_NO_BRE")) { svf.bre = nullptr; svf.bim = nullptr; }     // experiment: the training forward without the Bre / Bim stores (saved-set diet)
        timeit("block_fwd", 20 * VC, 0, [&](int it) { DC(blk_f(&mb, &bp, xr[it % NROT], o0r[it % NROT], &svf, ws, wsb, st)); });
        if (check) {   // fingerprint of the output (compare runs with DN_F16=0 / 1 / masks): mean |out| and three entries
            auto got = host(o0r[0], (size_t)V * C); double ma = 0; for (float v : got) ma += fabs(v);
            printf("  mean|out| %.9e  out[0] %.7e out[V/2] %.7e out[-1] %.7e", ma / got.size(), got[0], got[(size_t)(V / 2) * C + 5], got.back());
        }
        endl_();
    }
    if (want("block_bwd")) {
        auto f = L.sym<int (*)(const dn_mesh_batch_t*, const dn_block_params_t*, const float*, const dn_block_saved_t*, const float*, const dn_block_grads_t*, void*, size_t, void*)>("dn_block_bwd_f32");
        DC(blk_f(&mb, &bp, x, o0, &sv, ws, wsb, st));
        timeit("block_bwd", 40 * VC, 0, [&](int it) { DC(f(&mb, &bp, x, &sv, yr[it % NROT], &gr, ws, wsb, st)); });
        if (check) {
            auto got = host(o4, (size_t)V * C); double ma = 0, mx = 0; size_t nbad = 0, imx = 0;
            for (size_t i = 0; i < got.size(); ++i) { const float v = got[i]; if (!(fabs(v) < 1e30f)) { ++nbad; continue; } ma += fabs(v); if (fabs(v) > mx) { mx = fabs(v); imx = i; } }
            auto gw = host(dW, (size_t)3 * C * C); double mw = 0; for (float v : gw) mw += fabs(v);
            printf("  mean|d_x| %.9e  d_x[V/2] %.7e  mean|dW0| %.9e  max|d_x| %.4e at row %zu  non-finite %zu", ma / got.size(), got[(size_t)(V / 2) * C + 5], mw / gw.size(), mx, imx / C, nbad);
        }
        endl_();
    }
    if (trace) {   // libraries built with EXTRA=-DDN_WS_TRACE=<block>: stamps of the wave-specialised row GEMM (MFMA wave 0: 3 per slice, loader wave 4: 5)
        auto rd = (int (*)(unsigned long long*, int))dlsym(L.h, "dn_debug_rd_trace_read");
        if (rd) {
            for (const char* which : {"linear", "from_basis"}) {
                if (std::string(which) == "linear") DC(lin(&mb, x, C, W, b, C, 1, nullptr, o0, st));
                else DC((L.sym<int (*)(const dn_mesh_batch_t*, const float*, int, int, float*, void*)>("dn_from_basis_f32"))(&mb, spec, C, 0, o0, st));
                HC(hipStreamSynchronize(st));
                std::vector<unsigned long long> tb(12 * 256); rd(tb.data(), 12 * 256);
                printf("== trace %s\n", which);
                auto dump = [&](int w, int per, const char* names) {
                    printf(" wave %d (%s), deltas per iteration:\n", w, names);
                    const unsigned long long* t = &tb[(size_t)w * 256];
                    for (int it = 0; it < 20 && (it + 1) * per < 256; ++it) {
                        printf("  it %2d:", it);
                        for (int k = 0; k < per; ++k) printf(" %6llu", t[it * per + k + 1] - t[it * per + k]);
                        printf("   | iter %6llu\n", t[(it + 1) * per] - t[it * per]);
                    }
                };
                dump(0, 3, "MFMA: reads+mma issue | park | barrier");
                const int per = getenv("WS_TR_PER") ? atoi(getenv("WS_TR_PER")) : 6;   // 6: DN_WS_EARLY=1 (wait|split|request|LDS put|pieces|barrier); 5: =0 (wait|stage|pieces|request|barrier)
                dump(4, per, per == 6 ? "loader: vm wait | split | advance+request | LDS writes | pieces | barrier" : "loader: vm wait | split+LDS writes | pieces | advance+request | barrier");
            }
        }
    }
    if (getenv("KB_CLK")) {   // libraries built with EXTRA=-DDN_CLK_TRACE: shader clock each kernel ran at = d(s_memtime) / d(s_memrealtime at 100 MHz), last launch
        for (const char* nm : {"chain_fwd", "chain_bwd", "tn_multi", "tn_da"}) {
            auto rd = (int (*)(unsigned long long*, int))dlsym(L.h, (std::string("dn_debug_clk_read_") + nm).c_str());
            if (!rd) continue;
            std::vector<unsigned long long> b(1024, 0);
            if (rd(b.data(), 1024)) continue;
            double lo = 1e30, hi = 0, sum = 0, dur = 0; int n = 0;
            for (int w = 0; w < 256; ++w) {
                const double dt = (double)(b[4 * w + 2] - b[4 * w]), dr = (double)(b[4 * w + 3] - b[4 * w + 1]);
                if (b[4 * w + 1] == 0 || dr <= 0 || dt <= 0) continue;
                const double ghz = dt / (dr * 10.0);      // cycles per nanosecond
                lo = std::min(lo, ghz); hi = std::max(hi, ghz); sum += ghz; dur += dr * 0.01; ++n;
            }
            if (n) printf("== clock %-10s %3d workgroups: shader clock %.3f GHz (min %.3f, max %.3f) over a workgroup life of %.1f us on average\n", nm, n, sum / n, lo, hi, dur / n);
        }
    }
    if (trace) {   // libraries built with EXTRA=-DDN_CH_TRACE=<block>: s_memtime stamps of wave 0 of that workgroup of the chained forward kernel (last call)
        auto rd = (int (*)(unsigned long long*, int))dlsym(L.h, "dn_debug_ch_trace_read");
        if (rd) {
            HC(hipStreamSynchronize(st));
            std::vector<unsigned long long> tb(512, 0);
            if (rd(tb.data(), 512) == 0) {
                printf("== chain trace (cycles between consecutive stamps; 0-stamps end the list)\n");
                int n = 0; while (n < 512 && tb[n]) ++n;
                for (int i = 1; i < n; ++i) printf(" %7llu%s", tb[i] - tb[i - 1], (i % 12 == 0) ? "\n" : "");
                printf("\n total %llu cycles over %d stamps\n", n > 1 ? tb[n - 1] - tb[0] : 0ull, n);
            }
        }
    }
    HC(hipStreamSynchronize(st));
    return 0;
}
