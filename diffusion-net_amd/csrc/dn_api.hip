// dn_api.hip -- extern "C" entry points of libdiffnet_hip.so (see include/diffnet_hip.h).
// Host-side orchestration only: every function enqueues kernels of dn_gemm / dn_sparse /
// dn_pointwise on the caller's stream; nothing here allocates, frees or synchronises.
#include "dn_common.h"
#include "../../include/diffnet_hip.h"
#include <string.h>
#include <stdlib.h>
#include <atomic>
#include <stdio.h>
#define DN_MIN_TIME 1e-8f       /* layers.py:49 */

static_assert(sizeof(dn_tile_t) == sizeof(DnTile), "tile layout");
#define DN_ERR_INVALID 1   /* hipErrorInvalidValue */
#define DN_SMALLN_BLOCKS 2048
#define DN_CHECK(expr) do { int _e = (expr); if (_e) return _e; } while (0)

// ---- tuning options (dn_set_option / dn_get_option): relaxed atomic ints read at call time; nothing on the compute path writes them or reads
// the environment.  Kernel selection for A/B measurements and for tests that must run a particular kernel.  A value must not change between a
// *_workspace_bytes() query and the call it sizes, nor between a graph capture's warm-up and the capture (include/diffnet_hip.h); a caller that
// needs different engines for different models in one process uses dn_block_params_t.flags, which are per call.
enum { F16_TOB = 1, F16_FROMB = 2, F16_GF = 4, F16_MLP = 8, F16_LBI = 16, F16_GFB = 32, F16_TOB_B = 64, F16_FROMB_B = 128 };
namespace {
struct Opt { const char* name; std::atomic<int> value; };
enum { O_CHAIN, O_CHAIN_MIN_ROWS, O_CHAIN_SMALL_ROWS, O_CHAIN_NW, O_CHAIN_HH, O_F16, O_F16_MASK, O_F16_WGRAD, O_DIFFUSE, O_DIFFUSE_GROUPS, O_DIFFUSE_ORDER, O_DIFFUSE_FLAGS, O_DIFFUSE_SPLIT, O_SPECTRAL_GRAD, O_COUNT };
Opt g_opt[O_COUNT] = {
    {"chain", 1}, {"chain_min_rows", 0}, {"chain_small_rows", 0}, {"chain_nw", 0}, {"chain_hh", 0}, {"f16", 1},
    {"f16_mask", F16_GF | F16_MLP | F16_LBI | F16_GFB | F16_FROMB_B}, {"f16_wgrad", 0},
    {"diffuse", 2}, {"diffuse_groups", 1}, {"diffuse_order", 0}, {"diffuse_flags", DN_DF_FLAG_DEFER}, {"diffuse_split", 0}, {"spectral_grad", 1},
};
inline int opt(int i) { return g_opt[i].value.load(std::memory_order_relaxed); }
}  // namespace
int dn_opt_chain_nw(void) { return opt(O_CHAIN_NW); }

namespace {
struct Bump {
    char* p; size_t left; bool ok;
    Bump(void* ws, size_t n) : p((char*)ws), left(n), ok(true) {
        size_t mis = (256 - ((uintptr_t)p & 255)) & 255;
        if (mis > left) { ok = false; left = 0; } else { p += mis; left -= mis; }
    }
    float* f(size_t nfloat) {
        size_t bytes = (nfloat * sizeof(float) + 255) & ~(size_t)255;
        if (bytes > left) { ok = false; return nullptr; }
        float* r = (float*)p; p += bytes; left -= bytes; return r;
    }
};
inline size_t pad256(size_t nfloat) { return ((nfloat * sizeof(float) + 255) & ~(size_t)255); }
inline bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }
inline const DnTile* T(const dn_tile_t* t) { return reinterpret_cast<const DnTile*>(t); }
inline hipStream_t S(void* s) { return (hipStream_t)s; }

RgArgs rg_new(const dn_mesh_batch_t* mb) {
    RgArgs g;
    memset(&g, 0, sizeof(g));
    g.tiles = T(mb->tiles);
    g.acct_rows = mb->v_total;
    for (int o = 0; o < 2; ++o) for (int s = 0; s < 3; ++s) g.bsign[o][s] = 1.f;
    g.scale = 1.f;
    return g;
}
// Operand magnitudes of one product on the split-fp16 engine (see DnAmax).  A default-constructed F16 means "split-bf16 engine".
struct F16 {
    bool on = false;
    DnAmax a{}, b{};
    float* o = nullptr;     // receives max |output| (a word zeroed earlier in the same call), or null
};
F16 f16_of(const float* a0, const float* b0, float* o = nullptr) { F16 f; f.on = a0 && b0; f.a.p[0] = a0; f.b.p[0] = b0; f.o = o; return f; }
void rg_f16(RgArgs& g, const F16& f) { g.f16 = f.on ? 1 : 0; g.a_amax = f.a; g.b_amax = f.b; g.o_amax = f.o; }
void tn_f16(TnArgs& g, const F16& f) { g.f16 = f.on ? 1 : 0; g.a_amax = f.a; g.b_amax = f.b; }
void rg_seg(RgArgs& g, const float* p, const float* q, int w, int ld) {
    RgSeg& s = g.a[g.nseg++];
    s.p = p; s.q = q; s.w = w; s.ld = ld;
}
void rg_finish(RgArgs& g, int nout) {
    bool ok = (g.ldb % 4 == 0) && (g.b_mesh_stride % 4 == 0) && (g.b_colk || g.N % 4 == 0);
    for (int s = 0; s < g.nseg; ++s) {
        ok = ok && (g.a[s].w % 32 == 0) && (g.a[s].ld % 4 == 0) && al16(g.a[s].p) && al16(g.a[s].q);
        for (int o = 0; o < nout; ++o) ok = ok && al16(g.b[o][s]);
    }
    g.aligned = ok ? 1 : 0;
}
TnArgs tn_new(const dn_mesh_batch_t* mb) {
    TnArgs g;
    memset(&g, 0, sizeof(g));
    g.chunks = T(mb->chunks);
    g.acct_rows = mb->v_total;
    return g;
}
void tn_a(TnArgs& g, const float* p, const float* q, int w, int ld) {
    TnSeg& s = g.a[g.na++]; s.p = p; s.q = q; s.w = w; s.ld = ld; g.M += w;
}
void tn_b(TnArgs& g, const float* p, const float* q, int w, int ld) {
    TnSeg& s = g.b[g.nb++]; s.p = p; s.q = q; s.w = w; s.ld = ld; g.N += w;
}
void tn_finish(TnArgs& g) {
    bool ok = true;
    for (int i = 0; i < g.na; ++i) ok = ok && g.a[i].w % 4 == 0 && g.a[i].ld % 4 == 0 && al16(g.a[i].p) && al16(g.a[i].q);
    for (int i = 0; i < g.nb; ++i) ok = ok && g.b[i].w % 4 == 0 && g.b[i].ld % 4 == 0 && al16(g.b[i].p) && al16(g.b[i].q);
    g.aligned = ok ? 1 : 0;
}

// ---- dn_diffuse.hip, for batches that carry a plan, K = C = 128, 16-byte aligned operands.  Option "diffuse":
//   2 (default)  the back-projection of every diffusion (and of dn_from_basis_f32) is the DIRECT row product launch (bp_ok)
//   1            the whole operator as one persistent launch (diffuse_ok): correct on any residency, measured slower than the launches on
//                every batch but many-small-meshes ones -- kept selectable
//   0            the wave-specialised row GEMM of rounds 1-4
bool diffuse_ok(const dn_mesh_batch_t* mb, int C) {
    return opt(O_DIFFUSE) == 1 && mb->df_plan && mb->df_n_wg > 0 && mb->df_n_groups > 0 && mb->df_n_groups <= DN_DF_MAX_GROUPS && mb->k_eig == 128 && C == 128 &&
           mb->df_n_wg == dn_num_cus() && mb->df_v_total == mb->v_total;
}
// (the plan is caller data the kernels index memory with: it must have been made for this device's workgroup count and -- the stamp
// dn_mesh_batch_t.df_v_total, set by whoever called dn_diffusion_plan() -- for this batch's row count; anything else takes the row GEMM)
bool plan_matches(const dn_mesh_batch_t* mb) { return mb->df_n_wg == dn_num_cus() && mb->df_v_total == mb->v_total; }
bool bp_ok(const dn_mesh_batch_t* mb, int C) {
    return opt(O_DIFFUSE) != 0 && mb->df_plan && mb->df_n_wg > 0 && mb->df_n_groups == 1 && mb->k_eig == 128 && C == 128 && plan_matches(mb);
}
size_t diffuse_ws_floats(const dn_mesh_batch_t* mb) { return dn_diffuse_ws_bytes(mb->df_n_wg, mb->df_n_groups, mb->n_mesh) / sizeof(float); }
int diffuse_dt_rows(const dn_mesh_batch_t* mb) { return dn_diffuse_dt_rows(mb->df_n_wg, mb->df_n_groups); }
DfLaunch diffuse_new(const dn_mesh_batch_t* mb, void* ws) {
    DfLaunch L;
    memset(&L, 0, sizeof(L));
    L.plan = T(mb->df_plan); L.n_wg = mb->df_n_wg; L.n_groups = mb->df_n_groups; L.n_mesh = mb->n_mesh;
    L.evecs = mb->evecs; L.mass = mb->mass; L.evals = mb->evals; L.ws = ws;
    L.order = opt(O_DIFFUSE_ORDER); L.flags = opt(O_DIFFUSE_FLAGS); L.split = opt(O_DIFFUSE_SPLIT); L.acct_rows = mb->v_total;
    return L;
}
bool diffuse_aligned(const void* a, const void* b, const void* c, const void* d, const void* e, const void* f) {
    return al16(a) && al16(b) && al16(c) && al16(d) && al16(e) && al16(f);
}

// ---- building blocks shared by the per-op and the fused-block entry points ----
int to_basis_partials(const dn_mesh_batch_t* mb, const float* x, int C, bool use_mass, float* partial, hipStream_t st, const F16& f = F16()) {
    TnArgs g = tn_new(mb);
    tn_f16(g, f);
    tn_a(g, mb->evecs, nullptr, mb->k_eig, mb->k_eig);
    tn_b(g, x, nullptr, C, C);
    g.b_rowscale = use_mass ? mb->mass : nullptr;
    g.partial = partial;
    tn_finish(g);
    return dn_launch_tngemm(g, mb->n_chunks, st);
}
// the forward back-projection at K = C = 256 as the ring kernel of dn_backproject_wide.hip (3-term engine; option "diffuse" != 0; needs a
// workspace for the split spectrum: the block calls and dn_diffusion_f32 pass one, callers without take the row GEMM)
bool bw_ok(const dn_mesh_batch_t* mb, int C) {
    return opt(O_DIFFUSE) != 0 && mb->tiles && dn_backproject_wide_ok(mb->k_eig, C, mb->n_tiles) && al16(mb->evecs);
}
int from_basis(const dn_mesh_batch_t* mb, const float* spec, int C, float* out, const float* add, bool mass_epi, hipStream_t st,
               const F16& f = F16(), float* wide_ws = nullptr) {
    if (wide_ws && !mass_epi && !add && !f.on && bw_ok(mb, C) && al16(spec) && al16(out) && al16(wide_ws))
        return dn_launch_backproject_wide(T(mb->tiles), mb->n_tiles, mb->n_mesh, mb->evecs, spec, wide_ws, out, f.o, mb->k_eig, C, mb->v_total, st);
    if (bp_ok(mb, C) && al16(spec) && al16(out) && al16(add) && al16(mb->evecs))       // direct row product, on the engine f asks for
        return dn_launch_backproject(T(mb->df_plan), mb->df_n_wg, mb->evecs, spec, out, add, mass_epi ? mb->mass : nullptr, f.o, mb->v_total, st,
                                     f.on ? 1 : 0, &f.a, &f.b);
    RgArgs g = rg_new(mb);
    rg_f16(g, f);
    rg_seg(g, mb->evecs, nullptr, mb->k_eig, mb->k_eig);
    g.b[0][0] = spec; g.ldb = C; g.b_colk = 0; g.b_mesh_stride = (long long)mb->k_eig * C; g.N = C;
    g.o0 = out; g.ldo = C; g.ldr = C;
    if (mass_epi) { g.mode = DN_EPI_MASS_ADD; g.r0 = add; g.rowv = mb->mass; }
    else g.mode = DN_EPI_STORE;
    rg_finish(g, 1);
    return dn_launch_rowgemm(g, mb->n_tiles, 1, st);
}
int grad_apply_fwd(const dn_mesh_batch_t* mb, const float* x, int C, float* gx, float* gy, hipStream_t st, float* o_amax = nullptr,
                   const float* x_amax = nullptr) {
    SpArgs s; memset(&s, 0, sizeof(s));
    s.o_amax = o_amax; s.in_amax = x_amax; s.op_norm = x_amax ? mb->grad_norm : nullptr;   // the bound ||G||_inf max|x| when both are known
    s.rowptr = mb->g_rowptr; s.col = mb->g_col; s.va = mb->g_vx; s.vb = mb->g_vy;
    s.x1 = x; s.o1 = gx; s.o2 = gy; s.nrows = mb->v_total; s.C = C; s.ldx = C; s.ldo = C; s.mode = DN_SP_FWD2; s.div = 1.f; s.acct_nnz = mb->g_nnz;
    return dn_launch_spmm(s, st);
}
int grad_apply_bwd(const dn_mesh_batch_t* mb, const float* dgx, const float* dgy, const float* add, int C, float* dx, hipStream_t st,
                   float* o_amax = nullptr) {
    SpArgs s; memset(&s, 0, sizeof(s));
    s.o_amax = o_amax;
    s.rowptr = mb->gt_rowptr; s.col = mb->gt_col; s.va = mb->gt_vx; s.vb = mb->gt_vy;
    s.x1 = dgx; s.x2 = dgy; s.add = add; s.o1 = dx; s.nrows = mb->v_total; s.C = C; s.ldx = C; s.ldo = C; s.mode = DN_SP_BWD2; s.div = 1.f; s.acct_nnz = mb->g_nnz;
    return dn_launch_spmm(s, st);
}
int gradfeat_fwd(const dn_mesh_batch_t* mb, const float* gx, const float* gy, const float* A_re, const float* A_im, int C,
                 float* g_out, float* bre, float* bim, hipStream_t st, const F16& f = F16()) {
    RgArgs g = rg_new(mb);
    rg_f16(g, f);
    rg_seg(g, gx, nullptr, C, C);
    rg_seg(g, gy, nullptr, C, C);
    if (A_im) {   // Bre = gx A_re^T - gy A_im^T ; Bim = gx A_im^T + gy A_re^T   (layers.py:122-123)
        g.b[0][0] = A_re; g.bsign[0][0] = 1.f;  g.b[0][1] = A_im; g.bsign[0][1] = -1.f;
        g.b[1][0] = A_im; g.bsign[1][0] = 1.f;  g.b[1][1] = A_re; g.bsign[1][1] = 1.f;
    } else {      // Bre = gx A^T ; Bim = gy A^T                                 (layers.py:125-126)
        g.b[0][0] = A_re; g.bsign[0][0] = 1.f;  g.b[0][1] = A_re; g.bsign[0][1] = 0.f;
        g.b[1][0] = A_re; g.bsign[1][0] = 0.f;  g.b[1][1] = A_re; g.bsign[1][1] = 1.f;
    }
    g.ldb = C; g.b_colk = 1; g.N = C;
    g.mode = DN_EPI_GRADFEAT; g.r0 = gx; g.r1 = gy; g.ldr = C;
    g.o0 = g_out; g.o1 = bre; g.o2 = bim; g.ldo = C;
    if (!bre || !bim) { g.o1 = nullptr; g.o2 = nullptr; }
    rg_finish(g, 2);
    return dn_launch_rowgemm(g, mb->n_tiles, 2, st);
}
// d_dots = d(pre-tanh inner product).  d_gx = d_dots*Bre + dBre A_re + dBim A_im ; d_gy = d_dots*Bim - dBre A_im + dBim A_re
int gradfeat_bwd_inputs(const dn_mesh_batch_t* mb, const float* ddots, const float* gx, const float* gy, const float* bre,
                        const float* bim, const float* A_re, const float* A_im, int C, float* dgx, float* dgy, hipStream_t st,
                        const F16& f = F16()) {
    RgArgs g = rg_new(mb);
    rg_f16(g, f);
    rg_seg(g, ddots, gx, C, C);   // dBre = d_dots * gx
    rg_seg(g, ddots, gy, C, C);   // dBim = d_dots * gy
    if (A_im) {
        g.b[0][0] = A_re; g.bsign[0][0] = 1.f;   g.b[0][1] = A_im; g.bsign[0][1] = 1.f;
        g.b[1][0] = A_im; g.bsign[1][0] = -1.f;  g.b[1][1] = A_re; g.bsign[1][1] = 1.f;
    } else {
        g.b[0][0] = A_re; g.bsign[0][0] = 1.f;   g.b[0][1] = A_re; g.bsign[0][1] = 0.f;
        g.b[1][0] = A_re; g.bsign[1][0] = 0.f;   g.b[1][1] = A_re; g.bsign[1][1] = 1.f;
    }
    g.ldb = C; g.b_colk = 0; g.N = C;
    g.mode = DN_EPI_GRADFEAT_BWD; g.r0 = ddots; g.r1 = bre; g.r2 = bim; g.ldr = C;
    g.o0 = dgx; g.o1 = dgy; g.ldo = C;
    rg_finish(g, 2);
    return dn_launch_rowgemm(g, mb->n_tiles, 2, st);
}
// dd_amax / g_amax: magnitudes of ddots and of gx, gy for the split-fp16 engine (both null: split-bf16)
int gradfeat_bwd_weights(const dn_mesh_batch_t* mb, const float* ddots, const float* gx, const float* gy, int C,
                         float* dA_re, float* dA_im, float* partial, float* psum, hipStream_t st, MrJobs* defer = nullptr,
                         const float* dd_amax = nullptr, const float* g_amax = nullptr) {
    if (C == 128 && dA_im && al16(ddots) && al16(gx) && al16(gy) && al16(partial)) {
        // one pass over ddots, gx, gy: every workgroup computes all four quadrants for its row range (dn_tn_da.hip)
        int nwg = dn_num_cus();
        if (nwg > 2 * mb->n_chunks) nwg = 2 * mb->n_chunks;      // the workspace holds n_chunks * 4 C^2 floats
        DN_CHECK(dn_launch_tn_da(ddots, gx, gy, mb->v_total, partial, nwg, st, dd_amax, g_amax));
        if (defer && defer->push(partial, nwg, 2LL * C * C, dA_re, dA_im, (long long)C * C)) return 0;
        return dn_launch_reduce_split(partial, nwg, dA_re, dA_im, (long long)C * C, st);
    }
    TnArgs g = tn_new(mb);
    tn_a(g, ddots, gx, C, C);
    tn_a(g, ddots, gy, C, C);
    tn_b(g, gx, nullptr, C, C);
    tn_b(g, gy, nullptr, C, C);
    if (dd_amax && g_amax) { F16 f = f16_of(dd_amax, g_amax); f.a.mul = g_amax; tn_f16(g, f); }
    g.partial = partial;
    g.group = dn_tn_global_group_mn(mb->n_chunks, g.M, g.N);
    tn_finish(g);
    DN_CHECK(dn_launch_tngemm(g, mb->n_chunks, st));
    DN_CHECK(dn_launch_reduce(partial, psum, dn_tn_npartial(mb->n_chunks, g.group), 4LL * C * C, 4LL * C * C, st));
    return dn_launch_combine_dA(psum, dA_re, dA_im, C, st);
}
// y = act(sum_s x_s W[:, off_s:off_s+w_s]^T + b)
int linear_fwd(const dn_mesh_batch_t* mb, const float* const* xs, const int* ws_, int nseg, const float* W, int ldw,
               const float* b, int C_out, int mode, const uint8_t* mask, const float* resid, float* out, hipStream_t st,
               unsigned long long rng_seed = 0, const unsigned long long* rng_seed_dev = nullptr, const F16& f = F16()) {
    RgArgs g = rg_new(mb);
    rg_f16(g, f);
    int off = 0;
    for (int s = 0; s < nseg; ++s) {
        rg_seg(g, xs[s], nullptr, ws_[s], ws_[s]);
        g.b[0][s] = W + off;
        off += ws_[s];
    }
    g.ldb = ldw; g.b_colk = 1; g.N = C_out;
    g.mode = mode; g.bias = b; g.mask = mask; g.rng_seed = mask ? 0ull : rng_seed; g.rng_seed_dev = rng_seed_dev; g.scale = (mask || rng_seed) ? 2.f : 1.f; g.r0 = resid; g.ldr = C_out;
    g.o0 = out; g.ldo = C_out;
    rg_finish(g, 1);
    return dn_launch_rowgemm(g, mb->n_tiles, 1, st);
}
// d_in[:, n-range] = epi( d_a W[:, col_off : col_off+N] )
int linear_bwd_input(const dn_mesh_batch_t* mb, const float* d_a, int C_out, const float* W, int ldw, int col_off, int N,
                     int mode, const float* r0, float scale, float* out, hipStream_t st, const F16& f = F16()) {
    RgArgs g = rg_new(mb);
    rg_f16(g, f);
    rg_seg(g, d_a, nullptr, C_out, C_out);
    g.b[0][0] = W + col_off; g.ldb = ldw; g.b_colk = 0; g.N = N;
    g.mode = mode; g.r0 = r0; g.ldr = N; g.scale = scale;
    g.o0 = out; g.ldo = N;
    rg_finish(g, 1);
    return dn_launch_rowgemm(g, mb->n_tiles, 1, st);
}
// dW[o][i] = sum_r d_a[r,o] in[r,i] ; db[o] = sum_r d_a[r,o]
// launches collected for ONE multi-problem launch (dn_launch_tngemm_multi)
struct TnBatch { TnArgs g[3]; int nchunks[3]; int count = 0; };
int linear_bwd_weights(const dn_mesh_batch_t* mb, const float* d_a, int C_out, const float* const* ins, const int* ws_, int nseg,
                       float* dW, float* db, float* partial, float* colsum, hipStream_t st, MrJobs* defer = nullptr, const F16& f = F16(),
                       TnBatch* batch = nullptr) {
    TnArgs g = tn_new(mb);
    tn_f16(g, f);
    tn_a(g, d_a, nullptr, C_out, C_out);
    for (int s = 0; s < nseg; ++s) tn_b(g, ins[s], nullptr, ws_[s], ws_[s]);
    g.partial = partial; g.colsum = db ? colsum : nullptr;
    g.group = dn_tn_global_group_mn(mb->n_chunks, g.M, g.N);
    // several products in one launch: the launch has 3-5x the workgroups of one product, so two chunks per workgroup still fill the device and
    // halve the partial tiles written here and read by the deferred sum (measured, block backward at 160k vertices: 1 / 2 / 3 / 4 chunks per
    // workgroup 889 / 879 / 891 / 909 us)
    if (batch && g.group < 2 && mb->n_chunks >= dn_num_cus()) g.group = 2;
    tn_finish(g);
    const int npart = dn_tn_npartial(mb->n_chunks, g.group);
    const bool can_defer = defer && g.M % 4 == 0 && defer->count + 2 <= DN_MR_MAX_JOBS && al16(partial) && al16(colsum);
    // (a product whose launch is handed to the caller must have its sums deferred too: they run after the caller's launch)
    if (batch && batch->count < 3 && can_defer) { batch->g[batch->count] = g; batch->nchunks[batch->count] = mb->n_chunks; ++batch->count; }
    else DN_CHECK(dn_launch_tngemm(g, mb->n_chunks, st));
    if (can_defer && defer->push(partial, npart, (long long)g.M * g.N, dW)) {          // summed with the block's other gradients, one launch
        if (db && !defer->push(colsum, npart, g.M, db)) return dn_launch_reduce(colsum, db, npart, g.M, g.M, st);
        return 0;
    }
    if (db) return dn_launch_reduce_pair(partial, dW, (long long)g.M * g.N, colsum, db, g.M, npart, st);   // one launch for both
    return dn_launch_reduce(partial, dW, npart, (long long)g.M * g.N, (long long)g.M * g.N, st);
}
// per-layer key of the in-kernel dropout (0 stays 0 = off; never maps a live seed to 0)
unsigned long long layer_seed(unsigned long long seed, int layer) {
    if (!seed) return 0ull;
    unsigned long long s = seed + 0x9E3779B97F4A7C15ull * (unsigned long long)(layer + 1);
    return s ? s : 0x9E3779B97F4A7C15ull;
}
int max_width(const dn_block_params_t* p) {
    int m = p->C;
    for (int i = 1; i <= p->n_mlp; ++i) if (p->widths[i] > m) m = p->widths[i];
    return m;
}
bool block_params_ok(const dn_block_params_t* p) {
    if (!p || p->C <= 0 || p->n_mlp < 1 || p->n_mlp > DN_MAX_MLP_LAYERS) return false;
    const int in0 = (p->with_grad ? 3 : 2) * p->C;
    return p->widths[0] == in0 && p->widths[p->n_mlp] == p->C;
}
}  // namespace

// ---- opt-in per-kernel timing --------------------------------------------------------------
#ifndef DN_EMULATE
#include <vector>
namespace {
struct ProfRec { hipEvent_t e0, e1; int kind; };
struct ProfState {
    bool on = false;
    std::vector<ProfRec> pool;      // reused event pairs
    size_t used = 0;
    double ms[DN_K_COUNT] = {0}, flops[DN_K_COUNT] = {0}, bytes[DN_K_COUNT] = {0};
    long long launches[DN_K_COUNT] = {0};
    bool pending = false;
} g_prof;
void prof_drain() {
    for (size_t i = 0; i < g_prof.used; ++i) {
        float t = 0.f;
        if (hipEventSynchronize(g_prof.pool[i].e1) == hipSuccess &&
            hipEventElapsedTime(&t, g_prof.pool[i].e0, g_prof.pool[i].e1) == hipSuccess)
            g_prof.ms[g_prof.pool[i].kind] += t;
    }
    g_prof.used = 0;
}
}  // namespace
void dn_prof_begin(int kind, hipStream_t stream) {
    if (!g_prof.on) return;
    if (g_prof.used == g_prof.pool.size()) {
        if (g_prof.pool.size() >= 65536) prof_drain();
        else {
            ProfRec r; r.kind = kind;
            if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess) { g_prof.on = false; return; }
            g_prof.pool.push_back(r);
        }
    }
    g_prof.pool[g_prof.used].kind = kind;
    (void)hipEventRecord(g_prof.pool[g_prof.used].e0, stream);
    g_prof.pending = true;
}
void dn_prof_end(int kind, hipStream_t stream, double flops, double bytes) {
    if (!g_prof.on || !g_prof.pending) return;
    (void)hipEventRecord(g_prof.pool[g_prof.used].e1, stream);
    g_prof.used++;
    g_prof.pending = false;
    g_prof.launches[kind]++;
    g_prof.flops[kind] += flops;
    g_prof.bytes[kind] += bytes;
}
#endif

extern "C" {

int dn_prof_enable(int on) {
#ifndef DN_EMULATE
    g_prof.on = on != 0;
#endif
    return 0;
}
int dn_prof_reset(void) {
#ifndef DN_EMULATE
    prof_drain();
    for (int k = 0; k < DN_K_COUNT; ++k) { g_prof.ms[k] = g_prof.flops[k] = g_prof.bytes[k] = 0; g_prof.launches[k] = 0; }
#endif
    return 0;
}
/* out[0..3] = {milliseconds, launches, algorithmic flops, algorithmic bytes} summed since the last reset */
int dn_prof_read(int kind, double* out) {
    if (kind < 0 || kind >= DN_K_COUNT || !out) return DN_ERR_INVALID;
#ifndef DN_EMULATE
    prof_drain();
    out[0] = g_prof.ms[kind]; out[1] = (double)g_prof.launches[kind]; out[2] = g_prof.flops[kind]; out[3] = g_prof.bytes[kind];
#else
    out[0] = out[1] = out[2] = out[3] = 0;
#endif
    return 0;
}
const char* dn_prof_kind_name(int kind) {
    static const char* names[DN_K_COUNT] = {"rowgemm_kernel<*,1>", "rowgemm_kernel<*,2>", "tngemm_kernel", "spmm_kernel", "small", "chain_fwd_kernel", "chain_bwd_kernel", "diffuse_kernel", "tngemm_x3_multi_kernel", "tngemm_da_kernel", "backproject_kernel", "spectral_apply_kernel"};
    return (kind >= 0 && kind < DN_K_COUNT) ? names[kind] : "";
}

int dn_set_option(const char* name, int value) {
    if (!name) return DN_ERR_INVALID;
    for (int i = 0; i < O_COUNT; ++i) if (strcmp(name, g_opt[i].name) == 0) { g_opt[i].value.store(value, std::memory_order_relaxed); return 0; }
    return DN_ERR_INVALID;
}
int dn_get_option(const char* name, int* value) {
    if (!name || !value) return DN_ERR_INVALID;
    for (int i = 0; i < O_COUNT; ++i) if (strcmp(name, g_opt[i].name) == 0) { *value = opt(i); return 0; }
    return DN_ERR_INVALID;
}

int dn_version(void) { return 600; }
int dn_tile_rows(void) { return DN_TM; }
int dn_tn_target_chunks(void) { return 2 * dn_num_cus(); }
int dn_tn_target_chunks_k(int k_eig) { return (k_eig >= 256 ? 1 : 2) * dn_num_cus(); }
int dn_diffusion_plan_wgs(void) { return dn_num_cus(); }
int dn_diffusion_plan(const int32_t* sizes, int n_mesh, int n_wg, int n_groups, dn_tile_t* plan) {
    if (!sizes || !plan) return 0;
    return dn_diffuse_plan_host(sizes, n_mesh, n_wg, n_groups > 0 ? n_groups : opt(O_DIFFUSE_GROUPS), reinterpret_cast<DnTile*>(plan));
}

// ------------------------------------------------------------------ to_basis / from_basis
size_t dn_to_basis_workspace_bytes(const dn_mesh_batch_t* mb, int C) {
    return pad256((size_t)mb->n_chunks * mb->k_eig * C) + 512;
}
int dn_to_basis_f32(const dn_mesh_batch_t* mb, const float* x, int C, int use_mass, float* spec, void* ws, size_t ws_bytes,
                    void* stream) {
    Bump b(ws, ws_bytes);
    float* partial = b.f((size_t)mb->n_chunks * mb->k_eig * C);
    if (!b.ok) return DN_ERR_INVALID;
    DN_CHECK(to_basis_partials(mb, x, C, use_mass != 0, partial, S(stream)));
    return dn_launch_spec_fwd(partial, mb->mesh_chunk_off, mb->evals, nullptr, spec, nullptr, mb->n_mesh, mb->k_eig, C, S(stream));
}
int dn_from_basis_f32(const dn_mesh_batch_t* mb, const float* spec, int C, int scale_rows_by_mass, float* out, void* stream) {
    return from_basis(mb, spec, C, out, nullptr, scale_rows_by_mass != 0, S(stream));
}

// ------------------------------------------------------------------ learned-time diffusion
size_t dn_diffusion_workspace_bytes(const dn_mesh_batch_t* mb, int C) {
    size_t n = pad256((size_t)mb->n_chunks * mb->k_eig * C) + pad256((size_t)mb->n_mesh * mb->k_eig * C) +
               pad256((size_t)dn_spec_bwd_dt_rows(mb->n_mesh, mb->k_eig) * C) + 512;
    if (diffuse_ok(mb, C)) n += pad256(diffuse_ws_floats(mb)) + pad256((size_t)diffuse_dt_rows(mb) * C);      // (the hybrid form uses both sets)
    if (bw_ok(mb, C)) n += pad256(dn_backproject_wide_ws_floats(mb->n_mesh, mb->k_eig, C));                   // (K = C = 256: the split spectrum)
    return n;
}
int dn_diffusion_fwd_f32(const dn_mesh_batch_t* mb, const float* x, const float* time, int C, float* xs, float* xd,
                         void* ws, size_t ws_bytes, void* stream) {
    Bump b(ws, ws_bytes);
    if (diffuse_ok(mb, C) && time && diffuse_aligned(x, xd, xs, mb->evecs, time, mb->evals)) {      // one launch
        float* dws = b.f(diffuse_ws_floats(mb));
        if (!b.ok) return DN_ERR_INVALID;
        DfLaunch L = diffuse_new(mb, dws);
        L.x = x; L.time = time; L.xs = xs; L.out = xd;
        return dn_launch_diffuse(L, S(stream));
    }
    float* partial = b.f((size_t)mb->n_chunks * mb->k_eig * C);
    float* ys = b.f((size_t)mb->n_mesh * mb->k_eig * C);
    float* wide_ws = bw_ok(mb, C) ? b.f(dn_backproject_wide_ws_floats(mb->n_mesh, mb->k_eig, C)) : nullptr;
    if (!b.ok) return DN_ERR_INVALID;
    DN_CHECK(to_basis_partials(mb, x, C, true, partial, S(stream)));
    DN_CHECK(dn_launch_spec_fwd(partial, mb->mesh_chunk_off, mb->evals, time, xs, ys, mb->n_mesh, mb->k_eig, C, S(stream)));
    return from_basis(mb, ys, C, xd, nullptr, false, S(stream), F16(), wide_ws);
}
int dn_diffusion_bwd_f32(const dn_mesh_batch_t* mb, const float* d_xd, const float* xs, const float* time, int C,
                         const float* d_x_add, float* d_x, float* d_time, void* ws, size_t ws_bytes, void* stream) {
    Bump b(ws, ws_bytes);
    if (diffuse_ok(mb, C) && xs && diffuse_aligned(d_xd, d_x, xs, mb->evecs, time, d_x_add)) {      // one launch + the d_t row sum
        float* dws = b.f(diffuse_ws_floats(mb));
        float* dtp = b.f((size_t)diffuse_dt_rows(mb) * C);
        if (!b.ok) return DN_ERR_INVALID;
        DfLaunch L = diffuse_new(mb, dws);
        L.bwd = 1; L.x = d_xd; L.time = time; L.xs = const_cast<float*>(xs); L.out = d_x; L.add = d_x_add; L.dt_part = dtp;
        DN_CHECK(dn_launch_diffuse(L, S(stream)));
        return dn_launch_reduce(dtp, d_time, diffuse_dt_rows(mb), C, C, S(stream));
    }
    float* partial = b.f((size_t)mb->n_chunks * mb->k_eig * C);
    float* dxs = b.f((size_t)mb->n_mesh * mb->k_eig * C);
    const int dt_rows = dn_spec_bwd_dt_rows(mb->n_mesh, mb->k_eig);
    float* dtp = b.f((size_t)dt_rows * C);
    if (!b.ok) return DN_ERR_INVALID;
    DN_CHECK(to_basis_partials(mb, d_xd, C, false, partial, S(stream)));
    if (dn_spec_bwd_fused_ok(partial, time, xs, dxs, dtp, C)) {
        DN_CHECK(dn_launch_spec_bwd_fused(partial, mb->mesh_chunk_off, mb->evals, time, xs, dxs, dtp, mb->n_mesh, mb->k_eig, C, S(stream), nullptr));
        DN_CHECK(dn_launch_reduce(dtp, d_time, dt_rows, C, C, S(stream)));
    } else {
        DN_CHECK(dn_launch_seg_reduce(partial, mb->mesh_chunk_off, mb->n_mesh, 0, dxs, (long long)mb->k_eig * C, S(stream)));
        DN_CHECK(dn_launch_spec_bwd(dxs, mb->evals, time, xs, dtp, mb->n_mesh, mb->k_eig, C, S(stream)));
        DN_CHECK(dn_launch_reduce(dtp, d_time, mb->n_mesh, C, C, S(stream)));
    }
    return from_basis(mb, dxs, C, d_x, d_x_add, true, S(stream));   // d_x_add may be NULL
}

// ------------------------------------------------------------------ spectral-gradient operands (dn_spectral.hip)
int dn_spectral_grad_supported(int k_eig, int C) { return dn_chain_sg_eligible(C, k_eig, 1, C >= 256 ? 2 : 1) ? 1 : 0; }
int dn_spectral_units(const int32_t* sizes, int n_mesh, int k_eig, dn_tile_t* units) {
    if (!sizes || n_mesh <= 0) return 0;
    return dn_sg_units_host(sizes, n_mesh, k_eig, reinterpret_cast<DnTile*>(units));
}
size_t dn_spectral_pack_bytes(int n_units, int k_eig) { return dn_sg_pack_elems(n_units, k_eig) * sizeof(uint4); }
size_t dn_spectral_pack_workspace_bytes(const dn_mesh_batch_t* mb) { return 2 * pad256((size_t)mb->v_total * mb->k_eig) + 512; }
int dn_spectral_pack_f32(const dn_mesh_batch_t* mb, const dn_tile_t* units, int n_units, void* sg_pack, float* sg_amax, void* ws, size_t ws_bytes,
                         void* stream) {
    if (!mb || !units || n_units <= 0 || !sg_pack || !sg_amax || !mb->evecs || !mb->g_rowptr || !mb->g_col || !mb->g_vx || !mb->g_vy ||
        mb->k_eig % 32 != 0 || mb->k_eig > 256 || !al16(mb->evecs) || !al16(sg_pack))
        return DN_ERR_INVALID;
    Bump b(ws, ws_bytes);
    float* gpx = b.f((size_t)mb->v_total * mb->k_eig);
    float* gpy = b.f((size_t)mb->v_total * mb->k_eig);
    if (!b.ok) return DN_ERR_INVALID;
    return dn_launch_sg_pack(T(units), n_units, mb->n_mesh, mb->k_eig, mb->evecs, mb->g_rowptr, mb->g_col, mb->g_vx, mb->g_vy, gpx, gpy, sg_amax,
                             reinterpret_cast<uint4*>(sg_pack), S(stream));
}

// ------------------------------------------------------------------ gradient apply
int dn_grad_apply_fwd_f32(const dn_mesh_batch_t* mb, const float* x, int C, float* gx, float* gy, void* stream) {
    return grad_apply_fwd(mb, x, C, gx, gy, S(stream));
}
int dn_grad_apply_bwd_f32(const dn_mesh_batch_t* mb, const float* d_gx, const float* d_gy, const float* add, int C,
                          float* d_x, void* stream) {
    return grad_apply_bwd(mb, d_gx, d_gy, add, C, d_x, S(stream));
}

// ------------------------------------------------------------------ gradient features
size_t dn_gradfeat_workspace_bytes(const dn_mesh_batch_t* mb, int C) {
    return pad256((size_t)mb->n_chunks * 4 * C * C) + pad256((size_t)mb->v_total * C) + pad256((size_t)4 * C * C) + 512;
}
int dn_gradfeat_fwd_f32(const dn_mesh_batch_t* mb, const float* gx, const float* gy, const float* A_re, const float* A_im,
                        int C, float* g, float* bre, float* bim, void* stream) {
    return gradfeat_fwd(mb, gx, gy, A_re, A_im, C, g, bre, bim, S(stream));
}
int dn_gradfeat_bwd_f32(const dn_mesh_batch_t* mb, const float* d_g, const float* g, const float* gx, const float* gy,
                        const float* bre, const float* bim, const float* A_re, const float* A_im, int C,
                        float* d_gx, float* d_gy, float* dA_re, float* dA_im, void* ws, size_t ws_bytes, void* stream) {
    Bump b(ws, ws_bytes);
    float* partial = b.f((size_t)mb->n_chunks * 4 * C * C);
    float* ddots = b.f((size_t)mb->v_total * C);
    float* psum = b.f((size_t)4 * C * C);
    if (!b.ok) return DN_ERR_INVALID;
    // d_dots = d_g * (1 - g^2)   (the fused block folds this into the epilogue of the d_h0 product)
    DN_CHECK(dn_launch_dtanh(d_g, g, ddots, (long long)mb->v_total * C, S(stream)));
    DN_CHECK(gradfeat_bwd_weights(mb, ddots, gx, gy, C, dA_re, A_im ? dA_im : nullptr, partial, psum, S(stream)));
    return gradfeat_bwd_inputs(mb, ddots, gx, gy, bre, bim, A_re, A_im, C, d_gx, d_gy, S(stream));
}

// ------------------------------------------------------------------ nn.Linear
static size_t linear_partial_elems(const dn_mesh_batch_t* mb, int C_in, int C_out) {
    size_t part = (size_t)mb->n_chunks * C_in * C_out;
    if (C_in <= 16) {   // thin-input route: smalln partials, and a 4-column product for the bias column sums
        if ((size_t)DN_SMALLN_BLOCKS * C_in * C_out > part) part = (size_t)DN_SMALLN_BLOCKS * C_in * C_out;
        if ((size_t)mb->n_chunks * C_out * 4 > part) part = (size_t)mb->n_chunks * C_out * 4;
    }
    return part;
}
size_t dn_linear_workspace_bytes(const dn_mesh_batch_t* mb, int C_in, int C_out) {
    return pad256(linear_partial_elems(mb, C_in, C_out)) + pad256((size_t)mb->n_chunks * C_out) + 512;
}
// max |t| of a tensor nobody tracked while producing it: one measuring pass, accumulated into *word
static int measure_amax(const float* t, long long n, float* word, hipStream_t st) {
    AmaxJobs jobs; jobs.count = 0;
    jobs.push(t, n, word);
    return dn_launch_amax(jobs, st);
}
int dn_linear_fwd_amax_f32(const dn_mesh_batch_t* mb, const float* x, int C_in, const float* W, const float* b, int C_out,
                           int relu, const uint8_t* mask, float* out, float* out_amax, void* stream) {
    if (C_in <= 16 && C_out <= 1024 && !relu && !mask)   // thin contraction (first_lin: xyz / hks features): bandwidth-bound VALU kernel
        return dn_launch_smallk_rows(x, C_in, W, 0, b, C_out, out, mb->v_total, S(stream), out_amax);
    const float* xs[1] = {x};
    const int ws_[1] = {C_in};
    DN_CHECK(linear_fwd(mb, xs, ws_, 1, W, C_in, b, C_out, relu ? DN_EPI_BIAS_RELU : DN_EPI_STORE, mask, nullptr, out, S(stream)));
    return out_amax ? measure_amax(out, (long long)mb->v_total * C_out, out_amax, S(stream)) : 0;
}
int dn_linear_fwd_f32(const dn_mesh_batch_t* mb, const float* x, int C_in, const float* W, const float* b, int C_out,
                      int relu, const uint8_t* mask, float* out, void* stream) {
    return dn_linear_fwd_amax_f32(mb, x, C_in, W, b, C_out, relu, mask, out, nullptr, stream);
}
int dn_linear_bwd_f32(const dn_mesh_batch_t* mb, const float* d_out, const float* x, const float* W, int C_in, int C_out,
                      float* d_x, float* dW, float* db, void* ws, size_t ws_bytes, void* stream) {
    return dn_linear_bwd_amax_f32(mb, d_out, x, W, C_in, C_out, d_x, dW, db, nullptr, ws, ws_bytes, stream);
}
int dn_linear_bwd_amax_f32(const dn_mesh_batch_t* mb, const float* d_out, const float* x, const float* W, int C_in, int C_out,
                           float* d_x, float* dW, float* db, float* d_x_amax, void* ws, size_t ws_bytes, void* stream) {
    Bump b(ws, ws_bytes);
    float* partial = b.f(linear_partial_elems(mb, C_in, C_out));
    float* colsum = b.f((size_t)mb->n_chunks * C_out);
    if (!b.ok) return DN_ERR_INVALID;
    const float* ins[1] = {x};
    const int ws_[1] = {C_in};
    int nthin = mb->n_chunks < 512 ? mb->n_chunks : 512;    // workgroups of the thin products (their partials fit the regions above)
    if (C_in <= 32 && C_out >= 32 && C_out % 4 == 0 && al16(d_out) && (size_t)nthin * C_in * C_out <= linear_partial_elems(mb, C_in, C_out)) {
        // thin input (first_lin: xyz or hks features): dW = d_out^T x and db = column sums of d_out from one streaming pass over d_out per 8 inputs
        DN_CHECK(dn_launch_thin_tn(d_out, C_out, x, C_in, mb->v_total, 0, 1, dW, db, partial, colsum, nthin, S(stream)));
    } else if (C_out <= 32 && C_in >= 32 && C_in % 4 == 0 && al16(x)) {
        // thin output (last_lin, up to 32 classes): dW = d_out^T x and db = column sums of d_out from one streaming pass over x per 8 classes
        DN_CHECK(dn_launch_thin_tn(x, C_in, d_out, C_out, mb->v_total, 1, 0, dW, db, partial, colsum, nthin, S(stream)));
    } else if (C_in <= 16) {
        // thin input (hks features, C_in = 16): dW = d_out^T x streams d_out once on the VALU
        DN_CHECK(dn_launch_smalln_tn(d_out, C_out, x, C_in, mb->v_total, dW, partial, DN_SMALLN_BLOCKS, S(stream)));
        if (db) {   // db[o] = sum_r d_out[r,o]: column sums of a minimal (4-column) split-V product
            TnArgs g = tn_new(mb);
            tn_a(g, d_out, nullptr, C_out, C_out);
            tn_b(g, d_out, nullptr, C_out < 4 ? C_out : 4, C_out);
            g.partial = partial; g.colsum = colsum; g.group = dn_tn_global_group(mb->n_chunks);
            tn_finish(g);
            DN_CHECK(dn_launch_tngemm(g, mb->n_chunks, S(stream)));
            DN_CHECK(dn_launch_reduce(colsum, db, dn_tn_npartial(mb->n_chunks, g.group), C_out, C_out, S(stream)));
        }
    } else {
        DN_CHECK(linear_bwd_weights(mb, d_out, C_out, ins, ws_, 1, dW, db, partial, colsum, S(stream)));
    }
    if (d_x) {
        if (C_out <= 16 && C_in <= 1024) {
            DN_CHECK(dn_launch_smallk_rows(d_out, C_out, W, 1, nullptr, C_in, d_x, mb->v_total, S(stream), d_x_amax));
        } else {
            DN_CHECK(linear_bwd_input(mb, d_out, C_out, W, C_in, 0, C_in, DN_EPI_STORE, nullptr, 1.f, d_x, S(stream)));
            if (d_x_amax) DN_CHECK(measure_amax(d_x, (long long)mb->v_total * C_in, d_x_amax, S(stream)));
        }
    }
    return 0;
}

// ------------------------------------------------------------------ fused DiffusionNetBlock
// ---- split-fp16 engine of the fused block (round 3): every dense product of the block takes two-term fp16 splits of its operands,
// scaled by powers of two from "amax words" -- device floats holding the largest magnitude of a tensor, written by the kernel that
// produced it (atomic max in its epilogue) or measured by one small launch for the weights.  The words of a call live in its
// workspace; those of the saved activations in dn_block_saved_t.amax; the block input / output (and d_out / d_x) words travel
// through dn_block_params_t / dn_block_grads_t, and are measured by the call when the caller passes none.
// Eligible: aligned operands and widths the split kernels take (C, K and every MLP width multiples of 32 and >= 128); anything else
// -- and option "f16" = 0 -- runs the split-bf16 / exact-f32 engines exactly as before.
enum { AW_IN = 0, AW_YS, AW_MISC, AW_WA, AW_W0, AW_D0 = AW_W0 + DN_MAX_MLP_LAYERS, AW_COUNT = AW_D0 + DN_MAX_MLP_LAYERS + 1 };   // call-local words:
// [0, AW_WA) and [AW_D0, ..) are accumulated into (zeroed by the start-of-call launch), [AW_WA, AW_D0) are STORED by that same launch -- the two
// sets must not overlap: the zeroing workgroup runs concurrently with the storing ones (it once wiped the weight magnitude in 1 call of 2000)
enum { SW_X = 0, SW_XD, SW_G, SW_H0 };                                                                                          // saved words
static_assert(SW_H0 + DN_MAX_MLP_LAYERS <= DN_BLOCK_AMAX_WORDS, "saved amax words");
static bool block_f16_ok(const dn_mesh_batch_t* mb, const dn_block_params_t* p) {
    if (!opt(O_F16) || (p->flags & DN_BLOCK_NO_F16)) return false;
    auto ok = [](int w) { return w >= 128 && w % 32 == 0; };
    if (!ok(p->C) || !ok(mb->k_eig)) return false;
    for (int j = 1; j < p->n_mlp; ++j) if (!ok(p->widths[j])) return false;
    return true;
}
static size_t amax_ws(void) { return pad256(AW_COUNT + DN_BLOCK_AMAX_WORDS + 2); }
// The chained row kernels (dn_chain.hip forward, dn_chain_bwd.hip backward) take the block's row work -- gradient gather, gradient features,
// MiniMLP and their gradients -- when the shapes are the ones they are written for and the magnitude words exist.  Option "chain" = 0
// keeps the unfused launches (dn_set_option: the tests flip it at run time).
// kind: 0 = forward without saved activations (inference), 1 = forward saving activations (training), 2 = backward.
// Measured on MI355X (tools/kbench, block at C = K = 128, chained / unfused, us; round 5, profiles/r05_chain_hh_sweep.txt -- the chained kernels
// with one 16-row half per wave where that is faster, see chain_hh below):
//     vertices        7k        20k       40k       80k       160k
//     inference     64/86     88/121   122/188   203/299   365/509
//     training      66/90     99/120   145/194   230/307   447/507
//     backward     114/139   185/243   290/376   471/574   822/933
// The chained kernels win at every size in every role since the one-half-per-wave form exists (round 4: the training forward lost between
// 20k and 80k rows and was only taken from 100k); "chain_min_rows" / "chain_small_rows" (both 0 now) can still carve a window for the
// unfused training forward: the tests' "mixed" mode uses them.
static bool chain_aligned(const dn_block_params_t* p, const dn_block_saved_t* sv, const float* a, const float* b_, const float* c = nullptr, const float* d = nullptr) {
    bool ok = al16(a) && al16(b_) && al16(c) && al16(d) && al16(p->A_re) && al16(p->A_im) && al16(p->time);
    for (int j = 0; j < p->n_mlp; ++j) ok = ok && al16(p->W[j]) && al16(p->b[j]) && al16(p->mask[j]);
    if (sv) {
        ok = ok && al16(sv->xs) && al16(sv->xd) && al16(sv->gx) && al16(sv->gy) && al16(sv->g) && al16(sv->bre) && al16(sv->bim);
        for (int j = 0; j < p->n_mlp; ++j) ok = ok && al16(sv->h[j]);
    }
    return ok;
}
// 16-row halves per wave of the chained kernels: 2 (a weight fragment read feeds two MFMAs) for batches that fill the device, 1 for small ones
// (twice the waves, half the serial product chain each): one ~7k-vertex mesh per step is 219 32-row waves on 1024 SIMDs.  Option "chain_hh"
// forces either (tests, A/B).
// Measured (tools/kbench, block forward / backward, us, HH = 2 -> 1; profiles/r05_chain_hh_sweep.txt): 7k rows 90 -> 66 / 134 -> 114,
// 20k 117 -> 99 / 198 -> 185, 40k 168 -> 145 / 319 -> 290, 80k 267 -> 230 / 502 -> 471, 160k 457 -> 447 / 822 -> 833: the forward takes one
// half per wave at every size measured, the backward up to ~100k rows.
static int chain_hh(const dn_mesh_batch_t* mb, bool backward = false, int C = 128) {
    if (C >= 256) return 2;       // (the one form of the C = 256 forward)
    const int f = opt(O_CHAIN_HH);
    if (f == 1 || f == 2) return f;
    return mb->v_total <= (backward ? 100000 : 262144) ? 1 : 2;
}
static bool block_chain_ok(const dn_mesh_batch_t* mb, const dn_block_params_t* p, int kind) {
    if (!opt(O_CHAIN) || !opt(O_F16) || (p->with_grad && !mb->grad_norm)) return false;      // "f16" = 0: split-bf16 engine everywhere (A/B runs)
    if (p->flags & (DN_BLOCK_NO_CHAIN | DN_BLOCK_NO_F16)) return false;                        // per-call engine choice (include/diffnet_hip.h)
    if (kind == 1 && mb->v_total < opt(O_CHAIN_MIN_ROWS) && mb->v_total > opt(O_CHAIN_SMALL_ROWS)) return false;
    return dn_chain_eligible(p->C, p->n_mlp, p->widths, p->with_grad, mb->g_nnz, mb->v_total, kind == 2);
}
// The spectral-gradient form of the chained forward (dn_spectral.hip, dn_chain.hip KE > 0): batches that carry the packed operands, shapes the
// kernel is instantiated for, one 16-row half per wave (the forward's form up to 262144 rows); not with the one-launch diffusion operator.
// Option "spectral_grad": 1 (default) = the inference forward at every size, the training forward up to 65536 rows; 2 = both at every size; 0 = never.
// Measured (tools/kbench block_inf / block_fwd, us, spectral / gather form, same box; profiles/r06_sg_sweep.txt):
//     vertices      7k          20k         40k         80k         160k        240k       64 x 2k
//     inference   58.9/65.3   75.2/82.4   103/117     169/185     323/334     448/476     274/276
//     training    60.8/66.7   88.5/92.3   122/132     212/203     389/390     545/581     326/317
// The inference forward gains at every size (no back-projection launch, no xd round trip); the training forward also writes xd, gx, gy from the
// kernel (13 instead of 10 arrays of [V, C] through it) and is level with back-projection + gather from ~80k rows on (bench.py headline, two
// runs each on one box: 30.82 / 30.85 M vertices/s against 31.02 / 31.02).
static bool block_sg_ok(const dn_mesh_batch_t* mb, const dn_block_params_t* p, int kind) {
    const int o = (p->flags & DN_BLOCK_NO_SPECTRAL_GRAD) ? 0 : ((p->flags & DN_BLOCK_SPECTRAL_GRAD_ALWAYS) ? 2 : opt(O_SPECTRAL_GRAD));
    // (C = K = 256, the two-launch form: measured SLOWER than back-projection + gather -- BASELINE config 4 at 28.2 M vertices/s against 29.1 M: its
    // spectral launch reads 0.61 GB of operands and writes 0.61 GB of xd / gx / gy that the chain reads back, 366 us where ~250 would pay -- and
    // therefore only taken when asked for: option value 2 / DN_BLOCK_SPECTRAL_GRAD_ALWAYS)
    return o && kind < 2 && (o >= 2 || (p->C < 256 && (kind == 0 || mb->v_total <= 65536))) && mb->sg_pack && mb->sg_units && mb->sg_amax && mb->sg_n_units > 0 && al16(mb->sg_pack) && al16(mb->sg_amax) &&
           mb->sg_n_units <= 100 * dn_num_cus() &&      // (a workgroup's pass table lives in LDS: DN_CH_SG_MAXP = 64 passes of 2 x CUs workgroups)
           block_chain_ok(mb, p, kind) && dn_chain_sg_eligible(p->C, mb->k_eig, p->with_grad, chain_hh(mb, false, p->C)) && !diffuse_ok(mb, p->C);
}
static size_t sg_piece_floats(const dn_mesh_batch_t* mb, int C) { return (size_t)mb->n_mesh * (mb->k_eig / 32) * (2 * (C / 16) * 64) * 4; }
// Product classes of the block; option "f16_mask" (diagnostic) selects which of them run on the split-fp16 engine.
// Default: the row products (gradient features, MLP, input gradients, backward back-projection).  Not the split-V projections
// evecs^T x (their lock-step kernel gains nothing: 55 -> 54.6 us, and leaving them out spares the transposed gather a magnitude
// pass that cost it +60 %), and not the FORWARD back-projection x_diffuse = evecs * spectrum: its output is what the sparse gradient operators
// difference (gx = G_X x_diffuse, row sums ~0), which amplifies its rounding noise, and its B operand -- a spectrum whose entries
// decay over many orders of magnitude with the eigenvalue -- is the one tensor a single power-of-two scale serves badly.  Measured on
// the trained-checkpoint golden (error against fp64, fp32 reference = 8.4e-5 on the worst tensor): every class on fp16 3.2e-4, every
// class but this one 6e-5 ... 1.0e-4 -- the level of the split-bf16 engine and of the reference itself.
static int f16_mask(void) { return opt(O_F16_MASK); }
static F16 f16_if(int bit, const F16& f) { if (f16_mask() & bit) return f; F16 r; r.o = f.o; return r; }   // (the magnitude of the output is still recorded)

int dn_block_tracks_amax(const dn_mesh_batch_t* mb, const dn_block_params_t* p, int call) {
    if (!mb || !block_params_ok(p) || call < 0 || call > 2) return 0;
    if (block_f16_ok(mb, p)) return 1;
    return (call < 2 && block_chain_ok(mb, p, call)) ? 1 : 0;      // (the chained backward alone does not produce max |d_x|)
}
size_t dn_block_fwd_workspace_bytes(const dn_mesh_batch_t* mb, const dn_block_params_t* p, int with_saved) {
    if (!block_params_ok(p)) return 0;
    const size_t VC = (size_t)mb->v_total * p->C;
    size_t n = pad256((size_t)mb->n_chunks * mb->k_eig * p->C) + pad256((size_t)mb->n_mesh * mb->k_eig * p->C) + amax_ws() + 512;
    if (block_chain_ok(mb, p, with_saved ? 1 : 0)) n += pad256(dn_chain_ws_bytes(p->C, p->with_grad, p->with_rot, p->n_mlp) / sizeof(float));
    if (block_sg_ok(mb, p, with_saved ? 1 : 0)) n += pad256(sg_piece_floats(mb, p->C)) + pad256((size_t)2 * mb->n_mesh);
    if (diffuse_ok(mb, p->C)) n += pad256(diffuse_ws_floats(mb));
    if (bw_ok(mb, p->C)) n += pad256(dn_backproject_wide_ws_floats(mb->n_mesh, mb->k_eig, p->C));
    if (!with_saved) {
        n += pad256((size_t)mb->n_mesh * mb->k_eig * p->C) + 4 * pad256(VC);          // xs, xd, gx, gy, g
        n += 2 * pad256((size_t)mb->v_total * max_width(p));                            // hidden ping-pong
    }
    return n;
}
int dn_block_fwd_f32(const dn_mesh_batch_t* mb, const dn_block_params_t* p, const float* x, float* out,
                     const dn_block_saved_t* sv, void* ws, size_t ws_bytes, void* stream) {
    if (!block_params_ok(p)) return DN_ERR_INVALID;
    hipStream_t st = S(stream);
    const int C = p->C, K = mb->k_eig;
    const size_t VC = (size_t)mb->v_total * C;
    Bump b(ws, ws_bytes);
    float* partial = b.f((size_t)mb->n_chunks * K * C);
    float* ys = b.f((size_t)mb->n_mesh * K * C);
    float* aw = b.f(AW_COUNT + DN_BLOCK_AMAX_WORDS + 2);
    const bool chain = block_chain_ok(mb, p, sv ? 1 : 0) && (!sv || sv->amax) && chain_aligned(p, sv, x, out);
    float* chain_ws = block_chain_ok(mb, p, sv ? 1 : 0) ? b.f(dn_chain_ws_bytes(p->C, p->with_grad, p->with_rot, p->n_mlp) / sizeof(float)) : nullptr;
    const bool sg_ws = block_sg_ok(mb, p, sv ? 1 : 0);
    float* ysp = sg_ws ? b.f(sg_piece_floats(mb, C)) : nullptr;          // the scaled spectrum as weight pieces + its per-mesh magnitudes
    float* ysa = sg_ws ? b.f((size_t)2 * mb->n_mesh) : nullptr;
    const bool sg = chain && sg_ws;                                       // xd, gx, gy computed inside the chained kernel (dn_spectral.hip)
    float* diffuse_ws = diffuse_ok(mb, C) ? b.f(diffuse_ws_floats(mb)) : nullptr;
    float* wide_ws = bw_ok(mb, C) ? b.f(dn_backproject_wide_ws_floats(mb->n_mesh, K, C)) : nullptr;   // the spectrum split into ring pieces (K = C = 256)
    float *xs, *xd, *gx = nullptr, *gy = nullptr, *gf = nullptr, *bre = nullptr, *bim = nullptr;
    float* hbuf[2] = {nullptr, nullptr};
    if (sv) {
        xs = sv->xs; xd = sv->xd; gx = sv->gx; gy = sv->gy; gf = sv->g; bre = sv->bre; bim = sv->bim;
    } else {
        xs = b.f((size_t)mb->n_mesh * K * C); xd = b.f(VC);
        gx = b.f(VC); gy = b.f(VC); gf = b.f(VC);
        hbuf[0] = b.f((size_t)mb->v_total * max_width(p));
        hbuf[1] = b.f((size_t)mb->v_total * max_width(p));
    }
    if (!b.ok) return DN_ERR_INVALID;

    // ---- operand magnitudes for the split-fp16 engine
    const bool f16 = block_f16_ok(mb, p) && (!sv || sv->amax);          // split-fp16 engine for the unfused products
    const bool words = f16 || chain;                                      // magnitude words are tracked by this call
    float* sw = sv ? sv->amax : aw + AW_COUNT;                 // magnitudes of the saved activations (kept for the backward)
    const float *x_amax = nullptr, *ev_amax = nullptr, *ms_amax = nullptr;
    ChainPrepArgs pa; memset(&pa, 0, sizeof(pa));
    int chain_np = 0, chain_ngf = 0;
    const bool use_chain = chain;
    if (!words && p->clamp_time) {           // no bookkeeping launch on this path: the clamp is the launch
        AmaxInit in; memset(&in, 0, sizeof(in));
        in.clamp_p = const_cast<float*>(p->time); in.clamp_n = C; in.clamp_min = DN_MIN_TIME;
        DN_CHECK(dn_launch_amax_init(in, st));
    }
    if (words) {
        // one launch: weight magnitudes (stored), the words the kernels below accumulate into zeroed, the input's word forwarded to the
        // saved set (the backward multiplies by x again).  With the chained row kernel that launch is its weight-preparation kernel.
        AmaxInit in; memset(&in, 0, sizeof(in));
        if (!use_chain) {
            if (p->with_grad) { in.jobs.push(p->A_re, (long long)C * C, aw + AW_WA); if (p->with_rot) in.jobs.push(p->A_im, (long long)C * C, aw + AW_WA); }
            for (int j = 0; j < p->n_mlp && in.jobs.count < DN_AMAX_MAX_JOBS; ++j) in.jobs.push(p->W[j], (long long)p->widths[j] * p->widths[j + 1], aw + AW_W0 + j);
        }
        in.zero_range(aw, AW_WA);                                                  // AW_IN, AW_YS, AW_MISC
        in.zero_range(aw + AW_D0, DN_MAX_MLP_LAYERS + 1 + DN_BLOCK_AMAX_WORDS + 2);
        if (sv) in.zero_range(sw, DN_BLOCK_AMAX_WORDS);
        in.zero_range(p->out_amax, 1);
        x_amax = p->x_amax;
        ev_amax = mb->evecs_amax; ms_amax = mb->mass_amax;
        const bool measure = !x_amax || (f16 && (!ev_amax || !ms_amax));
        if (x_amax && sv && !measure) { in.copy_src = x_amax; in.copy_dst = sw + SW_X; }
        if (p->clamp_time) { in.clamp_p = const_cast<float*>(p->time); in.clamp_n = C; in.clamp_min = DN_MIN_TIME; }     // layers.py:48-49, in this launch
        if (use_chain) {
            const int NK = C / 32;
            auto piece = [&](const float* Wm, const float* Wm2, float* word, int ld, int col0) {
                ChainPrepPiece& q = pa.pc[chain_np++]; q.W = Wm; q.W2 = Wm2; q.amax = word; q.ld = ld; q.col0 = col0; };
            if (p->with_grad)
                for (int T = 0; T < NK; ++T) {      // (streamed once per 16-row half of a wave by the kernel)
                    piece(p->A_re, p->with_rot ? p->A_im : nullptr, aw + AW_WA, C, 32 * T);
                    if (p->with_rot) piece(p->A_im, p->A_re, aw + AW_WA, C, 32 * T);
                }
            chain_ngf = chain_np;
            for (int sg = 0; sg < (p->with_grad ? 3 : 2); ++sg) {      // layer 0: the g segment first (columns 2C..), then x, xd
                const int seg = p->with_grad ? (sg + 2) % 3 : sg;
                for (int T = 0; T < NK; ++T) piece(p->W[0], nullptr, aw + AW_W0, p->widths[0], seg * C + 32 * T);
            }
            for (int j = 1; j < p->n_mlp; ++j)
                for (int T = 0; T < NK; ++T) piece(p->W[j], nullptr, aw + AW_W0 + j, C, 32 * T);
            pa.out = reinterpret_cast<uint4*>(chain_ws);
            for (int r = 0; r < in.nzero; ++r) pa.zero_range(in.zero[r], in.zero_n[r]);
            pa.copy_src = in.copy_src; pa.copy_dst = in.copy_dst;
            pa.clamp_p = in.clamp_p; pa.clamp_n = in.clamp_n; pa.clamp_min = in.clamp_min;
            DN_CHECK(dn_launch_chain_prep(pa, chain_np, C, st));
        } else {
            DN_CHECK(dn_launch_amax_init(in, st));
        }
        if (measure) {   // a caller without magnitudes (plain C users, the first block of a net): one extra pass over what is missing
            AmaxJobs jobs; jobs.count = 0;
            float* fx = sv ? sw + SW_X : aw + AW_IN;
            if (!x_amax) { jobs.push(x, (long long)VC, fx); x_amax = fx; }
            if (f16 && !ev_amax) { jobs.push(mb->evecs, (long long)mb->v_total * K, aw + AW_COUNT + DN_BLOCK_AMAX_WORDS); ev_amax = aw + AW_COUNT + DN_BLOCK_AMAX_WORDS; }
            if (f16 && !ms_amax) { jobs.push(mb->mass, (long long)mb->v_total, aw + AW_COUNT + DN_BLOCK_AMAX_WORDS + 1); ms_amax = aw + AW_COUNT + DN_BLOCK_AMAX_WORDS + 1; }
            DN_CHECK(dn_launch_amax(jobs, st));
            if (p->x_amax && sv) DN_CHECK((int)hipMemcpyAsync(sw + SW_X, p->x_amax, sizeof(float), hipMemcpyDeviceToDevice, st));
        }
    }
    auto W = [&](int j) { return (const float*)(aw + AW_W0 + j); };

    // diffusion (layers.py:210): one persistent launch (dn_diffuse.hip) when the batch carries its plan -- both products on the 3-term
    // engine, as the three-launch form runs them by default -- else projection, spectral step, back-projection
    if (diffuse_ws && !(f16 && (f16_mask() & (F16_TOB | F16_FROMB))) && diffuse_aligned(x, xd, xs, mb->evecs, p->time, mb->evals)) {
        DfLaunch L = diffuse_new(mb, diffuse_ws);
        L.x = x; L.time = p->time; L.xs = xs; L.out = xd;
        L.out_amax = words ? sw + SW_XD : nullptr;                           // (the chained kernel scales xd by its magnitude)
        DN_CHECK(dn_launch_diffuse(L, st));
    } else {
        {
            F16 f;
            if (f16) { f = f16_of(ev_amax, x_amax); f.b.mul = ms_amax; }
            DN_CHECK(to_basis_partials(mb, x, C, true, partial, st, f16_if(F16_TOB, f)));
        }
        DN_CHECK(dn_launch_spec_fwd(partial, mb->mesh_chunk_off, mb->evals, p->time, xs, ys, mb->n_mesh, K, C, st,
                                    (f16 && (f16_mask() & F16_FROMB)) ? aw + AW_YS : nullptr));   // only the split-fp16 back-projection needs max |ys|
        if (sg) {
            // no back-projection launch: the chained kernel multiplies [evecs | gradX evecs | gradY evecs] by the spectrum itself
            DN_CHECK(dn_launch_spec_pieces(ys, mb->n_mesh, K, C, reinterpret_cast<uint4*>(ysp), ysa, st));
        } else {
            F16 fb;
            if (f16) fb = f16_if(F16_FROMB, f16_of(ev_amax, aw + AW_YS, sw + SW_XD));
            else if (words) fb.o = sw + SW_XD;                                  // (the chained kernel scales xd by its magnitude)
            DN_CHECK(from_basis(mb, ys, C, xd, nullptr, false, st, fb, wide_ws));
        }
    }
    if (use_chain) {   // gather -> gradient features -> MiniMLP + residual in one launch (layers.py:213-239)
        ChainArgs ca; memset(&ca, 0, sizeof(ca));
        ca.rowptr = mb->g_rowptr; ca.col = mb->g_col; ca.vx = mb->g_vx; ca.vy = mb->g_vy;
        ca.x = x; ca.xd = xd; ca.V = mb->v_total;
        ca.with_grad = p->with_grad; ca.with_rot = p->with_rot; ca.n_mlp = p->n_mlp;
        ca.wp = reinterpret_cast<const uint4*>(chain_ws);
        // pieces the kernel streams once per 16-row half: the gradient-feature ones, and at C = 256 layer 0's g segment behind them (dn_chain.hip, G0)
        ca.n_gf = chain_ngf + ((C >= 256 && p->with_grad) ? C / 32 : 0);
        ca.wa_amax = aw + AW_WA;
        for (int j = 0; j < p->n_mlp; ++j) {
            ca.w_amax[j] = aw + AW_W0 + j; ca.bias[j] = p->b[j];
            if (j < p->n_mlp - 1) {
                ca.mask[j] = p->mask[j + 1];
                ca.seed[j] = p->mask[j + 1] ? 0ull : layer_seed(p->drop_seed, j + 1);
                ca.h[j] = sv ? sv->h[j] : nullptr;
                ca.h_amax[j] = sw + SW_H0 + j;
            }
        }
        ca.seed_dev = (const unsigned long long*)p->drop_seed_dev;
        if (p->with_grad && sv) { ca.gx = gx; ca.gy = gy; ca.g = gf; ca.bre = bre; ca.bim = bim; }
        if (sg && C >= 256) { ca.gx = gx; ca.gy = gy; }      // (the C = 256 spectral form hands gx, gy from its spectral phase to its gradient-feature stage through memory)
        // (Measured and rejected: gathering gx, gy with the stand-alone CSR kernel first and letting the training chain read them -- the chained
        // kernel drops from 357 to 293 us, but the 87 us gather launch more than eats it: block forward 480 vs 465 us, profiles/r04_pregather.txt.)
        ca.out = out;
        ca.x_amax = x_amax; ca.xd_amax = sw + SW_XD; ca.grad_norm = mb->grad_norm;
        ca.g_amax = sw + SW_G; ca.out_amax = p->out_amax;
        if (sg) {
            // (C = 256: the kernel writes xd to the buffer and reads it back in layer 0; C <= 128: xd stays in registers, written only when saved)
            ca.xd = C >= 256 ? xd : nullptr;
            ca.sg_pack = reinterpret_cast<const uint4*>(mb->sg_pack); ca.sg_units = T(mb->sg_units); ca.sg_n_units = mb->sg_n_units; ca.sg_amax = mb->sg_amax;
            ca.sg_unit_rows = dn_sg_unit_rows(K); ca.sg_n_mesh = mb->n_mesh;
            ca.ysp = reinterpret_cast<const uint4*>(ysp); ca.ys_amax = ysa;
            ca.xd_out = (sv || C >= 256) ? xd : nullptr; ca.xd_amax_out = sw + SW_XD;
        }
        return dn_launch_chain_fwd(chain_np, ca, C, st, chain_hh(mb, false, C));
    }
    // gradient features (layers.py:213-226)
    if (p->with_grad) {
        DN_CHECK(grad_apply_fwd(mb, xd, C, gx, gy, st, f16 ? sw + SW_G : nullptr, f16 ? sw + SW_XD : nullptr));
        DN_CHECK(gradfeat_fwd(mb, gx, gy, p->A_re, p->with_rot ? p->A_im : nullptr, C, gf, bre, bim, st,
                              f16 ? f16_if(F16_GF, f16_of(sw + SW_G, aw + AW_WA)) : F16()));
    }
    // MiniMLP on [x | xd | g] + residual (layers.py:229-239)
    const float* in_ptr[3] = {x, xd, gf};
    int in_w[3] = {C, C, C};
    int nseg = p->with_grad ? 3 : 2;
    for (int j = 0; j < p->n_mlp; ++j) {
        const bool last = (j == p->n_mlp - 1);
        float* dst = last ? out : (sv ? sv->h[j] : hbuf[j & 1]);
        F16 f;
        if (f16) {
            f = f16_of(j == 0 ? x_amax : sw + SW_H0 + j - 1, W(j), last ? p->out_amax : sw + SW_H0 + j);
            if (j == 0) { f.a.p[1] = sw + SW_XD; f.a.c = p->with_grad ? 1.f : 0.f; }      // [x | xd | g], |g| = |tanh| <= 1
        }
        DN_CHECK(linear_fwd(mb, in_ptr, in_w, nseg, p->W[j], p->widths[j], p->b[j], p->widths[j + 1],
                            last ? DN_EPI_BIAS_RESID : DN_EPI_BIAS_RELU, last ? nullptr : p->mask[j + 1],
                            last ? x : nullptr, dst, st, last ? 0ull : layer_seed(p->drop_seed, j + 1), (const unsigned long long*)p->drop_seed_dev,
                            f16_if(F16_MLP, f)));
        in_ptr[0] = dst; in_w[0] = p->widths[j + 1]; nseg = 1;
    }
    return 0;
}

size_t dn_block_bwd_workspace_bytes(const dn_mesh_batch_t* mb, const dn_block_params_t* p) {
    if (!block_params_ok(p)) return 0;
    const size_t VC = (size_t)mb->v_total * p->C;
    size_t n = 2 * pad256((size_t)mb->v_total * max_width(p));     // d_a ping-pong
    n += 5 * pad256(VC);                                            // d_xacc, d_xd, d_dots, d_gx, d_gy
    for (int j = 0; j < p->n_mlp; ++j)                               // TN partials + bias partials, one region per layer: the sums
        n += pad256((size_t)mb->n_chunks * p->widths[j] * p->widths[j + 1]) + pad256((size_t)mb->n_chunks * p->widths[j + 1]);   // are deferred
    if (p->with_grad) n += pad256((size_t)mb->n_chunks * 4 * p->C * p->C);
    n += pad256((size_t)mb->n_chunks * mb->k_eig * p->C);           // split-V partials of the diffusion backward
    n += pad256((size_t)mb->n_mesh * mb->k_eig * p->C) + pad256((size_t)dn_spec_bwd_dt_rows(mb->n_mesh, mb->k_eig) * p->C);
    n += pad256((size_t)4 * p->C * p->C) + amax_ws();
    if (block_chain_ok(mb, p, 2)) n += pad256((size_t)dn_chain_bwd_pieces(p->C, p->with_grad, p->with_rot, p->n_mlp) * (2 * (p->C / 16) * 64) * 4);
    if (diffuse_ok(mb, p->C)) n += pad256(diffuse_ws_floats(mb)) + pad256((size_t)diffuse_dt_rows(mb) * p->C);
    return n + 512;
}
int dn_block_bwd_f32(const dn_mesh_batch_t* mb, const dn_block_params_t* p, const float* x, const dn_block_saved_t* sv,
                     const float* d_out, const dn_block_grads_t* gr, void* ws, size_t ws_bytes, void* stream) {
    if (!block_params_ok(p) || !sv || !gr) return DN_ERR_INVALID;
    hipStream_t st = S(stream);
    const int C = p->C, K = mb->k_eig;
    const size_t VC = (size_t)mb->v_total * C;
    Bump b(ws, ws_bytes);
    float* da[2] = {b.f((size_t)mb->v_total * max_width(p)), b.f((size_t)mb->v_total * max_width(p))};
    float* d_xacc = b.f(VC);
    float* d_xd = b.f(VC);
    float* d_dots = b.f(VC);
    float* d_gx = b.f(VC);
    float* d_gy = b.f(VC);
    float* part_w[DN_MAX_MLP_LAYERS]; float* part_b[DN_MAX_MLP_LAYERS];
    for (int j = 0; j < p->n_mlp; ++j) {
        part_w[j] = b.f((size_t)mb->n_chunks * p->widths[j] * p->widths[j + 1]);
        part_b[j] = b.f((size_t)mb->n_chunks * p->widths[j + 1]);
    }
    float* part_a = p->with_grad ? b.f((size_t)mb->n_chunks * 4 * C * C) : nullptr;
    float* partial = b.f((size_t)mb->n_chunks * K * C);
    float* dxs = b.f((size_t)mb->n_mesh * K * C);
    float* dtp = b.f((size_t)dn_spec_bwd_dt_rows(mb->n_mesh, K) * C);
    float* psum = b.f((size_t)4 * C * C);
    float* aw = b.f(AW_COUNT + DN_BLOCK_AMAX_WORDS + 2);
    float* chain_ws = block_chain_ok(mb, p, 2) ? b.f((size_t)dn_chain_bwd_pieces(C, p->with_grad, p->with_rot, p->n_mlp) * (2 * (C / 16) * 64) * 4) : nullptr;
    float* diffuse_ws = diffuse_ok(mb, C) ? b.f(diffuse_ws_floats(mb)) : nullptr;
    float* diffuse_dtp = diffuse_ok(mb, C) ? b.f((size_t)diffuse_dt_rows(mb) * C) : nullptr;
    if (!b.ok) return DN_ERR_INVALID;

    // ---- operand magnitudes for the split-fp16 engine (the saved activations' words come from the forward)
    const bool f16 = block_f16_ok(mb, p) && sv->amax;
    // The parameter gradients (dW = d_a^T h, dA) are sums over ALL vertices with heavy cancellation: their error is the operand
    // precision times a condition number of ~1e3, and the two-term fp16 split carries 22 bits against fp32's 24 -- measured on the
    // trained-checkpoint golden: 2.9e-4 from fp64 where the fp32 reference itself is 0.8e-4 away.  They stay on the split-bf16
    // engine (24 bits); the row products (activations, input gradients: 128..384-term sums) take the split-fp16 one.
    const bool wgrad_f16 = opt(O_F16_WGRAD) != 0;
    const float* sw = sv->amax;
    const float *dout_amax = nullptr, *ev_amax = nullptr;
    // the chained backward kernel (dn_chain_bwd.hip) takes the row-local part -- MiniMLP input gradients, tanh', gradient-feature backward --
    // under the conditions of the chained forward (two hidden-gradient buffers: MiniMLPs of up to three layers)
    const bool chainb = block_chain_ok(mb, p, 2) && sv->amax && !wgrad_f16 && p->n_mlp <= 3 && chain_aligned(p, sv, x, d_out, gr->d_x) &&
                        // (the chained backward does not record max |d_xd|: the diagnostic class that needs it keeps the unfused launches)
                        !(block_f16_ok(mb, p) && (f16_mask() & F16_TOB_B) && !p->with_grad);
    const bool words = f16 || chainb;
    ChainPrepArgs pa; memset(&pa, 0, sizeof(pa));
    int chain_np = 0;
    if (words) {
        AmaxInit in; memset(&in, 0, sizeof(in));
        if (!chainb) {
            if (p->with_grad) { in.jobs.push(p->A_re, (long long)C * C, aw + AW_WA); if (p->with_rot) in.jobs.push(p->A_im, (long long)C * C, aw + AW_WA); }
            for (int j = 0; j < p->n_mlp && in.jobs.count < DN_AMAX_MAX_JOBS; ++j) in.jobs.push(p->W[j], (long long)p->widths[j] * p->widths[j + 1], aw + AW_W0 + j);
        }
        in.zero_range(aw, AW_WA);
        in.zero_range(aw + AW_D0, DN_MAX_MLP_LAYERS + 1 + DN_BLOCK_AMAX_WORDS + 2);
        if (f16) in.zero_range(gr->d_x_amax, 1);
        if (chainb) {      // the weight-preparation kernel of the chained backward does this call's bookkeeping (one launch)
            const int NK = C / 32;
            auto piece = [&](const float* Wm, const float* Wm2, float* word, int ld, int row0, int T) {
                ChainPrepPiece& q = pa.pc[chain_np++]; q.W = Wm; q.W2 = Wm2; q.amax = word; q.ld = ld; q.col0 = 32 * T; q.transposed = 1; q.row0 = row0; };
            for (int j = p->n_mlp - 1; j >= 1; --j)
                for (int T = 0; T < NK; ++T) piece(p->W[j], nullptr, aw + AW_W0 + j, C, 0, T);
            for (int sg = 0; sg < (p->with_grad ? 3 : 2); ++sg)
                for (int T = 0; T < NK; ++T) piece(p->W[0], nullptr, aw + AW_W0, p->widths[0], sg * C, T);
            if (p->with_grad)
                for (int rep = 0; rep < chain_hh(mb, true); ++rep)
                    for (int T = 0; T < NK; ++T) {
                        piece(p->A_re, p->with_rot ? p->A_im : nullptr, aw + AW_WA, C, 0, T);
                        if (p->with_rot) piece(p->A_im, p->A_re, aw + AW_WA, C, 0, T);
                    }
            pa.out = reinterpret_cast<uint4*>(chain_ws);
            for (int r = 0; r < in.nzero; ++r) pa.zero_range(in.zero[r], in.zero_n[r]);
            DN_CHECK(dn_launch_chain_prep(pa, chain_np, C, st));
        } else {
            DN_CHECK(dn_launch_amax_init(in, st));
        }
        dout_amax = gr->d_out_amax;
        ev_amax = mb->evecs_amax;
        if (!dout_amax || (f16 && !ev_amax)) {
            AmaxJobs jobs; jobs.count = 0;
            if (!dout_amax) { jobs.push(d_out, (long long)VC, aw + AW_IN); dout_amax = aw + AW_IN; }
            if (f16 && !ev_amax) { jobs.push(mb->evecs, (long long)mb->v_total * K, aw + AW_COUNT + DN_BLOCK_AMAX_WORDS); ev_amax = aw + AW_COUNT + DN_BLOCK_AMAX_WORDS; }
            DN_CHECK(dn_launch_amax(jobs, st));
        }
    }
    auto W = [&](int j) { return (const float*)(aw + AW_W0 + j); };
    auto D = [&](int j) { return aw + AW_D0 + j; };   // magnitude of d(pre-activation of layer j-1) = the d_a consumed by layer j-1; D(n_mlp) unused

    // The fixed-order sums of the weight / bias / rotation-matrix partials are deferred: every product writes its partials to its
    // own region and ONE launch reduces them all once the last one is written (5 launches -> 1 per block).
    MrJobs jobs; jobs.count = 0;
    if (chainb) {
        // ---- row-local gradients in one launch, then the weight-gradient products over what it wrote
        ChainBwdArgs cb; memset(&cb, 0, sizeof(cb));
        cb.d_out = d_out; cb.V = mb->v_total; cb.with_grad = p->with_grad; cb.with_rot = p->with_rot; cb.n_mlp = p->n_mlp;
        for (int j = 1; j < p->n_mlp; ++j) {
            cb.h[j - 1] = sv->h[j - 1];
            cb.dscale[j - 1] = (p->mask[j] || p->drop_seed) ? 2.f : 1.f;
            cb.d_a[j - 1] = da[j & 1];
        }
        cb.g = sv->g; cb.gx = sv->gx; cb.gy = sv->gy; cb.bre = sv->bre; cb.bim = sv->bim;
        cb.wp = reinterpret_cast<const uint4*>(chain_ws);
        cb.wa_amax = aw + AW_WA;
        for (int j = 0; j < p->n_mlp; ++j) cb.w_amax[j] = aw + AW_W0 + j;
        cb.d_out_amax = dout_amax;
        cb.d_xacc = d_xacc; cb.d_xd = d_xd; cb.d_dots = d_dots; cb.d_gx = d_gx; cb.d_gy = d_gy;
        DN_CHECK(dn_launch_chain_bwd(chain_np, cb, C, st, chain_hh(mb, true)));
        // every d_a exists now: the (up to three) weight-gradient products of the MiniMLP go out as ONE launch
        TnBatch tb;
        const float* d_a = d_out;
        for (int j = p->n_mlp - 1; j >= 0; --j) {
            const int wo = p->widths[j + 1], wi = p->widths[j];
            if (j > 0) {
                const float* ins[1] = {sv->h[j - 1]};
                const int iw[1] = {wi};
                DN_CHECK(linear_bwd_weights(mb, d_a, wo, ins, iw, 1, gr->dW[j], gr->db[j], part_w[j], part_b[j], st, &jobs, F16(), &tb));
                d_a = da[j & 1];
            } else {
                const float* ins[3] = {x, sv->xd, sv->g};
                const int iw[3] = {C, C, C};
                DN_CHECK(linear_bwd_weights(mb, d_a, wo, ins, iw, p->with_grad ? 3 : 2, gr->dW[0], gr->db[0], part_w[0], part_b[0], st, &jobs, F16(), &tb));
            }
        }
        DN_CHECK(dn_launch_tngemm_multi(tb.g, tb.nchunks, tb.count, st));
        if (p->with_grad) {
            DN_CHECK(gradfeat_bwd_weights(mb, d_dots, sv->gx, sv->gy, C, gr->dA_re, p->with_rot ? gr->dA_im : nullptr, part_a, psum, st, &jobs));
            DN_CHECK(grad_apply_bwd(mb, d_gx, d_gy, d_xd, C, d_xd, st, (f16 && (f16_mask() & F16_TOB_B)) ? aw + AW_MISC : nullptr));   // d_xd += gradX^T d_gx + gradY^T d_gy (in place)
        }
    } else {
        // ---- MiniMLP backward (autograd of layers.py:236); d_a = gradient w.r.t. a layer's pre-activation output
        const float* d_a = d_out;   // last layer has no activation; the residual branch is added into d_xacc below
        const float* da_amax = dout_amax;
        for (int j = p->n_mlp - 1; j >= 0; --j) {
            const int wo = p->widths[j + 1], wi = p->widths[j];
            if (j > 0) {
                const float* ins[1] = {sv->h[j - 1]};
                const int iw[1] = {wi};
                DN_CHECK(linear_bwd_weights(mb, d_a, wo, ins, iw, 1, gr->dW[j], gr->db[j], part_w[j], part_b[j], st, &jobs,
                                            (f16 && wgrad_f16) ? f16_of(da_amax, sw + SW_H0 + j - 1) : F16()));
                float* nxt = da[j & 1];
                // d(pre-act of layer j-1) = (d_a W_j) * relu'(.) * dropout scale; h>0 <=> kept and active
                DN_CHECK(linear_bwd_input(mb, d_a, wo, p->W[j], wi, 0, wi, DN_EPI_MUL_DFAC, sv->h[j - 1],
                                          (p->mask[j] || p->drop_seed) ? 2.f : 1.f, nxt, st, f16 ? f16_if(F16_LBI, f16_of(da_amax, W(j), D(j))) : F16()));
                d_a = nxt; da_amax = D(j);
            } else {
                const float* ins[3] = {x, sv->xd, sv->g};
                const int iw[3] = {C, C, C};
                F16 fw;
                if (f16 && wgrad_f16) { fw = f16_of(da_amax, sw + SW_X); fw.b.p[1] = sw + SW_XD; fw.b.c = p->with_grad ? 1.f : 0.f; }
                DN_CHECK(linear_bwd_weights(mb, d_a, wo, ins, iw, p->with_grad ? 3 : 2, gr->dW[0], gr->db[0], part_w[0], part_b[0], st, &jobs, fw));
                // d_h0 = d_a W_0 split into its column groups [x | xd | g]
                const F16 fi = f16 ? f16_if(F16_LBI, f16_of(da_amax, W(0))) : F16();
                DN_CHECK(linear_bwd_input(mb, d_a, wo, p->W[0], wi, 0, C, DN_EPI_ADD, d_out, 1.f, d_xacc, st, fi));       // + residual
                F16 fxd = fi; if (f16 && !p->with_grad) fxd.o = aw + AW_MISC;     // without gradient features this IS the d_xd the diffusion backward reads
                DN_CHECK(linear_bwd_input(mb, d_a, wo, p->W[0], wi, C, C, DN_EPI_STORE, nullptr, 1.f, d_xd, st, fxd));
                if (p->with_grad) {
                    F16 fd = fi; if (f16) fd.o = D(0);                             // D(0): magnitude of d_dots
                    DN_CHECK(linear_bwd_input(mb, d_a, wo, p->W[0], wi, 2 * C, C, DN_EPI_DTANH, sv->g, 1.f, d_dots, st, fd));
                }
            }
        }
        // ---- gradient features + gradient apply backward
        if (p->with_grad) {
            const float* A_im = p->with_rot ? p->A_im : nullptr;
            DN_CHECK(gradfeat_bwd_weights(mb, d_dots, sv->gx, sv->gy, C, gr->dA_re, p->with_rot ? gr->dA_im : nullptr, part_a, psum, st, &jobs,
                                          (f16 && wgrad_f16) ? D(0) : nullptr, (f16 && wgrad_f16) ? sw + SW_G : nullptr));
            F16 fg;
            if (f16) { fg = f16_of(D(0), aw + AW_WA); fg.a.mul = sw + SW_G; }     // A = d_dots * (gx | gy)
            DN_CHECK(gradfeat_bwd_inputs(mb, d_dots, sv->gx, sv->gy, sv->bre, sv->bim, p->A_re, A_im, C, d_gx, d_gy, st, f16_if(F16_GFB, fg)));
            DN_CHECK(grad_apply_bwd(mb, d_gx, d_gy, d_xd, C, d_xd, st, (f16 && (f16_mask() & F16_TOB_B)) ? aw + AW_MISC : nullptr));   // d_xd += gradX^T d_gx + gradY^T d_gy (in place)
        }
    }
    // ---- diffusion backward: one persistent launch (3-term engine throughout) when the batch carries its plan; its d_t rows join the block's
    // deferred gradient sums
    if (diffuse_ws && !(f16 && (f16_mask() & F16_TOB_B)) && diffuse_aligned(d_xd, gr->d_x, sv->xs, mb->evecs, p->time, d_xacc)) {
        DfLaunch L = diffuse_new(mb, diffuse_ws);
        L.bwd = 1; L.x = d_xd; L.time = p->time; L.xs = sv->xs; L.out = gr->d_x; L.add = d_xacc; L.dt_part = diffuse_dtp;
        L.out_amax = f16 ? gr->d_x_amax : nullptr;
        DN_CHECK(dn_launch_diffuse(L, st));
        if (!jobs.push(diffuse_dtp, diffuse_dt_rows(mb), C, gr->d_time))
            DN_CHECK(dn_launch_reduce(diffuse_dtp, gr->d_time, diffuse_dt_rows(mb), C, C, st));
        return dn_launch_multi_reduce(jobs, st);          // every parameter gradient of the block: one fixed-order reduction launch
    }
    DN_CHECK(to_basis_partials(mb, d_xd, C, false, partial, st, f16 ? f16_if(F16_TOB_B, f16_of(ev_amax, aw + AW_MISC)) : F16()));
    if (dn_spec_bwd_fused_ok(partial, p->time, sv->xs, dxs, dtp, C)) {
        // one launch: per-mesh sums of the partials, exp(-lambda t), d_t contributions; their sum over (mesh, eigenvalue group) joins the block's
        // other deferred gradient sums below
        DN_CHECK(dn_launch_spec_bwd_fused(partial, mb->mesh_chunk_off, mb->evals, p->time, sv->xs, dxs, dtp, mb->n_mesh, K, C, st, f16 ? aw + AW_YS : nullptr));
        if (!jobs.push(dtp, dn_spec_bwd_dt_rows(mb->n_mesh, K), C, gr->d_time))
            DN_CHECK(dn_launch_reduce(dtp, gr->d_time, dn_spec_bwd_dt_rows(mb->n_mesh, K), C, C, st));
    } else {
        DN_CHECK(dn_launch_seg_reduce(partial, mb->mesh_chunk_off, mb->n_mesh, 0, dxs, (long long)K * C, st));
        DN_CHECK(dn_launch_spec_bwd(dxs, mb->evals, p->time, sv->xs, dtp, mb->n_mesh, K, C, st, f16 ? aw + AW_YS : nullptr));
        DN_CHECK(dn_launch_reduce(dtp, gr->d_time, mb->n_mesh, C, C, st));
    }
    DN_CHECK(dn_launch_multi_reduce(jobs, st));       // every parameter gradient of the block: one fixed-order reduction launch
    return from_basis(mb, dxs, C, gr->d_x, d_xacc, true, st, f16 ? f16_if(F16_FROMB_B, f16_of(ev_amax, aw + AW_YS, gr->d_x_amax)) : F16());
}

// ------------------------------------------------------------------ head / loss next to the path (dn_head.hip)
#define DN_HEAD_BLOCKS 2048
size_t dn_head_workspace_bytes(void) { return pad256(2 * DN_HEAD_BLOCKS) + 512; }
int dn_head_fwd_f32(const float* x, int n_src, int C, const int32_t* rowptr, const int32_t* col, int n_out, float div, int log_softmax,
                    const int64_t* labels, float smoothing, float* logp, float* loss, float* count, void* ws, size_t ws_bytes, void* stream) {
    if (!x || n_src < 0 || n_out < 0 || C <= 0 || (rowptr && !col) || div <= 0.f || (labels && (!loss || !count))) return DN_ERR_INVALID;
    if (!rowptr && n_out != n_src) return DN_ERR_INVALID;
    Bump b(ws, ws_bytes);
    float* partial = b.f(2 * DN_HEAD_BLOCKS);
    if (!b.ok) return DN_ERR_INVALID;
    HeadArgs a; memset(&a, 0, sizeof(a));
    a.x = x; a.ldx = C; a.rowptr = rowptr; a.col = col; a.inv_div = 1.f / div; a.labels = (const long long*)labels; a.smoothing = smoothing;
    a.n_out = n_out; a.n_src = n_src; a.C = C; a.lsm = log_softmax != 0; a.logp = logp; a.partial = partial;
    return dn_launch_head_fwd(a, DN_HEAD_BLOCKS, loss, count, S(stream));
}
int dn_head_bwd_f32(const float* logp, int n_out, int C, const int32_t* t_rowptr, const int32_t* t_col, int n_src, float div, int log_softmax,
                    const int64_t* labels, float smoothing, const float* d_logp, const float* d_loss, const float* count, float* d_x,
                    void* stream) {
    if (n_src < 0 || n_out < 0 || C <= 0 || !d_x || (t_rowptr && !t_col) || div <= 0.f || (log_softmax && !logp) ||
        (labels && d_loss && !count) || (!t_rowptr && n_out != n_src))
        return DN_ERR_INVALID;
    HeadArgs a; memset(&a, 0, sizeof(a));
    a.inv_div = 1.f / div; a.labels = (const long long*)labels; a.smoothing = smoothing; a.n_out = n_out; a.n_src = n_src; a.C = C;
    a.lsm = log_softmax != 0; a.logp = const_cast<float*>(logp); a.t_rowptr = t_rowptr; a.t_col = t_col; a.d_logp = d_logp; a.g_loss = d_loss;
    a.count = count; a.d_x = d_x;
    return dn_launch_head_bwd(a, S(stream));
}

// ------------------------------------------------------------------ input features next to the path
int dn_hks_f32(const float* evals, const float* evecs, const float* scales, int B, int V, int K, int n_scales, int scales_per_batch,
               float* out, void* stream) {
    if (!evals || !evecs || !scales || !out || B < 0 || V < 0 || K <= 0 || n_scales <= 0) return DN_ERR_INVALID;
    return dn_launch_hks(evals, evecs, scales, B, V, K, n_scales, scales_per_batch ? (long long)n_scales : 0LL, out, S(stream));
}

// ------------------------------------------------------------------ operator packing
size_t dn_coo_to_csr_workspace_bytes(int64_t nnz, int n_cols) { return dn_pack_ws_bytes(nnz, n_cols) + 512; }
int dn_coo_to_csr_i64(const int64_t* rows, int row_div, const int64_t* cols, const float* vx, const float* vy, int64_t nnz, int n_rows, int n_cols,
                      int32_t* rowptr, int32_t* col, int32_t* t_rowptr, int32_t* t_col, float* t_vx, float* t_vy, int32_t* status,
                      void* ws, size_t ws_bytes, void* stream) {
    if (nnz < 0 || n_rows < 0 || n_cols < 0 || nnz >= 2147483647LL || (!rows && row_div <= 0) || !rowptr || !t_rowptr || !status ||
        (nnz > 0 && (!cols || !col || !t_col || (vx && !t_vx) || (vy && !t_vy))))
        return DN_ERR_INVALID;
    Bump b(ws, ws_bytes);
    int* w = (int*)b.f((size_t)n_cols + 1 + (size_t)nnz);
    if (!b.ok) return DN_ERR_INVALID;
    return dn_launch_coo_to_csr((const long long*)rows, row_div, (const long long*)cols, vx, vy, nnz, n_rows, n_cols, rowptr, col, t_rowptr, t_col,
                                t_vx, t_vy, status, w, S(stream));
}

// 128-bit content checksum of a device buffer, ACCUMULATED into acc[0..1] (the caller zeroes them; several operands with different
// salts share one accumulator).  nbytes must be a multiple of 4 and data 4-byte aligned (every operand of the path is fp32 / int32 / int64).
int dn_checksum128(const void* data, size_t nbytes, uint64_t salt, uint64_t* acc, void* stream) {
    if (!acc || (nbytes && !data) || nbytes % 4 != 0 || ((uintptr_t)data & 3) != 0 || ((uintptr_t)acc & 7) != 0) return DN_ERR_INVALID;
    return dn_launch_checksum(data, (long long)(nbytes / 4), (unsigned long long)salt, (unsigned long long*)acc, S(stream));
}

// the same for up to DN_CK_MAX_JOBS buffers in one launch (host arrays of device pointers, byte counts and salts)
int dn_checksum128_multi(int n, const void* const* data, const size_t* nbytes, const uint64_t* salts, uint64_t* acc, void* stream) {
    if (n < 0 || n > DN_CK_MAX_JOBS || !acc || ((uintptr_t)acc & 7) != 0 || (n && (!data || !nbytes || !salts))) return DN_ERR_INVALID;
    long long nwords[DN_CK_MAX_JOBS];
    unsigned long long s64[DN_CK_MAX_JOBS];
    for (int j = 0; j < n; ++j) {
        if ((nbytes[j] && !data[j]) || nbytes[j] % 4 != 0 || ((uintptr_t)data[j] & 3) != 0) return DN_ERR_INVALID;
        nwords[j] = (long long)(nbytes[j] / 4);
        s64[j] = (unsigned long long)salts[j];
    }
    return dn_launch_checksum_multi(n, data, nwords, s64, (unsigned long long*)acc, S(stream));
}

// ------------------------------------------------------------------ output remaps
int dn_csr_mean_f32(const int32_t* rowptr, const int32_t* col, int n_rows, const float* x, int C, float div, float* out,
                    void* stream) {
    SpArgs s; memset(&s, 0, sizeof(s));
    s.rowptr = rowptr; s.col = col; s.x1 = x; s.o1 = out; s.nrows = n_rows; s.C = C; s.ldx = C; s.ldo = C;
    s.mode = DN_SP_ONE; s.div = div;
    return dn_launch_spmm(s, S(stream));
}
int dn_mass_mean_fwd_f32(const dn_mesh_batch_t* mb, const float* x, int C, float* out, float* mass_sum, void* stream) {
    return dn_launch_mass_mean_fwd(T(mb->mesh_rows), mb->mass, x, out, mass_sum, mb->n_mesh, C, S(stream));
}
int dn_mass_mean_bwd_f32(const dn_mesh_batch_t* mb, const float* mass_sum, const float* d_out, int C, float* d_x, void* stream) {
    return dn_launch_mass_mean_bwd(T(mb->tiles), mb->n_tiles, mb->mass, mass_sum, d_out, d_x, C, S(stream));
}

}  // extern "C"
