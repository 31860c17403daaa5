/* diffnet_hip.h -- C ABI of libdiffnet_hip.so: the MI355X (gfx950) DiffusionNet hot path.
 *
 * The reference (nmwsharp/diffusion-net) has no native layer: its hot path is Python calling
 * stock ATen kernels.  This header is the boundary a maintainer binds instead (ctypes stub in
 * INTEGRATION.md).  Every entry point cites the reference interface it replaces
 * (file:line under src/diffusion_net/).
 *
 * Conventions (all entry points):
 *   - return value: hipError_t as int, 0 = success; never throws;
 *   - never allocates, frees or synchronises; all work is enqueued on `stream` (a hipStream_t);
 *   - every pointer inside dn_mesh_batch_t / dn_block_* and every float* / int* argument is a
 *     DEVICE pointer owned by the caller and kept alive until the stream has drained; the structs
 *     themselves live in host memory and are read during the call only;
 *   - dense arrays are row-major contiguous fp32; index arrays int32; masks uint8 (1 = keep);
 *   - `ws` is caller-provided device scratch of at least the size the matching
 *     dn_*_workspace_bytes() reports; contents are undefined on return;
 *   - re-entrant: the compute path reads no environment variables and writes no global state.  Three process-wide pieces of
 *     bookkeeping exist: the opt-in timing of dn_prof_* (off by default), a per-device "large-LDS attribute set" bitmap per kernel
 *     instantiation (set-once, idempotent), and the table of tuning options behind dn_set_option() (kernel selection for A/B runs and
 *     tests; relaxed atomics written only by that call, read by the entry points at call time -- a value must not change between a
 *     *_workspace_bytes() query and the call it sizes, nor between the warm-up and the capture of a HIP graph; per-call engine choices
 *     go through dn_block_params_t.flags instead);
 *   - every struct below must be ZERO-INITIALISED by the caller before its fields are filled (memset): fields are only ever appended, and
 *     an appended field's zero value selects the behaviour of the versions before it.  dn_version() identifies the layout generation.
 *
 * A "mesh batch" is a ragged batch of independent meshes whose vertex axes are concatenated:
 * mesh m owns rows [mesh_rows[m].row0, +nrows) of every [v_total, *] array; the sparse gradient
 * operators use global (concatenated) row/column indices, i.e. they are block-diagonal.
 * Table invariants: meshes, tiles and chunks are listed in row order and each table covers [0, v_total) without gaps or
 * overlap; a tile / chunk never straddles two meshes.
 */
#ifndef DIFFNET_HIP_H
#define DIFFNET_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DN_MAX_MLP_LAYERS 8

/* a run of rows that belongs to one mesh (tiles: <= dn_tile_rows() rows; chunks: any length) */
typedef struct { int32_t row0, nrows, mesh, aux; } dn_tile_t;

/* Per-batch geometry: what geometry.get_operators() returns per mesh (geometry.py:426-570),
 * packed for the device.  gradX/gradY share one CSR pattern (geometry.py:381-382). */
typedef struct dn_mesh_batch {
    int32_t n_mesh, v_total, k_eig, n_tiles, n_chunks, g_nnz;  /* g_nnz: non-zeros of the gradient pattern */
    const dn_tile_t* tiles;          /* [n_tiles]  row tiles of <= dn_tile_rows() rows              */
    const dn_tile_t* chunks;         /* [n_chunks] split-V chunks, grouped by mesh                   */
    const int32_t*   mesh_chunk_off; /* [n_mesh+1] chunk range of every mesh                         */
    const dn_tile_t* mesh_rows;      /* [n_mesh]   {row0, nrows, m, 0}                               */
    const float* mass;               /* [v_total]            lumped vertex areas                     */
    const float* evals;              /* [n_mesh, k_eig]      Laplacian eigenvalues                   */
    const float* evecs;              /* [v_total, k_eig]     mass-orthonormal eigenbasis             */
    const int32_t* g_rowptr;  const int32_t* g_col;  const float* g_vx;  const float* g_vy;   /* CSR of gradX/gradY  */
    const int32_t* gt_rowptr; const int32_t* gt_col; const float* gt_vx; const float* gt_vy;  /* CSR of transposes   */
    /* Optional (round 3): one device float each holding max |evecs| and max |mass| (any upper bound within ~2^10 is as good) -- the
     * operand magnitudes the split-fp16 matrix engine of the fused block scales by.  NULL: the block calls measure them themselves
     * (one extra pass over the eigenbasis per call). */
    const float* evecs_amax; const float* mass_amax;
    /* Optional: one device float holding the infinity norm of the stacked gradient operators, max_i sum_j max(|gradX_ij|, |gradY_ij|)
     * summed as sum_j |gradX_ij| resp. sum_j |gradY_ij| (the larger of the two row sums): max |gradX x|, |gradY x| <= it * max |x| is
     * the magnitude bound the split-fp16 gradient-feature products scale by (about 3x the true maximum on a mesh).  NULL: measured. */
    const float* grad_norm;
    /* Optional (round 5): the work plan of the one-launch diffusion operator (dn_diffusion_plan()): device array of df_n_groups * df_n_wg
     * entries.  NULL / 0: the diffusion runs as three launches (projection, spectral step, back-projection). */
    const dn_tile_t* df_plan; int32_t df_n_wg, df_n_groups;
    /* Optional (round 6): the spectral-gradient operands of the batch (dn_spectral_pack_f32): the spatial gradient apply re-associated
     * through the eigenbasis, gradX (evecs ys) = (gradX evecs) ys.  sg_pack: [evecs | gradX evecs | gradY evecs] as fp16 (hi, lo) operand
     * fragments, 16 rows per group, every mesh padded to whole units of 64 (k_eig <= 128) / 128 (k_eig = 256) rows; sg_units: [sg_n_units] the units ({row0, nrows, mesh, 0},
     * dn_spectral_units()); sg_amax: [n_mesh][4] floats, the per-mesh magnitudes the fragments were scaled by.  With them the block forward
     * computes xd, gx, gy inside its chained row kernel (no back-projection launch, no CSR gather) for the shapes
     * dn_spectral_grad_supported() names.  NULL / 0: back-projection + gather, as before. */
    const void* sg_pack; const dn_tile_t* sg_units; const float* sg_amax; int32_t sg_n_units;
    /* Stamp of df_plan (round 6): v_total of the batch dn_diffusion_plan() was called for.  The plan names rows the kernels write: a plan whose
     * stamp is not this batch's v_total, or whose df_n_wg is not dn_diffusion_plan_wgs() of the current device, is IGNORED (row GEMM instead) --
     * a caller that leaves the field zero therefore never takes the planned kernels. */
    int32_t df_v_total;
} dn_mesh_batch_t;

/* Weights of one DiffusionNetBlock (layers.py:167-198), nn.Linear layout: W[out][in]. */
typedef struct dn_block_params {
    int32_t C, n_mlp, with_grad, with_rot;
    int32_t widths[DN_MAX_MLP_LAYERS + 1]; /* MiniMLP sizes: widths[0] = 3C (2C w/o gradient features), ..., widths[n_mlp] = C */
    const float* time;                     /* [C] diffusion_time, clamped to >= 1e-8 (layers.py:48-49) -- by the caller, or see clamp_time */
    const float* A_re; const float* A_im;  /* [C,C] (with_rot=0: A_re holds A, A_im ignored), layers.py:110-113 */
    const float* W[DN_MAX_MLP_LAYERS];     /* [widths[i+1], widths[i]] */
    const float* b[DN_MAX_MLP_LAYERS];     /* [widths[i+1]] */
    const uint8_t* mask[DN_MAX_MLP_LAYERS];/* mask[i], i>=1: dropout keep-mask of layer i's INPUT [v_total, widths[i]]
                                              (layers.py:143-147), NULL = no mask given; kept values are scaled by 2 */
    uint64_t drop_seed;                    /* != 0 and mask[i] == NULL: nn.Dropout(p=.5) of layer i's input is drawn IN the
                                              producing kernel's epilogue, keep = bit of hash(drop_seed, i, row, column): no mask
                                              tensor is written or read (the backward needs none: a saved activation is > 0
                                              exactly where it was kept and active).  0 = no dropout where mask[i] == NULL. */
    const uint64_t* drop_seed_dev;         /* optional DEVICE word added to drop_seed inside the kernels: a captured HIP graph of the training
                                              step advances it with a graph node, so that every replay draws new masks without host work */
    /* Optional (round 3), magnitudes for the split-fp16 engine: x_amax = device float holding max |x| of the block input (the previous
     * block's out_amax); NULL: measured by the call (one extra pass over x).  out_amax: device float that receives max |out| (zeroed by
     * the call when dn_block_tracks_amax() says the call tracks magnitudes, untouched otherwise), or NULL. */
    const float* x_amax; float* out_amax;
    /* Optional (round 4): != 0 makes dn_block_fwd_f32 raise `time` to >= 1e-8 IN PLACE before anything reads it -- the clamp the reference
     * applies to the parameter in every forward (layers.py:48-49: the Parameter keeps the clamped values) -- as part of the call's first
     * launch, instead of a separate elementwise kernel of the caller.  `time` must then be writable.  The backward ignores the field. */
    int32_t clamp_time;
    /* Optional (round 6): per-call engine choice, overriding the process-wide dn_set_option() table for THIS call (two models in one process may
     * want different engines; the table is for A/B runs).  0 = follow the table.  Workspace queries take the same params, so a call and its
     * query always agree. */
    uint32_t flags;
} dn_block_params_t;
#define DN_BLOCK_NO_CHAIN 1u                 /* unfused launches instead of the chained row kernels                  */
#define DN_BLOCK_NO_F16 2u                   /* split-bf16 engine throughout (implies the unfused launches)          */
#define DN_BLOCK_NO_SPECTRAL_GRAD 4u         /* back-projection + CSR gather even if the batch carries sg_pack       */
#define DN_BLOCK_SPECTRAL_GRAD_ALWAYS 8u     /* spectral-gradient forward at every size (option "spectral_grad" = 2) */

/* Activations the forward saves for the backward (caller-allocated). */
typedef struct dn_block_saved {
    float* xs;                         /* [n_mesh, k_eig, C]  spectrum of x                     */
    float* xd;                         /* [v_total, C]        diffused features                 */
    float* gx; float* gy;              /* [v_total, C]        spatial gradients   (with_grad)   */
    float* g; float* bre; float* bim;  /* [v_total, C]        tanh features, rotated gradients  */
    float* h[DN_MAX_MLP_LAYERS];       /* h[i], i < n_mlp-1: [v_total, widths[i+1]] post-ReLU(+dropout) */
    float* amax;                       /* optional [DN_BLOCK_AMAX_WORDS] device floats: magnitudes of the saved activations (written by the
                                          forward, read by the backward).  NULL: both calls use the split-bf16 engine throughout. */
} dn_block_saved_t;
#define DN_BLOCK_AMAX_WORDS 16

typedef struct dn_block_grads {
    float* d_x;                        /* [v_total, C] */
    float* d_time;                     /* [C] */
    float* dA_re; float* dA_im;        /* [C,C] (with_rot=0: dA_re holds dA) */
    float* dW[DN_MAX_MLP_LAYERS];
    float* db[DN_MAX_MLP_LAYERS];
    const float* d_out_amax;           /* optional: device float with max |d_out| (the next block's d_x_amax); NULL: measured by the call */
    float* d_x_amax;                   /* optional: device float that receives max |d_x| */
} dn_block_grads_t;

int dn_version(void);         /* 600: round-6 layout (dn_mesh_batch_t.sg_pack ...); 500: round 5 (dn_mesh_batch_t.df_plan ..., dn_set_option) */
int dn_tile_rows(void);      /* rows per entry of dn_mesh_batch_t.tiles (128) */
int dn_tn_target_chunks(void);  /* how many entries dn_mesh_batch_t.chunks should have on the current device for the split-V products to fill it in
                                   whole rounds: one workgroup slot per CU (wave-specialised kernel).  Any chunk list is CORRECT; this one is fastest. */
int dn_tn_target_chunks_k(int k_eig);   /* (round 6) the same for a batch of k_eig eigenvectors per mesh: from k_eig = 256 on half as many chunks -- every chunk
                                   writes a k_eig x C partial that the spectral step reads back (one 200k-vertex mesh, K = C = 256: projection 210 -> 190 us,
                                   block inference 1388 -> 1347 us with 240 instead of 489 chunks; profiles/r06_chunks256.txt) */

/* ---- tuning options (no reference counterpart): process-wide integers the entry points read at call time, for A/B measurements and for
 *      tests that must run a particular kernel.  Unknown names return non-zero.  Defaults in parentheses.
 *        "chain" (1)            chained row kernels of the fused block (0: unfused launches)
 *        "chain_min_rows" (0)   smallest batch the TRAINING forward takes the chained kernel for ...
 *        "chain_small_rows" (0)      ... and the largest small batch below that which takes it all the same (a window for the unfused launches: tests)
 *        "chain_hh" (0)         16-row halves per wave of the chained kernels (0: by batch size and direction, dn_api.hip; the forward
 *                               kernel of C = 256 has one shape -- two halves, four waves -- and ignores this option and the next)
 *        "chain_nw" (0)         waves per workgroup of the chained kernels (0: chosen by batch size)
 *        "f16" (1)              split-fp16 matrix engine for the row products of the fused block (0: split-bf16 everywhere)
 *        "f16_mask" (188)       product classes on the split-fp16 engine (diagnostic bit mask, dn_api.hip)
 *        "f16_wgrad" (0)        parameter-gradient products on the split-fp16 engine (diagnostic)
 *        "diffuse" (2)          batches that carry a plan (dn_diffusion_plan), K = C = 128: 2 = the back-projection of the diffusion is the direct
 *                               row-product launch (dn_diffuse.hip: backproject_kernel); 1 = the whole operator as ONE persistent launch
 *                               (diffuse_kernel; any number of groups); 0 = the wave-specialised row GEMM of the earlier rounds
 *        "diffuse_groups" (1)   mesh groups dn_diffusion_plan() deals the batch into when asked for 0 (the direct back-projection needs 1)
 *        "diffuse_order" (0)    schedule of the one-launch operator (0: interleaved groups, 1: projections first)
 *        "diffuse_flags" (1)    one-launch kernel: bit 0 = arrivals posted from inside the next projection loop; bits 1, 2 = tests (forced solo path)
 *        "diffuse_split" (0)    one-launch kernel: bit i = a kernel boundary after schedule step i (measurements)
 *        "spectral_grad" (1)    batches that carry spectral-gradient operands (dn_spectral_pack_f32): the chained forward kernel computes xd, gx, gy
 *                               from them (no back-projection launch, no CSR gather) -- 1 = in the inference forward at every size and in the training
 *                               forward up to 65536 rows, both at C <= 128 (where it measures faster), 2 = wherever a form exists (incl. the two-launch
 *                               form of k_eig = C = 256, which measures slower), 0 = never (back-projection + gather) */
int dn_set_option(const char* name, int value);
int dn_get_option(const char* name, int* value);

/* ---- opt-in per-kernel timing for benchmarks (no reference counterpart; the one piece of mutable global
 *      state, off by default): every launch is bracketed by hipEvents on its stream and summed per kernel
 *      kernel kind in [0, 12) (dn_prof_kind_name(kind) is "" past the last one).  read: out[0..3] = {ms, launches, algorithmic flops, algorithmic bytes}. */
int dn_prof_enable(int on);
int dn_prof_reset(void);
int dn_prof_read(int kind, double* out);
const char* dn_prof_kind_name(int kind);

/* ---- geometry.to_basis (geometry.py:572-583): spec[m] = evecs_m^T (x_m * mass_m); use_mass=0 drops the mass
 *      (the autograd transpose of from_basis).  spec: [n_mesh, k_eig, C]. */
size_t dn_to_basis_workspace_bytes(const dn_mesh_batch_t* mb, int C);
int dn_to_basis_f32(const dn_mesh_batch_t* mb, const float* x, int C, int use_mass, float* spec,
                    void* ws, size_t ws_bytes, void* stream);
/* ---- geometry.from_basis (geometry.py:586-598): out_m = evecs_m spec[m].  out: [v_total, C].
 *      scale_rows_by_mass=1 multiplies row v by mass[v] (the autograd transpose of to_basis). */
int dn_from_basis_f32(const dn_mesh_batch_t* mb, const float* spec, int C, int scale_rows_by_mass, float* out, void* stream);

/* ---- LearnedTimeDiffusion.forward, method='spectral' (layers.py:44-67) and its gradient.
 *      fwd: xs = to_basis(x), xd = from_basis(exp(-evals*time) * xs).
 *      bwd: d_x = d_x_add + mass * (evecs (coef * evecs^T d_xd)), d_time[c] = -sum lambda*coef*xs*(evecs^T d_xd). */
/*      dn_diffuse.hip (direct back-projection launch; one-launch form): taken when mb->df_plan is set, k_eig = C = 128 and every operand is
 *      16-byte aligned (option "diffuse").
 *      dn_diffusion_plan_wgs(): workgroups the plan should be made for on the current device (one per CU).
 *      dn_diffusion_plan(): HOST arithmetic -- sizes[n_mesh] = vertices per mesh in row order; n_groups = 0: the "diffuse_groups" option;
 *      plan: HOST array of DN_DIFFUSION_MAX_GROUPS * n_wg entries; returns the number of groups used (0: batch not plannable, leave
 *      df_plan NULL).  The caller uploads plan[0 .. groups * n_wg) and sets df_plan / df_n_wg / df_n_groups. */
#define DN_DIFFUSION_MAX_GROUPS 4
int dn_diffusion_plan_wgs(void);
int dn_diffusion_plan(const int32_t* sizes, int n_mesh, int n_wg, int n_groups, dn_tile_t* plan);
size_t dn_diffusion_workspace_bytes(const dn_mesh_batch_t* mb, int C);
int dn_diffusion_fwd_f32(const dn_mesh_batch_t* mb, const float* x, const float* time, int C,
                         float* xs, float* xd, void* ws, size_t ws_bytes, void* stream);
int dn_diffusion_bwd_f32(const dn_mesh_batch_t* mb, const float* d_xd, const float* xs, const float* time, int C,
                         const float* d_x_add, float* d_x, float* d_time, void* ws, size_t ws_bytes, void* stream);

/* ---- spectral-gradient operands (dn_spectral.hip; no reference counterpart: layers.py:213-223 applies gradX / gradY to x_diffuse = evecs ys,
 *      which lies in the span of evecs -- gradX x_diffuse = (gradX evecs) ys exactly).  Built ONCE per mesh batch:
 *      dn_spectral_grad_supported(k_eig, C): 1 if the block forward takes the operands at this shape (k_eig = 128 with C in {64, 128};
 *      k_eig = C = 256).
 *      dn_spectral_units(): HOST arithmetic -- sizes[n_mesh] = vertices per mesh in row order -> units[] (NULL: count only) of <= 64 rows
 *      (k_eig <= 128) or <= 128 rows (k_eig = 256) of one mesh each; returns the count.
 *      dn_spectral_pack_bytes(n_units, k_eig): bytes of the packed operand.  dn_spectral_pack_workspace_bytes(): scratch of the pack call.
 *      dn_spectral_pack_f32(): mb needs evecs, the gradient CSR and k_eig % 32 == 0; units = DEVICE copy of the unit table; writes
 *      sg_pack[dn_spectral_pack_bytes] and sg_amax[4 n_mesh]; gradX evecs / gradY evecs are accumulated in fp64 in entry order (deterministic).
 *      The caller then sets mb->sg_pack / sg_units / sg_amax / sg_n_units. */
int dn_spectral_grad_supported(int k_eig, int C);
int dn_spectral_units(const int32_t* sizes, int n_mesh, int k_eig, dn_tile_t* units);
size_t dn_spectral_pack_bytes(int n_units, int k_eig);
size_t dn_spectral_pack_workspace_bytes(const dn_mesh_batch_t* mb);
int dn_spectral_pack_f32(const dn_mesh_batch_t* mb, const dn_tile_t* units, int n_units, void* sg_pack, float* sg_amax,
                         void* ws, size_t ws_bytes, void* stream);

/* ---- spatial gradient apply, layers.py:217-223 (gx = gradX x, gy = gradY x) and its transpose
 *      d_x = add + gradX^T d_gx + gradY^T d_gy (add may be NULL). */
int dn_grad_apply_fwd_f32(const dn_mesh_batch_t* mb, const float* x, int C, float* gx, float* gy, void* stream);
int dn_grad_apply_bwd_f32(const dn_mesh_batch_t* mb, const float* d_gx, const float* d_gy, const float* add, int C,
                          float* d_x, void* stream);

/* ---- SpatialGradientFeatures.forward (layers.py:117-130): g = tanh(gx*Bre + gy*Bim).
 *      A_im = NULL selects with_gradient_rotations=False (layers.py:125-126).  bre/bim may be NULL (inference). */
size_t dn_gradfeat_workspace_bytes(const dn_mesh_batch_t* mb, int C);
int dn_gradfeat_fwd_f32(const dn_mesh_batch_t* mb, const float* gx, const float* gy, const float* A_re, const float* A_im,
                        int C, float* g, float* bre, float* bim, void* stream);
int dn_gradfeat_bwd_f32(const dn_mesh_batch_t* mb, const float* d_g, const float* g, const float* gx, const float* gy,
                        const float* bre, const float* bim, const float* A_re, const float* A_im, int C,
                        float* d_gx, float* d_gy, float* dA_re, float* dA_im, void* ws, size_t ws_bytes, void* stream);

/* ---- nn.Linear (first_lin / last_lin, layers.py:297-298,366,373; MiniMLP layers, layers.py:150-164):
 *      out = act(x W^T + b) [* 2*mask];  relu/mask as in MiniMLP.  bwd: d_x = d_out W (skipped when d_x NULL),
 *      dW = d_out^T x, db = column sums of d_out. */
size_t dn_linear_workspace_bytes(const dn_mesh_batch_t* mb, int C_in, int C_out);
int dn_linear_fwd_f32(const dn_mesh_batch_t* mb, const float* x, int C_in, const float* W, const float* b, int C_out,
                      int relu, const uint8_t* mask, float* out, void* stream);
int dn_linear_bwd_f32(const dn_mesh_batch_t* mb, const float* d_out, const float* x, const float* W, int C_in, int C_out,
                      float* d_x, float* dW, float* db, void* ws, size_t ws_bytes, void* stream);
/* The same two products, additionally leaving max |out| (max |d_x|) in a device word for the split-fp16 engine of the block that
 * consumes the result (dn_block_params_t.x_amax / dn_block_grads_t.d_out_amax): first_lin feeds block 0, last_lin's input gradient feeds
 * the last block's backward.  The word is ACCUMULATED into (atomic max): the caller zeroes it.  NULL = not tracked (the plain forms).
 * Thin products (C_in <= 16 forward, C_out <= 16 backward) track inside their kernel; others take one measuring pass. */
int dn_linear_fwd_amax_f32(const dn_mesh_batch_t* mb, const float* x, int C_in, const float* W, const float* b, int C_out,
                           int relu, const uint8_t* mask, float* out, float* out_amax, void* stream);
int dn_linear_bwd_amax_f32(const dn_mesh_batch_t* mb, const float* d_out, const float* x, const float* W, int C_in, int C_out,
                           float* d_x, float* dW, float* db, float* d_x_amax, void* ws, size_t ws_bytes, void* stream);

/* ---- DiffusionNetBlock.forward (layers.py:200-241) and its backward, fused orchestration.
 *      saved = NULL runs inference (intermediates live in ws). */
size_t dn_block_fwd_workspace_bytes(const dn_mesh_batch_t* mb, const dn_block_params_t* p, int with_saved);
/*      1 if the block call `call` (0: dn_block_fwd_f32 without saved activations, 1: dn_block_fwd_f32 with them, 2: dn_block_bwd_f32) on this batch
 *      and these parameters writes the optional magnitude words (out_amax and saved->amax / d_x_amax), 0 if it leaves them untouched (shapes
 *      neither the split-fp16 engine nor the chained kernels take): a caller that hands the words on to the next block must not hand on words
 *      nobody wrote. */
int dn_block_tracks_amax(const dn_mesh_batch_t* mb, const dn_block_params_t* p, int call);
size_t dn_block_bwd_workspace_bytes(const dn_mesh_batch_t* mb, const dn_block_params_t* p);
int dn_block_fwd_f32(const dn_mesh_batch_t* mb, const dn_block_params_t* p, const float* x, float* out,
                     const dn_block_saved_t* saved, void* ws, size_t ws_bytes, void* stream);
int dn_block_bwd_f32(const dn_mesh_batch_t* mb, const dn_block_params_t* p, const float* x,
                     const dn_block_saved_t* saved, const float* d_out, const dn_block_grads_t* grads,
                     void* ws, size_t ws_bytes, void* stream);

/* ---- operator packing: sparse COO (int64 indices, as the reference holds gradX/gradY: utils.py:55, geometry.py:375-382) -> int32 CSR of
 *      the pattern and CSR of its transpose, values carried along; also builds the gather patterns of the faces/edges remap.
 *      rows: [nnz] non-decreasing row ids (coalesced COO order), or NULL: entry j belongs to row j / row_div (a dense [n_rows, row_div]
 *      index array such as faces).  cols: [nnz].  vx / vy: [nnz] values on the shared pattern (either may be NULL).
 *      Out: rowptr [n_rows+1], col [nnz]; transpose: t_rowptr [n_cols+1], t_col [nnz] (row ids, ascending inside a transposed row),
 *      t_vx / t_vy [nnz].  status: one device int32, non-zero afterwards if an index was outside [0,n_rows) x [0,n_cols) or the
 *      rows were not sorted (the outputs are then undefined; the caller checks it at its next synchronisation point). */
size_t dn_coo_to_csr_workspace_bytes(int64_t nnz, int n_cols);
int dn_coo_to_csr_i64(const int64_t* rows, int row_div, const int64_t* cols, const float* vx, const float* vy, int64_t nnz, int n_rows,
                      int n_cols, int32_t* rowptr, int32_t* col, int32_t* t_rowptr, int32_t* t_col, float* t_vx, float* t_vy,
                      int32_t* status, void* ws, size_t ws_bytes, void* stream);

/* ---- content checksum for the device-resident operator cache (the reference re-uploads the operators of a mesh every step,
 *      human_segmentation_original.py:111-120, and caches them on disk by a hash of the mesh, geometry.py:476-494: the device-side analogue).
 *      Adds a 128-bit, order-independent checksum of the nbytes at `data` (multiple of 4, 4-byte aligned) to acc[0..1] (device, zeroed by
 *      the caller).  `salt` separates operands that share one accumulator.  EVERY byte of the operand enters the sum. */
int dn_checksum128(const void* data, size_t nbytes, uint64_t salt, uint64_t* acc, void* stream);
/*      The same for up to 16 buffers in ONE launch: data / nbytes / salts are HOST arrays of n entries (device pointers, byte counts, salts).
 *      Gives exactly the sum the n single calls would. */
int dn_checksum128_multi(int n, const void* const* data, const size_t* nbytes, const uint64_t* salts, uint64_t* acc, void* stream);

/* ---- output remaps (layers.py:379-397).
 *      csr_mean: out[i] = (sum_{j in row i} x[col[j]]) / div  -- faces (div=3) / edges (div=2) gather-mean, and with the
 *      transposed pattern its gradient.  mass_mean: out[m] = sum_v mass*x / sum_v mass per mesh (global_mean). */
int dn_csr_mean_f32(const int32_t* rowptr, const int32_t* col, int n_rows, const float* x, int C, float div,
                    float* out, void* stream);
int dn_mass_mean_fwd_f32(const dn_mesh_batch_t* mb, const float* x, int C, float* out, float* mass_sum, void* stream);
int dn_mass_mean_bwd_f32(const dn_mesh_batch_t* mb, const float* mass_sum, const float* d_out, int C, float* d_x,
                         void* stream);

/* ---- the head on the far side of the path, one pass each way (dn_head.hip): [gather-mean of the 2/3 vertices of an edge/face,
 *      layers.py:379-391] -> [log_softmax, the scripts' last_activation, human_segmentation_original.py:75] -> log-probabilities ->
 *      [NLL, F.nll_loss at human_segmentation_original.py:136 / rna_mesh_segmentation.py:135, or with smoothing > 0 the label-smoothed
 *      log loss of utils.py:18-24; mean over the rows whose label lies in [0, C) -- others (ignore_index) neither contribute nor count].
 *      Every bracketed stage is optional: rowptr = NULL: output i = row i (n_out == n_src); log_softmax = 0: x already holds
 *      log-probabilities; labels = NULL: no loss; logp = NULL: log-probabilities not written.  div: entries per output (3 faces, 2 edges).
 *      loss / count: device scalars (count = number of valid rows, needed by the backward).
 *      bwd: d_x[v] = sum_{i gathered v} (1/div) (dlp_i - [log_softmax] exp(logp_i) sum_c dlp_i[c]),
 *           dlp_i = d_logp_i (optional) - d_loss/count * smoothed-one-hot(label_i) (optional); t_rowptr/t_col: transposed gather. */
size_t dn_head_workspace_bytes(void);
int dn_head_fwd_f32(const float* x, int n_src, int C, const int32_t* rowptr, const int32_t* col, int n_out, float div, int log_softmax,
                    const int64_t* labels, float smoothing, float* logp, float* loss, float* count, void* ws, size_t ws_bytes, void* stream);
int dn_head_bwd_f32(const float* logp, int n_out, int C, const int32_t* t_rowptr, const int32_t* t_col, int n_src, float div, int log_softmax,
                    const int64_t* labels, float smoothing, const float* d_logp, const float* d_loss, const float* count, float* d_x,
                    void* stream);

/* ---- geometry.compute_hks (geometry.py:600-628; compute_hks_autoscale :630-633 passes scales = logspace(-2, 0, S)):
 *      out[b][v][s] = sum_k exp(-evals[b][k] * scales[.][s]) * evecs[b][v][k]^2.  evals [B,K], evecs [B,V,K], out [B,V,S];
 *      scales [S] shared by the batch (scales_per_batch = 0) or [B,S] (scales_per_batch = 1).  Forward only (an input feature). */
int dn_hks_f32(const float* evals, const float* evecs, const float* scales, int B, int V, int K, int S, int scales_per_batch,
               float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DIFFNET_HIP_H */
