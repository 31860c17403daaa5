// dn_diffusion_fused.hip -- LearnedTimeDiffusion.forward (layers.py:44-67 + geometry.py:572-598) as ONE persistent launch for the
// K = C = 128 configuration:  x_diffuse = Phi (exp(-lambda t) * (Phi^T (M x))).
//
// The three-launch form (split-V product -> partial reduction + scaling -> row product) streams Phi twice from HBM, writes and re-reads
// one 64 KiB partial per chunk (514 of them on the benchmark batch), and pays three launch gaps and three prologues for ~100 us of work.
// Here one workgroup per CU owns a contiguous row range of ONE mesh for the whole operator:
//   phase 1  partial[wg] = Phi[rows]^T (m x)[rows]          (the split-V engine of dn_tngemm.hip, double-buffered)
//   sync A   all workgroups of the mesh have published their partial              (agent-scope release / acquire, one counter per mesh)
//   phase 2  workgroup i of the mesh's n reduces rows [128 i / n, 128 (i+1) / n) of the spectrum over the n partials IN ORDER (fixed
//            assignment + fixed order: bitwise reproducible), writes xs (kept for the backward) and ys = exp(-lambda t) xs
//   sync B   the scaled spectrum of the mesh is complete
//   phase 3  x_diffuse[rows] = Phi[rows] ys                 (direct row product, dn_direct_tiles.h: Phi fragments straight from memory --
//            the rows this workgroup streamed 20-30 us earlier, so mostly still in the Infinity Cache -- spectrum planes resident in LDS)
// Inter-workgroup visibility follows MI355X_MICROARCH.md (workgroup dispatch section): plain stores -> __syncthreads -> lane 0 agent-scope
// release fence + explicit vmcnt(0) -> relaxed agent atomic arrive; consumer: relaxed polls with s_sleep -> ONE agent-scope acquire
// fence -> __syncthreads -> plain loads.  Every spin is bounded: on a timeout the workgroup sets *status and carries on (wrong
// numbers, never a hang).  The launch needs every workgroup resident: grid = the plan's size <= number of CUs, one workgroup (120 KiB
// of LDS) per CU; the plan (rows of each workgroup, its mesh, its position among the mesh's workgroups) is built once per mesh batch on
// the host.  The emulator tier cannot run it (its workgroups execute one after the other) and keeps the three-launch form.
#include "dn_tn_tiles.h"
#include "dn_direct_tiles.h"

#if defined(DN_DF_TRACE)   // development build only: s_memtime stamps of workgroups 0, 100, 255 (tid 0), read with dn_debug_df_trace_read
__device__ unsigned long long dn_df_trace_buf[16 * 16];
extern "C" int dn_debug_df_trace_read(unsigned long long* out, int n) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(dn_df_trace_buf), sizeof(unsigned long long) * n); }
#define DF_T(i_)                                                                                                  \
    do {                                                                                                          \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime();                                               \
        if (threadIdx.x == 0 && blockIdx.x < 16) dn_df_trace_buf[blockIdx.x * 16 + (i_)] = t_;                         \
    } while (0)
#else
#define DF_T(i_) do {} while (0)
#endif

// Elements between two workgroups' partials: 128 x 128 plus 1280 bytes.  With a power-of-two stride every workgroup writes (phase 1) and
// every reducer reads (phase 2) the same offset of sixteen slabs at the same time = the same memory channel: measured 17-27k cycles for
// the 64 KiB store of the unlucky workgroups against 3k for the others, and everybody waits for the slowest at the hand-off.
#define DN_DF_PSTRIDE (128 * 128 + 320)

struct DfArgs {
    const DnTile* plan;          // [n_wg] {row0, nrows, mesh, aux = first_wg_of_mesh * 1024 + n_wg_of_mesh}
    const float* evecs;          // [V, 128]
    const float* x;              // [V, 128]
    const float* mass;           // [V] (phase 1 row scale) or null
    const float* evals;          // [n_mesh, 128]
    const float* time;           // [128] or null (no scaling)
    float* xs;                   // [n_mesh, 128, 128] unscaled spectrum or null
    float* ys;                   // [n_mesh, 128, 128] scaled spectrum (scratch)
    float* xd;                   // [V, 128] out
    float* partial;              // [n_wg, 128, 128] scratch
    int* counters;               // [2 n_mesh] zeroed before the launch
    int* status;                 // device int, or-ed with 1 on a spin timeout
};

// 16-byte WRITE-THROUGH store (sc1): the data leaves the XCD's L2 with the store, so publishing it needs no release fence -- a fence
// writes back every dirty line of the L2 (measured here: 20 us per hand-off with 64 KiB freshly written per workgroup; the
// microarchitecture guide's "publish-large" row: 8.2 us vs 3.0 us).  The compiler does not count inline-asm stores: wait explicitly.
typedef float df_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void df_store_wt(float* p, float4 v) {
#ifdef DN_EMULATE
    *reinterpret_cast<float4*>(p) = v;
#else
    const df_f4 w = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(w) : "memory");
#endif
}
__device__ __forceinline__ void df_drain_stores() {
#ifndef DN_EMULATE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
}

__device__ __forceinline__ void df_arrive_and_wait(int* ctr, int n, int* status) {
    // caller: every wave has drained its write-through stores, then __syncthreads(); __syncthreads() again after (lane 0 acquired)
#ifndef DN_EMULATE
    __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int spins = 0;
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < n) {
        __builtin_amdgcn_s_sleep(4);
        if (++spins > (1 << 21)) { atomicOr(status, 1); break; }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
}

__global__ __launch_bounds__(DN_TX_THREADS) DN_WAVES_PER_EU(2) void diffusion_fused_kernel(DfArgs a) {
    constexpr int K = 128, C = 128;
    constexpr int SBUF = 6 * DN_TX_PLANE;   // bytes of one (A,B) step buffer (3 planes each); two buffers in phase 1
    DN_DYN_SMEM(smem_raw);
    unsigned char* smem = reinterpret_cast<unsigned char*>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = DN_UNIFORM(tid >> 6);
    const DnTile me = a.plan[blockIdx.x];
    const int grp_first = me.aux >> 10, grp_n = me.aux & 1023, grp_i = (int)blockIdx.x - grp_first;
    DF_T(0);

    // ------------------------------------------------------------------ phase 1: partial = Phi[rows]^T (m x)[rows]
    {
        const int wr = wave >> 2, wc = wave & 3;           // 2 x 4 waves, 64 x 32 outputs each
        const int li = lane & 31;
        f32x16 acc[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        const int q = tid & 31, kr0 = tid >> 5;            // this thread stages column group q of rows kr0, kr0 + 16
        const float* ap = a.evecs + 4 * q;
        const float* bp = a.x + 4 * q;
        TnArgs tg;
        tg.b_rowscale = a.mass;
        float4 csum = dn_f4_zero();
        TxRegs R;
        DnTile ch;
        ch.row0 = me.row0; ch.nrows = me.nrows; ch.mesh = me.mesh; ch.aux = 0;
        const int nsteps = (ch.nrows + DN_KB - 1) / DN_KB;
        if (a.mass) {
            tx_load<DN_TN_ROWSCALE>(tg, ch, 0, kr0, true, true, ap, ap, K, bp, C, R);
            tx_store<DN_TN_ROWSCALE>(smem, smem + 3 * DN_TX_PLANE, kr0, q, R, csum);
            if (nsteps > 1) tx_load<DN_TN_ROWSCALE>(tg, ch, 1, kr0, true, true, ap, ap, K, bp, C, R);
        } else {
            tx_load<DN_TN_PLAIN>(tg, ch, 0, kr0, true, true, ap, ap, K, bp, C, R);
            tx_store<DN_TN_PLAIN>(smem, smem + 3 * DN_TX_PLANE, kr0, q, R, csum);
            if (nsteps > 1) tx_load<DN_TN_PLAIN>(tg, ch, 1, kr0, true, true, ap, ap, K, bp, C, R);
        }
        __syncthreads();
        // pipeline: regs(step+1) -> LDS[other]; loads(step+2) -> regs; MFMAs on LDS[cur]; one barrier per step
        for (int st = 0; st < nsteps; ++st) {
            unsigned char* cur = smem + (st & 1) * SBUF;
            unsigned char* nxt = smem + ((st & 1) ^ 1) * SBUF;
            if (st + 1 < nsteps) {
                if (a.mass) tx_store<DN_TN_ROWSCALE>(nxt, nxt + 3 * DN_TX_PLANE, kr0, q, R, csum);
                else tx_store<DN_TN_PLAIN>(nxt, nxt + 3 * DN_TX_PLANE, kr0, q, R, csum);
            }
            if (st + 2 < nsteps) {
                if (a.mass) tx_load<DN_TN_ROWSCALE>(tg, ch, st + 2, kr0, true, true, ap, ap, K, bp, C, R);
                else tx_load<DN_TN_PLAIN>(tg, ch, st + 2, kr0, true, true, ap, ap, K, bp, C, R);
            }
            tx_compute(cur, cur + 3 * DN_TX_PLANE, wr, wc, lane, acc);
            __syncthreads();
        }
        DF_T(7);
        // the partial goes out through LDS (the step buffers are free: the loop ended on a barrier) as 16-byte write-through stores
        float* sP = reinterpret_cast<float*>(smem);
        const int n = wc * 32 + li;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) sP[((wr * 2 + i) * 32 + dn_acc_row(r, lane)) * C + n] = acc[i][r];
        __syncthreads();
        float* out = a.partial + (long long)blockIdx.x * DN_DF_PSTRIDE;
#pragma unroll
        for (int i = 0; i < K * C / 4 / DN_TX_THREADS; ++i) {
            const int e = tid + i * DN_TX_THREADS;
            df_store_wt(out + 4 * e, *reinterpret_cast<const float4*>(sP + 4 * e));
        }
        df_drain_stores();
    }
    // ------------------------------------------------------------------ sync A: the mesh's partials are published
    DF_T(1);
    __syncthreads();
    if (tid == 0) df_arrive_and_wait(a.counters + 2 * me.mesh, grp_n, a.status);
    __syncthreads();
    DF_T(2);
    // ------------------------------------------------------------------ phase 2: my rows of the spectrum, partials summed in order
    {
        const int k_beg = (int)((long long)grp_i * K / grp_n), k_end = (int)((long long)(grp_i + 1) * K / grp_n);
        const float* pbase = a.partial + (long long)grp_first * DN_DF_PSTRIDE;
        for (int e = tid; e < (k_end - k_beg) * (C / 4); e += DN_TX_THREADS) {
            const int k = k_beg + e / (C / 4), c4 = e % (C / 4);
            const long long off = (long long)k * C + 4 * c4;
            float4 s = dn_f4_zero();
            for (int j0 = 0; j0 < grp_n; j0 += 8) {              // eight partials in flight, summed in order
                float4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(pbase + (long long)(j0 + u < grp_n ? j0 + u : j0) * DN_DF_PSTRIDE + off);
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (j0 + u < grp_n) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
            }
            const long long o = (long long)me.mesh * K * C + off;
            if (a.xs) *reinterpret_cast<float4*>(a.xs + o) = s;
            if (a.time) {
                const float lam = a.evals[me.mesh * K + k];
                const float4 t = *reinterpret_cast<const float4*>(a.time + 4 * c4);
                s = make_float4(expf(-lam * t.x) * s.x, expf(-lam * t.y) * s.y, expf(-lam * t.z) * s.z, expf(-lam * t.w) * s.w);
            }
            df_store_wt(a.ys + o, s);
        }
        df_drain_stores();
    }
    // ------------------------------------------------------------------ sync B: the mesh's scaled spectrum is complete
    DF_T(3);
    __syncthreads();
    if (tid == 0) df_arrive_and_wait(a.counters + 2 * me.mesh + 1, grp_n, a.status);
    __syncthreads();
    // ------------------------------------------------------------------ phase 3: x_diffuse[rows] = Phi[rows] ys
    {
        DF_T(4);
        rd_stage_b_nn<DN_TX_THREADS>(a.ys + (long long)me.mesh * K * C, C, smem, tid);
        __syncthreads();
        DF_T(5);
        RgArgs g;
        g.o0 = a.xd; g.ldo = C; g.ldr = C; g.N = C; g.r0 = nullptr; g.rowv = nullptr; g.bias = nullptr; g.mask = nullptr; g.rng_seed = 0ull;
        g.scale = 1.f;
        rd_run_rows<DN_EPI_STORE>(g, smem, a.evecs, K, me.row0, me.row0 + me.nrows, 0, wave, lane);
        DF_T(6);
    }
}

size_t dn_diffusion_fused_ws_bytes(int n_wg, int n_mesh) {
    return (((size_t)n_wg * DN_DF_PSTRIDE * sizeof(float) + 255) & ~(size_t)255) + (((size_t)(2 * n_mesh + 1) * sizeof(int) + 255) & ~(size_t)255);
}

// ws: partial [n_wg, DN_DF_PSTRIDE] floats, then 2 n_mesh counters + 1 status int.  Returns hipError_t as int.
int dn_launch_diffusion_fused(const DnTile* plan, int n_wg, int n_mesh, const float* evecs, const float* x, const float* mass, const float* evals,
                              const float* time, float* xs, float* ys, float* xd, void* ws, hipStream_t stream) {
#ifdef DN_EMULATE
    (void)plan; (void)n_wg; (void)n_mesh; (void)evecs; (void)x; (void)mass; (void)evals; (void)time; (void)xs; (void)ys; (void)xd; (void)ws; (void)stream;
    return DN_ERR_BAD_MODE;
#else
    DfArgs a;
    a.plan = plan; a.evecs = evecs; a.x = x; a.mass = mass; a.evals = evals; a.time = time; a.xs = xs; a.ys = ys; a.xd = xd;
    a.partial = reinterpret_cast<float*>(ws);
    a.counters = reinterpret_cast<int*>(reinterpret_cast<char*>(ws) + (((size_t)n_wg * DN_DF_PSTRIDE * sizeof(float) + 255) & ~(size_t)255));
    a.status = a.counters + 2 * n_mesh;
    const size_t smem = (size_t)2 * 6 * DN_TX_PLANE;   // 120 KiB (phase 1: two step buffers; phase 3 uses the first 96 KiB)
    static unsigned long long lds_opt_in = 0;
    { const int oe_ = dn_lds_opt_in(reinterpret_cast<const void*>(&diffusion_fused_kernel), smem, &lds_opt_in); if (oe_) return oe_; }
    hipError_t e = hipMemsetAsync(a.counters, 0, (size_t)(2 * n_mesh + 1) * sizeof(int), stream);
    if (e != hipSuccess) return (int)e;
    dn_prof_begin(DN_K_SMALL, stream);
    DN_LAUNCH(diffusion_fused_kernel, dim3(n_wg, 1, 1), dim3(DN_TX_THREADS, 1, 1), smem, stream, a);
    dn_prof_end(DN_K_SMALL, stream, 0.0, 0.0);
    return (int)hipGetLastError();
#endif
}
