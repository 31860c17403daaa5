// dn_diffuse.hip -- LearnedTimeDiffusion, method='spectral' (layers.py:44-67 + geometry.py:572-598) and its gradient as ONE persistent
// launch for K = C = 128:
//     forward   x_diffuse = Phi (exp(-lambda t) * (Phi^T (M x)))                      xs = Phi^T (M x) kept for the backward
//     backward  d_x = add + M * (Phi (exp(-lambda t) * (Phi^T d_xd))),   d_t rows = -lambda * coef * xs * (Phi^T d_xd) summed per workgroup
//
// The three-launch form (split-V product -> per-mesh sum of partials + scaling -> row product) is ~100 us of kernels for 325 MB of
// algorithmic traffic on the benchmark batch, and its own timing does not change when the operands come out of the Infinity Cache: it
// is bound by fixed costs (three ramps and tails, two dependent boundaries, the row product's LDS round trip for Phi), not by bytes.
// Round 2 built the obvious fusion -- every workgroup owns a row range of one mesh: project, hand off, reduce a slice of the spectrum,
// hand off, back-project -- and measured 109-112 us: with ONE mesh group every hand-off is exposed (45k + 20k cycles of 200k: the
// write-through drain of the 64 KiB partials and the arrival skew, with the HBM idle meanwhile).  This kernel keeps that decomposition and
// removes the exposure:
//   * the meshes of the batch are dealt into G groups and every workgroup owns a row range in EACH group; its schedule interleaves the
//     groups (G = 3:  P1a P1b Ra P1c Rb P3a Rc P3b P3c) so that every dependency -- "all partials of my mesh are published", "the
//     scaled spectrum of my mesh is complete" -- was satisfied one long phase earlier by construction: the polls succeed on their first
//     read and the publishing stores drain under the next phase's loads (the arrival is posted from inside the next projection loop);
//   * P1 (projection)       partial[slot] = Phi[rows]^T (m x)[rows]: the split-V engine of dn_tngemm.hip (k-major bf16 planes, transpose
//                            reads), double-buffered, 16-byte write-through stores of the partial (MI355X_MICROARCH.md "publish-large");
//   * R  (slice reduce)      workgroup i of the mesh's n sums rows [128 i / n, 128 (i + 1) / n) of the n partials IN SLOT ORDER (fixed
//                            assignment + fixed order: bitwise reproducible), applies exp(-lambda t), writes xs (forward) or the d_t
//                            contributions (backward), publishes its slice of the scaled spectrum;
//   * P3 (back-projection)   out[rows] = Phi[rows] ys: direct row product (dn_direct_tiles.h: Phi fragments straight from memory --
//                            the rows this workgroup streamed a phase earlier -- spectrum planes resident in LDS), epilogue store or
//                            add + mass * acc, max |out| for the split-fp16 consumers.
// Inter-workgroup visibility (MI355X_MICROARCH.md, "Workgroup dispatch, XCD placement & inter-workgroup visibility"): payloads are
// written with sc1 (write-through) stores and read with sc1 loads (8-byte agent-scope relaxed atomics, L1-bypassing); a counter per
// (mesh, stage) is raised by one relaxed agent-scope atomic per workgroup after every storing wave has drained (explicit vmcnt(0)) and
// the workgroup has met at a barrier; consumers poll it with relaxed loads + s_sleep from one lane.
// NO CO-RESIDENCY ASSUMPTION: a poll that runs out (another process holds CUs, the grid does not fit, a debugger) does not end in a
// hang or in wrong numbers -- the waiting workgroup computes the missing inputs itself ("solo": the projection of EVERY row range of its
// mesh in slot order, the same sums bit for bit) and carries on; it still publishes its own contributions, so every workgroup always
// makes progress whatever the residency.  The emulator tier, whose workgroups run one after the other, executes the schedule one step
// per launch and covers the solo path with the DN_DF_FLAG_SOLO_* test flags.
#include "dn_tn_tiles.h"
#include "dn_direct_tiles.h"
#include <string.h>

#define DN_DF_PSTRIDE (128 * 128 + 320)   // floats between two slots' partials: a power-of-two stride puts every workgroup's stores (and
                                          // every reducer's reads) of the same offset on one memory channel (round 2: 17-27k vs 3k cycles)
#define DN_DF_SPINS 4096                  // polls (s_sleep 8 + one L2 round trip each, ~1 us) before a workgroup goes solo: ~4 ms

#if defined(DN_DF_TRACE) && !defined(DN_EMULATE)   // development build only: s_memtime stamps of the first 16 workgroups (thread 0)
__device__ unsigned long long dn_df_trace_buf[16 * 32];
extern "C" int dn_debug_df_trace_read(unsigned long long* out, int n) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(dn_df_trace_buf), sizeof(unsigned long long) * n); }
__device__ unsigned long long dn_df_wg_times[512 * 2];   // s_memrealtime (100 MHz, chip-wide) at the start / end of every workgroup of backproject_kernel
extern "C" int dn_debug_df_wg_times_read(unsigned long long* out, int n) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(dn_df_wg_times), sizeof(unsigned long long) * n); }
#define DF_WG(i_) do { if (threadIdx.x == 0 && blockIdx.x < 512) dn_df_wg_times[blockIdx.x * 2 + (i_)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define DF_T(i_)                                                                                                  \
    do {                                                                                                          \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime();                                               \
        if (threadIdx.x == 0 && blockIdx.x < 16 && (i_) < 32) dn_df_trace_buf[blockIdx.x * 32 + (i_)] = t_;       \
    } while (0)
#else
#define DF_T(i_) do {} while (0)
#define DF_WG(i_) do {} while (0)
#endif

struct DfArgs {
    const DnTile* plan;          // [n_groups * n_wg] {row0, nrows, mesh, aux = first_slot_of_mesh * 1024 + n_slots_of_mesh}; mesh < 0: idle
    int n_wg, n_groups;
    int sched[DN_DF_MAX_SCHED];  // (op << 8) | group
    int n_sched, s_begin, s_end;
    int flags;
    const float* evecs;          // [V, 128]
    const float* x;              // [V, 128] projected operand (x forward, d_xd backward)
    const float* mass_in;        // [V] row scale of the projected operand (forward) or null
    const float* evals;          // [n_mesh, 128]
    const float* time;           // [128]
    float* xs_out;               // forward: [n_mesh, 128, 128] unscaled spectrum, or null
    const float* xs_in;          // backward: the forward's spectrum
    float* ys;                   // [n_mesh, 128, 128] scaled spectrum (scratch)
    float* out;                  // [V, 128]
    const float* add;            // backward: optional addend [V, 128]
    const float* rowv;           // backward: mass [V]
    float* partial;              // [n_groups * n_wg, DN_DF_PSTRIDE] scratch
    float* solo;                 // [n_wg, 128 * 128] private scratch of the solo path
    float* dt_part;              // backward: [n_groups * n_wg, 128] d_t contributions (every row written)
    int* counters;               // [2 n_mesh] zeroed before the launch
    float* out_amax;             // optional: receives max |out| (atomic max; zeroed by the caller)
};

typedef float df_f4 __attribute__((ext_vector_type(4)));
// 16-byte WRITE-THROUGH store (sc1): the data leaves the XCD's L2 with the store, publishing it needs no release fence (a fence writes
// back every dirty line of the L2: 20 us per hand-off measured in round 2).  The compiler does not count inline-asm stores: df_drain().
__device__ __forceinline__ void df_store_wt(float* p, float4 v) {
#ifdef DN_EMULATE
    *reinterpret_cast<float4*>(p) = v;
#else
    const df_f4 w = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(w) : "memory");
#endif
}
__device__ __forceinline__ void df_drain() {
#ifndef DN_EMULATE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
}
__device__ __forceinline__ void df_st2_coherent(float* p, float x, float y) {
#ifdef DN_EMULATE
    p[0] = x; p[1] = y;
#else
    const unsigned long long u = (unsigned long long)__float_as_uint(x) | ((unsigned long long)__float_as_uint(y) << 32);
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}
__device__ __forceinline__ void df_arrive(int* ctr) {
#ifdef DN_EMULATE
    *ctr += 1;
#else
    __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}
// true: the counter reached n.  false: the polls ran out (or the test flag says so): the caller computes what it waited for itself.
// Two barriers: the verdict travels through one LDS word.
__device__ __forceinline__ bool df_wait(int* ctr, int n, bool fail, volatile int* sflag, int tid) {
    if (tid == 0) {
        int ok = 0;
        if (!fail) {
#ifdef DN_EMULATE
            ok = *ctr >= n;
#else
            for (int spins = 0; spins < DN_DF_SPINS; ++spins) {
                if (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= n) { ok = 1; break; }
                __builtin_amdgcn_s_sleep(8);
            }
#endif
        }
        *sflag = ok;
    }
    __syncthreads();
    const bool r = *sflag != 0;
    __syncthreads();
    return r;
}
// every storing wave drained, the workgroup met, one arrival
__device__ __forceinline__ void df_flush(int*& pend, int tid) {
    if (pend) {
        df_drain();
        __syncthreads();
        if (tid == 0) df_arrive(pend);
        pend = nullptr;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// P1: the 128 x 128 partial of rows [row0, row0 + nrows) into LDS (sP = smem as [128][128] floats), all waves past a barrier on return.
// pend: an arrival of an EARLIER phase whose write-through stores are still draining: it is posted from inside the loop (second step: the
// wait that the staging of that step needs anyway covers the old stores), or at the end when the loop is too short.
template <bool MASS>
__device__ __forceinline__ void df_p1(const DfArgs& a, unsigned char* smem, int row0, int nrows, int tid, int lane, int wave, int*& pend) {
    constexpr int K = 128, C = 128;
    constexpr int SBUF = 6 * DN_TX_PLANE;   // bytes of one (A,B) step buffer (3 planes each); two buffers
    constexpr int FL = MASS ? DN_TN_ROWSCALE : DN_TN_PLAIN;
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
    tg.b_rowscale = a.mass_in;
    float4 csum = dn_f4_zero();
    TxRegs R;
    DnTile ch;
    ch.row0 = row0; ch.nrows = nrows; ch.mesh = 0; ch.aux = 0;
    const int nsteps = (nrows + DN_KB - 1) / DN_KB;
    tx_load<FL>(tg, ch, 0, kr0, true, true, ap, ap, K, bp, C, R);
    tx_store<FL>(smem, smem + 3 * DN_TX_PLANE, kr0, q, R, csum);
    if (nsteps > 1) tx_load<FL>(tg, ch, 1, kr0, true, true, ap, ap, K, bp, C, R);
    __syncthreads();
    // pipeline: regs(step+1) -> LDS[other]; loads(step+2) -> regs; MFMAs on LDS[cur]; one barrier per step
    for (int st = 0; st < nsteps; ++st) {
        unsigned char* cur = smem + (st & 1) * SBUF;
        unsigned char* nxt = smem + ((st & 1) ^ 1) * SBUF;
        const bool post = pend != nullptr && st == 1;      // uniform
        if (post) df_drain();
        if (st + 1 < nsteps) tx_store<FL>(nxt, nxt + 3 * DN_TX_PLANE, kr0, q, R, csum);
        if (st + 2 < nsteps) tx_load<FL>(tg, ch, st + 2, kr0, true, true, ap, ap, K, bp, C, R);
        tx_compute(cur, cur + 3 * DN_TX_PLANE, wr, wc, lane, acc);
        __syncthreads();
        if (post) { if (tid == 0) df_arrive(pend); pend = nullptr; }
    }
    df_flush(pend, tid);
    // the partial goes through LDS (the step buffers are free: the loop ended on a barrier) so that it leaves as 16-byte stores
    float* sP = reinterpret_cast<float*>(smem);
    const int n = wc * 32 + li;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) sP[((wr * 2 + i) * 32 + dn_acc_row(r, lane)) * C + n] = acc[i][r];
    __syncthreads();
}

// the spectral scaling both directions share (layers.py:62-64)
__device__ __forceinline__ float df_coef(float lam, float t) { return expf(-lam * t); }

// The whole unscaled spectrum of the mesh whose slots are [first, first + n) into S (private to this workgroup): every slot's partial
// in slot order, S = ((0 + p_0) + p_1) + ... -- the sums the slice reducers form, bit for bit.
template <bool MASS>
__device__ void df_solo_sum(const DfArgs& a, unsigned char* smem, int first, int n, float* S, int tid, int lane, int wave) {
    const float* sP = reinterpret_cast<const float*>(smem);
    int* none = nullptr;
    for (int j = 0; j < n; ++j) {
        const DnTile t = a.plan[first + j];
        df_p1<MASS>(a, smem, t.row0, t.nrows, tid, lane, wave, none);
#pragma unroll
        for (int i = 0; i < 128 * 128 / 4 / DN_TX_THREADS; ++i) {
            const int e = tid + i * DN_TX_THREADS;
            const float4 v = *reinterpret_cast<const float4*>(sP + 4 * e);
            float4 s = j ? *reinterpret_cast<const float4*>(S + 4 * e) : dn_f4_zero();
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            *reinterpret_cast<float4*>(S + 4 * e) = s;
        }
        __syncthreads();
    }
}

// R: rows [kb, ke) of the mesh's spectrum.  S != null: the sums are read from the solo buffer instead of the partials.
template <bool BWD>
__device__ void df_reduce(const DfArgs& a, unsigned char* smem, int mesh, int first, int n, int idx, int slot, const float* S, int tid) {
    constexpr int K = 128, C = 128;
    const int kb = (int)((long long)idx * K / n), ke = (int)((long long)(idx + 1) * K / n);
    const int units = (ke - kb) * (C / 2);
    float* sdt = reinterpret_cast<float*>(smem);           // [ke - kb][C] d_t contributions (backward)
    const float* pbase = a.partial + (long long)first * DN_DF_PSTRIDE;
    for (int u = tid; u < units; u += DN_TX_THREADS) {
        const int k = kb + u / (C / 2), c = 2 * (u % (C / 2));
        const long long off = (long long)k * C + c;
        float sx = 0.f, sy = 0.f;
        if (S) { sx = S[off]; sy = S[off + 1]; }
        else {
            for (int j0 = 0; j0 < n; j0 += 16) {            // sixteen partials in flight, summed in slot order
                float2 v[16];
#pragma unroll
                for (int w = 0; w < 16; ++w) v[w] = dn_ld2_coherent(pbase + (long long)(j0 + w < n ? j0 + w : j0) * DN_DF_PSTRIDE + off);
#pragma unroll
                for (int w = 0; w < 16; ++w)
                    if (j0 + w < n) { sx += v[w].x; sy += v[w].y; }
            }
        }
        const long long o = (long long)mesh * K * C + off;
        const float lam = a.evals[mesh * K + k];
        const float cx = df_coef(lam, a.time[c]), cy = df_coef(lam, a.time[c + 1]);
        if (BWD) {
            const float2 xin = *reinterpret_cast<const float2*>(a.xs_in + o);
            sdt[(k - kb) * C + c] = -(lam * sx * cx * xin.x);
            sdt[(k - kb) * C + c + 1] = -(lam * sy * cy * xin.y);
        } else if (a.xs_out) {
            *reinterpret_cast<float2*>(a.xs_out + o) = make_float2(sx, sy);
        }
        df_st2_coherent(a.ys + o, cx * sx, cy * sy);
    }
    if (BWD) {
        __syncthreads();
        if (tid < C) {
            float s = 0.f;
            for (int kk = 0; kk < ke - kb; ++kk) s += sdt[kk * C + tid];
            a.dt_part[(long long)slot * C + tid] = s;
        }
        __syncthreads();
    }
}

template <bool BWD>
__global__ __launch_bounds__(DN_TX_THREADS) DN_WAVES_PER_EU(2) void diffuse_kernel(DfArgs a) {
    constexpr int K = 128, C = 128;
    constexpr int SMEM = 2 * 6 * DN_TX_PLANE;              // 120 KiB: two projection step buffers; the back-projection uses the first 96 KiB
    DN_DYN_SMEM(smem_raw);
    unsigned char* smem = reinterpret_cast<unsigned char*>(smem_raw);
    volatile int* sflag = reinterpret_cast<volatile int*>(smem + SMEM);
    float* swmax = reinterpret_cast<float*>(smem + SMEM + 16);
    const int tid = threadIdx.x, lane = tid & 63, wave = DN_UNIFORM(tid >> 6);
    const int wg = blockIdx.x;
    float* S = a.solo + (long long)wg * K * C;
    int* pend = nullptr;           // arrival whose stores are still draining
    float om = 0.f;
    DF_T(0);
    for (int si = a.s_begin; si < a.s_end; ++si) {
        const int op = a.sched[si] >> 8, g = a.sched[si] & 255;
        const int slot = g * a.n_wg + wg;
        const DnTile me = a.plan[slot];
        const int first = me.aux >> 10, n = me.aux & 1023, idx = slot - first;
        if (me.mesh < 0 || n == 0) {                                      // no work in this group
            if (BWD && op == DN_DF_OP_R && tid < C) a.dt_part[(long long)slot * C + tid] = 0.f;
            continue;
        }
        int* cnt_p = a.counters + 2 * me.mesh;
        int* cnt_y = cnt_p + 1;
        if (op == DN_DF_OP_P1) {
            if (!(a.flags & DN_DF_FLAG_DEFER)) df_flush(pend, tid);
            df_p1<!BWD>(a, smem, me.row0, me.nrows, tid, lane, wave, pend);
            DF_T(2 * si + 1);
            const float* sP = reinterpret_cast<const float*>(smem);
            float* out = a.partial + (long long)slot * DN_DF_PSTRIDE;
#pragma unroll
            for (int i = 0; i < K * C / 4 / DN_TX_THREADS; ++i) {
                const int e = tid + i * DN_TX_THREADS;
                df_store_wt(out + 4 * e, *reinterpret_cast<const float4*>(sP + 4 * e));
            }
            __syncthreads();                                              // sP is read: the next phase may stage into it
            pend = cnt_p;
        } else if (op == DN_DF_OP_R) {
            df_flush(pend, tid);
            const bool have = df_wait(cnt_p, n, (a.flags & DN_DF_FLAG_SOLO_R) != 0 && ((wg + me.mesh) & 1) == 0, sflag, tid);
            DF_T(2 * si + 1);
            if (!have) df_solo_sum<!BWD>(a, smem, first, n, S, tid, lane, wave);
            df_reduce<BWD>(a, smem, me.mesh, first, n, idx, slot, have ? nullptr : S, tid);
            pend = cnt_y;
        } else {
            df_flush(pend, tid);
            const bool have = df_wait(cnt_y, n, (a.flags & DN_DF_FLAG_SOLO_P3) != 0 && ((wg + me.mesh) & 1) == 1, sflag, tid);
            DF_T(2 * si + 1);
            const float* ysrc = a.ys + (long long)me.mesh * K * C;
            if (!have) {                                                  // the whole scaled spectrum, privately
                df_solo_sum<!BWD>(a, smem, first, n, S, tid, lane, wave);
                for (int e = tid; e < K * C; e += DN_TX_THREADS) S[e] = df_coef(a.evals[me.mesh * K + e / C], a.time[e % C]) * S[e];
                __syncthreads();
                ysrc = S;
            }
            if (have) rd_stage_b_nn<DN_TX_THREADS, true, 3>(ysrc, C, smem, tid);
            else rd_stage_b_nn<DN_TX_THREADS, false, 3>(ysrc, C, smem, tid);
            __syncthreads();
            RgArgs rg;
            rg.o0 = a.out; rg.ldo = C; rg.ldr = C; rg.N = C; rg.r0 = a.add; rg.rowv = a.rowv; rg.bias = nullptr; rg.mask = nullptr; rg.rng_seed = 0ull;
            rg.scale = 1.f;
            const float m = rd_run_rows<BWD ? DN_EPI_MASS_ADD : DN_EPI_STORE>(rg, smem, a.evecs, K, me.row0, me.row0 + me.nrows, 0, wave, lane);
            om = m > om ? m : om;
            __syncthreads();                                              // the planes are read: the next phase may stage
        }
        DF_T(2 * si + 2);
    }
    df_flush(pend, tid);
    if (a.out_amax) {      // one check-first atomic per workgroup (a posted atomic per wave serialises on the word's channel)
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) { const float o = __shfl_xor(om, d, 64); om = o > om ? o : om; }
        if (lane == 0) swmax[wave] = om;
        __syncthreads();
        if (tid == 0) {
            float mm = 0.f;
            for (int w = 0; w < DN_TX_THREADS / 64; ++w) mm = swmax[w] > mm ? swmax[w] : mm;
            if (mm > 0.f && mm > *reinterpret_cast<volatile float*>(a.out_amax)) atomicMax(reinterpret_cast<unsigned*>(a.out_amax), __float_as_uint(mm));
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The back-projection as a launch of its own (the shipped form of the diffusion operator: projection and spectral step stay the split-V
// kernel + per-mesh reduce of dn_tngemm.hip / dn_pointwise.hip): out[rows] = Phi[rows] ys[mesh] (+ the backward's add + mass * acc) with
// the direct row product.  One 512-thread workgroup per CU owns the contiguous rows its plan entry names; the spectrum planes of its mesh
// are split once into LDS (96 KiB), Phi fragments come straight from memory -- the rows the projection kernel streamed two launches
// earlier: Infinity-Cache hits.  A kernel of its own rather than the third phase of diffuse_kernel: the register allocator sizes a
// kernel for its worst phase and spills in the hottest one -- the unit loop ran 68k cycles inside the one-launch kernel (27-51 spilled
// registers, every reload behind a vmcnt(0) that drains the operand prefetch) against 38k when it is alone.
// ---------------------------------------------------------------------------------------------------------------------------------
struct BpArgs {
    const DnTile* plan; int n_wg;
    const float* evecs; const float* ys; float* out; const float* add; const float* rowv; float* out_amax;
    DnAmax a_amax, b_amax;       // NP = 2 (split-fp16 engine): magnitude bounds of Phi and of the spectrum (power-of-two operand scales)
};
// NP = 3: 3-term split-bf16 (the forward: its output is what the gradient operators difference); NP = 2: 2-term split-fp16 with
// power-of-two operand scales from the producers' magnitude words (the backward of the fused block, as in rounds 3-4): half the MFMAs,
// 6 instead of 11 split instructions per operand pair, 64 instead of 96 KiB of planes.
template <int MODE, int NP>
__global__ __launch_bounds__(DN_TX_THREADS) DN_WAVES_PER_EU(2) void backproject_kernel(BpArgs a) {
    constexpr int K = 128, C = 128;
    DN_DYN_SMEM(smem_raw);
    unsigned char* smem = reinterpret_cast<unsigned char*>(smem_raw);
    float* swmax = reinterpret_cast<float*>(smem + DN_RD_LDS_B);
    const int tid = threadIdx.x, lane = tid & 63, wave = DN_UNIFORM(tid >> 6);
    const DnTile me = a.plan[blockIdx.x];
    float om = 0.f;
    float sa = 1.f, sb = 1.f, so = 1.f;
    if constexpr (NP == 2) {
        sa = dn_pow2_scale(dn_amax_eval(a.a_amax));
        sb = dn_pow2_scale(dn_amax_eval(a.b_amax));
        so = (1.f / sa) * (1.f / sb);
    }
    DF_T(0);
    DF_WG(0);
    if (me.mesh >= 0 && me.nrows > 0) {
        // the wave's first two 16-row units of Phi are requested BEFORE the planes are staged (they do not depend on them): 98.3 vs 100.1 us
        // for the forward operator, 115.7 vs 116.7 backward (profiles/r05_backproject_ab.txt; streaming stores / loads measured too: slower)
        RdStart st;
        rd_rows_begin(a.evecs, K, me.row0, me.row0 + me.nrows, wave, lane, st);
        rd_stage_b_nn<DN_TX_THREADS, false, NP>(a.ys + (long long)me.mesh * K * C, C, smem, tid, sb);
        __syncthreads();
        DF_T(1);
        RgArgs rg;
        rg.o0 = a.out; rg.ldo = C; rg.ldr = C; rg.N = C; rg.r0 = a.add; rg.rowv = a.rowv; rg.bias = nullptr; rg.mask = nullptr; rg.rng_seed = 0ull;
        rg.scale = 1.f;
        om = rd_rows_run<MODE, NP>(rg, smem, a.evecs, K, me.row0, me.row0 + me.nrows, 0, lane, st, sa, so);
        DF_T(2);
        __syncthreads();
        DF_WG(1);
    }
    if (a.out_amax) {      // one check-first atomic per workgroup
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) { const float o = __shfl_xor(om, d, 64); om = o > om ? o : om; }
        __syncthreads();
        if (lane == 0) swmax[wave] = om;
        __syncthreads();
        if (tid == 0) {
            float mm = 0.f;
            for (int w = 0; w < DN_TX_THREADS / 64; ++w) mm = swmax[w] > mm ? swmax[w] : mm;
            if (mm > 0.f && mm > *reinterpret_cast<volatile float*>(a.out_amax)) atomicMax(reinterpret_cast<unsigned*>(a.out_amax), __float_as_uint(mm));
        }
    }
}
template <int MODE, int NP>
static int bp_launch(const BpArgs& a, hipStream_t stream) {
    const size_t smem = (size_t)DN_RD_LDS_B + 64;
#ifndef DN_EMULATE
    static unsigned long long lds_opt_in = 0;
    { const int oe_ = dn_lds_opt_in(reinterpret_cast<const void*>(&backproject_kernel<MODE, NP>), smem, &lds_opt_in); if (oe_) return oe_; }
#endif
    DN_LAUNCH((backproject_kernel<MODE, NP>), dim3(a.n_wg, 1, 1), dim3(DN_TX_THREADS, 1, 1), smem, stream, a);
    return (int)hipGetLastError();
}
// out = evecs ys (mass == null) or add + mass * (evecs ys); plan: the FIRST group of a dn_diffuse_plan_host() plan made with ONE group
// f16: run on the split-fp16 engine with operand magnitudes a_amax (Phi) / b_amax (the spectrum)
int dn_launch_backproject(const DnTile* plan, int n_wg, const float* evecs, const float* ys, float* out, const float* add, const float* mass,
                          float* out_amax, double acct_rows, hipStream_t stream, int f16, const DnAmax* a_amax, const DnAmax* b_amax) {
    if (!plan || n_wg <= 0 || (f16 && (!a_amax || !b_amax))) return DN_ERR_BAD_MODE;
    BpArgs a;
    memset(&a, 0, sizeof(a));
    a.plan = plan; a.n_wg = n_wg; a.evecs = evecs; a.ys = ys; a.out = out; a.add = add; a.rowv = mass; a.out_amax = out_amax;
    if (f16) { a.a_amax = *a_amax; a.b_amax = *b_amax; }
    dn_prof_begin(DN_K_BACKPROJECT, stream);
    const int err = f16 ? (mass ? bp_launch<DN_EPI_MASS_ADD, 2>(a, stream) : bp_launch<DN_EPI_STORE, 2>(a, stream))
                        : (mass ? bp_launch<DN_EPI_MASS_ADD, 3>(a, stream) : bp_launch<DN_EPI_STORE, 3>(a, stream));
    dn_prof_end(DN_K_BACKPROJECT, stream, 2.0 * acct_rows * 128 * 128, 4.0 * acct_rows * ((mass ? (add ? 3 : 2) : 2) * 128.0 + (mass ? 1 : 0)));
    return err;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------------------
static size_t df_pad(size_t bytes) { return (bytes + 255) & ~(size_t)255; }
size_t dn_diffuse_ws_bytes(int n_wg, int n_groups, int n_mesh) {
    return df_pad((size_t)n_groups * n_wg * DN_DF_PSTRIDE * sizeof(float)) + df_pad((size_t)n_mesh * 128 * 128 * sizeof(float)) +
           df_pad((size_t)n_wg * 128 * 128 * sizeof(float)) + df_pad((size_t)2 * n_mesh * sizeof(int)) + 256;
}
int dn_diffuse_dt_rows(int n_wg, int n_groups) { return n_wg * n_groups; }

// The schedule for G groups.  order 0 (default): every hand-off one long phase apart (P1a P1b Ra P1c Rb P3a Rc P3b P3c);
// order 1: all projections first, then (R, P3) per group.
int dn_diffuse_schedule(int G, int order, int* sched) {
    int ns = 0;
    auto put = [&](int op, int g) { sched[ns++] = (op << 8) | g; };
    if (order == 1) {
        for (int g = 0; g < G; ++g) put(DN_DF_OP_P1, g);
        for (int g = 0; g < G; ++g) { put(DN_DF_OP_R, g); put(DN_DF_OP_P3, g); }
        return ns;
    }
    put(DN_DF_OP_P1, 0);
    for (int g = 1; g < G; ++g) {
        put(DN_DF_OP_P1, g);
        put(DN_DF_OP_R, g - 1);
        if (g >= 2) put(DN_DF_OP_P3, g - 2);
    }
    put(DN_DF_OP_R, G - 1);
    if (G >= 2) put(DN_DF_OP_P3, G - 2);
    put(DN_DF_OP_P3, G - 1);
    return ns;
}

template <bool BWD>
static int df_launch(const DfArgs& a, int split, hipStream_t stream) {
    const size_t smem = (size_t)2 * 6 * DN_TX_PLANE + 64;
#ifndef DN_EMULATE
    static unsigned long long lds_opt_in = 0;
    { const int oe_ = dn_lds_opt_in(reinterpret_cast<const void*>(&diffuse_kernel<BWD>), smem, &lds_opt_in); if (oe_) return oe_; }
#else
    split = ~0;   // the emulator runs one workgroup after the other: one schedule step per launch (every dependency points to an earlier step)
#endif
    // split: bit i set = a kernel boundary after schedule step i (the polls of the next launch then succeed on their first read)
    int s0 = 0;
    for (int si = 0; si < a.n_sched; ++si) {
        if (si + 1 < a.n_sched && !((split >> si) & 1)) continue;
        DfArgs b = a;
        b.s_begin = s0; b.s_end = si + 1;
        DN_LAUNCH(diffuse_kernel<BWD>, dim3(a.n_wg, 1, 1), dim3(DN_TX_THREADS, 1, 1), smem, stream, b);
        s0 = si + 1;
    }
    return (int)hipGetLastError();
}

// ws: dn_diffuse_ws_bytes(n_wg, n_groups, n_mesh) bytes.  Returns hipError_t as int.
int dn_launch_diffuse(const DfLaunch& L, hipStream_t stream) {
    if (L.n_wg <= 0 || L.n_groups <= 0 || L.n_groups > DN_DF_MAX_GROUPS || !L.plan || !L.ws || !L.time) return DN_ERR_BAD_MODE;
    DfArgs a;
    memset(&a, 0, sizeof(a));
    a.plan = L.plan; a.n_wg = L.n_wg; a.n_groups = L.n_groups;
    a.n_sched = dn_diffuse_schedule(L.n_groups, L.order, a.sched);
    a.flags = L.flags;
    a.s_begin = 0; a.s_end = a.n_sched;
    a.evecs = L.evecs; a.x = L.x; a.mass_in = L.bwd ? nullptr : L.mass; a.evals = L.evals; a.time = L.time;
    a.xs_out = L.bwd ? nullptr : L.xs; a.xs_in = L.bwd ? L.xs : nullptr;
    a.out = L.out; a.add = L.bwd ? L.add : nullptr; a.rowv = L.bwd ? L.mass : nullptr;
    a.dt_part = L.dt_part; a.out_amax = L.out_amax;
    char* p = reinterpret_cast<char*>(L.ws);
    p += (256 - ((uintptr_t)p & 255)) & 255;
    a.partial = reinterpret_cast<float*>(p); p += df_pad((size_t)L.n_groups * L.n_wg * DN_DF_PSTRIDE * sizeof(float));
    a.ys = reinterpret_cast<float*>(p); p += df_pad((size_t)L.n_mesh * 128 * 128 * sizeof(float));
    a.solo = reinterpret_cast<float*>(p); p += df_pad((size_t)L.n_wg * 128 * 128 * sizeof(float));
    a.counters = reinterpret_cast<int*>(p);
    if (L.bwd && (!L.dt_part || !L.xs || !L.mass)) return DN_ERR_BAD_MODE;
    hipError_t e = hipMemsetAsync(a.counters, 0, (size_t)2 * L.n_mesh * sizeof(int), stream);
    if (e != hipSuccess) return (int)e;
    double rows = L.acct_rows;
    dn_prof_begin(DN_K_DIFFUSE, stream);
    const int err = L.bwd ? df_launch<true>(a, L.split, stream) : df_launch<false>(a, L.split, stream);
    dn_prof_end(DN_K_DIFFUSE, stream, 4.0 * rows * 128 * 128, 4.0 * (rows * (4 * 128 + 1) + (double)L.n_mesh * (2 * 128 * 128 + 256)));
    return err;
}

// ---- the plan: which rows of which mesh every workgroup owns in every group (host arithmetic; the caller uploads it)
// sizes: vertices per mesh (row order).  plan: [DN_DF_MAX_GROUPS * n_wg] entries; returns the number of groups used (0: this batch is
// not taken -- more meshes in a group than workgroups).
int dn_diffuse_plan_host(const int* sizes, int n_mesh, int n_wg, int n_groups, DnTile* plan) {
    if (n_mesh <= 0 || n_wg <= 0) return 0;
    int G = n_groups < 1 ? 1 : (n_groups > DN_DF_MAX_GROUPS ? DN_DF_MAX_GROUPS : n_groups);
    if (G > n_mesh) G = n_mesh;
    long long V = 0;
    for (int m = 0; m < n_mesh; ++m) { if (sizes[m] <= 0) return 0; V += sizes[m]; }
    // consecutive meshes per group, nearly equal row totals, every group non-empty
    int gb[DN_DF_MAX_GROUPS + 1];
    gb[0] = 0;
    {
        long long cum = 0;
        int m = 0;
        for (int g = 0; g < G; ++g) {
            const long long target = V * (g + 1) / G;
            const int last_allowed = n_mesh - (G - 1 - g);            // leave one mesh for every later group
            int end = m + 1;
            cum += sizes[m];
            while (end < last_allowed && cum + sizes[end] / 2 < target) { cum += sizes[end]; ++end; }
            if (g == G - 1) { while (end < n_mesh) { cum += sizes[end]; ++end; } }
            gb[g + 1] = end;
            m = end;
        }
    }
    long long row0 = 0;
    for (int g = 0; g < G; ++g) {
        const int m0 = gb[g], m1 = gb[g + 1], nm = m1 - m0;
        if (nm > n_wg) return 0;
        long long rows_g = 0;
        for (int m = m0; m < m1; ++m) rows_g += sizes[m];
        // workgroups per mesh: proportional to its rows, at least 1, at most 128 (a slice reducer owns >= 1 eigenvalue row) and rows / 64
        int cnt[4096], cap[4096];
        if (nm > 4096) return 0;
        int sum = 0;
        for (int i = 0; i < nm; ++i) {
            const int v = sizes[m0 + i];
            cap[i] = v / 64 < 1 ? 1 : (v / 64 > 128 ? 128 : v / 64);
            long long q = (long long)n_wg * v / rows_g;
            cnt[i] = q < 1 ? 1 : (q > cap[i] ? cap[i] : (int)q);
            sum += cnt[i];
        }
        while (sum > n_wg) {                                           // (only when many tiny meshes forced the minimum of one)
            int big = 0;
            for (int i = 1; i < nm; ++i) if (cnt[i] > cnt[big]) big = i;
            if (cnt[big] <= 1) return 0;
            --cnt[big]; --sum;
        }
        bool prog = true;
        while (sum < n_wg && prog) {                                   // remainder to the meshes with the most rows per workgroup
            prog = false;
            int best = -1;
            double bestv = 0.0;
            for (int i = 0; i < nm; ++i)
                if (cnt[i] < cap[i]) { const double r = (double)sizes[m0 + i] / cnt[i]; if (r > bestv) { bestv = r; best = i; } }
            if (best >= 0) { ++cnt[best]; ++sum; prog = true; }
        }
        int slot = g * n_wg;
        for (int i = 0; i < nm; ++i) {
            const int v = sizes[m0 + i], c = cnt[i], first = slot;
            int prev = 0;
            for (int j = 0; j < c; ++j) {
                int end = (j == c - 1) ? v : (int)(((long long)v * (j + 1) / c + 8) / 16 * 16);
                if (end <= prev) end = prev + 1;
                if (end > v - (c - 1 - j)) end = v - (c - 1 - j);
                DnTile t;
                t.row0 = (int)(row0 + prev); t.nrows = end - prev; t.mesh = m0 + i; t.aux = first * 1024 + c;
                plan[slot++] = t;
                prev = end;
            }
            row0 += v;
        }
        for (; slot < (g + 1) * n_wg; ++slot) { DnTile t; t.row0 = 0; t.nrows = 0; t.mesh = -1; t.aux = 0; plan[slot] = t; }
    }
    return G;
}
