// Last-arriver tickets: how a kernel of libccvpe_hip.so finishes work that needs the results of ALL its workgroups without a second
// launch and without anybody waiting - the squeeze-excite MLP behind the fused MBConv fronts (efficientnet_pytorch/model.py:113-118)
// and the sum over the K slices of a split-K convolution (models.py:407-446 at small batch).  Every producer publishes its partial
// result, then draws a ticket on a device-scope counter; whoever draws the last one does the combining step.  No spin loop anywhere,
// so a grid that is only partly resident (two streams share the chip) cannot hang.
//
// Visibility (cdna_hip_programming.md, Guideline 16; the L2 of an XCD is not coherent with the other seven):
//   producer   every handed-off byte is stored write-through (`sc1`: st_sc1 / st_sc1_f4), every storing wave drains its stores
//              (`s_waitcnt vmcnt(0)`), the workgroup meets at a barrier, ONE lane adds to the counter (agent-scope atomic);
//   consumer   the lane whose add returned the last ticket issues ONE agent-scope acquire (invalidates this CU's L1) and waits for it,
//              the workgroup meets at a barrier, then every wave reads - with `sc1` loads (ld_sc1 / ld_sc1_f4), so a line another CU
//              rewrote is never served from a stale L1 copy either way.
// The counters are zero between launches: the last arriver puts its counter back to zero (the memory is zeroed once, when the plan
// that owns it is built), so a captured hipGraph needs no memset node and a replay starts from the same state.
// Results do not depend on who arrives last: the combining step reads EVERY partial from memory in a fixed order.
#pragma once
#include <hip/hip_runtime.h>

namespace ccvpe {

typedef float tk_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned tk_u32x4 __attribute__((__vector_size__(4 * sizeof(unsigned))));

__device__ __forceinline__ void st_sc1(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_sc1(const float* p) { return __hip_atomic_load(const_cast<float*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// 16-byte forms through a buffer descriptor (aux bit 4 = sc1); `off` in bytes, out-of-range offsets store nothing / load zeros
__device__ __forceinline__ void st_sc1_f4(__amdgpu_buffer_rsrc_t rsrc, unsigned off, tk_f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(tk_u32x4, v), rsrc, off, 0, 16);
}
__device__ __forceinline__ tk_f32x4 ld_sc1_f4(__amdgpu_buffer_rsrc_t rsrc, unsigned off) {
    return __builtin_bit_cast(tk_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 16));
}

// Every thread of the workgroup calls this after its last hand-off store.  `n` tickets are drawn at once (a workgroup that finished n
// of the `total` work items of this counter).  Returns - to every thread the same value - whether this draw completed the count; the
// caller then reads the partials with the ld_sc1 forms.  `flag`: one LDS word nobody touches between the call and the next barrier
// the caller executes after reading the result.
__device__ __forceinline__ bool ticket_arrive(unsigned* counter, unsigned n, unsigned total, unsigned* flag) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave: its sc1 stores have left the CU
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned old = __hip_atomic_fetch_add(counter, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool last = old + n == total;
        if (last) {
            __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // back to zero for the next launch
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");                              // this CU's L1 holds nothing stale
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                // ... once the invalidate has completed
        }
        *flag = last ? 1u : 0u;
    }
    __syncthreads();
    const bool last = *flag != 0u;
    __syncthreads();   // (the callers' flag word is scratch they reuse at once: nobody rewrites it before everybody has read it)
    return last;
}

// ------------------------------------------------------------------------------------------------------------------------------
// Squeeze-excite behind a fused MBConv front kernel (model.py:113-118: global average pool -> 1x1 conv C -> SQ + swish -> 1x1 conv
// SQ -> C + sigmoid).  The front kernels leave per-workgroup channel sums in `pool` ([B][S][C] partial rows, sc1 stores); the
// workgroup that draws the last ticket of sample b reduces the S rows in row order, runs the two small matrix-vector products and
// writes gate[b][0..C).  The project conv of the block (the next launch) multiplies the gate into its A operand.
// ------------------------------------------------------------------------------------------------------------------------------
struct SeTicket {
    unsigned* counter;         // [B] tickets drawn per sample; nullptr = the squeeze-excite runs as its own launches (launch_se)
    int per_sample;            // tickets that complete a sample
    const float* pool;         // [B][S][C] partial sums of the depthwise output (written by this launch); unused when `sqpart` is set
    int S, C, SQ;              // C % 4 == 0, SQ <= 64
    float inv_hw;
    const float* w1;           // [SQ][C]
    const float* b1;           // [SQ]
    const float* w2;           // [SQ][C]
    const float* b2;           // [C]
    float* gate;               // [B][C]
    // Distributed squeeze (mbconv_image_kernel): the first 1x1 conv is linear in the pooled sums, so every work item (a strip x 16
    // channels) adds ITS channels' share sum_c w1[j][c] * poolsum[c] as a row of SQ floats; the last arriver only sums the rows (in item
    // order), applies bias + swish and runs the second conv.  The [SQ][C] matrix w1 - 221 KB for the 1152-channel blocks - is then read
    // by the ~72 workgroups that own its columns instead of by one CU at the end of the launch.
    float* sqpart;             // [B][per_sample][SQ] or nullptr
    int spec;                  // latency plans: request the excite weights before the ticket is drawn (se_arrive_and_finish_parts_spec)
};
// LDS scratch of the combining step: [ partial products / group partials 1024 | squeezed 64 | flag 4 | pooled C <= 1152 ]; the form
// with the distributed squeeze needs the first three only
static constexpr int SE_SCR_PART = 0, SE_SCR_SQ = 1024, SE_SCR_FLAG = 1088, SE_SCR_POOLED = 1092;
static constexpr int SE_SCRATCH_PARTS_FLOATS = 1092;
static constexpr int SE_SCRATCH_FLOATS = 1092 + 1152;

__device__ __forceinline__ float se_swish(float v) { return v * __builtin_amdgcn_rcpf(1.f + __expf(-v)); }

// The combining step runs on ONE workgroup while the rest of the chip waits for the launch to end, and every step of it is a round trip
// to memory (~1-2 us each under load): the code below keeps the round trips few and overlapped - row sums read in batches of independent
// loads that are added in row order afterwards, and the first batch of the excite weights (read-only, never handed off) requested before
// the sums they will be multiplied with exist.

// sum of rows r0, r0 + stride, ... < n of a [n][ld] array at column `col`, in row order; the loads go out UB at a time
template <int UB>
__device__ __forceinline__ float se_column_sum(const float* base, int ld, int col, int r0, int stride, int n) {
    float a = 0.f;
    for (int r = r0; r < n; r += UB * stride) {
        float v[UB];
#pragma unroll
        for (int u = 0; u < UB; ++u) v[u] = r + u * stride < n ? ld_sc1(base + (size_t)(r + u * stride) * ld + col) : 0.f;
#pragma unroll
        for (int u = 0; u < UB; ++u) a += v[u];     // (a row past the end adds +0: the sum of the real rows is unchanged)
    }
    return a;
}

// excite (second 1x1 conv + sigmoid, model.py:116-118): a thread owns 4 consecutive channels and walks the SQ weight rows JB at a time (JB
// independent 16-byte loads in flight).  Split in two so that the first batch can be requested early: se_excite_fetch before the squeezed
// vector exists, se_excite_apply once it is in LDS.
template <int JB>
struct SeExciteRegs { tk_f32x4 w[JB]; tk_f32x4 b2; float b1; };   // + the two biases this thread will need: every load that waits behind a
                                                                     //   barrier of the combining step is one more trip to memory on the launch's tail
template <int NT, int JB>
__device__ __forceinline__ void se_excite_fetch(const SeTicket& t, SeExciteRegs<JB>& r) {
    const int c = min((int)threadIdx.x * 4, t.C - 4);      // (threads past the last channel quad re-read it: no branch around the loads)
#pragma unroll
    for (int u = 0; u < JB; ++u) r.w[u] = *reinterpret_cast<const tk_f32x4*>(t.w2 + (size_t)min(u, t.SQ - 1) * t.C + c);
    r.b2 = *reinterpret_cast<const tk_f32x4*>(t.b2 + c);
    r.b1 = t.b1[min((int)threadIdx.x, t.SQ - 1)];
}
template <int NT, int JB>
__device__ __forceinline__ void se_excite_apply(const SeTicket& t, int b, const float* sq, SeExciteRegs<JB>& r) {
    bool first = true;
    for (int c = threadIdx.x * 4; c < t.C; c += NT * 4) {
        tk_f32x4 a = first ? r.b2 : *reinterpret_cast<const tk_f32x4*>(t.b2 + c);
        for (int j0 = 0; j0 < t.SQ; j0 += JB) {
            if (!first) {
#pragma unroll
                for (int u = 0; u < JB; ++u) r.w[u] = *reinterpret_cast<const tk_f32x4*>(t.w2 + (size_t)min(j0 + u, t.SQ - 1) * t.C + c);
            }
            first = false;
#pragma unroll
            for (int u = 0; u < JB; ++u) {
                const float sv = j0 + u < t.SQ ? sq[j0 + u] : 0.f;
                a[0] = fmaf(r.w[u][0], sv, a[0]); a[1] = fmaf(r.w[u][1], sv, a[1]); a[2] = fmaf(r.w[u][2], sv, a[2]); a[3] = fmaf(r.w[u][3], sv, a[3]);
            }
        }
        tk_f32x4 g;
#pragma unroll
        for (int e = 0; e < 4; ++e) g[e] = 1.f / (1.f + __expf(-a[e]));
        *reinterpret_cast<tk_f32x4*>(t.gate + (size_t)b * t.C + c) = g;
    }
}

// Called by ALL NT threads of the workgroup that drew the last ticket of sample b; `scr` = SE_SCRATCH_FLOATS floats of LDS that are
// free (16-byte aligned).  Pooling partial rows -> gates (the few-channel blocks: stem_dw_kernel, mbconv_front_wave_kernel).
template <int NT, int JB>
__device__ __forceinline__ void se_finish_jb(const SeTicket& t, int b, float* scr) {
    const int tid = threadIdx.x;
    float* pooled = scr + SE_SCR_POOLED;
    float* part = scr + SE_SCR_PART;
    float* sq = scr + SE_SCR_SQ;
    const float* pp = t.pool + (size_t)b * t.S * t.C;
    SeExciteRegs<JB> er;
    se_excite_fetch<NT, JB>(t, er);
    // squeeze weights of this thread's first (output j, channel quad) product, requested before the pooled sums exist
    const int c4n = t.C >> 2, nprod = t.SQ * c4n;
    const int pj = min(tid, nprod - 1) / c4n, pc = (min(tid, nprod - 1) - pj * c4n) * 4;
    const tk_f32x4 w1v = *reinterpret_cast<const tk_f32x4*>(t.w1 + (size_t)pj * t.C + pc);
    // (1) pooled mean: the S partial rows in row order.  Few channels and many rows (stem, block 1): G groups of C threads take
    // rows g, g + G, ... and their sums meet in LDS in group order; else a thread owns channels tid, tid + NT, ...
    if (t.C * 2 <= NT) {
        const int G = NT / t.C;
        const int g = tid / t.C, c = tid - g * t.C;
        if (g < G) part[g * t.C + c] = se_column_sum<32>(pp, t.C, c, g, G, t.S);   // (block 1 at batch 1: 128 rows over two groups - two trips, not eight)
        __syncthreads();
        if (tid < t.C) {
            float v = 0.f;
            for (int g2 = 0; g2 < G; ++g2) v += part[g2 * t.C + tid];
            pooled[tid] = v * t.inv_hw;
        }
    } else {
        for (int c = tid; c < t.C; c += NT) pooled[c] = se_column_sum<16>(pp, t.C, c, 0, 1, t.S) * t.inv_hw;
    }
    __syncthreads();
    // (2) squeeze: one (output j, channel quad) product per thread and round into LDS, then thread j adds its C / 4 products in
    // channel order
    if (nprod <= 1024) {
        for (int i = tid; i < nprod; i += NT) {
            const int j = i / c4n, c = (i - j * c4n) * 4;
            const tk_f32x4 wv = i == tid ? w1v : *reinterpret_cast<const tk_f32x4*>(t.w1 + (size_t)j * t.C + c);
            const tk_f32x4 pv = *reinterpret_cast<const tk_f32x4*>(pooled + c);
            part[i] = fmaf(wv[3], pv[3], fmaf(wv[2], pv[2], fmaf(wv[1], pv[1], wv[0] * pv[0])));
        }
        __syncthreads();
        if (tid < t.SQ) {
            float a = 0.f;
            for (int k = 0; k < c4n; ++k) a += part[tid * c4n + k];
            sq[tid] = se_swish(a + er.b1);
        }
    } else {   // wide blocks without the distributed squeeze: wave w owns outputs w, w + NT / 64, ...
        const int lane = tid & 63, wave = tid >> 6;
        for (int j = wave; j < t.SQ; j += NT / 64) {
            const float* wr = t.w1 + (size_t)j * t.C;
            float a = 0.f;
            for (int c = lane * 4; c < t.C; c += 256) {
                const tk_f32x4 wv = *reinterpret_cast<const tk_f32x4*>(wr + c);
                const tk_f32x4 pv = *reinterpret_cast<const tk_f32x4*>(pooled + c);
                a = fmaf(wv[0], pv[0], a); a = fmaf(wv[1], pv[1], a); a = fmaf(wv[2], pv[2], a); a = fmaf(wv[3], pv[3], a);
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off);
            if (lane == 0) sq[j] = se_swish(a + t.b1[j]);
        }
    }
    __syncthreads();
    se_excite_apply<NT, JB>(t, b, sq, er);
}
// (The front kernels call these AFTER their persistent loops, never inside: next to the loops' prefetch state the ~100 registers of the
//  combining step spill, and a kernel that touches scratch - also through a non-inlined call - starts ~10 us later: measured on every
//  launch of the batch-1 plan.)
static constexpr int SE_JB = 16;   // excite weight rows in flight per thread (SQ <= 16: all of them; the 480 .. 1152-channel blocks: 20 .. 48 rows)
template <int NT>
__device__ __forceinline__ void se_finish(const SeTicket& t, int b, float* scr) { se_finish_jb<NT, SE_JB>(t, b, scr); }

// The same for a launch with the distributed squeeze: sum the per_sample rows of SQ floats in row order (G groups of SQ threads take
// rows g, g + G, ...; the groups meet in LDS in group order), bias + swish, excite.
template <int NT, int JB, bool PRE>
__device__ __forceinline__ void se_finish_parts_jb(const SeTicket& t, int b, float* scr, SeExciteRegs<JB>& er) {
    const int tid = threadIdx.x;
    float* part = scr + SE_SCR_PART;
    float* sq = scr + SE_SCR_SQ;
    if (!PRE) se_excite_fetch<NT, JB>(t, er);
    const float* rows = t.sqpart + (size_t)b * t.per_sample * t.SQ;
    const int G = min(NT / t.SQ, 1024 / t.SQ);
    const int g = tid / t.SQ, j = tid - g * t.SQ;
    if (g < G) part[g * t.SQ + j] = se_column_sum<24>(rows, t.SQ, j, g, G, t.per_sample);   // (latency plans with strips: ~200 rows over ten groups - one trip)
    __syncthreads();
    if (tid < t.SQ) {
        float v = 0.f;
        for (int g2 = 0; g2 < G; ++g2) v += part[g2 * t.SQ + tid];
        sq[tid] = se_swish(v * t.inv_hw + er.b1);
    }
    __syncthreads();
    se_excite_apply<NT, JB>(t, b, sq, er);
}
template <int NT>
__device__ __forceinline__ void se_finish_parts(const SeTicket& t, int b, float* scr) {
    SeExciteRegs<SE_JB> er;
    se_finish_parts_jb<NT, SE_JB, false>(t, b, scr, er);
}
// Latency plans (the launch's workgroups do not even fill the chip once: batch <= 4): a workgroup whose items all belong to ONE sample
// draws its ticket behind its loop, and requests the excite weights BEFORE it knows whether the ticket is the last one - the request
// then travels under the store drain, the ticket's round trip and the acquire instead of behind them.  All SQ rows at once (SE_JB_SPEC).
static constexpr int SE_JB_SPEC = 32;   // (48 - every row of the widest blocks - spills: 192 registers of weights)
#ifndef CCVPE_SE_CLOCK
#define CCVPE_SE_CLOCK 0   // dev builds (tools/build_variant.sh): 1 = the last arriver stamps s_memrealtime (100 MHz) along the combining step
#endif
#if CCVPE_SE_CLOCK
__device__ unsigned long long g_se_clk[16];
#define CCVPE_SE_STAMP(i_) { if (threadIdx.x == 0) stamp[i_] = wall_clock64(); }
#else
#define CCVPE_SE_STAMP(i_)
#endif
template <int NT>
__device__ __forceinline__ void se_arrive_and_finish_parts_spec(const SeTicket& t, int b, unsigned n, float* scr) {
    unsigned* flag = reinterpret_cast<unsigned*>(scr + SE_SCR_FLAG);
#if CCVPE_SE_CLOCK
    unsigned long long stamp[8];
#endif
    CCVPE_SE_STAMP(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // as ticket_arrive: every storing wave has drained its sc1 stores
    __syncthreads();
    CCVPE_SE_STAMP(1);
    unsigned old = 0;
    if (threadIdx.x == 0) old = __hip_atomic_fetch_add(t.counter + b, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    SeExciteRegs<SE_JB_SPEC> er;
    se_excite_fetch<NT, SE_JB_SPEC>(t, er);            // read-only weights, never handed off: in flight under the ticket and the acquire
    if (threadIdx.x == 0) {
        const bool last = old + n == (unsigned)t.per_sample;
        CCVPE_SE_STAMP(2);
        if (last) {
            __hip_atomic_store(t.counter + b, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        CCVPE_SE_STAMP(3);
        *flag = last ? 1u : 0u;
    }
    __syncthreads();
    if (*flag != 0u) {
        CCVPE_SE_STAMP(4);
        se_finish_parts_jb<NT, SE_JB_SPEC, true>(t, b, scr, er);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        CCVPE_SE_STAMP(5);
#if CCVPE_SE_CLOCK
        if (threadIdx.x == 0) {
            for (int i = 0; i < 6; ++i) g_se_clk[i] = stamp[i];
        }
#endif
    }
}

}  // namespace ccvpe
