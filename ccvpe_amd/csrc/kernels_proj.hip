// Deep-K / narrow-N pointwise GEMM for the MBConv project convolutions of blocks 5-15 (efficientnet_pytorch/model.py:121-122 after
// the squeeze-excite gate of :113-118): [B HW x K] x [K x N] with K = 240 .. 1152 expanded channels and N = 80 .. 320 output
// channels, M = B HW = 6 400 .. 32 768 rows at batch 32.
//
// conv_igemm_kernel runs these layers at 40-78 TFLOP/s (profiles/r03_kernel_trace_b32_fp32.md): with N this narrow a 64 x 64 tile
// grid is 1-3 workgroups per CU deep, every workgroup stages A and B through LDS with one barrier per 16-32 channels, and the
// A operand is re-read once per column block.  Here a workgroup owns RT x 16 rows x ALL N columns, and K - not M or N - is what
// its four waves split: each wave walks its quarter of the channels with both operands straight from L2 / HBM in the MFMA
// fragment order (one 16-byte load per row tile / column tile and 16 channels, the next step in flight under the current step's
// RT x CT x 4 MFMAs), no LDS and no barrier in the K loop, the SE gate multiplied into the A registers.  The four partial sums meet
// once in LDS (two halves of the column tiles, so two workgroups share a CU) and leave through emit_out4 (bias, residual, up to three
// concat destinations).  A is read exactly once per layer; the K x N weight panel (<= 1.5 MB) is streamed from L2 by every workgroup.
#include "igemm_common.h"

#include <algorithm>

namespace ccvpe {

// weights: [16-channel step s][column tile t][lane = (c % 16) / 4 * 16 + n % 16][c % 4]  (1 KiB per (s, t)): A-operand fragments of the
// four MFMA k-steps of a step (k-step j of lane (k, n) holds channel 16 s + 4 k + j - the same permutation the activations' 16-byte
// loads give the B operand)
template <int RT, int CT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_proj_kernel(const ConvParams p) {
    constexpr unsigned OOB = 0x80000000u;
    constexpr int CH = (CT + 1) / 2;           // column tiles per exchange half
    constexpr int UH = RT * CH;                // (row tile, column tile) units per half
    extern __shared__ __attribute__((aligned(16))) float smem[];   // [4 waves][UH][64 lanes][4]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // the K range and the weight offsets are scalar (soffset) operands
    const int kq4 = 4 * (lane >> 4);
    const int hw = p.OH * p.OW;
    const int nsteps = p.Cin >> 4;
    const int per = (nsteps + 3) >> 2;
    const int s_begin = min(wave * per, nsteps), s_end = min(s_begin + per, nsteps);
    const int m0 = xcd_remap(blockIdx.x, gridDim.x) * (16 * RT);

    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, p.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t g_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.gate), 0, p.gate_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.proj_w), 0, p.proj_bytes, 0x00020000);

    unsigned a_off[RT], g_off[RT];
#pragma unroll
    for (int r = 0; r < RT; ++r) {
        const int m = m0 + r * 16 + (lane & 15);
        const bool ok = m < p.M;
        a_off[r] = ok ? (unsigned)m * (unsigned)p.in_ld * 4u + (unsigned)kq4 * 4u : OOB;
        g_off[r] = ok ? (unsigned)(m / hw) * (unsigned)p.Cin * 4u + (unsigned)kq4 * 4u : OOB;
    }
    const unsigned w_lane = (unsigned)lane * 16u;

    f32x4 acc[RT][CT];
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int t = 0; t < CT; ++t) acc[r][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    // Operands: the activations (and their gates) of the NEXT step are requested when a step begins (a second register set, copied over);
    // a column tile's weights are refilled IN PLACE for the next step as soon as its RT x 4 MFMAs have issued - one weight set, a whole
    // step of latency cover.  Every load is unconditional (the last step re-requests its own operands): loads under a runtime
    // condition inside the loop make hipcc drain vmcnt at the joins.
    f32x4 a0[RT], a1[RT], g0[RT], g1[RT], w[CT];
#define CCVPE_PJ_LOAD_A(s_, a_, g_)                                                                                     \
    _Pragma("unroll") for (int r = 0; r < RT; ++r) {                                                                  \
        a_[r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, a_off[r], (s_) * 64, 0));      \
        g_[r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(g_rsrc, g_off[r], (s_) * 64, 0));       \
    }
#define CCVPE_PJ_LOAD_W(s_, t_) w[t_] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, w_lane, ((s_) * CT + (t_)) * 1024, 0))
#define CCVPE_PJ_STEP(a_, g_, sn_)   /* MFMAs of one step on (a_, g_, w); w refilled for step sn_ */                      \
    {                                                                                                                   \
        _Pragma("unroll") for (int r = 0; r < RT; ++r) a_[r] *= g_[r];                                                \
        _Pragma("unroll") for (int t = 0; t < CT; ++t) {                                                              \
            _Pragma("unroll") for (int r = 0; r < RT; ++r) {                                                          \
                acc[r][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[t].x, a_[r].x, acc[r][t], 0, 0, 0);                  \
                acc[r][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[t].y, a_[r].y, acc[r][t], 0, 0, 0);                  \
                acc[r][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[t].z, a_[r].z, acc[r][t], 0, 0, 0);                  \
                acc[r][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[t].w, a_[r].w, acc[r][t], 0, 0, 0);                  \
            }                                                                                                           \
            __builtin_amdgcn_sched_barrier(0);                                                                          \
            CCVPE_PJ_LOAD_W(sn_, t);                                                                                    \
        }                                                                                                               \
    }
    if (s_begin < s_end) {
        CCVPE_PJ_LOAD_A(s_begin, a0, g0);
#pragma unroll
        for (int t = 0; t < CT; ++t) CCVPE_PJ_LOAD_W(s_begin, t);
#pragma unroll 1
        for (int s = s_begin; s < s_end; ++s) {   // one exit, accumulators updated in place (a two-exit ping-pong form made hipcc spill them)
            const int sn = min(s + 1, s_end - 1);
            CCVPE_PJ_LOAD_A(sn, a1, g1);
            __builtin_amdgcn_sched_barrier(0);
            CCVPE_PJ_STEP(a0, g0, sn);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < RT; ++r) { a0[r] = a1[r]; g0[r] = g1[r]; }
        }
    }
#undef CCVPE_PJ_LOAD_A
#undef CCVPE_PJ_LOAD_W
#undef CCVPE_PJ_STEP

    // ---- the four K-partials meet in LDS, one half of the column tiles at a time; wave w finishes units w, w + 4, ... ----
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int t_lo = half * CH;
        const int nct = half == 0 ? CH : CT - CH;      // column tiles of this half
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
            for (int tt = 0; tt < CH; ++tt)
                if (t_lo + tt < CT) *reinterpret_cast<f32x4*>(smem + ((wave * UH + r * CH + tt) * 64 + lane) * 4) = acc[r][t_lo + tt];
        __syncthreads();
        for (int u = wave; u < RT * CH; u += 4) {
            const int r = u / CH, tt = u - r * CH;
            if (tt >= nct) continue;
            f32x4 v = *reinterpret_cast<const f32x4*>(smem + ((0 * UH + u) * 64 + lane) * 4);
#pragma unroll
            for (int w = 1; w < 4; ++w) v += *reinterpret_cast<const f32x4*>(smem + ((w * UH + u) * 64 + lane) * 4);
            const int m = m0 + r * 16 + (lane & 15);
            const int n = (t_lo + tt) * 16 + kq4;
            if (m < p.M && n < p.N) {
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = apply_act(v[i] + (n + i < p.N ? p.bias[n + i] : 0.f), p.act);
                emit_out4(p, m, n, v);
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Latency form (round 4): the same GEMM when M is a few hundred rows (batch 1: 16 x 16 or 32 x 32 pixels).  conv_proj_kernel then has
// 4-16 workgroups whose waves walk 18 dependent steps; the implicit GEMM needs split-K 4-8 with slabs and a reduce launch (17 us for
// block 12 at batch 1, a third of it the second launch).  Here a workgroup is SIXTEEN waves that split K sixteen ways (72 steps of 16
// channels -> 4-5 steps per wave), owns one row tile and CTB column tiles, requests EVERY operand of its waves up front (<= 5 steps x
// (activations + gates + CTB weight tiles) - one trip to L2 / HBM for the whole K loop), runs its 4-5 x CTB x 4 MFMAs and meets the other
// fifteen partial sums in LDS, in wave order.  No slab, no second launch; the grid is (row tiles, column blocks) = 16 x 6-12 workgroups.
// ---------------------------------------------------------------------------------------------------------------------------------
static constexpr int PL_WAVES = 16;
// LS = steps a wave requests at once (a group); K deeper than 16 x LS x 16 channels takes several groups.  The same kernel serves
// (a) the gated project convs (1x1), (b) 1x1 layers without a gate: the ground descriptor heads (models.py:355-395) and the transposed
// convs of the first decoder levels (models.py:407-446; k2s2 transposed = 1x1 with a pixel-shuffle epilogue, emit_out4), (c) the aerial
// descriptor conv k2s2 (models.py:471-482): four taps on disjoint pixels, K = (tap, channel), a lane keeps one base address per tap.
// Channels are padded to 16 per tap (zero weights; the activation read runs into the next pixel's first channels - finite numbers - or
// past the tensor - zeros).
// SEP (GATE must be set too): the squeeze-excite gates are computed HERE, once per workgroup, with ONE trip to memory for everything the
// workgroup needs.  Requested together at the start: the front kernel's squeeze rows (a row of SQ floats per work item; G groups of SQ
// threads take rows g, g + G, ...), both bias vectors, the excite matrix (thread = (channel quad q, third h of the SQ rows): <= 16 loads of
// 16 bytes) and the activations of the wave's K steps.  Then, through LDS: row sums -> sq[j] = swish(b1[j] + mean) -> per-third partial
// excite products -> gates[c] = sigmoid(b2 + the three thirds, in that order); the weight fragments are requested when the excite rows have
// left their registers (L2 hits, under the last two barriers), and every wave reads the gate quads of its steps from LDS.  (The first
// version - every wave computing the gates of its own K slice with a butterfly over 16 lanes - chained four trips: rows, late excite rows,
// excite bias, activations: 20.9 us against 9.7 for the gated form.)
// RTB > 1 (no gate): RTB row tiles per workgroup - a weight fragment serves RTB MFMAs and the layer's weights are read M / (16 RTB) times
// instead of M / 16 times.  The level-6 transposed convs and the aerial descriptor conv at batch 1 (64 rows, 21-26 MB of weights) ran at
// 1 TB/s with four row-tile workgroups per column tile on four different XCDs, each pulling the panel through its own L2.
template <int CTB, bool GATE, int LS, bool SEP, int RTB = 1>
__global__ __launch_bounds__(64 * PL_WAVES) void conv_proj_lat_kernel(const ConvParams p) {
    static_assert(!SEP || GATE, "SEP computes what GATE multiplies");
    static_assert(RTB == 1 || !GATE, "one gate vector per row tile: the multi-row form is for the layers without a gate");
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) float smem[];   // [16 waves][CTB][64 lanes][4]; SEP: group partials and sq[] first
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kq4 = 4 * (lane >> 4);
    const int ohw = p.OH * p.OW;
    const int taps = p.KH * p.KW;
    const int spt = (p.Cin + 15) >> 4;                             // steps per tap
    const int nsteps = spt * taps;
    // self-reducing split-K (layers without a gate; igemm_common.h): gridDim.z workgroups share an output tile, each takes a run of steps,
    // leaves its partial tile in its slab and draws a ticket - 64 rows x 1280 columns x K 5120 (the aerial descriptor conv at batch 1) are 80
    // workgroups otherwise, each pulling 1.6 MB of operands through one CU
    const int nz = (!GATE && p.splitk > 1) ? (int)gridDim.z : 1;
    const int st_lo = nz > 1 ? (int)((long long)nsteps * blockIdx.z / nz) : 0, st_hi = nz > 1 ? (int)((long long)nsteps * (blockIdx.z + 1) / nz) : nsteps;
    const int per = (st_hi - st_lo + PL_WAVES - 1) / PL_WAVES;
    const int s_begin = min(st_lo + wave * per, st_hi), s_end = min(s_begin + per, st_hi);
    const int m0 = blockIdx.x * 16 * RTB;
    const int ct_all = (p.N + 15) >> 4, t0 = blockIdx.y * CTB;

    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, p.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t g_rsrc = SEP ? __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.se_w2), 0, (unsigned)((size_t)p.se_sq * p.Cin * 4), 0x00020000)
                                              : __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(GATE ? p.gate : p.in), 0, GATE ? p.gate_bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.proj_w), 0, p.proj_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t b2_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(SEP ? p.se_b2 : p.in), 0, SEP ? (unsigned)(p.Cin * 4) : 0u, 0x00020000);
    const int m = m0 + (lane & 15);                                // (row of tile 0; tile r: + 16 r)
    const bool ok = m < p.M;
    // base address of this lane's row in every row tile (1x1: the row itself; k2s2: the first pixel of its 2 x 2 patch - the other taps
    // are a uniform distance away and ride in the scalar offset)
    unsigned a_row[RTB];
#pragma unroll
    for (int r = 0; r < RTB; ++r) {
        const int mr = m + 16 * r;
        if (SEP) { a_row[r] = mr < p.M ? (unsigned)mr * (unsigned)p.in_ld * 4u + (unsigned)kq4 * 4u : OOB; continue; }   // (SEP: 1x1 only - the row itself)
        const int mm = mr < p.M ? mr : 0;
        const int b = mm / ohw, rem = mm - b * ohw;
        const int oy = rem / p.OW, ox = rem - oy * p.OW;
        a_row[r] = mr < p.M ? (unsigned)((b * p.H + oy * p.stride) * p.W + ox * p.stride) * (unsigned)p.in_ld * 4u + (unsigned)kq4 * 4u : OOB;
    }
    const unsigned g_off = ok ? (unsigned)(m / ohw) * (unsigned)p.Cin * 4u + (unsigned)kq4 * 4u : OOB;
    const unsigned w_lane = (unsigned)lane * 16u;

    f32x4 acc[CTB][RTB];
#pragma unroll
    for (int t = 0; t < CTB; ++t)
#pragma unroll
        for (int r = 0; r < RTB; ++r) acc[t][r] = f32x4{0.f, 0.f, 0.f, 0.f};
    // SEP: this thread's share of the squeeze rows, of the excite matrix and the biases - requested first, together with the activations below
    const int se_g = SEP ? tid / max(p.se_sq, 1) : 0, se_j = SEP ? tid - se_g * p.se_sq : 0;
    const int se_G = SEP ? min(64 * PL_WAVES / max(p.se_sq, 1), 16) : 1;      // groups: rows g, g + G, ...
    float se_b1v = 0.f;                                            // bias of squeeze output tid
    constexpr int SEU = 9;                                         // squeeze rows per thread: <= 9 x 16 groups
    constexpr int SET = 8;                                         // excite rows per thread and pass: SQ <= 3 x 16 takes two passes (the 64 registers of one do not exist)
    float se_v[SEP ? SEU : 1];
    f32x4 se_w[SEP ? SET : 1], se_b2v = {0.f, 0.f, 0.f, 0.f};
    const int c4n = p.Cin >> 2;
    const int se_h = SEP ? tid / max(c4n, 1) : 0, se_q = SEP ? tid - se_h * c4n : 0;   // third of the SQ rows, channel quad
    const int se_jt = SEP ? (p.se_sq + 2) / 3 : 0;                  // rows per third (<= 16)
    const unsigned wv = (SEP && se_h < 3) ? (unsigned)((se_h * se_jt * p.Cin + 4 * se_q) * 4) : OOB;   // (rows past this third / past SQ: loaded or zero, never summed)
    if (SEP) {
        // (buffer loads with 32-bit offsets: per-thread 64-bit addresses would be registers this kernel does not have)
        const __amdgpu_buffer_rsrc_t r_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.se_rows) + (size_t)(m0 / ohw) * p.se_nrows * p.se_sq, 0,
                                                                                 (unsigned)((size_t)p.se_nrows * p.se_sq * 4), 0x00020000);   // (a row tile lies inside one sample: checked by the host)
        // (one address register per thread, the step between a thread's loads rides in the scalar offset; rows past the last are past the
        //  descriptor's range: zeros, no traffic)
        const unsigned rv = se_g < se_G ? (unsigned)((se_g * p.se_sq + se_j) * 4) : OOB;
#pragma unroll
        for (int u = 0; u < SEU; ++u) se_v[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_rsrc, rv, u * se_G * p.se_sq * 4, 0));
        const __amdgpu_buffer_rsrc_t b1_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.se_b1), 0, (unsigned)(p.se_sq * 4), 0x00020000);
        se_b1v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(b1_rsrc, (unsigned)(tid * 4), 0, 0));   // (threads past SQ: out of range, zero)
#pragma unroll
        for (int u = 0; u < SET; ++u) se_w[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(g_rsrc, wv, u * p.Cin * 4, 0));
        se_b2v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(b2_rsrc, se_h == 0 ? (unsigned)(16 * se_q) : OOB, 0, 0));
    }
    int sg = s_begin;
#pragma unroll 1
    do {   // (at least one pass per wave, also for a wave without steps - all its requests out of range: the SEP barriers below are for everybody)
        // every operand of the group, requested at once (steps past s_end and column tiles past the layer's ask for out-of-range
        // offsets: zeros, no traffic)
        f32x4 a[LS][RTB], g[LS], w[LS][CTB];
#pragma unroll
        for (int i = 0; i < LS; ++i) {
            const int s = sg + i;
            const bool live = s < s_end;
            const int tap = (SEP || taps == 1) ? 0 : s / spt;        // (scalar: s is wave-uniform; SEP: one tap)
            const int sub = s - tap * spt;
            const int tap_off = (SEP || taps == 1) ? 0 : ((tap / p.KW) * p.W + tap % p.KW) * p.in_ld * 4;   // (scalar)
#pragma unroll
            for (int r = 0; r < RTB; ++r) a[i][r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, live ? a_row[r] : OOB, tap_off + sub * 64, 0));
            if (GATE && !SEP) g[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(g_rsrc, live ? g_off : OOB, sub * 64, 0));
            if (!SEP) {   // (SEP: the weight fragments follow once the excite rows have left their registers)
#pragma unroll
                for (int t = 0; t < CTB; ++t)
                    w[i][t] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, (live && t0 + t < ct_all) ? w_lane : OOB, (s * ct_all + t0 + t) * 1024, 0));
            }
        }
        __builtin_amdgcn_sched_barrier(0);   // (hipcc otherwise sinks the requests between the MFMAs to save registers: one memory latency per step)
        if (SEP) {
            // LDS behind the exchange area (a fast wave may already write its partial sums while a slow one still reads the gates):
            // [G][SQ] group sums | sq[64] | [3][Cin] per-third excite products | gates[Cin]
            float* part = smem + PL_WAVES * CTB * 256;
            float* sq = part + 16 * 64;
            float* gpart = sq + 64;
            float* gates = gpart + 3 * p.Cin;
            {                                                        // (one group per wave: the launcher)
                float sum = 0.f;
#pragma unroll
                for (int u = 0; u < SEU; ++u) sum += se_v[u];
                if (se_g < se_G) part[se_g * p.se_sq + se_j] = sum;
                __syncthreads();
                if (tid < p.se_sq) {
                    float v = 0.f;
                    for (int g2 = 0; g2 < se_G; ++g2) v += part[g2 * p.se_sq + tid];
                    v = v * p.se_inv_hw + se_b1v;
                    sq[tid] = v * __builtin_amdgcn_rcpf(1.f + __expf(-v));
                }
                __syncthreads();
                f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int u = 0; u < SET; ++u) {
                    const int jj = se_h * se_jt + u;
                    const float sv = (se_h < 3 && u < se_jt && jj < p.se_sq) ? sq[jj] : 0.f;
                    z[0] = fmaf(se_w[u][0], sv, z[0]); z[1] = fmaf(se_w[u][1], sv, z[1]); z[2] = fmaf(se_w[u][2], sv, z[2]); z[3] = fmaf(se_w[u][3], sv, z[3]);
                }
                __builtin_amdgcn_sched_barrier(0);
                const bool pass2 = se_jt > SET;                      // (uniform: the blocks with more than 24 squeezed channels; L2 hits - every workgroup reads this matrix)
                if (pass2) {
#pragma unroll
                    for (int u = 0; u < SET; ++u) se_w[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(g_rsrc, wv, (SET + u) * p.Cin * 4, 0));
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < LS; ++i) {                       // the weight fragments: L2 hits, under the two barriers below
                    const int s = sg + i;
#pragma unroll
                    for (int t = 0; t < CTB; ++t)
                        w[i][t] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, (s < s_end && t0 + t < ct_all) ? w_lane : OOB, (s * ct_all + t0 + t) * 1024, 0));
                }
                __builtin_amdgcn_sched_barrier(0);
                if (pass2) {
#pragma unroll
                    for (int u = 0; u < SET; ++u) {
                        const int jj = se_h * se_jt + SET + u;
                        const float sv = (se_h < 3 && SET + u < se_jt && jj < p.se_sq) ? sq[jj] : 0.f;
                        z[0] = fmaf(se_w[u][0], sv, z[0]); z[1] = fmaf(se_w[u][1], sv, z[1]); z[2] = fmaf(se_w[u][2], sv, z[2]); z[3] = fmaf(se_w[u][3], sv, z[3]);
                    }
                }
                if (se_h < 3 && se_q < c4n) *reinterpret_cast<f32x4*>(gpart + se_h * p.Cin + 4 * se_q) = z;
                __syncthreads();
                if (se_h == 0 && se_q < c4n) {
                    const f32x4 zz = se_b2v + *reinterpret_cast<const f32x4*>(gpart + 4 * se_q) + *reinterpret_cast<const f32x4*>(gpart + p.Cin + 4 * se_q) +
                                     *reinterpret_cast<const f32x4*>(gpart + 2 * p.Cin + 4 * se_q);
                    f32x4 gq;
#pragma unroll
                    for (int c = 0; c < 4; ++c) gq[c] = 1.f / (1.f + __expf(-zz[c]));
                    *reinterpret_cast<f32x4*>(gates + 4 * se_q) = gq;
                }
                __syncthreads();
            }
#pragma unroll
            for (int i = 0; i < LS; ++i) {
                const int s = sg + i;
                if (s < s_end) a[i][0] *= *reinterpret_cast<const f32x4*>(gates + 16 * s + kq4);   // (channels 16 s + 4 kq .. + 3 of this lane's fragment)
            }
        }
#pragma unroll
        for (int i = 0; i < LS; ++i) {
            if (GATE && !SEP) a[i][0] *= g[i];
#pragma unroll
            for (int t = 0; t < CTB; ++t)
#pragma unroll
                for (int r = 0; r < RTB; ++r) {
                    acc[t][r] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[i][t].x, a[i][r].x, acc[t][r], 0, 0, 0);
                    acc[t][r] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[i][t].y, a[i][r].y, acc[t][r], 0, 0, 0);
                    acc[t][r] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[i][t].z, a[i][r].z, acc[t][r], 0, 0, 0);
                    acc[t][r] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[i][t].w, a[i][r].w, acc[t][r], 0, 0, 0);
                }
        }
        __builtin_amdgcn_sched_barrier(0);
        sg += LS;
    } while (sg < s_end);
    // ---- the sixteen K-partials meet in LDS; wave t < CTB x RTB adds (column tile, row tile) t in wave order and stores it ----
    constexpr int NTL = CTB * RTB;
    static_assert(NTL <= PL_WAVES, "one wave per tile of the workgroup in the epilogue");
#pragma unroll
    for (int t = 0; t < CTB; ++t)
#pragma unroll
        for (int r = 0; r < RTB; ++r) *reinterpret_cast<f32x4*>(smem + ((wave * NTL + t * RTB + r) * 64 + lane) * 4) = acc[t][r];
    __syncthreads();
    const int ct_w = wave / RTB, rt_w = wave - ct_w * RTB;           // (scalar)
    if (wave < NTL && t0 + ct_w < ct_all) {
        f32x4 v = *reinterpret_cast<const f32x4*>(smem + ((0 * NTL + wave) * 64 + lane) * 4);
#pragma unroll
        for (int w2 = 1; w2 < PL_WAVES; ++w2) v += *reinterpret_cast<const f32x4*>(smem + ((w2 * NTL + wave) * 64 + lane) * 4);
        const int n = (t0 + ct_w) * 16 + kq4;
        const int mo = m + 16 * rt_w;
        if (mo < p.M && n < p.N) {
            if (nz > 1) {   // partial sums of this K slice -> its slab, write-through
                float* dst = p.partial + ((size_t)blockIdx.z * p.M + mo) * p.N + n;
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (n + i < p.N) st_sc1(dst + i, v[i]);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = apply_act(v[i] + (n + i < p.N ? p.bias[n + i] : 0.f), p.act);
                emit_out4(p, mo, n, v);
            }
        }
    }
    if (!GATE && nz > 1) {   // (uniform: every thread of the workgroup; the exchange area is dead behind the ticket's first barrier)
        if (splitk_ticket(p, blockIdx.y * gridDim.x + blockIdx.x, reinterpret_cast<unsigned*>(smem)))
            splitk_finish<64 * PL_WAVES>(p, m0, 1, 16 * RTB, 0, t0 * 16, 16 * CTB);
    }
}

static int proj_lat_steps(const ConvParams& p) { return ((p.Cin + 15) / 16) * p.KH * p.KW; }

template <int CTB, bool GATE, int LS, bool SEP = false, int RTB = 1>
static void launch_proj_lat2(const ConvParams& p, hipStream_t s) {
    const size_t lds = ((size_t)PL_WAVES * CTB * RTB * 64 * 4 + (SEP ? 16 * 64 + 64 + 4 * (size_t)p.Cin : 0)) * sizeof(float);
    dim3 grid((p.M + 16 * RTB - 1) / (16 * RTB), ((p.N + 15) / 16 + CTB - 1) / CTB);
    if (!GATE && p.splitk > 1) grid.z = p.splitk;   // (launch_conv_igemm: only with ticket counters, a slab and <= CONV_TICKETS tiles)
    static LdsAttr attr;
    auto kern = conv_proj_lat_kernel<CTB, GATE, LS, SEP, RTB>;
    ensure_dynamic_lds(attr, reinterpret_cast<const void*>(kern), lds);
    CCVPE_LAUNCH(kern, grid, dim3(64 * PL_WAVES), lds, s, p);
}
template <int CTB>
static void launch_proj_lat(const ConvParams& p, hipStream_t s) {
    if (p.se_rows) { launch_proj_lat2<1, true, 5, true>(p, s); return; }   // (gates in the prologue: one column tile per workgroup - registers)
    if (p.gate) { launch_proj_lat2<(CTB > 2 ? 2 : CTB), true, 5>(p, s); return; }   // (never CTB 4 with a gate: conv_proj_supported)
    if (CTB == 1 && proj_lat_steps(p) > PL_WAVES * 5) { launch_proj_lat2<1, false, 10>(p, s); return; }   // deep K: ten steps per group
    launch_proj_lat2<CTB, false, 5>(p, s);
}

struct ProjCfg { int rt, ct; };
static constexpr ProjCfg PROJ_CFGS[] = {{2, 5}, {4, 5}, {2, 7}, {4, 7}, {1, 12}, {2, 12}, {1, 20}};
static constexpr int PROJ_NCFG = (int)(sizeof(PROJ_CFGS) / sizeof(PROJ_CFGS[0]));

template <int RT, int CT>
static void launch_proj_cfg(const ConvParams& p, hipStream_t s) {
    constexpr size_t lds = (size_t)4 * RT * ((CT + 1) / 2) * 64 * 4 * sizeof(float);
    static LdsAttr attr;
    auto kern = conv_proj_kernel<RT, CT>;
    ensure_dynamic_lds(attr, reinterpret_cast<const void*>(kern), lds);
    const int wgs = (p.M + 16 * RT - 1) / (16 * RT);
    CCVPE_LAUNCH(kern, dim3(wgs), dim3(256), lds, s, p);
}

// column tiles of a layer the packed weights were made for
static int proj_ct(const ConvParams& p) { return (p.N + 15) / 16; }

// rt >= 100: the latency form with rt - 100 column tiles per workgroup (any layer width; K <= 1280; a few thousand rows at most - beyond
// that its (row tile, column block) grid re-reads the operands too often to be worth timing)
bool conv_proj_supported(const ConvParams& p, int rt) {
    if (rt >= 100) {
        const bool one = p.KH == 1 && p.KW == 1 && p.stride == 1 && p.OH == p.H && p.OW == p.W;
        const bool k2s2 = p.KH == 2 && p.KW == 2 && p.stride == 2 && p.OH * 2 == p.H && p.OW * 2 == p.W && p.gate == nullptr && p.mode == MODE_CONV;
        if (!(p.proj_w != nullptr && (one || k2s2) && p.pad_t == 0 && p.pad_l == 0 && !p.in_split && (p.mode == MODE_CONV || p.mode == MODE_DECONV) &&
              p.in_ld % 4 == 0 && p.Cin >= 64 && p.M <= 4096)) return false;
        if ((p.gate != nullptr || p.se_rows != nullptr) && p.Cin % 16 != 0) return false;      // (the gate vector has exactly Cin entries per sample)
        if (rt == 104 && p.gate != nullptr) return false;              // (gates + four column tiles of weights per step: past the 128 registers of a wave)
        if (rt > 104 && (p.gate != nullptr || p.se_rows != nullptr || p.M < 16 * ((rt - 100) / 10) || p.M > 1024)) return false;   // the multi-row forms: no gate, at least one full workgroup of rows
        if (p.se_rows != nullptr) {   // gates computed in the prologue: one group per wave, every wave busy, a row tile inside one sample, <= 64 squeeze outputs
            const int steps = proj_lat_steps(p);
            if (rt != 101 || !one || steps > PL_WAVES * 5 || p.se_sq > 48 || p.se_sq < 1 || p.se_nrows > 9 * std::min(1024 / p.se_sq, 16) ||
                3 * (p.Cin / 4) > 64 * PL_WAVES || !(p.B == 1 || (p.OH * p.OW) % 16 == 0)) return false;
        }
        const int steps = proj_lat_steps(p);
        return steps <= PL_WAVES * 40;
    }
    if (!(p.proj_w != nullptr && p.gate != nullptr && p.KH == 1 && p.KW == 1 && p.stride == 1 && p.pad_t == 0 && p.pad_l == 0 && p.OH == p.H && p.OW == p.W &&
          !p.in_split && p.mode == MODE_CONV && p.Cin % 16 == 0 && p.in_ld % 4 == 0 && p.Cin >= 64)) return false;
    for (int i = 0; i < PROJ_NCFG; ++i)
        if (PROJ_CFGS[i].rt == rt && PROJ_CFGS[i].ct == proj_ct(p)) return true;
    return false;
}

bool conv_proj_has(int rt, int N) {   // an instantiated (row tiles, column tiles) pair
    if (rt >= 100) return true;
    for (int i = 0; i < PROJ_NCFG; ++i)
        if (PROJ_CFGS[i].rt == rt && PROJ_CFGS[i].ct == (N + 15) / 16) return true;
    return false;
}

void launch_proj(const ConvParams& p, int rt, hipStream_t s) {
    if (rt == 101) { launch_proj_lat<1>(p, s); return; }
    if (rt == 102) { launch_proj_lat<2>(p, s); return; }
    if (rt == 104) { launch_proj_lat<4>(p, s); return; }
    if (rt == 121) { launch_proj_lat2<1, false, 8, false, 2>(p, s); return; }   // two / four row tiles per workgroup (no gate)
    if (rt == 141) { launch_proj_lat2<1, false, 4, false, 4>(p, s); return; }
    const int ct = proj_ct(p);
    if (rt == 2 && ct == 5) launch_proj_cfg<2, 5>(p, s);
    else if (rt == 4 && ct == 5) launch_proj_cfg<4, 5>(p, s);
    else if (rt == 2 && ct == 7) launch_proj_cfg<2, 7>(p, s);
    else if (rt == 4 && ct == 7) launch_proj_cfg<4, 7>(p, s);
    else if (rt == 1 && ct == 12) launch_proj_cfg<1, 12>(p, s);
    else if (rt == 2 && ct == 12) launch_proj_cfg<2, 12>(p, s);
    else if (rt == 1 && ct == 20) launch_proj_cfg<1, 20>(p, s);
}

// ... and for the latency form: any 1x1 layer (projects, descriptor heads, transposed convs) or k2s2 conv (aerial descriptor map) whose
// K is deep enough that the implicit GEMM splits it at batch 1
bool conv_proj_lat_wanted(int taps, int KH, int KW, int cinp) {
    return taps * cinp >= 480 && ((taps == 1 && KH == 1 && KW == 1) || (taps == 4 && KH == 2 && KW == 2));
}

// layers the packer makes the fragment-order copy for: the gated project convs with a deep K and one of the widths above
bool conv_proj_wanted(int N, int cin) {
    const int ct = (N + 15) / 16;
    return cin % 16 == 0 && cin >= 192 && (ct == 5 || ct == 7 || ct == 12 || ct == 20);
}

// `get(n, k)` returns the folded weight of output channel n at GEMM column k = tap * cin + channel; every tap's channels are padded to a
// multiple of 16 (zero weights): [step = tap * ceil(cin / 16) + c / 16][column tile][lane = (c % 16) / 4 * 16 + n % 16][c % 4]
size_t conv_proj_pack(int N, int cin, const std::function<float(int, int)>& get, std::vector<float>& out, int taps) {
    const int ct = (N + 15) / 16, spt = (cin + 15) / 16, nsteps = spt * taps;
    out.assign((size_t)nsteps * ct * 256, 0.f);
    for (int n = 0; n < N; ++n)
        for (int t = 0; t < taps; ++t)
            for (int c = 0; c < cin; ++c) {
                const int s = t * spt + c / 16, k = (c % 16) / 4, j = c % 4;
                out[(((size_t)s * ct + n / 16) * 64 + k * 16 + (n % 16)) * 4 + j] = get(n, t * cin + c);
            }
    return out.size();
}

}  // namespace ccvpe
