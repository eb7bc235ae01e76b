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
static constexpr int PL_STEPS = 5;             // steps a wave holds at most: K <= 16 x 5 x 16 = 1280 channels
template <int CTB, bool GATE>
__global__ __launch_bounds__(64 * PL_WAVES) void conv_proj_lat_kernel(const ConvParams p) {
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) float smem[];   // [16 waves][CTB][64 lanes][4]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kq4 = 4 * (lane >> 4);
    const int hw = p.OH * p.OW;
    const int nsteps = p.Cin >> 4;
    const int per = (nsteps + PL_WAVES - 1) / PL_WAVES;            // <= PL_STEPS (checked by the launcher)
    const int s_begin = min(wave * per, nsteps), s_end = min(s_begin + per, nsteps);
    const int m0 = blockIdx.x * 16;
    const int ct_all = (p.N + 15) >> 4, t0 = blockIdx.y * CTB;

    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, p.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t g_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(GATE ? p.gate : p.in), 0, GATE ? p.gate_bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.proj_w), 0, p.proj_bytes, 0x00020000);
    const int m = m0 + (lane & 15);
    const bool ok = m < p.M;
    const unsigned a_off = ok ? (unsigned)m * (unsigned)p.in_ld * 4u + (unsigned)kq4 * 4u : OOB;
    const unsigned g_off = ok ? (unsigned)(m / hw) * (unsigned)p.Cin * 4u + (unsigned)kq4 * 4u : OOB;
    const unsigned w_lane = (unsigned)lane * 16u;

    // every operand of this wave's steps, requested at once (steps past s_end and column tiles past the layer's ask for out-of-range
    // offsets: zeros, no traffic)
    f32x4 a[PL_STEPS], g[PL_STEPS], w[PL_STEPS][CTB];
#pragma unroll
    for (int i = 0; i < PL_STEPS; ++i) {
        const int s = s_begin + i;
        const bool live = s < s_end;
        a[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, live ? a_off : OOB, s * 64, 0));
        if (GATE) g[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(g_rsrc, live ? g_off : OOB, s * 64, 0));
#pragma unroll
        for (int t = 0; t < CTB; ++t)
            w[i][t] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, (live && t0 + t < ct_all) ? w_lane : OOB, (s * ct_all + t0 + t) * 1024, 0));
    }
    __builtin_amdgcn_sched_barrier(0);   // (hipcc otherwise sinks the requests between the MFMAs to save registers: one memory latency per step)
    f32x4 acc[CTB];
#pragma unroll
    for (int t = 0; t < CTB; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < PL_STEPS; ++i) {
        if (GATE) a[i] *= g[i];
#pragma unroll
        for (int t = 0; t < CTB; ++t) {
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[i][t].x, a[i].x, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[i][t].y, a[i].y, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[i][t].z, a[i].z, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[i][t].w, a[i].w, acc[t], 0, 0, 0);
        }
    }
    // ---- the sixteen K-partials meet in LDS; wave t < CTB adds column tile t in wave order and stores it ----
#pragma unroll
    for (int t = 0; t < CTB; ++t) *reinterpret_cast<f32x4*>(smem + ((wave * CTB + t) * 64 + lane) * 4) = acc[t];
    __syncthreads();
    if (wave < CTB && t0 + wave < ct_all) {
        f32x4 v = *reinterpret_cast<const f32x4*>(smem + ((0 * CTB + wave) * 64 + lane) * 4);
#pragma unroll
        for (int w2 = 1; w2 < PL_WAVES; ++w2) v += *reinterpret_cast<const f32x4*>(smem + ((w2 * CTB + wave) * 64 + lane) * 4);
        const int n = (t0 + wave) * 16 + kq4;
        if (ok && n < p.N) {
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = apply_act(v[i] + (n + i < p.N ? p.bias[n + i] : 0.f), p.act);
            emit_out4(p, m, n, v);
        }
    }
}

template <int CTB>
static void launch_proj_lat(const ConvParams& p, hipStream_t s) {
    constexpr size_t lds = (size_t)PL_WAVES * CTB * 64 * 4 * sizeof(float);
    const dim3 grid((p.M + 15) / 16, ((p.N + 15) / 16 + CTB - 1) / CTB);
    static LdsAttr attr_g, attr_n;
    if (p.gate) {
        ensure_dynamic_lds(attr_g, reinterpret_cast<const void*>(conv_proj_lat_kernel<CTB, true>), lds);
        hipLaunchKernelGGL((conv_proj_lat_kernel<CTB, true>), grid, dim3(64 * PL_WAVES), lds, s, p);
    } else {
        ensure_dynamic_lds(attr_n, reinterpret_cast<const void*>(conv_proj_lat_kernel<CTB, false>), lds);
        hipLaunchKernelGGL((conv_proj_lat_kernel<CTB, false>), grid, dim3(64 * PL_WAVES), lds, s, p);
    }
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
    hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), lds, s, p);
}

// column tiles of a layer the packed weights were made for
static int proj_ct(const ConvParams& p) { return (p.N + 15) / 16; }

// rt >= 100: the latency form with rt - 100 column tiles per workgroup (any layer width; K <= 1280; a few thousand rows at most - beyond
// that its (row tile, column block) grid re-reads the operands too often to be worth timing)
bool conv_proj_supported(const ConvParams& p, int rt) {
    if (rt >= 100)
        return p.proj_w != nullptr && p.KH == 1 && p.KW == 1 && p.stride == 1 && p.pad_t == 0 && p.pad_l == 0 && p.OH == p.H && p.OW == p.W && !p.in_split &&
               p.mode == MODE_CONV && p.Cin % 16 == 0 && p.in_ld % 4 == 0 && p.Cin >= 64 && (p.Cin >> 4) <= PL_WAVES * PL_STEPS && p.M <= 4096;
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
    const int ct = proj_ct(p);
    if (rt == 2 && ct == 5) launch_proj_cfg<2, 5>(p, s);
    else if (rt == 4 && ct == 5) launch_proj_cfg<4, 5>(p, s);
    else if (rt == 2 && ct == 7) launch_proj_cfg<2, 7>(p, s);
    else if (rt == 4 && ct == 7) launch_proj_cfg<4, 7>(p, s);
    else if (rt == 1 && ct == 12) launch_proj_cfg<1, 12>(p, s);
    else if (rt == 2 && ct == 12) launch_proj_cfg<2, 12>(p, s);
    else if (rt == 1 && ct == 20) launch_proj_cfg<1, 20>(p, s);
}

// layers the packer makes the fragment-order copy for: the gated project convs with a deep K and one of the widths above
bool conv_proj_wanted(int N, int cin) {
    const int ct = (N + 15) / 16;
    return cin % 16 == 0 && cin >= 192 && (ct == 5 || ct == 7 || ct == 12 || ct == 20);
}

// `get(n, c)` returns the folded 1x1 weight
size_t conv_proj_pack(int N, int cin, const std::function<float(int, int)>& get, std::vector<float>& out) {
    const int ct = (N + 15) / 16, nsteps = cin / 16;
    out.assign((size_t)nsteps * ct * 256, 0.f);
    for (int n = 0; n < N; ++n)
        for (int c = 0; c < cin; ++c) {
            const int s = c / 16, k = (c % 16) / 4, j = c % 4;
            out[(((size_t)s * ct + n / 16) * 64 + k * 16 + (n % 16)) * 4 + j] = get(n, c);
        }
    return out.size();
}

}  // namespace ccvpe
