// Winograd F(4x4, 3x3), xi-split form for the NARROW decoder layers (<= 128 output channels: conv2 / conv3 and their _ori twins,
// reference models.py:42-47, 417-421, 438-442): the 36 Winograd positions are dealt to the waves of a workgroup (nine each)
// instead of the output channels.
//
// Why: kernels_wino4.hip gives every wave one 16-channel slice x all 36 xi, so a layer with 40 output channels (3 slices)
// idles one wave of four - and with it one SIMD's matrix pipe - through the whole MFMA phase, a layer with 80 (5 slices of an
// 8-wave workgroup) runs 72 MFMAs per k-step on one SIMD and 36 on the other three.  Here a wave owns xi 9q .. 9q + 8 of ALL the
// workgroup's slices (S of them: 27 MFMAs per k-step and wave for 40 channels, 45 per SIMD for 80), the V operand of an xi is
// read from LDS once per wave instead of once per slice (5 ds_read_b64 per k-step instead of 18), and the weights arrive in
// ceil(9 S / 4) 16-byte loads.  The price is the epilogue: the inverse transform A^T M A needs all 36 xi of a (tile, channel), so
// the accumulators cross LDS once per tile - one 16-channel slice per round, thread = (tile, channel), which also spreads the
// transform's ~100 VALU operations per (tile, channel) over every lane of the workgroup.
//
// One n-block per layer (the configuration is picked by the layer's width), everything else - raw-patch staging, the input
// transform B^T d B in registers, the persistent XCD-aware grid, split-K - as kernels_wino4.hip.
#include "igemm_common.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <type_traits>

namespace ccvpe {

typedef float f32x2 __attribute__((ext_vector_type(2)));

static constexpr int X4_PITCH = 20;                 // floats per raw row (18 used)
static constexpr int X4_PLANE = 385;                // floats per raw channel plane (1 mod 64)
static constexpr int X4_GCH = 16;                   // input channels per group (4 k-steps)
static constexpr int X4_VFLOATS = 4 * 4 * 5 * 128;  // V image: [k-step][xi quarter][pair slot 5][lane][2]
static constexpr int X4_RAW_F4 = 18 * 18 * 4;

__device__ __forceinline__ void x4_bt(const float d0, const float d1, const float d2, const float d3, const float d4, const float d5,
                                      float& t0, float& t1, float& t2, float& t3, float& t4, float& t5) {
    const float a = fmaf(-4.f, d2, d4), b = fmaf(-4.f, d1, d3);
    const float c = d4 - d2, e = d3 - d1;
    t0 = fmaf(4.f, d0, fmaf(-5.f, d2, d4));
    t1 = a + b;
    t2 = a - b;
    t3 = fmaf(2.f, e, c);
    t4 = fmaf(-2.f, e, c);
    t5 = fmaf(4.f, d1, fmaf(-5.f, d3, d5));
}
__device__ __forceinline__ void x4_at(const float m0, const float m1, const float m2, const float m3, const float m4, const float m5,
                                      float& y0, float& y1, float& y2, float& y3) {
    const float s1 = m1 + m2, d1 = m1 - m2, s2 = m3 + m4, d2 = m3 - m4;
    y0 = m0 + s1 + s2;
    y1 = fmaf(2.f, d2, d1);
    y2 = fmaf(4.f, s2, s1);
    y3 = fmaf(8.f, d2, d1) + m5;
}

// float offset of xi inside one k-step of the V image: quarter q = xi / 9, position xl + (q & 1) inside it (odd quarters start at an
// odd xi: shifting them by one keeps every (even xi, xi + 1) pair of one quarter on an aligned slot)
__host__ __device__ constexpr int x4_vpos(int xi) {
    const int q = xi / 9, pos = xi - 9 * q + (q & 1);
    return (q * 5 + (pos >> 1)) * 128 + (pos & 1);
}

// weights: bytes of one k-step = sum over the waves of their 16-byte loads (1 KiB per load and wave)
__host__ __device__ constexpr int x4_loads(int s) { return (9 * s + 3) / 4; }
__host__ __device__ constexpr int x4_step_loads(int nw, int s0, int s1) { return 4 * x4_loads(s0) + (nw == 8 ? 4 * x4_loads(s1) : 0); }

#ifndef CCVPE_X4_DEEP
#define CCVPE_X4_DEEP 1     // dev builds: 0 = one k-step of weight prefetch everywhere
#endif
#ifndef CCVPE_X4_SWITCH
#define CCVPE_X4_SWITCH 0   // dev builds, timing only (wrong results): bit 0 no epilogue, 1 no transform, 2 no raw staging, 3 no MFMAs, 4 no weight refills
#endif
#ifndef CCVPE_X4_CLOCK
#define CCVPE_X4_CLOCK 0   // dev builds (tools/build_variant.sh): 1 = every workgroup stamps s_memtime / s_memrealtime around its work loop
#endif
#if CCVPE_X4_CLOCK
__device__ unsigned long long g_x4_stamps[4 * 1024];
#endif

template <int NW, int S0, int S1>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_wino4x_kernel(const ConvParams p) {
#if CCVPE_X4_CLOCK
    const unsigned long long clk0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
#endif
    static_assert((NW == 4 && S1 == 0) || (NW == 8 && S1 > 0), "waves 0-3 own slices [0, S0), waves 4-7 slices [S0, S0 + S1)");
    static_assert(S0 >= S1 && S0 <= 4, "at most 36 accumulator tiles per wave");
    constexpr int NT = NW * 64;
    constexpr int NS = S0 + S1;                            // 16-channel slices of the workgroup = of the layer
    constexpr int RS = NW / 4;                             // slices per epilogue round (thread = (slice of the round, tile, channel))
    constexpr int ROUNDS = (NS + RS - 1) / RS;
    constexpr int RAW_ITEMS = NW == 4 ? 6 : 3;
    constexpr int RAW_HALVES = RAW_ITEMS / 3;
    // NW 4 stages six float4 per thread and group.  EARLY: all six are requested when the group's MFMA phase begins and stored when
    // it ends (a whole group of cover for the HBM latency under load, 24 registers held); otherwise in two halves of three through
    // the same 12 registers (the second half requested after the first k-step: one k-step of cover for the first half)
    constexpr bool EARLY = NW == 4 && S0 <= 3;
    // DEEP: the weights are requested TWO k-steps ahead (two register sets).  vmcnt counts a wave's loads and stores in issue order, so
    // the first wait for a weight load that is younger than the raw-patch loads (or than the previous tile's output stores) also waits
    // for those: with one k-step of weight prefetch the HBM latency of the patch was exposed after one k-step whatever the patch's own
    // prefetch distance (switch experiments: staging 22 %, epilogue 21 % of conv2.0); two k-steps of cover hide most of it.
    constexpr bool DEEP = CCVPE_X4_DEEP && S0 <= 3;      // (the 64- and 128-channel configurations have no registers left for the second set)
    constexpr int NBQ = DEEP ? 2 : 1;
    constexpr int MXQ = 72;                                // floats of one (xi, channel quad): 16 tiles x 4 channels + 8 pad (the readers'
                                                           // four quads then start on banks 0 / 8 / 16 / 24: conflict-free ds_read_b32)
    constexpr int MXI = 4 * MXQ;                           // floats per xi
    constexpr int MX_FLOATS = RS * 36 * MXI;               // exchange image of one round: [slice][xi][channel quad][tile][4]
    constexpr int V_REGION = X4_VFLOATS > MX_FLOATS ? X4_VFLOATS : MX_FLOATS;
    constexpr unsigned OOB = 0x80000000u;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Vs = smem;                 // V image; the epilogue's exchange image aliases it
    float* Rs = smem + V_REGION;      // [16][X4_PLANE]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int q = wave & 3;           // xi quarter of this wave
    const bool hi = wave >= 4;        // second slice group (NW 8)

    const int mbx = p.W >> 4, mby = p.H >> 4;
    const int total = p.B * mbx * mby;

    const int xcd = blockIdx.x & 7;
    const int stride = ((int)gridDim.x >> 3) + (xcd < ((int)gridDim.x & 7) ? 1 : 0);
    const int item_begin = xcd * (total >> 3) + min(xcd, total & 7);
    const int item_end = item_begin + (total >> 3) + (xcd < (total & 7) ? 1 : 0);
    int item = item_begin + ((int)blockIdx.x >> 3);
    if (item >= item_end) return;

    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, p.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wino4x_w), 0, p.wino4x_bytes, 0x00020000);

    int b, by, bx;
#define CCVPE_X4_DECODE(it_, b_, by_, bx_)                                                               \
    {                                                                                                    \
        b_ = (it_) / (mbx * mby);                                                                        \
        const int rem_ = (it_) - b_ * (mbx * mby);                                                       \
        by_ = rem_ / mbx;                                                                                \
        bx_ = rem_ - by_ * mbx;                                                                          \
    }
    CCVPE_X4_DECODE(item, b, by, bx);

    // ---- raw patch staging (kernels_wino4.hip) ----
    unsigned r_off[RAW_ITEMS];
    const int r_ch = (tid & 3) * 4;
    const int r_px0 = tid >> 2;
#define CCVPE_X4_ROFF(b_, by_, bx_, live_)                                                               \
    _Pragma("unroll") for (int i = 0; i < RAW_ITEMS; ++i) {                                              \
        int j = tid + i * NT;                                                                            \
        asm volatile("" : "+v"(j));   /* recomputed per item, not hoisted into scratch (kernels_wino4.hip) */ \
        const int px = j >> 2, qq = j & 3;                                                               \
        const int py = px / 18, pxx = px - py * 18;                                                      \
        const int y = (by_) * 16 - 1 + py, x = (bx_) * 16 - 1 + pxx;                                     \
        const bool ok = (live_) && j < X4_RAW_F4 && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W; \
        r_off[i] = ok ? (unsigned)(((((b_) * p.H + y) * p.W + x) * p.in_ld + qq * 4) * 4) : OOB;         \
    }
    CCVPE_X4_ROFF(b, by, bx, true);

    // ---- transform item of this lane: channel 4 * (wave & 3) + (lane & 3) of the group, tile (ty, tx) = (lane >> 4, (lane >> 2) & 3);
    //      NW 8: waves w and w + 4 share the item (output rows 0-2 / 3-5)
    const int tk = lane & 3, ttx = (lane >> 2) & 3, tty = lane >> 4;
    const int tks = wave & 3;
    const float* t_src = Rs + (4 * tks + tk) * X4_PLANE + (4 * tty) * X4_PITCH + 4 * ttx;
    float* t_dst = Vs + tks * (20 * 128) + (tk * 16 + tty * 4 + ttx) * 2;   // + x4_vpos(xi)

    // weights: [k-step][wave][load][lane][4]; unit u = 4 load + e of a wave = (xi 9q + u / S, slice u % S of its group)
    constexpr unsigned w_step_b = (unsigned)x4_step_loads(NW, S0, S1) * 1024u;
    const unsigned w_base = (unsigned)((wave < 4 ? wave * x4_loads(S0) : 4 * x4_loads(S0) + (wave - 4) * x4_loads(S1)) * 1024 + lane * 16);

    const int ngr_all = (p.Cin + X4_GCH - 1) / X4_GCH;
    int g_begin = 0, g_end = ngr_all;
    if (p.splitk > 1) {
        const int per = (ngr_all + p.splitk - 1) / p.splitk;
        g_begin = min((int)blockIdx.z * per, ngr_all);
        g_end = min(g_begin + per, ngr_all);
    }
    if (g_begin >= g_end) return;
    const int tail_ks = (p.Cin - (ngr_all - 1) * X4_GCH + 3) >> 2;

    f32x4 raw[EARLY ? 6 : 3];
    float bq[NBQ][9 * S0];    // B fragments (weights) of the current k-step(s), refilled in place for the k-step NBQ ahead
#define CCVPE_X4_LOAD_RAW(c0, half)                                                                      \
    _Pragma("unroll") for (int i = 0; i < 3; ++i) {                                                    \
        const unsigned o_ = ((c0) + r_ch < p.Cin) ? r_off[3 * (half) + i] : OOB;                         \
        raw[(EARLY ? 3 * (half) : 0) + i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, o_, (c0) * 4, 0)); \
    }
#define CCVPE_X4_STORE_RAW(half)                                                                         \
    _Pragma("unroll") for (int i = 0; i < 3; ++i) {                                                    \
        const int px_ = r_px0 + (NT / 4) * (3 * (half) + i);                                             \
        if (px_ < 18 * 18) {                                                                             \
            const int py_ = (px_ * 3641) >> 16;                                                          \
            float* d_ = Rs + r_ch * X4_PLANE + py_ * X4_PITCH + (px_ - py_ * 18);                        \
            const f32x4 rv_ = raw[(EARLY ? 3 * (half) : 0) + i];                                         \
            d_[0] = rv_.x; d_[X4_PLANE] = rv_.y; d_[2 * X4_PLANE] = rv_.z; d_[3 * X4_PLANE] = rv_.w;     \
        }                                                                                                \
    }
#define CCVPE_X4_LOAD_B(set_, ks, ld_)   /* 16-byte load ld_ of k-step ks (global index) into register set set_: units 4 ld_ .. 4 ld_ + 3 */ \
    {                                                                                                    \
        const f32x4 t4_ = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, w_base, (ks) * w_step_b + (ld_) * 1024u, 0)); \
        if (4 * (ld_) + 0 < 9 * S0) bq[set_][4 * (ld_) + 0] = t4_.x;                                     \
        if (4 * (ld_) + 1 < 9 * S0) bq[set_][4 * (ld_) + 1] = t4_.y;                                     \
        if (4 * (ld_) + 2 < 9 * S0) bq[set_][4 * (ld_) + 2] = t4_.z;                                     \
        if (4 * (ld_) + 3 < 9 * S0) bq[set_][4 * (ld_) + 3] = t4_.w;                                     \
    }

    f32x4 acc[9 * S0];
#pragma unroll
    for (int x = 0; x < 9 * S0; ++x) acc[x] = f32x4{0.f, 0.f, 0.f, 0.f};

    CCVPE_X4_LOAD_RAW(g_begin * X4_GCH, 0);
#pragma unroll
    for (int l = 0; l < x4_loads(S0); ++l) { CCVPE_X4_LOAD_B(0, g_begin * 4, l); }
    if (DEEP) {
#pragma unroll
        for (int l = 0; l < x4_loads(S0); ++l) { CCVPE_X4_LOAD_B(NBQ - 1, g_begin * 4 + 1, l); }
    }
    CCVPE_X4_STORE_RAW(0);
    if (RAW_HALVES == 2) {
        CCVPE_X4_LOAD_RAW(g_begin * X4_GCH, 1);
        CCVPE_X4_STORE_RAW(1);
    }
    __syncthreads();

    const bool split = p.splitk > 1;
    const int ld = split ? p.N : p.dst[0].ld;
    const int act = split ? ACT_NONE : p.act;
    // V fragments of this wave: pair slots 0 .. 4 of its quarter
    const float* va0 = Vs + q * (5 * 128) + lane * 2;

    while (true) {
        const int item_n = item + stride;
        const bool have_n = item_n < item_end;
        int b_n, by_n, bx_n;
        CCVPE_X4_DECODE(have_n ? item_n : item, b_n, by_n, bx_n);

        for (int g = g_begin; g < g_end; ++g) {
            const bool last_group = g == g_end - 1;
            const int nks = g == ngr_all - 1 ? tail_ks : 4;
            // ---- transform B^T d B (kernels_wino4.hip): one half-item = output rows 3 HF .. 3 HF + 2 ----
            auto half_item = [&](auto hfc) {
                constexpr int HF = decltype(hfc)::value;
                float t[3][6];
                float dc[2][6];
#pragma unroll
                for (int r = 0; r < 6; ++r) dc[0][r] = t_src[r * X4_PITCH];
#pragma unroll
                for (int c = 0; c < 6; ++c) {
                    if (c + 1 < 6) {
#pragma unroll
                        for (int r = 0; r < 6; ++r) dc[(c + 1) & 1][r] = t_src[r * X4_PITCH + c + 1];
                    }
                    float u0, u1, u2, u3, u4, u5;
                    x4_bt(dc[c & 1][0], dc[c & 1][1], dc[c & 1][2], dc[c & 1][3], dc[c & 1][4], dc[c & 1][5], u0, u1, u2, u3, u4, u5);
                    t[0][c] = HF ? u3 : u0; t[1][c] = HF ? u4 : u1; t[2][c] = HF ? u5 : u2;
                }
#pragma unroll
                for (int ii = 0; ii < 3; ++ii) {
                    float v[6];
                    x4_bt(t[ii][0], t[ii][1], t[ii][2], t[ii][3], t[ii][4], t[ii][5], v[0], v[1], v[2], v[3], v[4], v[5]);
                    // xi = 6 i + j, i = 3 HF + ii; pairs (xi even, xi + 1) of one quarter are one 8-byte write, the two pairs that
                    // straddle a quarter boundary ((8, 9) and (26, 27)) two 4-byte writes
#pragma unroll
                    for (int j = 0; j < 6; j += 2) {
                        const int xi = 6 * (3 * HF + ii) + j;
                        if (xi / 9 == (xi + 1) / 9) {
                            *reinterpret_cast<f32x2*>(t_dst + x4_vpos(xi)) = f32x2{v[j], v[j + 1]};
                        } else {
                            t_dst[x4_vpos(xi)] = v[j];
                            t_dst[x4_vpos(xi + 1)] = v[j + 1];
                        }
                    }
                }
            };
            if (tks >= nks || (CCVPE_X4_SWITCH & 2)) {
            } else if (NW == 4) {
                half_item(std::integral_constant<int, 0>{});
                half_item(std::integral_constant<int, 1>{});
            } else if (wave < 4) {
                half_item(std::integral_constant<int, 0>{});
            } else {
                half_item(std::integral_constant<int, 1>{});
            }
            __syncthreads();   // V image complete; every wave is done with the raw patch
            const int c0n = last_group ? g_begin * X4_GCH : (g + 1) * X4_GCH;
            if (last_group) { CCVPE_X4_ROFF(b_n, by_n, bx_n, have_n); }
            if (!(CCVPE_X4_SWITCH & (4 | 32))) {
                CCVPE_X4_LOAD_RAW(c0n, 0);
                if (EARLY) { CCVPE_X4_LOAD_RAW(c0n, 1); }
            }
            __builtin_amdgcn_sched_barrier(0);
            // ---- MFMA phase: per k-step 9 xi x S slices; PAR = parity of the quarter (position of xi 9q + xl inside the slots),
            //      S = slices of this wave: both compile-time inside ----
            auto mfma_phase = [&](auto parc, auto sc) {
                constexpr int PAR = decltype(parc)::value;
                constexpr int S = decltype(sc)::value;
                constexpr int NL = x4_loads(S);
#pragma unroll 1
                for (int ks0 = 0; ks0 < nks; ks0 += NBQ) {   // nks is 2 or 4
#pragma unroll
                    for (int hb = 0; hb < NBQ; ++hb) {
                        const int ks = ks0 + hb;
                        int ksn;             // k-step (global index) this set is refilled for: NBQ k-steps ahead in the workgroup's stream
                        if (ks + NBQ < nks) ksn = g * 4 + ks + NBQ;
                        else ksn = (last_group ? g_begin : g + 1) * 4 + (ks + NBQ - nks);
                        const float* va = va0 + ks * (20 * 128);
                        f32x2 fa[5];
#pragma unroll
                        for (int s5 = 0; s5 < 5; ++s5) fa[s5] = *reinterpret_cast<const f32x2*>(va + s5 * 128);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int xl = 0; xl < 9; ++xl) {
                            const float v = ((xl + PAR) & 1) ? fa[(xl + PAR) >> 1].y : fa[(xl + PAR) >> 1].x;
#pragma unroll
                            for (int sl = 0; sl < S; ++sl) {
                                const int u = xl * S + sl;
                                // weights as the A operand: D[channel][tile]; a lane ends up with 4 consecutive channels of ONE tile
                                if (!(CCVPE_X4_SWITCH & 8)) acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(bq[hb][u], v, acc[u], 0, 0, 0);
                                else acc[u][0] += v;
                                if (((u & 3) == 3 || u == 9 * S - 1) && !(CCVPE_X4_SWITCH & 16)) {   // the units of a 16-byte weight load have issued: refill it
                                    __builtin_amdgcn_sched_barrier(0);
                                    CCVPE_X4_LOAD_B(hb, ksn, u >> 2);
                                    (void)NL;
                                }
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        if (RAW_HALVES == 2 && !EARLY && ks == 1 && !(CCVPE_X4_SWITCH & 4)) {
                            CCVPE_X4_STORE_RAW(0);
                            CCVPE_X4_LOAD_RAW(c0n, 1);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                }
            };
            if (!hi) {
                if (q & 1) mfma_phase(std::integral_constant<int, 1>{}, std::integral_constant<int, S0>{});
                else mfma_phase(std::integral_constant<int, 0>{}, std::integral_constant<int, S0>{});
            } else {
                if (q & 1) mfma_phase(std::integral_constant<int, 1>{}, std::integral_constant<int, (S1 > 0 ? S1 : 1)>{});
                else mfma_phase(std::integral_constant<int, 0>{}, std::integral_constant<int, (S1 > 0 ? S1 : 1)>{});
            }
            if (!(CCVPE_X4_SWITCH & (4 | 64))) {
                if (EARLY) { CCVPE_X4_STORE_RAW(0); }
                CCVPE_X4_STORE_RAW(RAW_HALVES - 1);
            } else if (CCVPE_X4_SWITCH & 64) {
                float keep_ = 0.f;
#pragma unroll
                for (int i = 0; i < (EARLY ? 6 : 3); ++i) keep_ += raw[i].x + raw[i].y + raw[i].z + raw[i].w;
                if (keep_ == 12345.f) Rs[tid] = keep_;
            }
            __syncthreads();   // V image free again; next raw patch complete
        }

        if (CCVPE_X4_SWITCH & 1) {   // timing only: one store per lane keeps the accumulators alive
            float sum_ = 0.f;
#pragma unroll
            for (int x = 0; x < 9 * S0; ++x) sum_ += acc[x][0] + acc[x][1] + acc[x][2] + acc[x][3];
            if (sum_ == 12345.f) p.dst[0].ptr[tid] = sum_;
        } else
        // ---- epilogue: accumulators -> LDS (one round = RS slices), thread = (slice of the round, tile, channel) gathers its 36 xi,
        //      applies A^T M A, bias, activation, and stores its channel of the tile's 4 x 4 output pixels ----
        {
            const size_t pix0 = ((size_t)b * p.H + (size_t)by * 16) * p.W + (size_t)bx * 16;
            float* obase = split ? p.partial + ((size_t)blockIdx.z * p.M + pix0) * p.N : p.dst[0].ptr + pix0 * ld + p.dst[0].coff;
            const __amdgpu_buffer_rsrc_t o_rsrc = __builtin_amdgcn_make_buffer_rsrc(obase, 0, 0x7fffffff, 0x00020000);
            const int e_ch = tid & 15, e_tile = (tid >> 4) & 15, e_sl = tid >> 8;   // e_sl: 0 (NW 4) or 0 / 1 (NW 8)
            const int oty = e_tile >> 2, otx = e_tile & 3;
            const float lo = act == ACT_RELU ? 0.f : -__builtin_inff();
            const int wq_tile = lane & 15, wq_cq = lane >> 4;
#pragma unroll
            for (int r = 0; r < ROUNDS; ++r) {
                // writers: slice sg = r * RS + e of the round lives at Mx + e * 36 * 256; the accumulator index is compile-time in
                // either slice group (a runtime index would send the accumulators to scratch)
#pragma unroll
                for (int e = 0; e < RS; ++e) {
                    const int sg = r * RS + e;
                    if (sg >= NS) continue;
                    float* mxw = Vs + e * (36 * MXI) + (9 * q) * MXI + wq_cq * MXQ + wq_tile * 4;   // 16 lanes = 256 contiguous bytes
                    if (sg < S0) {
                        if (!hi) {
#pragma unroll
                            for (int xl = 0; xl < 9; ++xl) *reinterpret_cast<f32x4*>(mxw + xl * MXI) = acc[xl * S0 + sg];
                        }
                    } else if (S1 > 0) {
                        if (hi) {
#pragma unroll
                            for (int xl = 0; xl < 9; ++xl) *reinterpret_cast<f32x4*>(mxw + xl * MXI) = acc[xl * (S1 > 0 ? S1 : 1) + (sg - S0)];
                        }
                    }
                }
                __syncthreads();
                const int sg = r * RS + e_sl;
                const int n = sg * 16 + e_ch;
                if (sg < NS && n < p.N) {
                    const float* mx = Vs + e_sl * (36 * MXI) + (e_ch >> 2) * MXQ + e_tile * 4 + (e_ch & 3);
                    float tt[4][6];
#pragma unroll
                    for (int c = 0; c < 6; ++c)
                        x4_at(mx[(0 * 6 + c) * MXI], mx[(1 * 6 + c) * MXI], mx[(2 * 6 + c) * MXI], mx[(3 * 6 + c) * MXI], mx[(4 * 6 + c) * MXI], mx[(5 * 6 + c) * MXI],
                              tt[0][c], tt[1][c], tt[2][c], tt[3][c]);
                    const float bias = split ? 0.f : p.bias[n];
                    const unsigned o_lane = (unsigned)((((oty * 4) * p.W + otx * 4) * ld + n) * 4);
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        float yy[4];
                        x4_at(tt[a][0], tt[a][1], tt[a][2], tt[a][3], tt[a][4], tt[a][5], yy[0], yy[1], yy[2], yy[3]);
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const float t = yy[c] + bias;
                            const int soff = ((a * p.W + c) * ld) * 4;
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, t < lo ? lo : t), o_rsrc, o_lane, soff, 0);
                        }
                    }
                }
                __syncthreads();   // the exchange image is free for the next round / the next tile's V image
            }
        }
        if (!have_n) break;
#pragma unroll
        for (int x = 0; x < 9 * S0; ++x) acc[x] = f32x4{0.f, 0.f, 0.f, 0.f};
        item = item_n; b = b_n; by = by_n; bx = bx_n;
    }
#if CCVPE_X4_CLOCK
    if (tid == 0 && blockIdx.x < 1024 && blockIdx.z == 0) {
        g_x4_stamps[4 * blockIdx.x + 0] = __builtin_amdgcn_s_memtime() - clk0;
        g_x4_stamps[4 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - rt0;
        g_x4_stamps[4 * blockIdx.x + 2] = rt0;
    }
#endif
#undef CCVPE_X4_DECODE
#undef CCVPE_X4_ROFF
#undef CCVPE_X4_LOAD_RAW
#undef CCVPE_X4_STORE_RAW
#undef CCVPE_X4_LOAD_B
}

struct X4Config { int nw, s0, s1; };
static constexpr X4Config X4_CONFIGS[] = {{4, 2, 0}, {4, 3, 0}, {4, 4, 0}, {8, 3, 2}, {8, 3, 3}, {8, 4, 4}};
static constexpr int X4_NCFG = (int)(sizeof(X4_CONFIGS) / sizeof(X4_CONFIGS[0]));

// configuration index for a layer of N output channels (-1: wider than 128)
int conv_wino4x_config(int N) {
    for (int i = 0; i < X4_NCFG; ++i)
        if (N <= 16 * (X4_CONFIGS[i].s0 + X4_CONFIGS[i].s1)) return i;
    return -1;
}

template <int NW, int S0, int S1>
static void launch_wino4x_cfg(const ConvParams& p_in, hipStream_t s) {
    ConvParams p = p_in;
    if (p.splitk > 1) {
        const int ngr = (p.Cin + X4_GCH - 1) / X4_GCH;
        const int per = (ngr + p.splitk - 1) / p.splitk;
        p.splitk = (ngr + per - 1) / per;
    }
    constexpr int mx = (NW / 4) * 36 * 4 * 72;
    constexpr size_t lds = ((X4_VFLOATS > mx ? X4_VFLOATS : mx) + X4_GCH * X4_PLANE) * sizeof(float);
    static_assert((NW == 4 ? 2 : 1) * lds <= 160 * 1024, "workgroups per CU");
    static LdsAttr attr;
    auto kern = conv_wino4x_kernel<NW, S0, S1>;
    ensure_dynamic_lds(attr, reinterpret_cast<const void*>(kern), lds);
    const int mblocks = p.B * (p.W >> 4) * (p.H >> 4);
    static const int grid_override = getenv("CCVPE_X4_GRID") ? std::atoi(getenv("CCVPE_X4_GRID")) : 0;   // dev: workgroups of the persistent grid
    const int resident = grid_override > 0 ? grid_override : (NW == 4 ? 2 : 1) * 256 / (p.splitk > 1 ? p.splitk : 1);
    dim3 grid(std::min(mblocks, std::max(resident, 8)), 1, p.splitk > 1 ? p.splitk : 1);
    CCVPE_LAUNCH(kern, grid, dim3(NW * 64), lds, s, p);
#if CCVPE_X4_CLOCK
    {   // in-kernel clock = shader cycles / (100 MHz reference ticks) x 100 MHz, median over the workgroups (MI355X_MICROARCH.md, DVFS item 6)
        static int calls = 0;
        if (++calls % 16 == 0) {
            (void)hipStreamSynchronize(s);
            std::vector<unsigned long long> h(4 * 1024);
            (void)hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_x4_stamps), h.size() * 8);
            std::vector<double> mhz;
            for (unsigned i = 0; i < grid.x && i < 1024; ++i) if (h[4 * i + 1]) mhz.push_back(100.0 * (double)h[4 * i] / (double)h[4 * i + 1]);
            std::sort(mhz.begin(), mhz.end());
            unsigned long long s0 = ~0ull, s1 = 0, e1 = 0;
            for (unsigned i = 0; i < grid.x && i < 1024; ++i) { s0 = std::min(s0, h[4 * i + 2]); s1 = std::max(s1, h[4 * i + 2]); e1 = std::max(e1, h[4 * i + 2] + h[4 * i + 1]); }
            std::fprintf(stderr, "  start skew %.1f us, first start -> last end %.1f us, median workgroup %.1f us\n", (s1 - s0) * 0.01, (e1 - s0) * 0.01, h[4 * (grid.x / 2) + 1] * 0.01);
            if (!mhz.empty()) std::fprintf(stderr, "conv_wino4x<%d,%d,%d>: in-kernel clock median %.0f MHz (min %.0f, max %.0f), %zu workgroups, %llu cycles each\n", NW, S0, S1,
                                           mhz[mhz.size() / 2], mhz.front(), mhz.back(), mhz.size(), h[0]);
        }
    }
#endif
    if (p.splitk > 1) launch_splitk_reduce(p, s);
}

void launch_wino4x(const ConvParams& p, hipStream_t s) {
    switch (p.wino4x_cfg) {
        case 0: launch_wino4x_cfg<4, 2, 0>(p, s); break;
        case 1: launch_wino4x_cfg<4, 3, 0>(p, s); break;
        case 2: launch_wino4x_cfg<4, 4, 0>(p, s); break;
        case 3: launch_wino4x_cfg<8, 3, 2>(p, s); break;
        case 4: launch_wino4x_cfg<8, 3, 3>(p, s); break;
        case 5: launch_wino4x_cfg<8, 4, 4>(p, s); break;
        default: break;
    }
}

bool conv_wino4x_supported(const ConvParams& p) {
    return p.wino4x_w != nullptr && p.wino4x_cfg >= 0 && p.wino4x_cfg < X4_NCFG && conv_wino4x_config(p.N) == p.wino4x_cfg &&
           p.KH == 3 && p.KW == 3 && p.stride == 1 && p.pad_t == 1 && p.pad_l == 1 && p.mode == MODE_CONV &&
           p.gate == nullptr && p.resid == nullptr && p.ndst == 1 && !p.dst[0].split && !p.in_split && p.OH == p.H && p.OW == p.W &&
           p.W % 16 == 0 && p.H % 16 == 0 && p.Cin % 8 == 0;
}

// Host-side weight transform (as conv_wino4_pack) into [k-step][wave][load][lane = (cin % 4) * 16 + cout % 16][4]
size_t conv_wino4x_pack(int N, int cin, const std::function<float(int, int, int)>& get, std::vector<float>& out, int* cfg_out) {
    static const double G[6][3] = {{0.25, 0.0, 0.0}, {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                                   {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0.0, 0.0, 1.0}};
    const int cfg = conv_wino4x_config(N);
    *cfg_out = cfg;
    out.clear();
    if (cfg < 0) return 0;
    const X4Config c = X4_CONFIGS[cfg];
    const int ksteps = ((cin + 15) / 16) * 4;
    const int step_loads = x4_step_loads(c.nw, c.s0, c.s1);
    out.assign((size_t)ksteps * step_loads * 256, 0.f);
    for (int n = 0; n < N; ++n)
        for (int ci = 0; ci < cin; ++ci) {
            double g[3][3], tmp[6][3];
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) g[i][j] = (double)get(n, i * 3 + j, ci);
            for (int r = 0; r < 6; ++r)
                for (int j = 0; j < 3; ++j) tmp[r][j] = G[r][0] * g[0][j] + G[r][1] * g[1][j] + G[r][2] * g[2][j];
            const int ks = ci / 4, k = ci % 4;
            const int slice = n / 16;
            const bool hi = slice >= c.s0;
            const int S = hi ? c.s1 : c.s0, sl = hi ? slice - c.s0 : slice;
            for (int r = 0; r < 6; ++r)
                for (int qq = 0; qq < 6; ++qq) {
                    const double uv = tmp[r][0] * G[qq][0] + tmp[r][1] * G[qq][1] + tmp[r][2] * G[qq][2];
                    const int xi = r * 6 + qq;
                    const int wq = xi / 9, xl = xi % 9;
                    const int wave = (hi ? 4 : 0) + wq;
                    const int wave_load0 = wave < 4 ? wave * x4_loads(c.s0) : 4 * x4_loads(c.s0) + (wave - 4) * x4_loads(c.s1);
                    const int u = xl * S + sl;
                    const size_t idx = (((size_t)ks * step_loads + wave_load0 + u / 4) * 64 + k * 16 + (n % 16)) * 4 + (u & 3);
                    out[idx] = (float)uv;
                }
        }
    return out.size();
}

}  // namespace ccvpe
