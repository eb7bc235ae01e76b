// Winograd F(4x4, 3x3) convolution on the fp32 matrix cores of gfx950, fully fused (no transformed tensors in HBM).
//
// Serves the WIDE decoder double_conv layers of the reference (models.py:42-47: conv6 / conv5 / conv4 and their _ori
// twins, conv3_ori; models.py:407-446) where the matrix pipe is the bound:  Y = A^T [ sum_c (G g G^T) (.) (B^T d B) ] A
// on 4x4 output tiles needs 36 multiplies per tile and channel pair instead of 144 - 4x fewer matrix-core cycles than
// the direct form, 1.78x fewer than the F(2x2,3x3) kernel (kernels_wino.hip).  Everything stays fp32; the weights are
// transformed once on the host in double precision.  The larger transform constants (up to 8) cost accuracy:
// ~1.4e-5 of the output scale per layer against fp64 (F(2x2): 7e-7, implicit GEMM: 4e-7) - tests/test_ops_gpu.py
// holds every tile to 1e-4, the path's contract is 1e-3.
//
// Work decomposition (NW = 4; the NW = 8 form doubles the channels per workgroup and halves the transform work per
// output channel: 8 waves, one 512-thread workgroup per CU, each lane transforms half an item): a workgroup of 4 wave64
// owns a 16 x 16 pixel block of one image (4 x 4 tiles = the 16 rows of
// one v_mfma_f32_16x16x4_f32) x 64 output channels (16 per wave), for all 36 Winograd positions xi: 144 accumulator
// registers per lane, so the inverse transform is a per-lane affair.  K is walked in groups of 16 input channels:
//   * the raw 18 x 18 x 16 patch is staged channel-major in LDS (plane stride 385, row pitch 20: the 64 lanes of a
//     transform wave hit 64 different banks);
//   * wave w transforms channels 4w..4w+3 of the group: one (channel, tile) item per lane, 18 ds_read2_b32, B^T d B in
//     registers, 18 ds_write_b64 into the V image [k-step][xi pair][A-fragment lane][2];
//   * every wave then runs, per k-step (4 channels), 36 MFMAs from 18 ds_read_b64 (two xi per read); the weights never
//     touch LDS: the host layout [k-step][xi/4][16-channel slice][lane][xi%4] is the B-fragment layout of four xi, so
//     a lane loads its 16 bytes straight into the MFMA operand registers, refilled for the next k-step as soon as a
//     quad's MFMAs have issued.
// Two barriers per group (V image single-buffered, 61.5 KB of LDS -> two workgroups per CU cover each other's
// transform phase); the next group's raw patch is in flight under the MFMA phase.  Persistent XCD-aware grid and
// split-K exactly as kernels_wino.hip.
#include "igemm_common.h"

#include <algorithm>
#include <thread>
#include <type_traits>

namespace ccvpe {

typedef float f32x2 __attribute__((ext_vector_type(2)));

static constexpr int W4_PITCH = 20;                 // floats per raw row (18 used)
static constexpr int W4_PLANE = 385;                // floats per raw channel plane (18 * 20 = 360, padded to 1 mod 64)
static constexpr int W4_GCH = 16;                   // input channels per group (4 k-steps)
static constexpr int W4_VFLOATS = 4 * 18 * 64 * 2;  // V image: [k-step][xi pair][lane][2]
static constexpr int W4_RAW_F4 = 18 * 18 * 4;       // float4 items of one raw group

// 1-D input transform B^T (Lavin & Gray, points 0, +-1, +-2, inf)
__device__ __forceinline__ void w4_bt(const float d0, const float d1, const float d2, const float d3, const float d4, const float d5,
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
// 1-D output transform A^T
__device__ __forceinline__ void w4_at(const float m0, const float m1, const float m2, const float m3, const float m4, const float m5,
                                      float& y0, float& y1, float& y2, float& y3) {
    const float s1 = m1 + m2, d1 = m1 - m2, s2 = m3 + m4, d2 = m3 - m4;
    y0 = m0 + s1 + s2;
    y1 = fmaf(2.f, d2, d1);
    y2 = fmaf(4.f, s2, s1);
    y3 = fmaf(8.f, d2, d1) + m5;
}

// FUSED: a split-K launch that reduces itself (igemm_common.h) - slab stores write-through, a ticket per item, the sums behind the loop.
// A template parameter, not a run-time flag: the kernel sits at exactly 256 registers and any extra path in its loop spills.
// PANEL (split-K launches with at least eight (channel block, K slice) pairs; one-dimensional grid): the work units are dealt
// PANEL-major - XCD x owns the weight panels x, x + 8, ... and walks ALL their pixel blocks - instead of one K slice per blockIdx.z with
// the items of a slice cut into eight runs.  A panel (conv6.0 at batch 32: 448 channels x 128 output channels x 36 positions = 8.3 MB) is
// then streamed into ONE L2 once; cut into runs, every XCD met three of the fifteen panels per slice and the launch fetched 786 MB for
// 189 MB of algorithmic traffic (profiles/r03_traffic.json), the weights 4-5 times.
template <int NW, bool FUSED = false, bool PANEL = false>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_wino4_kernel(const ConvParams p) {
    static_assert(NW == 4 || NW == 8, "waves w and w + 4 share the k-step w & 3 of a 16-channel group");
    constexpr int NT = NW * 64;
    constexpr int RAW_ITEMS = NW == 4 ? 6 : 3;             // float4 items per thread of one raw group (1296 in all)
    constexpr int RAW_HALVES = RAW_ITEMS / 3;              // staged three items at a time
    constexpr unsigned OOB = 0x80000000u;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Vs = smem;                 // [4][18][64][2]
    float* Rs = smem + W4_VFLOATS;    // [16][W4_PLANE]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;

    const int mbx = p.W >> 4, mby = p.H >> 4;
    const int mblocks = p.B * mbx * mby;
    const int total = mblocks * (p.wino_nb ? p.wino_nb : (p.wino_n16 + NW - 1) / NW);   // wino_nb: a sub-launch over some n-blocks

    // persistent work loop (see kernels_wino.hip): XCD x owns a contiguous run of items, channel block slowest
    const int xcd = blockIdx.x & 7;
    const int stride = ((int)gridDim.x >> 3) + (xcd < ((int)gridDim.x & 7) ? 1 : 0);
    const int S = p.splitk > 1 ? p.splitk : 1;
    // PANEL: unit u of this XCD = (its k-th panel, pixel block): panel x + 8 k = (channel block, K slice)
    const int npanels = (total / mblocks) * S;
    const int item_begin = PANEL ? 0 : xcd * (total >> 3) + min(xcd, total & 7);
    const int xpanels = max((npanels - xcd + 7) >> 3, 1);   // panels of this XCD
    const int item_end = PANEL ? xpanels * mblocks : item_begin + (total >> 3) + (xcd < (total & 7) ? 1 : 0);
    int item = item_begin + ((int)blockIdx.x >> 3);
    if (item >= item_end) return;
    const int item_first = item;
    int ordinal = 0;                     // items this workgroup has finished
    unsigned long long fin_mask = 0;     // self-reducing split-K: ordinals of the regions it is the last K slice of

    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, p.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wino4_w), 0, p.wino4_bytes, 0x00020000);

    int nb, b, by, bx, z = PANEL ? 0 : (int)blockIdx.z, z_n = z;
#define CCVPE_W4_DECODE(it_, nb_, b_, by_, bx_, z_)                                                      \
    {                                                                                                    \
        int q_, mb_;                                                                                     \
        if (PANEL) {   /* panel-minor inside the XCD: its workgroups stream all its panels side by side (all 32 on ONE panel met in the same L2 lines at the same time) */ \
            mb_ = (it_) / xpanels; q_ = xcd + 8 * ((it_) - mb_ * xpanels);                               \
            nb_ = q_ / S; z_ = q_ - nb_ * S;                                                             \
        } else { q_ = (it_) / mblocks; mb_ = (it_) - q_ * mblocks; nb_ = q_; }                           \
        b_ = mb_ / (mbx * mby);                                                                          \
        const int rem_ = mb_ - b_ * (mbx * mby);                                                         \
        by_ = rem_ / mbx;                                                                                \
        bx_ = rem_ - by_ * mbx;                                                                          \
    }
    CCVPE_W4_DECODE(item, nb, b, by, bx, z);

    // ---- raw patch staging: float4 j = tid + i*NT -> pixel j / 4 of the 18 x 18 region, channels 4 * (j % 4) of the group
    // (register budget: 144 accumulators + 36 weight registers leave ~70 for everything else at two waves per SIMD, so the
    //  LDS offset of an item is recomputed when it is stored and the patch travels in two halves of three float4)
    unsigned r_off[RAW_ITEMS];
    const int r_ch = (tid & 3) * 4;   // first channel inside the group (NT % 4 == 0: the same for every item of a thread)
    const int r_px0 = tid >> 2;       // pixel of item 0; item i is pixel r_px0 + (NT / 4) i
#define CCVPE_W4_ROFF(b_, by_, bx_, live_)                                                               \
    _Pragma("unroll") for (int i = 0; i < RAW_ITEMS; ++i) {                                              \
        int j = tid + i * NT;                                                                            \
        asm volatile("" : "+v"(j));   /* recomputed per item: hoisted, the six (row, column) pairs of a thread lived in scratch (12 spills - and a kernel that touches scratch starts ~5 us later: tools/ubench_scratch.hip) */ \
        const int px = j >> 2, q = j & 3;                                                                \
        const int py = px / 18, pxx = px - py * 18;                                                      \
        const int y = (by_) * 16 - 1 + py, x = (bx_) * 16 - 1 + pxx;                                     \
        const bool ok = (live_) && j < W4_RAW_F4 && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W; \
        r_off[i] = ok ? (unsigned)(((((b_) * p.H + y) * p.W + x) * p.in_ld + q * 4) * 4) : OOB;          \
    }
    CCVPE_W4_ROFF(b, by, bx, true);

    // ---- transform item of this lane: channel 4*wave + (lane & 3) of the group, tile (ty, tx) = (lane >> 4, (lane >> 2) & 3)
    const int tk = lane & 3, ttx = (lane >> 2) & 3, tty = lane >> 4;
    const int tks = wave & 3;         // k-step of the group this wave transforms (NW 8: waves w and w + 4 split its output rows)
    const float* t_src = Rs + (4 * tks + tk) * W4_PLANE + (4 * tty) * W4_PITCH + 4 * ttx;
    // V position of the item = A-fragment lane (k * 16 + tile); xi pair xp lives 128 floats further per step
    float* t_dst = Vs + ((tks * 18) * 64 + tk * 16 + tty * 4 + ttx) * 2;

    // weights: [k-step][xi/4][n16 slice][lane][xi%4], one 16-byte load per (k-step, quad)
    const unsigned w_quad_b = (unsigned)p.wino_n16 * 1024u;      // bytes between consecutive quads
    const unsigned w_step_b = w_quad_b * 9u;                     // bytes per k-step
#define CCVPE_W4_WBASE(nb_) ((nb_) * NW + wave + p.wino_n16_off < p.wino_n16 ? (unsigned)((nb_) * NW + wave + p.wino_n16_off) * 1024u + (unsigned)lane * 16u : OOB)
    unsigned w_base = CCVPE_W4_WBASE(nb);

    // split-K over channel groups (blockIdx.z)
    const int ngr_all = (p.Cin + W4_GCH - 1) / W4_GCH;
    const int g_per = (ngr_all + S - 1) / S;          // channel groups per K slice (the host never launches an empty slice)
    int g_begin = min(z * g_per, ngr_all), g_end = min(g_begin + g_per, ngr_all);
    if (g_begin >= g_end) return;
    // the layer's last group may hold only 8 real channels (Cin % 16 == 8: 40, 56, 104, 200 ...): its k-steps 2 and 3 would
    // multiply zeros, so they are neither transformed nor run (conv2.2: 10 k-steps instead of 12)
    const int tail_ks = (p.Cin - (ngr_all - 1) * W4_GCH + 3) >> 2;

    f32x4 raw[3];
    float bq[36];             // B fragments of the current k-step, refilled in place for the next one
    // channels past Cin (last group of a layer whose Cin is not a multiple of 16) are forced to zero
#define CCVPE_W4_LOAD_RAW(c0, half)   /* items 3*half .. 3*half+2 */                                     \
    _Pragma("unroll") for (int i = 0; i < 3; ++i) {                                                    \
        const unsigned o_ = ((c0) + r_ch < p.Cin) ? r_off[3 * (half) + i] : OOB;                         \
        raw[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, o_, (c0) * 4, 0)); \
    }
#define CCVPE_W4_STORE_RAW(half)                                                                         \
    _Pragma("unroll") for (int i = 0; i < 3; ++i) {                                                    \
        const int px_ = r_px0 + (NT / 4) * (3 * (half) + i);                                             \
        if (px_ < 18 * 18) {                                                                             \
            const int py_ = (px_ * 3641) >> 16;              /* px / 18, exact for px < 324 */            \
            float* d_ = Rs + r_ch * W4_PLANE + py_ * W4_PITCH + (px_ - py_ * 18);                        \
            d_[0] = raw[i].x; d_[W4_PLANE] = raw[i].y; d_[2 * W4_PLANE] = raw[i].z; d_[3 * W4_PLANE] = raw[i].w; \
        }                                                                                                \
    }
#define CCVPE_W4_LOAD_B(wb, ks, qd)   /* k-step ks (global index), quad qd = xi 4qd .. 4qd+3 */           \
    {                                                                                                    \
        const f32x4 t4_ = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, wb, (ks) * w_step_b + (qd) * w_quad_b, 0)); \
        bq[4 * (qd)] = t4_.x; bq[4 * (qd) + 1] = t4_.y; bq[4 * (qd) + 2] = t4_.z; bq[4 * (qd) + 3] = t4_.w; \
    }

    f32x4 acc[36];
#pragma unroll
    for (int x = 0; x < 36; ++x) acc[x] = f32x4{0.f, 0.f, 0.f, 0.f};

    CCVPE_W4_LOAD_RAW(g_begin * W4_GCH, 0);
#pragma unroll
    for (int qd = 0; qd < 9; ++qd) { CCVPE_W4_LOAD_B(w_base, g_begin * 4, qd); }
    CCVPE_W4_STORE_RAW(0);
    if (RAW_HALVES == 2) {
        CCVPE_W4_LOAD_RAW(g_begin * W4_GCH, 1);
        CCVPE_W4_STORE_RAW(1);
    }
    __syncthreads();

    const float* va0 = Vs + lane * 2;
    const bool split = p.splitk > 1;
    const int ld = split ? p.N : p.dst[0].ld;
    const int act = split ? ACT_NONE : p.act;

    while (true) {
        const int item_n = item + stride;
        const bool have_n = item_n < item_end;
        int nb_n, b_n, by_n, bx_n;
        CCVPE_W4_DECODE(have_n ? item_n : item, nb_n, b_n, by_n, bx_n, z_n);
        const unsigned w_base_n = have_n ? CCVPE_W4_WBASE(nb_n) : OOB;
        const int g_begin_n = PANEL ? min(z_n * g_per, ngr_all) : g_begin;       // (PANEL: the next unit may belong to another K slice)

        for (int g = g_begin; g < g_end; ++g) {
            const bool last_group = g == g_end - 1;
            const int nks = g == ngr_all - 1 ? tail_ks : 4;
            // ---- transform: B^T d B of this lane's (channel, tile), 36 values -> V image ----
            // Two passes over the 6 x 6 patch, output rows 0-2 then 3-5 (18 live intermediates instead of 36); within a pass
            // the column transform is software pipelined: column c + 1 is being read while column c is transformed.
            // one half-item: output rows 3 HF .. 3 HF + 2.  HF is a compile-time constant inside, so only the three needed rows
            // of every column transform are computed (6 instead of 12 operations per column)
            auto half_item = [&](auto hfc) {
                constexpr int HF = decltype(hfc)::value;
                float t[3][6];
                float dc[2][6];
#pragma unroll
                for (int r = 0; r < 6; ++r) dc[0][r] = t_src[r * W4_PITCH];
#pragma unroll
                for (int c = 0; c < 6; ++c) {
                    if (c + 1 < 6) {
#pragma unroll
                        for (int r = 0; r < 6; ++r) dc[(c + 1) & 1][r] = t_src[r * W4_PITCH + c + 1];
                    }
                    float u0, u1, u2, u3, u4, u5;
                    w4_bt(dc[c & 1][0], dc[c & 1][1], dc[c & 1][2], dc[c & 1][3], dc[c & 1][4], dc[c & 1][5], u0, u1, u2, u3, u4, u5);
                    t[0][c] = HF ? u3 : u0; t[1][c] = HF ? u4 : u1; t[2][c] = HF ? u5 : u2;
                }
#pragma unroll
                for (int ii = 0; ii < 3; ++ii) {
                    float v0, v1, v2, v3, v4, v5;
                    w4_bt(t[ii][0], t[ii][1], t[ii][2], t[ii][3], t[ii][4], t[ii][5], v0, v1, v2, v3, v4, v5);
                    // row i = 3 HF + ii; xi = 6 i + j; pairs (6i, 6i+1), (6i+2, 6i+3), (6i+4, 6i+5) = xp 3i .. 3i+2
                    float* td = t_dst + HF * (9 * 128) + ii * (3 * 128);
                    *reinterpret_cast<f32x2*>(td) = f32x2{v0, v1};
                    *reinterpret_cast<f32x2*>(td + 128) = f32x2{v2, v3};
                    *reinterpret_cast<f32x2*>(td + 256) = f32x2{v4, v5};
                }
            };
            if (tks >= nks) {           // k-step beyond the layer's channels: nothing to transform (wave-uniform)
            } else if (NW == 4) {       // both halves, one after the other (18 live intermediates instead of 36)
                half_item(std::integral_constant<int, 0>{});
                half_item(std::integral_constant<int, 1>{});
            } else if (wave < 4) {      // NW 8: waves w and w + 4 share the item (wave-uniform branch)
                half_item(std::integral_constant<int, 0>{});
            } else {
                half_item(std::integral_constant<int, 1>{});
            }
            __syncthreads();   // V image complete; every wave is done with the raw patch
            // ---- the patch after this one: next group of this tile, or the first group of the next tile ----
            const int c0n = last_group ? g_begin_n * W4_GCH : (g + 1) * W4_GCH;
            if (last_group) { CCVPE_W4_ROFF(b_n, by_n, bx_n, have_n); }
            CCVPE_W4_LOAD_RAW(c0n, 0);
            __builtin_amdgcn_sched_barrier(0);
            // ---- MFMA phase: 4 k-steps x 36 xi (a real loop: unrolled, hipcc renames the 144 accumulator and 36 weight
            //      registers per k-step and spills a hundred of them) ----
#pragma unroll 1
            for (int ks = 0; ks < nks; ++ks) {
                const bool last_step = last_group && ks == nks - 1;
                const unsigned wb = last_step ? w_base_n : w_base;
                const int ksn = last_step ? g_begin_n * 4 : g * 4 + ks + 1;
                const float* va = va0 + ks * (18 * 128);
                f32x2 fa[3];              // V pairs, read two pairs (4 MFMAs = 128 cycles) ahead of their use
                fa[0] = *reinterpret_cast<const f32x2*>(va);
                fa[1] = *reinterpret_cast<const f32x2*>(va + 128);
#pragma unroll
                for (int xp = 0; xp < 18; ++xp) {
                    if (xp + 2 < 18) fa[(xp + 2) % 3] = *reinterpret_cast<const f32x2*>(va + (xp + 2) * 128);
                    __builtin_amdgcn_sched_barrier(0);   // keep the next pair's read above this pair's MFMAs
                    // weights as the A operand, V as the B operand: D[channel][tile], so a lane ends up with 4 consecutive
                    // channels of ONE tile and the epilogue stores 16-byte pieces (the fragment layouts of A and B coincide)
                    acc[2 * xp] = __builtin_amdgcn_mfma_f32_16x16x4f32(bq[2 * xp], fa[xp % 3].x, acc[2 * xp], 0, 0, 0);
                    acc[2 * xp + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(bq[2 * xp + 1], fa[xp % 3].y, acc[2 * xp + 1], 0, 0, 0);
                    if (xp & 1) {   // a quad's MFMAs have issued: refill its weight registers for the next k-step
                        __builtin_amdgcn_sched_barrier(0);
                        CCVPE_W4_LOAD_B(wb, ksn, xp >> 1);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                if (RAW_HALVES == 2 && ks == 1) {   // first half of the next patch -> LDS (two k-steps of cover), second half into the same registers
                    CCVPE_W4_STORE_RAW(0);
                    CCVPE_W4_LOAD_RAW(c0n, 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            CCVPE_W4_STORE_RAW(RAW_HALVES - 1);
            __syncthreads();   // V image free again; next raw patch complete
        }

        // ---- inverse transform A^T M A, bias, activation, store ----
        // lane = (channel quad cq = lane >> 4, tile (ty, tx) = ((lane >> 2) & 3, lane & 3)): accumulator element i of xi is
        // channel 4 cq + i of that tile, so the 4 x 4 output pixels of the tile leave as 16 float4 stores per lane
        {
            const int n = (nb * NW + wave) * 16 + 4 * (lane >> 4);
            const bool nok = n < p.N;                                   // N % 4 == 0: the whole quad is in or out
            const f32x4 bias = (split || !nok) ? f32x4{0.f, 0.f, 0.f, 0.f} : *reinterpret_cast<const f32x4*>(p.bias + n);
            const float lo = act == ACT_RELU ? 0.f : -__builtin_inff();   // ReLU as a select: no branch per store
            const size_t pix0 = ((size_t)b * p.H + (size_t)by * 16) * p.W + (size_t)bx * 16;
            float* obase = split ? p.partial + ((size_t)z * p.M + pix0) * p.N : p.dst[0].ptr + pix0 * ld + p.dst[0].coff;
            const __amdgpu_buffer_rsrc_t o_rsrc = __builtin_amdgcn_make_buffer_rsrc(obase, 0, 0x7fffffff, 0x00020000);
            const int oty = (lane >> 2) & 3, otx = lane & 3;
            const unsigned o_lane = nok ? (unsigned)((((oty * 4) * p.W + otx * 4) * ld + n) * 4) : OOB;
            // the 16 outputs of channel i go back into element i of acc[0..15] (dead by then): no second register set, and
            // acc[4 a + c] ends up as the float4 of pixel (a, c)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float tt[4][6];
#pragma unroll
                for (int q = 0; q < 6; ++q)
                    w4_at(acc[0 * 6 + q][i], acc[1 * 6 + q][i], acc[2 * 6 + q][i], acc[3 * 6 + q][i], acc[4 * 6 + q][i], acc[5 * 6 + q][i],
                          tt[0][q], tt[1][q], tt[2][q], tt[3][q]);
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    float yy[4];
                    w4_at(tt[a][0], tt[a][1], tt[a][2], tt[a][3], tt[a][4], tt[a][5], yy[0], yy[1], yy[2], yy[3]);
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[4 * a + c][i] = yy[c];
                }
            }
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    f32x4 v;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float t = acc[4 * a + c][i] + bias[i];
                        v[i] = t < lo ? lo : t;
                    }
                    const int soff = ((a * p.W + c) * ld) * 4;   // uniform
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, v), o_rsrc, o_lane, soff, FUSED ? 16 : 0);   // (FUSED: slab, write-through - ticket.h)
                    // gfx950: a vector instruction that overwrites the data registers of a 16-byte buffer store in the very next
                    // issue slot can reach the registers before the store has read them (seen with v_pk_add_f32 behind a store
                    // with an SGPR soffset, which hipcc's hazard recogniser exempts): lanes 12-15 of every row stored the NEXT
                    // pixel's value.  Two wait states after each store; tests/test_isa_hazard.py checks the generated code.
                    __builtin_amdgcn_sched_barrier(0);
                    asm volatile("s_nop 1");
                    __builtin_amdgcn_sched_barrier(0);
                }
        }
        if (FUSED) {
            // self-reducing split-K (igemm_common.h): the last K slice of this (pixel block, channel block) sums the slabs and stores the
            // block; the V image is free here (its first word is the ticket's flag), the raw patch of the next item is not touched
            // (the sum itself runs behind the loop: next to the loop's prefetch state its registers spill)
            if (splitk_ticket(p, nb * mblocks + (b * mby + by) * mbx + bx, reinterpret_cast<unsigned*>(Vs))) fin_mask |= 1ull << ordinal;
        }
        if (FUSED) ++ordinal;
        if (!have_n) break;
#pragma unroll
        for (int x = 0; x < 36; ++x) acc[x] = f32x4{0.f, 0.f, 0.f, 0.f};
        item = item_n; nb = nb_n; b = b_n; by = by_n; bx = bx_n; w_base = w_base_n;
        if (PANEL) { z = z_n; g_begin = g_begin_n; g_end = min(g_begin + g_per, ngr_all); }
    }
    // the regions whose last ticket this workgroup drew (the launcher keeps a workgroup's items <= 64 when the launch reduces itself)
    while (FUSED && fin_mask) {
        const int k = __builtin_ctzll(fin_mask);
        fin_mask &= fin_mask - 1;
        int nb_f, b_f, by_f, bx_f, z_f = 0;
        CCVPE_W4_DECODE(item_first + k * stride, nb_f, b_f, by_f, bx_f, z_f);
        (void)z_f;
        splitk_finish<NT>(p, (b_f * p.H + by_f * 16) * p.W + bx_f * 16, 16, 16, p.W, nb_f * NW * 16, NW * 16);
    }
#undef CCVPE_W4_DECODE
#undef CCVPE_W4_ROFF
#undef CCVPE_W4_WBASE
#undef CCVPE_W4_LOAD_RAW
#undef CCVPE_W4_STORE_RAW
#undef CCVPE_W4_LOAD_B
}

template <int NW>
static void launch_wino4_plain(const ConvParams& p_in, hipStream_t s) {
    ConvParams p = p_in;
    if (p.splitk > 1) {   // never launch an empty K slice (it would leave its slab unwritten): 84 groups over 16 slices = 14 x 6
        const int ngr = (p.Cin + W4_GCH - 1) / W4_GCH;
        const int per = (ngr + p.splitk - 1) / p.splitk;
        p.splitk = (ngr + per - 1) / per;
    }
    constexpr size_t lds = (W4_VFLOATS + W4_GCH * W4_PLANE) * sizeof(float);
    static_assert(2 * lds <= 160 * 1024, "two workgroups per CU");
    static LdsAttr attr, attr_f, attr_p, attr_pf;
    const int mblocks = p.B * (p.W >> 4) * (p.H >> 4);
    const int nblocks = p.wino_nb ? p.wino_nb : (p.wino_n16 + NW - 1) / NW;
    const int S = p.splitk > 1 ? p.splitk : 1;
    // Opt-in (CCVPE_WINO4_PANEL=1), measured in round 4 on conv6.0 at batch 32 (profiles/r04_traffic.json): fabric-side traffic 786 MB ->
    // 529 MB per launch (414 MB with the XCD's workgroups all on one panel at a time, but that form runs 4 % slower: 32 CUs meet in the
    // same L2 lines), the launch alone 1.7 % faster, the whole two-stream step 0.3 % slower - so the default stays one K slice per blockIdx.z.
    static const bool want_panel = getenv("CCVPE_WINO4_PANEL") && std::atoi(getenv("CCVPE_WINO4_PANEL")) == 1;
    const bool panel = S > 1 && nblocks * S >= 8 && want_panel;   // panel-major units, one-dimensional grid (see the kernel's comment)
    const int resident = (NW == 4 ? 2 : 1) * 256 / (panel ? 1 : S);   // 256-thread workgroups: two per CU, 512-thread: one
    dim3 grid(std::min(mblocks * nblocks * (panel ? S : 1), std::max(resident, 8)), 1, panel ? 1 : S);
    const int per_wg = (mblocks * nblocks * (panel ? S : 1) + (int)grid.x - 1) / (int)grid.x + 8;   // (upper bound of a workgroup's units)
    if (S <= 1 || p.tickets == nullptr || mblocks * nblocks > CONV_TICKETS || per_wg > 64) p.split_fused = 0;
    const dim3 block(NW * 64);
    if (panel && p.split_fused) { auto kern = conv_wino4_kernel<NW, true, true>; ensure_dynamic_lds(attr_pf, reinterpret_cast<const void*>(kern), lds); CCVPE_LAUNCH(kern, grid, block, lds, s, p); return; }
    if (panel) { auto kern = conv_wino4_kernel<NW, false, true>; ensure_dynamic_lds(attr_p, reinterpret_cast<const void*>(kern), lds); CCVPE_LAUNCH(kern, grid, block, lds, s, p); launch_splitk_reduce(p, s); return; }
    if (p.split_fused) { auto kern = conv_wino4_kernel<NW, true, false>; ensure_dynamic_lds(attr_f, reinterpret_cast<const void*>(kern), lds); CCVPE_LAUNCH(kern, grid, block, lds, s, p); return; }
    auto kern = conv_wino4_kernel<NW, false, false>;
    ensure_dynamic_lds(attr, reinterpret_cast<const void*>(kern), lds);
    CCVPE_LAUNCH(kern, grid, block, lds, s, p);
    if (S > 1) launch_splitk_reduce(p, s);
}

// cfg split code 255 = "tail split": when the work items are a whole number of rounds of the resident workgroups plus a
// remainder that is a whole number of n-blocks (conv5.0: 128 pixel blocks x 5 channel blocks = 640 items on 512 workgroups),
// the full rounds run unsplit and only the remaining n-blocks are split over K - 1.3 rounds of work instead of 2, and only
// their slice of the output goes through slabs.
static thread_local bool g_tail_applied = false;
bool conv_wino4_tail_applied() { return g_tail_applied; }   // did the last F(4x4) launch of this thread with split code 255 split its tail?

template <int NW>
static void launch_wino4(const ConvParams& p, hipStream_t s) {
    g_tail_applied = false;
    if (p.splitk != 255) { launch_wino4_plain<NW>(p, s); return; }
    ConvParams a = p;
    a.splitk = 1;
    const int mblocks = p.B * (p.W >> 4) * (p.H >> 4);
    const int nblocks = (p.wino_n16 + NW - 1) / NW;
    const int resident = (NW == 4 ? 2 : 1) * 256;
    const int rem = (mblocks * nblocks) % resident;
    const int ngr = (p.Cin + W4_GCH - 1) / W4_GCH;
    const int tail_nb = mblocks > 0 ? rem / mblocks : 0;
    int split = tail_nb > 0 ? std::min(resident / rem, ngr / 2) : 0;
    const int n0 = (nblocks - tail_nb) * NW * 16;                    // first channel of the tail
    if (rem == 0 || rem % mblocks != 0 || tail_nb >= nblocks || split < 2 || p.partial == nullptr ||
        (size_t)split * p.M * (size_t)(p.N - n0) > p.partial_floats || (n0 & 3) != 0) {
        launch_wino4_plain<NW>(a, s);
        return;
    }
    g_tail_applied = true;
    a.wino_nb = nblocks - tail_nb; a.split_fused = 0;
    launch_wino4_plain<NW>(a, s);
    ConvParams b = p;
    b.wino_nb = tail_nb; b.wino_n16_off = (nblocks - tail_nb) * NW;
    b.N = p.N - n0; b.bias = p.bias + n0; b.dst[0].coff = p.dst[0].coff + n0;
    b.splitk = split;
    launch_wino4_plain<NW>(b, s);
}

void launch_wino4_64(const ConvParams& p, hipStream_t s) { launch_wino4<4>(p, s); }
void launch_wino4_128(const ConvParams& p, hipStream_t s) { launch_wino4<8>(p, s); }

bool conv_wino4_supported(const ConvParams& p) {
    return p.wino4_w != nullptr && p.KH == 3 && p.KW == 3 && p.stride == 1 && p.pad_t == 1 && p.pad_l == 1 && p.mode == MODE_CONV &&
           p.gate == nullptr && p.resid == nullptr && p.ndst == 1 && !p.dst[0].split && !p.in_split && p.OH == p.H && p.OW == p.W &&
           p.W % 16 == 0 && p.H % 16 == 0 && p.Cin % 8 == 0 && p.N % 4 == 0 && p.dst[0].ld % 4 == 0 && p.dst[0].coff % 4 == 0;
}

// Host-side weight transform: U = G g G^T (6 x 6) per (cout, cin) in double precision, stored as the B-fragment layout
//   [k-step = cin/4][xi/4][n16][lane = (cin%4)*16 + cout%16][xi%4]       (1024 B per (k-step, quad, n16 slice))
// with K padded to a multiple of 16 channels (zero weights).  `get(n, tap, c)` returns the 3x3 weight (tap = ky*3 + kx).
size_t conv_wino4_pack(int N, int cin, const std::function<float(int, int, int)>& get, std::vector<float>& out) {
    static const double G[6][3] = {{0.25, 0.0, 0.0}, {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                                   {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0.0, 0.0, 1.0}};
    const int n16 = (N + 15) / 16;
    const int ksteps = ((cin + 15) / 16) * 4;
    out.assign((size_t)ksteps * 9 * n16 * 256, 0.f);
    auto work = [&](int n_lo, int n_hi) {
        for (int n = n_lo; n < n_hi; ++n)
            for (int c = 0; c < cin; ++c) {
                double g[3][3], tmp[6][3];
                for (int i = 0; i < 3; ++i)
                    for (int j = 0; j < 3; ++j) g[i][j] = (double)get(n, i * 3 + j, c);
                for (int r = 0; r < 6; ++r)
                    for (int j = 0; j < 3; ++j) tmp[r][j] = G[r][0] * g[0][j] + G[r][1] * g[1][j] + G[r][2] * g[2][j];
                const int ks = c / 4, k = c % 4;
                for (int r = 0; r < 6; ++r)
                    for (int q = 0; q < 6; ++q) {
                        const double uv = tmp[r][0] * G[q][0] + tmp[r][1] * G[q][1] + tmp[r][2] * G[q][2];
                        const int xi = r * 6 + q;
                        const size_t idx = ((((size_t)ks * 9 + xi / 4) * n16 + n / 16) * 64 + k * 16 + (n % 16)) * 4 + (xi & 3);
                        out[idx] = (float)uv;
                    }
            }
    };
    const int nthreads = std::max(1, std::min(16, (int)std::thread::hardware_concurrency()));
    if (nthreads == 1 || (long long)N * cin < 4096) { work(0, N); return out.size(); }
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; ++t) th.emplace_back(work, (int)((long long)N * t / nthreads), (int)((long long)N * (t + 1) / nthreads));
    for (auto& x : th) x.join();
    return out.size();
}

}  // namespace ccvpe
