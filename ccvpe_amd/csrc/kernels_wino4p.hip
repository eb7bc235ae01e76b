// Winograd F(4x4, 3x3), split form: the input transform V = B^T d B is written ONCE per layer by a small HBM-bound
// kernel, in the exact order the matrix kernel's LDS image has; the matrix kernel then is a batched GEMM over the 36
// Winograd positions with no transform and no raw-patch staging in its K loop (V groups arrive by LDS-DMA,
// `buffer_load_dwordx4 ... lds`, double-buffered, one barrier per 16-channel group) and keeps the inverse transform
// A^T M A in its epilogue.  Same call sites as kernels_wino4.hip (decoder double_conv layers, reference models.py:42-47,
// 407-446); the fused form re-transforms the 18 x 18 x Cin patch once per 64- / 128-channel workgroup (5x on conv5.0 /
// conv6.0), which this form removes at the price of 2.25x the input bytes written and read back (through the
// Infinity Cache for the layers it is picked for).  The autotuner times both forms per layer.
//
// V layout in HBM: [16 x 16 pixel block][16-channel group][k-step 4][xi pair 18][lane 64 = (channel % 4) * 16 + tile][2]
// = 36 KiB per (block, group), the LDS image of kernels_wino4.hip verbatim.
#include "igemm_common.h"

#include <algorithm>

namespace ccvpe {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* LdsPtrW4;

#ifndef CCVPE_W4P_SWITCH
#define CCVPE_W4P_SWITCH 0   // dev builds (tools/build_variant.sh): bit 0 no weight refills, bit 1 no V reads, bit 2 no V DMA in the K loop - timing only, wrong results
#endif

static constexpr int W4P_PITCH = 20;                 // floats per raw row (18 used)
static constexpr int W4P_PLANE = 385;                // floats per raw channel plane
static constexpr int W4P_GCH = 16;                   // input channels per group (4 k-steps)
static constexpr int W4P_VFLOATS = 4 * 18 * 64 * 2;  // one V group: [k-step][xi pair][lane][2]

__device__ __forceinline__ void w4p_bt(const float d0, const float d1, const float d2, const float d3, const float d4, const float d5,
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
__device__ __forceinline__ void w4p_at(const float m0, const float m1, const float m2, const float m3, const float m4, const float m5,
                                       float& y0, float& y1, float& y2, float& y3) {
    const float s1 = m1 + m2, d1 = m1 - m2, s2 = m3 + m4, d2 = m3 - m4;
    y0 = m0 + s1 + s2;
    y1 = fmaf(2.f, d2, d1);
    y2 = fmaf(4.f, s2, s1);
    y3 = fmaf(8.f, d2, d1) + m5;
}

// ---------------------------------------------------------------------------------------------------------------
// Input transform: one 256-thread workgroup per (pixel block, channel group); HBM-bound (reads 1.27x the group's input
// bytes incl. the halo, writes 2.25x).  The arithmetic is the fused kernel's (same operation order -> same bits).
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void wino4_input_transform_kernel(const ConvParams p, float* __restrict__ vout, const int ngr) {
    __shared__ float Rs[W4P_GCH * W4P_PLANE];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int mbx = p.W >> 4, mby = p.H >> 4;
    const int g = blockIdx.x % ngr;
    const int mb = blockIdx.x / ngr;
    const int b = mb / (mbx * mby);
    const int rem = mb - b * (mbx * mby);
    const int by = rem / mbx, bx = rem - by * mbx;
    const int c0 = g * W4P_GCH;
    constexpr unsigned OOB = 0x80000000u;
    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, p.in_bytes, 0x00020000);
    const int r_ch = (tid & 3) * 4;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int j = tid + i * 256;
        const int px = j >> 2;
        if (px < 18 * 18) {
            const int py = px / 18, pxx = px - py * 18;
            const int y = by * 16 - 1 + py, x = bx * 16 - 1 + pxx;
            const bool ok = (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W && c0 + r_ch < p.Cin;
            const unsigned o = ok ? (unsigned)((((b * p.H + y) * p.W + x) * p.in_ld + r_ch) * 4) : OOB;
            const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, o, c0 * 4, 0));
            float* d = Rs + r_ch * W4P_PLANE + py * W4P_PITCH + pxx;
            d[0] = v.x; d[W4P_PLANE] = v.y; d[2 * W4P_PLANE] = v.z; d[3 * W4P_PLANE] = v.w;
        }
    }
    __syncthreads();
    // item of this lane: channel 4 * wave + (lane & 3) of the group, tile (ty, tx) = (lane >> 4, (lane >> 2) & 3)
    const int tk = lane & 3, ttx = (lane >> 2) & 3, tty = lane >> 4;
    const float* t_src = Rs + (4 * wave + tk) * W4P_PLANE + (4 * tty) * W4P_PITCH + 4 * ttx;
    float* t_dst = vout + ((size_t)mb * ngr + g) * W4P_VFLOATS + ((wave * 18) * 64 + tk * 16 + tty * 4 + ttx) * 2;
    if (c0 + 4 * wave >= p.Cin) {   // k-step beyond the layer's channels (Cin % 16 == 8): never read by the matrix kernel
        return;
    }
    float t[6][6];
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        float d[6];
#pragma unroll
        for (int r = 0; r < 6; ++r) d[r] = t_src[r * W4P_PITCH + c];
        w4p_bt(d[0], d[1], d[2], d[3], d[4], d[5], t[0][c], t[1][c], t[2][c], t[3][c], t[4][c], t[5][c]);
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        float v0, v1, v2, v3, v4, v5;
        w4p_bt(t[i][0], t[i][1], t[i][2], t[i][3], t[i][4], t[i][5], v0, v1, v2, v3, v4, v5);
        float* td = t_dst + i * (3 * 128);
        *reinterpret_cast<f32x2*>(td) = f32x2{v0, v1};
        *reinterpret_cast<f32x2*>(td + 128) = f32x2{v2, v3};
        *reinterpret_cast<f32x2*>(td + 256) = f32x2{v4, v5};
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Matrix kernel: workgroup = 16 x 16 pixel block x 16 NW output channels, wave = one 16-channel slice x all 36 xi
// (144 accumulators per lane; the inverse transform is per lane, as in kernels_wino4.hip).
// ---------------------------------------------------------------------------------------------------------------
template <int NW>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_wino4p_kernel(const ConvParams p, const float* __restrict__ vpre, const unsigned vpre_bytes) {
    static_assert(NW == 4 || NW == 8, "");
    constexpr unsigned OOB = 0x80000000u;
    constexpr int PIECES = W4P_VFLOATS * 4 / 1024;   // 36 LDS-DMA pieces of 1 KiB per group

    extern __shared__ __attribute__((aligned(16))) float smem[];   // [2][W4P_VFLOATS]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);   // LDS-DMA destinations and scalar offsets must be wave-uniform

    const int mbx = p.W >> 4, mby = p.H >> 4;
    const int mblocks = p.B * mbx * mby;
    const int total = mblocks * (p.wino_nb ? p.wino_nb : (p.wino_n16 + NW - 1) / NW);

    const int xcd = blockIdx.x & 7;
    const int stride = ((int)gridDim.x >> 3) + (xcd < ((int)gridDim.x & 7) ? 1 : 0);
    const int item_begin = xcd * (total >> 3) + min(xcd, total & 7);
    const int item_end = item_begin + (total >> 3) + (xcd < (total & 7) ? 1 : 0);
    int item = item_begin + ((int)blockIdx.x >> 3);
    if (item >= item_end) return;

    const __amdgpu_buffer_rsrc_t v_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(vpre), 0, vpre_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wino4_w), 0, p.wino4_bytes, 0x00020000);

    int nb = item / mblocks;
    int mb = item - nb * mblocks;

    const unsigned w_quad_b = (unsigned)p.wino_n16 * 1024u;
    const unsigned w_step_b = w_quad_b * 9u;
#define CCVPE_W4P_WBASE(nb_) ((nb_) * NW + wave + p.wino_n16_off < p.wino_n16 ? (unsigned)((nb_) * NW + wave + p.wino_n16_off) * 1024u + (unsigned)lane * 16u : OOB)
    unsigned w_base = CCVPE_W4P_WBASE(nb);

    const int ngr_all = (p.Cin + W4P_GCH - 1) / W4P_GCH;
    int g_begin = 0, g_end = ngr_all;
    if (p.splitk > 1) {
        const int per = (ngr_all + p.splitk - 1) / p.splitk;
        g_begin = min((int)blockIdx.z * per, ngr_all);
        g_end = min(g_begin + per, ngr_all);
    }
    if (g_begin >= g_end) return;
    const int tail_ks = (p.Cin - (ngr_all - 1) * W4P_GCH + 3) >> 2;

    // LDS-DMA of one V group: piece q (1 KiB, one wave instruction) -> buffer `buf`; wave w copies pieces w, w + NW, ...
    const unsigned v_lane = (unsigned)lane * 16u;
#define CCVPE_W4P_DMA(mb_, g_, buf_, live_)                                                              \
    {                                                                                                    \
        const unsigned gb_ = (unsigned)((mb_) * ngr_all + (g_)) * (unsigned)(W4P_VFLOATS * 4);           \
        _Pragma("unroll") for (int q_ = 0; q_ < (PIECES + NW - 1) / NW; ++q_) {                          \
            const int piece_ = wave_u + q_ * NW;                                                           \
            if (piece_ < PIECES) {                                                                       \
                LdsPtrW4 d_ = (LdsPtrW4)(smem + (buf_) * W4P_VFLOATS + piece_ * 256);                    \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(v_rsrc, d_, 16, (live_) ? v_lane : OOB, gb_ + piece_ * 1024, 0, 0); \
            }                                                                                            \
        }                                                                                                \
    }

    float bq[36];
#define CCVPE_W4P_LOAD_B(wb, ks, qd)                                                                     \
    {                                                                                                    \
        const f32x4 t4_ = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, wb, (ks) * w_step_b + (qd) * w_quad_b, 0)); \
        bq[4 * (qd)] = t4_.x; bq[4 * (qd) + 1] = t4_.y; bq[4 * (qd) + 2] = t4_.z; bq[4 * (qd) + 3] = t4_.w; \
    }

    f32x4 acc[36];
#pragma unroll
    for (int x = 0; x < 36; ++x) acc[x] = f32x4{0.f, 0.f, 0.f, 0.f};

    int vbuf = 0;
    CCVPE_W4P_DMA(mb, g_begin, 0, true);
#pragma unroll
    for (int qd = 0; qd < 9; ++qd) { CCVPE_W4P_LOAD_B(w_base, g_begin * 4, qd); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const bool split = p.splitk > 1;
    const int ld = split ? p.N : p.dst[0].ld;
    const int act = split ? ACT_NONE : p.act;

    while (true) {
        const int item_n = item + stride;
        const bool have_n = item_n < item_end;
        const int nb_n = (have_n ? item_n : item) / mblocks;
        const int mb_n = (have_n ? item_n : item) - nb_n * mblocks;
        const unsigned w_base_n = have_n ? CCVPE_W4P_WBASE(nb_n) : OOB;

        for (int g = g_begin; g < g_end; ++g) {
            const bool last_group = g == g_end - 1;
            const int nks = g == ngr_all - 1 ? tail_ks : 4;
            // the group after this one (of this item, or the first one of the next item) -> the other buffer; every wave is past
            // the barrier that ended that buffer's last reads
            if (!(CCVPE_W4P_SWITCH & 4)) {
                if (last_group) { CCVPE_W4P_DMA(mb_n, g_begin, vbuf ^ 1, have_n); }
                else { CCVPE_W4P_DMA(mb, g + 1, vbuf ^ 1, true); }
            }
            __builtin_amdgcn_sched_barrier(0);
            const float* va0 = smem + vbuf * W4P_VFLOATS + lane * 2;
#pragma unroll 1
            for (int ks = 0; ks < nks; ++ks) {
                const bool last_step = last_group && ks == nks - 1;
                const unsigned wb = last_step ? w_base_n : w_base;
                const int ksn = last_step ? g_begin * 4 : g * 4 + ks + 1;
                const float* va = va0 + ks * (18 * 128);
                // the last k-step of a group: everything this wave has in flight is at least one k-step old (the DMA pieces
                // three or more) - wait for it here, so the barrier below publishes complete V pieces
                if (ks == nks - 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                f32x2 fa[3];
                fa[0] = *reinterpret_cast<const f32x2*>(va);
                fa[1] = *reinterpret_cast<const f32x2*>(va + 128);
#pragma unroll
                for (int xp = 0; xp < 18; ++xp) {
                    if (xp + 2 < 18 && !(CCVPE_W4P_SWITCH & 2)) fa[(xp + 2) % 3] = *reinterpret_cast<const f32x2*>(va + (xp + 2) * 128);
                    __builtin_amdgcn_sched_barrier(0);
                    acc[2 * xp] = __builtin_amdgcn_mfma_f32_16x16x4f32(bq[2 * xp], fa[xp % 3].x, acc[2 * xp], 0, 0, 0);
                    acc[2 * xp + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(bq[2 * xp + 1], fa[xp % 3].y, acc[2 * xp + 1], 0, 0, 0);
                    if ((xp & 1) && !(CCVPE_W4P_SWITCH & 1)) {
                        __builtin_amdgcn_sched_barrier(0);
                        CCVPE_W4P_LOAD_B(wb, ksn, xp >> 1);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();   // the other buffer is complete (every wave waited for its pieces); this one is free
            vbuf ^= 1;
        }

        // ---- inverse transform A^T M A, bias, activation, store (kernels_wino4.hip) ----
        {
            const int b = mb / (mbx * mby);
            const int rem = mb - b * (mbx * mby);
            const int by = rem / mbx, bx = rem - by * mbx;
            const int n = (nb * NW + wave) * 16 + 4 * (lane >> 4);
            const bool nok = n < p.N;
            const f32x4 bias = (split || !nok) ? f32x4{0.f, 0.f, 0.f, 0.f} : *reinterpret_cast<const f32x4*>(p.bias + n);
            const float lo = act == ACT_RELU ? 0.f : -__builtin_inff();
            const size_t pix0 = ((size_t)b * p.H + (size_t)by * 16) * p.W + (size_t)bx * 16;
            float* obase = split ? p.partial + ((size_t)blockIdx.z * p.M + pix0) * p.N : p.dst[0].ptr + pix0 * ld + p.dst[0].coff;
            const __amdgpu_buffer_rsrc_t o_rsrc = __builtin_amdgcn_make_buffer_rsrc(obase, 0, 0x7fffffff, 0x00020000);
            const int oty = (lane >> 2) & 3, otx = lane & 3;
            const unsigned o_lane = nok ? (unsigned)((((oty * 4) * p.W + otx * 4) * ld + n) * 4) : OOB;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float tt[4][6];
#pragma unroll
                for (int q = 0; q < 6; ++q)
                    w4p_at(acc[0 * 6 + q][i], acc[1 * 6 + q][i], acc[2 * 6 + q][i], acc[3 * 6 + q][i], acc[4 * 6 + q][i], acc[5 * 6 + q][i],
                           tt[0][q], tt[1][q], tt[2][q], tt[3][q]);
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    float yy[4];
                    w4p_at(tt[a][0], tt[a][1], tt[a][2], tt[a][3], tt[a][4], tt[a][5], yy[0], yy[1], yy[2], yy[3]);
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
                    const int soff = ((a * p.W + c) * ld) * 4;
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, v), o_rsrc, o_lane, soff, 0);
                    // gfx950: two wait states before anything may overwrite the store's data registers (tests/test_isa_hazard.py)
                    __builtin_amdgcn_sched_barrier(0);
                    asm volatile("s_nop 1");
                    __builtin_amdgcn_sched_barrier(0);
                }
        }
        if (!have_n) break;
#pragma unroll
        for (int x = 0; x < 36; ++x) acc[x] = f32x4{0.f, 0.f, 0.f, 0.f};
        item = item_n; nb = nb_n; mb = mb_n; w_base = w_base_n;
    }
#undef CCVPE_W4P_WBASE
#undef CCVPE_W4P_DMA
#undef CCVPE_W4P_LOAD_B
}

static size_t w4p_v_floats(const ConvParams& p) {
    return (size_t)p.B * (p.W >> 4) * (p.H >> 4) * ((p.Cin + W4P_GCH - 1) / W4P_GCH) * W4P_VFLOATS;
}

template <int NW>
static void launch_wino4p_plain(const ConvParams& p_in, hipStream_t s) {
    ConvParams p = p_in;
    if (p.splitk > 1) {
        const int ngr = (p.Cin + W4P_GCH - 1) / W4P_GCH;
        const int per = (ngr + p.splitk - 1) / p.splitk;
        p.splitk = (ngr + per - 1) / per;
    }
    constexpr size_t lds = 2 * W4P_VFLOATS * sizeof(float);
    static_assert(2 * lds <= 160 * 1024, "two workgroups per CU");
    static LdsAttr attr;
    auto kern = conv_wino4p_kernel<NW>;
    ensure_dynamic_lds(attr, reinterpret_cast<const void*>(kern), lds);
    const int mblocks = p.B * (p.W >> 4) * (p.H >> 4);
    const int nblocks = p.wino_nb ? p.wino_nb : (p.wino_n16 + NW - 1) / NW;
    const int resident = (NW == 4 ? 2 : 1) * 256 / (p.splitk > 1 ? p.splitk : 1);
    dim3 grid(std::min(mblocks * nblocks, std::max(resident, 8)), 1, p.splitk > 1 ? p.splitk : 1);
    CCVPE_LAUNCH(kern, grid, dim3(NW * 64), lds, s, p, (const float*)p.wino4_v, (unsigned)(w4p_v_floats(p) * sizeof(float)));
    if (p.splitk > 1) launch_splitk_reduce(p, s);
}

static thread_local bool g_tailp_applied = false;
bool conv_wino4p_tail_applied() { return g_tailp_applied; }

template <int NW>
static void launch_wino4p(const ConvParams& p, hipStream_t s) {
    g_tailp_applied = false;
    {   // V = B^T d B, once per layer
        const int ngr = (p.Cin + W4P_GCH - 1) / W4P_GCH;
        const int mblocks = p.B * (p.W >> 4) * (p.H >> 4);
        CCVPE_LAUNCH(wino4_input_transform_kernel, dim3(mblocks * ngr), dim3(256), 0, s, p, p.wino4_v, ngr);
    }
    if (p.splitk != 255) { launch_wino4p_plain<NW>(p, s); return; }
    ConvParams a = p;
    a.splitk = 1;
    const int mblocks = p.B * (p.W >> 4) * (p.H >> 4);
    const int nblocks = (p.wino_n16 + NW - 1) / NW;
    const int resident = (NW == 4 ? 2 : 1) * 256;
    const int rem = (mblocks * nblocks) % resident;
    const int ngr = (p.Cin + W4P_GCH - 1) / W4P_GCH;
    const int tail_nb = mblocks > 0 ? rem / mblocks : 0;
    int split = tail_nb > 0 ? std::min(resident / rem, ngr / 2) : 0;
    const int n0 = (nblocks - tail_nb) * NW * 16;
    if (rem == 0 || rem % mblocks != 0 || tail_nb >= nblocks || split < 2 || p.partial == nullptr ||
        (size_t)split * p.M * (size_t)(p.N - n0) > p.partial_floats || (n0 & 3) != 0) {
        launch_wino4p_plain<NW>(a, s);
        return;
    }
    g_tailp_applied = true;
    a.wino_nb = nblocks - tail_nb;
    launch_wino4p_plain<NW>(a, s);
    ConvParams b = p;
    b.wino_nb = tail_nb; b.wino_n16_off = (nblocks - tail_nb) * NW;
    b.N = p.N - n0; b.bias = p.bias + n0; b.dst[0].coff = p.dst[0].coff + n0;
    b.splitk = split;
    launch_wino4p_plain<NW>(b, s);
}

void launch_wino4p_64(const ConvParams& p, hipStream_t s) { launch_wino4p<4>(p, s); }
void launch_wino4p_128(const ConvParams& p, hipStream_t s) { launch_wino4p<8>(p, s); }

// the split form needs the per-stream V scratch of the plan (ConvParams::wino4_v) and byte offsets below 4 GiB
bool conv_wino4p_supported(const ConvParams& p) {
    return conv_wino4_supported(p) && p.wino4_v != nullptr && w4p_v_floats(p) <= p.wino4_v_floats && w4p_v_floats(p) * sizeof(float) < 0xffff0000ull;
}

}  // namespace ccvpe
