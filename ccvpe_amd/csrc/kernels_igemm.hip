// Implicit-GEMM convolution / transposed convolution on the fp32 matrix cores of gfx950.
//
// Replaces F.conv2d / ConvTranspose2d / Linear call sites of the reference hot path:
//   decoder double_conv 3x3 (models.py:42-47), deconvN k2 s2 (models.py:407-446), MBConv expand /
//   project / head 1x1 (efficientnet_pytorch/model.py:103-106,121-122,299), ground descriptor 1x1
//   heads (models.py:355-395) and the aerial descriptor Linear(5120,D) == conv k2 s2 (models.py:400-402,
//   471-482).
//
// Design (MI355X):  one workgroup = 4 wave64 = BM x BN output tile, K walked in 32-deep tiles that
// are 4 "chunks" of 8 input channels of one filter tap, gathered straight from the NHWC activation
// (no im2col buffer).  The halo / ragged rows are handled without branches: activations are read with
// raw buffer loads whose out-of-range offsets return zero.  Tiles are register-staged
// (buffer_load_dwordx4 -> ds_write_b128) into a double-buffered LDS image with 144-byte rows
// (conflict-free ds_read_b128), one barrier per K tile, the next tile's loads in flight under the
// current tile's MFMAs.  v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32 are exact fp32 (bitwise an
// fmaf chain) at 64 FLOP/clk/SIMD; the 16x16 form serves narrow outputs (N = 16, 40, 80 ...) where a
// 32-wide tile would waste matrix-core cycles on padding.  One ds_read_b128 per operand feeds four
// MFMAs: lane l holds k = 4*(l / MT) + j of its row for j = 0..3, the same k permutation on A and B.
#include "igemm_common.h"

#include <algorithm>

namespace ccvpe {

template <int MT> struct Mfma;
template <> struct Mfma<32> {
    using acc_t = f32x16;
    static constexpr int NACC = 16;
    static __device__ __forceinline__ acc_t run(float a, float b, acc_t c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
    // D[n][m] of the swapped product: lane holds column m = lane & 31 and, per quad q = 0..3, rows n = 8 q + 4 (lane >> 5) + 0..3
    static constexpr int NQ = 4;
    static __device__ __forceinline__ int mrow(int lane) { return lane & 31; }
    static __device__ __forceinline__ int ncol(int q, int lane) { return 8 * q + 4 * (lane >> 5); }
};
template <> struct Mfma<16> {
    using acc_t = f32x4;
    static constexpr int NACC = 4;
    static __device__ __forceinline__ acc_t run(float a, float b, acc_t c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
    // lane holds column m = lane & 15 and rows n = 4 (lane >> 4) + 0..3
    static constexpr int NQ = 1;
    static __device__ __forceinline__ int mrow(int lane) { return lane & 15; }
    static __device__ __forceinline__ int ncol(int, int lane) { return 4 * (lane >> 4); }
};

// NS = LDS stages: 2 (next tile stored while the current one is multiplied, one barrier per K tile) or 1 (half the
// LDS, two barriers per K tile: for the 1-3 tile deep encoder / deconv layers, which are bound by how many workgroups
// a CU can keep in flight, not by the K loop)
template <int BM, int BN, int WGM, int WGN, int MT, bool GATE, int NS = 2>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvParams p) {
    static_assert(WGM * WGN == 4, "4 waves per workgroup");
    constexpr int WM = BM / WGM, WN = BN / WGN;
    constexpr int TM = WM / MT, TN = WN / MT;
    static_assert(TM >= 1 && TN >= 1 && TM * MT * WGM == BM && TN * MT * WGN == BN, "tile must be whole MFMA tiles");
    constexpr int AR = BM / 32;          // float4 rows each thread stages for A
    constexpr int BR = (BN + 31) / 32;   // ... for B (last one partially used when BN % 32 != 0)
    constexpr int KSTEP = 64 / MT * 4;   // k covered by one ds_read_b128 across the wave: 8 (MT 32) or 16 (MT 16)
    constexpr int NKK = BK / KSTEP;
    using M = Mfma<MT>;
    using acc_t = typename M::acc_t;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                  // [2][BM][LDK]
    float* Bs = smem + NS * BM * LDK;  // [NS][BN][LDK]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int m0 = xcd_remap(blockIdx.x, gridDim.x) * BM;
    const int n0 = blockIdx.y * BN;
    const int kq = tid & 7;    // float4 slot inside the 32-deep K tile
    const int r0 = tid >> 3;   // 0..31

    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, p.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t gate_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(GATE ? p.gate : p.in), 0, GATE ? p.gate_bytes : 0, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;   // beyond any tensor (< 2 GiB): the buffer load returns zeros

    // ---- per-row gather state (rows r0 + 32*j of the A tile) ----
    int a_base[AR], a_iy[AR], a_ix[AR], a_gb[AR];
    const int ohw = p.OH * p.OW;
#pragma unroll
    for (int j = 0; j < AR; ++j) {
        const int m = m0 + r0 + 32 * j;
        const bool ok = m < p.M;
        const int mm = ok ? m : 0;
        const int b = mm / ohw;
        const int rem = mm - b * ohw;
        const int oy = rem / p.OW;
        const int ox = rem - oy * p.OW;
        const int iy0 = oy * p.stride - p.pad_t;
        const int ix0 = ox * p.stride - p.pad_l;
        a_base[j] = (((b * p.H + iy0) * p.W + ix0) * p.in_ld + (kq & 1) * 4) * 4;   // bytes
        a_iy[j] = ok ? iy0 : -(1 << 28);
        a_ix[j] = ix0;
        a_gb[j] = (b * p.Cin + (kq & 1) * 4) * 4;
    }
    const float* wrow[BR];
#pragma unroll
    for (int j = 0; j < BR; ++j) wrow[j] = p.wpk + (size_t)min(n0 + r0 + 32 * j, p.Npad - 1) * p.Kpad + kq * 4;

    f32x4 ra[AR], rb[BR];          // plain LLVM vectors (HIP's float4 struct arrays ended up in scratch)
    f32x4 rg[GATE ? AR : 1];

#define CCVPE_LOAD_TILE(kt, live_)                                                                              \
    {                                                                                                    \
        const int g = (kt) * 4 + (kq >> 1);                                                              \
        int tap, c0;                                                                                     \
        chunk_to_tap(p, g, tap, c0);                                                                     \
        const int ky = (tap * p.div_kw_mul) >> 5;                                                        \
        const int kx = tap - ky * p.KW;                                                                  \
        const bool gok = (g < p.nchunks) & (live_);                                                      \
        const int koff = ((ky * p.W + kx) * p.in_ld + c0) * 4;                                           \
        _Pragma("unroll") for (int j = 0; j < AR; ++j) {                                                 \
            const int iy = a_iy[j] + ky, ix = a_ix[j] + kx;                                              \
            const bool ok = gok & ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W);       \
            const unsigned off = ok ? (unsigned)(a_base[j] + koff) : OOB;                                \
            ra[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, off, 0, 0)); \
            if (GATE) {                                                                                  \
                rg[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(gate_rsrc, ok ? (unsigned)(a_gb[j] + c0 * 4) : OOB, 0, 0)); \
            }                                                                                            \
        }                                                                                                \
        _Pragma("unroll") for (int j = 0; j < BR; ++j) rb[j] = *reinterpret_cast<const f32x4*>(wrow[j] + (kt) * BK); \
    }
#define CCVPE_STORE_TILE(stage)                                                                          \
    {                                                                                                    \
        float* as_ = As + (stage) * BM * LDK;                                                            \
        float* bs_ = Bs + (stage) * BN * LDK;                                                            \
        _Pragma("unroll") for (int j = 0; j < AR; ++j) {                                                 \
            f32x4 v_ = ra[j];                                                                            \
            if (GATE) v_ *= rg[j];                                                                       \
            *reinterpret_cast<f32x4*>(as_ + (r0 + 32 * j) * LDK + kq * 4) = v_;                          \
        }                                                                                                \
        _Pragma("unroll") for (int j = 0; j < BR; ++j)                                                   \
            if (BN % 32 == 0 || r0 + 32 * j < BN) *reinterpret_cast<f32x4*>(bs_ + (r0 + 32 * j) * LDK + kq * 4) = rb[j]; \
    }

    acc_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < M::NACC; ++r) acc[i][j][r] = 0.f;

    // split-K: blockIdx.z owns K tiles [kt0, kt1); raw partial sums go to a slab, splitk_reduce_kernel finishes
    const int nkt_all = p.Kpad / BK;
    int kt0 = 0, kt1 = nkt_all;
    if (p.splitk > 1) {
        const int per = (nkt_all + p.splitk - 1) / p.splitk;
        kt0 = min((int)blockIdx.z * per, nkt_all);
        kt1 = min(kt0 + per, nkt_all);
    }
    {
        const int kfirst = min(kt0, nkt_all - 1);
        CCVPE_LOAD_TILE(kfirst, true);
    }
    CCVPE_STORE_TILE(0);
    __syncthreads();

    const int a_row = wm * WM + (lane % MT);
    const int b_row = wn * WN + (lane % MT);
    const int k_lane = (lane / MT) * 4;

    for (int kt = kt0; kt < kt1; ++kt) {
        const int stage = NS == 2 ? (kt - kt0) & 1 : 0;
        // unconditional prefetch keeps the loop free of branches (and the staging registers out of scratch); in the
        // last iteration the activation loads are pointed out of range (zero fill, no memory traffic) - for the
        // 1-3 tile deep encoder layers a real re-read of the tile was 33-100 % extra L2 traffic
        const int ktn = min(kt + 1, kt1 - 1);
        CCVPE_LOAD_TILE(ktn, kt + 1 < kt1);
        // keep the prefetch ABOVE the MFMA block: without this fence hipcc sinks the loads to just before
        // the ds_writes (to shorten register live ranges) and every K tile eats a full memory latency
        __builtin_amdgcn_sched_barrier(0);
        const float* as = As + stage * BM * LDK + a_row * LDK + k_lane;
        const float* bs = Bs + stage * BN * LDK + b_row * LDK + k_lane;
        // fragments double-buffered in registers: kk+1's ds_reads are in flight under kk's MFMAs
        f32x4 a[2][TM], b[2][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) a[0][i] = *reinterpret_cast<const f32x4*>(as + i * MT * LDK);
#pragma unroll
        for (int j = 0; j < TN; ++j) b[0][j] = *reinterpret_cast<const f32x4*>(bs + j * MT * LDK);
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) {
            const int cur = kk & 1, nxt = cur ^ 1;
            if (kk + 1 < NKK) {
#pragma unroll
                for (int i = 0; i < TM; ++i) a[nxt][i] = *reinterpret_cast<const f32x4*>(as + i * MT * LDK + (kk + 1) * KSTEP);
#pragma unroll
                for (int j = 0; j < TN; ++j) b[nxt][j] = *reinterpret_cast<const f32x4*>(bs + j * MT * LDK + (kk + 1) * KSTEP);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    // weights as the A operand: accumulators are channel-major (element r of a quad = 4 consecutive n of ONE
                    // row m), so the epilogue writes 16-byte pieces of C rows
                    acc[i][j] = M::run(b[cur][j].x, a[cur][i].x, acc[i][j]);
                    acc[i][j] = M::run(b[cur][j].y, a[cur][i].y, acc[i][j]);
                    acc[i][j] = M::run(b[cur][j].z, a[cur][i].z, acc[i][j]);
                    acc[i][j] = M::run(b[cur][j].w, a[cur][i].w, acc[i][j]);
                }
        }
        // ... and the LDS stores (which wait for those loads) BELOW it
        __builtin_amdgcn_sched_barrier(0);
        if (NS == 1) __syncthreads();   // single stage: every wave is done reading before the tile is replaced
        CCVPE_STORE_TILE(NS == 2 ? stage ^ 1 : 0);
        __syncthreads();
    }
#undef CCVPE_LOAD_TILE
#undef CCVPE_STORE_TILE

    // ---- epilogue ----
    // (1) accumulators (+bias, activation) -> LDS C tile [BM][BN+4] (the staging buffers are dead now);
    // (2) rows leave as 16-byte stores: coalesced full rows instead of 4-byte-per-lane column slivers,
    //     with the residual read and the pixel-shuffle / concat addressing done per float4.
    constexpr int LDC = BN + 4;
    float* Cs = smem;
    const bool split = p.splitk > 1;
    const int act = split ? ACT_NONE : p.act;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
#pragma unroll
        for (int q = 0; q < M::NQ; ++q) {
            const int nl = wn * WN + j * MT + M::ncol(q, lane);
            const int n = n0 + nl;
            f32x4 bias = {0.f, 0.f, 0.f, 0.f};
            if (!split) {
#pragma unroll
                for (int e = 0; e < 4; ++e) bias[e] = n + e < p.N ? p.bias[n + e] : 0.f;
            }
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int ml = wm * WM + i * MT + M::mrow(lane);
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * q + e] + bias[e];
                if (act == ACT_SWISH) {          // uniform: one branch per quad instead of one per element
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = apply_act(v[e], ACT_SWISH);
                } else if (act == ACT_RELU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                }
                *reinterpret_cast<f32x4*>(Cs + ml * LDC + nl) = v;
            }
        }
    }
    __syncthreads();
    constexpr int C4 = BN / 4;
    const __amdgpu_buffer_rsrc_t slab_rsrc = __builtin_amdgcn_make_buffer_rsrc(split ? p.partial : const_cast<float*>(p.in), 0,
                                                                                split ? (unsigned)p.splitk * (unsigned)p.M * (unsigned)p.N * 4u : 0u, 0x00020000);
    for (int it = tid; it < BM * C4; it += 256) {
        const int ml = it / C4, c4 = it - ml * C4;
        const int m = m0 + ml, n = n0 + c4 * 4;
        if (m >= p.M || n >= p.N) continue;
        const f32x4 v = *reinterpret_cast<const f32x4*>(Cs + ml * LDC + c4 * 4);
        if (split) {
            float* dst = p.partial + ((size_t)blockIdx.z * p.M + m) * p.N + n;
            if (p.split_fused) {   // write-through: the slabs are handed to the last arriver of this tile (ticket.h)
                if ((p.N & 3) == 0) st_sc1_f4(slab_rsrc, (unsigned)(((size_t)blockIdx.z * p.M + m) * p.N + n) * 4u, v);
                else
                    for (int e = 0; e < 4 && n + e < p.N; ++e) st_sc1(dst + e, v[e]);
            } else if ((p.N & 3) == 0) *reinterpret_cast<f32x4*>(dst) = v;
            else
                for (int e = 0; e < 4 && n + e < p.N; ++e) dst[e] = v[e];
        } else {
            emit_out4(p, m, n, v);
        }
    }
    if (split && p.split_fused) {
        // self-reducing split-K: the last of this tile's K slices sums the slabs (slice order) and runs the epilogue; the C tile in
        // LDS is dead behind the ticket's first barrier, its first word serves as the flag
        if (splitk_ticket(p, blockIdx.y * gridDim.x + blockIdx.x, reinterpret_cast<unsigned*>(Cs)))
            splitk_finish<256>(p, m0, 1, BM, 0, n0, BN);
    }
}

// second half of a split-K launch: sum the slabs, then bias / activation / placement as the fused epilogue
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const ConvParams p) {
    const int n4 = (p.N + 3) >> 2;
    const long long total = (long long)p.M * n4;
    for (long long it = (long long)blockIdx.x * 256 + threadIdx.x; it < total; it += (long long)gridDim.x * 256) {
        const int m = (int)(it / n4);
        const int n = (int)(it - (long long)m * n4) * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if ((p.N & 3) == 0) {
            // eight slabs at a time: eight independent loads, added in slice order (as one load per iteration the loop waited a full
            // memory latency per slab: 12 slabs = 8 of the 9.6 us this kernel took per launch in a batch-1 frame)
            const size_t zs = (size_t)p.M * p.N;
            const float* src = p.partial + (size_t)m * p.N + n;
            for (int z0 = 0; z0 < p.splitk; z0 += 8) {
                f32x4 t[8];
#pragma unroll
                for (int z = 0; z < 8; ++z) t[z] = *reinterpret_cast<const f32x4*>(src + (size_t)min(z0 + z, p.splitk - 1) * zs);
#pragma unroll
                for (int z = 0; z < 8; ++z) if (z0 + z < p.splitk) v += t[z];
            }
        } else {
            for (int z = 0; z < p.splitk; ++z) {
                const float* src = p.partial + ((size_t)z * p.M + m) * p.N + n;
                for (int e = 0; e < 4 && n + e < p.N; ++e) v[e] += src[e];
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = apply_act(v[e] + (n + e < p.N ? p.bias[n + e] : 0.f), p.act);
        emit_out4(p, m, n, v);
    }
}

void launch_splitk_reduce(const ConvParams& p, hipStream_t s) {
    const long long total = (long long)p.M * ((p.N + 3) / 4);
    const int blocks = (int)std::min<long long>((total + 255) / 256, 4096);
    CCVPE_LAUNCH(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, s, p);
}

template <int BM, int BN, int WGM, int WGN, int MT, bool GATE, int NS>
static void launch_cfg2(const ConvParams& p, hipStream_t s) {
    constexpr size_t lds = std::max<size_t>(NS * (BM + BN) * LDK, BM * (BN + 4)) * sizeof(float);   // stages | epilogue C tile
    static LdsAttr attr;
    auto kern = conv_igemm_kernel<BM, BN, WGM, WGN, MT, GATE, NS>;
    ensure_dynamic_lds(attr, reinterpret_cast<const void*>(kern), lds);
    dim3 grid((p.M + BM - 1) / BM, (p.N + BN - 1) / BN, p.splitk > 1 ? p.splitk : 1);
    if (p.splitk <= 1 || p.tickets == nullptr || (long long)grid.x * grid.y > CONV_TICKETS) {
        if (p.split_fused) { ConvParams q = p; q.split_fused = 0; CCVPE_LAUNCH(kern, grid, dim3(256), lds, s, q); if (q.splitk > 1) launch_splitk_reduce(q, s); return; }
    }
    CCVPE_LAUNCH(kern, grid, dim3(256), lds, s, p);
    if (p.splitk > 1 && !p.split_fused) launch_splitk_reduce(p, s);
}

template <int BM, int BN, int WGM, int WGN, int MT, int NS = 2>
static void launch_cfg(const ConvParams& p, hipStream_t s) {
    if (p.gate) launch_cfg2<BM, BN, WGM, WGN, MT, true, NS>(p, s);
    else launch_cfg2<BM, BN, WGM, WGN, MT, false, NS>(p, s);
}

// Tile table.  id = index + 1 (0 is TILE_AUTO).  `intrinsic` is only the prior used when the plan has not been
// autotuned (ccvpe_api.hip times every candidate per layer on the device and keeps the fastest).
struct TileCfg { int bm, bn; double intrinsic; const char* name; void (*launch)(const ConvParams&, hipStream_t); };
static const TileCfg TILES[] = {
    {128, 128, 0.95, "conv_igemm_128x128_m32", launch_cfg<128, 128, 2, 2, 32>},
    {128, 64, 0.85, "conv_igemm_128x64_m32", launch_cfg<128, 64, 2, 2, 32>},
    {64, 64, 0.85, "conv_igemm_64x64_m32", launch_cfg<64, 64, 2, 2, 32>},
    {128, 32, 0.85, "conv_igemm_128x32_m32", launch_cfg<128, 32, 4, 1, 32>},
    {256, 16, 0.60, "conv_igemm_256x16_m16", launch_cfg<256, 16, 4, 1, 16>},
    {128, 48, 0.85, "conv_igemm_128x48_m16", launch_cfg<128, 48, 4, 1, 16>},
    {128, 80, 1.00, "conv_igemm_128x80_m16", launch_cfg<128, 80, 4, 1, 16>},
    {256, 32, 0.70, "conv_igemm_256x32_m32", launch_cfg<256, 32, 4, 1, 32>},
    {128, 128, 1.00, "conv_igemm_128x128_m16", launch_cfg<128, 128, 2, 2, 16>},
    {128, 64, 0.95, "conv_igemm_128x64_m16", launch_cfg<128, 64, 2, 2, 16>},
    {128, 32, 0.85, "conv_igemm_128x32_m16", launch_cfg<128, 32, 4, 1, 16>},
    {128, 16, 0.60, "conv_igemm_128x16_m16", launch_cfg<128, 16, 4, 1, 16>},
    {128, 96, 1.00, "conv_igemm_128x96_m16", launch_cfg<128, 96, 4, 1, 16>},
    {128, 112, 1.00, "conv_igemm_128x112_m16", launch_cfg<128, 112, 4, 1, 16>},
    {64, 64, 0.85, "conv_igemm_64x64_m16", launch_cfg<64, 64, 2, 2, 16>},
    {64, 32, 0.70, "conv_igemm_64x32_m16", launch_cfg<64, 32, 2, 2, 16>},
    {256, 48, 0.85, "conv_igemm_256x48_m16", launch_cfg<256, 48, 4, 1, 16>},
    {64, 128, 0.90, "conv_igemm_64x128_m16", launch_cfg<64, 128, 2, 2, 16>},
    // single-stage forms of the tiles the shallow layers use (prior 0: only the autotuner picks them)
    {64, 64, 0.0, "conv_igemm_64x64_m16_s1", launch_cfg<64, 64, 2, 2, 16, 1>},
    {64, 32, 0.0, "conv_igemm_64x32_m16_s1", launch_cfg<64, 32, 2, 2, 16, 1>},
    {64, 64, 0.0, "conv_igemm_64x64_m32_s1", launch_cfg<64, 64, 2, 2, 32, 1>},
    {128, 48, 0.0, "conv_igemm_128x48_m16_s1", launch_cfg<128, 48, 4, 1, 16, 1>},
    {128, 16, 0.0, "conv_igemm_128x16_m16_s1", launch_cfg<128, 16, 4, 1, 16, 1>},
    {128, 80, 0.0, "conv_igemm_128x80_m16_s1", launch_cfg<128, 80, 4, 1, 16, 1>},
};
static constexpr int NTILES = (int)(sizeof(TILES) / sizeof(TILES[0]));

int conv_igemm_npad() { return 128; }
int conv_igemm_num_tiles() { return NTILES + bf16x3_num_tiles() + wino_num_tiles() + pw_num_tiles(); }
bool conv_igemm_tile_is_wino(int tile) { tile &= 0xff; return tile > NTILES + bf16x3_num_tiles() && tile <= NTILES + bf16x3_num_tiles() + wino_num_tiles(); }
static int pw_index(int tile) { return (tile & 0xff) - NTILES - bf16x3_num_tiles() - wino_num_tiles() - 1; }
bool conv_igemm_tile_is_pw(int tile) { const int i = pw_index(tile); return i >= 0 && i < pw_num_tiles(); }
bool conv_igemm_tile_is_proj(int tile) { return conv_igemm_tile_is_pw(tile) && pw_tile(pw_index(tile))->proj_rt > 0; }
int conv_igemm_tile_proj_rt(int tile) { return conv_igemm_tile_is_pw(tile) ? conv_pw_tile_proj_rt(pw_index(tile)) : 0; }
int conv_proj_lat_tile() {
    for (int i = 0; i < pw_num_tiles(); ++i)
        if (pw_tile(i)->proj_rt == 101) return NTILES + bf16x3_num_tiles() + wino_num_tiles() + i + 1;
    return 0;
}
bool conv_igemm_tile_is_wino4(int tile) { return conv_igemm_tile_is_wino(tile) && wino_tile((tile & 0xff) - NTILES - bf16x3_num_tiles() - 1)->f == 4; }
bool conv_igemm_tile_is_wino4p(int tile) { return conv_igemm_tile_is_wino4(tile) && wino_tile((tile & 0xff) - NTILES - bf16x3_num_tiles() - 1)->pre; }
static int wino4x_cfg_of(int tile) { return conv_igemm_tile_is_wino(tile) ? wino_tile((tile & 0xff) - NTILES - bf16x3_num_tiles() - 1)->xcfg : -1; }
bool conv_igemm_tile_is_wino4x(int tile) { return wino4x_cfg_of(tile) >= 0; }
int conv_igemm_tile_wino4x_cfg(int tile) { return wino4x_cfg_of(tile); }
bool conv_wino_tile_supported(const ConvParams& p, int tile) {
    if (!conv_igemm_tile_is_wino(tile)) return false;
    if (conv_igemm_tile_is_wino4x(tile)) return conv_wino4x_supported(p) && p.wino4x_cfg == wino4x_cfg_of(tile);
    if (conv_igemm_tile_is_wino4p(tile)) return conv_wino4p_supported(p);
    return conv_igemm_tile_is_wino4(tile) ? conv_wino4_supported(p) : conv_wino_supported(p);
}
bool conv_igemm_tile_is_bf16x3(int tile) { tile &= 0xff; return tile > NTILES && tile <= NTILES + bf16x3_num_tiles(); }
// kernels that can reduce their own split-K (split code SPLIT_FUSED + S): the fp32 implicit GEMM and the fused F(4x4) / F(2x2) Winograd kernels
bool conv_igemm_tile_can_fuse_split(int tile) {
    tile &= 0xff;
    return (tile >= 1 && tile <= NTILES) || (conv_igemm_tile_is_wino(tile) && !conv_igemm_tile_is_wino4p(tile) && !conv_igemm_tile_is_wino4x(tile)) ||
           conv_igemm_tile_proj_rt(tile) >= 100;   // (the latency form of the deep-K GEMM: layers without a gate)
}
static void tile_dims(int tile, int& bm, int& bn) {
    if (tile >= 1 && tile <= NTILES) { bm = TILES[tile - 1].bm; bn = TILES[tile - 1].bn; }
    else if (conv_igemm_tile_is_bf16x3(tile)) { bm = (*bf16x3_tile(tile - NTILES - 1)).bm; bn = (*bf16x3_tile(tile - NTILES - 1)).bn; }
    else if (conv_igemm_tile_is_wino(tile)) { bm = wino_tile(tile - NTILES - bf16x3_num_tiles() - 1)->bm; bn = wino_tile(tile - NTILES - bf16x3_num_tiles() - 1)->bn; }
    else if (conv_igemm_tile_is_pw(tile)) { bm = pw_tile(pw_index(tile))->bm; bn = pw_tile(pw_index(tile))->bn; }
    else { bm = bn = 0; }
}

// fraction of the launched MFMA work that is useful (padding of M and N to the tile)
double conv_igemm_tile_util(const ConvParams& p, int tile) {
    int bm, bn;
    tile_dims(tile, bm, bn);
    if (!bm) return 0.0;
    double gm = (p.M + bm - 1) / bm, gn = (p.N + bn - 1) / bn;
    return ((double)p.M * p.N) / (gm * bm * gn * bn);
}

long long conv_igemm_tile_blocks(const ConvParams& p, int tile) {
    int bm, bn;
    tile_dims(tile, bm, bn);
    if (!bm) return 0;
    return (long long)((p.M + bm - 1) / bm) * ((p.N + bn - 1) / bn);
}

static int pick_tile(const ConvParams& p) {
    const double cus = 256.0;
    int best = 3;
    double best_score = -1.0;
    for (int t = 1; t <= NTILES; ++t) {
        const TileCfg& c = TILES[t - 1];
        double gm = (p.M + c.bm - 1) / c.bm, gn = (p.N + c.bn - 1) / c.bn;
        double blocks = gm * gn;
        double util = conv_igemm_tile_util(p, t);
        double rounds = (double)(long long)((blocks + cus - 1) / cus);
        double quant = blocks / (rounds * cus);
        double score = util * quant * c.intrinsic;
        if (score > best_score) { best_score = score; best = t; }
    }
    return best;
}

static thread_local int g_last_tile = 0;
int conv_igemm_last_tile() { int t = g_last_tile; g_last_tile = 0; return t; }
const char* conv_igemm_tile_name(int tile) {
    tile &= 0xff;
    if (tile >= 1 && tile <= NTILES) return TILES[tile - 1].name;
    if (conv_igemm_tile_is_bf16x3(tile)) return (*bf16x3_tile(tile - NTILES - 1)).name;
    if (conv_igemm_tile_is_wino(tile)) return wino_tile(tile - NTILES - bf16x3_num_tiles() - 1)->name;
    if (conv_igemm_tile_is_pw(tile)) return pw_tile(pw_index(tile))->name;
    return "";
}

// exact small-range division by multiplication: q = (g * mul) >> 20 for 0 <= g < limit
static int find_div_mul(int d, int limit) {
    const unsigned mul = ((1u << 20) + d - 1) / d;
    for (int g = 0; g < limit; ++g)
        if ((int)(((unsigned)g * mul) >> 20) != g / d) return -1;
    return (int)mul;
}

static void make_fast_div(unsigned d, unsigned& mul, unsigned& shift) {
    unsigned l = 0;
    while ((1u << l) < d) ++l;
    mul = (unsigned)((((unsigned long long)1 << 32) * (((unsigned long long)1 << l) - d)) / d + 1);
    shift = l;
}

int conv_igemm_prepare(ConvParams& p) {
    make_fast_div((unsigned)std::max(p.W, 1), p.fdw_mul, p.fdw_shift);
    make_fast_div((unsigned)std::max(p.H, 1), p.fdh_mul, p.fdh_shift);
    const int taps = p.KH * p.KW;
    const int limit = p.Kpad / 8 + 8;
    p.taps4 = 4 * taps;
    p.kfull_chunks = (p.Cin / 32) * 4 * taps;
    p.kfull_c0 = (p.Cin / 32) * 32;
    p.knc = (p.Cin % 32) / 8;
    const int mul = find_div_mul(p.taps4, limit);
    if (mul < 0 || (long long)limit * mul >= (1LL << 32)) return -1;
    p.div_4t_mul = mul;
    p.div_nc_mul = 256;   // knc == 0 or 1: idx / 1
    if (p.knc > 1) {
        int m = -1;
        for (int c = 1; c < 512 && m < 0; ++c) {
            bool ok = true;
            for (int i = 0; i < 4 * 16 && ok; ++i) ok = ((i * c) >> 8) == i / p.knc;
            if (ok) m = c;
        }
        if (m < 0) return -1;
        p.div_nc_mul = m;
    }
    // tap / KW for tap < 16: (tap * m) >> 5
    int kwm = -1;
    for (int m = 1; m < 64 && kwm < 0; ++m) {
        bool ok = true;
        for (int t = 0; t < 16 && ok; ++t) ok = ((t * m) >> 5) == t / p.KW;
        if (ok) kwm = m;
    }
    if (kwm < 0) return -1;
    p.div_kw_mul = kwm;
    return 0;
}

// host-side inverse of chunk_to_tap(): GEMM k index of (tap, channel c) for a layer with `cin` (padded)
// input channels and `taps` filter taps
int conv_igemm_k_index(int cin, int taps, int tap, int c) {
    const int full = cin / 32;
    const int cg = c / 32;
    int g;
    if (cg < full) g = cg * 4 * taps + tap * 4 + (c % 32) / 8;
    else g = full * 4 * taps + tap * ((cin % 32) / 8) + (c - full * 32) / 8;
    return g * 8 + (c % 8);
}

int launch_conv_igemm(const ConvParams& p_in, int tile, hipStream_t s) {
    ConvParams p = p_in;
    if (p.KH * p.KW > 16 || p.Cin % 8 || conv_igemm_prepare(p) != 0) return -1;   // geometry outside the supported range
    p.vec_epi = (p.N % 4 == 0) && (p.resid == nullptr || p.resid_ld % 4 == 0) && (p.mode != MODE_DECONV || p.deconv_cout % 4 == 0);
    for (int d = 0; d < p.ndst; ++d) p.vec_epi = p.vec_epi && p.dst[d].ld % 4 == 0 && p.dst[d].coff % 4 == 0;
    for (int d = 0; d < p.ndst; ++d)
        if (p.dst[d].split && !p.vec_epi) return -1;   // split destinations exist only on the 16-byte path
    int splitk = (tile >> 8) & 0xff;
    tile &= 0xff;
    bool fused = false;
    if (splitk > SPLIT_FUSED && splitk <= SPLIT_FUSED + 32) { fused = p.tickets != nullptr; splitk -= SPLIT_FUSED; }   // (no counters: the reduce launch)
    if (conv_igemm_tile_is_bf16x3(tile) && (p.w_hi == nullptr || p.w_lo == nullptr)) tile = 0;   // planes not packed: fp32 path
    if (conv_igemm_tile_is_wino(tile) && !conv_wino_tile_supported(p, tile)) tile = 0;
    if (conv_igemm_tile_is_pw(tile) && !conv_pw_tile_ok(pw_index(tile), p)) tile = 0;   // not a layer this pointwise tile takes
    if (conv_igemm_tile_is_pw(tile)) {   // the pointwise persistent tiles keep K whole; the latency form of the deep-K GEMM splits it only self-reducing, without a gate
        const bool lat = conv_igemm_tile_proj_rt(tile) >= 100;
        const int ct = lat ? std::max(1, (conv_igemm_tile_proj_rt(tile) - 100) % 10) : 1, rt = lat ? std::max(1, (conv_igemm_tile_proj_rt(tile) - 100) / 10) : 1;
        const long long regions = (long long)((p.M + 16 * rt - 1) / (16 * rt)) * (((p.N + 15) / 16 + ct - 1) / ct);
        if (!(lat && fused && p.gate == nullptr && p.se_rows == nullptr && p.partial != nullptr && regions <= CONV_TICKETS)) splitk = 1;
    }
    if (tile < 1 || tile > conv_igemm_num_tiles()) tile = pick_tile(p);
    if (p.in_split) {   // pre-split bf16 input: only the bf16x3 kernels can read it
        if (p.w_hi == nullptr) return -1;
        if (!conv_igemm_tile_is_bf16x3(tile)) tile = NTILES + 3;   // conv_bf16x3_64x64_m32: valid for any shape
    }
    if (splitk == 255) { if (!conv_igemm_tile_is_wino4(tile) || conv_igemm_tile_is_wino4x(tile) || p.partial == nullptr) splitk = 1; }   // F(4x4) tail split: sized by its launcher
    else if (splitk < 2 || p.partial == nullptr || (size_t)splitk * p.M * p.N > p.partial_floats) splitk = 1;
    p.splitk = splitk;
    // which kernels reduce their own split: the fp32 implicit GEMM, the F(4x4) and F(2x2) Winograd kernels
    fused = fused && splitk > 1 && splitk != 255 && conv_igemm_tile_can_fuse_split(tile);
    p.split_fused = (fused || (splitk == 255 && p.tickets != nullptr)) ? 1 : 0;       // (the tail split's K-split sub-launch reduces itself whenever it can)
    g_last_tile = tile | ((fused ? splitk + SPLIT_FUSED : splitk) << 8);
    if (tile <= NTILES) TILES[tile - 1].launch(p, s);
    else if (conv_igemm_tile_is_pw(tile)) pw_tile(pw_index(tile))->launch(p, s);
    else if (conv_igemm_tile_is_wino(tile)) wino_tile(tile - NTILES - bf16x3_num_tiles() - 1)->launch(p, s);
    else (*bf16x3_tile(tile - NTILES - 1)).launch(p, s);
    if (splitk == 255 && !(conv_igemm_tile_is_wino4p(tile) ? conv_wino4p_tail_applied() : conv_wino4_tail_applied())) g_last_tile = tile | (1 << 8);   // the tail split did not apply: a plain launch
    return 0;
}

}  // namespace ccvpe
