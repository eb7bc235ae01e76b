// Pointwise (1x1) convolution / k2s2 transposed convolution with a shallow K on the fp32 matrix cores: persistent,
// wave-local, no barrier in the loop.
//
// Call sites (all 1x1 GEMMs of the path): MBConv expand / project (efficientnet_pytorch/model.py:103-106, 121-122), the
// encoder head (:299), the ground descriptor heads (models.py:355-395) and the decoder deconvN = ConvTranspose2d(k2, s2)
// (models.py:407-446) as a GEMM [pixels x Cin] x [Cin x 4 Cout] with a pixel-shuffle epilogue.
//
// conv_igemm_kernel walks these layers in 1-6 K tiles of 32 channels: the tile prologue (gather state, first loads), the
// two staging hops and the C-tile epilogue dominate, the launches sit at 50-60 TFLOP/s and 1-3 TB/s - far from either
// roof (profiles/r02_kernel_trace_b32_fp32.md).  Here:
//   * the weight slab of the workgroup's column block ([16 TN][K], K <= 512) is staged in LDS ONCE per workgroup;
//   * a wave owns 16 consecutive GEMM rows (pixels) per step and loads them straight into the MFMA A-operand registers
//     (lane (l & 15, l >> 4) = row l & 15, channels 4 (l >> 4) .. + 3 of every 16-channel step - 64-byte pieces of the
//     NHWC rows), four 16-channel steps per chunk, the next chunk (of this or of the next row tile) in flight under the
//     current chunk's 16 TN MFMAs; the SE gate multiplies the A registers (project convs);
//   * the 16 x 16 TN result goes through a wave-private LDS patch and leaves as 16-byte row pieces through the same
//     emit_out4 as the implicit GEMM (bias, activation, residual, pixel shuffle, up to 3 concat destinations);
//   * persistent grid: a workgroup keeps its column block and strides over the row tiles, so nothing but s_waitcnt
//     separates the stages and 8 waves per CU cover each other's latencies.
#include "igemm_common.h"

#include <algorithm>

namespace ccvpe {

static constexpr int PW_KMAX = 512;        // deepest K (channels) the weight slab is staged for

template <int TN, bool GATE>
__global__ __launch_bounds__(256) void conv_pw_kernel(const ConvParams p) {
    constexpr int BN = 16 * TN;
    constexpr int LDC = BN + 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int Kp = p.Kpad;                 // multiple of 32
    const int ldb = Kp + 4;
    float* Bs = smem;                      // [BN][Kp + 4]
    float* Cs = smem + BN * ldb;           // [4 waves][16][LDC]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = blockIdx.y * BN;

    // ---- weight slab of this column block -> LDS, once ----
    {
        const int k4n = Kp >> 2;
        for (int i = tid; i < BN * k4n; i += 256) {
            const int n = i / k4n, k4 = i - n * k4n;
            *reinterpret_cast<f32x4*>(Bs + n * ldb + k4 * 4) = *reinterpret_cast<const f32x4*>(p.wpk + (size_t)min(n0 + n, p.Npad - 1) * Kp + k4 * 4);
        }
    }
    __syncthreads();

    const int mtiles = (p.M + 15) >> 4;
    const int nsteps = (p.Cin + 15) >> 4;               // 16-channel k-steps
    const int nchunks = (nsteps + 3) >> 2;              // chunks of 4 k-steps
    const int hw = p.OH * p.OW;
    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, p.in_bytes, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;
    const int kq4 = 4 * (lane >> 4);

    // row tile of this wave: tiles are dealt to (workgroup, wave) round robin; XCD-contiguous runs keep neighbours in one L2
    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    int mt = wg * 4 + wave;
    const int mt_stride = gridDim.x * 4;
    if (mt >= mtiles) return;

    float* cw = Cs + wave * 16 * LDC;
    const float* bp = Bs + (lane & 15) * ldb + kq4;

    // A registers of one chunk: a[j] = channels 16 (4 chunk + j) + kq4 .. + 3 of row (mt * 16 + (lane & 15))
    f32x4 acur[4], anext[4];
    auto load_chunk = [&](int mtile, int chunk, f32x4* dst) {
        const int m = mtile * 16 + (lane & 15);
        const bool rok = mtile < mtiles && m < p.M;
        const unsigned rbase = (unsigned)m * (unsigned)p.in_ld * 4u;
        const int b = GATE ? (rok ? m / hw : 0) : 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = (chunk * 4 + j) * 16 + kq4;
            const bool ok = rok && c < p.Cin;
            dst[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, ok ? rbase + (unsigned)c * 4u : OOB, 0, 0));
            if (GATE) {
                const f32x4 gv = ok ? *reinterpret_cast<const f32x4*>(p.gate + (size_t)b * p.Cin + c) : f32x4{0.f, 0.f, 0.f, 0.f};
                dst[j] *= gv;
            }
        }
    };
    load_chunk(mt, 0, acur);

    f32x4 acc[TN];
    while (mt < mtiles) {
#pragma unroll
        for (int t = 0; t < TN; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int ch = 0; ch < nchunks; ++ch) {
            // the chunk after this one: next chunk of this row tile, or the first chunk of the wave's next row tile
            const bool last = ch == nchunks - 1;
            load_chunk(last ? mt + mt_stride : mt, last ? 0 : ch + 1, anext);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int ks = ch * 4 + j;
                if (ks < nsteps) {
#pragma unroll
                    for (int t = 0; t < TN; ++t) {
                        const f32x4 bv = *reinterpret_cast<const f32x4*>(bp + t * 16 * ldb + ks * 16);
                        // weights as the A operand: channel-major accumulators (4 consecutive columns of ONE row per lane)
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv.x, acur[j].x, acc[t], 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv.y, acur[j].y, acc[t], 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv.z, acur[j].z, acc[t], 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv.w, acur[j].w, acc[t], 0, 0, 0);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; ++j) acur[j] = anext[j];
        }
        // ---- epilogue: bias + activation -> wave-private C patch -> 16-byte row pieces ----
#pragma unroll
        for (int t = 0; t < TN; ++t) {
            const int n = n0 + t * 16 + kq4;
            f32x4 v;
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = acc[t][i] + (n + i < p.N ? p.bias[n + i] : 0.f);
            if (p.act == ACT_SWISH) {            // uniform
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = apply_act(v[i], ACT_SWISH);
            } else if (p.act == ACT_RELU) {
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.f);
            }
            *reinterpret_cast<f32x4*>(cw + (lane & 15) * LDC + t * 16 + kq4) = v;
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0) only: LDS traffic of this wave is done; global loads / stores stay in flight
        __builtin_amdgcn_wave_barrier();
        constexpr int C4 = BN / 4;
        for (int it = lane; it < 16 * C4; it += 64) {
            const int row = it / C4, c4 = it - row * C4;
            const int m = mt * 16 + row, n = n0 + c4 * 4;
            if (m < p.M && n < p.N) emit_out4(p, m, n, *reinterpret_cast<const f32x4*>(cw + row * LDC + c4 * 4));
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0) only: LDS traffic of this wave is done; global loads / stores stay in flight
        __builtin_amdgcn_wave_barrier();
        mt += mt_stride;
    }
}

bool conv_pw_supported(const ConvParams& p) {
    return p.KH == 1 && p.KW == 1 && p.stride == 1 && p.pad_t == 0 && p.pad_l == 0 && p.OH == p.H && p.OW == p.W && !p.in_split &&
           p.Kpad <= PW_KMAX && p.Cin % 4 == 0 && p.in_ld % 4 == 0 && (p.mode == MODE_CONV || p.mode == MODE_DECONV);
}

template <int TN, bool GATE>
static void launch_pw2(const ConvParams& p, hipStream_t s) {
    constexpr int BN = 16 * TN;
    const size_t lds = ((size_t)BN * (p.Kpad + 4) + 4 * 16 * (BN + 4)) * sizeof(float);
    static LdsAttr attr;
    auto kern = conv_pw_kernel<TN, GATE>;
    ensure_dynamic_lds(attr, reinterpret_cast<const void*>(kern), lds);
    const int mtiles = (p.M + 15) / 16;
    const int nblocks = (p.N + BN - 1) / BN;
    const int wgs_needed = (mtiles + 3) / 4;
    const int per_cu = std::max(1, std::min(5, (int)(150 * 1024 / lds)));            // workgroups that fit a CU's LDS
    const int gx = std::max(1, std::min(wgs_needed, (256 * per_cu + nblocks - 1) / nblocks));
    CCVPE_LAUNCH(kern, dim3(gx, nblocks), dim3(256), lds, s, p);
}
template <int TN>
static void launch_pw(const ConvParams& p, hipStream_t s) {
    if (p.gate) launch_pw2<TN, true>(p, s);
    else launch_pw2<TN, false>(p, s);
}

// bm = 16: rows are padded to one MFMA tile only; bn = 16 TN
template <int RT>
static void launch_proj_rt(const ConvParams& p, hipStream_t s) { launch_proj(p, RT, s); }
static const PwTile PW_TILES[] = {
    {16, 16, "conv_pw_16", launch_pw<1>, 0},
    {16, 32, "conv_pw_32", launch_pw<2>, 0},
    {16, 48, "conv_pw_48", launch_pw<3>, 0},
    {16, 80, "conv_pw_80", launch_pw<5>, 0},
    {16, 128, "conv_pw_128", launch_pw<8>, 0},
    {16, 64, "conv_pw_64", launch_pw<4>, 0},
    {16, 160, "conv_pw_160", launch_pw<10>, 0},
    // kernels_proj.hip: RT x 16 rows x ALL columns per workgroup, K split over its four waves (gated project convs with a deep K)
    {16, 16, "conv_proj_r1", launch_proj_rt<1>, 1},
    {32, 16, "conv_proj_r2", launch_proj_rt<2>, 2},
    {64, 16, "conv_proj_r4", launch_proj_rt<4>, 4},
    // ... and its latency form: one row tile x 1 / 2 / 4 column tiles per workgroup of sixteen waves that split K (batch <= 4)
    {16, 16, "conv_projl_1", launch_proj_rt<101>, 101},
    {16, 32, "conv_projl_2", launch_proj_rt<102>, 102},
    {16, 64, "conv_projl_4", launch_proj_rt<104>, 104},
    // ... with two / four row tiles and one column tile per workgroup: the weights of a 64-row layer are read once (layers without a gate)
    {32, 16, "conv_projl_r2", launch_proj_rt<121>, 121},
    {64, 16, "conv_projl_r4", launch_proj_rt<141>, 141},
};
int conv_pw_tile_proj_rt(int i) { return i >= 0 && i < (int)(sizeof(PW_TILES) / sizeof(PW_TILES[0])) ? PW_TILES[i].proj_rt : 0; }
int pw_num_tiles() { return (int)(sizeof(PW_TILES) / sizeof(PW_TILES[0])); }
const PwTile* pw_tile(int i) { return &PW_TILES[i]; }
bool conv_pw_tile_ok(int i, const ConvParams& p) {
    if (i < 0 || i >= pw_num_tiles()) return false;
    if (PW_TILES[i].proj_rt > 0) return conv_proj_supported(p, PW_TILES[i].proj_rt);
    return conv_pw_supported(p) && conv_pw_fits(PW_TILES[i].bn, p.Kpad);
}
// the weight slab [bn][Kpad + 4] and the four C patches must fit the LDS of a CU
bool conv_pw_fits(int bn, int kpad) { return kpad <= PW_KMAX && ((size_t)bn * (kpad + 4) + 4 * 16 * (bn + 4)) * sizeof(float) <= 150 * 1024; }

}  // namespace ccvpe
