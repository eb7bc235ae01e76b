// Fused MBConv front half for the SMALL-spatial blocks (6..15 of EfficientNet-B0: 32x32 ... 5x8 pixels, 480-1152
// expanded channels): 1x1 expand (+BN+swish) -> depthwise k x k (+BN+swish) -> SE pooling sums, with the expanded
// activation of one whole image x 16 channels resident in LDS.
//
// Reference: MBConvBlock.forward, efficientnet_pytorch/model.py:103-110 (expand conv + bn0 + swish, static same /
// horizontally circular padding utils.py:254-358, depthwise conv + bn1 + swish) and :114 (global average pool).
//
// Unfused, these blocks write the 6x-expanded tensor to HBM and read it straight back (block 9, batch 32: 88 MB each
// way, two launches of 75 + 60 us).  The tile form of kernels_mbconv.hip does not pay here: an 8x8 tile of a 5x5 layer
// recomputes 2.25x of the expand GEMM.  At these resolutions the WHOLE image of a 16-channel chunk fits in LDS
// (36 x 36 pixels x 80 B = 104 KB), so nothing is recomputed and there is no halo logic on the input at all:
//   * a 512-thread workgroup owns (sample, every CG-th chunk of 16 expanded channels);
//   * expand: wave w takes m-tiles w, w + 8, ... of the [pixels x Cin] x [Cin x 16] GEMM, two at a time; the A operand
//     (the block input, NHWC) goes from global memory / L2 straight into the MFMA registers, the B operand (16 rows of
//     the expand weight) sits in registers for the chunk; bias + swish, then the accumulator rows are scattered into the
//     zero-padded (or horizontally wrapped) E image through a pixel -> offset table;
//   * depthwise: thread = (4 channels, TX x TY output patch); the K x K taps come from LDS once per chunk, every input
//     row of the patch window is read once and used by all taps; bias + swish, 16-byte stores, channel sums for the
//     squeeze-excite pool reduced in a fixed order (wave shuffles, then 8 partials through LDS) -> deterministic;
//   * two barriers per chunk; the next chunk's expand weights are requested before the depthwise phase starts.
// Samples are pinned to XCDs (sample b only on workgroups with id % 8 == b % 8) so the ~40-70 passes over a sample's
// input hit that XCD's L2.
#include "igemm_common.h"

#include <algorithm>
#include <cstdlib>

namespace ccvpe {

static constexpr int EPS = 20;        // floats per pixel of the E image (16 + 4 pad: the MFMA-layout scatter is conflict-free)

__device__ __forceinline__ float swish_i(float v) { return v * __builtin_amdgcn_rcpf(1.f + __expf(-v)); }

struct MbImgParams {
    MbFrontParams f;
    int NMT;                   // m-tiles of 16 pixels a strip (with its halo rows) needs at most
    int NST, RO;               // strips per image, output rows per strip (one strip = the whole image for the small blocks)
    int PT, PL;                // rows / columns of padding in front of the image inside the E image
    int WPa, HPa;              // allocated E image size in pixels (>= the static padding; >= the patch windows of ragged sizes)
    int NPX, NPY;              // output patches per row / column
    int CG;                    // unused
    int nchunks;               // mid / 16
};

// NT = 512 threads (8 waves, one workgroup per CU) for the large images, 256 (two workgroups per CU, out of phase) when two
// E images fit the LDS.  (Measured and dropped: keeping the input m-tiles of a <= 256-pixel image in registers across all
// chunks - the expand and depthwise phases of a chunk simply add up, 55 + 40 us on block 12, and one workgroup per CU has
// nothing to overlap them with.)
template <int K, int S, int KCH, int TX, int TY, int NT>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(2))) void mbconv_image_kernel(const MbImgParams q) {
    constexpr int NWV = NT / 64;
    constexpr int WW = (TX - 1) * S + K;
    const MbFrontParams& p = q.f;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int e_floats = (q.HPa * q.WPa + 1) * EPS;      // + one sink pixel
    float* Es = smem;
    int* ptab = reinterpret_cast<int*>(smem + e_floats);  // [NMT*16] pixel -> float offset in Es (sink past P)
    int* dtab = ptab + q.NMT * 16;                         // [NMT*16] circular layers: offset of the wrapped copy (sink if none)
    float* wks = reinterpret_cast<float*>(dtab + q.NMT * 16);   // [K*K][16] depthwise taps of the chunk
    float* bds = wks + K * K * 16;                         // [16] depthwise bias
    float* red = bds + 16;                                 // [NT/64][16] per-wave pooling sums

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // Work = the flattened list of (unit, chunk) items, unit = (sample, strip), unit-major; a workgroup takes an equal
    // contiguous share, and the shares are dealt XCD-major (XCD x owns a contiguous eighth of the list): a sample's 40-70
    // passes over its input stay in ONE L2 (dealt round robin, every L2 saw every sample: block 9 0.091 -> 0.134 ms)
    const int total = p.B * q.NST * q.nchunks;
    const int wgx = xcd_remap(blockIdx.x, gridDim.x);
    const int it_lo = (int)((long long)total * wgx / gridDim.x), it_hi = (int)((long long)total * (wgx + 1) / gridDim.x);
    if (it_lo >= it_hi) return;
    const int sink = q.HPa * q.WPa * EPS;
    const unsigned a_lane = (unsigned)(((lane & 15) * p.Cin + 4 * (lane >> 4)) * 4);    // row (lane & 15) of an m-tile, channels 4 (lane >> 4)
    const unsigned a_mt = (unsigned)(16 * p.Cin * 4);                                    // bytes per m-tile
    const int dq4 = tid & 3, slot = tid >> 2;          // depthwise: channel quad, patch slot (NT / 4 slots)
    const int npatch = q.NPX * q.NPY;

    f32x4 wf[KCH];                                      // B fragments: We[16 ch + (lane & 15)][16 kc + 4 (lane >> 4) + e]
#define CCVPE_MI_LOAD_W(c_)                                                                                      \
    _Pragma("unroll") for (int kc = 0; kc < KCH; ++kc)                                                           \
        wf[kc] = *reinterpret_cast<const f32x4*>(p.we + (size_t)((c_) * 16 + (lane & 15)) * p.cinp + kc * 16 + 4 * (lane >> 4));
    // depthwise taps / bias of a chunk: fetched into registers a phase ahead, parked in LDS after the chunk barrier
    const bool tap_thread = tid < K * K * 4, bias_thread = tid >= NT - 16;
    f32x4 tapv = {0.f, 0.f, 0.f, 0.f};
    float biasv = 0.f;
#define CCVPE_MI_LOAD_TAPS(c_)                                                                                   \
    {                                                                                                            \
        if (tap_thread) tapv = *reinterpret_cast<const f32x4*>(p.wd + (size_t)(tid >> 2) * p.mid + (c_) * 16 + (tid & 3) * 4); \
        if (bias_thread) biasv = p.bd[(c_) * 16 + tid - (NT - 16)];                                               \
    }
    // A operand of one m-tile: KCH 16-byte pieces per lane; two register sets, the next m-tile always in flight
    f32x4 abuf[2][KCH];
#define CCVPE_MI_LOAD_A(set_, mt_)                                                                               \
    _Pragma("unroll") for (int kc = 0; kc < KCH; ++kc)                                                           \
        abuf[set_][kc] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, (mt_) < NMT ? a_lane + (unsigned)(mt_) * a_mt : 0x80000000u, kc * 64, 0));
    // one m-tile: 4 KCH dependent MFMAs (the SIMD's other wave fills the dependent-issue gaps; the expand weights are the A
    // operand, so a lane ends up with 4 consecutive channels of ONE pixel), bias + swish, one 16-byte scatter per lane
#define CCVPE_MI_SCATTER(acc_, mt_)                                                                              \
    {                                                                                                            \
        const int pos = ptab[(mt_) * 16 + (lane & 15)];                                                          \
        f32x4 v;                                                                                                 \
        _Pragma("unroll") for (int r = 0; r < 4; ++r) v[r] = swish_i(acc_[r] + be[r]);                           \
        *reinterpret_cast<f32x4*>(Es + pos + 4 * (lane >> 4)) = v;                                               \
        if (p.circular) *reinterpret_cast<f32x4*>(Es + dtab[(mt_) * 16 + (lane & 15)] + 4 * (lane >> 4)) = v;   \
    }
#define CCVPE_MI_MTILE(set_, mt_)                                                                                \
    {                                                                                                            \
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};                                                                        \
        _Pragma("unroll") for (int kc = 0; kc < KCH; ++kc)                                                       \
            _Pragma("unroll") for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[kc][e], abuf[set_][kc][e], acc, 0, 0, 0); \
        CCVPE_MI_SCATTER(acc, mt_);                                                                              \
    }

    // state of the current unit (set at its first item)
    int unit = -1, b = 0, oyA = 0, oyB = 0, NMT = 0;
    __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, 0, 0x00020000);

    for (int it = it_lo; it < it_hi; ++it) {
        const int u = it / q.nchunks, ch = it - u * q.nchunks;
        if (u != unit) {
            // ---- new (sample, strip): output rows [oyA, oyB), input rows [yA, yB) (clamped to the image: rows outside stay
            // zero in the E image); E row 0 is input row ey0 (negative in the first strip: the static top padding) ----
            unit = u;
            b = u / q.NST;
            const int st = u - b * q.NST;
            oyA = st * q.RO; oyB = min(oyA + q.RO, p.OH);
            const int ey0 = oyA * S - q.PT;
            const int yA = max(ey0, 0), yB = min((oyB - 1) * S - q.PT + K, p.H);
            const int P = (yB - yA) * p.W;
            NMT = (P + 15) >> 4;                         // <= q.NMT
            x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x + ((size_t)b * p.H + yA) * p.W * p.Cin), 0, (unsigned)((size_t)P * p.Cin * 4), 0x00020000);
            // (the barrier that ended the previous item already guarantees nobody still reads the E image or the tables)
            for (int i = tid * 4; i < e_floats; i += NT * 4) *reinterpret_cast<f32x4*>(Es + i) = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int i = tid; i < q.NMT * 16; i += NT) {
                int po = sink, dq = sink;
                if (i < P) {
                    const int yl = i / p.W, x = i - yl * p.W;
                    const int y = yA + yl - ey0;           // row inside the E image
                    po = (y * q.WPa + x + q.PL) * EPS;
                    if (p.circular) {
                        const int pr = (p.OW - 1) * S + K - q.PL - p.W;       // columns of padding behind the image
                        if (x >= p.W - q.PL) dq = (y * q.WPa + x - (p.W - q.PL)) * EPS;
                        else if (x < pr) dq = (y * q.WPa + q.PL + p.W + x) * EPS;
                    }
                }
                ptab[i] = po; dtab[i] = dq;
            }
            CCVPE_MI_LOAD_W(ch);
            CCVPE_MI_LOAD_TAPS(ch);
            CCVPE_MI_LOAD_A(0, wave);
            __syncthreads();
        }
        const int ch0 = ch * 16;
        // depthwise taps and bias of this chunk -> LDS (read after the barrier below)
        if (tap_thread) *reinterpret_cast<f32x4*>(wks + (tid >> 2) * 16 + (tid & 3) * 4) = tapv;
        if (bias_thread) bds[tid - (NT - 16)] = biasv;
        const f32x4 be = *reinterpret_cast<const f32x4*>(p.be + ch0 + 4 * (lane >> 4));   // channel-major accumulators: 4 channels of one pixel per lane

        // ---- expand: m-tiles wave, wave + NWV, ...; set 0 holds the first one already ----
        for (int mt = wave; mt < NMT;) {
            CCVPE_MI_LOAD_A(1, mt + NWV);
            CCVPE_MI_MTILE(0, mt);
            mt += NWV;
            if (mt >= NMT) break;
            CCVPE_MI_LOAD_A(0, mt + NWV);
            CCVPE_MI_MTILE(1, mt);
            mt += NWV;
        }
        // the next item's weights, taps and (same unit) first m-tile land under the depthwise phase
        if (it + 1 < it_hi) {
            const int un = (it + 1) / q.nchunks, chn = it + 1 - un * q.nchunks;
            if (un == unit) { CCVPE_MI_LOAD_W(chn); CCVPE_MI_LOAD_TAPS(chn); CCVPE_MI_LOAD_A(0, wave); }
        }
        __syncthreads();                                   // E image, taps and bias of this chunk complete

        // ---- depthwise K x K, stride S, from the E image ----
        const f32x4 bd = *reinterpret_cast<const f32x4*>(bds + dq4 * 4);
        f32x4 pool = {0.f, 0.f, 0.f, 0.f};
        for (int pi = slot; pi < npatch; pi += NT / 4) {
            const int py = pi / q.NPX, px = pi - py * q.NPX;
            const int oy0 = py * TY, ox0 = px * TX;      // output row inside the strip
            const float* wp = Es + ((oy0 * S) * q.WPa + ox0 * S) * EPS + dq4 * 4;
            f32x4 acc[TY][TX];
#pragma unroll
            for (int ty = 0; ty < TY; ++ty)
#pragma unroll
                for (int tx = 0; tx < TX; ++tx) acc[ty][tx] = bd;
            // one tap row at a time (a real loop: with all K*K taps and the whole window in registers the 5x5 forms spill, and
            // hipcc hoists every LDS read of an unrolled form to the top whatever fences are placed)
#pragma unroll 1
            for (int ky = 0; ky < K; ++ky) {
                f32x4 tap[K];
#pragma unroll
                for (int kx = 0; kx < K; ++kx) tap[kx] = *reinterpret_cast<const f32x4*>(wks + (ky * K + kx) * 16 + dq4 * 4);
#pragma unroll
                for (int ty = 0; ty < TY; ++ty) {
                    const float* rp = wp + ((ty * S + ky) * q.WPa) * EPS;
                    f32x4 row[WW];
#pragma unroll
                    for (int wx = 0; wx < WW; ++wx) row[wx] = *reinterpret_cast<const f32x4*>(rp + wx * EPS);
#pragma unroll
                    for (int tx = 0; tx < TX; ++tx)
#pragma unroll
                        for (int kx = 0; kx < K; ++kx) acc[ty][tx] = __builtin_elementwise_fma(row[tx * S + kx], tap[kx], acc[ty][tx]);
                }
            }
#pragma unroll
            for (int ty = 0; ty < TY; ++ty)
#pragma unroll
                for (int tx = 0; tx < TX; ++tx) {
                    const int oy = oyA + oy0 + ty, ox = ox0 + tx;
                    if (oy < oyB && ox < p.OW) {
                        f32x4 ov;
#pragma unroll
                        for (int e = 0; e < 4; ++e) ov[e] = swish_i(acc[ty][tx][e]);
                        pool += ov;
                        *reinterpret_cast<f32x4*>(p.out + (((size_t)b * p.OH + oy) * p.OW + ox) * p.mid + ch0 + dq4 * 4) = ov;
                    }
                }
        }
        // channel sums: lanes with equal quad inside the wave (xor over lane bits 2..5), then the waves through LDS
#pragma unroll
        for (int off = 4; off < 64; off <<= 1)
#pragma unroll
            for (int e = 0; e < 4; ++e) pool[e] += __shfl_xor(pool[e], off);
        if (lane < 4) *reinterpret_cast<f32x4*>(red + wave * 16 + lane * 4) = pool;
        __syncthreads();                                   // E image free again; wave partials visible
        if (tid < 16) {
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < NT / 64; ++w) s += red[w * 16 + tid];
            p.pool[(size_t)unit * p.mid + ch0 + tid] = s;  // [B][NST][mid]: one partial row per strip
        }
        // (red is rewritten only after the next item's first barrier, which wave 0 reaches after these reads)
    }
#undef CCVPE_MI_LOAD_W
#undef CCVPE_MI_LOAD_TAPS
#undef CCVPE_MI_LOAD_A
#undef CCVPE_MI_MTILE
#undef CCVPE_MI_SCATTER
}

// Geometry for RO output rows per strip (RO a multiple of ty, or the whole image).
static size_t img_geometry_ro(const MbFrontParams& p, int tx, int ty, int ro, MbImgParams& q) {
    q.f = p;
    q.RO = ro;
    q.NST = (p.OH + ro - 1) / ro;
    q.PT = p.pad_t; q.PL = p.pad_l;
    q.NPX = (p.OW + tx - 1) / tx; q.NPY = (ro + ty - 1) / ty;
    q.WPa = std::max(p.W + q.PL, (q.NPX * tx - 1) * p.s + p.k);
    q.HPa = (q.NPY * ty - 1) * p.s + p.k;                          // rows the patch windows of one strip reach
    const int rows_in = std::min(p.H, (ro - 1) * p.s + p.k);         // input rows of a strip incl. halo
    q.NMT = (rows_in * p.W + 15) / 16;
    q.nchunks = p.mid / 16;
    return ((size_t)(q.HPa * q.WPa + 1) * EPS + 2 * (size_t)q.NMT * 16 + (size_t)p.k * p.k * 16 + 16 + 8 * 16) * sizeof(float);
}

// Whole image if it fits the LDS of a CU, else the fewest equal strips that do (their halo rows are expanded twice).
static bool img_geometry(const MbFrontParams& p, int tx, int ty, MbImgParams& q, size_t& lds) {
    const size_t cap = 150 * 1024;
    for (int nst = 1; nst <= p.OH; ++nst) {
        int ro = (p.OH + nst - 1) / nst;
        ro = (ro + ty - 1) / ty * ty;
        lds = img_geometry_ro(p, tx, ty, ro, q);
        if (lds <= cap) return (ro - 1) * p.s + p.k <= 4 * ro * p.s;   // not worth it once the halo is 4x the strip
        if (ro <= ty) break;
    }
    return false;
}

// Patch per thread: 4 x 2 outputs (13 LDS reads per output for a 5x5 layer) - also on a 16 x 16 image, where it keeps only half
// of a 256-thread workgroup busy but halves the LDS traffic of the 2 x 1 form (27 reads per output); 2 x 1 for the small
// stride-2 images.
static void patch_sel(int oh, int ow, int s, int& tx, int& ty) {
    if (oh * ow >= 512 || s == 1) { tx = 4; ty = 2; } else { tx = 2; ty = 1; }
}
// Cin need not be a multiple of 16: the expand weights are zero padded to cinp and the last 16-byte pieces of a pixel then
// read the first channels of the next pixel (finite activations; past the end the buffer descriptor returns zeros).
bool mbconv_image_supported(const MbFrontParams& p) {
    const int kch = p.cinp / 16;
    if (p.Cin % 8 != 0 || p.cinp % 16 != 0 || p.cinp < p.Cin || p.mid % 16 != 0 || p.W < 2 * p.k) return false;
    const bool combo = (p.k == 3 && p.s == 1 && (kch == 2 || kch == 5 || kch == 12)) || (p.k == 5 && p.s == 1 && (kch == 3 || kch == 5 || kch == 7 || kch == 12)) ||
                       (p.k == 5 && p.s == 2 && (kch == 2 || kch == 7)) || (p.k == 3 && p.s == 2 && kch == 3);
    if (!combo) return false;
    int tx, ty;
    patch_sel(p.OH, p.OW, p.s, tx, ty);
    MbImgParams q;
    size_t lds;
    if (!img_geometry(p, tx, ty, q, lds)) return false;
    // measured (batch 32): a stride-2 layer wider than 128 pixels gets strips of only 4 output rows (1.4x halo rows, items too
    // short for their two barriers): ground block 3 (80 x 160) 0.195 ms fused against 0.182 ms as two launches - left unfused
    if (q.NST > 1 && p.s == 2 && p.W > 128) return false;
    return true;
}

int mbconv_image_strips(const MbFrontParams& p) {
    int tx, ty;
    patch_sel(p.OH, p.OW, p.s, tx, ty);
    MbImgParams q;
    size_t lds;
    return (mbconv_image_supported(p) && img_geometry(p, tx, ty, q, lds)) ? q.NST : 0;
}

template <int K, int S, int KCH, int TX, int TY, int NT>
static void launch_img(const MbFrontParams& p, hipStream_t s) {
    MbImgParams q;
    size_t lds;
    img_geometry(p, TX, TY, q, lds);
    static LdsAttr attr;
    auto kern = mbconv_image_kernel<K, S, KCH, TX, TY, NT>;
    ensure_dynamic_lds(attr, reinterpret_cast<const void*>(kern), lds);
    const int total = std::max(1, p.B * q.NST * q.nchunks);
    q.CG = 0;
    hipLaunchKernelGGL(kern, dim3(std::min(total, NT == 512 ? 256 : 512)), dim3(NT), lds, s, q);   // persistent: 8 waves per CU either way
}

template <int K, int S, int KCH>
static void launch_img_p(const MbFrontParams& p, hipStream_t s) {
    int tx, ty;
    patch_sel(p.OH, p.OW, p.s, tx, ty);
    MbImgParams q;
    size_t lds;
    img_geometry(p, tx, ty, q, lds);
    const bool two = 2 * lds <= 150 * 1024;      // two workgroups of 256 threads per CU
    if (tx == 4) { if (two) launch_img<K, S, KCH, 4, 2, 256>(p, s); else launch_img<K, S, KCH, 4, 2, 512>(p, s); }
    else { if (two) launch_img<K, S, KCH, 2, 1, 256>(p, s); else launch_img<K, S, KCH, 2, 1, 512>(p, s); }
}

void launch_mbconv_image(const MbFrontParams& p, hipStream_t s) {
    const int kch = p.cinp / 16;
    if (p.k == 3 && p.s == 1) { if (kch == 2) launch_img_p<3, 1, 2>(p, s); else if (kch == 5) launch_img_p<3, 1, 5>(p, s); else launch_img_p<3, 1, 12>(p, s); }
    else if (p.k == 5 && p.s == 1) { if (kch == 3) launch_img_p<5, 1, 3>(p, s); else if (kch == 5) launch_img_p<5, 1, 5>(p, s); else if (kch == 7) launch_img_p<5, 1, 7>(p, s); else launch_img_p<5, 1, 12>(p, s); }
    else if (p.k == 5) { if (kch == 2) launch_img_p<5, 2, 2>(p, s); else launch_img_p<5, 2, 7>(p, s); }
    else launch_img_p<3, 2, 3>(p, s);
}

}  // namespace ccvpe
