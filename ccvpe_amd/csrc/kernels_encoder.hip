// HBM-bound encoder kernels: stem conv, depthwise conv (+BN+swish+SE pooling), squeeze-excite MLP,
// ground-descriptor row reduction.  fp32, NHWC, float4 (16 B/lane) accesses, BN pre-folded.
//
// Reference call sites: efficientnet_pytorch/model.py:289 (stem), :108-110 (depthwise + bn1 + swish),
// :113-118 (squeeze-excite), utils.py:254-358 (static zero / horizontal-circular padding),
// models.py:355-395 (ground descriptor heads: permute + Conv2d(H_f,1,1) + flatten).
#include "kernels.h"

#include <algorithm>
#include <cstdlib>

namespace ccvpe {

// x * sigmoid(x) with the hardware reciprocal (v_rcp_f32, 1 ulp) instead of an IEEE division: the division expands to ~10
// instructions (v_div_scale / v_div_fmas / v_div_fixup + Newton steps) per activation, and the encoder applies ~10^9 of them
__device__ __forceinline__ float swishf(float v) { return v * __builtin_amdgcn_rcpf(1.f + __expf(-v)); }

// ------------------------------------------------------------------------------------------------
// Stem: 3x3 stride 2, 3 -> 32 channels, NCHW input -> NHWC output, BN + swish.
// thread = (output pixel, group of 4 output channels); 8 consecutive lanes share a pixel so the
// 27 input taps are broadcast loads and the store is a fully coalesced 16 B/lane stream.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void stem_kernel(const StemParams p) {
    // The grid stride is a multiple of 8, so a thread keeps its channel group for the whole launch and
    // holds its 27 x 4 weights in registers (the LDS broadcast reads were the bottleneck before).
    const int cg = threadIdx.x & 7;
    float4 w[27];
#pragma unroll
    for (int i = 0; i < 27; ++i) w[i] = *reinterpret_cast<const float4*>(p.w + i * 32 + cg * 4);
    const float4 bias = *reinterpret_cast<const float4*>(p.bias + cg * 4);
    const long long total = (long long)p.B * p.OH * p.OW * 8;
    for (long long it = (long long)blockIdx.x * 256 + threadIdx.x; it < total; it += (long long)gridDim.x * 256) {
        long long pix = it >> 3;
        const int ox = (int)(pix % p.OW);
        long long t = pix / p.OW;
        const int oy = (int)(t % p.OH);
        const int b = (int)(t / p.OH);
        float4 acc = bias;
        const float* inb = p.in + (size_t)b * 3 * p.H * p.W;
        const int plane = p.H * p.W;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = oy * 2 - p.pad_t + ky;
            const bool oky = (unsigned)iy < (unsigned)p.H;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                int ix = ox * 2 - p.pad_l + kx;
                bool ok = oky;
                if (p.circular) {
                    if (ix < 0) ix += p.W;
                    if (ix >= p.W) ix -= p.W;
                } else {
                    ok = ok & ((unsigned)ix < (unsigned)p.W);
                }
                const int o = ok ? iy * p.W + ix : 0;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    float v = inb[c * plane + o];
                    v = ok ? v : 0.f;
                    const float4 wv = w[(c * 3 + ky) * 3 + kx];
                    acc.x = fmaf(v, wv.x, acc.x);
                    acc.y = fmaf(v, wv.y, acc.y);
                    acc.z = fmaf(v, wv.z, acc.z);
                    acc.w = fmaf(v, wv.w, acc.w);
                }
            }
        }
        acc.x = swishf(acc.x); acc.y = swishf(acc.y); acc.z = swishf(acc.z); acc.w = swishf(acc.w);
        *reinterpret_cast<float4*>(p.out + pix * 32 + cg * 4) = acc;
    }
}

// Stem, tile form (default): a workgroup owns 32 x 8 output pixels x all 32 channels.  The (65 x 17 x 3)-float input patch is
// staged in LDS once (coalesced rows; zero / wrapped outside the image), so the 27 taps of an output are LDS reads instead
// of 27 global loads each touching 8 addresses (the pixel form above sat at ~2 TB/s on load issue).  Thread = (channel quad,
// column of the tile): its 8 output rows share input rows (17 x 9 = 153 reads for 8 outputs), the 27 x 4 weights stay in
// registers, stores are 16 B per lane with the 8 quads of a pixel contiguous.  Persistent grid.
typedef float f32x4s __attribute__((ext_vector_type(4)));
static constexpr int ST_TW = 32, ST_TH = 8;
static constexpr int ST_PW = 2 * ST_TW + 1, ST_PH = 2 * ST_TH + 1;   // input patch
static constexpr int ST_PITCH = ST_PW + 2;                            // floats per patch row

__global__ __launch_bounds__(256) void stem_tile_kernel(const StemParams p) {
    __shared__ float patch[3 * ST_PH * ST_PITCH];
    const int tid = threadIdx.x;
    const int cg = tid & 7, col = tid >> 3;                           // channel quad, tile column
    f32x4s w[27];
#pragma unroll
    for (int i = 0; i < 27; ++i) w[i] = *reinterpret_cast<const f32x4s*>(p.w + i * 32 + cg * 4);
    const f32x4s bias = *reinterpret_cast<const f32x4s*>(p.bias + cg * 4);
    const int tiles_x = (p.OW + ST_TW - 1) / ST_TW, tiles_y = (p.OH + ST_TH - 1) / ST_TH;
    const int tiles = p.B * tiles_x * tiles_y;
    const int plane = p.H * p.W;
    // the next tile's patch is fetched into registers while this one is computed (NPRE values per thread)
    constexpr int NPRE = (3 * ST_PH * ST_PW + 255) / 256;
    float pre[NPRE];
    auto fetch = [&](int tl) {
        const int b = tl / (tiles_x * tiles_y);
        const int r = tl - b * (tiles_x * tiles_y);
        const int ty = r / tiles_x, tx = r - ty * tiles_x;
        const int iy0 = ty * ST_TH * 2 - p.pad_t, ix0 = tx * ST_TW * 2 - p.pad_l;
        const float* inb = p.in + (size_t)b * 3 * plane;
#pragma unroll
        for (int j = 0; j < NPRE; ++j) {
            const int i = tid + j * 256;
            const int c = i / (ST_PH * ST_PW), rem = i - c * (ST_PH * ST_PW);
            const int y = rem / ST_PW, x = rem - y * ST_PW;
            const int iy = iy0 + y;
            int ix = ix0 + x;
            bool ok = i < 3 * ST_PH * ST_PW && (unsigned)iy < (unsigned)p.H;
            if (p.circular) { if (ix < 0) ix += p.W; else if (ix >= p.W) ix -= p.W; }
            ok = ok && (unsigned)ix < (unsigned)p.W;
            pre[j] = ok ? inb[c * plane + iy * p.W + ix] : 0.f;
        }
    };
    if ((int)blockIdx.x < tiles) fetch(blockIdx.x);
    for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int b = tile / (tiles_x * tiles_y);
        const int r = tile - b * (tiles_x * tiles_y);
        const int ty = r / tiles_x, tx = r - ty * tiles_x;
        const int oy0 = ty * ST_TH, ox0 = tx * ST_TW;
        __syncthreads();                                              // previous tile's readers are done with the patch
#pragma unroll
        for (int j = 0; j < NPRE; ++j) {
            const int i = tid + j * 256;
            if (i < 3 * ST_PH * ST_PW) {
                const int c = i / (ST_PH * ST_PW), rem = i - c * (ST_PH * ST_PW);
                const int y = rem / ST_PW, x = rem - y * ST_PW;
                patch[(c * ST_PH + y) * ST_PITCH + x] = pre[j];
            }
        }
        __syncthreads();
        if (tile + (int)gridDim.x < tiles) fetch(tile + gridDim.x);
        f32x4s acc[ST_TH];
#pragma unroll
        for (int i = 0; i < ST_TH; ++i) acc[i] = bias;
        const float* pp = patch + 2 * col;
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int y = 0; y < ST_PH; ++y) {
                const float v0 = pp[(c * ST_PH + y) * ST_PITCH], v1 = pp[(c * ST_PH + y) * ST_PITCH + 1], v2 = pp[(c * ST_PH + y) * ST_PITCH + 2];
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    if ((y - ky) < 0 || ((y - ky) & 1) || (y - ky) / 2 >= ST_TH) continue;
                    const int i = (y - ky) / 2;                       // output row fed by patch row y through tap row ky
                    // 4 channels per tap on two v_pk_fma_f32 (the input value is broadcast through op_sel_hi, no move)
                    acc[i] = __builtin_elementwise_fma(f32x4s{v0, v0, v0, v0}, w[(c * 3 + ky) * 3], acc[i]);
                    acc[i] = __builtin_elementwise_fma(f32x4s{v1, v1, v1, v1}, w[(c * 3 + ky) * 3 + 1], acc[i]);
                    acc[i] = __builtin_elementwise_fma(f32x4s{v2, v2, v2, v2}, w[(c * 3 + ky) * 3 + 2], acc[i]);
                }
            }
        const int ox = ox0 + col;
        if (ox < p.OW) {
#pragma unroll
            for (int i = 0; i < ST_TH; ++i) {
                const int oy = oy0 + i;
                if (oy >= p.OH) break;
                f32x4s o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = swishf(acc[i][e]);
                *reinterpret_cast<f32x4s*>(p.out + (((size_t)b * p.OH + oy) * p.OW + ox) * 32 + cg * 4) = o;
            }
        }
    }
}

void launch_stem(const StemParams& p, hipStream_t s) {
    static const bool pixel_form = getenv("CCVPE_STEM_TILE") && std::atoi(getenv("CCVPE_STEM_TILE")) == 0;   // A/B switch
    if (!pixel_form) {
        const int tiles = p.B * ((p.OW + ST_TW - 1) / ST_TW) * ((p.OH + ST_TH - 1) / ST_TH);
        CCVPE_LAUNCH(stem_tile_kernel, dim3(std::min(tiles, 256 * 6)), dim3(256), 0, s, p);
        return;
    }
    long long total = (long long)p.B * p.OH * p.OW * 8;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 256 * 16) blocks = 256 * 16;
    CCVPE_LAUNCH(stem_kernel, dim3(blocks), dim3(256), 0, s, p);
}

// ------------------------------------------------------------------------------------------------
// Stem + the depthwise conv of block 0 (round 3).  Block 0 of EfficientNet-B0 has expand ratio 1 (model.py:103: no expand conv), so its
// 3x3 / stride-1 depthwise conv (model.py:108-110) reads the stem output (model.py:314: conv_stem + bn0 + swish) directly.  As two
// launches the 32-channel half-resolution tensor is written and read back (268 MB each way for the aerial encoder at batch 32); both are
// bound by that traffic (4.4 / 5.2 TB/s).  Here a workgroup owns 32 x 8 outputs of the depthwise conv: it computes the 34 x 10 stem
// pixels they depend on from a 69 x 21 x 3 input patch in LDS (the halo is recomputed: 1.33x of the stem's FMAs), keeps them in LDS
// (zero outside the stem output where the reference pads with zeros, wrapped columns for the circular encoder: the patch fetch wraps
// the INPUT column, which is the same thing since W = 2 OW), runs the depthwise taps from there and stores only its result plus one
// pooling partial row per tile.  Per output the FMA order of both convs is the one of the separate kernels (same bits up to the
// squeeze-excite pooling order).  Thread = (channel quad, column); the stem phase computes 10 rows per thread plus, for 160 threads, one
// pixel of the two halo columns 32 / 33; the next tile's patch arrives by LDS-DMA under the depthwise phase.
// ------------------------------------------------------------------------------------------------
static constexpr int SD_TW = 32, SD_TH = 8;
static constexpr int SD_SW = SD_TW + 2, SD_SH = SD_TH + 2;               // stem pixels a tile needs
static constexpr int SD_PW = 2 * SD_SW + 1, SD_PH = 2 * SD_SH + 1;       // input patch
static constexpr int SD_PITCH = SD_PW + 2;

__global__ __launch_bounds__(256, 2) void stem_dw_kernel(const StemDwParams q) {
    typedef __attribute__((address_space(3))) void* lds_ptr;
    const StemParams& p = q.st;
    extern __shared__ __attribute__((aligned(16))) float sd_smem[];
    constexpr int PATCH = (3 * SD_PH * SD_PITCH + 3) & ~3;
    float* patch = sd_smem;                                           // [3][SD_PH][SD_PITCH]
    float* st = sd_smem + PATCH;                                      // [SD_SH][SD_SW][32] stem output (+ swish), halo included
    float* red = st + SD_SH * SD_SW * 32;                             // [4][32]
    float* wds = red + 4 * 32;                                        // [9][32] depthwise taps
    float* wst = wds + 9 * 32;                                        // [27][32] stem weights: 108 registers if held per thread (the kernel then spills);
                                                                      // from LDS the nine vectors of one input channel at a time
    const int tid = threadIdx.x;
    const int cg = tid & 7, col = tid >> 3;                           // channel quad, tile column
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const f32x4s bias = *reinterpret_cast<const f32x4s*>(p.bias + cg * 4);
    const f32x4s bd = *reinterpret_cast<const f32x4s*>(q.bd + cg * 4);
    for (int i = tid; i < 9 * 32; i += 256) wds[i] = q.wd[i];      // (visible after the first barrier of the tile loop)
    for (int i = tid; i < 27 * 32; i += 256) wst[i] = p.w[i];
    const int tiles_x = (p.OW + SD_TW - 1) / SD_TW, tiles_y = (p.OH + SD_TH - 1) / SD_TH;
    const int tps = tiles_x * tiles_y;                                // tiles per sample = pooling partial rows
    const int tiles = p.B * tps;
    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, (unsigned)((size_t)p.B * 3 * p.H * p.W * 4), 0x00020000);
    // The input patch of a tile arrives by LDS-DMA, requested when the previous tile's stem phase is over (the patch is read there only)
    // and landing under its depthwise phase: a (channel, row) of 69 floats is two instructions (lanes = columns 0..63 and 64..68); the
    // column of a lane is wrapped / bounds-checked once per tile, rows outside the image and columns outside a non-circular image get
    // the out-of-range offset (the descriptor returns zeros).  No registers held across the tile, no per-element index arithmetic.
    auto fetch = [&](int tl) {
        float* dst = patch;
        const int b = tl / tps;
        const int r = tl - b * tps;
        const int ty = r / tiles_x, tx = r - ty * tiles_x;
        const int iy0 = (ty * SD_TH - 1) * 2 - p.pad_t, ix0 = (tx * SD_TW - 1) * 2 - p.pad_l;
        unsigned xo[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            int ix = ix0 + h * 64 + lane;
            if (p.circular) { if (ix < 0) ix += p.W; else if (ix >= p.W) ix -= p.W; }
            const bool ok = h * 64 + lane < SD_PW && (unsigned)ix < (unsigned)p.W;
            xo[h] = ok ? (unsigned)ix * 4u : 0x80000000u;
        }
        for (int pr = wave; pr < 3 * SD_PH; pr += 4) {                // (channel, patch row) pairs of this wave
            const int c = pr / SD_PH, y = pr - c * SD_PH;
            const int iy = iy0 + y;
            const bool rok = (unsigned)iy < (unsigned)p.H;
            const unsigned rowb = (unsigned)((((size_t)b * 3 + c) * p.H + (rok ? iy : 0)) * p.W) * 4u;
            float* d = dst + pr * SD_PITCH;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(in_rsrc, (lds_ptr)d, 4, rok ? xo[0] : 0x80000000u, rowb, 0, 0);
            // (only the lanes that hold a column: an out-of-range lane still WRITES its zero, here into the next row of the patch)
            if (lane < SD_PW - 64) __builtin_amdgcn_raw_ptr_buffer_load_lds(in_rsrc, (lds_ptr)(d + 64), 4, rok ? xo[1] : 0x80000000u, rowb, 0, 0);
        }
    };
    // one stem pixel (row r, column c of the halo tile) from the patch: taps in the order of stem_tile_kernel (c, ky, kx ascending)
    auto stem_px = [&](const float* patch, int r, int c) {
        f32x4s a = bias;
        const float* pp = patch + 2 * c;
#pragma unroll 1
        for (int ch = 0; ch < 3; ++ch)
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const float* row = pp + (ch * SD_PH + 2 * r + ky) * SD_PITCH;
                const float* wr = wst + ((ch * 3 + ky) * 3) * 32 + cg * 4;
                a = __builtin_elementwise_fma(f32x4s{row[0], row[0], row[0], row[0]}, *reinterpret_cast<const f32x4s*>(wr), a);
                a = __builtin_elementwise_fma(f32x4s{row[1], row[1], row[1], row[1]}, *reinterpret_cast<const f32x4s*>(wr + 32), a);
                a = __builtin_elementwise_fma(f32x4s{row[2], row[2], row[2], row[2]}, *reinterpret_cast<const f32x4s*>(wr + 64), a);
            }
        return a;
    };
    // A workgroup owns a contiguous run of tiles (round 4; strided before): its tiles belong to one sample, or to two neighbours, so the
    // squeeze-excite ticket below is drawn once or twice per workgroup instead of once per tile
    const int t_lo = (int)((long long)tiles * blockIdx.x / gridDim.x), t_hi = (int)((long long)tiles * (blockIdx.x + 1) / gridDim.x);
    if (t_lo < t_hi) fetch(t_lo);
    for (int tile = t_lo; tile < t_hi; ++tile) {
        const int b = tile / tps;
        const int tr = tile - b * tps;
        const int ty = tr / tiles_x, tx = tr - ty * tiles_x;
        const int oy0 = ty * SD_TH, ox0 = tx * SD_TW;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // this wave's share of the patch has landed
        __syncthreads();                                              // ... everybody's; the previous tile's readers are done with the stem tile
        // ---- stem: rows 0..9 of halo column `col` (patch rows shared between the output rows they feed) ----
        {
            f32x4s acc[SD_SH];
#pragma unroll
            for (int i = 0; i < SD_SH; ++i) acc[i] = bias;
            const float* pp = patch + 2 * col;
#pragma unroll 1
            for (int c = 0; c < 3; ++c) {                             // (a real loop: one input channel's nine weight vectors in registers at a time)
                f32x4s w[9];
#pragma unroll
                for (int t = 0; t < 9; ++t) w[t] = *reinterpret_cast<const f32x4s*>(wst + (c * 9 + t) * 32 + cg * 4);
                const float* pc = pp + c * SD_PH * SD_PITCH;
#pragma unroll
                for (int y = 0; y < SD_PH; ++y) {
                    const float v0 = pc[y * SD_PITCH], v1 = pc[y * SD_PITCH + 1], v2 = pc[y * SD_PITCH + 2];
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky) {
                        if ((y - ky) < 0 || ((y - ky) & 1) || (y - ky) / 2 >= SD_SH) continue;
                        const int i = (y - ky) / 2;
                        acc[i] = __builtin_elementwise_fma(f32x4s{v0, v0, v0, v0}, w[ky * 3], acc[i]);
                        acc[i] = __builtin_elementwise_fma(f32x4s{v1, v1, v1, v1}, w[ky * 3 + 1], acc[i]);
                        acc[i] = __builtin_elementwise_fma(f32x4s{v2, v2, v2, v2}, w[ky * 3 + 2], acc[i]);
                    }
                }
            }
            const int sx = ox0 - 1 + col;
            const bool colok = p.circular || (unsigned)sx < (unsigned)p.OW;
#pragma unroll
            for (int i = 0; i < SD_SH; ++i) {
                const int sy = oy0 - 1 + i;
                const bool ok = colok && (unsigned)sy < (unsigned)p.OH;
                f32x4s o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = ok ? swishf(acc[i][e]) : 0.f;
                *reinterpret_cast<f32x4s*>(st + (i * SD_SW + col) * 32 + cg * 4) = o;
            }
        }
        if (tid < 8 * 2 * SD_SH) {                                    // the two halo columns past the 32 thread columns: one pixel per thread
            const int e2 = tid >> 3, r = e2 >> 1, c = SD_TW + (e2 & 1);
            const f32x4s a = stem_px(patch, r, c);
            const int sx = ox0 - 1 + c, sy = oy0 - 1 + r;
            const bool ok = (p.circular || (unsigned)sx < (unsigned)p.OW) && (unsigned)sy < (unsigned)p.OH;
            f32x4s o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = ok ? swishf(a[e]) : 0.f;
            *reinterpret_cast<f32x4s*>(st + (r * SD_SW + c) * 32 + cg * 4) = o;
        }
        __syncthreads();
        if (tile + 1 < t_hi) fetch(tile + 1);                         // the patch is free: the next tile's lands under the depthwise phase
        // ---- depthwise 3x3 from the stem tile: 8 outputs of column `col`; per output ky, kx ascending as in depthwise_kernel ----
        f32x4s pool = {0.f, 0.f, 0.f, 0.f};
        {
            f32x4s wd[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) wd[t] = *reinterpret_cast<const f32x4s*>(wds + t * 32 + cg * 4);
            f32x4s acc[SD_TH];
#pragma unroll
            for (int i = 0; i < SD_TH; ++i) acc[i] = bd;
#pragma unroll
            for (int r = 0; r < SD_SH; ++r) {
                f32x4s v[3];
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) v[kx] = *reinterpret_cast<const f32x4s*>(st + (r * SD_SW + col + kx) * 32 + cg * 4);
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const int i = r - ky;
                    if (i < 0 || i >= SD_TH) continue;
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) acc[i] = __builtin_elementwise_fma(v[kx], wd[ky * 3 + kx], acc[i]);   // (v_pk_fma_f32: per element the same fused operation)
                }
            }
            const int ox = ox0 + col;
            if (ox < p.OW) {
#pragma unroll
                for (int i = 0; i < SD_TH; ++i) {
                    const int oy = oy0 + i;
                    if (oy >= p.OH) break;
                    f32x4s o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = swishf(acc[i][e]);
                    pool += o;
                    *reinterpret_cast<f32x4s*>(q.out + (((size_t)b * p.OH + oy) * p.OW + ox) * 32 + cg * 4) = o;
                }
            }
        }
        // pooling partial of the tile: columns of a wave (lane bits 3..5), then the four waves through LDS; fixed order -> deterministic
#pragma unroll
        for (int off = 8; off < 64; off <<= 1)
#pragma unroll
            for (int e = 0; e < 4; ++e) pool[e] += __shfl_xor(pool[e], off);
        if (lane < 8) *reinterpret_cast<f32x4s*>(red + wave * 32 + lane * 4) = pool;
        __syncthreads();
        if (tid < 32) st_sc1(q.pool_partial + ((size_t)b * tps + tr) * 32 + tid, red[tid] + red[32 + tid] + red[64 + tid] + red[96 + tid]);
    }
    // squeeze-excite by ticket (ticket.h), behind the loop: one ticket per sample this workgroup's run of tiles touched; the workgroup that
    // completes a sample's tps tiles computes its gates (the stem tile `st` is free: scratch)
    if (q.se.counter != nullptr && t_lo < t_hi) {
        const int b_lo = t_lo / tps, b_hi = (t_hi - 1) / tps;
        for (int bb = b_lo; bb <= b_hi; ++bb) {
            const int n = min(t_hi, (bb + 1) * tps) - max(t_lo, bb * tps);
            if (ticket_arrive(q.se.counter + bb, (unsigned)n, (unsigned)q.se.per_sample, reinterpret_cast<unsigned*>(st + SE_SCR_FLAG))) {
                se_finish<256>(q.se, bb, st);
                __syncthreads();
            }
        }
    }
}

int stem_dw_tiles(int OH, int OW) { return ((OW + SD_TW - 1) / SD_TW) * ((OH + SD_TH - 1) / SD_TH); }

void launch_stem_dw(const StemDwParams& q, hipStream_t s) {
    const int tiles = q.st.B * stem_dw_tiles(q.st.OH, q.st.OW);
    const size_t lds = (size_t)(((3 * SD_PH * SD_PITCH + 3) & ~3) + SD_SH * SD_SW * 32 + 4 * 32 + 9 * 32 + 27 * 32) * sizeof(float);
    static LdsAttr attr;
    ensure_dynamic_lds(attr, reinterpret_cast<const void*>(stem_dw_kernel), lds);
    CCVPE_LAUNCH(stem_dw_kernel, dim3(std::min(tiles, 512)), dim3(256), lds, s, q);
}

// ------------------------------------------------------------------------------------------------
// Depthwise k x k (k = 3 | 5), stride 1 | 2, static zero / horizontal-circular padding, BN + swish,
// plus per-(sample, strip-lane, channel) partial sums for the squeeze-excite average pool.
// thread = (4 channels, patch of TX x TY output pixels).  The thread keeps its k*k filter taps (4 channels each)
// in registers for all the patches it walks, and every input row of a patch (k + (TX-1)*stride columns) is
// loaded once and feeds all the output rows it contributes to: 48 16-byte loads per 8 outputs for 5x5 / s1
// instead of 130 (the kernel is bound by the load path, not by HBM).  A thread walks patches lane, lane+S, ...
// of one sample, so its pooled partial sum is private and deterministic; the per-output FMA order (ky, kx
// ascending) is the same as a plain loop nest.
// ------------------------------------------------------------------------------------------------
template <int K, int STRIDE, int TX, int TY>
__global__ __launch_bounds__(256) void depthwise_kernel(const DwParams p) {
    constexpr int NCOL = K + (TX - 1) * STRIDE;
    constexpr int NROW = K + (TY - 1) * STRIDE;
    const int cg_n = p.C >> 2;
    const int sx_n = (p.OW + TX - 1) / TX;
    const int nstrips = ((p.OH + TY - 1) / TY) * sx_n;
    const long long total = (long long)p.B * p.S * cg_n;
    for (long long it = (long long)blockIdx.x * 256 + threadIdx.x; it < total; it += (long long)gridDim.x * 256) {
        const int cg = (int)(it % cg_n);
        long long t = it / cg_n;
        const int lane_s = (int)(t % p.S);
        const int b = (int)(t / p.S);
        const int c0 = cg * 4;
        const float4 bias = *reinterpret_cast<const float4*>(p.bias + c0);
        float4 w[K * K];
#pragma unroll
        for (int i = 0; i < K * K; ++i) w[i] = *reinterpret_cast<const float4*>(p.w + i * p.C + c0);
        float4 pool = make_float4(0.f, 0.f, 0.f, 0.f);
        const float* inb = p.in + (size_t)b * p.H * p.W * p.C + c0;
        float* outb = p.out + (size_t)b * p.OH * p.OW * p.C + c0;
        for (int strip = lane_s; strip < nstrips; strip += p.S) {
            const int sy = strip / sx_n;
            const int oy0 = sy * TY;
            const int ox0 = (strip - sy * sx_n) * TX;
            float4 acc[TY][TX];
#pragma unroll
            for (int j = 0; j < TY; ++j)
#pragma unroll
                for (int i = 0; i < TX; ++i) acc[j][i] = bias;
            const int ix0 = ox0 * STRIDE - p.pad_l;
#pragma unroll
            for (int r = 0; r < NROW; ++r) {
                const int iy = oy0 * STRIDE - p.pad_t + r;
                if ((unsigned)iy >= (unsigned)p.H) continue;
                const float* row = inb + (size_t)iy * p.W * p.C;
                float4 col[NCOL];
#pragma unroll
                for (int j = 0; j < NCOL; ++j) {
                    int ix = ix0 + j;
                    bool ok = true;
                    if (p.circular) {
                        if (ix < 0) ix += p.W;
                        else if (ix >= p.W) ix -= p.W;
                        ok = (unsigned)ix < (unsigned)p.W;   // patches past the right edge
                    } else {
                        ok = (unsigned)ix < (unsigned)p.W;
                    }
                    col[j] = ok ? *reinterpret_cast<const float4*>(row + (size_t)ix * p.C) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int ty = 0; ty < TY; ++ty) {
                    const int ky = r - ty * STRIDE;      // compile-time after unrolling
                    if (ky < 0 || ky >= K) continue;
#pragma unroll
                    for (int kx = 0; kx < K; ++kx) {
                        const float4 wv = w[ky * K + kx];
#pragma unroll
                        for (int i = 0; i < TX; ++i) {
                            const float4 v = col[i * STRIDE + kx];
                            acc[ty][i].x = fmaf(v.x, wv.x, acc[ty][i].x);
                            acc[ty][i].y = fmaf(v.y, wv.y, acc[ty][i].y);
                            acc[ty][i].z = fmaf(v.z, wv.z, acc[ty][i].z);
                            acc[ty][i].w = fmaf(v.w, wv.w, acc[ty][i].w);
                        }
                    }
                }
            }
#pragma unroll
            for (int ty = 0; ty < TY; ++ty)
#pragma unroll
                for (int i = 0; i < TX; ++i) {
                    if (oy0 + ty < p.OH && ox0 + i < p.OW) {
                        float4 o;
                        o.x = swishf(acc[ty][i].x); o.y = swishf(acc[ty][i].y); o.z = swishf(acc[ty][i].z); o.w = swishf(acc[ty][i].w);
                        pool.x += o.x; pool.y += o.y; pool.z += o.z; pool.w += o.w;
                        *reinterpret_cast<float4*>(outb + ((size_t)(oy0 + ty) * p.OW + ox0 + i) * p.C) = o;
                    }
                }
        }
        *reinterpret_cast<float4*>(p.pool_partial + ((size_t)b * p.S + lane_s) * p.C + c0) = pool;
    }
}

static constexpr int DW_TX = 4;
static constexpr int DW_TY = 4;   // measured at batch 32: 1.58 ms (1 row), 1.49 (2), 1.33 (4) over the 26 launches

int depthwise_strip_lanes(int B, int OH, int OW, int C, int k, int stride) {
    (void)k; (void)stride;
    const int ty = DW_TY;
    const int sx_n = (OW + DW_TX - 1) / DW_TX;
    const int nstrips = ((OH + ty - 1) / ty) * sx_n;
    const long long target = 256LL * 1024;   // ~4 waves per SIMD over the chip (the kernel holds ~170 VGPRs)
    long long s = target / ((long long)B * (C / 4));
    if (s < 1) s = 1;
    if (s > nstrips) s = nstrips;
    return (int)s;
}

void launch_depthwise(const DwParams& p, hipStream_t s) {
    long long total = (long long)p.B * p.S * (p.C / 4);
    int blocks = (int)((total + 255) / 256);
    if (p.k == 3 && p.stride == 1) CCVPE_LAUNCH((depthwise_kernel<3, 1, DW_TX, DW_TY>), dim3(blocks), dim3(256), 0, s, p);
    else if (p.k == 3 && p.stride == 2) CCVPE_LAUNCH((depthwise_kernel<3, 2, DW_TX, DW_TY>), dim3(blocks), dim3(256), 0, s, p);
    else if (p.k == 5 && p.stride == 1) CCVPE_LAUNCH((depthwise_kernel<5, 1, DW_TX, DW_TY>), dim3(blocks), dim3(256), 0, s, p);
    else CCVPE_LAUNCH((depthwise_kernel<5, 2, DW_TX, DW_TY>), dim3(blocks), dim3(256), 0, s, p);
}

// ------------------------------------------------------------------------------------------------
// Squeeze-excite: (1) reduce the depthwise kernel's per-lane partial sums to the pooled mean, 64
// channels per block with the S rows split over the block's 4 waves (coalesced 256 B rows);
// (2) per sample: 1x1 (C->SQ) + swish -> 1x1 (SQ->C) + sigmoid.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void se_pool_kernel(const SeParams p, float* pooled) {
    // grid (C/64, B, SC): block z reduces rows z, z+SC, ... of the [S][C] partial sums -> pooled[b][z][c]
    __shared__ float red[4][64];
    const int b = blockIdx.y, z = blockIdx.z, SC = gridDim.z;
    const int cl = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    float acc = 0.f;
    if (c < p.C) {
        const float* pp = p.pool_partial + (size_t)b * p.S * p.C + c;
        for (int s = z * 4 + sl; s < p.S; s += 4 * SC) acc += pp[(size_t)s * p.C];
    }
    red[sl][cl] = acc;
    __syncthreads();
    if (sl == 0 && c < p.C) pooled[((size_t)b * SC + z) * p.C + c] = red[0][cl] + red[1][cl] + red[2][cl] + red[3][cl];
}

// squeeze: grid (SQ, B), one block per (j, sample): 256 threads reduce over C (<= 5 loads each)
__global__ __launch_bounds__(256) void se_squeeze_kernel(const SeParams p, const float* pooled_g, float* sq_g) {
    __shared__ float red[4];
    const int j = blockIdx.x, b = blockIdx.y;
    float acc = 0.f;
    for (int c = threadIdx.x; c < p.C; c += 256) {
        float pv = 0.f;
        for (int z = 0; z < p.SC; ++z) pv += pooled_g[((size_t)b * p.SC + z) * p.C + c];
        acc = fmaf(p.w1[(size_t)j * p.C + c], pv * p.inv_hw, acc);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) sq_g[(size_t)b * p.SQ + j] = swishf(red[0] + red[1] + red[2] + red[3] + p.b1[j]);
}

// excite: grid (ceil(C/256), B), thread per channel; w2 is [SQ][C] so lanes read consecutive channels
__global__ __launch_bounds__(256) void se_excite_kernel(const SeParams p, const float* sq_g) {
    __shared__ float sq[64];
    const int b = blockIdx.y;
    if (threadIdx.x < p.SQ) sq[threadIdx.x] = sq_g[(size_t)b * p.SQ + threadIdx.x];
    __syncthreads();
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= p.C) return;
    float acc = p.b2[c];
    int j = 0;
    for (; j + 8 <= p.SQ; j += 8) {
        float w[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) w[u] = p.w2[(size_t)(j + u) * p.C + c];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = fmaf(w[u], sq[j + u], acc);
    }
    for (; j < p.SQ; ++j) acc = fmaf(p.w2[(size_t)j * p.C + c], sq[j], acc);
    p.gate[(size_t)b * p.C + c] = 1.f / (1.f + __expf(-acc));
}

void launch_se(const SeParams& p, hipStream_t s) {
    if (p.S <= 16) {   // the image-resident front kernels leave 1-16 partial rows per sample: the squeeze kernel sums them itself
        SeParams q = p;
        q.SC = p.S;
        CCVPE_LAUNCH(se_squeeze_kernel, dim3(p.SQ, p.B), dim3(256), 0, s, q, (const float*)p.pool_partial, p.sq);
        CCVPE_LAUNCH(se_excite_kernel, dim3((p.C + 255) / 256, p.B), dim3(256), 0, s, p, (const float*)p.sq);
        return;
    }
    CCVPE_LAUNCH(se_pool_kernel, dim3((p.C + 63) / 64, p.B, p.SC), dim3(256), 0, s, p, p.pooled);
    CCVPE_LAUNCH(se_squeeze_kernel, dim3(p.SQ, p.B), dim3(256), 0, s, p, (const float*)p.pooled, p.sq);
    CCVPE_LAUNCH(se_excite_kernel, dim3((p.C + 255) / 256, p.B), dim3(256), 0, s, p, (const float*)p.sq);
}

// ------------------------------------------------------------------------------------------------
// Ground descriptors: weighted sum over the H_f feature rows, layout d[w*c + ch].
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void grd_desc_kernel(const GrdDescParams p) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= p.Ltot) return;
    int k = 0;
#pragma unroll
    for (int j = 1; j < 6; ++j)
        if (j < p.nlev && i >= p.loff[j]) k = j;
    const int local = i - p.loff[k];
    if (local >= p.Wf * p.c[k]) {   // alignment gap between two levels
        p.desc[(size_t)b * p.Ltot + i] = 0.f;
        return;
    }
    const int w = local / p.c[k];
    const int ch = local - w * p.c[k];
    float acc = p.b2[k];
    const float* y = p.y + ((size_t)b * p.Hf * p.Wf + w) * p.Ntot + p.off[k] + ch;
    for (int h = 0; h < p.Hf; ++h) acc = fmaf(p.wh[k][h], y[(size_t)h * p.Wf * p.Ntot], acc);
    p.desc[(size_t)b * p.Ltot + i] = acc;
}

void launch_grd_desc(const GrdDescParams& p, hipStream_t s) {
    CCVPE_LAUNCH(grd_desc_kernel, dim3((p.Ltot + 255) / 256, p.B), dim3(256), 0, s, p);
}

}  // namespace ccvpe
