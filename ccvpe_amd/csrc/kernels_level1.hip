// Fused last decoder level:  ConvTranspose2d(k2,s2, CX->16) -> conv3x3(16->16)+ReLU -> conv3x3(16->{1,2})
// (-> F.normalize for the orientation branch), writing the NCHW outputs directly.
//
// Reference: deconv1 / conv1 (models.py:422-425, 625-626) and deconv1_ori / conv1_ori + normalize
// (models.py:443-446, 648-650); KITTI / Oxford copies at :727-728, :748-749, :1026-1028, :1046-1048.
//
// Unfused, this level moves three 512x512x16 fp32 tensors per sample through HBM and runs its 3x3 conv as
// an N=16 GEMM with 5 K tiles (prologue/epilogue dominated).  Here one workgroup owns a 16x16 output
// tile: the 10x10 input pixels it depends on are staged in LDS once, the transposed conv (a [100 x CX] x
// [CX x 64] GEMM) and the 3x3 conv (a [324 x 144] x [144 x 16] GEMM whose A operand is gathered from the
// LDS tile at 9 shifted positions) run on v_mfma_f32_16x16x4_f32, the final 16->{1,2} conv as per-tap dot products on
// the same MFMAs plus nine adds per output (round 3; on the VALU before), and only the 1-2 output channels leave the chip.  Zero padding of both 3x3 convs is applied where the
// reference applies it: intermediate pixels outside the 512x512 image are forced to 0 (not bias).
#include "kernels.h"

#include <algorithm>
#include <cstdio>

namespace ccvpe {

typedef float f32x4 __attribute__((ext_vector_type(4)));

static constexpr int T = 16;          // output tile
static constexpr int DT = T + 4;      // deconv-output tile (halo 2)
static constexpr int AT = T + 2;      // conv_a-output tile (halo 1)
static constexpr int XT = DT / 2;     // input tile (10 x 10)
static constexpr int PS = 20;         // floats per pixel in the D / A tiles (16 + 4 pad: conflict-free b128)
static constexpr int DSINK = DT + 2;  // sink rows behind the D tile: offset (dy*DT + dx) past row DT*DT stays inside
static constexpr int AROWS = ((AT * AT + 15) / 16) * 16;   // A tile rows incl. the padding rows of the last m-tile
typedef float f32x2 __attribute__((ext_vector_type(2)));


#ifndef CCVPE_L1_CLOCK
#define CCVPE_L1_CLOCK 0   // dev builds (tools/build_variant.sh): 1 = every wave sums s_memtime per stage (input -> LDS + barrier, deconv, barrier, conv_a, barrier, tail conv + stores, end barrier)
#endif
#if CCVPE_L1_CLOCK
__device__ unsigned long long g_l1_clk[10];
#define CCVPE_L1_STAMP(i_) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); clk[i_] += t_ - tprev; tprev = t_; }
#else
#define CCVPE_L1_STAMP(i_)
#endif

static constexpr int KCH_MAX = 4;     // input channels <= 64 (16 per k-chunk); the host falls back to the unfused path beyond
static constexpr int XI_MAX = (XT * XT * KCH_MAX * 4 + 255) / 256;   // float4 items per thread of one X tile

// Persistent: 2 workgroups per CU loop over the 16x16 output tiles (XCD x owns a contiguous run, so neighbouring
// tiles - which share their input halo - meet in one L2).  Per-workgroup constants (both weight sets as MFMA B
// fragments, biases, the pixel -> LDS offset table) are set up once, and the next tile's input pixels are loaded
// into registers while the current tile runs its three stages.  Measured before this: with one tile per workgroup
// 0.36 of the 1.06 ms was launch + weight staging + exposed load latency that nothing overlapped.
template <int COUT>
__global__ __launch_bounds__(256) void level1_kernel(const Level1Params p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int CXP = p.cxp;                 // input channels padded to a multiple of 16
    const int XS = CXP + 4;                // row stride of the X tile
    const int r0f = max(XT * XT * XS, AROWS * PS);   // region 0 holds the X tile, later the aliased A tile
    float* Xs = smem;                      // [XT*XT][XS]
    float* As = smem;                      // [AROWS][PS]  (aliases Xs, dead after the deconv stage; rows >= 324 are a sink)
    float* Ds = smem + r0f;                // [DT*DT + DSINK][PS]  (tail rows: sink for the 12 padding rows of the last m-tile)
    int* dtab = reinterpret_cast<int*>(Ds + (DT * DT + DSINK) * PS);   // [112] X pixel -> float offset of its (dy,dx) = (0,0) D pixel

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int H = p.H, W = p.W;            // output size (512); input is H/2 x W/2
    const int IH = H >> 1, IW = W >> 1;
    const int tiles_x = W / T, tiles_y = H / T;
    const int tiles = p.B * tiles_x * tiles_y;
    const int xcd = blockIdx.x & 7;
    const int stride = ((int)gridDim.x >> 3) + (xcd < ((int)gridDim.x & 7) ? 1 : 0);
    const int t_begin = xcd * (tiles >> 3) + min(xcd, tiles & 7);
    const int t_end = t_begin + (tiles >> 3) + (xcd < (tiles & 7) ? 1 : 0);
    int tile = t_begin + ((int)blockIdx.x >> 3);
    if (tile >= t_end) return;

    // ---- once per workgroup ----
    // Index arithmetic on the vector ALU costs matrix-pipe issue slots on gfx950, so the pixel -> LDS offset maps of
    // the two epilogues are tabulated (stage 1) or affine (stage 2); tiles whose halo lies fully inside the image
    // (88 % of them) also skip every bounds test.
    if (tid < 112) dtab[tid] = tid < XT * XT ? ((2 * (tid / XT)) * DT + 2 * (tid % XT)) * PS : DT * DT * PS;
    // Last conv (16 -> COUT, 3x3) in two steps (round 3): P[pixel][tap, co] = sum_c A[pixel][c] wt[tap][co][c] for every pixel of the 18 x 18
    // conv_a tile - a [324 x 16] x [16 x 9 COUT] GEMM whose B operand is the conv_a accumulator AS IT STANDS in registers (lane = pixel,
    // 4 channels: exactly the operand layout), 4 MFMAs per 16 pixels and 16 (tap, co) columns - and out[y][x][co] = bt + sum_tap P[(y + dy,
    // x + dx)][tap, co]: nine LDS reads and adds per output.  The P tile takes the A tile's place in LDS (18 of its 20 floats per pixel).
    // Before: 72 COUT packed FMAs, 36 16-byte A reads and 36 COUT weight reads per output pixel on the vector pipe - a quarter of the
    // kernel's time (in-kernel stamps, CCVPE_L1_CLOCK).
    constexpr int NPT = (9 * COUT + 15) / 16;                     // 16-column tiles of P
    f32x4 wtf[NPT];                                               // A fragments: wt[n = 16 nt + (lane & 15)][4 (lane >> 4) + e], n = tap * COUT + co
#pragma unroll
    for (int nt = 0; nt < NPT; ++nt) {
        const int n = nt * 16 + (lane & 15);
        wtf[nt] = n < 9 * COUT ? *reinterpret_cast<const f32x4*>(p.wt + (size_t)n * 16 + 4 * (lane >> 4)) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int kch = CXP >> 4;
    // deconv weights: wave w owns output parity (dy,dx) = (w>>1, w&1); B operand of k-chunk kc, MFMA j is
    // Wd[n = w*16 + (lane&15)][16*kc + 4*(lane>>4) + j]
    f32x4 wd[KCH_MAX];
#pragma unroll
    for (int kc = 0; kc < KCH_MAX; ++kc)
        wd[kc] = kc < kch ? *reinterpret_cast<const f32x4*>(p.wd + (size_t)(wave * 16 + (lane & 15)) * CXP + kc * 16 + 4 * (lane >> 4)) : f32x4{0.f, 0.f, 0.f, 0.f};
    // conv_a weights for this lane: B operand of MFMA j at tap t is Wa[n = lane&15][t*16 + 4*(lane>>4) + j]
    f32x4 wa[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) wa[t] = *reinterpret_cast<const f32x4*>(p.wa + (size_t)(lane & 15) * 144 + t * 16 + 4 * (lane >> 4));
    // channel-major accumulators (weights are the A operand of the MFMAs): a lane holds channels 4 (lane >> 4) .. + 3 of ONE pixel
    const f32x4 bd = *reinterpret_cast<const f32x4*>(p.bd + 4 * (lane >> 4));
    const f32x4 ba = *reinterpret_cast<const f32x4*>(p.ba + 4 * (lane >> 4));

    // X tile staging: float4 item i = tid + it*256 -> pixel i / c4n, channels 4*(i % c4n)
    const int c4n = CXP >> 2;
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, (unsigned)((size_t)p.B * IH * IW * p.x_ld * 4), 0x00020000);
    int x_lds[XI_MAX], x_rc[XI_MAX];   // LDS float offset (-1: no item), (row << 8 | col) inside the 10x10 tile, channel in bits 16+
#pragma unroll
    for (int it = 0; it < XI_MAX; ++it) {
        const int i = tid + it * 256;
        const int px = i / c4n, c4 = i - px * c4n;
        const bool live = i < XT * XT * c4n;
        x_lds[it] = live ? px * XS + c4 * 4 : -1;
        x_rc[it] = ((px / XT) << 8) | (px % XT) | ((c4 * 4 < p.cx ? c4 * 4 : 0x7fff) << 16);
    }
    f32x4 xv[XI_MAX];
#define CCVPE_L1_LOAD_X(tl)                                                                              \
    {                                                                                                    \
        const int b_ = (tl) / (tiles_x * tiles_y);                                                       \
        const int r_ = (tl) - b_ * (tiles_x * tiles_y);                                                  \
        const int ty_ = r_ / tiles_x, tx_ = r_ - ty_ * tiles_x;                                          \
        const int xr0_ = ty_ * (T / 2) - 1, xc0_ = tx_ * (T / 2) - 1;                                    \
        _Pragma("unroll") for (int it = 0; it < XI_MAX; ++it) {                                          \
            const int xr = xr0_ + ((x_rc[it] >> 8) & 0xff), xc = xc0_ + (x_rc[it] & 0xff), ch = x_rc[it] >> 16; \
            const bool ok = x_lds[it] >= 0 && (unsigned)xr < (unsigned)IH && (unsigned)xc < (unsigned)IW && ch != 0x7fff; \
            const unsigned off = ok ? (unsigned)((((b_ * IH + xr) * IW + xc) * p.x_ld + ch) * 4) : 0x80000000u; \
            xv[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, off, 0, 0)); \
        }                                                                                                \
    }
    CCVPE_L1_LOAD_X(tile);

    const int oy = tid >> 4, ox = tid & 15;      // stage 3: one output pixel per thread
    const size_t hw = (size_t)H * W;
    constexpr int NMT = (AT * AT + 15) / 16;   // 21

#if CCVPE_L1_CLOCK
    unsigned long long clk[7] = {0, 0, 0, 0, 0, 0, 0}, tprev = __builtin_amdgcn_s_memtime();
    int ntiles = 0;
#endif
    while (true) {
#if CCVPE_L1_CLOCK
        ++ntiles;
#endif
        const int b = tile / (tiles_x * tiles_y);
        const int rem = tile - b * (tiles_x * tiles_y);
        const int Y0 = (rem / tiles_x) * T, X0 = (rem % tiles_x) * T;
        const bool interior = Y0 >= 2 && Y0 + T + 2 <= H && X0 >= 2 && X0 + T + 2 <= W;

        // ---- stage 0: this tile's input pixels registers -> LDS; start fetching the next tile's ----
#pragma unroll
        for (int it = 0; it < XI_MAX; ++it)
            if (x_lds[it] >= 0) *reinterpret_cast<f32x4*>(Xs + x_lds[it]) = xv[it];
        __syncthreads();
        CCVPE_L1_STAMP(0);
        const int tile_n = tile + stride;
        const bool have_n = tile_n < t_end;
        if (have_n) { CCVPE_L1_LOAD_X(tile_n); }

        // ---- stage 1: transposed conv as GEMM [100 x CXP] x [CXP x 64]; unit = (m-tile, (dy,dx)) ----
        // two independent accumulator chains per wave (m-tiles mt and mt+1) hide the dependent-MFMA latency
        {
            const int dy = wave >> 1, dx = wave & 1;
            float* dsub = Ds + (dy * DT + dx) * PS + 4 * (lane >> 4);
            for (int mt0 = 0; mt0 < 7; mt0 += 2) {
                const int mt1 = mt0 + 1;               // may be 7 (invalid): computed on clamped rows, never stored
                const float* ap0 = Xs + min(mt0 * 16 + (lane & 15), XT * XT - 1) * XS + 4 * (lane >> 4);
                const float* ap1 = Xs + min(mt1 * 16 + (lane & 15), XT * XT - 1) * XS + 4 * (lane >> 4);
                f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
                if (mt1 < 7) {
#pragma unroll
                    for (int kc = 0; kc < KCH_MAX; ++kc) {
                        if (kc >= kch) break;
                        const f32x4 a0 = *reinterpret_cast<const f32x4*>(ap0 + kc * 16);
                        const f32x4 a1 = *reinterpret_cast<const f32x4*>(ap1 + kc * 16);
                        const f32x4 w = wd[kc];
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w.x, a0.x, acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w.x, a1.x, acc1, 0, 0, 0);
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w.y, a0.y, acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w.y, a1.y, acc1, 0, 0, 0);
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w.z, a0.z, acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w.z, a1.z, acc1, 0, 0, 0);
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w.w, a0.w, acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w.w, a1.w, acc1, 0, 0, 0);
                    }
                } else {   // the seventh m-tile has no partner: one chain
#pragma unroll
                    for (int kc = 0; kc < KCH_MAX; ++kc) {
                        if (kc >= kch) break;
                        const f32x4 a0 = *reinterpret_cast<const f32x4*>(ap0 + kc * 16);
                        const f32x4 w = wd[kc];
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w.x, a0.x, acc0, 0, 0, 0);
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w.y, a0.y, acc0, 0, 0, 0);
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w.z, a0.z, acc0, 0, 0, 0);
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w.w, a0.w, acc0, 0, 0, 0);
                    }
                }
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    const int mt = h2 ? mt1 : mt0;
                    if (mt >= 7) continue;
                    const f32x4 acc = h2 ? acc1 : acc0;
                    const int px = mt * 16 + (lane & 15);           // X pixel of this lane's accumulator (>= 100: sink row)
                    const int dto = dtab[px];
                    if (interior) {
                        *reinterpret_cast<f32x4*>(dsub + dto) = acc + bd;
                    } else if (px < XT * XT) {
                        const int dr = 2 * (px / XT) + dy, dc = 2 * (px % XT) + dx;          // position in the D tile
                        const int gy = Y0 - 2 + dr, gx = X0 - 2 + dc;                        // position in the image
                        const bool in = (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
                        *reinterpret_cast<f32x4*>(dsub + dto) = in ? acc + bd : f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                }
            }
        }
        CCVPE_L1_STAMP(1);
        __syncthreads();
        CCVPE_L1_STAMP(2);

        // ---- stage 2: conv3x3 16->16 + ReLU on the 18x18 halo tile: 21 m-tiles of 16 pixels, 36 MFMAs each ----
        for (int mt0 = wave; mt0 < NMT; mt0 += 8) {
            const int mt1 = mt0 + 4;                   // second chain (may be >= NMT: clamped reads, no stores)
            const int pa0 = min(mt0 * 16 + (lane & 15), AT * AT - 1);
            const int pa1 = min(mt1 * 16 + (lane & 15), AT * AT - 1);
            const float* dp0 = Ds + ((pa0 / AT) * DT + pa0 % AT) * PS + 4 * (lane >> 4);
            const float* dp1 = Ds + ((pa1 / AT) * DT + pa1 % AT) * PS + 4 * (lane >> 4);
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
            if (mt1 < NMT) {
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int toff = ((t / 3) * DT + (t % 3)) * PS;
                    const f32x4 a0 = *reinterpret_cast<const f32x4*>(dp0 + toff);
                    const f32x4 a1 = *reinterpret_cast<const f32x4*>(dp1 + toff);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[t].x, a0.x, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[t].x, a1.x, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[t].y, a0.y, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[t].y, a1.y, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[t].z, a0.z, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[t].z, a1.z, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[t].w, a0.w, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[t].w, a1.w, acc1, 0, 0, 0);
                }
            } else {   // the last m-tile of a wave has no partner (21 m-tiles over 4 waves x 2 chains): one chain, no padding MFMAs
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int toff = ((t / 3) * DT + (t % 3)) * PS;
                    const f32x4 a0 = *reinterpret_cast<const f32x4*>(dp0 + toff);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[t].x, a0.x, acc0, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[t].y, a0.y, acc0, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[t].z, a0.z, acc0, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[t].w, a0.w, acc0, 0, 0, 0);
                }
            }
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                const int mt = h2 ? mt1 : mt0;
                if (mt >= NMT) continue;
                const f32x4 acc = h2 ? acc1 : acc0;
                const int q = mt * 16 + (lane & 15);                // A pixel of this lane's accumulator (>= 324: sink row)
                f32x4 v = acc + ba;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
                if (!interior) {
                    const int gy = Y0 - 1 + q / AT, gx = X0 - 1 + q % AT;
                    const bool in = q < AT * AT && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
                    if (!in) v = f32x4{0.f, 0.f, 0.f, 0.f};
                }
                // the last conv's per-tap dot products of this pixel: v is the B operand as it stands
#pragma unroll
                for (int nt = 0; nt < NPT; ++nt) {
                    f32x4 pa = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int e = 0; e < 4; ++e) pa = __builtin_amdgcn_mfma_f32_16x16x4f32(wtf[nt][e], v[e], pa, 0, 0, 0);
                    // columns 16 nt + 4 (lane >> 4) .. + 3 of pixel q; 9 COUT <= 18 columns are real: the second tile keeps two
                    if (nt == 0) *reinterpret_cast<f32x4*>(As + q * PS + 4 * (lane >> 4)) = pa;
                    else if ((lane >> 4) == 0) *reinterpret_cast<f32x2*>(As + q * PS + 16) = f32x2{pa[0], pa[1]};
                }
            }
        }
        CCVPE_L1_STAMP(3);
        __syncthreads();
        CCVPE_L1_STAMP(4);

        // ---- stage 3: out = bias + the nine taps' dot products of the shifted pixels, one output pixel per thread, NCHW store ----
        float o[COUT];
#pragma unroll
        for (int c = 0; c < COUT; ++c) o[c] = p.bt[c];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const float* pp = As + ((oy + t / 3) * AT + ox + t % 3) * PS + t * COUT;
            if (COUT == 2) { const f32x2 v = *reinterpret_cast<const f32x2*>(pp); o[0] += v.x; o[COUT - 1] += v.y; }
            else o[0] += pp[0];
        }
        const size_t opix = (size_t)(Y0 + oy) * W + X0 + ox;
        if (p.raw) {
#pragma unroll
            for (int c = 0; c < COUT; ++c) p.raw[((size_t)b * COUT + c) * hw + opix] = o[c];
        }
        if (p.normalize) {
            float n2 = 0.f;
#pragma unroll
            for (int c = 0; c < COUT; ++c) n2 = fmaf(o[c], o[c], n2);
            const float inv = 1.f / fmaxf(sqrtf(n2), 1e-12f);
#pragma unroll
            for (int c = 0; c < COUT; ++c) o[c] *= inv;
        }
#pragma unroll
        for (int c = 0; c < COUT; ++c) p.out[((size_t)b * COUT + c) * hw + opix] = o[c];

        CCVPE_L1_STAMP(5);
        if (!have_n) break;
        __syncthreads();   // the A tile (aliasing Xs) is fully consumed before the next X tile lands
        CCVPE_L1_STAMP(6);
        tile = tile_n;
    }
#if CCVPE_L1_CLOCK
    if (lane == 0) {
        for (int i = 0; i < 7; ++i) atomicAdd(&g_l1_clk[i], clk[i]);
        atomicAdd(&g_l1_clk[7], 1ull);
        atomicAdd(&g_l1_clk[8], (unsigned long long)ntiles);
    }
#endif
#undef CCVPE_L1_LOAD_X
}

bool level1_supported(int cxp) { return cxp >= 16 && cxp <= 16 * KCH_MAX && cxp % 16 == 0; }

size_t level1_lds_bytes(int cxp, int cout) {
    const int XS = cxp + 4;
    const size_t r0f = std::max<size_t>((size_t)XT * XT * XS, (size_t)AROWS * PS);
    (void)cout;
    return (r0f + (size_t)(DT * DT + DSINK) * PS + 112) * sizeof(float);
}

void launch_level1(const Level1Params& p, hipStream_t s) {
    const size_t lds = level1_lds_bytes(p.cxp, p.cout);
    const int tiles = (p.W / T) * (p.H / T) * p.B;
    dim3 grid(std::min(tiles, 2 * 256));   // persistent: two workgroups per CU
#if CCVPE_L1_CLOCK
    static int calls = 0;
    const bool stamp = ++calls % 4 == 0;
    if (stamp) { (void)hipStreamSynchronize(s); unsigned long long z[10] = {}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_l1_clk), z, sizeof z); }
#endif
    if (p.cout == 1) {
        static LdsAttr attr1;
        ensure_dynamic_lds(attr1, reinterpret_cast<const void*>(level1_kernel<1>), lds);
        CCVPE_LAUNCH(level1_kernel<1>, grid, dim3(256), lds, s, p);
    } else {
        static LdsAttr attr2;
        ensure_dynamic_lds(attr2, reinterpret_cast<const void*>(level1_kernel<2>), lds);
        CCVPE_LAUNCH(level1_kernel<2>, grid, dim3(256), lds, s, p);
    }
#if CCVPE_L1_CLOCK
    if (stamp) {
        (void)hipStreamSynchronize(s);
        unsigned long long h[10];
        (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_l1_clk), sizeof h);
        double tot = 0;
        for (int i = 0; i < 7; ++i) tot += (double)h[i];
        std::fprintf(stderr, "level1<%d> cx %d: %.0f cycles/tile and wave: input->LDS+barrier %.0f %% deconv %.0f %% barrier %.0f %% conv_a %.0f %% barrier %.0f %% tail conv %.0f %% end barrier %.0f %%\n",
                     p.cout, p.cx, tot / std::max(1.0, (double)h[8]), 100 * h[0] / tot, 100 * h[1] / tot, 100 * h[2] / tot, 100 * h[3] / tot, 100 * h[4] / tot, 100 * h[5] / tot, 100 * h[6] / tot);
    }
#endif
}

}  // namespace ccvpe
