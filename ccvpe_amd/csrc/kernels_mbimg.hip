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
//   * a workgroup owns (sample, every CG-th chunk of 16 expanded channels);
//   * expand: wave w takes m-tiles w, w + NWV, ... of the [pixels x Cin] x [Cin x 16] GEMM.  Both operands reach the MFMA
//     registers THROUGH LDS (round 3): in the MFMA layout lane l holds row (l & 15), so four consecutive lanes of a direct
//     16-byte load hit four different rows - the texture addresser then moves 18 B/clk per CU (tools/ubench_ta.hip; the loop
//     spent 80 % of its time issuing loads, tools/time_ops.py + CCVPE_MI_CLOCK).  The same bytes with lane l -> row (l >> 2),
//     piece (l & 3) move at 60 B/clk: every wave brings its m-tile in with LDS-DMA in that order (one instruction per 16 input
//     channels, pieces rotated by (row >> 2) so the MFMA-layout read back is bank-conflict free), reads it into registers,
//     requests the following tile into the same buffer and runs the MFMA chain; the chunk's expand weights take the same road
//     once per workgroup instead of once per wave.  (Also built and measured: the tiles through registers - coalesced loads a step
//     ahead into two register sets, 16-byte LDS writes in lane order and the same read back right before use; 10-14 KB per wave in
//     flight instead of the buffer's 4-12 KB, yet 1.4 % slower over the 29 launches: 2.415 against 2.381 ms at batch 32.)
//   * bias + swish, then the accumulator rows are scattered into the zero-padded (or horizontally wrapped) E image through a
//     pixel -> offset table;
//   * depthwise: thread = (4 channels, TX x TY output patch); the K x K taps come from LDS once per chunk, every input
//     row of the patch window is read once and used by all taps; bias + swish, 16-byte stores, channel sums for the
//     squeeze-excite pool reduced in a fixed order (wave shuffles, then 8 partials through LDS) -> deterministic;
//   * two barriers per chunk; the next chunk's expand weights and first m-tiles land under the depthwise phase.
// Samples are pinned to XCDs (sample b only on workgroups with id % 8 == b % 8) so the ~40-70 passes over a sample's
// input hit that XCD's L2.
#include "igemm_common.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>

namespace ccvpe {

#ifndef CCVPE_MI_CLOCK
#define CCVPE_MI_CLOCK 0   // dev builds (tools/build_variant.sh): 1 = every wave sums s_memtime per phase (unit set-up, expand, barrier, depthwise, tail)
#endif
#if CCVPE_MI_CLOCK
__device__ unsigned long long g_mi_clk[12];
#define CCVPE_MI_STAMP(i_) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); clk[i_] += t_ - tprev; tprev = t_; }
#else
#define CCVPE_MI_STAMP(i_)
#endif

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
// workgroups' LDS fits.  SK = 16-channel groups per staging step: the whole m-tile (SK = KCH) where the LDS has room for
// NT / 64 x KCH KB beside the E image, else half of it (two steps per tile).  (Measured and dropped: keeping the input m-tiles of a
// <= 256-pixel image in registers across all chunks; TWO chunks per item on the small images of blocks 12-15 - every staged m-tile
// feeding two MFMA chains, one workgroup of 512 threads per CU whose depthwise phase is one pass of all threads: 0.627 against 0.612 ms
// for blocks 11-15 at batch 32, the overlap of two out-of-phase workgroups is worth as much as the halved input traffic; the two halves of a
// tile as a ring - a half requested for the next tile as soon as it is read, `vmcnt(other half)` waits - 2.361 against 2.353 ms over the 29
// launches: spreading the requests changes nothing; the items of the strip layers dealt round robin inside an XCD, so that the nine chunks
// of a strip - which read the same 150 KB of input - run on nine workgroups at once instead of one after the other (FETCH_SIZE of blocks 2 / 3
// is 9x their input): 0.859 against 0.765 ms for blocks 2-5 - every item then starts with a new strip's zero-fill and tables.)
template <int K, int S, int KCH, int SK, int TX, int TY, int NT>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(2))) void mbconv_image_kernel(const MbImgParams q) {
    constexpr int NWV = NT / 64;
    constexpr int WW = (TX - 1) * S + K;
    constexpr int NS = (KCH + SK - 1) / SK;              // staging steps per m-tile
    static_assert(NS <= 2, "whole or half tiles");
    typedef __attribute__((address_space(3))) void* lds_ptr;
    const MbFrontParams& p = q.f;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int e_floats = (q.HPa * q.WPa + 1) * EPS;      // + one sink pixel
    float* Es = smem;
    int* ptab = reinterpret_cast<int*>(smem + e_floats);  // [NMT*16] pixel -> float offset in Es (sink past P)
    int* dtab = ptab + q.NMT * 16;                         // [NMT*16] circular layers: offset of the wrapped copy (sink if none)
    float* wks = reinterpret_cast<float*>(dtab + q.NMT * 16);   // [K*K][16] depthwise taps of the chunk
    float* bds = wks + K * K * 16;                         // [16] depthwise bias
    float* red = bds + 16;                                 // [NT/64][16] per-wave pooling sums
    float* Wst = red + NWV * 16;                           // [KCH][256] expand weights of the chunk, staged image (below)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);        // (uniform for the compiler too: LDS-DMA bases and offsets are scalars)
    float* stg = Wst + KCH * 256 + wave * (SK * 256);      // [SK][256] this wave's m-tile (or half of it)

    // Work = the flattened list of (unit, chunk) items, unit = (sample, strip), unit-major; a workgroup takes an equal
    // contiguous share, and the shares are dealt XCD-major (XCD x owns a contiguous eighth of the list): a sample's 40-70
    // passes over its input stay in ONE L2 (dealt round robin, every L2 saw every sample: block 9 0.091 -> 0.134 ms)
    const int total = p.B * q.NST * q.nchunks;
    const int wgx = xcd_remap(blockIdx.x, gridDim.x);
    const int it_lo = (int)((long long)total * wgx / gridDim.x), it_hi = (int)((long long)total * (wgx + 1) / gridDim.x);
    if (it_lo >= it_hi) return;
    const int sink = q.HPa * q.WPa * EPS;
    // Staged image of a [16 rows x 16 channels] block (1 KB, one LDS-DMA instruction): DMA lane l = 4 r + t fetches row r, 16-byte
    // piece (t - (r >> 2)) & 3 of the block - the four lanes of a row read one 64-byte run - and lands at byte 16 l; the MFMA lane
    // (g = l >> 4, r = l & 15) finds its piece g at 64 r + 16 ((g + (r >> 2)) & 3): the 16 lanes of one read phase hit 16 different
    // 16-byte bank groups.  The K order (channel 16 kc + 4 g + e at MFMA e of group kc) is the same on both operands.
    const int dr = lane >> 2, dpc = ((lane & 3) - (dr >> 2)) & 3;
    const unsigned a_dma = (unsigned)((dr * p.Cin + 4 * dpc) * 4), w_dma = (unsigned)((dr * p.cinp + 4 * dpc) * 4);
    const unsigned a_mt = (unsigned)(16 * p.Cin * 4), w_ch = (unsigned)(16 * p.cinp * 4);      // bytes per m-tile / per chunk of weights
    const int rd_off = 16 * (lane & 15) + 4 * (((lane >> 4) + ((lane & 15) >> 2)) & 3);         // floats
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.we), 0, (unsigned)((size_t)p.mid * p.cinp * 4), 0x00020000);
    const int dq4 = tid & 3, slot = tid >> 2;          // depthwise: channel quad, patch slot (NT / 4 slots)
    const int npatch = q.NPX * q.NPY;

    // expand weights of chunk c_ -> Wst: the workgroup's waves share the KCH instructions
#define CCVPE_MI_DMA_W(c_)                                                                                       \
    for (int kc = wave; kc < KCH; kc += NWV)                                                                     \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_ptr)(Wst + kc * 256), 16, w_dma + (unsigned)(c_) * w_ch, kc * 64, 0, 0);
    // staging step st_ of m-tile mt_ (rows past the unit's pixels and tiles past its last: the descriptor returns zeros)
#define CCVPE_MI_DMA_A(rsrc_, nmt_, mt_, st_)                                                                    \
    {                                                                                                            \
        const unsigned vo_ = (mt_) < (nmt_) ? a_dma + (unsigned)(mt_) * a_mt : 0x80000000u;                     \
        const __amdgpu_buffer_rsrc_t rs_ = rsrc_;   /* (a struct member as the argument: hipcc's host pass drops the kernel's stub without a word) */ \
        _Pragma("unroll") for (int j = 0; j < SK; ++j)                                                           \
            if ((st_) * SK + j < KCH) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_, (lds_ptr)(stg + j * 256), 16, vo_, ((st_) * SK + j) * 64, 0, 0); \
    }
    // depthwise taps / bias of a chunk: fetched into registers a phase ahead, parked in LDS after the chunk barrier
    // (every thread loads - clamped addresses - and only the owners store: a load under a condition makes hipcc copy the loop-carried
    // registers at the join and wait for the loads right there)
    const bool tap_thread = tid < K * K * 4, bias_thread = tid >= NT - 16;
    const int tap_row = min(tid >> 2, K * K - 1), bias_col = max(tid - (NT - 16), 0);
    f32x4 tapv;
    float biasv;
#define CCVPE_MI_LOAD_TAPS(c_)                                                                                   \
    {                                                                                                            \
        tapv = *reinterpret_cast<const f32x4*>(p.wd + (size_t)tap_row * p.mid + (c_) * 16 + (tid & 3) * 4);      \
        biasv = p.bd[(c_) * 16 + bias_col];                                                                      \
    }
    // bias + swish on the accumulator rows (the expand weights are the A operand, so a lane ends up with 4 consecutive channels of
    // ONE pixel), one 16-byte scatter per lane
#define CCVPE_MI_SCATTER(acc_, mt_)                                                                              \
    {                                                                                                            \
        const int pos = ptab[(mt_) * 16 + (lane & 15)];                                                          \
        f32x4 v;                                                                                                 \
        _Pragma("unroll") for (int r = 0; r < 4; ++r) v[r] = swish_i(acc_[r] + be[r]);                           \
        *reinterpret_cast<f32x4*>(Es + pos + 4 * (lane >> 4)) = v;                                               \
        if (p.circular) *reinterpret_cast<f32x4*>(Es + dtab[(mt_) * 16 + (lane & 15)] + 4 * (lane >> 4)) = v;   \
    }
#if CCVPE_MI_CLOCK
#define CCVPE_MI_VMWAIT() { const unsigned long long t0_ = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); clkw += __builtin_amdgcn_s_memtime() - t0_; }
#else
#define CCVPE_MI_VMWAIT() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#endif

    // A unit = (sample, strip): output rows [oyA, oyB), input rows [yA, yB) (clamped to the image: rows outside stay zero in the E
    // image); E row 0 is input row ey0 (negative in the first strip: the static top padding)
    struct Unit { int b, oyA, oyB, ey0, yA, P, NMT; __amdgpu_buffer_rsrc_t rsrc; };
    auto decode_unit = [&](int u) {
        Unit r;
        r.b = u / q.NST;
        const int st = u - r.b * q.NST;
        r.oyA = st * q.RO; r.oyB = min(r.oyA + q.RO, p.OH);
        r.ey0 = r.oyA * S - q.PT;
        r.yA = max(r.ey0, 0);
        const int yB = min((r.oyB - 1) * S - q.PT + K, p.H);
        r.P = (yB - r.yA) * p.W;
        r.NMT = (r.P + 15) >> 4;                          // <= q.NMT
        r.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x + ((size_t)r.b * p.H + r.yA) * p.W * p.Cin), 0, (unsigned)((size_t)r.P * p.Cin * 4), 0x00020000);
        return r;
    };
    // Everything an item needs from global memory is requested ONE item ahead, at fixed places and unconditionally (the last item
    // requests itself again): its expand weights after the barrier that ends the previous expand phase, its taps / bias and every
    // wave's first m-tile at the end of that phase.
    int unit = -1;
    const bool se_on = p.se.counter != nullptr;            // per_sample == NST * nchunks items complete a sample
    const bool rows_on = p.se.sqpart != nullptr;           // squeeze rows per item: for the ticket's combining step, or (no ticket) for the project GEMM's prologue
    const bool sq_lane = rows_on && wave == 0 && lane < p.se.SQ;
    const __amdgpu_buffer_rsrc_t se_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(rows_on ? p.se.w1 : p.we), 0, rows_on ? (unsigned)((size_t)p.se.SQ * p.mid * 4) : 0u, 0x00020000);
    Unit cur = decode_unit(it_lo / q.nchunks);
    {
        const int ch = it_lo - (it_lo / q.nchunks) * q.nchunks;
        CCVPE_MI_DMA_W(ch);
        CCVPE_MI_LOAD_TAPS(ch);
        CCVPE_MI_DMA_A(cur.rsrc, cur.NMT, wave, 0);
    }
#if CCVPE_MI_CLOCK
    unsigned long long clk[5] = {0, 0, 0, 0, 0}, clkw = 0, tprev = __builtin_amdgcn_s_memtime();
#endif
    for (int it = it_lo; it < it_hi; ++it) {
        const int u = it / q.nchunks, ch = it - u * q.nchunks;
        const int b = cur.b, oyA = cur.oyA, oyB = cur.oyB, NMT = cur.NMT;
        const __amdgpu_buffer_rsrc_t x_rsrc = cur.rsrc;
        if (u != unit) {
            unit = u;
            const int P = cur.P, yA = cur.yA, ey0 = cur.ey0;
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
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (first item: the weights of its chunk have landed before anyone reads Wst)
            __syncthreads();
        }
        CCVPE_MI_STAMP(0);
        const int ch0 = ch * 16;
        // the item after this one (itself again behind the last): its unit's pixels, for the tile requests below
        const int itn = min(it + 1, it_hi - 1);
        const int un = itn / q.nchunks, chn = itn - un * q.nchunks;
        if (un != unit) cur = decode_unit(un);             // (scalar state only; this item keeps the copies made above)
        // depthwise taps and bias of this chunk -> LDS (read after the barrier below)
        if (tap_thread) *reinterpret_cast<f32x4*>(wks + (tid >> 2) * 16 + (tid & 3) * 4) = tapv;
        if (bias_thread) bds[tid - (NT - 16)] = biasv;
        const f32x4 be = *reinterpret_cast<const f32x4*>(p.be + ch0 + 4 * (lane >> 4));   // channel-major accumulators: 4 channels of one pixel per lane
        f32x4 wf[KCH];                                      // A fragments: We[16 ch + (lane & 15)][16 kc + 4 (lane >> 4) + e]
#pragma unroll
        for (int kc = 0; kc < KCH; ++kc) wf[kc] = *reinterpret_cast<const f32x4*>(Wst + kc * 256 + rd_off);

        // ---- expand: m-tiles wave, wave + NWV, ...; the first one was requested an item ago ----
        for (int mt = wave; mt < NMT; mt += NWV) {
            // the tile this wave stages next: its following tile of this item, else its first tile of the next item
            const bool last = mt + NWV >= NMT;
            const __amdgpu_buffer_rsrc_t n_rsrc = last ? cur.rsrc : x_rsrc;
            const int n_nmt = last ? cur.NMT : NMT, n_mt = last ? wave : mt + NWV;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int st = 0; st < NS; ++st) {
                CCVPE_MI_VMWAIT();                          // this step's DMA has landed
                f32x4 r[SK];
#pragma unroll
                for (int j = 0; j < SK; ++j) if (st * SK + j < KCH) r[j] = *reinterpret_cast<const f32x4*>(stg + j * 256 + rd_off);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // ... and is in registers before the next request may overwrite it
                __builtin_amdgcn_sched_barrier(0);          // (hipcc otherwise spreads the reads over the MFMA chain and the request goes out behind it)
                if (st + 1 < NS) { CCVPE_MI_DMA_A(x_rsrc, NMT, mt, st + 1); }
                else { CCVPE_MI_DMA_A(n_rsrc, n_nmt, n_mt, 0); }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < SK; ++j)
                    if (st * SK + j < KCH) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[st * SK + j][e], r[j][e], acc, 0, 0, 0);
                    }
            }
            CCVPE_MI_SCATTER(acc, mt);
        }
        if (wave >= NMT) { CCVPE_MI_DMA_A(cur.rsrc, cur.NMT, wave, 0); }   // (a wave without tiles here may have one in the next unit)
        CCVPE_MI_LOAD_TAPS(chn);
        CCVPE_MI_STAMP(1);
        __syncthreads();                                   // E image, taps and bias of this chunk complete; everybody holds the weights
        CCVPE_MI_STAMP(2);
        CCVPE_MI_DMA_W(chn);                               // lands under the depthwise phase
        // distributed squeeze (ticket.h): lane j < SQ of wave 0 fetches w1[j][ch0 .. ch0 + 15] (64 bytes) for the end of the item; every
        // other lane asks for an out-of-range offset (no traffic, no branch around loads)
        f32x4 sqw[4];
#pragma unroll
        for (int e = 0; e < 4; ++e)
            sqw[e] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(se_rsrc, sq_lane ? (unsigned)((lane * p.mid + ch0) * 4) : 0x80000000u, e * 16, 0));

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
        CCVPE_MI_STAMP(3);
        // channel sums: lanes with equal quad inside the wave (xor over lane bits 2..5), then the waves through LDS
#pragma unroll
        for (int off = 4; off < 64; off <<= 1)
#pragma unroll
            for (int e = 0; e < 4; ++e) pool[e] += __shfl_xor(pool[e], off);
        if (lane < 4) *reinterpret_cast<f32x4*>(red + wave * 16 + lane * 4) = pool;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the next chunk's weights (this wave's share) have landed
        __syncthreads();                                   // E image free again; wave partials and the weights visible
        if (wave == 0) {
            float s = 0.f;
            if (lane < 16) {
#pragma unroll
                for (int w = 0; w < NT / 64; ++w) s += red[w * 16 + lane];
            }
            if (!rows_on) {
                if (lane < 16) p.pool[(size_t)unit * p.mid + ch0 + lane] = s;   // [B][NST][mid]: one partial row per strip (launch_se reads them)
            } else {
                // this item's share of the squeeze conv (model.py:115 is linear in the pooled sums): lane j adds w1[j][ch0 + c] * sum[c]
                // over the item's 16 channels (sum[c] sits in lane c); one row of SQ floats per item, write-through for the ticket below
                float a = 0.f;
#pragma unroll
                for (int c = 0; c < 16; ++c) a = fmaf(sqw[c >> 2][c & 3], __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, s), c)), a);   // (the builtin moves 32 bits: integers by its prototype)
                if (lane < p.se.SQ) st_sc1(p.se.sqpart + ((size_t)b * p.se.per_sample + (it - b * p.se.per_sample)) * p.se.SQ + lane, a);
            }
        }
        // (red is rewritten only after the next item's first barrier, which wave 0 reaches after these reads)
        CCVPE_MI_STAMP(4);
    }
    // ---- squeeze-excite by ticket (ticket.h), behind the loop: one ticket per sample this workgroup's contiguous share of the item list
    // touched (how many of the sample's NST x nchunks items were ours follows from the share's bounds); whoever completes a sample computes
    // its gates.  Inside the loop a ticket cost the front kernels 10-20 % at batch 32 (its store drain also waits for the next item's
    // prefetched operands, its barriers stop the workgroup twice per sample boundary); here every store has long left.  The whole dynamic
    // LDS is free (the last item's re-requested operands have landed behind the first ticket's drain).
    if (se_on) {
        const int b_lo = it_lo / p.se.per_sample, b_hi = (it_hi - 1) / p.se.per_sample;
        if (b_lo == b_hi && p.se.spec) {   // latency plans: the excite weights are requested before the ticket is drawn
            se_arrive_and_finish_parts_spec<NT>(p.se, b_lo, (unsigned)(it_hi - it_lo), Es);
        } else {
            for (int bb = b_lo; bb <= b_hi; ++bb) {
                const int n = min(it_hi, (bb + 1) * p.se.per_sample) - max(it_lo, bb * p.se.per_sample);
                if (ticket_arrive(p.se.counter + bb, (unsigned)n, (unsigned)p.se.per_sample, reinterpret_cast<unsigned*>(Es + SE_SCR_FLAG))) {
                    se_finish_parts<NT>(p.se, bb, Es);
                    __syncthreads();
                }
            }
        }
    }
#if CCVPE_MI_CLOCK
    if (lane == 0) {
        for (int i = 0; i < 5; ++i) atomicAdd(&g_mi_clk[i], clk[i]);
        atomicAdd(&g_mi_clk[5], 1ull);
        atomicAdd(&g_mi_clk[6], (unsigned long long)(it_hi - it_lo));
        atomicAdd(&g_mi_clk[7], clkw);
    }
#endif
#undef CCVPE_MI_DMA_W
#undef CCVPE_MI_DMA_A
#undef CCVPE_MI_LOAD_TAPS
#undef CCVPE_MI_SCATTER
#undef CCVPE_MI_VMWAIT
}

// Geometry for RO output rows per strip (RO a multiple of ty, or the whole image), nwv waves per workgroup and sk 16-channel groups
// per staging step.
static size_t img_geometry_ro(const MbFrontParams& p, int tx, int ty, int ro, int nwv, int sk, MbImgParams& q) {
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
    q.CG = 0;
    const int kch = p.cinp / 16;
    return ((size_t)(q.HPa * q.WPa + 1) * EPS + 2 * (size_t)q.NMT * 16 + (size_t)p.k * p.k * 16 + 16 + (size_t)nwv * 16 + (size_t)(kch + nwv * sk) * 256) * sizeof(float);
}

// Patch per thread: 4 x 2 outputs (13 LDS reads per output for a 5x5 layer) from 512 pixels up; 4 x 1 on the smaller stride-1 images
// (blocks 12-15: 32 patches of 4 x 2 keep a quarter of the waves busy in the depthwise phase; 16 reads per output, 65 -> 61 us at batch
// 32; 2 x 1 measured no better); 2 x 1 for the small stride-2 images.
static void patch_sel(int oh, int ow, int s, int& tx, int& ty) {
    if (oh * ow >= 512) { tx = 4; ty = 2; }
    else if (s == 1) { tx = 4; ty = 1; }
    else { tx = 2; ty = 1; }
}

struct ImgPlan {
    MbImgParams q;
    size_t lds;
    int tx, ty, nt, sk;        // patch, threads per workgroup, 16-channel groups per staging step (kch = whole tiles)
};

// Whole image if it fits the LDS of a CU, else the fewest equal strips that do (their halo rows are expanded twice).  Per strip count:
// whole-tile staging, else half tiles; two 256-thread workgroups per CU where both fit (out of phase: one's depthwise phase beside the
// other's expand phase), else one of 512.
static void img_plan_spread(const MbFrontParams& p, ImgPlan& pl);
static bool img_plan(const MbFrontParams& p, ImgPlan& pl) {
    const size_t cap = 158 * 1024;
    const int kch = p.cinp / 16;
    patch_sel(p.OH, p.OW, p.s, pl.tx, pl.ty);
    for (int nst = 1; nst <= p.OH; ++nst) {
        int ro = (p.OH + nst - 1) / nst;
        ro = (ro + pl.ty - 1) / pl.ty * pl.ty;
        const int hk = (kch + 1) / 2;
        // latency plans (round 4): when the (sample, strip, chunk) items do not even fill the 256 CUs once, two workgroups per CU buy
        // nothing - eight waves per item halve its m-tiles per wave, and the squeeze-excite tail of the last arriver has twice the loads in flight
        const bool few = (long long)p.B * ((p.OH + ro - 1) / ro) * (p.mid / 16) <= 256;
        if (few && img_geometry_ro(p, pl.tx, pl.ty, ro, 8, kch, pl.q) <= cap) { pl.nt = 512; pl.sk = kch; }
        else if (2 * img_geometry_ro(p, pl.tx, pl.ty, ro, 4, kch, pl.q) <= cap) { pl.nt = 256; pl.sk = kch; }
        else if (kch > 2 && 2 * img_geometry_ro(p, pl.tx, pl.ty, ro, 4, hk, pl.q) <= cap) { pl.nt = 256; pl.sk = hk; }
        else if (img_geometry_ro(p, pl.tx, pl.ty, ro, 8, kch, pl.q) <= cap) { pl.nt = 512; pl.sk = kch; }
        else if (kch > 2 && img_geometry_ro(p, pl.tx, pl.ty, ro, 8, hk, pl.q) <= cap) { pl.nt = 512; pl.sk = hk; }
        else { if (ro <= pl.ty) break; continue; }
        pl.lds = img_geometry_ro(p, pl.tx, pl.ty, ro, pl.nt / 64, pl.sk, pl.q);
        if ((ro - 1) * p.s + p.k > 4 * ro * p.s) return false;   // not worth it once the halo is 4x the strip
        img_plan_spread(p, pl);
        return true;
    }
    return false;
}

// Latency plans (round 4).  With one sample a block is 15-72 chunks - as many workgroups on 256 CUs - and the expand GEMM of a chunk runs
// on ONE CU's matrix pipe: 256 pixels x 192 channels x 16 = 6100 cycles of fp32 MFMAs, 1024 x 112 x 16 = 14300 (in-kernel stamps, batch 1:
// the expand phase is 50-65 % of a workgroup's 7-15 us and under 1 % of it waits for memory).  More, shorter strips put the same chunk on
// several CUs: the halo rows are expanded twice, the chip has the room.  Taken when the items still fit the chip in one round and a strip
// reads at least a fifth fewer rows than the plan above.
static void img_plan_spread(const MbFrontParams& p, ImgPlan& pl) {
    const int cap_items = p.spread;   // items a launch may spread to - half the chip by default: the other stream runs the other encoder's front
                                      // beside it (256: 1.25, 192: 1.22, 128: 1.20 ms per batch-1 frame, 0 = off: 1.27)
    const int chunks = p.mid / 16, kch = p.cinp / 16;
    if ((long long)p.B * pl.q.NST * chunks >= cap_items) return;
    const size_t cap = 158 * 1024;
    int best_rows = std::min(p.H, (pl.q.RO - 1) * p.s + p.k);
    for (int nst = pl.q.NST + 1; nst <= p.OH; ++nst) {
        const int ro0 = (p.OH + nst - 1) / nst;
        int tx, ty;
        patch_sel(ro0, p.OW, p.s, tx, ty);
        const int ro = (ro0 + ty - 1) / ty * ty;
        const int n = (p.OH + ro - 1) / ro;
        if ((long long)p.B * n * chunks > cap_items) break;
        const int rows = std::min(p.H, (ro - 1) * p.s + p.k);
        if (rows * 5 > best_rows * 4 || rows > 4 * ro * p.s) continue;
        ImgPlan c = pl;
        c.tx = tx; c.ty = ty; c.nt = 512; c.sk = kch;
        if (img_geometry_ro(p, tx, ty, ro, 8, kch, c.q) > cap) continue;
        c.lds = img_geometry_ro(p, tx, ty, ro, 8, kch, c.q);
        pl = c;
        best_rows = rows;
    }
}

// Cin need not be a multiple of 16: the expand weights are zero padded to cinp and the last 16-byte pieces of a pixel then
// read the first channels of the next pixel (finite activations; past the end the buffer descriptor returns zeros).
bool mbconv_image_supported(const MbFrontParams& p) {
    const int kch = p.cinp / 16;
    if (p.Cin % 8 != 0 || p.cinp % 16 != 0 || p.cinp < p.Cin || p.mid % 16 != 0 || p.W < 2 * p.k) return false;
    const bool combo = (p.k == 3 && p.s == 1 && (kch == 2 || kch == 5 || kch == 12)) || (p.k == 5 && p.s == 1 && (kch == 3 || kch == 5 || kch == 7 || kch == 12)) ||
                       (p.k == 5 && p.s == 2 && (kch == 2 || kch == 7)) || (p.k == 3 && p.s == 2 && kch == 3);
    if (!combo) return false;
    ImgPlan pl;
    if (!img_plan(p, pl)) return false;
    // measured (batch 32): a stride-2 layer wider than 128 pixels gets strips of only 4 output rows (1.4x halo rows, items too
    // short for their two barriers): ground block 3 (80 x 160) 0.195 ms fused against 0.182 ms as two launches - left unfused
    // (batch <= 4 fuses it all the same: there the three launches of the unfused form - expand GEMM, depthwise, squeeze-excite - are what
    //  costs, 48 us against 25 us for the aerial encoder's block 3 at batch 1)
    if (pl.q.NST > 1 && p.s == 2 && p.W > 128 && p.B > 4) return false;
    return true;
}

int mbconv_image_strips(const MbFrontParams& p) {
    ImgPlan pl;
    return (mbconv_image_supported(p) && img_plan(p, pl)) ? pl.q.NST : 0;
}

// The ticket's scratch is the E image (free between two items): the launch takes a ticket when that image is large enough
int mbconv_image_ticket_rows(const MbFrontParams& p) {
    ImgPlan pl;
    if (!mbconv_image_supported(p) || !img_plan(p, pl)) return 0;
    if ((pl.q.HPa * pl.q.WPa + 1) * EPS < SE_SCRATCH_PARTS_FLOATS || p.mid > 1152 || p.mid % 4 != 0) return 0;
    return pl.q.NST;
}

template <int K, int S, int KCH, int SK, int TX, int TY, int NT>
static void launch_img(const MbFrontParams& p, const ImgPlan& pl, hipStream_t s) {
    const MbImgParams& q = pl.q;
    static LdsAttr attr;
    auto kern = mbconv_image_kernel<K, S, KCH, SK, TX, TY, NT>;
    ensure_dynamic_lds(attr, reinterpret_cast<const void*>(kern), pl.lds);
    const int total = std::max(1, p.B * q.NST * q.nchunks);
#if CCVPE_MI_CLOCK
    static int calls = 0;
    const bool stamp = ++calls % 8 == 0;
    if (stamp) { (void)hipStreamSynchronize(s); unsigned long long z[12] = {}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_mi_clk), z, sizeof z); }
#endif
    CCVPE_LAUNCH(kern, dim3(std::min(total, NT == 512 ? 256 : 512)), dim3(NT), pl.lds, s, q);   // persistent: 8 waves per CU either way
#if CCVPE_SE_CLOCK
    if (p.se.counter && p.se.spec && p.mid >= 480) {
        (void)hipStreamSynchronize(s);
        unsigned long long h[16];
        (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_se_clk), sizeof h);
        std::fprintf(stderr, "se tail mid %d: drain+barrier %.2f  ticket %.2f  reset+acquire %.2f  barrier %.2f  sums+excite %.2f us\n", p.mid, (h[1] - h[0]) * 0.01, (h[2] - h[1]) * 0.01,
                     (h[3] - h[2]) * 0.01, (h[4] - h[3]) * 0.01, (h[5] - h[4]) * 0.01);
    }
#endif
#if CCVPE_MI_CLOCK
    if (stamp) {
        (void)hipStreamSynchronize(s);
        unsigned long long h[12];
        (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_mi_clk), sizeof h);
        const double w = (double)std::max(1ull, h[5]), tot = (double)(h[0] + h[1] + h[2] + h[3] + h[4]);
        std::fprintf(stderr, "mbconv_image<%d,%d,%d,%d,%d,%d,%d> %dx%d cin %d mid %d strips %d lds %zu: %.1f items/workgroup, %.0f cycles/wave: set-up %.0f %% expand %.0f %% barrier %.0f %% depthwise %.0f %% tail %.0f %%; %.0f cycles/item, %.0f of them waiting for staged m-tiles\n",
                     K, S, KCH, SK, TX, TY, NT, p.H, p.W, p.Cin, p.mid, q.NST, pl.lds, (double)h[6] / w, tot / w, 100.0 * h[0] / tot, 100.0 * h[1] / tot, 100.0 * h[2] / tot, 100.0 * h[3] / tot,
                     100.0 * h[4] / tot, tot / std::max(1.0, (double)h[6]), (double)h[7] / std::max(1.0, (double)h[6]));
    }
#endif
}

template <int K, int S, int KCH, int TX, int TY>
static void launch_img_t(const MbFrontParams& p, const ImgPlan& pl, hipStream_t s) {
    constexpr int HK = (KCH + 1) / 2;
    const bool half = pl.sk != KCH;
    if (!half) { if (pl.nt == 256) launch_img<K, S, KCH, KCH, TX, TY, 256>(p, pl, s); else launch_img<K, S, KCH, KCH, TX, TY, 512>(p, pl, s); }
    else if constexpr (KCH > 2) { if (pl.nt == 256) launch_img<K, S, KCH, HK, TX, TY, 256>(p, pl, s); else launch_img<K, S, KCH, HK, TX, TY, 512>(p, pl, s); }
}

template <int K, int S, int KCH>
static void launch_img_p(const MbFrontParams& p, hipStream_t s) {
    ImgPlan pl;
    img_plan(p, pl);
    if (pl.tx == 4 && pl.ty == 1) { if constexpr (S == 1) launch_img_t<K, S, KCH, 4, 1>(p, pl, s); }
    else if (pl.tx == 4) launch_img_t<K, S, KCH, 4, 2>(p, pl, s);
    else launch_img_t<K, S, KCH, 2, 1>(p, pl, s);
}

void launch_mbconv_image(const MbFrontParams& p, hipStream_t s) {
    const int kch = p.cinp / 16;
    if (p.k == 3 && p.s == 1) { if (kch == 2) launch_img_p<3, 1, 2>(p, s); else if (kch == 5) launch_img_p<3, 1, 5>(p, s); else launch_img_p<3, 1, 12>(p, s); }
    else if (p.k == 5 && p.s == 1) { if (kch == 3) launch_img_p<5, 1, 3>(p, s); else if (kch == 5) launch_img_p<5, 1, 5>(p, s); else if (kch == 7) launch_img_p<5, 1, 7>(p, s); else launch_img_p<5, 1, 12>(p, s); }
    else if (p.k == 5) { if (kch == 2) launch_img_p<5, 2, 2>(p, s); else launch_img_p<5, 2, 7>(p, s); }
    else launch_img_p<3, 2, 3>(p, s);
}

}  // namespace ccvpe
