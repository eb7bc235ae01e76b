// Fused MBConv front half: 1x1 expand (+BN+swish) -> depthwise k x k (+BN+swish) -> SE pooling partials,
// for the large-spatial blocks (1..4) of EfficientNet-B0 where the 6x-expanded tensor dominates HBM traffic.
//
// Reference: MBConvBlock.forward, efficientnet_pytorch/model.py:103-110 (expand conv + bn0 + swish, static
// same/circular padding utils.py:254-358, depthwise conv + bn1 + swish) and :114 (global average pool).
//
// Unfused, block 1 writes a 96 x 256 x 256 fp32 tensor per aerial image (25 MB) only to read it back once.
// Here a workgroup owns an output tile of the depthwise conv: the input pixels it depends on (tile + halo,
// Cin <= 48 channels) are staged in LDS once; for each chunk of 48 expanded channels the expand GEMM
// [pixels x Cin] x [Cin x 48] runs on v_mfma_f32_16x16x4_f32 into an LDS tile (positions outside the image are
// forced to 0: the reference zero-pads the EXPANDED activation; horizontally-circular layers wrap instead),
// the depthwise conv reads that tile on the VALU, and only the depthwise output leaves the chip together with
// deterministic per-tile channel sums for the squeeze-excite pool.  The halo is recomputed (1.2-2.3x expand
// FLOPs, which are negligible next to the saved traffic).
#include "kernels.h"

#include <algorithm>
#include <cstdlib>

namespace ccvpe {

typedef float f32x4 __attribute__((ext_vector_type(4)));

static constexpr int MC = 48;          // expanded channels per chunk (96, 144, 240 are multiples of 48)
static constexpr int ES = MC + 4;      // floats per pixel in the E tile (208 B: conflict-free b128)

__device__ __forceinline__ float swish_f(float v) { return v * __builtin_amdgcn_rcpf(1.f + __expf(-v)); }   // v_rcp_f32, not an IEEE division

// TW x TH output tile; input tile IW x IH = ((TW-1)*S + K) x ((TH-1)*S + K)
template <int K, int S, int TW, int TH>
__global__ __launch_bounds__(256) void mbconv_front_kernel(const MbFrontParams p) {
    constexpr int IW = (TW - 1) * S + K, IH = (TH - 1) * S + K;
    constexpr int NIP = IW * IH;                 // input pixels in the tile
    constexpr int NMT = (NIP + 15) / 16;         // m-tiles of the expand GEMM
    constexpr int NOP = TW * TH;                 // output pixels
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int CP = p.cinp;                       // Cin padded to 16
    const int XS = CP + 4;
    float* Xs = smem;                            // [NIP][XS]
    float* Ws = Xs + NIP * XS;                   // [MC][XS]   expand weights of the current chunk
    float* Es = Ws + MC * XS;                    // [NIP][ES]
    float* Rs = Es + NIP * ES;                   // [16][MC]   pooling reduction
    unsigned char* Vs = reinterpret_cast<unsigned char*>(Rs + 16 * MC);   // [NIP] 1 = inside the image

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_x = (p.OW + TW - 1) / TW;
    const int tile = blockIdx.x, b = blockIdx.y;
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int iy0 = oy0 * S - p.pad_t, ix0 = ox0 * S - p.pad_l;

    // ---- stage the input tile (zero outside; horizontal wrap for circular layers) ----
    const int c4n = CP >> 2;
    for (int i = tid; i < NIP * c4n; i += 256) {
        const int px = i / c4n, c4 = i - px * c4n;
        const int iy = iy0 + px / IW;
        int ix = ix0 + px % IW;
        bool ok = (unsigned)iy < (unsigned)p.H;
        if (p.circular) { if (ix < 0) ix += p.W; else if (ix >= p.W) ix -= p.W; }
        ok = ok && (unsigned)ix < (unsigned)p.W;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (ok && c4 * 4 < p.Cin) v = *reinterpret_cast<const f32x4*>(p.x + (((size_t)b * p.H + iy) * p.W + ix) * p.Cin + c4 * 4);
        *reinterpret_cast<f32x4*>(Xs + px * XS + c4 * 4) = v;
        if (c4 == 0) Vs[px] = ok ? 1 : 0;
    }

    const int kch = CP >> 4;
    // depthwise thread mapping: 192 threads = 12 channel groups x 16 pixel slots (fixed group per thread so the
    // pooling sums stay private and deterministic)
    const int dg = tid % 12, dslot = tid / 12;
    const bool dw_active = tid < 192;

    for (int ch0 = 0; ch0 < p.mid; ch0 += MC) {
        // expand weights of this chunk -> LDS
        for (int i = tid; i < MC * c4n; i += 256) {
            const int n = i / c4n, c4 = i - n * c4n;
            *reinterpret_cast<f32x4*>(Ws + n * XS + c4 * 4) = *reinterpret_cast<const f32x4*>(p.we + (size_t)(ch0 + n) * CP + c4 * 4);
        }
        __syncthreads();
        // ---- expand GEMM: units = (m-tile, n-tile of 16 channels), two chains per wave ----
        for (int u0 = wave; u0 < NMT * 3; u0 += 8) {
            const int u1 = u0 + 4;
            const int mt0 = u0 / 3, nt0 = u0 - mt0 * 3;
            const int mt1 = u1 / 3, nt1 = u1 - mt1 * 3;
            const float* a0p = Xs + min(mt0 * 16 + (lane & 15), NIP - 1) * XS + 4 * (lane >> 4);
            const float* a1p = Xs + min(mt1 * 16 + (lane & 15), NIP - 1) * XS + 4 * (lane >> 4);
            const float* b0p = Ws + (nt0 * 16 + (lane & 15)) * XS + 4 * (lane >> 4);
            const float* b1p = Ws + (min(nt1, 2) * 16 + (lane & 15)) * XS + 4 * (lane >> 4);
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
            for (int kc = 0; kc < kch; ++kc) {
                const f32x4 a0 = *reinterpret_cast<const f32x4*>(a0p + kc * 16);
                const f32x4 a1 = *reinterpret_cast<const f32x4*>(a1p + kc * 16);
                const f32x4 w0 = *reinterpret_cast<const f32x4*>(b0p + kc * 16);
                const f32x4 w1 = *reinterpret_cast<const f32x4*>(b1p + kc * 16);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, w0.x, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, w1.x, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, w0.y, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, w1.y, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, w0.z, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, w1.z, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, w0.w, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, w1.w, acc1, 0, 0, 0);
            }
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                const int u = h2 ? u1 : u0;
                if (u >= NMT * 3) continue;
                const int mt = h2 ? mt1 : mt0, nt = h2 ? nt1 : nt0;
                const f32x4 acc = h2 ? acc1 : acc0;
                const float be = p.be[ch0 + nt * 16 + (lane & 15)];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int px = mt * 16 + (lane >> 4) * 4 + r;
                    if (px < NIP) Es[px * ES + nt * 16 + (lane & 15)] = Vs[px] ? swish_f(acc[r] + be) : 0.f;
                }
            }
        }
        __syncthreads();
        // ---- depthwise k x k from the E tile, BN + swish, store, pooling partial ----
        if (dw_active) {
            // thread = (channel group dg, strip of PPT adjacent output pixels of one row): every filter tap's weight
            // is loaded once per chunk and the strip's input columns are read from LDS once per filter row
            constexpr int PPT = NOP / 16;                 // 4 (8x8), 2 (8x4) or 1 (8x2)
            constexpr int SPR = TW / PPT;                 // strips per row
            constexpr int NCOL = (PPT - 1) * S + K;
            const int c = ch0 + dg * 4;
            const f32x4 bd = *reinterpret_cast<const f32x4*>(p.bd + c);
            const int oyl = dslot / SPR, oxl = (dslot % SPR) * PPT;
            f32x4 acc[PPT];
#pragma unroll
            for (int i = 0; i < PPT; ++i) acc[i] = bd;
            const float* ep = Es + ((oyl * S) * IW + oxl * S) * ES + dg * 4;
#pragma unroll
            for (int ky = 0; ky < K; ++ky) {
                f32x4 col[NCOL];
#pragma unroll
                for (int j = 0; j < NCOL; ++j) col[j] = *reinterpret_cast<const f32x4*>(ep + (ky * IW + j) * ES);
#pragma unroll
                for (int kx = 0; kx < K; ++kx) {
                    const f32x4 w = *reinterpret_cast<const f32x4*>(p.wd + (size_t)(ky * K + kx) * p.mid + c);
#pragma unroll
                    for (int i = 0; i < PPT; ++i) acc[i] += col[i * S + kx] * w;
                }
            }
            f32x4 pool = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < PPT; ++i) {
                const int oy = oy0 + oyl, ox = ox0 + oxl + i;
                if (oy < p.OH && ox < p.OW) {
                    f32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = swish_f(acc[i][e]);
                    pool += o;
                    *reinterpret_cast<f32x4*>(p.out + (((size_t)b * p.OH + oy) * p.OW + ox) * p.mid + c) = o;
                }
            }
            *reinterpret_cast<f32x4*>(Rs + dslot * MC + dg * 4) = pool;
        }
        __syncthreads();
        if (tid < MC) {
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) s += Rs[i * MC + tid];
            p.pool[((size_t)b * gridDim.x + tile) * p.mid + ch0 + tid] = s;
        }
        // the next chunk's first barrier (after the weight load) orders these reads before Rs/Es/Ws are rewritten
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Wave-local form (round 2).  The kernel above runs three workgroup barriers per 48-channel chunk of a 32-pixel tile and
// measured 7x off its HBM roofline (block 1 of the aerial encoder: 0.41 ms for 335 MB).  Here ONE WAVE owns an output tile
// for all expanded channels: it stages its input tile (+ halo) in its own LDS region, then per chunk of 16 expanded channels
// runs the expand GEMM on v_mfma_f32_16x16x4_f32 into its own E tile, the depthwise conv from that tile, the store and the
// pooling partial - nothing but s_waitcnt between the stages, no barrier anywhere, eight single-wave workgroups per CU.
// Same tile shapes as above, so the pooling partial layout [B][tiles][mid] is unchanged.  The input tile never touches LDS: it
// is loaded straight into the MFMA A-operand registers (lane (l & 15, l >> 4) = pixel l & 15 of every m-tile, channels
// 4 (l >> 4) .. + 3 of every 16-channel step), so the 12.8 KB E tile is all the LDS a wave needs and 8-12 waves fit a CU
// (a first version with the input tile in LDS ran 6 waves per CU and was slower than the workgroup form).
// Measured at batch 32: block 1 of the aerial encoder 0.374 -> 0.218 ms, the six fused launches 1.38 -> 0.95 ms per step.
// ---------------------------------------------------------------------------------------------------------------------
static constexpr int ES2 = 20;        // floats per pixel of the 16-channel E tile (conflict-free epilogue writes and b128 reads)

// NWG = 1: one wave per workgroup, one pooling partial row per tile (squeeze-excite as its own launches).  NWG = 4 (round 4, the
// ticket form): four INDEPENDENT waves per workgroup - still no barrier in the tile loop - whose channel sums meet in LDS, so a
// workgroup leaves ONE partial row and draws ONE squeeze-excite ticket (ticket.h); a sample's last workgroup computes its gates.
template <int K, int S, int TW, int TH, int KCH, int NWG>
__global__ __launch_bounds__(64 * NWG, KCH == 1 ? 3 : 2) void mbconv_front_wave_kernel(const MbFrontParams p) {
    constexpr int IW = (TW - 1) * S + K, IH = (TH - 1) * S + K;
    constexpr int NIP = IW * IH;
    constexpr int NMT = (NIP + 15) / 16;
    constexpr int NOP = TW * TH;
    constexpr int OPL = NOP / 16;            // outputs per lane slot (2 for 8x4, 4 for 8x8)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int CP = KCH * 16;             // input channels padded to 16
    constexpr int ESZ = NMT * 16 * ES2;      // floats of one wave's E tile
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* Es = smem + wave * ESZ;           // [NMT * 16][ES2] (rows >= NIP: sink of the last m-tile's padding rows)
    float* psum = smem + NWG * ESZ;          // NWG > 1: [NWG][mid] channel sums of the workgroup's tiles

    const int tiles_x = (p.OW + TW - 1) / TW;
    const int tiles = tiles_x * ((p.OH + TH - 1) / TH);
    const int tile = blockIdx.x * NWG + wave, b = blockIdx.y;
    const bool live = tile < tiles;          // (ragged last workgroup: a wave without a tile contributes zeros)
    // latency plans (NWG > 1, gridDim.z > 1): the 16-channel chunks of a tile are dealt to gridDim.z workgroups - a wave then walks
    // 2 chunks instead of 6 (block 1 at batch 1: every chunk is a chain of weight, tap and LDS latencies nothing else on the CU hides)
    const int nch = p.mid >> 4;
    const int c_lo = NWG > 1 ? 16 * (int)(((long long)nch * blockIdx.z) / gridDim.z) : 0;
    const int c_hi = NWG > 1 ? 16 * (int)(((long long)nch * (blockIdx.z + 1)) / gridDim.z) : p.mid;
    if (NWG > 1 && !live)
        for (int c = c_lo + lane; c < c_hi; c += 64) psum[wave * p.mid + c] = 0.f;
    if (live) {
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int iy0 = oy0 * S - p.pad_t, ix0 = ox0 * S - p.pad_l;

    // ---- input tile straight into the MFMA A-operand registers: lane (l & 15, l >> 4) holds channels 16 kc + 4 (l >> 4) .. + 3
    // of pixel 16 mt + (l & 15) for every m-tile (zero outside the image; horizontal wrap for circular layers).  No LDS
    // copy of the input: the E tile is the only LDS the wave needs (12.8 KB -> twelve waves per CU).
    f32x4 xa[NMT][KCH];
#pragma unroll
    for (int mt = 0; mt < NMT; ++mt) {
        const int px = min(mt * 16 + (lane & 15), NIP - 1);
        const int iy = iy0 + px / IW;
        int ix = ix0 + px % IW;
        bool ok = (unsigned)iy < (unsigned)p.H;
        if (p.circular) { if (ix < 0) ix += p.W; else if (ix >= p.W) ix -= p.W; }
        ok = ok && (unsigned)ix < (unsigned)p.W;
        const float* src = p.x + (((size_t)b * p.H + iy) * p.W + ix) * p.Cin + 4 * (lane >> 4);
#pragma unroll
        for (int kc = 0; kc < KCH; ++kc) {
            const int c = kc * 16 + 4 * (lane >> 4);
            xa[mt][kc] = (ok && c < p.Cin) ? *reinterpret_cast<const f32x4*>(src + kc * 16) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    // validity of the NMT pixels this lane's accumulators cover (the reference zero-pads the EXPANDED activation).  The
    // expand weights are the A operand of the MFMAs, so the accumulators are channel-major: a lane holds channels
    // 4 (lane >> 4) .. + 3 of pixel 16 mt + (lane & 15) and writes them to the E tile as one 16-byte piece.
    unsigned vmask = 0;                          // bit mt
#pragma unroll
    for (int mt = 0; mt < NMT; ++mt) {
        const int px = mt * 16 + (lane & 15);
        const int iy = iy0 + px / IW;
        int ix = ix0 + px % IW;
        if (p.circular) { if (ix < 0) ix += p.W; else if (ix >= p.W) ix -= p.W; }
        const bool ok = px < NIP && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        vmask |= (ok ? 1u : 0u) << mt;
    }
    const int q = lane & 3, slot = lane >> 2;    // depthwise: channel quad q of the chunk, output slot (16 slots x OPL outputs)
    for (int ch0 = c_lo; ch0 < c_hi; ch0 += 16) {
        // ---- expand: E[px][n] = swish(sum_c X[px][c] We[ch0 + n][c] + be) for the NIP input pixels ----
        f32x4 wfrag[KCH];                        // B fragments: We[ch0 + (lane & 15)][16 kc + 4 (lane >> 4) + e]
#pragma unroll
        for (int kc = 0; kc < KCH; ++kc)
            wfrag[kc] = *reinterpret_cast<const f32x4*>(p.we + (size_t)(ch0 + (lane & 15)) * CP + kc * 16 + 4 * (lane >> 4));
        const f32x4 be = *reinterpret_cast<const f32x4*>(p.be + ch0 + 4 * (lane >> 4));
        // four m-tiles at a time: four independent accumulator chains hide the 40-cycle dependent-MFMA latency
#pragma unroll
        for (int mt0 = 0; mt0 < NMT; mt0 += 4) {
            f32x4 acc[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kc = 0; kc < KCH; ++kc)
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (mt0 + u < NMT) acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(wfrag[kc][e], xa[mt0 + u][kc][e], acc[u], 0, 0, 0);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int mt = mt0 + u;
                if (mt >= NMT) continue;
                const float keep = (float)((vmask >> mt) & 1u);   // a multiply, not a branch around the swish
                f32x4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = swish_f(acc[u][r] + be[r]) * keep;
                *reinterpret_cast<f32x4*>(Es + (mt * 16 + (lane & 15)) * ES2 + 4 * (lane >> 4)) = v;   // rows >= NIP: sink rows
            }
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0) only: LDS traffic of this wave is done; global loads / stores stay in flight
        __builtin_amdgcn_wave_barrier();
        // ---- depthwise K x K from the E tile, BN + swish, store, pooling partial ----
        const int c = ch0 + q * 4;
        const f32x4 bd = *reinterpret_cast<const f32x4*>(p.bd + c);
        f32x4 wk[K * K];
#pragma unroll
        for (int t = 0; t < K * K; ++t) wk[t] = *reinterpret_cast<const f32x4*>(p.wd + (size_t)t * p.mid + c);
        f32x4 pool = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < OPL; ++i) {
            const int o = slot + 16 * i;
            const int oyl = o / TW, oxl = o - oyl * TW;
            const float* ep = Es + ((oyl * S) * IW + oxl * S) * ES2 + q * 4;
            f32x4 acc = bd;
#pragma unroll
            for (int ky = 0; ky < K; ++ky)
#pragma unroll
                for (int kx = 0; kx < K; ++kx) acc += *reinterpret_cast<const f32x4*>(ep + (ky * IW + kx) * ES2) * wk[ky * K + kx];
            const int oy = oy0 + oyl, ox = ox0 + oxl;
            if (oy < p.OH && ox < p.OW) {
                f32x4 ov;
#pragma unroll
                for (int e = 0; e < 4; ++e) ov[e] = swish_f(acc[e]);
                pool += ov;
                *reinterpret_cast<f32x4*>(p.out + (((size_t)b * p.OH + oy) * p.OW + ox) * p.mid + c) = ov;
            }
        }
        // sum over the 16 slots (lanes with equal q): xor-shuffles over lane bits 2..5; fixed order -> deterministic
#pragma unroll
        for (int off = 4; off < 64; off <<= 1)
#pragma unroll
            for (int e = 0; e < 4; ++e) pool[e] += __shfl_xor(pool[e], off);
        if (slot == 0) {
            if (NWG > 1) *reinterpret_cast<f32x4*>(psum + wave * p.mid + c) = pool;
            else *reinterpret_cast<f32x4*>(p.pool + ((size_t)b * gridDim.x + tile) * p.mid + c) = pool;
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): the E tile is rewritten by the next chunk's expand (the output stores stay in flight)
        __builtin_amdgcn_wave_barrier();
    }
    }   // live
    if constexpr (NWG > 1) {
        // one partial row per workgroup (waves in wave order: deterministic), then the squeeze-excite ticket; the E tiles are free
        __syncthreads();
        for (int c = c_lo + threadIdx.x; c < c_hi; c += 64 * NWG) {   // (the workgroups of a tile group write disjoint channels of one row)
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < NWG; ++w) s += psum[w * p.mid + c];
            st_sc1(p.pool + ((size_t)b * gridDim.x + blockIdx.x) * p.mid + c, s);
        }
        if (p.se.counter != nullptr) {
            // (the ticket's first barrier also orders the psum reads above before the scratch writes of se_finish)
            if (ticket_arrive(p.se.counter + b, 1u, (unsigned)p.se.per_sample, reinterpret_cast<unsigned*>(smem + SE_SCR_FLAG)))
                se_finish<64 * NWG>(p.se, b, smem);
        }
    }
}

static constexpr int WAVE_NWG = 4;   // waves per workgroup of the ticket form
int mbconv_front_ticket_split(const MbFrontParams& p);

template <int K, int S, int TW, int TH, int KCH>
static void launch_mb_wave(const MbFrontParams& p, hipStream_t s) {
    constexpr int IW = (TW - 1) * S + K, IH = (TH - 1) * S + K, NIP = IW * IH;
    const size_t esz = (size_t)((NIP + 15) / 16 * 16) * ES2 * sizeof(float);
    const int tiles = ((p.OW + TW - 1) / TW) * ((p.OH + TH - 1) / TH);
    if (p.se.counter != nullptr) {   // ticket form: four waves per workgroup, one pooling partial row each
        const size_t lds = std::max(WAVE_NWG * esz + (size_t)WAVE_NWG * p.mid * sizeof(float), (size_t)SE_SCRATCH_FLOATS * sizeof(float));
        static LdsAttr attr4;
        auto kern = mbconv_front_wave_kernel<K, S, TW, TH, KCH, WAVE_NWG>;
        ensure_dynamic_lds(attr4, reinterpret_cast<const void*>(kern), lds);
        CCVPE_LAUNCH(kern, dim3((tiles + WAVE_NWG - 1) / WAVE_NWG, p.B, mbconv_front_ticket_split(p)), dim3(64 * WAVE_NWG), lds, s, p);
        return;
    }
    static LdsAttr attr;
    auto kern = mbconv_front_wave_kernel<K, S, TW, TH, KCH, 1>;
    ensure_dynamic_lds(attr, reinterpret_cast<const void*>(kern), esz);
    CCVPE_LAUNCH(kern, dim3(tiles, p.B), dim3(64), esz, s, p);
}

template <int K, int S, int TW, int TH>
static void launch_mb(const MbFrontParams& p, hipStream_t s) {
    constexpr int IW = (TW - 1) * S + K, IH = (TH - 1) * S + K, NIP = IW * IH;
    const int XS = p.cinp + 4;
    const size_t lds = ((size_t)NIP * XS + (size_t)MC * XS + (size_t)NIP * ES + 16 * MC) * sizeof(float) + ((NIP + 15) & ~15);
    static LdsAttr attr;
    auto kern = mbconv_front_kernel<K, S, TW, TH>;
    ensure_dynamic_lds(attr, reinterpret_cast<const void*>(kern), lds);
    const int tiles = ((p.OW + TW - 1) / TW) * ((p.OH + TH - 1) / TH);
    CCVPE_LAUNCH(kern, dim3(tiles, p.B), dim3(256), lds, s, p);
}

// output tile per (k, stride): stride-1 blocks 8x8, stride-2 blocks 8x4 (k3) / 8x2 (k5) - their input tiles are
// (2T+k-2)^2-ish and must leave room for two workgroups per CU
static void tile_of(int k, int s, int& tw, int& th) { tw = 8; th = s == 1 ? 8 : (k == 3 ? 4 : 2); }

int mbconv_front_tiles(int k, int s, int OH, int OW) {
    int tw, th;
    tile_of(k, s, tw, th);
    return ((OW + tw - 1) / tw) * ((OH + th - 1) / th);
}

bool mbconv_front_supported(int k, int s, int cin, int mid) {
    return (k == 3 || k == 5) && (s == 1 || s == 2) && cin % 8 == 0 && cin <= 48 && mid % MC == 0;
}

// Measured on MI355X at batch 32 (round 1): the fused kernel wins for the 3x3 blocks (block 1: 0.41 vs 0.65 ms,
// block 2: 0.33 vs 0.35, block 5: 0.11 vs 0.13) and loses for the 5x5 blocks (block 3: 0.44 vs 0.30, block 4:
// 0.37 vs 0.19 - 25-tap depthwise from LDS plus 2.3x halo recompute make it latency-bound), so only k = 3 fuses
// by default.
bool mbconv_front_profitable(int k) { return k == 3; }

static bool wave_form_serves(const MbFrontParams& p) {
    static const bool wave_form = !(getenv("CCVPE_MBCONV_WAVE") && std::atoi(getenv("CCVPE_MBCONV_WAVE")) == 0);   // CCVPE_MBCONV_WAVE=0: workgroup form
    return wave_form && p.k == 3 && p.cinp <= 48 && p.cinp % 16 == 0 && p.mid % 16 == 0;
}

// only the wave-local form takes a squeeze-excite ticket (four tiles per workgroup share a pooling partial row)
int mbconv_front_ticket_rows(const MbFrontParams& p) {
    if (!wave_form_serves(p) || p.mid > 1152) return 0;
    return (mbconv_front_tiles(p.k, p.s, p.OH, p.OW) + WAVE_NWG - 1) / WAVE_NWG;
}

// ... and how many workgroups share a tile group's channels (each draws a ticket: SeTicket::per_sample = rows x this).  More than one only
// in latency plans: the launch then has ~384 workgroups of four waves, three per CU beside the other encoder's
int mbconv_front_ticket_split(const MbFrontParams& p) {
    const int rows = mbconv_front_ticket_rows(p);
    if (rows == 0 || p.spread <= 0 || (long long)rows * p.B > 192) return 1;
    return std::max(1, std::min(p.mid / 16, 3 * p.spread / (rows * p.B)));   // (four-wave workgroups: three fit a CU)
}

void launch_mbconv_front(const MbFrontParams& p, hipStream_t s) {
    if (wave_form_serves(p)) {
        const int kch = p.cinp / 16;
        if (p.s == 1) { if (kch == 1) launch_mb_wave<3, 1, 8, 8, 1>(p, s); else if (kch == 2) launch_mb_wave<3, 1, 8, 8, 2>(p, s); else launch_mb_wave<3, 1, 8, 8, 3>(p, s); }
        else { if (kch == 1) launch_mb_wave<3, 2, 8, 4, 1>(p, s); else if (kch == 2) launch_mb_wave<3, 2, 8, 4, 2>(p, s); else launch_mb_wave<3, 2, 8, 4, 3>(p, s); }
        return;
    }
    if (p.k == 3 && p.s == 1) launch_mb<3, 1, 8, 8>(p, s);
    else if (p.k == 3 && p.s == 2) launch_mb<3, 2, 8, 4>(p, s);
    else if (p.k == 5 && p.s == 1) launch_mb<5, 1, 8, 8>(p, s);
    else launch_mb<5, 2, 8, 2>(p, s);
}

}  // namespace ccvpe
