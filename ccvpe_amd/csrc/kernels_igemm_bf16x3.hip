// fp32-accurate implicit GEMM on the bf16 matrix cores ("bf16x3"):  every fp32 operand x is split into
// hi = bf16(x), lo = bf16(x - hi) and  a*b  is evaluated as  a_lo*b_hi + a_hi*b_lo + a_hi*b_hi  with fp32
// accumulation (v_mfma_f32_32x32x16_bf16 / v_mfma_f32_16x16x32_bf16).  The three partial products are exact
// in fp32 (8+8 significand bits), only a_lo*b_lo (<= 2^-16 relative) is dropped, so a K-term dot product
// carries a relative error of ~2^-16 per term instead of fp32's 2^-24 - measured end to end on the CCVPE
// decoder: 1.3e-5 of the logits' scale (the path's contract is 1e-3).  The bf16 matrix pipe runs at 16x the
// fp32-MFMA rate, so three products still leave up to 5.3x of matrix throughput.
//
// Opt-in (precision mode "bf16x3" of ccvpe_create / CCVPE_PRECISION): the default path stays exact fp32.
// Same gather / tiling / epilogue as kernels_igemm.hip; what changes:
//   * activations are split to (hi, lo) bf16 while they are staged into LDS (VALU, beside the MFMAs);
//     weights are split once at ccvpe_finalize_weights and stored as two bf16 planes [Npad][Kpad];
//   * LDS images are 64-byte rows (32 bf16) with a 16-byte-chunk XOR swizzle (chunk ^ ((row>>2)&3)) instead of
//     padding: conflict-free ds_read_b128 fragments and 2 workgroups per CU at 128x128.
#include "igemm_common.h"

#include <algorithm>

namespace ccvpe {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* LdsPtr;

template <int MT> struct MfmaB;
template <> struct MfmaB<32> {
    using acc_t = f32x16;
    static constexpr int NACC = 16;
    static constexpr int KSTEPS = 2;   // K = 16 per MFMA
    static __device__ __forceinline__ acc_t run(bf16x8 a, bf16x8 b, acc_t c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ int row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }
    static __device__ __forceinline__ int chunk(int ks, int lane) { return ks * 2 + (lane >> 5); }
};
template <> struct MfmaB<16> {
    using acc_t = f32x4;
    static constexpr int NACC = 4;
    static constexpr int KSTEPS = 1;   // K = 32 per MFMA
    static __device__ __forceinline__ acc_t run(bf16x8 a, bf16x8 b, acc_t c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ int row(int r, int lane) { return (lane >> 4) * 4 + r; }
    static __device__ __forceinline__ int chunk(int, int lane) { return lane >> 4; }
};

// byte offset of 16-byte chunk `ch` (0..3) of row `r` in a swizzled [rows][32 bf16] image (4 rows per 256-B bank
// row).  A ds_read_b128 is served in 16-lane groups {0-3,12-15,20-27}, ...: for the 32x32x16 fragment a group reads
// 16 rows at ONE chunk -> XOR with (r>>2)&3 spreads them; for the 16x16x32 fragment a group reads rows 0-3 and
// 12-15 at chunk c and rows 4-11 at chunk c+1 -> rotate by 2 for rows 8-15 instead (XOR would collide 2-way).
template <int MT>
__device__ __forceinline__ int swz(int r, int ch) {
    if (MT == 32) return r * 64 + ((ch ^ ((r >> 2) & 3)) << 4);
    return r * 64 + (((ch + ((r >> 3) & 1) * 2) & 3) << 4);
}

// SPLIT: the activation already lives in HBM as two bf16 planes (hi at p.in, lo at p.in + in_plane elements of
// bf16) written by the producing layer's epilogue - the K loop then carries no conversion VALU at all (with fp32
// inputs the split costs ~5 VALU instructions per MFMA and the kernel is VALU-issue bound).
template <int BM, int BN, int WGM, int WGN, int MT, bool GATE, bool SPLIT>
__global__ __launch_bounds__(256) void conv_igemm_bf16x3_kernel(const ConvParams p) {
    static_assert(WGM * WGN == 4, "4 waves per workgroup");
    constexpr int WM = BM / WGM, WN = BN / WGN;
    constexpr int TM = WM / MT, TN = WN / MT;
    static_assert(TM >= 1 && TN >= 1 && TM * MT * WGM == BM && TN * MT * WGN == BN, "tile must be whole MFMA tiles");
    constexpr int AR = BM / 32;          // float4 rows each thread stages for A (8 threads per 32-float row)
    constexpr int BRH = (BN + 63) / 64;  // 16-byte rows each thread stages per weight plane (4 threads per row)
    static_assert(!SPLIT || (BM % 64 == 0 && BN % 16 == 0), "LDS-DMA staging copies whole 16-row blocks");
    static_assert(!(GATE && SPLIT), "pre-split activations carry no squeeze-excite gate");
    using M = MfmaB<MT>;
    using acc_t = typename M::acc_t;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
    // stage layout: [Ah | Al | Bh | Bl], each [rows][64 B]
    constexpr int STAGE = (2 * BM + 2 * BN) * 64;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int m0 = xcd_remap(blockIdx.x, gridDim.x) * BM;
    const int n0 = blockIdx.y * BN;
    const int kq = tid & 7;
    const int r0 = tid >> 3;

    // SPLIT: one descriptor over both planes (hi | lo), element size 2 bytes
    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, p.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t gate_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(GATE ? p.gate : p.in), 0, GATE ? p.gate_bytes : 0, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;

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
        a_base[j] = (((b * p.H + iy0) * p.W + ix0) * p.in_ld + (kq & 1) * 4) * 4;
        a_iy[j] = ok ? iy0 : -(1 << 28);
        a_ix[j] = ix0;
        a_gb[j] = (b * p.Cin + (kq & 1) * 4) * 4;
    }
    // weight planes (and SPLIT activation planes): thread -> (row = tid>>2 (+64j), 16-byte chunk = tid&3)
    const int wch = tid & 3, wr0 = tid >> 2;
    // SPLIT (LDS-DMA staging): wave w copies 16-row blocks w, w+4, ... of every plane straight into LDS with
    // buffer_load_dwordx4 ... lds (no VGPR round trip, no ds_write): lane l fills row (l>>2), physical 16-byte
    // slot (l&3) of its block, i.e. the logical chunk the swizzle maps there (the swizzle is an involution)
    constexpr int ARD = BM / 64;             // A row blocks per wave (BM/16 blocks over 4 waves)
    constexpr int BRD = (BN / 16 + 3) / 4;   // B row blocks per wave (upper bound)
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    int s_base[SPLIT ? ARD : 1], s_iy[SPLIT ? ARD : 1], s_ix[SPLIT ? ARD : 1], s_ch[SPLIT ? ARD : 1];
    int w_off[SPLIT ? BRD : 1], w_ch[SPLIT ? BRD : 1];
    if constexpr (SPLIT) {
#pragma unroll
        for (int j = 0; j < ARD; ++j) {
            const int r = (wave_u + 4 * j) * 16 + (lane >> 2);
            const int m = m0 + r;
            const bool ok = m < p.M;
            const int mm = ok ? m : 0;
            const int b = mm / ohw;
            const int rem = mm - b * ohw;
            const int oy = rem / p.OW;
            const int ox = rem - oy * p.OW;
            const int iy0 = oy * p.stride - p.pad_t;
            const int ix0 = ox * p.stride - p.pad_l;
            s_base[j] = (((b * p.H + iy0) * p.W + ix0) * p.in_ld) * 2;   // bytes inside a bf16 plane
            s_iy[j] = ok ? iy0 : -(1 << 28);
            s_ix[j] = ix0;
            s_ch[j] = (swz<MT>(r, lane & 3) >> 4) & 3;                   // logical chunk stored at this slot
        }
#pragma unroll
        for (int j = 0; j < BRD; ++j) {
            const int r = (wave_u + 4 * j) * 16 + (lane >> 2);
            w_ch[j] = (swz<MT>(r, lane & 3) >> 4) & 3;
            w_off[j] = (min(n0 + r, p.Npad - 1) * p.Kpad + w_ch[j] * 8) * 2;   // bytes inside a weight plane
        }
    }
    const __amdgpu_buffer_rsrc_t wh_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.w_hi), 0, SPLIT ? p.w_plane_bytes : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t wl_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.w_lo), 0, SPLIT ? p.w_plane_bytes : 0, 0x00020000);
    const unsigned short* wrow_h[BRH];
    const unsigned short* wrow_l[BRH];
#pragma unroll
    for (int j = 0; j < BRH; ++j) {
        const size_t off = (size_t)min(n0 + wr0 + 64 * j, p.Npad - 1) * p.Kpad + wch * 8;
        wrow_h[j] = p.w_hi + off;
        wrow_l[j] = p.w_lo + off;
    }

    f32x4 ra[AR];
    f32x4 rg[GATE ? AR : 1];
    u32x4 rbh[BRH], rbl[BRH];

#define CCVPE_LOAD_TILE(kt)                                                                              \
    {                                                                                                    \
        if constexpr (!SPLIT) {                                                                          \
            const int g = (kt) * 4 + (kq >> 1);                                                          \
            int tap, c0;                                                                                 \
            chunk_to_tap(p, g, tap, c0);                                                                 \
            const int ky = (tap * p.div_kw_mul) >> 5;                                                    \
            const int kx = tap - ky * p.KW;                                                              \
            const bool gok = g < p.nchunks;                                                              \
            const int koff = ((ky * p.W + kx) * p.in_ld + c0) * 4;                                       \
            _Pragma("unroll") for (int j = 0; j < AR; ++j) {                                             \
                const int iy = a_iy[j] + ky, ix = a_ix[j] + kx;                                          \
                const bool ok = gok & ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W);   \
                const unsigned off = ok ? (unsigned)(a_base[j] + koff) : OOB;                            \
                ra[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, off, 0, 0)); \
                if (GATE) {                                                                              \
                    rg[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(gate_rsrc, ok ? (unsigned)(a_gb[j] + c0 * 4) : OOB, 0, 0)); \
                }                                                                                        \
            }                                                                                            \
        }                                                                                                \
        if constexpr (!SPLIT) {                                                                          \
            _Pragma("unroll") for (int j = 0; j < BRH; ++j) {                                            \
                rbh[j] = *reinterpret_cast<const u32x4*>(wrow_h[j] + (kt) * 32);                         \
                rbl[j] = *reinterpret_cast<const u32x4*>(wrow_l[j] + (kt) * 32);                         \
            }                                                                                            \
        }                                                                                                \
    }
// SPLIT: asynchronous global -> LDS copies of K tile `kt` into stage `st` (out-of-range offsets write zeros)
#define CCVPE_DMA_TILE(kt, st)                                                                           \
    {                                                                                                    \
        unsigned char* sb_ = smem_b + (st) * STAGE;                                                      \
        _Pragma("unroll") for (int j = 0; j < ARD; ++j) {                                                \
            const int g = (kt) * 4 + s_ch[j];                                                            \
            int tap, c0;                                                                                 \
            chunk_to_tap(p, g, tap, c0);                                                                 \
            const int ky = (tap * p.div_kw_mul) >> 5;                                                    \
            const int kx = tap - ky * p.KW;                                                              \
            const int iy = s_iy[j] + ky, ix = s_ix[j] + kx;                                              \
            const bool ok = (g < p.nchunks) & ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W); \
            const unsigned off = ok ? (unsigned)(s_base[j] + ((ky * p.W + kx) * p.in_ld + c0) * 2) : OOB; \
            LdsPtr dh = (LdsPtr)(sb_ + (wave_u + 4 * j) * 1024);                                         \
            LdsPtr dl = (LdsPtr)(sb_ + BM * 64 + (wave_u + 4 * j) * 1024);                               \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(in_rsrc, dh, 16, off, 0, 0, 0);                     \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(in_rsrc, dl, 16, ok ? off + p.in_plane_bytes : OOB, 0, 0, 0); \
        }                                                                                                \
        _Pragma("unroll") for (int j = 0; j < BRD; ++j) {                                                \
            if (wave_u + 4 * j < BN / 16) {                                                              \
                const unsigned off = (unsigned)(w_off[j] + (kt) * 64);                                   \
                LdsPtr dh = (LdsPtr)(sb_ + 2 * BM * 64 + (wave_u + 4 * j) * 1024);                       \
                LdsPtr dl = (LdsPtr)(sb_ + 2 * BM * 64 + BN * 64 + (wave_u + 4 * j) * 1024);             \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(wh_rsrc, dh, 16, off, 0, 0, 0);                 \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(wl_rsrc, dl, 16, off, 0, 0, 0);                 \
            }                                                                                            \
        }                                                                                                \
    }
#define CCVPE_STORE_TILE(stage)                                                                          \
    {                                                                                                    \
        unsigned char* sb = smem_b + (stage) * STAGE;                                                    \
        if constexpr (!SPLIT) {                                                                          \
            _Pragma("unroll") for (int j = 0; j < AR; ++j) {                                             \
                f32x4 v_ = ra[j];                                                                        \
                if (GATE) v_ *= rg[j];                                                                   \
                bf16x4 h_, l_;                                                                           \
                _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                          \
                    h_[e] = (__bf16)v_[e];                                                               \
                    l_[e] = (__bf16)(v_[e] - (float)h_[e]);                                              \
                }                                                                                        \
                const int r_ = r0 + 32 * j;                                                              \
                const int o_ = swz<MT>(r_, kq >> 1) + (kq & 1) * 8;                                          \
                *reinterpret_cast<bf16x4*>(sb + o_) = h_;                                                \
                *reinterpret_cast<bf16x4*>(sb + BM * 64 + o_) = l_;                                      \
            }                                                                                            \
        }                                                                                                \
        if constexpr (!SPLIT) {                                                                          \
            _Pragma("unroll") for (int j = 0; j < BRH; ++j) {                                            \
                const int r_ = wr0 + 64 * j;                                                             \
                if (BN % 64 == 0 || r_ < BN) {                                                           \
                    *reinterpret_cast<u32x4*>(sb + 2 * BM * 64 + swz<MT>(r_, wch)) = rbh[j];             \
                    *reinterpret_cast<u32x4*>(sb + 2 * BM * 64 + BN * 64 + swz<MT>(r_, wch)) = rbl[j];   \
                }                                                                                        \
            }                                                                                            \
        }                                                                                                \
    }

    acc_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < M::NACC; ++r) acc[i][j][r] = 0.f;

    const int nkt_all = p.Kpad / BK;
    int kt0 = 0, kt1 = nkt_all;
    if (p.splitk > 1) {
        const int per = (nkt_all + p.splitk - 1) / p.splitk;
        kt0 = min((int)blockIdx.z * per, nkt_all);
        kt1 = min(kt0 + per, nkt_all);
    }
    {
        const int kfirst = min(kt0, nkt_all - 1);
        if constexpr (SPLIT) {
            CCVPE_DMA_TILE(kfirst, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            CCVPE_LOAD_TILE(kfirst);
            CCVPE_STORE_TILE(0);
        }
    }
    __syncthreads();

    const int a_row = wm * WM + (lane % MT);
    const int b_row = wn * WN + (lane % MT);

    for (int kt = kt0; kt < kt1; ++kt) {
        const int stage = (kt - kt0) & 1;
        const int ktn = min(kt + 1, kt1 - 1);
        if constexpr (SPLIT) {
            // stage^1 was last read one barrier ago: the copies of the next K tile land while this tile's MFMAs run
            if (kt + 1 < kt1) CCVPE_DMA_TILE(ktn, stage ^ 1);
        } else {
            CCVPE_LOAD_TILE(ktn);
        }
        __builtin_amdgcn_sched_barrier(0);
        const unsigned char* sb = smem_b + stage * STAGE;
#pragma unroll
        for (int ks = 0; ks < M::KSTEPS; ++ks) {
            const int ch = M::chunk(ks, lane);
            bf16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int o = swz<MT>(a_row + i * MT, ch);
                ah[i] = *reinterpret_cast<const bf16x8*>(sb + o);
                al[i] = *reinterpret_cast<const bf16x8*>(sb + BM * 64 + o);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int o = swz<MT>(b_row + j * MT, ch);
                bh[j] = *reinterpret_cast<const bf16x8*>(sb + 2 * BM * 64 + o);
                bl[j] = *reinterpret_cast<const bf16x8*>(sb + 2 * BM * 64 + BN * 64 + o);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = M::run(al[i], bh[j], acc[i][j]);
                    acc[i][j] = M::run(ah[i], bl[j], acc[i][j]);
                    acc[i][j] = M::run(ah[i], bh[j], acc[i][j]);
                }
        }
        // ... and the LDS stores (which wait for those loads) BELOW it
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (SPLIT) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's LDS-DMA copies have landed
        } else {
            CCVPE_STORE_TILE(stage ^ 1);
        }
        __syncthreads();
    }
#undef CCVPE_LOAD_TILE
#undef CCVPE_STORE_TILE
#undef CCVPE_DMA_TILE

    // ---- epilogue (as kernels_igemm.hip): C tile through LDS, 16-byte row stores ----
    constexpr int LDC = BN + 4;
    float* Cs = reinterpret_cast<float*>(smem_b);
    const bool split = p.splitk > 1;
    const int col = lane % MT;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int nl = wn * WN + j * MT + col;
        const int n = n0 + nl;
        const float bias = (!split && n < p.N) ? p.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < M::NACC; ++r) {
                const int ml = wm * WM + i * MT + M::row(r, lane);
                const float v = acc[i][j][r] + bias;
                Cs[ml * LDC + nl] = split ? v : apply_act(v, p.act);
            }
        }
    }
    __syncthreads();
    constexpr int C4 = BN / 4;
    for (int it = tid; it < BM * C4; it += 256) {
        const int ml = it / C4, c4 = it - ml * C4;
        const int m = m0 + ml, n = n0 + c4 * 4;
        if (m >= p.M || n >= p.N) continue;
        const f32x4 v = *reinterpret_cast<const f32x4*>(Cs + ml * LDC + c4 * 4);
        if (split) {
            float* dst = p.partial + ((size_t)blockIdx.z * p.M + m) * p.N + n;
            if ((p.N & 3) == 0) *reinterpret_cast<f32x4*>(dst) = v;
            else
                for (int e = 0; e < 4 && n + e < p.N; ++e) dst[e] = v[e];
        } else {
            emit_out4(p, m, n, v);
        }
    }
}

template <int BM, int BN, int WGM, int WGN, int MT, bool GATE, bool SPLIT>
static void launch_b2(const ConvParams& p, hipStream_t s) {
    constexpr size_t stage_bytes = 2 * (size_t)(2 * BM + 2 * BN) * 64;
    constexpr size_t c_bytes = (size_t)BM * (BN + 4) * sizeof(float);
    constexpr size_t lds = stage_bytes > c_bytes ? stage_bytes : c_bytes;
    static LdsAttr attr;
    auto kern = conv_igemm_bf16x3_kernel<BM, BN, WGM, WGN, MT, GATE, SPLIT>;
    ensure_dynamic_lds(attr, reinterpret_cast<const void*>(kern), lds);
    dim3 grid((p.M + BM - 1) / BM, (p.N + BN - 1) / BN, p.splitk > 1 ? p.splitk : 1);
    CCVPE_LAUNCH(kern, grid, dim3(256), lds, s, p);
    if (p.splitk > 1) launch_splitk_reduce(p, s);
}

template <int BM, int BN, int WGM, int WGN, int MT>
static void launch_b(const ConvParams& p, hipStream_t s) {
    if (p.in_split) launch_b2<BM, BN, WGM, WGN, MT, false, true>(p, s);
    else if (p.gate) launch_b2<BM, BN, WGM, WGN, MT, true, false>(p, s);
    else launch_b2<BM, BN, WGM, WGN, MT, false, false>(p, s);
}

static const Bf16x3Tile BF16X3_TILES_[] = {
    {128, 128, "conv_bf16x3_128x128_m32", launch_b<128, 128, 2, 2, 32>},
    {128, 64, "conv_bf16x3_128x64_m32", launch_b<128, 64, 2, 2, 32>},
    {64, 64, "conv_bf16x3_64x64_m32", launch_b<64, 64, 2, 2, 32>},
    {128, 128, "conv_bf16x3_128x128_m16", launch_b<128, 128, 2, 2, 16>},
    {128, 64, "conv_bf16x3_128x64_m16", launch_b<128, 64, 2, 2, 16>},
    {128, 80, "conv_bf16x3_128x80_m16", launch_b<128, 80, 4, 1, 16>},
    {128, 48, "conv_bf16x3_128x48_m16", launch_b<128, 48, 4, 1, 16>},
    {128, 32, "conv_bf16x3_128x32_m16", launch_b<128, 32, 4, 1, 16>},
    {128, 16, "conv_bf16x3_128x16_m16", launch_b<128, 16, 4, 1, 16>},
    {64, 64, "conv_bf16x3_64x64_m16", launch_b<64, 64, 2, 2, 16>},
    {64, 32, "conv_bf16x3_64x32_m16", launch_b<64, 32, 2, 2, 16>},
    {256, 64, "conv_bf16x3_256x64_m32", launch_b<256, 64, 4, 1, 32>},
};
int bf16x3_num_tiles() { return (int)(sizeof(BF16X3_TILES_) / sizeof(BF16X3_TILES_[0])); }
const Bf16x3Tile* bf16x3_tile(int i) { return &BF16X3_TILES_[i]; }

}  // namespace ccvpe
