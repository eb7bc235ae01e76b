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
// (no im2col buffer) with zero fill for the halo.  Tiles are register-staged (global_load_dwordx4 ->
// ds_write_b128) into a double-buffered LDS image with 144-byte rows (conflict-free ds_read_b128), one
// barrier per K tile, next tile's loads in flight under the current tile's MFMAs.
// v_mfma_f32_32x32x2_f32 is exact fp32 (bitwise an fmaf chain) at 64 FLOP/clk/SIMD.  One ds_read_b128
// per operand feeds four MFMAs: lane l holds k = 4*(l>>5)+j of its row for j = 0..3, the same k
// permutation on A and B, so the contraction is complete after 4 issues.
#include "kernels.h"

namespace ccvpe {

typedef float f32x16 __attribute__((ext_vector_type(16)));

static constexpr int BK = 32;
static constexpr int LDK = 36;  // floats per LDS row: 32 + 4 pad -> 144 B, (144/16)=9 odd => b128 reads conflict-free

__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == ACT_RELU) return fmaxf(v, 0.f);
    if (act == ACT_SWISH) return v / (1.f + __expf(-v));
    return v;
}

template <int BM, int BN, int WGM, int WGN>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvParams p) {
    static_assert(WGM * WGN == 4, "4 waves per workgroup");
    constexpr int WM = BM / WGM, WN = BN / WGN;
    constexpr int TM = WM / 32, TN = WN / 32;
    static_assert(TM >= 1 && TN >= 1, "wave tile must hold a 32x32 MFMA tile");
    constexpr int AR = BM / 32;  // float4 rows each thread stages for A
    constexpr int BR = BN / 32;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                  // [2][BM][LDK]
    float* Bs = smem + 2 * BM * LDK;   // [2][BN][LDK]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int m0 = blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;
    const int kq = tid & 7;    // float4 slot inside the 32-deep K tile
    const int r0 = tid >> 3;   // 0..31

    // ---- per-row gather state (rows r0 + 32*j of the A tile) ----
    int a_base[AR], a_iy[AR], a_ix[AR], a_gb[AR];
    const int ohw = p.OH * p.OW;
#pragma unroll
    for (int j = 0; j < AR; ++j) {
        int m = m0 + r0 + 32 * j;
        bool ok = m < p.M;
        int mm = ok ? m : 0;
        int b = mm / ohw;
        int rem = mm - b * ohw;
        int oy = rem / p.OW;
        int ox = rem - oy * p.OW;
        int iy0 = oy * p.stride - p.pad_t;
        int ix0 = ox * p.stride - p.pad_l;
        a_base[j] = ((b * p.H + iy0) * p.W + ix0) * p.in_ld;
        a_iy[j] = ok ? iy0 : -(1 << 28);
        a_ix[j] = ix0;
        a_gb[j] = b * p.Cin;
    }
    // chunk walk state: chunk g = kt*4 + (kq>>1) -> (tap = g / cin8, cc = g % cin8)
    const int cin8 = p.Cin >> 3;
    const int c4 = (kq & 1) * 4;
    int g = kq >> 1;
    int tap = g / cin8;
    int cc = g - tap * cin8;
    int ky = tap / p.KW;
    int kx = tap - ky * p.KW;

    const float* wrow[BR];
#pragma unroll
    for (int j = 0; j < BR; ++j) wrow[j] = p.wpk + (size_t)(n0 + r0 + 32 * j) * p.Kpad + kq * 4;

    float4 ra[AR], rb[BR];

    auto load_tile = [&](int kt) {
        const bool gok = g < p.nchunks;
        const int koff = (ky * p.W + kx) * p.in_ld + cc * 8 + c4;
#pragma unroll
        for (int j = 0; j < AR; ++j) {
            int iy = a_iy[j] + ky, ix = a_ix[j] + kx;
            bool ok = gok && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ok) {
                v = *reinterpret_cast<const float4*>(p.in + (a_base[j] + koff));
                if (p.gate) {
                    float4 gv = *reinterpret_cast<const float4*>(p.gate + a_gb[j] + cc * 8 + c4);
                    v.x *= gv.x; v.y *= gv.y; v.z *= gv.z; v.w *= gv.w;
                }
            }
            ra[j] = v;
        }
#pragma unroll
        for (int j = 0; j < BR; ++j) rb[j] = *reinterpret_cast<const float4*>(wrow[j] + kt * BK);
        // advance the chunk walk by one K tile (4 chunks)
        g += 4;
        cc += 4;
        while (cc >= cin8) {
            cc -= cin8;
            if (++kx == p.KW) { kx = 0; ++ky; }
        }
    };
    auto store_tile = [&](int stage) {
        float* as = As + stage * BM * LDK;
        float* bs = Bs + stage * BN * LDK;
#pragma unroll
        for (int j = 0; j < AR; ++j) *reinterpret_cast<float4*>(as + (r0 + 32 * j) * LDK + kq * 4) = ra[j];
#pragma unroll
        for (int j = 0; j < BR; ++j) *reinterpret_cast<float4*>(bs + (r0 + 32 * j) * LDK + kq * 4) = rb[j];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nkt = p.Kpad / BK;
    load_tile(0);
    store_tile(0);
    __syncthreads();

    const int a_row = wm * WM + (lane & 31);
    const int b_row = wn * WN + (lane & 31);
    const int k_lane = (lane >> 5) * 4;

    for (int kt = 0; kt < nkt; ++kt) {
        const int stage = kt & 1;
        if (kt + 1 < nkt) load_tile(kt + 1);
        const float* as = As + stage * BM * LDK + a_row * LDK + k_lane;
        const float* bs = Bs + stage * BN * LDK + b_row * LDK + k_lane;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            float4 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const float4*>(as + i * 32 * LDK + kk * 8);
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const float4*>(bs + j * 32 * LDK + kk * 8);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, b[j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, b[j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, b[j].w, acc[i][j], 0, 0, 0);
                }
        }
        if (kt + 1 < nkt) store_tile(stage ^ 1);
        __syncthreads();
    }

    // ---- epilogue: C/D layout col = lane&31 (n), row = (r&3) + 8*(r>>2) + 4*(lane>>5) (m) ----
    const int col = lane & 31;
    const int rhalf = (lane >> 5) * 4;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * WN + j * 32 + col;
        const bool nok = n < p.N;
        const float bias = nok ? p.bias[n] : 0.f;
        int q = 0, o = n;
        if (p.mode == MODE_DECONV) {
            q = n / p.deconv_cout;
            o = n - q * p.deconv_cout;
        }
        const int dy = q >> 1, dx = q & 1;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + rhalf;
                if (!nok || m >= p.M) continue;
                float v = apply_act(acc[i][j][r] + bias, p.act);
                int opix = m;
                if (p.mode == MODE_DECONV) {
                    int x = m % p.W;
                    int t = m / p.W;
                    int y = t % p.H;
                    int b = t / p.H;
                    opix = (b * 2 * p.H + 2 * y + dy) * (2 * p.W) + 2 * x + dx;
                } else if (p.resid) {
                    v += p.resid[(size_t)m * p.resid_ld + n];
                }
#pragma unroll
                for (int d = 0; d < 3; ++d)
                    if (d < p.ndst) p.dst[d].ptr[(size_t)opix * p.dst[d].ld + p.dst[d].coff + o] = v;
            }
        }
    }
}

template <int BM, int BN, int WGM, int WGN>
static void launch_cfg(const ConvParams& p, hipStream_t s) {
    constexpr size_t lds = 2 * (BM + BN) * LDK * sizeof(float);
    static bool attr_done = false;
    auto kern = conv_igemm_kernel<BM, BN, WGM, WGN>;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    dim3 grid((p.M + BM - 1) / BM, (p.N + BN - 1) / BN);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, p);
}

int conv_igemm_npad() { return 128; }

static int pick_tile(const ConvParams& p) {
    struct Cand { int id, bm, bn; double intrinsic; };
    static const Cand cands[] = {
        {TILE_128x128, 128, 128, 1.00}, {TILE_128x64, 128, 64, 0.95}, {TILE_64x64, 64, 64, 0.85},
        {TILE_128x32, 128, 32, 0.85},
    };
    const double cus = 256.0;
    int best = TILE_64x64;
    double best_score = -1.0;
    for (const Cand& c : cands) {
        double gm = (p.M + c.bm - 1) / c.bm, gn = (p.N + c.bn - 1) / c.bn;
        double blocks = gm * gn;
        double util = ((double)p.M * p.N) / (gm * c.bm * gn * c.bn);
        double rounds = (double)(long long)((blocks + cus - 1) / cus);
        double quant = blocks / (rounds * cus);
        double score = util * quant * c.intrinsic;
        if (score > best_score) { best_score = score; best = c.id; }
    }
    return best;
}

static thread_local int g_last_tile = 0;
int conv_igemm_last_tile() { int t = g_last_tile; g_last_tile = 0; return t; }
const char* conv_igemm_tile_name(int tile) {
    switch (tile) {
        case TILE_128x128: return "conv_igemm_128x128";
        case TILE_128x64: return "conv_igemm_128x64";
        case TILE_64x64: return "conv_igemm_64x64";
        case TILE_128x32: return "conv_igemm_128x32";
        default: return "";
    }
}

void launch_conv_igemm(const ConvParams& p, int tile, hipStream_t s) {
    if (tile == TILE_AUTO) tile = pick_tile(p);
    g_last_tile = tile;
    switch (tile) {
        case TILE_128x128: launch_cfg<128, 128, 2, 2>(p, s); break;
        case TILE_128x64:  launch_cfg<128, 64, 2, 2>(p, s); break;
        case TILE_128x32:  launch_cfg<128, 32, 4, 1>(p, s); break;
        case TILE_64x64:
        default:           launch_cfg<64, 64, 2, 2>(p, s); break;
    }
}

}  // namespace ccvpe
