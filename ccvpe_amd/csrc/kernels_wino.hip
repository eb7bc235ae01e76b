// Winograd F(2x2, 3x3) convolution on the fp32 matrix cores of gfx950, fully fused (no transformed tensors
// in HBM).
//
// Serves the decoder double_conv 3x3 / stride 1 / pad 1 call sites of the reference (models.py:42-47 used by
// conv6..conv2 and conv6_ori..conv2_ori, models.py:407-446): 72 % of the forward pass FLOPs.  The minimal
// filtering form  Y = A^T [ sum_c (G g G^T) (.) (B^T d B) ] A  needs 16 multiplies per 2x2 output tile and
// channel pair instead of 36, i.e. 2.25x fewer matrix-core cycles than the implicit GEMM for the same
// result (fp32 throughout; the transforms only add/subtract activations and the weights are transformed
// once on the host in double precision).
//
// Work decomposition: a workgroup of NW wave64 owns 32 output tiles (an 8 x 4 patch of 2x2 tiles = 16 x 8
// output pixels of one image) x 16*NW output channels, for all 16 Winograd positions xi.  K is walked in
// chunks of 8 input channels:
//   * each thread gathers the 4x4 input patch of one (tile, channel) with bounds-checked buffer loads (the
//     pad-1 halo returns zeros), applies B^T d B in registers (32 adds) and scatters the 16 values to LDS;
//   * the transformed weights of the chunk (host layout == LDS layout, 512 B per (xi, 16 channels)) are copied
//     with 16-byte loads;
//   * wave w then runs, for each xi, four v_mfma_f32_16x16x4_f32: 2 tile halves x 2 k-steps, from three
//     conflict-free ds_read_b64 (a lane's two k-steps are adjacent in LDS).  The wave keeps all
//     16 xi x 32 tiles x 16 channels accumulators (128 registers), so the inverse transform A^T M A is a
//     per-lane affair in the epilogue.
// Global loads of chunk i+1 are in flight under the MFMAs of chunk i (register staging, single LDS image,
// two barriers per chunk; two workgroups per CU cover each other's barriers).
#include "igemm_common.h"

#include <algorithm>

namespace ccvpe {

typedef float f32x2 __attribute__((ext_vector_type(2)));

#ifndef CCVPE_WINO_SCALAR_COLUMNS
#define CCVPE_WINO_SCALAR_COLUMNS 0   // 1: column pass of B^T d B as eight scalar adds instead of two packed ones (measured: DESIGN.md 4.0)
#endif

// NW channel slices (16 output channels each) x NM tile sets (8 x 4 tiles each, stacked vertically) per workgroup,
// one wave per (slice, set); GC = 8-channel chunks per raw-patch refresh.
template <int NW, int NM, int GC, bool FUSED = false>
__global__ __launch_bounds__(NW * NM * 64) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_wino_kernel(const ConvParams p) {   // FUSED: see conv_wino4_kernel
    constexpr int NT = NW * NM * 64;
    constexpr int NITEM = 256 * NM;                        // (tile, channel) transform items per chunk
    constexpr int ITEMS = (NITEM + NT - 1) / NT;
    constexpr int PXS = GC * 8 + 4;                        // floats per raw pixel in LDS (16 B pad: conflict-free reads)
    constexpr int PROWS = 8 * NM + 2;                      // raw patch: PROWS x 18 pixels
    constexpr int RAW_F4 = PROWS * 18 * GC * 2;            // float4s of one raw group (GC*8 channels per pixel)
    constexpr int RAW_ITEMS = (RAW_F4 + NT - 1) / NT;
    constexpr unsigned OOB = 0x80000000u;
    // V image double-buffered when it fits twice beside a second workgroup (single tile set): the transform of chunk
    // i+1 is then written while slower waves still read chunk i - one barrier per chunk instead of two
    constexpr bool VDB = NM == 1;
    constexpr int VSZ = 4096 * NM;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Vs = smem;                          // [VDB ? 2 : 1][NM sets][16 xi][2 halves][4 kq][16 tiles (swizzled)][2 kh]
    float* Rs = smem + VSZ * (VDB ? 2 : 1);    // [PROWS][18 cols][PXS]  raw input patch of the current channel group

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = (tid >> 6) % NW;      // channel slice
    const int wset = (tid >> 6) / NW;      // tile set

    const int mbx = p.W >> 4, mby = p.H / (8 * NM);
    const int mblocks = p.B * mbx * mby;
    const int total = mblocks * ((p.wino_n16 + NW - 1) / NW);

    // ---- persistent work loop: XCD x (blockIdx.x % 8) owns a contiguous run of tiles, its workgroups stride
    // through it.  Tile index = channel block * mblocks + m block (channel block slowest: workgroups that run
    // together share the weight panel in their L2).  While a tile's last chunks are on the matrix pipe the next
    // tile's first patch and weights are already in flight, and its stores drain under the next tile's MFMAs.
    const int xcd = blockIdx.x & 7;
    const int stride = ((int)gridDim.x >> 3) + (xcd < ((int)gridDim.x & 7) ? 1 : 0);
    const int item_begin = xcd * (total >> 3) + min(xcd, total & 7);
    const int item_end = item_begin + (total >> 3) + (xcd < (total & 7) ? 1 : 0);
    int item = item_begin + ((int)blockIdx.x >> 3);
    if (item >= item_end) return;
    const int item_first = item;
    int ordinal = 0;                     // items this workgroup has finished
    unsigned long long fin_mask = 0;     // self-reducing split-K: ordinals of the regions it is the last K slice of

    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, p.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wino_w), 0, p.wino_bytes, 0x00020000);

    int nb, b, by, bx;        // current tile (uniform)
#define CCVPE_WINO_DECODE(it_, nb_, b_, by_, bx_)                                                        \
    {                                                                                                    \
        nb_ = (it_) / mblocks;                                                                           \
        const int mb_ = (it_) - nb_ * mblocks;                                                           \
        b_ = mb_ / (mbx * mby);                                                                          \
        const int rem_ = mb_ - b_ * (mbx * mby);                                                         \
        by_ = rem_ / mbx;                                                                                \
        bx_ = rem_ - by_ * mbx;                                                                          \
    }
    CCVPE_WINO_DECODE(item, nb, b, by, bx);

    // ---- raw patch: float4 j = tid + i*NT  ->  pixel j / (2*GC) of the PROWS x 18 region, channels 4*(j % (2*GC)).
    // r_off is the per-thread part of the address (buffer voffset); the channel-group offset is uniform and travels
    // in the scalar soffset operand, so the loads cost no vector ALU work inside the K loop (non-MFMA VALU
    // instructions take matrix-pipe issue slots on gfx950)
    unsigned r_off[RAW_ITEMS];
#define CCVPE_WINO_ROFF(b_, by_, bx_, live_)                                                             \
    _Pragma("unroll") for (int i = 0; i < RAW_ITEMS; ++i) {                                              \
        int j = tid + i * NT;                                                                            \
        asm volatile("" : "+v"(j));   /* recomputed per item, not hoisted into scratch (kernels_wino4.hip) */ \
        const int px = j / (2 * GC), q = j - px * (2 * GC);                                              \
        const int py = px / 18, pxx = px - py * 18;                                                      \
        const int y = (by_) * 8 * NM - 1 + py, x = (bx_) * 16 - 1 + pxx;                                 \
        const bool ok = (live_) && j < RAW_F4 && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W; \
        r_off[i] = ok ? (unsigned)(((((b_) * p.H + y) * p.W + x) * p.in_ld + q * 4) * 4) : OOB;          \
    }
    CCVPE_WINO_ROFF(b, by, bx, true);
    // ---- transform items: (tile t, channel k of the chunk) ----
    int g_raw[ITEMS], g_lds[ITEMS];
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const int it = tid + i * NT;
        const int k = it & 7, tg = (it >> 3) & (32 * NM - 1), t = tg & 31;   // tg: tile inside the block, t: inside its set
        g_raw[i] = ((tg >> 3) * 2 * 18 + (tg & 7) * 2) * PXS + k;
        // tile slot XOR 8 for kq >= 2: the 8 tiles x 8 channels a wave scatters then cover all 64 banks once
        g_lds[i] = (tg >> 5) * 4096 + (t >> 4) * 128 + ((k & 3) * 16 + ((t & 15) ^ ((k & 2) << 2))) * 2 + (k >> 2);
    }
    // weights never touch LDS: the host layout is the B-fragment layout of two consecutive xi, so a wave reads the
    // 1024 bytes of its (xi pair, 16-channel slice) with one 16-byte load per lane, straight into the MFMA operand
    // registers
    const unsigned w_pair_b = (unsigned)p.wino_n16 * 1024u;          // bytes between consecutive xi pairs
    const unsigned w_chunk_b = w_pair_b * 8u;                         // bytes per chunk
#define CCVPE_WINO_WBASE(nb_) ((nb_) * NW + wave < p.wino_n16 ? (unsigned)((nb_) * NW + wave) * 1024u + (unsigned)lane * 16u : OOB)
    unsigned w_base = CCVPE_WINO_WBASE(nb);

    // split-K over chunks (blockIdx.z); channel groups of near-equal length, at most GC chunks each
    const int nch = p.Cin >> 3;
    int c_begin = 0, c_end = nch;
    if (p.splitk > 1) {
        const int per = (nch + p.splitk - 1) / p.splitk;
        c_begin = min((int)blockIdx.z * per, nch);
        c_end = min(c_begin + per, nch);
    }
    const int ngroups = (c_end - c_begin + GC - 1) / GC;
    const int glen_lo = ngroups > 0 ? (c_end - c_begin) / ngroups : 0;       // groups [n_hi, ngroups) have this length
    const int n_hi = ngroups > 0 ? (c_end - c_begin) - glen_lo * ngroups : 0;   // the first n_hi groups are one longer

    f32x4 raw[RAW_ITEMS];
    f32x2 bq[16];             // B fragments of the current chunk, refilled in place for the next one
    // channels past the group / past Cin land in Rs but are never read: the chunk loop stops at Cin / 8
#define CCVPE_WINO_LOAD_RAW(c0)   /* channel group starting at channel c0 */                             \
    _Pragma("unroll") for (int i = 0; i < RAW_ITEMS; ++i)                                                \
        raw[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, r_off[i], (c0) * 4, 0));
#define CCVPE_WINO_STORE_RAW()                                                                           \
    _Pragma("unroll") for (int i = 0; i < RAW_ITEMS; ++i) {                                              \
        const int j = tid + i * NT;                                                                      \
        if (RAW_ITEMS * NT == RAW_F4 || j < RAW_F4)                                                      \
            *reinterpret_cast<f32x4*>(Rs + (j / (2 * GC)) * PXS + (j % (2 * GC)) * 4) = raw[i];          \
    }
#define CCVPE_WINO_LOAD_B(wb, ch, g)   /* xi pair g = (2g, 2g+1) */                                      \
    {                                                                                                    \
        const f32x4 t4_ = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, wb, (ch) * w_chunk_b + (g) * w_pair_b, 0)); \
        bq[2 * (g)] = f32x2{t4_.x, t4_.y};                                                               \
        bq[2 * (g) + 1] = f32x2{t4_.z, t4_.w};                                                           \
    }

    f32x4 acc[16][2];
#pragma unroll
    for (int x = 0; x < 16; ++x) {
        acc[x][0] = f32x4{0.f, 0.f, 0.f, 0.f};
        acc[x][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (c_begin >= c_end) return;   // empty split-K slice (the host never launches one)

    CCVPE_WINO_LOAD_RAW(c_begin * 8);
#pragma unroll
    for (int g = 0; g < 8; ++g) { CCVPE_WINO_LOAD_B(w_base, c_begin, g); }
    CCVPE_WINO_STORE_RAW();
    __syncthreads();
    const float* va0 = Vs + wset * 4096 + ((lane & 48) + ((lane & 15) ^ ((lane & 32) >> 2))) * 2;   // same slot swizzle as g_lds
    int vbuf = 0;
    const bool split = p.splitk > 1;
    const int ld = split ? p.N : p.dst[0].ld;
    const int act = split ? ACT_NONE : p.act;

    while (true) {
        const int item_n = item + stride;
        const bool have_n = item_n < item_end;
        int nb_n, b_n, by_n, bx_n;
        CCVPE_WINO_DECODE(have_n ? item_n : item, nb_n, b_n, by_n, bx_n);
        const unsigned w_base_n = have_n ? CCVPE_WINO_WBASE(nb_n) : OOB;

        int gidx = 0, gstart = c_begin, glen = glen_lo + (n_hi > 0 ? 1 : 0);   // current channel group
        for (int ch = c_begin; ch < c_end; ++ch) {
            const int sub = ch - gstart;               // chunk inside the channel group held in Rs
            const bool last_chunk = ch == c_end - 1;
            if (sub == 0) {
                // a group has just begun: fetch the one after it - of this tile, or the first one of the next tile
                if (gidx == ngroups - 1) {
                    CCVPE_WINO_ROFF(b_n, by_n, bx_n, have_n);
                    CCVPE_WINO_LOAD_RAW(c_begin * 8);
                } else {
                    CCVPE_WINO_LOAD_RAW((gstart + glen) * 8);
                }
            }
            // gather the 4x4 patch of (tile, channel) from the raw image and apply B^T d B in registers
            float* Vw = Vs + (VDB ? vbuf * VSZ : 0);
            const float* va = va0 + (VDB ? vbuf * VSZ : 0);
            float v[VDB ? 1 : ITEMS][16];
#pragma unroll
            for (int i = 0; i < ITEMS; ++i) {
                const float* rp_ = Rs + g_raw[i] + sub * 8;
                // B^T d B on register pairs: D[r][cp] = patch pixels (r, 2cp) and (r, 2cp+1) (one ds_read2_b32 each),
                // row pass = 8 packed adds, column pass = 8 packed adds whose operand halves are picked with
                // op_sel / neg modifiers (hipcc would insert v_mov shuffles; every VALU slot here is a lost
                // matrix-pipe slot)
                f32x2 D[4][2], X[4][2];
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int cp = 0; cp < 2; ++cp) D[r][cp] = f32x2{rp_[(r * 18 + 2 * cp) * PXS], rp_[(r * 18 + 2 * cp + 1) * PXS]};
#pragma unroll
                for (int cp = 0; cp < 2; ++cp) {
                    X[0][cp] = D[0][cp] - D[2][cp];
                    X[1][cp] = D[1][cp] + D[2][cp];
                    X[2][cp] = D[2][cp] - D[1][cp];
                    X[3][cp] = D[1][cp] - D[3][cp];
                }
                float* vi = v[VDB ? 0 : i];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    // lo = (x0 - x2, x1 + x2), hi = (x2 - x1, x1 - x3) with (x0,x1) = X[r][0], (x2,x3) = X[r][1].
                    // gfx950: a packed fp32 instruction whose LOW result takes the HIGH half of src1 (op_sel[1] = 1) returns wrong
                    // values in lanes 48-63 while another wave of the SIMD runs a 16- / 8-bit MFMA (tools/repro_pk_mfma.hip), and
                    // any co-tenant of the chip may run those - so the crossed operand always sits in src0 (op_sel[0] = 1, the form
                    // the reproducer shows clean); tests/test_isa_hazard.py holds every kernel of the library to that.
#if CCVPE_WINO_SCALAR_COLUMNS
                    vi[r * 4 + 0] = X[r][0].x - X[r][1].x; vi[r * 4 + 1] = X[r][0].y + X[r][1].x;
                    vi[r * 4 + 2] = X[r][1].x - X[r][0].y; vi[r * 4 + 3] = X[r][0].y - X[r][1].y;
#else
                    f32x2 lo, hi;
                    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,0]" : "=v"(lo) : "v"(X[r][0]), "v"(X[r][1]));
                    asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1] neg_lo:[1,0] neg_hi:[0,1]" : "=v"(hi) : "v"(X[r][0]), "v"(X[r][1]));
                    vi[r * 4 + 0] = lo.x; vi[r * 4 + 1] = lo.y; vi[r * 4 + 2] = hi.x; vi[r * 4 + 3] = hi.y;
#endif
                }
                if (VDB && (NT * ITEMS == NITEM || tid + i * NT < NITEM)) {
#pragma unroll
                    for (int x = 0; x < 16; ++x) Vw[x * 256 + g_lds[i]] = vi[x];
                }
            }
            __syncthreads();   // VDB: this chunk's V image is complete; else: the previous chunk's image is free
            if (!VDB) {
#pragma unroll
                for (int i = 0; i < ITEMS; ++i)
                    if (NT * ITEMS == NITEM || tid + i * NT < NITEM) {
#pragma unroll
                        for (int x = 0; x < 16; ++x) Vw[x * 256 + g_lds[i]] = v[VDB ? 0 : i][x];
                    }
            }
            // every wave has also finished this chunk's Rs reads by now
            const bool group_end = sub == glen - 1;
            if (group_end) {   // bring in the next group's patch (loaded glen chunks ago)
                CCVPE_WINO_STORE_RAW();
                gstart += glen;
                ++gidx;
                glen = glen_lo + (gidx < n_hi ? 1 : 0);
            }
            if (!VDB || group_end) __syncthreads();
            vbuf ^= 1;
            // the next chunk's weights: of this tile, or chunk c_begin of the next tile's channel block
            const unsigned wb = last_chunk ? w_base_n : w_base;
            const int chn = last_chunk ? c_begin : ch + 1;
            __builtin_amdgcn_sched_barrier(0);
            // A fragments double-buffered in registers, two xi per step: the ds_reads of pair g+1 are in flight
            // under the eight MFMAs of pair g (hipcc otherwise waits out the LDS latency before every group);
            // a pair's B registers are refilled for the next chunk as soon as its MFMAs have issued, so the
            // weight loads have a whole chunk period to land
            f32x2 fa[2][2][2];
#define CCVPE_WINO_FRAGS(buf, g)                                                                         \
    _Pragma("unroll") for (int e = 0; e < 2; ++e) {                                                      \
        fa[buf][e][0] = *reinterpret_cast<const f32x2*>(va + ((g) * 2 + e) * 256);                       \
        fa[buf][e][1] = *reinterpret_cast<const f32x2*>(va + ((g) * 2 + e) * 256 + 128);                 \
    }
            CCVPE_WINO_FRAGS(0, 0);
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const int cur = g & 1;
                if (g + 1 < 8) { CCVPE_WINO_FRAGS(cur ^ 1, g + 1); }
                __builtin_amdgcn_sched_barrier(0);   // pin the reads above this pair's MFMAs
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int x = g * 2 + e;
                    acc[x][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(bq[x].x, fa[cur][e][0].x, acc[x][0], 0, 0, 0);
                    acc[x][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(bq[x].x, fa[cur][e][1].x, acc[x][1], 0, 0, 0);
                    acc[x][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(bq[x].y, fa[cur][e][0].y, acc[x][0], 0, 0, 0);
                    acc[x][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(bq[x].y, fa[cur][e][1].y, acc[x][1], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                CCVPE_WINO_LOAD_B(wb, chn, g);
            }
#undef CCVPE_WINO_FRAGS
            __builtin_amdgcn_sched_barrier(0);
        }

        // ---- inverse transform A^T M A, bias, activation, store ----
        // The weights are the A operand of the MFMAs, so the accumulators are channel-major: lane = (channel quad lane >> 4, tile
        // lane & 15 of the half: row (lane & 15) >> 3, column (lane & 15) & 7) and element i of an accumulator is channel
        // 4 (lane >> 4) + i.  The transform runs on the channel quads (packed fp32 adds) and every output pixel of the tile
        // leaves as ONE 16-byte buffer store (8 per lane instead of 32 dword stores with a branch on the activation each); the
        // (half, pixel) part of every address is uniform and rides in soffset.
        {
            const int n = (nb * NW + wave) * 16 + 4 * (lane >> 4);
            f32x4 bias = {0.f, 0.f, 0.f, 0.f};
            if (!split) {
#pragma unroll
                for (int i = 0; i < 4; ++i) bias[i] = n + i < p.N ? p.bias[n + i] : 0.f;
            }
            const float lo = act == ACT_RELU ? 0.f : -__builtin_inff();   // ReLU as a select: no branch per store
            // first output pixel of this wave's tile set
            const size_t pix0 = ((size_t)b * p.H + (size_t)(by * NM + wset) * 8) * p.W + (size_t)bx * 16;
            float* obase = split ? p.partial + ((size_t)blockIdx.z * p.M + pix0) * p.N : p.dst[0].ptr + pix0 * ld + p.dst[0].coff;
            const __amdgpu_buffer_rsrc_t o_rsrc = __builtin_amdgcn_make_buffer_rsrc(obase, 0, 0x7fffffff, 0x00020000);
            const int tl = lane & 15;
            const unsigned o_lane = n < p.N ? (unsigned)((((tl >> 3) * 2 * p.W + (tl & 7) * 2) * ld + n) * 4) : OOB;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                f32x4 tt[2][4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    tt[0][c] = acc[0 * 4 + c][h] + acc[1 * 4 + c][h] + acc[2 * 4 + c][h];
                    tt[1][c] = acc[1 * 4 + c][h] - acc[2 * 4 + c][h] - acc[3 * 4 + c][h];
                }
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    f32x4 y[2];
                    y[0] = tt[a][0] + tt[a][1] + tt[a][2] + bias;
                    y[1] = tt[a][1] - tt[a][2] - tt[a][3] + bias;
#pragma unroll
                    for (int dx = 0; dx < 2; ++dx) {
                        f32x4 v;
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[i] = y[dx][i] < lo ? lo : y[dx][i];
                        const int soff = (((h * 4 + a) * p.W + dx) * ld) * 4;   // uniform
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, v), o_rsrc, o_lane, soff, FUSED ? 16 : 0);   // (FUSED: slab, write-through - ticket.h)
                        // two wait states before anything may overwrite the store's data registers (gfx950: tests/test_isa_hazard.py)
                        __builtin_amdgcn_sched_barrier(0);
                        asm volatile("s_nop 1");
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        }
        if (FUSED) {
            // self-reducing split-K (igemm_common.h): the last K slice of this workgroup's region (NM tile sets of 16 x 8 pixels, NW x 16
            // channels) sums the slabs and stores it.  The ticket's first barrier ends every wave's reads of the V image, whose first
            // word then serves as the flag; the next item's raw patch (Rs) is not touched.
            if (splitk_ticket(p, item, reinterpret_cast<unsigned*>(Vs))) fin_mask |= 1ull << ordinal;   // (summed behind the loop: registers)
        }
        if (FUSED) ++ordinal;
        if (!have_n) break;
#pragma unroll
        for (int x = 0; x < 16; ++x) {
            acc[x][0] = f32x4{0.f, 0.f, 0.f, 0.f};
            acc[x][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        item = item_n; nb = nb_n; b = b_n; by = by_n; bx = bx_n; w_base = w_base_n;
    }
    while (FUSED && fin_mask) {   // the regions whose last ticket this workgroup drew (<= 64 items per workgroup: the launcher)
        const int k = __builtin_ctzll(fin_mask);
        fin_mask &= fin_mask - 1;
        int nb_f, b_f, by_f, bx_f;
        CCVPE_WINO_DECODE(item_first + k * stride, nb_f, b_f, by_f, bx_f);
        splitk_finish<NT>(p, (b_f * p.H + by_f * NM * 8) * p.W + bx_f * 16, 8 * NM, 16, p.W, nb_f * NW * 16, NW * 16);
    }
#undef CCVPE_WINO_LOAD_RAW
#undef CCVPE_WINO_STORE_RAW
#undef CCVPE_WINO_LOAD_B
#undef CCVPE_WINO_ROFF
#undef CCVPE_WINO_DECODE
#undef CCVPE_WINO_WBASE
}

template <int NW, int NM, int GC>
static void launch_wino(const ConvParams& p_in, hipStream_t s) {
    ConvParams p = p_in;
    if (p.splitk > 1) {   // never launch an empty K slice (it would leave its slab unwritten)
        const int nch = p.Cin >> 3;
        const int per = (nch + p.splitk - 1) / p.splitk;
        p.splitk = (nch + per - 1) / per;
    }
    constexpr size_t lds = (4096 * NM * (NM == 1 ? 2 : 1) + (8 * NM + 2) * 18 * (GC * 8 + 4)) * sizeof(float);
    static_assert(2 * lds <= 160 * 1024, "two workgroups per CU");
    static LdsAttr attr, attr_f;
    const int mblocks = p.B * (p.W >> 4) * (p.H / (8 * NM));
    const int nblocks = (p.wino_n16 + NW - 1) / NW;
    // persistent grid: two workgroups per CU (the register budget allows no more) loop over the tiles
    const int resident = 2 * 256 / (p.splitk > 1 ? p.splitk : 1);
    dim3 grid(std::min(mblocks * nblocks, std::max(resident, 8)), 1, p.splitk > 1 ? p.splitk : 1);
    if (p.splitk <= 1 || p.tickets == nullptr || mblocks * nblocks > CONV_TICKETS || (mblocks * nblocks + (int)grid.x - 1) / (int)grid.x + 1 > 64) p.split_fused = 0;
    if (p.split_fused) {
        auto kern = conv_wino_kernel<NW, NM, GC, true>;
        ensure_dynamic_lds(attr_f, reinterpret_cast<const void*>(kern), lds);
        CCVPE_LAUNCH(kern, grid, dim3(NW * NM * 64), lds, s, p);
        return;
    }
    auto kern = conv_wino_kernel<NW, NM, GC, false>;
    ensure_dynamic_lds(attr, reinterpret_cast<const void*>(kern), lds);
    CCVPE_LAUNCH(kern, grid, dim3(NW * NM * 64), lds, s, p);
    if (p.splitk > 1) launch_splitk_reduce(p, s);
}

// bm = output pixels per workgroup (32 tiles x 4 pixels per set), bn = output channels per workgroup
static const WinoTile WINO_TILES[] = {
    {256, 32, "conv_wino_64x32", launch_wino<2, 2, 2>, 2, false, -1},
    {128, 48, "conv_wino_32x48", launch_wino<3, 1, 2>, 2, false, -1},
    {128, 64, "conv_wino_32x64", launch_wino<4, 1, 4>, 2, false, -1},
    {128, 80, "conv_wino_32x80", launch_wino<5, 1, 4>, 2, false, -1},
    {256, 48, "conv_wino_64x48", launch_wino<3, 2, 2>, 2, false, -1},
    // F(4x4,3x3), kernels_wino4.hip: 16 tiles of 4x4 pixels x 64 channels per workgroup
    {256, 64, "conv_wino4_16x64", launch_wino4_64, 4, false, -1},
    {256, 128, "conv_wino4_16x128", launch_wino4_128, 4, false, -1},
    // F(4x4,3x3) split form, kernels_wino4p.hip: V = B^T d B written once per layer, matrix kernel without a transform
    {256, 64, "conv_wino4p_16x64", launch_wino4p_64, 4, true, -1},
    {256, 128, "conv_wino4p_16x128", launch_wino4p_128, 4, true, -1},
    // F(4x4,3x3) xi-split form, kernels_wino4x.hip: one n-block = the whole (narrow) layer, nine xi per wave
    {256, 32, "conv_wino4x_32", launch_wino4x, 4, false, 0},
    {256, 48, "conv_wino4x_48", launch_wino4x, 4, false, 1},
    {256, 64, "conv_wino4x_64", launch_wino4x, 4, false, 2},
    {256, 80, "conv_wino4x_80", launch_wino4x, 4, false, 3},
    {256, 96, "conv_wino4x_96", launch_wino4x, 4, false, 4},
    {256, 128, "conv_wino4x_128", launch_wino4x, 4, false, 5},
};
int wino_num_tiles() { return (int)(sizeof(WINO_TILES) / sizeof(WINO_TILES[0])); }
const WinoTile* wino_tile(int i) { return &WINO_TILES[i]; }

bool conv_wino_supported(const ConvParams& p) {
    return p.wino_w != nullptr && p.KH == 3 && p.KW == 3 && p.stride == 1 && p.pad_t == 1 && p.pad_l == 1 && p.mode == MODE_CONV &&
           p.gate == nullptr && p.resid == nullptr && p.ndst == 1 && !p.dst[0].split && !p.in_split && p.OH == p.H && p.OW == p.W &&
           p.W % 16 == 0 && p.H % 16 == 0 && p.Cin % 8 == 0 && p.N % 4 == 0 && p.dst[0].ld % 4 == 0 && p.dst[0].coff % 4 == 0;
}

// Host-side weight transform: U = G g G^T per (cout, cin) in double precision, stored in the LDS image order
//   [chunk = cin/8][xi/2][n16][lane = (cin%4)*16 + cout%16][xi%2][kh = (cin%8)/4]      (1024 B per (xi pair, n16) = one 16-byte
//   B-fragment load per lane)
// `get(n, tap, c)` returns the 3x3 weight (tap = ky*3 + kx).
size_t conv_wino_pack(int N, int cin, const std::function<float(int, int, int)>& get, std::vector<float>& out, int* n16_out) {
    static const double G[4][3] = {{1.0, 0.0, 0.0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0.0, 0.0, 1.0}};
    const int n16 = (N + 15) / 16;
    const int nch = cin / 8;
    out.assign((size_t)nch * 16 * n16 * 128, 0.f);
    for (int n = 0; n < N; ++n)
        for (int c = 0; c < cin; ++c) {
            double g[3][3], tmp[4][3];
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) g[i][j] = (double)get(n, i * 3 + j, c);
            for (int r = 0; r < 4; ++r)
                for (int j = 0; j < 3; ++j) tmp[r][j] = G[r][0] * g[0][j] + G[r][1] * g[1][j] + G[r][2] * g[2][j];
            const int chunk = c / 8, k = c % 8;
            for (int r = 0; r < 4; ++r)
                for (int q = 0; q < 4; ++q) {
                    const double uv = tmp[r][0] * G[q][0] + tmp[r][1] * G[q][1] + tmp[r][2] * G[q][2];
                    const int xi = r * 4 + q;
                    const size_t idx = ((((size_t)chunk * 8 + xi / 2) * n16 + n / 16) * 64 + (k & 3) * 16 + (n % 16)) * 4 + (xi & 1) * 2 + (k >> 2);
                    out[idx] = (float)uv;
                }
        }
    *n16_out = n16;
    return out.size();
}

}  // namespace ccvpe
