// Internal kernel interface of libccvpe_hip.so (gfx950 only).  All activations are fp32 NHWC.
#pragma once
#include <functional>
#include <vector>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "ticket.h"

namespace ccvpe {

// Every kernel launch of the library goes through this macro: ccvpe_launch_count() (include/ccvpe.h) lets a caller count the launches of
// a forward call - the batch-1 latency record of bench.py reports launches per frame.
extern thread_local unsigned long long g_launches;
#define CCVPE_LAUNCH(...) do { ++::ccvpe::g_launches; hipLaunchKernelGGL(__VA_ARGS__); } while (0)

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-device property of a kernel: the launchers raise it lazily, once per
// (kernel instantiation, device) - a process may hold handles on several devices.  `state` is a function-local static array
// (one slot per device ordinal); a failing hipFuncSetAttribute leaves the slot unset, the launch that follows then fails and
// the forward call reports it through hipGetLastError.
struct LdsAttr { size_t set[16] = {0}; };
inline void ensure_dynamic_lds(LdsAttr& state, const void* kernel, size_t bytes) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
    if (bytes > state.set[dev] && hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess) state.set[dev] = bytes;
}

enum Act { ACT_NONE = 0, ACT_RELU = 1, ACT_SWISH = 2 };

// A destination view of an NHWC tensor: element (pixel p, channel c) lives at ptr[p*ld + coff + c].
struct Dst {
    float* ptr;
    int ld;
    int coff;
    // split != 0: the tensor is stored as two bf16 planes (bf16x3 mode): hi at ptr, lo `plane` bf16 elements later
    int split;
    long long plane;
};

// ---------------------------------------------------------------------------------------------
// Implicit-GEMM convolution on fp32 MFMA (v_mfma_f32_32x32x2_f32).
//   out[m, n] = act( sum_k A[m, k] * Wp[n, k] + bias[n] ) (+ resid[m, n])
// m = output pixel (b, oy, ox) for MODE_CONV, input pixel for MODE_DECONV (k2 s2 transposed conv,
// n = (dy*2+dx)*cout + o, pixel-shuffled on store).  k = (ky*KW + kx)*Cin + c, Cin % 8 == 0.
// ---------------------------------------------------------------------------------------------
enum { MODE_CONV = 0, MODE_DECONV = 1 };

struct ConvParams {
    const float* in;
    int in_ld;                 // floats per input pixel (>= Cin)
    int B, H, W, Cin;          // input geometry
    int OH, OW;                // GEMM-row geometry (conv: output map; deconv: == H, W)
    int KH, KW, stride, pad_t, pad_l;
    const float* wpk;          // [Npad][Kpad], k contiguous
    const unsigned short* w_hi;   // bf16x3 mode: bf16(w) and bf16(w - hi), same [Npad][Kpad] layout (null = unavailable)
    const unsigned short* w_lo;
    int Kpad;                  // multiple of 32
    int Npad;                  // rows in wpk (multiple of conv_igemm_npad())
    int nchunks;               // KH*KW*Cin/8 valid 8-channel chunks
    const float* bias;         // [N]
    int N;
    int act;
    const float* gate;         // optional [B][Cin] multiplier on A (squeeze-excite), 1x1 only
    const float* resid;        // optional residual [M][resid_ld]
    int resid_ld;
    Dst dst[3];
    int ndst;
    int mode;
    int deconv_cout;
    int M;                     // B*OH*OW
    unsigned in_bytes;         // extent of `in` for the bounds-checked buffer loads (< 2 GiB)
    int in_split;              // input is a split bf16 tensor (hi plane | lo plane); bf16x3 tiles only
    unsigned in_plane_bytes;   // byte distance between the two planes
    unsigned w_plane_bytes;    // extent of one bf16 weight plane (Npad*Kpad*2)
    unsigned gate_bytes;
    // filled by launch_conv_igemm (conv_igemm_prepare): K-order arithmetic, see chunk_to_tap()
    int taps4;                 // 4 * KH*KW
    int kfull_chunks;          // chunks covered by whole 32-channel groups = (Cin/32) * 4 * taps
    int kfull_c0;              // (Cin/32)*32: first channel of the partial group
    int knc;                   // chunks per tap in the partial group (Cin%32)/8
    int div_4t_mul;            // g / taps4      == (g * mul) >> 20
    int div_nc_mul;            // idx / knc      == (idx * mul) >> 8
    int div_kw_mul;            // tap / KW       == (tap * mul) >> 5
    int vec_epi;               // filled by launch_conv_igemm: 16-byte epilogue stores are legal
    int splitk;                // filled by launch_conv_igemm from the cfg word: K split over gridDim.z
    float* partial;            // split-K slab scratch [splitk][M][N] (null = split-K unavailable)
    size_t partial_floats;
    // Self-reducing split-K (round 4, ticket.h): the K slices of an output region draw tickets on tickets[region]; the last one sums the
    // slabs in slice order and runs the epilogue itself - no splitk_reduce_kernel launch.  Chosen per launch by the cfg word (split code
    // 64 + S); needs `tickets` (>= CONV_TICKETS counters, zero between launches).
    unsigned* tickets;
    int split_fused;           // filled by launch_conv_igemm
    // exact division by W and H for the transposed-conv pixel shuffle (filled by conv_igemm_prepare): q = (umulhi(n, mul) + n) >> shift
    unsigned fdw_mul, fdw_shift, fdh_mul, fdh_shift;
    // Winograd F(2x2,3x3) form of a 3x3 / stride 1 / pad 1 layer (kernels_wino.hip); null = not packed
    const float* wino_w;       // [Cin/8][16][wino_n16][128]
    int wino_n16;              // ceil(N / 16)
    int wino_nb, wino_n16_off; // F(4x4) sub-launches: n-blocks this launch covers (0 = all), first 16-channel slice
    unsigned wino_bytes;
    // Winograd F(4x4,3x3) form (kernels_wino4.hip), packed for the wide decoder layers only; null = not packed
    const float* wino4_w;      // [ceil(Cin/16)*4 k-steps][9 xi quads][wino_n16][64 lanes][4]
    unsigned wino4_bytes;
    // xi-split F(4x4) form for layers of <= 128 output channels (kernels_wino4x.hip); null = not packed
    const float* wino4x_w;     // [k-step][wave][16-byte load][64 lanes][4]
    unsigned wino4x_bytes;
    int wino4x_cfg;            // configuration index the weights were packed for (conv_wino4x_config(N))
    // deep-K / narrow-N project GEMM (kernels_proj.hip): the 1x1 weights in MFMA fragment order, null = not packed
    const float* proj_w;       // [Cin / 16 steps][ceil(N / 16) column tiles][64 lanes][4]
    unsigned proj_bytes;
    // split F(4x4) form (kernels_wino4p.hip): per-stream scratch for the pre-transformed input V = B^T d B (null = unavailable)
    float* wino4_v;
    size_t wino4_v_floats;
    // Squeeze-excite in the prologue of the latency-form project GEMM (kernels_proj.hip, batch <= 4; model.py:113-118): instead of a gate
    // vector the launch gets the fused front kernel's per-item squeeze rows (ticket.h: SeTicket::sqpart) and the excite weights, and
    // every wave computes the gates of ITS K slice while its operands are in flight.  se_rows == null: `gate` as usual.
    const float* se_rows;      // [B][se_nrows][se_sq]
    int se_nrows, se_sq;
    float se_inv_hw;
    const float* se_b1;        // [SQ]
    const float* se_w2;        // [SQ][Cin]
    const float* se_b2;        // [Cin]
};

// tile = 0 picks automatically from (M, N); otherwise one of the TILE_* ids.
enum { TILE_AUTO = 0 };
static constexpr int CONV_TICKETS = 8192;   // ticket counters a convolution launch may use (output regions of a split-K launch)
static constexpr int SPLIT_FUSED = 64;       // split code 64 + S: S slices, reduced by the last arriver (S <= 32)      // tile ids are 1..conv_igemm_num_tiles(); a launch cfg word is tile | (splitk << 8)
int conv_igemm_num_tiles();
double conv_igemm_tile_util(const ConvParams& p, int tile);
long long conv_igemm_tile_blocks(const ConvParams& p, int tile);
int conv_igemm_prepare(ConvParams& p);
int conv_igemm_k_index(int cin, int taps, int tap, int c);   // packed-weight column of (tap, channel)
int launch_conv_igemm(const ConvParams& p, int tile, hipStream_t s);   // 0, or -1 for unsupported geometry
int conv_igemm_npad();       // row padding of packed weights (multiple every tile divides)
bool conv_igemm_tile_is_bf16x3(int tile);
bool conv_igemm_tile_can_fuse_split(int tile);   // the kernel reduces its own split-K when given the split code SPLIT_FUSED + S
void launch_splitk_reduce(const ConvParams& p, hipStream_t s);
struct Bf16x3Tile { int bm, bn; const char* name; void (*launch)(const ConvParams&, hipStream_t); };
int bf16x3_num_tiles();
const Bf16x3Tile* bf16x3_tile(int i);
struct WinoTile { int bm, bn; const char* name; void (*launch)(const ConvParams&, hipStream_t); int f; bool pre; int xcfg; };   // f: output tile edge, 2 or 4; pre: split form (V pre-transformed); xcfg >= 0: xi-split configuration
int wino_num_tiles();
const WinoTile* wino_tile(int i);
bool conv_wino_supported(const ConvParams& p);
bool conv_wino4_supported(const ConvParams& p);
bool conv_wino_tile_supported(const ConvParams& p, int tile);   // the form tile id `tile` needs is packed and the layer is shaped for it
bool conv_igemm_tile_is_wino4(int tile);
void launch_wino4_64(const ConvParams& p, hipStream_t s);
void launch_wino4_128(const ConvParams& p, hipStream_t s);
bool conv_wino4_tail_applied();
void launch_wino4p_64(const ConvParams& p, hipStream_t s);    // split form: input transform kernel + matrix kernel
void launch_wino4p_128(const ConvParams& p, hipStream_t s);
bool conv_wino4p_tail_applied();
bool conv_wino4p_supported(const ConvParams& p);
bool conv_igemm_tile_is_wino4p(int tile);
bool conv_igemm_tile_is_wino4x(int tile);
int conv_igemm_tile_wino4x_cfg(int tile);   // xi-split configuration of the tile, -1 for other tiles
void launch_wino4x(const ConvParams& p, hipStream_t s);       // xi-split form (kernels_wino4x.hip)
bool conv_wino4x_supported(const ConvParams& p);
int conv_wino4x_config(int N);
size_t conv_wino4x_pack(int N, int cin, const std::function<float(int, int, int)>& get, std::vector<float>& out, int* cfg_out);
size_t conv_wino4_pack(int N, int cin, const std::function<float(int, int, int)>& get, std::vector<float>& out);
bool conv_igemm_tile_is_wino(int tile);
size_t conv_wino_pack(int N, int cin, const std::function<float(int, int, int)>& get, std::vector<float>& out, int* n16_out);
// pointwise persistent kernel (kernels_pw.hip): 1x1 convs / k2s2 transposed convs with K <= 512
struct PwTile { int bm, bn; const char* name; void (*launch)(const ConvParams&, hipStream_t); int proj_rt; };   // proj_rt > 0: a kernels_proj.hip tile of that many row tiles
int pw_num_tiles();
const PwTile* pw_tile(int i);
bool conv_pw_supported(const ConvParams& p);
bool conv_pw_fits(int bn, int kpad);
bool conv_pw_tile_ok(int i, const ConvParams& p);   // tile i of the family can run this launch
int conv_pw_tile_proj_rt(int i);                    // row-tile code of a kernels_proj.hip tile (>= 100: its latency form), 0 for the others
int conv_igemm_tile_proj_rt(int tile);              // the same by tile id
int conv_proj_lat_tile();                           // tile id of conv_projl_1 (the latency form with one column tile per workgroup)
// deep-K project GEMM (kernels_proj.hip)
bool conv_proj_supported(const ConvParams& p, int rt);
void launch_proj(const ConvParams& p, int rt, hipStream_t s);
bool conv_proj_wanted(int N, int cin);
bool conv_proj_has(int rt, int N);
size_t conv_proj_pack(int N, int cin, const std::function<float(int, int)>& get, std::vector<float>& out, int taps = 1);
bool conv_proj_lat_wanted(int taps, int KH, int KW, int cinp);   // deep-K 1x1 / k2s2 layers the latency form may serve
bool conv_igemm_tile_is_pw(int tile);
bool conv_igemm_tile_is_proj(int tile);   // a kernels_proj.hip tile (member of the pointwise family)
int conv_igemm_last_tile();  // tile id of the most recent launch on this thread (then reset to 0)
const char* conv_igemm_tile_name(int tile);

// ---------------------------------------------------------------------------------------------
// Encoder pieces
// ---------------------------------------------------------------------------------------------
struct StemParams {
    const float* in;           // NCHW [B,3,H,W]
    int B, H, W, OH, OW;
    int pad_t, pad_l;
    int circular;
    const float* w;            // [27][32]  k = (c*3+ky)*3+kx, BN folded
    const float* bias;         // [32]
    float* out;                // NHWC [B,OH,OW,32]
};
void launch_stem(const StemParams& p, hipStream_t s);

// Stem + block 0's depthwise conv in one launch (block 0 has no expand conv: its depthwise conv reads the stem output directly)
struct StemDwParams {
    StemParams st;             // st.out unused
    const float* wd;           // [9][32] depthwise taps, BN folded
    const float* bd;           // [32]
    float* out;                // NHWC [B,OH,OW,32]: swish(dw(swish(stem)))
    float* pool_partial;       // [B][S][32] partial sums of `out`, S = stem_dw_tiles
    SeTicket se;               // se.counter != null: the workgroup that finishes a sample last runs its squeeze-excite (ticket.h)
};
int stem_dw_tiles(int OH, int OW);   // pooling partial rows per sample
void launch_stem_dw(const StemDwParams& p, hipStream_t s);

struct DwParams {
    const float* in;           // NHWC [B,H,W,C]
    int B, H, W, C, OH, OW;
    int k, stride, pad_t, pad_l, circular;
    const float* w;            // [k*k][C], BN folded
    const float* bias;         // [C]
    float* out;                // NHWC [B,OH,OW,C], swish applied
    float* pool_partial;       // [B][S][C] partial sums of `out` for squeeze-excite
    int S;                     // strip lanes per sample
    SeTicket se;               // se.counter != null: squeeze-excite by the last-arriving workgroup of a sample (ticket.h)
};
void launch_depthwise(const DwParams& p, hipStream_t s);
int depthwise_strip_lanes(int B, int OH, int OW, int C, int k, int stride);

// Fused expand (1x1 + BN + swish) + depthwise (k x k + BN + swish) + SE pooling partials (kernels_mbconv.hip)
struct MbFrontParams {
    const float* x;            // NHWC [B,H,W,Cin]
    int B, H, W, Cin, cinp;    // cinp = Cin rounded up to 16 (weights zero padded)
    int mid;                   // expanded channels, multiple of 48
    const float* we;           // [mid][cinp], BN folded
    const float* be;           // [mid]
    const float* wd;           // [k*k][mid], BN folded
    const float* bd;           // [mid]
    int k, s, pad_t, pad_l, circular, OH, OW;
    float* out;                // NHWC [B,OH,OW,mid]
    float* pool;               // [B][tiles][mid]
    SeTicket se;               // se.counter != null: squeeze-excite by the last-arriving workgroup of a sample (ticket.h); the kernels
                               // that take it are named by the *_ticket_rows functions below (0 = this launch cannot)
    int spread;                // latency plans: work items a launch may spread to by cutting its work finer (more strips of the image-resident
                               // form, channel ranges per tile group of the wave form); 0 = never.  Set by the plan (CCVPE_FRONT_SPREAD, default 128)
};
void launch_mbconv_front(const MbFrontParams& p, hipStream_t s);
int mbconv_front_tiles(int k, int s, int OH, int OW);
// pooling partial rows per sample when the launch runs with a ticket (the wave-local form then pre-reduces four tiles per workgroup);
// 0 = the kernel launch_mbconv_front / launch_mbconv_image would pick for these parameters does not take a ticket
int mbconv_front_ticket_rows(const MbFrontParams& p);
int mbconv_image_ticket_rows(const MbFrontParams& p);
int mbconv_front_ticket_split(const MbFrontParams& p);   // workgroups per tile group of the wave form (channel ranges): tickets per row
bool mbconv_front_supported(int k, int s, int cin, int mid);
bool mbconv_front_profitable(int k);
// image-resident form (kernels_mbimg.hip): the whole image, or horizontal strips of it, per 16-channel chunk in LDS
bool mbconv_image_supported(const MbFrontParams& p);
int mbconv_image_strips(const MbFrontParams& p);   // pooling partial rows per sample ([B][strips][mid])
void launch_mbconv_image(const MbFrontParams& p, hipStream_t s);

struct SeParams {
    const float* pool_partial; // [B][S][C]
    int B, S, C, SQ;
    float inv_hw;
    const float* w1;           // [SQ][C]
    const float* b1;           // [SQ]
    const float* w2;           // [SQ][C] (transposed at pack time)
    const float* b2;           // [C]
    float* gate;               // [B][C] sigmoid(...)
    float* pooled;             // [B][SC][C] scratch: second-stage partial sums
    int SC;                    // second-stage split of the S partial rows (<= 16)
    float* sq;                 // [B][SQ] scratch: squeezed activations
};
void launch_se(const SeParams& p, hipStream_t s);

// ground descriptor: d_k[b][w*c_k + ch] = b2_k + sum_h wh_k[h] * Y[b,h,w,off_k+ch]
struct GrdDescParams {
    const float* y;            // NHWC [B,Hf,Wf,Ntot] (1x1 conv output incl. bias)
    int B, Hf, Wf, Ntot;
    int nlev;
    int c[6], off[6];          // per level head width and channel offset in y
    const float* wh[6];        // [Hf]
    float b2[6];
    float* desc;               // [B][Ltot] levels back to back
    int loff[6];               // element offset of level k inside a sample's row
    int Ltot;
};
void launch_grd_desc(const GrdDescParams& p, hipStream_t s);

// ---------------------------------------------------------------------------------------------
// Rolling match + L2 normalise + concat
// ---------------------------------------------------------------------------------------------
struct MatchParams {
    const float* x;            // NHWC [B,HW,C] (ld = x_ld)
    int x_ld;
    int B, HW, C;
    const float* g;            // [B][g_ld] descriptor, first L used
    int g_ld, L;
    int R;
    int shift[32];             // window_r[c] = x[(c + shift[r]) mod C]
    uint32_t inmax;            // rolls that take part in the max
    float* ms;                 // NCHW [B,R,HW] or null
    // loc concat buffer: ch0 = max, ch1..7 = 0, ch 8.. = normalised x
    float* cat_max;  int cat_max_ld;
    // ori concat buffer: ch0..R-1 = scores, zero pad to rpad, then normalised x
    float* cat_all;  int cat_all_ld;  int rpad;
    int P;                     // pixels per block (power of two, divides HW)
    float* gg_scratch;         // [B][match_scratch_floats(C)]: rolled descriptor of the small-C register form / Gm, Mk of the MFMA form (null = LDS form only)
    int prep_done;             // 1: launch_match_prep already ran on these parameters (launch_match then skips its preparation launch)
    int no_wide;               // 1: never the latency forms (sixteen waves per workgroup / four waves per 64 pixels) - CCVPE_MATCH_WIDE=0, read per plan
};
void launch_match(const MatchParams& p, hipStream_t s);
// The preparation launch alone (rolled descriptor / Gm, Mk into gg_scratch): it depends on the ground descriptor only, so a plan may issue it
// long before the aerial features exist.  Same form decision as launch_match when given the same parameters; no launch for the LDS form.
void launch_match_prep(const MatchParams& p, hipStream_t s);
void launch_match_prep_all(const MatchParams* ps, int n, hipStream_t s);   // the preparation of up to six levels in one launch
int match_pixels_per_block(int HW, int C);
size_t match_scratch_floats(int C);   // per-sample floats of MatchParams::gg_scratch

// ---------------------------------------------------------------------------------------------
// Tail: 16 -> {1,2} 3x3 conv to NCHW (+ unit-normalise for ori), softmax, post-processing
// ---------------------------------------------------------------------------------------------
struct TailConvParams {
    const float* in;           // NHWC [B,H,W,16]
    int B, H, W;
    const float* w;            // [9][16][cout]
    float bias[2];
    int cout;                  // 1 or 2
    int normalize;             // L2 normalise over the cout channels (F.normalize, eps 1e-12)
    float* out;                // NCHW [B,cout,H,W]
    float* raw;                // optional NCHW un-normalised copy (debug tap) or null
};
void launch_tail_conv(const TailConvParams& p, hipStream_t s);

// Fused last decoder level (kernels_level1.hip): deconv k2s2 (cx -> 16) + conv3x3 (16 -> 16) + ReLU +
// conv3x3 (16 -> cout) [+ L2 normalise], NHWC input at H/2 x W/2, NCHW output at H x W.
struct Level1Params {
    const float* x;            // NHWC [B, H/2, W/2, x_ld], first cx channels used
    int x_ld, cx, cxp;         // cxp = cx rounded up to 16 (weights zero padded)
    int B, H, W;               // output geometry (512 x 512)
    const float* wd;           // [64][cxp]  n = (dy*2+dx)*16 + o
    const float* bd;           // [16]
    const float* wa;           // [16][144]  k = (ky*3+kx)*16 + c
    const float* ba;           // [16]
    const float* wt;           // [9][cout][16]
    float bt[2];
    int cout, normalize;
    float* out;                // NCHW [B, cout, H, W]
    float* raw;                // optional un-normalised copy (debug tap)
};
void launch_level1(const Level1Params& p, hipStream_t s);
bool level1_supported(int cxp);   // input channel count the fused kernel takes

struct SoftmaxParams {
    const float* logits;       // [B][n]
    int B, n;
    float* partial;            // [B][chunks][2]
    int chunks;
    float* out;                // [B][n]
};
void launch_softmax(const SoftmaxParams& p, hipStream_t s);

struct PreprocParams {
    const unsigned char* in;   // [B][H][W][3] uint8
    int B, H, W, crop_w;
    const int* shift;          // [B] roll in pixels (out[x] = in[(x - shift) mod W]) or null
    float mean[3], stdv[3];
    float* out;                // [B][3][H][crop_w] fp32
};
void launch_preprocess(const PreprocParams& p, hipStream_t s);
// resize (PIL bilinear, byte-exact) + ToTensor + Normalize + roll + crop (kernels_preproc.hip)
struct ResizeParams {
    const unsigned char* in;   // [B][IH][IW][3] uint8
    int B, IH, IW, OH, OW, crop_w;
    unsigned char* tmp;        // [B][IH][OW][3] uint8 scratch (the horizontally resampled image)
    const int* shift;          // [B] roll in output pixels or null
    float mean[3], stdv[3];
    float* out;                // [B][3][OH][crop_w] fp32
};
int launch_resize(const ResizeParams& p, hipStream_t s);   // -1: down-scaling factor above 8

void launch_scatter_channels(const float* src, int C, long long P, Dst d0, Dst d1, int ndst, hipStream_t s);

struct PoseOut { int32_t index; float prob, cos_v, sin_v, angle_deg; };
static constexpr int PP_MAX_BATCH = 4096;           // samples per launch_postprocess call
size_t postprocess_scratch_bytes(int B);            // partial (max, index) pairs + one ticket counter per sample (the counters zero before the first launch)
void launch_postprocess(const float* heat, const float* ori, int B, int n, PoseOut* out, float* rows, void* scratch, hipStream_t s);   // rows != null: [B][5] floats instead of `out`

struct MetricsOut { double pixel_distance, meter_distance, prob_at_gt, angle_pred_deg, angle_gt_deg, orientation_error_deg, longitudinal_m, lateral_m; };
void launch_metrics(const PoseOut* pose, const float* heat, int B, int W, int n, const int* gt_index, const float* gt_cos_sin,
                    const double* meter_per_pixel, const double* heading_deg, MetricsOut* out, hipStream_t s);

// several device-to-device copies in ONE launch (float counts, all pointers 16-byte aligned, counts multiples of 4): the staging
// copies around a hipGraph replay - 11 runtime copy launches of ~5 us each per batch-1 frame otherwise
struct MultiCopy { const float* src[12]; float* dst[12]; unsigned long long n[12]; int count; };
void launch_multi_copy(const MultiCopy& mc, hipStream_t s);
void launch_fill_random(float* p, size_t n, uint32_t seed, hipStream_t s);   // ~N(0,1) floats (autotune operands)
void launch_nhwc_to_nchw(const float* in, int in_ld, int coff, int C, int B, int HW, float* out, hipStream_t s);

}  // namespace ccvpe
