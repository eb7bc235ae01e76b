/*
 * ccvpe.h - C ABI of libccvpe_hip.so: the MI355X (gfx950) inference forward pass of CCVPE.
 *
 * The reference is pure Python/PyTorch and has no native interface of its own; the hot path it
 * exposes is `CVM_*.forward(grd, sat)` (reference models.py:150, 448, 752, 1051) on an nn.Module
 * whose parameters arrive through `load_state_dict` (reference train_VIGOR.py:248-254).  This
 * header is what a binding for that path binds:
 *
 *   reference interface                                   replaced by
 *   ----------------------------------------------------  -----------------------------------------
 *   CVM_VIGOR(device, circular_padding)        models.py:50    ccvpe_create (variant 0)
 *   CVM_VIGOR_ori_prior(device, ori_noise, circ) models.py:347 ccvpe_create (variant 1)
 *   CVM_KITTI(device)                          models.py:656   ccvpe_create (variant 2)
 *   CVM_OxfordRobotCar(device)                 models.py:955   ccvpe_create (variant 3)
 *   load_state_dict(state_dict)          train_VIGOR.py:252    ccvpe_set_weight x818 + ccvpe_finalize_weights
 *   forward(grd, sat) -> 9-tuple   models.py:150,448,752,1051  ccvpe_forward
 *   per-sample argmax / (cos,sin) lookup train_VIGOR.py:297-316 ccvpe_postprocess
 *
 * Conventions
 *   - all tensors are float32; inputs/outputs are NCHW-contiguous exactly as the reference's
 *     forward receives and returns them; pointers are DEVICE pointers borrowed from the caller
 *     (e.g. torch.Tensor.data_ptr()) and are never freed or retained by the library;
 *   - every function returns 0 on success or a negative CCVPE_E* code; ccvpe_last_error() gives
 *     a thread-local human-readable message; nothing throws across the ABI;
 *   - a handle is bound to one HIP device and is not re-entrant; launches go to the caller's
 *     stream, asynchronously, with no hidden synchronisation once the plan for a given
 *     (batch, ground size) exists (the first call with a new shape allocates workspace);
 *   - there is no CPU fallback: without a gfx950 device every compute entry point fails.
 */
#ifndef CCVPE_H
#define CCVPE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CCVPE_OK 0
#define CCVPE_EINVAL (-1)   /* bad argument / shape */
#define CCVPE_ESTATE (-2)   /* call order (e.g. forward before finalize) */
#define CCVPE_EHIP (-3)     /* HIP runtime error */
#define CCVPE_ENOMEM (-4)
#define CCVPE_EKEY (-5)     /* unknown / missing state_dict key */

#define CCVPE_VARIANT_VIGOR 0
#define CCVPE_VARIANT_VIGOR_ORI_PRIOR 1
#define CCVPE_VARIANT_KITTI 2
#define CCVPE_VARIANT_OXFORD 3

#define CCVPE_OUT_HW 512       /* localisation / orientation maps are 512 x 512 (models.py:319-341) */
#define CCVPE_SAT_HW 512       /* aerial input is 3 x 512 x 512 */

typedef struct ccvpe_handle_s* ccvpe_handle;

typedef struct ccvpe_config {
    int32_t variant;            /* CCVPE_VARIANT_* */
    int32_t circular_padding;   /* ground encoder: horizontal circular padding (models.py:53, utils.py:330-358) */
    float   ori_noise;          /* variant 1 only: rolls i = -n..n, n = int(ori_noise/18) (models.py:489) */
    int32_t device;             /* HIP device ordinal */
    int32_t micro_batch;        /* samples per internal pass (0 = library default); larger batches loop */
    int32_t reserved[3];        /* reserved[0] = precision mode of the dense contractions:
                                   0 exact fp32 MFMA (default); 1 "bf16x3": fp32 operands split into two bf16,
                                   three bf16 MFMAs per product, fp32 accumulate (~2^-16 relative per term) */
} ccvpe_config;

/* Caller-allocated outputs of one forward call (device memory, NCHW contiguous, batch-major).
 * Channel counts of ms[k] come from ccvpe_output_channels(). */
typedef struct ccvpe_outputs {
    float* logits_flattened;    /* [B, 512*512]                                  models.py:319 */
    float* heatmap;             /* [B, 1, 512, 512] softmax over all pixels       models.py:320 */
    float* ori;                 /* [B, 2, 512, 512] unit (cos, sin) field         models.py:341 */
    float* matching_score[6];   /* level k: [B, R_k, 8*2^k, 8*2^k], k = 0..5      models.py:343 */
} ccvpe_outputs;

/* Compact per-sample result of the test-loop post-processing (train_VIGOR.py:297-316). */
typedef struct ccvpe_pose {
    int32_t index;              /* argmax of the heatmap, y*512 + x */
    float   prob;               /* heatmap value there */
    float   cos_v, sin_v;       /* orientation field at that pixel */
    float   angle_deg;          /* acos/sign rule of train_VIGOR.py:307-311, in [0, 360) */
} ccvpe_pose;

/* Per-query evaluation record of the reference test loops (train_VIGOR.py:296-326, train_KITTI.py:309-343), double
 * precision like their numpy / math arithmetic.  NaN = the reference does not produce that number for the query. */
typedef struct ccvpe_metrics {
    double pixel_distance;         /* |argmax(gt) - argmax(heatmap)| in output pixels          train_VIGOR.py:299 */
    double meter_distance;         /* pixel_distance * meter_per_pixel[b]                       :301-309 */
    double prob_at_gt;             /* heatmap at the ground-truth pixel                         :326 */
    double angle_pred_deg;         /* acos / sign rule on the predicted (cos, sin)              :306-311 */
    double angle_gt_deg;           /* the same on the ground-truth (cos, sin)                   :312-317 */
    double orientation_error_deg;  /* min(|gt - pred|, 360 - |gt - pred|)                       :319 */
    double longitudinal_m;         /* KITTI / Oxford: error along the driving direction         train_KITTI.py:322-324 */
    double lateral_m;              /* ... and across it                                          train_KITTI.py:323-325 */
} ccvpe_metrics;

const char* ccvpe_last_error(void);
const char* ccvpe_version(void);
/* Kernel launches this THREAD has issued through the library so far (every entry point; eager launches - a replayed hipGraph issues
 * none).  The difference around a call = its launches: bench.py's `launches_per_frame`.  No reference counterpart. */
uint64_t ccvpe_launch_count(void);

int ccvpe_create(const ccvpe_config* cfg, ccvpe_handle* out);
int ccvpe_destroy(ccvpe_handle h);

/* One state_dict entry.  `data` may be a host or a device pointer (float32, contiguous, `shape`
 * as in the reference state_dict); BatchNorm `num_batches_tracked` (int64) entries and the unused
 * `_fc.*` classifier are accepted and ignored via ccvpe_skip_weight.  Returns CCVPE_EKEY for a key
 * that is not part of the 818-key layout, CCVPE_EINVAL for a shape mismatch. */
int ccvpe_set_weight(ccvpe_handle h, const char* key, const float* data, const int64_t* shape, int32_t ndim);
int ccvpe_skip_weight(ccvpe_handle h, const char* key);
/* Folds BatchNorm into the adjacent convolutions, repacks everything into the kernels' layouts and
 * uploads it.  Fails with CCVPE_EKEY (message lists the first missing key) if any key is unset. */
int ccvpe_finalize_weights(ccvpe_handle h);

/* Packed-weight cache (reference train_VIGOR.py:252 reloads and, here, re-packs the checkpoint on every start): after
 * ccvpe_finalize_weights, ccvpe_save_packed writes the folded / repacked / Winograd-transformed device weights to `path`;
 * in a later process ccvpe_load_packed(h, path) on a handle created with the same config replaces the 818 ccvpe_set_weight
 * calls and ccvpe_finalize_weights.  The caller keys the file by the checkpoint's content (see ccvpe_amd/models.py).
 * CCVPE_EINVAL if the file is missing, truncated, or was packed for another variant / precision / library build. */
int ccvpe_save_packed(ccvpe_handle h, const char* path);
int ccvpe_load_packed(ccvpe_handle h, const char* path);
/* The environment switches that change what the packer emits, as "NAME=value;" pairs in a fixed order ("" when none is set;
 * thread-local storage, valid until the thread's next call): part of the caller's cache key, so a file packed under one setting
 * is never loaded under another.  No reference counterpart. */
const char* ccvpe_pack_switches(void);

/* Tuning table.  The first forward of a new (batch, ground size) measures every tiled launch of its plan with each candidate
 * kernel tile and keeps the fastest (1-2 s at batch 32); the choices are the only thing about a forward that can differ between
 * two processes.  ccvpe_export_tuning writes them as text (one "op <launch key> <tile name> <split>" line per launch; call with
 * buf = NULL to get the size in *needed), ccvpe_import_tuning loads such text (returns the number of entries read): a launch
 * found in the table is not measured again and runs exactly as recorded, so two handles with the same table produce
 * bit-identical results.  ccvpe_tuning_generation counts the plans this handle has measured launches of (the host side saves
 * the table when it grows; see ccvpe_amd/tuning.py).  No reference counterpart. */
int ccvpe_import_tuning(ccvpe_handle h, const char* text);
int ccvpe_export_tuning(ccvpe_handle h, char* buf, size_t capacity, size_t* needed);
int ccvpe_tuning_generation(ccvpe_handle h);

/* Largest micro_batch whose intermediate tensors all stay below the 2 GiB the kernels address with 32-bit byte offsets
 * (ccvpe_forward refuses a larger one with CCVPE_EINVAL instead of wrapping offsets).  Pure host arithmetic, no device
 * needed.  Negative on a ground size the variant's descriptor heads cannot take. */
int ccvpe_max_micro_batch(int32_t variant, float ori_noise, int32_t grd_h, int32_t grd_w);

/* R_k of matching_score[level] (level 0..5) for this handle's variant / ori_noise. */
int ccvpe_output_channels(ccvpe_handle h, int32_t level);
/* Device workspace the library holds for a (batch, ground size) plan, in bytes (0 on error). */
size_t ccvpe_workspace_bytes(ccvpe_handle h, int32_t batch, int32_t grd_h, int32_t grd_w);

/* forward(grd, sat): grd [B,3,grd_h,grd_w], sat [B,3,512,512], device pointers, NCHW float32.
 * `stream` is a hipStream_t (NULL = default stream). */
int ccvpe_forward(ccvpe_handle h, const float* grd, int32_t grd_h, int32_t grd_w, const float* sat,
                  int32_t batch, const ccvpe_outputs* out, void* stream);

/* Device-side test-loop post-processing on forward outputs: poses[B] is DEVICE memory.  batch <= 4096 per call; one call in flight per
 * handle (the launch keeps per-sample partial results and ticket counters in a scratch buffer of the handle). */
int ccvpe_postprocess(ccvpe_handle h, const float* heatmap, const float* ori, int32_t batch,
                      ccvpe_pose* poses, void* stream);
/* The same five numbers per query as one float row - rows[B][5] = (index, prob, cos, sin, angle_deg), DEVICE memory: the 20-byte
 * record a data-parallel evaluation gathers (train_VIGOR.py:297-316 keeps them in Python lists).  batch <= 4096. */
int ccvpe_postprocess_rows(ccvpe_handle h, const float* heatmap, const float* ori, int32_t batch,
                           float* rows, void* stream);

/* Ground-truth side of the same test loop, on device: `poses` from ccvpe_postprocess, `gt_index[B]` = flat index of
 * argmax(gt map) (y*512 + x), `gt_cos_sin[B][2]` = ground-truth orientation at that pixel (NULL: no orientation error),
 * `meter_per_pixel[B]` = metres per OUTPUT pixel (VIGOR: city constant / 512 * 640, train_VIGOR.py:301-308; KITTI:
 * test_set.meter_per_pixel), `heading_deg[B]` = orientation_angle of train_KITTI.py:310 (NULL: no lateral / longitudinal
 * split).  All pointers are DEVICE memory; out[B] likewise. */
int ccvpe_eval_metrics(ccvpe_handle h, const ccvpe_pose* poses, const float* heatmap, int32_t batch, const int32_t* gt_index,
                       const float* gt_cos_sin, const double* meter_per_pixel, const double* heading_deg, ccvpe_metrics* out,
                       void* stream);

/* Aerial-side caching for streaming (reference datasets.py:306-317 reuses aerial tiles across frames, the
 * reference model still re-encodes them every call): encode once, then run ground encoder + matching +
 * decoders against the cached aerial encoding.  `cache` is caller-owned DEVICE memory of
 * ccvpe_aerial_cache_bytes(h, batch) bytes; batch <= micro_batch.  forward_cached(grd, encode(sat)) ==
 * forward(grd, sat). */
size_t ccvpe_aerial_cache_bytes(ccvpe_handle h, int32_t batch);
int ccvpe_encode_aerial(ccvpe_handle h, const float* sat, int32_t batch, void* cache, void* stream);
int ccvpe_forward_cached(ccvpe_handle h, const float* grd, int32_t grd_h, int32_t grd_w, const void* cache,
                         int32_t batch, const ccvpe_outputs* out, void* stream);

/* Input pre-processing on device (reference train_VIGOR.py:57-70 ToTensor + Normalize, datasets.py:118
 * torch.roll(grd, shift, dims=2), train_VIGOR.py:272-273 FoV crop): uint8 HWC images [B,H,W,3] (decoded and
 * resized on the host) -> float32 NCHW [B,3,H,crop_w] with out[..., x] = norm(in[..., (x - shift[b]) mod W, :]).
 * `shift` is DEVICE memory [B] or NULL; mean/std are host arrays of 3.  Bit-identical to torchvision's fp32
 * (x/255 - mean)/std. */
int ccvpe_preprocess(const uint8_t* hwc, int32_t batch, int32_t H, int32_t W, const int32_t* shift, int32_t crop_w,
                     const float mean[3], const float stdv[3], float* out_nchw, void* stream);

/* The same with the resize in front (reference train_VIGOR.py:57-70 transforms.Resize([320,640]) / Resize([512,512]), applied
 * to PIL images at datasets.py:106, i.e. PIL.Image.resize(BILINEAR)): uint8 HWC images [B,in_h,in_w,3] as decoded ->
 * Pillow's 8-bit bilinear resampler (triangle filter with support = the down-scaling factor, 22-bit fixed-point taps,
 * horizontal pass first, uint8 between the passes; byte-identical to PIL) -> ToTensor + Normalize + roll + crop as above ->
 * float32 NCHW [B,3,out_h,crop_w].  `scratch` is caller-owned DEVICE memory of batch*in_h*out_w*3 bytes (may be NULL when
 * in_w == out_w).  Down-scaling factors above 8 per axis are refused. */
int ccvpe_preprocess_resize(const uint8_t* hwc, int32_t batch, int32_t in_h, int32_t in_w, int32_t out_h, int32_t out_w,
                            const int32_t* shift, int32_t crop_w, const float mean[3], const float stdv[3], uint8_t* scratch,
                            float* out_nchw, void* stream);

/* Debug taps: when enabled, intermediate tensors of the next forward call stay resident and can be
 * copied out by name as NCHW float32 into HOST memory (`capacity` in floats).  Returns the number
 * of floats written via *n_out.  Names: see DESIGN.md (e.g. "sat_block15", "loc_level6"). */
int ccvpe_set_debug(ccvpe_handle h, int32_t enable);
/* Issue order of the next forward calls: 2 (default) = the aerial encoder and the orientation decoder run on an internal
 * second stream; 1 = everything in program order on the caller's stream.  Results are bit-identical (test hook and
 * diagnostic switch; no reference counterpart). */
int ccvpe_set_streams(ccvpe_handle h, int32_t n_streams);
int ccvpe_read_tap(ccvpe_handle h, const char* name, float* host_dst, size_t capacity, size_t* n_out,
                   int32_t shape_out[4]);

/* Diagnostic: after a synchronised forward, writes one text line per launch of its plan (name, stream, cross-stream
 * wait, tile) with an integer checksum of every tensor the launch touches.  Meaningful for plans whose tensors keep
 * their memory (debug plans, two-stream plans issued either way): diffing two dumps names the first launch whose
 * output differs.  No reference counterpart. */
int ccvpe_debug_dump_plan(ccvpe_handle h, const char* path);

/* Per-kernel device timing of the most recent ccvpe_profile_forward call: runs one forward with a
 * hipEvent pair around every launch and reports (name, milliseconds) rows.  Used by bench.py for
 * the roofline line.  Returns the number of rows; row i is copied into name_buf / ms. */
int ccvpe_profile_forward(ccvpe_handle h, const float* grd, int32_t grd_h, int32_t grd_w, const float* sat,
                          int32_t batch, const ccvpe_outputs* out, void* stream);
int ccvpe_profile_row(ccvpe_handle h, int32_t i, char* name_buf, size_t name_cap, float* ms,
                      double* flops, double* bytes);
/* FLOPs row i actually issued on the matrix pipe: M / N padded to the launch's tile, K to the packed depth, 16
 * products per 2x2 tile for the Winograd tiles, three MFMAs per product in bf16x3 mode (`flops` of
 * ccvpe_profile_row is the algorithmic direct-convolution count). */
int ccvpe_profile_row_issued(ccvpe_handle h, int32_t i, double* issued_flops);

/* Kernel-level hook for parity tests and tile tuning (not on the product path): one NHWC fp32
 * convolution through the implicit-GEMM MFMA kernel.  in [B,H,W,Cin] (device, Cin % 8 == 0),
 * w [Cout,Cin,KH,KW] and bias [Cout] in the reference's layout (host or device; bias may be NULL),
 * symmetric zero padding `pad`, out [B,OH,OW,Cout] (device) with OH = (H + 2*pad - KH)/stride + 1.
 * act: 0 none, 1 relu, 2 swish.  tile: 0 = heuristic, else an internal tile id (1..ccvpe_op_num_tiles()).
 * iters > 0 additionally times `iters` back-to-back launches with hipEvents and stores the mean
 * milliseconds in *ms. */
int ccvpe_op_num_tiles(void);
const char* ccvpe_op_tile_name(int32_t tile);   /* tile ids are 1..ccvpe_op_num_tiles() */
int ccvpe_op_conv2d(const float* in, int32_t B, int32_t H, int32_t W, int32_t Cin, const float* w, const float* bias,
                    int32_t Cout, int32_t KH, int32_t KW, int32_t stride, int32_t pad, int32_t act, int32_t tile,
                    float* out, int32_t iters, float* ms, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CCVPE_H */
