// C ABI of libccvpe_hip.so: handle, state_dict ingestion (BN folding + weight packing), execution
// plan with lifetime-based workspace reuse, forward orchestration.  See include/ccvpe.h.
//
// The orchestration restates CVM_*.forward (reference models.py:150-343, 448-652, 752-950, 1051-1244)
// as a static list of kernel launches over NHWC tensors; concatenations are channel-offset writes
// into pre-allocated buffers, the encoder taps are written by the producing GEMM's epilogue.
#include "ccvpe_internal.h"

// ------------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------------
namespace ccvpe { thread_local unsigned long long g_launches = 0; }
static thread_local std::string g_err_storage;
std::string& ccvpe_err() { return g_err_storage; }
int ccvpe_fail(int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err_storage = buf;
    return code;
}

// Integer checksum of a tensor's bytes (diagnostics: ccvpe_debug_dump_plan)
__global__ __launch_bounds__(256) void checksum_kernel(const uint32_t* __restrict__ p, size_t n, unsigned long long* out) {
    unsigned long long s = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) s += (unsigned long long)p[i] * (unsigned long long)((i & 1023) + 1);
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
    if ((threadIdx.x & 63) == 0) atomicAdd(out, s);
}


// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
extern "C" {

const char* ccvpe_last_error(void) { return ccvpe_err().c_str(); }
const char* ccvpe_version(void) { return "ccvpe-hip 0.1 (gfx950, fp32 MFMA)"; }
uint64_t ccvpe_launch_count(void) { return ccvpe::g_launches; }

int ccvpe_create(const ccvpe_config* cfg, ccvpe_handle* out) {
    if (!cfg || !out) return ccvpe_fail(CCVPE_EINVAL, "null argument");
    if (cfg->variant < 0 || cfg->variant > 3) return ccvpe_fail(CCVPE_EINVAL, "unknown variant %d", cfg->variant);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return ccvpe_fail(CCVPE_EHIP, "no HIP device visible: libccvpe_hip has no CPU fallback");
    if (cfg->device < 0 || cfg->device >= ndev) return ccvpe_fail(CCVPE_EINVAL, "device %d out of range (%d visible)", cfg->device, ndev);
    HIPCHK(hipSetDevice(cfg->device));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, cfg->device));
    if (std::string(prop.gcnArchName).find("gfx950") == std::string::npos)
        return ccvpe_fail(CCVPE_EHIP, "device %d is %s; this library carries gfx950 code objects only", cfg->device, prop.gcnArchName);
    auto* h = new ccvpe_handle_s();
    h->cfg = *cfg;
    if (h->cfg.micro_batch <= 0) h->cfg.micro_batch = 32;
    h->vs = make_variant(cfg->variant);
    if (const char* e = getenv("CCVPE_AUTOTUNE")) h->autotune = std::atoi(e) != 0;
    // the candidate-filter switches are diagnostics: with one of them set every plan is measured under it, table or not
    for (const char* sw : {"CCVPE_TUNE_PREFER_PW", "CCVPE_TUNE_PREFER_PROJ", "CCVPE_TUNE_SPLITK", "CCVPE_TUNE_NO_BF16X3", "CCVPE_TUNE_BF16_ONLY", "CCVPE_NO_PW", "CCVPE_TUNE_IGNORE_TABLE"})
        if (getenv(sw)) h->tuning_lookup = false;
    if (const char* e = getenv("CCVPE_GRAPH")) h->graph_mode = std::atoi(e) != 0;
    if (const char* e = getenv("CCVPE_FUSE_L1")) h->fuse_level1 = std::atoi(e) != 0;
    if (const char* e = getenv("CCVPE_WINOGRAD")) h->wino = std::atoi(e) != 0;
    if (const char* e = getenv("CCVPE_STREAMS")) h->two_streams = std::atoi(e) >= 2;
    if (const char* e = getenv("CCVPE_FUSE_MBCONV")) h->fuse_mbconv = std::atoi(e);
    if (const char* e = getenv("CCVPE_DIAG_SYNC_BEFORE")) h->diag_sync = e;
    if (const char* e = getenv("CCVPE_DIAG_SNAP")) h->diag_snap = e;
    if (const char* e = getenv("CCVPE_PRECISION")) h->cfg.reserved[0] = (std::string(e) == "bf16x3") ? 1 : 0;
    if (h->cfg.reserved[0] != 0 && h->cfg.reserved[0] != 1) { delete h; return ccvpe_fail(CCVPE_EINVAL, "unknown precision mode %d", cfg->reserved[0]); }
    const int n = (int)(cfg->ori_noise / 18.f);
    for (int k = 0; k < 6; ++k)
        h->rolls[k] = (cfg->variant == CCVPE_VARIANT_VIGOR_ORI_PRIOR && k > 0) ? 2 * n + 1 : h->vs.n_rolls;
    if (cfg->variant == CCVPE_VARIANT_VIGOR_ORI_PRIOR && (n < 0 || 2 * n + 1 > 32)) {
        delete h;
        return ccvpe_fail(CCVPE_EINVAL, "ori_noise %.1f out of range", cfg->ori_noise);
    }
    build_expect(h);
    *out = h;
    return 0;
}

int ccvpe_max_micro_batch(int32_t variant, float ori_noise, int32_t grd_h, int32_t grd_w) {
    if (variant < 0 || variant > 3) return ccvpe_fail(CCVPE_EINVAL, "unknown variant %d", variant);
    // a device-less stand-in handle: build_plan only sizes tensors and records launches, it never touches HIP.
    // fuse_mbconv = 0 sizes the unfused (largest) form of every MBConv block, so the bound holds for every plan.
    ccvpe_handle_s tmp;
    tmp.cfg.variant = variant; tmp.cfg.ori_noise = ori_noise; tmp.cfg.micro_batch = 1;
    tmp.vs = make_variant(variant);
    // fuse_level1 = false sizes the unfused last decoder level too (a handle may run with CCVPE_FUSE_L1=0 or a level-1 input the
    // fused kernel does not take), so the cap holds whatever flags the real handle has
    tmp.fuse_mbconv = 0; tmp.fuse_level1 = false; tmp.two_streams = false; tmp.graph_mode = 0;
    const int n = (int)(ori_noise / 18.f);
    for (int k = 0; k < 6; ++k) tmp.rolls[k] = (variant == CCVPE_VARIANT_VIGOR_ORI_PRIOR && k > 0) ? 2 * n + 1 : tmp.vs.n_rolls;
    int lo = 0, hi = 1024;   // invariant: lo fits (0 = nothing fits / bad geometry), hi does not
    while (hi - lo > 1) {
        const int mid = (lo + hi) / 2;
        Plan pl;
        if (build_plan(&tmp, pl, mid, grd_h, grd_w) == 0) lo = mid; else hi = mid;
    }
    if (lo == 0) return ccvpe_fail(CCVPE_EINVAL, "ground size %d x %d is not valid for variant %d: %s", grd_h, grd_w, variant, ccvpe_err().c_str());
    return lo;
}

int ccvpe_destroy(ccvpe_handle h) {
    if (!h) return 0;
    (void)hipSetDevice(h->cfg.device);
    for (void* p : h->dev_allocs) (void)hipFree(p);
    if (h->arena) (void)hipFree(h->arena);
    if (h->post_scratch) (void)hipFree(h->post_scratch);
    h->plans.clear();
    for (int k = 0; k < 2; ++k) if (h->snap[k]) (void)hipFree(h->snap[k]);
    if (h->capture_stream) (void)hipStreamDestroy(h->capture_stream);
    if (h->aux_stream) (void)hipStreamDestroy(h->aux_stream);
    delete h;
    return 0;
}

int ccvpe_output_channels(ccvpe_handle h, int32_t level) {
    if (!h || level < 0 || level > 5) return ccvpe_fail(CCVPE_EINVAL, "bad level");
    return h->rolls[level];
}

size_t ccvpe_workspace_bytes(ccvpe_handle h, int32_t batch, int32_t grd_h, int32_t grd_w) {
    if (!h || !h->finalized || batch <= 0) { ccvpe_fail(CCVPE_ESTATE, "handle not ready"); return 0; }
    Plan pl;
    const int mb = std::min(batch, h->cfg.micro_batch);
    if (build_plan(h, pl, mb, grd_h, grd_w)) return 0;
    return pl.total * sizeof(float);
}

// Issue a plan's ops: in order on one stream, or on two streams with event edges for the cross-stream dependencies.
// First issue of a plan: every tiled launch must run the (tile, split-K) its plan entry names.  launch_conv_igemm falls back to the shape
// heuristic when a tile cannot serve a launch (a hand-edited or foreign tuning table): legal, but then "same table -> same launches ->
// same bits" no longer holds, so it is said out loud once per plan.
static void check_issued_tile(Plan& pl, const Op& op) {
    const int got = conv_igemm_last_tile();
    if (!op.tile || (*op.tile & 0xff) == 0 || got == 0) return;
    const int want = *op.tile, ws = (want >> 8) & 0xff, gs = (got >> 8) & 0xff;
    if ((want & 0xff) != (got & 0xff) || (ws > 1 ? ws : 1) != (gs > 1 ? gs : 1))
        std::fprintf(stderr, "ccvpe: launch %s (batch %d) runs %s split %d instead of the planned %s split %d\n", op.name.c_str(), pl.B,
                     conv_igemm_tile_name(got), gs, conv_igemm_tile_name(want), ws);
}

static int run_ops(ccvpe_handle h, Plan& pl, const Ctx& base, hipStream_t s0) {
    const bool check = !pl.tiles_checked;
    pl.tiles_checked = true;
    if (check) (void)conv_igemm_last_tile();
    if (!pl.two_streams || h->serial_issue) {
        Ctx c = base;
        c.stream = s0;
        for (auto& op : pl.ops) { op.fn(c); if (check) check_issued_tile(pl, op); }
        base.conv_errors += c.conv_errors;
        return 0;
    }
    if (!h->aux_stream) HIPCHK(hipStreamCreateWithFlags(&h->aux_stream, hipStreamNonBlocking));
    if (pl.events.empty()) {
        pl.events.assign(pl.ops.size() + 2, nullptr);
        for (size_t i = 0; i < pl.events.size(); ++i)
            if (i >= pl.ops.size() || pl.ops[i].signal) HIPCHK(hipEventCreateWithFlags(&pl.events[i], hipEventDisableTiming));
    }
    Ctx c[2] = {base, base};
    hipStream_t st[2] = {s0, h->aux_stream};
    c[0].stream = st[0];
    c[1].stream = st[1];
    pl.set_scratch(c[1], 1);
    const size_t n = pl.ops.size();
    HIPCHK(hipEventRecord(pl.events[n], st[0]));            // fork: the second stream starts after the caller's prior work
    HIPCHK(hipStreamWaitEvent(st[1], pl.events[n], 0));
    for (size_t k = 0; k < n; ++k) {
        const size_t i = pl.issue_order.size() == n ? (size_t)pl.issue_order[k] : k;
        Op& op = pl.ops[i];
        for (int d : op.wait_on) HIPCHK(hipStreamWaitEvent(st[op.stream], pl.events[d], 0));
        if (!h->diag_sync.empty() && op.name.find(h->diag_sync) != std::string::npos) HIPCHK(hipDeviceSynchronize());
        const bool snap = !h->diag_snap.empty() && op.name == h->diag_snap;
        auto take_snap = [&](int which) -> int {
            if (!h->snap[0]) {
                h->snap_layout.clear();
                size_t o = 0;
                for (int id : op.uses) { h->snap_layout.push_back({id, o}); o += (pl.size[id] + 63) & ~(size_t)63; }
                h->snap_floats = o;
                for (int k = 0; k < 2; ++k) HIPCHK(hipMalloc((void**)&h->snap[k], o * sizeof(float)));
            }
            for (auto& e : h->snap_layout)
                HIPCHK(hipMemcpyAsync(h->snap[which] + e.second, base.arena + pl.off[e.first], pl.size[e.first] * sizeof(float), hipMemcpyDeviceToDevice, st[op.stream]));
            return 0;
        };
        if (snap) { if (int r = take_snap(0)) return r; }
        op.fn(c[op.stream]);
        if (check) check_issued_tile(pl, op);
        if (snap) { if (int r = take_snap(1)) return r; }
        if (op.signal) HIPCHK(hipEventRecord(pl.events[i], st[op.stream]));
    }
    HIPCHK(hipEventRecord(pl.events[n + 1], st[1]));        // join
    HIPCHK(hipStreamWaitEvent(st[0], pl.events[n + 1], 0));
    base.conv_errors += c[0].conv_errors + c[1].conv_errors;
    return 0;
}

static int run_forward(ccvpe_handle h, const float* grd, int gh, int gw, const float* sat, int batch,
                       const ccvpe_outputs* out, hipStream_t stream, bool profile, const float* cache = nullptr) {
    const int mode = cache ? 2 : 0;
    if (!h || !grd || (!sat && !cache) || !out) return ccvpe_fail(CCVPE_EINVAL, "null argument");
    if (cache && batch > h->cfg.micro_batch) return ccvpe_fail(CCVPE_EINVAL, "cached forward needs batch <= micro_batch (%d)", h->cfg.micro_batch);
    if (!h->finalized) return ccvpe_fail(CCVPE_ESTATE, "ccvpe_finalize_weights has not been called");
    if (batch <= 0) return ccvpe_fail(CCVPE_EINVAL, "batch must be positive");
    if (!out->logits_flattened || !out->heatmap || !out->ori) return ccvpe_fail(CCVPE_EINVAL, "null output buffer");
    for (int k = 0; k < 6; ++k) if (!out->matching_score[k]) return ccvpe_fail(CCVPE_EINVAL, "null matching_score[%d]", k);
    HIPCHK(hipSetDevice(h->cfg.device));
    if (profile) h->prof.clear();
    int mbmax = h->cfg.micro_batch;
    {   // never build a plan with a tensor of 2 GiB or more (32-bit byte offsets): larger batches loop
        auto it = h->mb_cap.find({gh, gw});
        if (it == h->mb_cap.end()) {
            const std::string keep = ccvpe_err();
            const int cap = ccvpe_max_micro_batch(h->cfg.variant, h->cfg.ori_noise, gh, gw);
            ccvpe_err() = keep;
            it = h->mb_cap.emplace(std::make_pair(gh, gw), cap).first;
        }
        if (it->second > 0) mbmax = std::min(mbmax, it->second);
    }
    // make sure every plan (and the largest arena) exists before the first launch
    for (int done = 0; done < batch;) {
        const int mb = std::min(mbmax, batch - done);
        Plan* pl; int rc = get_plan(h, mb, gh, gw, &pl, mode);
        if (rc) return rc;
        done += mb;
    }
    const size_t npx = (size_t)CCVPE_OUT_HW * CCVPE_OUT_HW;
    for (int done = 0; done < batch;) {
        const int mb = std::min(mbmax, batch - done);
        Plan* pl; int rc = get_plan(h, mb, gh, gw, &pl, mode);
        if (rc) return rc;
        h->last_plan = pl;
        Ctx c;
        c.cache_in = cache;
        c.arena = h->arena; c.off = &pl->off; c.stream = stream;
        c.tickets = pl->tickets;
        pl->set_scratch(c, 0);
        c.grd = grd + (size_t)done * 3 * gh * gw;
        c.sat = sat ? sat + (size_t)done * 3 * CCVPE_SAT_HW * CCVPE_SAT_HW : nullptr;
        c.out.logits_flattened = out->logits_flattened + done * npx;
        c.out.heatmap = out->heatmap + done * npx;
        c.out.ori = out->ori + done * 2 * npx;
        for (int k = 0; k < 6; ++k) {
            const size_t hw = (size_t)(8 << k) * (8 << k);
            c.out.matching_score[k] = out->matching_score[k] + (size_t)done * h->rolls[k] * hw;
        }
        if (!profile && pl->use_graph && !h->debug) {
            // latency mode: stage inputs, replay the captured launch sequence, copy the outputs out
            const ccvpe_outputs user = c.out;
            const float* ugrd = c.grd; const float* usat = c.sat;
            c.grd = c.ptr(pl->io_grd); c.sat = c.ptr(pl->io_sat);
            c.out.logits_flattened = c.ptr(pl->io_logits); c.out.heatmap = c.ptr(pl->io_heat); c.out.ori = c.ptr(pl->io_ori);
            for (int k = 0; k < 6; ++k) c.out.matching_score[k] = c.ptr(pl->io_ms[k]);
            {   // both inputs in one launch (sizes are multiples of 4 floats: 3 x H x W with even H or W - else the runtime copies)
                const size_t ng = (size_t)mb * 3 * gh * gw, ns = (size_t)mb * 3 * CCVPE_SAT_HW * CCVPE_SAT_HW;
                if (ng % 4 == 0 && ((uintptr_t)ugrd % 16) == 0 && ((uintptr_t)usat % 16) == 0) {
                    MultiCopy mc{};
                    mc.src[0] = ugrd; mc.dst[0] = (float*)c.grd; mc.n[0] = ng;
                    mc.src[1] = usat; mc.dst[1] = (float*)c.sat; mc.n[1] = ns;
                    mc.count = 2;
                    launch_multi_copy(mc, stream);
                } else {
                    HIPCHK(hipMemcpyAsync((void*)c.grd, ugrd, ng * sizeof(float), hipMemcpyDeviceToDevice, stream));
                    HIPCHK(hipMemcpyAsync((void*)c.sat, usat, ns * sizeof(float), hipMemcpyDeviceToDevice, stream));
                }
            }
            if (!pl->exec && pl->runs >= 1) {   // first call ran eagerly (lazy kernel attributes are set): capture now
                // capture on a private stream (the caller's may be the legacy null stream, which cannot capture)
                hipGraph_t graph = nullptr;
                if (!h->capture_stream) HIPCHK(hipStreamCreateWithFlags(&h->capture_stream, hipStreamNonBlocking));
                HIPCHK(hipStreamBeginCapture(h->capture_stream, hipStreamCaptureModeThreadLocal));
                const int rrc = run_ops(h, *pl, c, h->capture_stream);
                hipError_t ce = hipStreamEndCapture(h->capture_stream, &graph);
                if (rrc) ce = hipErrorUnknown;
                if (ce == hipSuccess && graph) {
                    hipGraphExec_t ex = nullptr;
                    if (hipGraphInstantiate(&ex, graph, nullptr, nullptr, 0) == hipSuccess) pl->exec = ex;
                    (void)hipGraphDestroy(graph);
                }
                if (!pl->exec) { (void)hipGetLastError(); pl->use_graph = false; }   // fall back to eager launches for good
            }
            if (pl->exec) HIPCHK(hipGraphLaunch(pl->exec, stream));
            else if (int rrc = run_ops(h, *pl, c, stream)) return rrc;
            pl->runs++;
            {   // the nine outputs in one launch
                MultiCopy mc{};
                auto add = [&](float* dst, const float* src, size_t n) { mc.src[mc.count] = src; mc.dst[mc.count] = dst; mc.n[mc.count] = n; ++mc.count; };
                add(user.logits_flattened, c.out.logits_flattened, (size_t)mb * npx);
                add(user.heatmap, c.out.heatmap, (size_t)mb * npx);
                add(user.ori, c.out.ori, (size_t)mb * 2 * npx);
                bool aligned = true;
                for (int k = 0; k < 6; ++k) add(user.matching_score[k], c.out.matching_score[k], (size_t)mb * h->rolls[k] * ((size_t)(8 << k) * (8 << k)));
                for (int i = 0; i < mc.count; ++i) aligned = aligned && mc.n[i] % 4 == 0 && ((uintptr_t)mc.dst[i] % 16) == 0 && ((uintptr_t)mc.src[i] % 16) == 0;
                if (aligned) launch_multi_copy(mc, stream);
                else
                    for (int i = 0; i < mc.count; ++i) HIPCHK(hipMemcpyAsync(mc.dst[i], mc.src[i], mc.n[i] * sizeof(float), hipMemcpyDeviceToDevice, stream));
            }
        } else if (!profile) {
            if (int rrc = run_ops(h, *pl, c, stream)) return rrc;
        } else {
            hipEvent_t e0, e1;
            HIPCHK(hipEventCreate(&e0));
            HIPCHK(hipEventCreate(&e1));
            for (auto& op : pl->ops) {
                (void)conv_igemm_last_tile();
                HIPCHK(hipEventRecord(e0, stream));
                op.fn(c);
                HIPCHK(hipEventRecord(e1, stream));
                HIPCHK(hipEventSynchronize(e1));
                float ms = 0.f;
                HIPCHK(hipEventElapsedTime(&ms, e0, e1));
                const int tile = conv_igemm_last_tile();
                std::string nm = op.name;
                double issued = op.flops;   // launches that are not tiled GEMMs: issued == algorithmic
                if (tile) {
                    nm += std::string("|") + conv_igemm_tile_name(tile);
                    if ((tile >> 8) == 255) nm += "_tailsplit";
                    else if ((tile >> 8) > SPLIT_FUSED) nm += "_splitk" + std::to_string((tile >> 8) - SPLIT_FUSED) + "r";   // r: reduces itself
                    else if ((tile >> 8) > 1) nm += "_splitk" + std::to_string(tile >> 8);
                    // FLOPs the launch puts on the matrix pipe: M and N padded to the tile, K to the packed depth;
                    // Winograd F(2x2,3x3): 16 products per 2x2 output tile and channel pair; bf16x3: three MFMAs per product
                    ConvParams q{};
                    q.M = op.gemm_m; q.N = op.gemm_n;
                    const double util = conv_igemm_tile_util(q, tile & 0xff);
                    const double mn_pad = util > 0 ? (double)op.gemm_m * op.gemm_n / util : 0.0;
                    if (conv_igemm_tile_is_wino4(tile)) issued = 2.0 * mn_pad * 2.25 * ((op.conv_cin + 3) / 4 * 4);   // 36 products per 4x4 tile; k-steps of 4 channels, all-zero ones skipped
                    else if (conv_igemm_tile_is_wino(tile)) issued = 2.0 * mn_pad * 4.0 * op.conv_cin;
                    else issued = 2.0 * mn_pad * op.gemm_kpad * (conv_igemm_tile_is_bf16x3(tile) ? 3.0 : 1.0);
                }
                h->prof.push_back({nm, ms, op.flops, op.bytes, issued});
            }
            (void)hipEventDestroy(e0);
            (void)hipEventDestroy(e1);
        }
        if (c.conv_errors) return ccvpe_fail(CCVPE_EINVAL, "%d convolution launches were refused (unsupported geometry)", c.conv_errors);
        done += mb;
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return ccvpe_fail(CCVPE_EHIP, "kernel launch failed: %s", hipGetErrorString(e));
    return 0;
}

int ccvpe_forward(ccvpe_handle h, const float* grd, int32_t grd_h, int32_t grd_w, const float* sat, int32_t batch,
                  const ccvpe_outputs* out, void* stream) {
    return run_forward(h, grd, grd_h, grd_w, sat, batch, out, (hipStream_t)stream, false);
}

int ccvpe_profile_forward(ccvpe_handle h, const float* grd, int32_t grd_h, int32_t grd_w, const float* sat, int32_t batch,
                          const ccvpe_outputs* out, void* stream) {
    int rc = run_forward(h, grd, grd_h, grd_w, sat, batch, out, (hipStream_t)stream, true);
    if (rc) return rc;
    return (int)h->prof.size();
}

int ccvpe_profile_row(ccvpe_handle h, int32_t i, char* name_buf, size_t name_cap, float* ms, double* flops, double* bytes) {
    if (!h || i < 0 || i >= (int)h->prof.size()) return ccvpe_fail(CCVPE_EINVAL, "row out of range");
    const auto& r = h->prof[i];
    if (name_buf && name_cap) { std::strncpy(name_buf, r.name.c_str(), name_cap - 1); name_buf[name_cap - 1] = 0; }
    if (ms) *ms = r.ms;
    if (flops) *flops = r.flops;
    if (bytes) *bytes = r.bytes;
    return 0;
}

int ccvpe_profile_row_issued(ccvpe_handle h, int32_t i, double* issued_flops) {
    if (!h || i < 0 || i >= (int)h->prof.size() || !issued_flops) return ccvpe_fail(CCVPE_EINVAL, "row out of range");
    *issued_flops = h->prof[i].issued;
    return 0;
}

// scratch of the post-processing launch: grows with the largest batch seen (the only synchronising step, once per size)
static int ensure_post_scratch(ccvpe_handle_s* h, int batch) {
    if (batch <= h->post_batch) return 0;
    HIPCHK(hipDeviceSynchronize());   // a launch in flight may still use the old buffer
    if (h->post_scratch) HIPCHK(hipFree(h->post_scratch));
    h->post_scratch = nullptr; h->post_batch = 0;
    const int cap = std::max(batch, 32);
    void* d = nullptr;
    if (hipMalloc(&d, postprocess_scratch_bytes(cap)) != hipSuccess) return ccvpe_fail(CCVPE_ENOMEM, "post-processing scratch");
    HIPCHK(hipMemset(d, 0, postprocess_scratch_bytes(cap)));
    HIPCHK(hipDeviceSynchronize());
    h->post_scratch = d; h->post_batch = cap;
    return 0;
}

static int postprocess_any(ccvpe_handle h, const float* heatmap, const float* ori, int32_t batch, ccvpe_pose* poses, float* rows, void* stream) {
    if (!h || !heatmap || !ori || (!poses && !rows) || batch <= 0 || batch > PP_MAX_BATCH) return ccvpe_fail(CCVPE_EINVAL, "bad argument (batch 1 .. 4096)");
    HIPCHK(hipSetDevice(h->cfg.device));
    static_assert(sizeof(ccvpe_pose) == sizeof(PoseOut), "pose layout");
    if (int rc = ensure_post_scratch(h, batch)) return rc;
    launch_postprocess(heatmap, ori, batch, CCVPE_OUT_HW * CCVPE_OUT_HW, reinterpret_cast<PoseOut*>(poses), rows, h->post_scratch, (hipStream_t)stream);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return ccvpe_fail(CCVPE_EHIP, "postprocess launch failed: %s", hipGetErrorString(e));
    return 0;
}

int ccvpe_postprocess(ccvpe_handle h, const float* heatmap, const float* ori, int32_t batch, ccvpe_pose* poses, void* stream) {
    if (!poses) return ccvpe_fail(CCVPE_EINVAL, "bad argument");
    return postprocess_any(h, heatmap, ori, batch, poses, nullptr, stream);
}

int ccvpe_postprocess_rows(ccvpe_handle h, const float* heatmap, const float* ori, int32_t batch, float* rows, void* stream) {
    if (!rows) return ccvpe_fail(CCVPE_EINVAL, "bad argument");
    return postprocess_any(h, heatmap, ori, batch, nullptr, rows, stream);
}

int ccvpe_eval_metrics(ccvpe_handle h, const ccvpe_pose* poses, const float* heatmap, int32_t batch, const int32_t* gt_index,
                       const float* gt_cos_sin, const double* meter_per_pixel, const double* heading_deg, ccvpe_metrics* out, void* stream) {
    if (!h || !poses || !heatmap || !gt_index || !meter_per_pixel || !out || batch <= 0) return ccvpe_fail(CCVPE_EINVAL, "bad argument");
    HIPCHK(hipSetDevice(h->cfg.device));
    static_assert(sizeof(ccvpe_metrics) == sizeof(MetricsOut), "metrics layout");
    launch_metrics(reinterpret_cast<const PoseOut*>(poses), heatmap, batch, CCVPE_OUT_HW, CCVPE_OUT_HW * CCVPE_OUT_HW, gt_index, gt_cos_sin,
                   meter_per_pixel, heading_deg, reinterpret_cast<MetricsOut*>(out), (hipStream_t)stream);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return ccvpe_fail(CCVPE_EHIP, "metrics launch failed: %s", hipGetErrorString(e));
    return 0;
}

int ccvpe_set_streams(ccvpe_handle h, int32_t n_streams) {
    if (!h) return ccvpe_fail(CCVPE_EINVAL, "null handle");
    if (n_streams != 1 && n_streams != 2) return ccvpe_fail(CCVPE_EINVAL, "n_streams must be 1 or 2");
    const bool serial = n_streams == 1;
    if (serial != h->serial_issue)   // captured graphs embed the issue order: drop them
        for (auto& q : h->plans)
            if (q->exec) { (void)hipGraphExecDestroy(q->exec); q->exec = nullptr; q->runs = 0; }
    h->serial_issue = serial;
    return 0;
}

int ccvpe_set_debug(ccvpe_handle h, int32_t enable) {
    if (!h) return ccvpe_fail(CCVPE_EINVAL, "null handle");
    h->debug = enable != 0;
    return 0;
}

int ccvpe_read_tap(ccvpe_handle h, const char* name, float* host_dst, size_t capacity, size_t* n_out, int32_t shape_out[4]) {
    if (!h || !name || !host_dst) return ccvpe_fail(CCVPE_EINVAL, "null argument");
    HIPCHK(hipSetDevice(h->cfg.device));
    Plan* pl = nullptr;
    for (auto it = h->plans.rbegin(); it != h->plans.rend(); ++it)
        if ((*it)->debug) { pl = it->get(); break; }
    if (!pl) return ccvpe_fail(CCVPE_ESTATE, "no debug plan: call ccvpe_set_debug(h, 1) before forward");
    auto it = pl->taps.find(name);
    if (it == pl->taps.end()) return ccvpe_fail(CCVPE_EKEY, "unknown tap '%s'", name);
    const TapInfo& ti = it->second;
    const Tensor& t = ti.t;
    float* base = h->arena + pl->off[t.id];
    HIPCHK(hipDeviceSynchronize());
    if (ti.C < 0) {   // already NCHW
        const size_t n = (size_t)t.B * t.H * t.W * t.C;
        if (n > capacity) return ccvpe_fail(CCVPE_EINVAL, "tap needs %zu floats", n);
        HIPCHK(hipMemcpy(host_dst, base, n * sizeof(float), hipMemcpyDeviceToHost));
        if (n_out) *n_out = n;
        if (shape_out) { shape_out[0] = t.B; shape_out[1] = t.H; shape_out[2] = t.W; shape_out[3] = t.C; }
        return 0;
    }
    const int hw = t.H * t.W;
    const size_t n = (size_t)t.B * ti.C * hw;
    if (n > capacity) return ccvpe_fail(CCVPE_EINVAL, "tap needs %zu floats", n);
    float* tmp = nullptr;
    HIPCHK(hipMalloc((void**)&tmp, n * sizeof(float)));
    launch_nhwc_to_nchw(base, t.C, ti.coff, ti.C, t.B, hw, tmp, nullptr);
    hipError_t e = hipMemcpy(host_dst, tmp, n * sizeof(float), hipMemcpyDeviceToHost);
    (void)hipFree(tmp);
    if (e != hipSuccess) return ccvpe_fail(CCVPE_EHIP, "tap copy failed: %s", hipGetErrorString(e));
    if (n_out) *n_out = n;
    if (shape_out) { shape_out[0] = t.B; shape_out[1] = ti.C; shape_out[2] = t.H; shape_out[3] = t.W; }
    return 0;
}

size_t ccvpe_aerial_cache_bytes(ccvpe_handle h, int32_t batch) {
    if (!h || batch <= 0) { ccvpe_fail(CCVPE_EINVAL, "bad argument"); return 0; }
    size_t off[6];
    return cache_layout(h->vs, batch, off) * sizeof(float);
}

int ccvpe_encode_aerial(ccvpe_handle h, const float* sat, int32_t batch, void* cache, void* stream) {
    if (!h || !sat || !cache || batch <= 0) return ccvpe_fail(CCVPE_EINVAL, "bad argument");
    if (!h->finalized) return ccvpe_fail(CCVPE_ESTATE, "ccvpe_finalize_weights has not been called");
    if (batch > h->cfg.micro_batch) return ccvpe_fail(CCVPE_EINVAL, "aerial encode needs batch <= micro_batch (%d)", h->cfg.micro_batch);
    HIPCHK(hipSetDevice(h->cfg.device));
    Plan* pl; int rc = get_plan(h, batch, 0, 0, &pl, 1);
    if (rc) return rc;
    Ctx c;
    c.arena = h->arena; c.off = &pl->off; c.stream = (hipStream_t)stream;
    c.tickets = pl->tickets;
    c.splitk_scratch = c.ptr(pl->scratch); c.splitk_floats = Plan::SPLITK_FLOATS;
    c.sat = sat; c.cache_out = (float*)cache;
    for (auto& op : pl->ops) op.fn(c);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return ccvpe_fail(CCVPE_EHIP, "kernel launch failed: %s", hipGetErrorString(e));
    return 0;
}

int ccvpe_forward_cached(ccvpe_handle h, const float* grd, int32_t grd_h, int32_t grd_w, const void* cache, int32_t batch,
                         const ccvpe_outputs* out, void* stream) {
    if (!cache) return ccvpe_fail(CCVPE_EINVAL, "null cache");
    return run_forward(h, grd, grd_h, grd_w, nullptr, batch, out, (hipStream_t)stream, false, (const float*)cache);
}

int ccvpe_preprocess(const uint8_t* hwc, int32_t batch, int32_t H, int32_t W, const int32_t* shift, int32_t crop_w,
                     const float mean[3], const float stdv[3], float* out_nchw, void* stream) {
    if (!hwc || !out_nchw || !mean || !stdv) return ccvpe_fail(CCVPE_EINVAL, "null argument");
    if (batch <= 0 || H <= 0 || W <= 0 || crop_w <= 0 || crop_w > W) return ccvpe_fail(CCVPE_EINVAL, "bad geometry");
    PreprocParams p{};
    p.in = hwc; p.B = batch; p.H = H; p.W = W; p.crop_w = crop_w; p.shift = shift; p.out = out_nchw;
    for (int c = 0; c < 3; ++c) { p.mean[c] = mean[c]; p.stdv[c] = stdv[c]; }
    launch_preprocess(p, (hipStream_t)stream);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return ccvpe_fail(CCVPE_EHIP, "preprocess launch failed: %s", hipGetErrorString(e));
    return 0;
}


int ccvpe_preprocess_resize(const uint8_t* hwc, int32_t batch, int32_t in_h, int32_t in_w, int32_t out_h, int32_t out_w,
                            const int32_t* shift, int32_t crop_w, const float mean[3], const float stdv[3], uint8_t* scratch,
                            float* out_nchw, void* stream) {
    if (!hwc || !out_nchw || !mean || !stdv) return ccvpe_fail(CCVPE_EINVAL, "null argument");
    if (batch <= 0 || in_h <= 0 || in_w <= 0 || out_h <= 0 || out_w <= 0 || crop_w <= 0 || crop_w > out_w) return ccvpe_fail(CCVPE_EINVAL, "bad geometry");
    if (in_w != out_w && !scratch) return ccvpe_fail(CCVPE_EINVAL, "scratch of batch*in_h*out_w*3 bytes is required when the width changes");
    if ((double)batch * in_h * std::max(in_w, out_w) * 3 >= 2147483647.0 * 2) return ccvpe_fail(CCVPE_EINVAL, "image batch too large");
    ResizeParams p{};
    p.in = hwc; p.B = batch; p.IH = in_h; p.IW = in_w; p.OH = out_h; p.OW = out_w; p.crop_w = crop_w; p.tmp = scratch; p.shift = shift; p.out = out_nchw;
    for (int c = 0; c < 3; ++c) { p.mean[c] = mean[c]; p.stdv[c] = stdv[c]; }
    if (launch_resize(p, (hipStream_t)stream) != 0) return ccvpe_fail(CCVPE_EINVAL, "down-scaling factors above 8 are not supported (%dx%d -> %dx%d)", in_h, in_w, out_h, out_w);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return ccvpe_fail(CCVPE_EHIP, "resize launch failed: %s", hipGetErrorString(e));
    return 0;
}

int ccvpe_debug_dump_plan(ccvpe_handle h, const char* path) {
    if (!h || !path) return ccvpe_fail(CCVPE_EINVAL, "null argument");
    Plan* pl = h->last_plan;
    if (!pl) return ccvpe_fail(CCVPE_ESTATE, "no forward has run on this handle");
    HIPCHK(hipSetDevice(h->cfg.device));
    HIPCHK(hipDeviceSynchronize());
    const size_t n = pl->size.size();
    unsigned long long* d = nullptr;
    HIPCHK(hipMalloc((void**)&d, n * sizeof(unsigned long long)));
    HIPCHK(hipMemset(d, 0, n * sizeof(unsigned long long)));
    for (size_t id = 0; id < n; ++id) {
        if (pl->size[id] == 0) continue;
        const int blocks = (int)std::min<size_t>((pl->size[id] + 255) / 256, 2048);
        CCVPE_LAUNCH(checksum_kernel, dim3(blocks), dim3(256), 0, nullptr, reinterpret_cast<const uint32_t*>(h->arena + pl->off[id]), pl->size[id], d + id);
    }
    std::vector<unsigned long long> sums(n);
    hipError_t e = hipMemcpy(sums.data(), d, n * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return ccvpe_fail(CCVPE_EHIP, "checksum copy failed: %s", hipGetErrorString(e));
    FILE* f = std::fopen(path, "w");
    if (!f) return ccvpe_fail(CCVPE_EINVAL, "cannot open %s", path);
    std::fprintf(f, "# plan B=%d grd=%dx%d mode=%d two_streams=%d tensors=%zu arena_floats=%zu\n", pl->B, pl->gh, pl->gw, pl->mode, (int)pl->two_streams, n, pl->total);
    for (size_t i = 0; i < pl->ops.size(); ++i) {
        const Op& op = pl->ops[i];
        std::fprintf(f, "op %zu %s stream=%d wait=%d signal=%d tile=%s", i, op.name.c_str(), op.stream, op.wait_on.empty() ? -1 : op.wait_on[0], (int)op.signal,
                     op.tile ? conv_igemm_tile_name(*op.tile & 0xff) : "-");
        if (op.tile && (*op.tile >> 8) == 255) std::fprintf(f, "_tailsplit");
        else if (op.tile && (*op.tile >> 8) > SPLIT_FUSED) std::fprintf(f, "_splitk%dr", (*op.tile >> 8) - SPLIT_FUSED);
        else if (op.tile && (*op.tile >> 8) > 1) std::fprintf(f, "_splitk%d", *op.tile >> 8);
        for (int id : op.uses) std::fprintf(f, " t%d[off=%zu,n=%zu]=%016llx", id, pl->off[id], pl->size[id], sums[id]);
        std::fprintf(f, "\n");
    }
    if (h->snap[0]) {   // CCVPE_DIAG_SNAP: what the named launch's tensors held right before / right after it, in stream order
        for (int which = 0; which < 2; ++which) {
            std::fprintf(f, "snap_%s %s", which ? "after" : "before", h->diag_snap.c_str());
            for (auto& e : h->snap_layout) {
                unsigned long long* dd = nullptr;
                unsigned long long v = 0;
                if (hipMalloc((void**)&dd, sizeof(v)) == hipSuccess) {
                    (void)hipMemset(dd, 0, sizeof(v));
                    const int blocks = (int)std::min<size_t>((pl->size[e.first] + 255) / 256, 2048);
                    CCVPE_LAUNCH(checksum_kernel, dim3(blocks), dim3(256), 0, nullptr, reinterpret_cast<const uint32_t*>(h->snap[which] + e.second), pl->size[e.first], dd);
                    (void)hipMemcpy(&v, dd, sizeof(v), hipMemcpyDeviceToHost);
                    (void)hipFree(dd);
                }
                std::fprintf(f, " t%d=%016llx", e.first, v);
            }
            std::fprintf(f, "\n");
        }
    }
    std::fclose(f);
    return 0;
}

int ccvpe_op_num_tiles(void) { return conv_igemm_num_tiles(); }
const char* ccvpe_op_tile_name(int32_t tile) { return conv_igemm_tile_name(tile); }

int ccvpe_op_conv2d(const float* in, int32_t B, int32_t H, int32_t W, int32_t Cin, const float* w, const float* bias,
                    int32_t Cout, int32_t KH, int32_t KW, int32_t stride, int32_t pad, int32_t act, int32_t tile,
                    float* out, int32_t iters, float* ms, void* stream) {
    if (!in || !w || !out) return ccvpe_fail(CCVPE_EINVAL, "null argument");
    if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cin % 8 || Cout <= 0 || KH <= 0 || KW <= 0 || stride <= 0 || pad < 0)
        return ccvpe_fail(CCVPE_EINVAL, "bad conv geometry (Cin must be a multiple of 8)");
    const int OH = (H + 2 * pad - KH) / stride + 1, OW = (W + 2 * pad - KW) / stride + 1;
    if (OH <= 0 || OW <= 0) return ccvpe_fail(CCVPE_EINVAL, "empty output");
    if ((double)B * H * W * Cin >= 2147483647.0 || (double)B * OH * OW * Cout >= 2147483647.0)
        return ccvpe_fail(CCVPE_EINVAL, "tensor exceeds 2^31 elements");
    const size_t nw = (size_t)Cout * Cin * KH * KW;
    std::vector<float> hw(nw), hb(Cout, 0.f);
    HIPCHK(hipMemcpy(hw.data(), w, nw * sizeof(float), hipMemcpyDefault));
    if (bias) HIPCHK(hipMemcpy(hb.data(), bias, Cout * sizeof(float), hipMemcpyDefault));
    ccvpe_handle_s tmp;   // only its dev_allocs list / precision flag are used by the packer
    tmp.cfg.reserved[0] = 1;   // also pack the bf16x3 planes so every tile id can be exercised
    PackedConv pc;
    const int taps = KH * KW;
    int rc = pack_conv(&tmp, pc, Cout, taps, Cin, Cin, identity_map(Cin),
                       [&](int n, int t, int c) { return hw[((size_t)n * Cin + c) * taps + t]; }, hb, KH, KW);
    auto cleanup = [&]() { for (void* p : tmp.dev_allocs) (void)hipFree(p); };
    if (rc) { cleanup(); return rc; }
    ConvParams p = conv_params(pc, in, Cin, B, H, W, OH, OW, stride, pad, pad, act);
    p.dst[0] = {out, Cout, 0}; p.ndst = 1;
    hipStream_t st = (hipStream_t)stream;
    if (((tile >> 8) & 0xff) > 1) {   // tile word = id | (split-K << 8): give the launch a slab (and ticket counters: 255 and the codes above SPLIT_FUSED reduce themselves)
        int sk = (tile >> 8) & 0xff;
        if (sk > SPLIT_FUSED && sk != 255) sk -= SPLIT_FUSED;
        {
            void* d = nullptr;
            if (hipMalloc(&d, CONV_TICKETS * sizeof(unsigned)) != hipSuccess || hipMemset(d, 0, CONV_TICKETS * sizeof(unsigned)) != hipSuccess) { cleanup(); return ccvpe_fail(CCVPE_ENOMEM, "ticket counters"); }
            tmp.dev_allocs.push_back(d);
            tmp.dev_alloc_bytes.push_back(CONV_TICKETS * sizeof(unsigned));
            p.tickets = (unsigned*)d;
        }
        const size_t fl = (size_t)(sk == 255 ? 8 : sk) * p.M * p.N;   // 255 = F(4x4) tail split: its slab is a fraction of 8 full ones
        void* d = nullptr;
        if (hipMalloc(&d, fl * sizeof(float)) != hipSuccess) { cleanup(); return ccvpe_fail(CCVPE_ENOMEM, "split-K slab"); }
        tmp.dev_allocs.push_back(d);
        tmp.dev_alloc_bytes.push_back(fl * sizeof(float));
        p.partial = (float*)d; p.partial_floats = fl;
    }
    if (conv_igemm_tile_is_wino4p(tile) && H % 16 == 0 && W % 16 == 0) {   // split Winograd form: scratch for V = B^T d B
        const size_t fl = (size_t)B * (H / 16) * (W / 16) * ((Cin + 15) / 16) * 9216;
        void* d = nullptr;
        if (hipMalloc(&d, fl * sizeof(float)) != hipSuccess) { cleanup(); return ccvpe_fail(CCVPE_ENOMEM, "Winograd V scratch"); }
        tmp.dev_allocs.push_back(d);
        tmp.dev_alloc_bytes.push_back(fl * sizeof(float));
        p.wino4_v = (float*)d; p.wino4_v_floats = fl;
    }
    if (conv_igemm_tile_is_wino(tile) && !conv_wino_tile_supported(p, tile)) { cleanup(); return ccvpe_fail(CCVPE_EINVAL, "layer is not Winograd-shaped (3x3, stride 1, pad 1, W %% 16 == 0, H %% 16 == 0, output channels a multiple of 4; F(4x4): >= 40 of them)"); }
    if (launch_conv_igemm(p, tile, st) != 0) { cleanup(); return ccvpe_fail(CCVPE_EINVAL, "unsupported conv geometry (KH*KW <= 16, Cin %% 8 == 0)"); }
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && iters > 0 && ms) {
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0, st);
        for (int i = 0; i < iters; ++i) launch_conv_igemm(p, tile, st);
        (void)hipEventRecord(e1, st);
        (void)hipEventSynchronize(e1);
        float t = 0.f;
        (void)hipEventElapsedTime(&t, e0, e1);
        *ms = t / iters;
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    }
    hipError_t e2 = hipStreamSynchronize(st);
    cleanup();
    if (e != hipSuccess) return ccvpe_fail(CCVPE_EHIP, "conv launch failed: %s", hipGetErrorString(e));
    if (e2 != hipSuccess) return ccvpe_fail(CCVPE_EHIP, "conv execution failed: %s", hipGetErrorString(e2));
    return 0;
}

}  // extern "C"

