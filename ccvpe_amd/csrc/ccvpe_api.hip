// C ABI of libccvpe_hip.so: handle, state_dict ingestion (BN folding + weight packing), execution
// plan with lifetime-based workspace reuse, forward orchestration.  See include/ccvpe.h.
//
// The orchestration restates CVM_*.forward (reference models.py:150-343, 448-652, 752-950, 1051-1244)
// as a static list of kernel launches over NHWC tensors; concatenations are channel-offset writes
// into pre-allocated buffers, the encoder taps are written by the producing GEMM's epilogue.
#include "../../include/ccvpe.h"
#include "kernels.h"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <set>
#include <string>
#include <type_traits>
#include <vector>

using namespace ccvpe;

// ------------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------------
static thread_local std::string g_err;
static int fail(int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
#define HIPCHK(expr)                                                                           \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess) return fail(CCVPE_EHIP, "%s failed: %s", #expr, hipGetErrorString(_e)); \
    } while (0)

// ------------------------------------------------------------------------------------------------
// static description of the network (mirrors ccvpe_amd/spec.py; reference lines cited there)
// ------------------------------------------------------------------------------------------------
struct BlockSpec { int e, k, s, cin, cout; };
static const BlockSpec B0[16] = {
    {1, 3, 1, 32, 16}, {6, 3, 2, 16, 24}, {6, 3, 1, 24, 24}, {6, 5, 2, 24, 40}, {6, 5, 1, 40, 40},
    {6, 3, 2, 40, 80}, {6, 3, 1, 80, 80}, {6, 3, 1, 80, 80}, {6, 5, 1, 80, 112}, {6, 5, 1, 112, 112},
    {6, 5, 1, 112, 112}, {6, 5, 2, 112, 192}, {6, 5, 1, 192, 192}, {6, 5, 1, 192, 192}, {6, 5, 1, 192, 192},
    {6, 3, 1, 192, 320},
};
static const int TAP_BLOCK[5] = {15, 10, 4, 2, 0};   // skip of decoder level 6..2 (models.py:465-469)
static const float BN_EPS = 1e-3f;                   // utils.py:666

struct DecLevel { int din, dout, skip, mid, out; };
struct VariantSpec {
    int feat_h;
    int head_ch[6];
    int sat_desc;
    int match_ch[6];
    int step[6];
    int n_rolls;
    int centre;
    DecLevel loc[6], ori[6];
};
static const DecLevel VIGOR_LOC[6] = {{1281, 1024, 320, 640, 640}, {641, 320, 112, 320, 320}, {321, 160, 40, 160, 160},
                                      {161, 80, 24, 80, 80},       {81, 40, 16, 40, 40},      {41, 16, 0, 16, 1}};
static const DecLevel VIGOR_ORI[6] = {{1300, 1024, 320, 640, 640}, {640, 256, 112, 256, 256}, {256, 128, 40, 128, 128},
                                      {128, 64, 24, 64, 64},       {64, 32, 16, 32, 32},      {32, 16, 0, 16, 2}};
static const DecLevel KITTI_LOC[6] = {{2049, 1024, 320, 512, 512}, {513, 256, 112, 256, 256}, {257, 128, 40, 128, 128},
                                      {129, 64, 24, 128, 128},     {129, 32, 16, 32, 32},     {33, 16, 0, 16, 1}};
static const DecLevel KITTI_ORI[6] = {{2064, 1024, 320, 512, 512}, {512, 256, 112, 256, 256}, {256, 128, 40, 128, 128},
                                      {128, 64, 24, 64, 64},       {64, 32, 16, 32, 32},      {32, 16, 0, 16, 2}};

static VariantSpec make_variant(int v) {
    VariantSpec s{};
    auto cp = [](DecLevel* d, const DecLevel* src) { for (int i = 0; i < 6; ++i) d[i] = src[i]; };
    if (v == CCVPE_VARIANT_KITTI) {
        s.feat_h = 8;
        int hc[6] = {16, 8, 4, 2, 1, 1}, mc[6] = {2048, 512, 256, 128, 128, 32}, st[6] = {128, 64, 32, 16, 8, 8};
        for (int i = 0; i < 6; ++i) { s.head_ch[i] = hc[i]; s.match_ch[i] = mc[i]; s.step[i] = st[i]; }
        s.sat_desc = 2048; s.n_rolls = 16; s.centre = 0;
        cp(s.loc, KITTI_LOC); cp(s.ori, KITTI_ORI);
    } else {
        int mc[6] = {1280, 640, 320, 160, 80, 40}, st[6] = {64, 32, 16, 8, 4, 2};
        int hv[6] = {64, 32, 16, 8, 4, 2}, ho[6] = {32, 16, 8, 4, 2, 1};
        for (int i = 0; i < 6; ++i) {
            s.match_ch[i] = mc[i]; s.step[i] = st[i];
            s.head_ch[i] = (v == CCVPE_VARIANT_OXFORD) ? ho[i] : hv[i];
        }
        s.feat_h = (v == CCVPE_VARIANT_OXFORD) ? 4 : 10;
        s.sat_desc = 1280; s.n_rolls = 20; s.centre = (v == CCVPE_VARIANT_OXFORD);
        cp(s.loc, VIGOR_LOC); cp(s.ori, VIGOR_ORI);
    }
    return s;
}

static void static_pad(int k, int s, int& lo, int& hi) {   // utils.py:261-277 with the nominal-224 walk
    if (s == 1) { lo = hi = (k - 1) / 2; return; }
    int total = k - 2;
    lo = total / 2; hi = total - lo;
}
static int conv_out(int n, int k, int s) {
    int lo, hi; static_pad(k, s, lo, hi);
    return (n + lo + hi - k) / s + 1;
}
static int se_squeeze(int cin) { return std::max(1, (int)(cin * 0.25)); }   // model.py:79
static int round_up(int a, int b) { return (a + b - 1) / b * b; }

// ------------------------------------------------------------------------------------------------
// weights
// ------------------------------------------------------------------------------------------------
struct PackedConv {
    float* w = nullptr;
    unsigned short *w_hi = nullptr, *w_lo = nullptr;   // bf16x3 planes (precision mode 1 only)
    float* bias = nullptr;
    int N = 0, Kpad = 0, nchunks = 0, cinp = 0, KH = 1, KW = 1;
    float* wino = nullptr;        // Winograd-domain weights of a 3x3 layer (kernels_wino.hip), fp32 mode only
    int wino_n16 = 0;
    size_t wino_bytes = 0;
    float* wino4 = nullptr;       // F(4x4,3x3) weights (kernels_wino4.hip): wide layers on maps up to 128 x 128 only
    size_t wino4_bytes = 0;
};
struct BlockW {
    PackedConv expand, project;
    float* exp_lin = nullptr;     // [mid][cinp16] expand weights for the fused expand+depthwise kernel
    int exp_cinp = 0;
    float *dw_w = nullptr, *dw_b = nullptr, *se_w1 = nullptr, *se_b1 = nullptr, *se_w2 = nullptr, *se_b2 = nullptr;
    int sq = 0;
};
struct EncoderW {
    float *stem_w = nullptr, *stem_b = nullptr;
    BlockW blk[16];
    PackedConv head;
};
struct DecoderW {
    PackedConv deconv[6], conva[6], convb[5];
    float* tail_w = nullptr;
    float* l1_wt = nullptr;       // the same weights as [9][cout][16] for the fused level-1 kernel (channel pairs contiguous)
    float tail_b[2] = {0.f, 0.f};
    // fused last level (kernels_level1.hip)
    float *l1_wd = nullptr, *l1_bd = nullptr, *l1_wa = nullptr, *l1_ba = nullptr;
    int l1_cx = 0, l1_cxp = 0;
};

struct Tensor {
    int id = -1; int B = 0, H = 0, W = 0, C = 0;
    bool split = false;   // bf16x3 mode: stored as two bf16 planes (hi | lo) instead of fp32
    long long numel() const { return (long long)B * H * W * C; }
};

struct Ctx {
    float* arena = nullptr;
    const std::vector<size_t>* off = nullptr;
    hipStream_t stream = nullptr;
    const float* grd = nullptr;
    const float* sat = nullptr;
    ccvpe_outputs out{};
    float* splitk_scratch = nullptr;
    size_t splitk_floats = 0;
    const float* cache_in = nullptr;   // aerial cache consumed by a "cached" plan
    float* cache_out = nullptr;        // aerial cache produced by an "encode" plan
    float* ptr(const Tensor& t) const { return arena + (*off)[t.id]; }
    Dst dst(const Tensor& t, int coff = 0) const { return Dst{ptr(t), t.C, coff, t.split ? 1 : 0, t.numel()}; }
    mutable int conv_errors = 0;   // launches refused by launch_conv_igemm (unsupported geometry)
    void launch_conv(ConvParams& p, int cfg) const {
        p.partial = splitk_scratch;
        p.partial_floats = splitk_floats;
        if (launch_conv_igemm(p, cfg, stream) != 0) ++conv_errors;
    }
};

struct Op {
    std::string name;
    std::function<void(const Ctx&)> fn;
    std::vector<int> uses;
    double flops = 0, bytes = 0;
    // implicit-GEMM launches: tile id the launch uses (0 = heuristic) - set by Plan::autotune
    std::shared_ptr<int> tile;
    int gemm_m = 0, gemm_n = 0, gemm_kpad = 0;
    int conv_cin = 0;             // 3x3 layers: input channels (issued-FLOP accounting of the Winograd tiles)
    bool bf16x3_only = false;     // the launch reads a pre-split bf16 tensor: exact-fp32 tiles cannot serve it
    bool wino_ok = false;         // 3x3 / stride 1 layer with Winograd-domain weights packed
    bool wino4_ok = false;        // ... with the F(4x4,3x3) weights packed as well
    bool is_pw = false;           // 1x1 conv / k2s2 transposed conv: the pointwise persistent tiles may serve it
    // two-stream execution (Plan::schedule): stream the op is issued on, ops of the other stream it must wait for,
    // and whether an op of the other stream waits for this one (then an event is recorded after it)
    int stream = 0;
    std::vector<int> wait_on;
    bool signal = false;
};

struct TapInfo { Tensor t; int coff; int C; };

struct Plan {
    int B = 0, gh = 0, gw = 0;
    int mode = 0;                 // 0 full forward, 1 aerial encode only, 2 forward from a cached aerial encoding
    bool debug = false;
    std::vector<size_t> size;     // floats per tensor
    std::vector<size_t> off;      // float offset in the arena
    std::vector<Op> ops;
    std::map<std::string, TapInfo> taps;
    size_t total = 0;             // floats
    float* arena = nullptr;
    Tensor scratch;               // split-K slab scratch shared by every conv launch (whole-plan lifetime)
    // hipGraph replay (latency mode): static staging copies of the inputs / outputs so the captured kernel
    // arguments never change; the caller's buffers are reached by D2D copies outside the graph
    bool use_graph = false;
    Tensor io_grd, io_sat, io_logits, io_heat, io_ori, io_ms[6];
    Tensor tune_cache;            // encode plans: stand-in for the caller's cache while the plan is being autotuned
    hipGraphExec_t exec = nullptr;
    int runs = 0;
    // Two-stream execution: the aerial encoder and the orientation decoder are issued on a second stream, so the
    // ramp-up / drain of the ~330 short kernels of one chain is filled by the other chain.  Dependencies come from
    // the ops' tensor lists (any two ops that touch the same tensor stay ordered), and a two-stream plan gives every
    // tensor its own memory (lifetime-based reuse would add hidden dependencies between the streams).
    bool two_streams = false;
    Tensor scratch2;              // split-K slab scratch of the second stream
    std::vector<hipEvent_t> events;   // one per signalling op + fork + join, created on first use
    ~Plan() {
        if (exec) (void)hipGraphExecDestroy(exec);
        for (hipEvent_t e : events) if (e) (void)hipEventDestroy(e);
    }
    void schedule() {
        if (!two_streams) return;
        bool any = false;
        for (auto& o : ops) {
            o.stream = (o.name.rfind("sat.", 0) == 0 || o.name.rfind("ori", 0) == 0) ? 1 : 0;
            any = any || o.stream == 1;
            o.wait_on.clear();
            o.signal = false;
        }
        if (!any) { two_streams = false; return; }
        std::map<int, int> last_use;   // tensor id -> most recent op that touches it
        for (int i = 0; i < (int)ops.size(); ++i) {
            int dep = -1;              // stream order already covers earlier ops of a stream: the latest one is enough
            for (int id : ops[i].uses) {
                auto it = last_use.find(id);
                if (it != last_use.end() && ops[it->second].stream != ops[i].stream) dep = std::max(dep, it->second);
            }
            if (dep >= 0) { ops[i].wait_on.push_back(dep); ops[dep].signal = true; }
            for (int id : ops[i].uses) last_use[id] = i;
        }
        if (getenv("CCVPE_LOG_SCHEDULE"))
            for (int i = 0; i < (int)ops.size(); ++i) {
                std::fprintf(stderr, "op %3d s%d %-28s wait=%d uses=", i, ops[i].stream, ops[i].name.c_str(), ops[i].wait_on.empty() ? -1 : ops[i].wait_on[0]);
                for (int id : ops[i].uses) std::fprintf(stderr, "%d ", id);
                std::fprintf(stderr, "\n");
            }
    }
    static constexpr size_t SPLITK_FLOATS = 32u << 20;   // 128 MiB: 16 slabs of M*N <= 2M outputs

    // every kernel addresses a tensor with 32-bit byte offsets (raw buffer loads, `unsigned in_bytes`, the 0x80000000
    // out-of-range sentinel): no tensor of a plan may reach 2 GiB - checked at the end of build_plan
    size_t max_tensor_bytes = 0;
    int max_dims[4] = {0, 0, 0, 0};
    Tensor alloc(int B_, int H, int W, int C) {
        Tensor t; t.id = (int)size.size(); t.B = B_; t.H = H; t.W = W; t.C = C;
        size.push_back((size_t)B_ * H * W * C);
        if (size.back() * sizeof(float) > max_tensor_bytes) {
            max_tensor_bytes = size.back() * sizeof(float);
            max_dims[0] = B_; max_dims[1] = H; max_dims[2] = W; max_dims[3] = C;
        }
        return t;
    }
    void add(const std::string& name, std::vector<Tensor> uses, std::function<void(const Ctx&)> fn, double flops = 0, double bytes = 0) {
        Op o; o.name = name; o.fn = std::move(fn); o.flops = flops; o.bytes = bytes;
        for (auto& t : uses) o.uses.push_back(t.id);
        ops.push_back(std::move(o));
    }
    void add_conv(const std::string& name, std::vector<Tensor> uses, int gemm_m, int gemm_n, int gemm_kpad,
                  std::function<void(const Ctx&, int)> fn, double flops, double bytes) {
        auto tp = std::make_shared<int>(TILE_AUTO);
        add(name, std::move(uses), [fn, tp](const Ctx& c) { fn(c, *tp); }, flops, bytes);
        ops.back().tile = tp;
        ops.back().gemm_m = gemm_m;
        ops.back().gemm_n = gemm_n;
        ops.back().gemm_kpad = gemm_kpad;
    }
    // Workspace layout.  Single-stream plans: first-fit with lifetime reuse over the program order.  Two-stream plans:
    // memory may only be recycled between tensors whose launches are ordered whichever way the plan is issued - i.e.
    // tensors touched by ONE stream only, recycled among tensors of the same stream (stream order == program order).
    // Tensors that both streams touch (the concat buffers the aerial encoder's taps land in, the descriptor map, the
    // level-1 score stack) keep private memory for the whole plan, so a recycled address never adds a dependency the
    // event edges do not know about.  Three regions: [stream-0 pool | stream-1 pool | cross-stream and pinned tensors].
    void assign() {
        const int n = (int)size.size();
        std::vector<int> first(n, 1 << 30), last(n, -1), smask(n, 0);
        for (int i = 0; i < (int)ops.size(); ++i)
            for (int id : ops[i].uses) {
                first[id] = std::min(first[id], i); last[id] = std::max(last[id], i);
                smask[id] |= 1 << (two_streams ? ops[i].stream : 0);
            }
        static const bool no_reuse = getenv("CCVPE_NO_REUSE") != nullptr;   // diagnostic: every tensor keeps its memory
        std::vector<bool> pinned(n, false);
        auto pin = [&](const Tensor& t) { if (t.id >= 0) { pinned[t.id] = true; first[t.id] = 0; last[t.id] = 1 << 30; } };
        if (debug || no_reuse) for (int i = 0; i < n; ++i) if (last[i] >= 0) pinned[i] = true;
        pin(scratch); pin(scratch2); pin(tune_cache);
        if (use_graph)
            for (const Tensor* t : {&io_grd, &io_sat, &io_logits, &io_heat, &io_ori, &io_ms[0], &io_ms[1], &io_ms[2], &io_ms[3], &io_ms[4], &io_ms[5]}) pin(*t);
        for (int i = 0; i < n; ++i) if (smask[i] == 3) pinned[i] = true;
        off.assign(n, 0);
        std::vector<int> order(n);
        for (int i = 0; i < n; ++i) order[i] = i;
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return first[a] < first[b]; });
        auto granule = [&](int id) { return (size[id] + 63) & ~(size_t)63; };   // 256-byte granules
        total = 0;
        for (int pool = 1; pool <= 2; ++pool) {   // recycled regions of stream 0 and stream 1
            const size_t base = total;
            std::vector<int> placed;
            for (int id : order) {
                if (last[id] < 0 || pinned[id] || smask[id] != pool) continue;
                const size_t sz = granule(id);
                // candidate offsets: the region start and the end of every live, lifetime-overlapping tensor
                std::vector<std::pair<size_t, size_t>> busy;
                for (int o : placed)
                    if (!(last[o] < first[id] || last[id] < first[o])) busy.push_back({off[o], off[o] + granule(o)});
                std::sort(busy.begin(), busy.end());
                size_t pos = base;
                for (auto& iv : busy) {
                    if (pos + sz <= iv.first) break;
                    pos = std::max(pos, iv.second);
                }
                off[id] = pos;
                total = std::max(total, pos + sz);
                placed.push_back(id);
            }
        }
        for (int id : order) {   // private memory
            if (last[id] < 0 || !pinned[id]) continue;
            off[id] = total;
            total += granule(id);
        }
    }
};

struct ccvpe_handle_s {
    ccvpe_config cfg{};
    VariantSpec vs{};
    int rolls[6] = {0};                               // R_k of the ms outputs
    std::map<std::string, std::vector<int64_t>> expect;   // key -> shape
    std::map<std::string, std::vector<float>> host;       // raw host copies until finalize
    std::set<std::string> skipped;
    bool finalized = false;
    bool debug = false;
    bool autotune = true;
    int fuse_mbconv = 1;          // CCVPE_FUSE_MBCONV: 0 never, 1 where measured profitable (3x3 blocks), 2 every supported block
    bool fuse_level1 = true;      // CCVPE_FUSE_L1=0 falls back to deconv / conv / tail launches
    // CCVPE_WINOGRAD=0 keeps the decoder 3x3 layers on the implicit GEMM.  fp32 plans only: conv_wino_kernel's column pass
    // is a v_pk_add_f32 with op_sel:[0,1], which gfx950 mis-executes beside another wave's bf16 MFMAs, so a
    // bf16x3 plan never contains it
    bool wino = true;
    int graph_mode = -1;          // -1 auto (plans of <= 4 samples replay a hipGraph), 0 never, 1 always
    hipStream_t capture_stream = nullptr;
    hipStream_t aux_stream = nullptr;   // second stream of two-stream plans
    bool two_streams = true;      // CCVPE_STREAMS=1 issues everything on the caller's stream
    bool serial_issue = false;    // ccvpe_set_streams(h, 1): run two-stream plans in program order on one stream
    std::vector<void*> dev_allocs;
    std::vector<size_t> dev_alloc_bytes;   // parallel to dev_allocs (packed-weight cache: ccvpe_save_packed)
    EncoderW grd_enc, sat_enc;
    PackedConv grd_heads, sat_desc;
    float* grd_wh[6] = {nullptr};
    float grd_b2[6] = {0};
    DecoderW loc, ori;
    std::vector<std::unique_ptr<Plan>> plans;
    Plan* last_plan = nullptr;    // plan of the most recent forward (ccvpe_debug_dump_plan)
    // diagnostics (environment, read at ccvpe_create): CCVPE_DIAG_SYNC_BEFORE=<name part> drains the device before matching
    // launches; CCVPE_DIAG_SNAP=<launch name> copies that launch's tensors aside (stream ordered) right before and right after it
    std::string diag_sync, diag_snap;
    float* snap[2] = {nullptr, nullptr};
    size_t snap_floats = 0;
    std::vector<std::pair<int, size_t>> snap_layout;   // (tensor id, float offset inside a snapshot buffer)
    std::map<std::pair<int, int>, int> mb_cap;   // ground size -> ccvpe_max_micro_batch (2 GiB tensor bound)
    float* arena = nullptr;
    size_t arena_floats = 0;
    // profiling rows of the last ccvpe_profile_forward
    struct Row { std::string name; float ms; double flops, bytes, issued; };
    std::vector<Row> prof;
};

// ---- expected state_dict layout -----------------------------------------------------------------
static void add_bn(std::map<std::string, std::vector<int64_t>>& m, const std::string& p, int c) {
    m[p + ".weight"] = {c}; m[p + ".bias"] = {c}; m[p + ".running_mean"] = {c}; m[p + ".running_var"] = {c};
    m[p + ".num_batches_tracked"] = {};
}
static void add_encoder(std::map<std::string, std::vector<int64_t>>& m, const std::string& p) {
    m[p + "._conv_stem.weight"] = {32, 3, 3, 3};
    add_bn(m, p + "._bn0", 32);
    for (int i = 0; i < 16; ++i) {
        const BlockSpec& b = B0[i];
        std::string q = p + "._blocks." + std::to_string(i);
        int mid = b.cin * b.e;
        if (b.e != 1) { m[q + "._expand_conv.weight"] = {mid, b.cin, 1, 1}; add_bn(m, q + "._bn0", mid); }
        m[q + "._depthwise_conv.weight"] = {mid, 1, b.k, b.k};
        add_bn(m, q + "._bn1", mid);
        int sq = se_squeeze(b.cin);
        m[q + "._se_reduce.weight"] = {sq, mid, 1, 1}; m[q + "._se_reduce.bias"] = {sq};
        m[q + "._se_expand.weight"] = {mid, sq, 1, 1}; m[q + "._se_expand.bias"] = {mid};
        m[q + "._project_conv.weight"] = {b.cout, mid, 1, 1};
        add_bn(m, q + "._bn2", b.cout);
    }
    m[p + "._conv_head.weight"] = {1280, 320, 1, 1};
    add_bn(m, p + "._bn1", 1280);
    m[p + "._fc.weight"] = {1000, 1280};
    m[p + "._fc.bias"] = {1000};
}
static void build_expect(ccvpe_handle_s* h) {
    auto& m = h->expect;
    add_encoder(m, "grd_efficientnet");
    add_encoder(m, "sat_efficientnet");
    for (int k = 0; k < 6; ++k) {
        std::string p = "grd_feature_to_descriptor" + std::to_string(k + 1);
        m[p + ".0.weight"] = {h->vs.head_ch[k], 1280, 1, 1}; m[p + ".0.bias"] = {h->vs.head_ch[k]};
        m[p + ".2.weight"] = {1, h->vs.feat_h, 1, 1};        m[p + ".2.bias"] = {1};
    }
    m["sat_feature_to_descriptors.1.weight"] = {h->vs.sat_desc, 5120};
    m["sat_feature_to_descriptors.1.bias"] = {h->vs.sat_desc};
    for (int d = 0; d < 2; ++d) {
        const DecLevel* lv = d ? h->vs.ori : h->vs.loc;
        std::string sfx = d ? "_ori" : "";
        for (int j = 0; j < 6; ++j) {
            std::string n = std::to_string(6 - j);
            m["deconv" + n + sfx + ".weight"] = {lv[j].din, lv[j].dout, 2, 2};
            m["deconv" + n + sfx + ".bias"] = {lv[j].dout};
            m["conv" + n + sfx + ".0.weight"] = {lv[j].mid, lv[j].dout + lv[j].skip, 3, 3};
            m["conv" + n + sfx + ".0.bias"] = {lv[j].mid};
            m["conv" + n + sfx + ".2.weight"] = {lv[j].out, lv[j].mid, 3, 3};
            m["conv" + n + sfx + ".2.bias"] = {lv[j].out};
        }
    }
}

// ---- upload helpers ------------------------------------------------------------------------------
static int upload(ccvpe_handle_s* h, const std::vector<float>& v, float** out) {
    void* d = nullptr;
    size_t bytes = std::max<size_t>(v.size(), 4) * sizeof(float);
    HIPCHK(hipMalloc(&d, bytes));
    h->dev_allocs.push_back(d);
    h->dev_alloc_bytes.push_back(bytes);
    HIPCHK(hipMemcpy(d, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice));
    *out = (float*)d;
    return 0;
}

// Generic packer: rows n < N, k = tap*cinp + cmap(c).  `get(n, tap, c)` returns the (already scaled) weight.
static int pack_conv(ccvpe_handle_s* h, PackedConv& pc, int N, int taps, int cin, int cinp, const std::vector<int>& cmap,
                     const std::function<float(int, int, int)>& get, const std::vector<float>& bias, int KH, int KW) {
    const int npad = round_up(N, conv_igemm_npad());
    const int K = taps * cinp;
    const int kpad = round_up(K, 32);
    std::vector<float> w((size_t)npad * kpad, 0.f);
    for (int n = 0; n < N; ++n)
        for (int t = 0; t < taps; ++t)
            for (int c = 0; c < cin; ++c) w[(size_t)n * kpad + conv_igemm_k_index(cinp, taps, t, cmap[c])] = get(n, t, c);
    pc.N = N; pc.Kpad = kpad; pc.nchunks = K / 8; pc.cinp = cinp; pc.KH = KH; pc.KW = KW;
    int rc = upload(h, w, &pc.w);
    if (rc) return rc;
    if (KH == 3 && KW == 3 && cin == cinp && cin % 8 == 0 && (size_t)(cin / 8) * 16 * ((N + 15) / 16) * 512 < (1u << 31)) {
        std::vector<float> u;
        conv_wino_pack(N, cin, get, u, &pc.wino_n16);
        pc.wino_bytes = u.size() * sizeof(float);
        if ((rc = upload(h, u, &pc.wino))) return rc;
        // F(4x4,3x3) pays where the layer fills at least three of a workgroup's four 16-channel slices: measured faster than
        // every F(2x2) tile down to 40 output channels (conv2: 0.52 vs 0.64 ms), slower at 32 (conv2_ori: 0.41 vs 0.37);
        // 4x the direct weights, ~0.5 GB for the layers that qualify
        static const int wino4_min_n = getenv("CCVPE_WINO4_MIN_N") ? std::atoi(getenv("CCVPE_WINO4_MIN_N")) : 40;
        if (N >= wino4_min_n && (size_t)((cin + 15) / 16) * 4 * 9 * ((N + 15) / 16) * 1024 < (1u << 31) && !getenv("CCVPE_NO_WINO4")) {
            std::vector<float> u4;
            conv_wino4_pack(N, cin, get, u4);
            pc.wino4_bytes = u4.size() * sizeof(float);
            if ((rc = upload(h, u4, &pc.wino4))) return rc;
        }
    }
    if (h->cfg.reserved[0] == 1) {   // bf16x3: hi = bf16(w), lo = bf16(w - hi), round to nearest even
        auto to_bf16 = [](float f) -> unsigned short {
            uint32_t u; std::memcpy(&u, &f, 4);
            if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);   // NaN stays NaN
            return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
        };
        std::vector<unsigned short> hi(w.size()), lo(w.size());
        for (size_t i = 0; i < w.size(); ++i) {
            hi[i] = to_bf16(w[i]);
            uint32_t hu = (uint32_t)hi[i] << 16; float hf; std::memcpy(&hf, &hu, 4);
            lo[i] = to_bf16(w[i] - hf);
        }
        for (int plane = 0; plane < 2; ++plane) {
            void* d = nullptr;
            HIPCHK(hipMalloc(&d, hi.size() * sizeof(unsigned short)));
            h->dev_allocs.push_back(d);
            h->dev_alloc_bytes.push_back(hi.size() * sizeof(unsigned short));
            HIPCHK(hipMemcpy(d, plane ? lo.data() : hi.data(), hi.size() * sizeof(unsigned short), hipMemcpyHostToDevice));
            (plane ? pc.w_lo : pc.w_hi) = (unsigned short*)d;
        }
    }
    return upload(h, bias, &pc.bias);
}
static std::vector<int> identity_map(int n) { std::vector<int> m(n); for (int i = 0; i < n; ++i) m[i] = i; return m; }

struct BnFold { std::vector<float> scale, shift; };
static BnFold fold_bn(ccvpe_handle_s* h, const std::string& p) {
    const auto& g = h->host[p + ".weight"]; const auto& b = h->host[p + ".bias"];
    const auto& mu = h->host[p + ".running_mean"]; const auto& var = h->host[p + ".running_var"];
    BnFold f; f.scale.resize(g.size()); f.shift.resize(g.size());
    for (size_t i = 0; i < g.size(); ++i) {
        float s = g[i] / std::sqrt(var[i] + BN_EPS);
        f.scale[i] = s; f.shift[i] = b[i] - mu[i] * s;
    }
    return f;
}

static int pack_pointwise_bn(ccvpe_handle_s* h, PackedConv& pc, const std::string& wkey, const std::string& bnkey, int cout, int cin) {
    const auto& w = h->host[wkey];
    BnFold f = fold_bn(h, bnkey);
    return pack_conv(h, pc, cout, 1, cin, cin, identity_map(cin),
                     [&](int n, int, int c) { return w[(size_t)n * cin + c] * f.scale[n]; }, f.shift, 1, 1);
}

static int build_encoder(ccvpe_handle_s* h, EncoderW& e, const std::string& p) {
    int rc;
    {   // stem: [32][3][3][3] -> [27][32], k = (c*3+ky)*3+kx
        const auto& w = h->host[p + "._conv_stem.weight"];
        BnFold f = fold_bn(h, p + "._bn0");
        std::vector<float> pk(27 * 32);
        for (int o = 0; o < 32; ++o)
            for (int k = 0; k < 27; ++k) pk[k * 32 + o] = w[o * 27 + k] * f.scale[o];
        if ((rc = upload(h, pk, &e.stem_w))) return rc;
        if ((rc = upload(h, f.shift, &e.stem_b))) return rc;
    }
    for (int i = 0; i < 16; ++i) {
        const BlockSpec& b = B0[i];
        BlockW& bw = e.blk[i];
        std::string q = p + "._blocks." + std::to_string(i);
        const int mid = b.cin * b.e;
        if (b.e != 1 && (rc = pack_pointwise_bn(h, bw.expand, q + "._expand_conv.weight", q + "._bn0", mid, b.cin))) return rc;
        if (b.e != 1 && (mbconv_front_supported(b.k, b.s, b.cin, mid) || b.cin % 8 == 0)) {   // linear copy for the fused front kernels
            const auto& w = h->host[q + "._expand_conv.weight"];
            BnFold f = fold_bn(h, q + "._bn0");
            bw.exp_cinp = round_up(b.cin, 16);
            std::vector<float> lin((size_t)mid * bw.exp_cinp, 0.f);
            for (int n = 0; n < mid; ++n)
                for (int c = 0; c < b.cin; ++c) lin[(size_t)n * bw.exp_cinp + c] = w[(size_t)n * b.cin + c] * f.scale[n];
            if ((rc = upload(h, lin, &bw.exp_lin))) return rc;
        }
        {
            const auto& w = h->host[q + "._depthwise_conv.weight"];
            BnFold f = fold_bn(h, q + "._bn1");
            const int kk = b.k * b.k;
            std::vector<float> pk((size_t)kk * mid);
            for (int c = 0; c < mid; ++c)
                for (int t = 0; t < kk; ++t) pk[(size_t)t * mid + c] = w[(size_t)c * kk + t] * f.scale[c];
            if ((rc = upload(h, pk, &bw.dw_w))) return rc;
            if ((rc = upload(h, f.shift, &bw.dw_b))) return rc;
        }
        bw.sq = se_squeeze(b.cin);
        if ((rc = upload(h, h->host[q + "._se_reduce.weight"], &bw.se_w1))) return rc;
        if ((rc = upload(h, h->host[q + "._se_reduce.bias"], &bw.se_b1))) return rc;
        {   // [C][SQ] -> [SQ][C] so the excite phase reads consecutive channels
            const auto& w2 = h->host[q + "._se_expand.weight"];
            std::vector<float> t((size_t)mid * bw.sq);
            for (int c = 0; c < mid; ++c)
                for (int j = 0; j < bw.sq; ++j) t[(size_t)j * mid + c] = w2[(size_t)c * bw.sq + j];
            if ((rc = upload(h, t, &bw.se_w2))) return rc;
        }
        if ((rc = upload(h, h->host[q + "._se_expand.bias"], &bw.se_b2))) return rc;
        if ((rc = pack_pointwise_bn(h, bw.project, q + "._project_conv.weight", q + "._bn2", b.cout, mid))) return rc;
    }
    return pack_pointwise_bn(h, e.head, p + "._conv_head.weight", p + "._bn1", 1280, 320);
}

static int score_pad(int nscore) { return round_up(nscore, 8); }

static int build_decoder(ccvpe_handle_s* h, DecoderW& d, const DecLevel* lv, const std::string& sfx, int nscore_l6, bool every_level_scored) {
    int rc;
    for (int j = 0; j < 6; ++j) {
        std::string n = std::to_string(6 - j);
        {   // ConvTranspose2d weight [cin][cout][2][2] -> rows n = (dy*2+dx)*cout + o
            const auto& w = h->host["deconv" + n + sfx + ".weight"];
            const auto& b = h->host["deconv" + n + sfx + ".bias"];
            const int cin = lv[j].din, cout = lv[j].dout;
            int nscore = 0;
            if (every_level_scored) nscore = 1;
            else if (j == 0) nscore = nscore_l6;
            const int spad = score_pad(nscore);
            const int cinp = spad + (cin - nscore);
            std::vector<int> cmap(cin);
            for (int c = 0; c < cin; ++c) cmap[c] = c < nscore ? c : c - nscore + spad;
            std::vector<float> bias(4 * cout);
            for (int qd = 0; qd < 4; ++qd) for (int o = 0; o < cout; ++o) bias[qd * cout + o] = b[o];
            if ((rc = pack_conv(h, d.deconv[j], 4 * cout, 1, cin, cinp, cmap,
                                [&](int nn, int, int c) { int qd = nn / cout, o = nn % cout; return w[((size_t)c * cout + o) * 4 + qd]; },
                                bias, 1, 1))) return rc;
        }
        if (j == 5) {   // dedicated layouts for the fused last level
            const auto& w = h->host["deconv" + n + sfx + ".weight"];
            const auto& b = h->host["deconv" + n + sfx + ".bias"];
            const int cin = lv[j].din, cout = lv[j].dout;   // cout == 16
            const int nscore = every_level_scored ? 1 : 0;
            const int spad = score_pad(nscore);
            d.l1_cx = spad + (cin - nscore);
            d.l1_cxp = round_up(d.l1_cx, 16);
            std::vector<float> wd((size_t)64 * d.l1_cxp, 0.f);
            for (int c = 0; c < cin; ++c) {
                const int cm = c < nscore ? c : c - nscore + spad;
                for (int o = 0; o < cout; ++o)
                    for (int qd = 0; qd < 4; ++qd) wd[(size_t)(qd * 16 + o) * d.l1_cxp + cm] = w[((size_t)c * cout + o) * 4 + qd];
            }
            if ((rc = upload(h, wd, &d.l1_wd))) return rc;
            if ((rc = upload(h, b, &d.l1_bd))) return rc;
            const auto& wa = h->host["conv" + n + sfx + ".0.weight"];   // [16][16][3][3] -> [16][144], k = tap*16 + c
            std::vector<float> pk(16 * 144);
            for (int o = 0; o < 16; ++o)
                for (int c = 0; c < 16; ++c)
                    for (int t = 0; t < 9; ++t) pk[o * 144 + t * 16 + c] = wa[((size_t)o * 16 + c) * 9 + t];
            if ((rc = upload(h, pk, &d.l1_wa))) return rc;
            if ((rc = upload(h, h->host["conv" + n + sfx + ".0.bias"], &d.l1_ba))) return rc;
        }
        {
            const auto& w = h->host["conv" + n + sfx + ".0.weight"];
            const int cin = lv[j].dout + lv[j].skip, cout = lv[j].mid;
            if ((rc = pack_conv(h, d.conva[j], cout, 9, cin, cin, identity_map(cin),
                                [&](int nn, int t, int c) { return w[((size_t)nn * cin + c) * 9 + t]; },
                                h->host["conv" + n + sfx + ".0.bias"], 3, 3))) return rc;
        }
        const auto& w2 = h->host["conv" + n + sfx + ".2.weight"];
        const auto& b2 = h->host["conv" + n + sfx + ".2.bias"];
        if (j < 5) {
            const int cin = lv[j].mid, cout = lv[j].out;
            if ((rc = pack_conv(h, d.convb[j], cout, 9, cin, cin, identity_map(cin),
                                [&](int nn, int t, int c) { return w2[((size_t)nn * cin + c) * 9 + t]; }, b2, 3, 3))) return rc;
        } else {   // tail: [cout][16][3][3] -> [9][16][cout]
            const int cout = lv[j].out;
            std::vector<float> pk(9 * 16 * cout);
            for (int o = 0; o < cout; ++o)
                for (int c = 0; c < 16; ++c)
                    for (int t = 0; t < 9; ++t) pk[(t * 16 + c) * cout + o] = w2[((size_t)o * 16 + c) * 9 + t];
            if ((rc = upload(h, pk, &d.tail_w))) return rc;
            std::vector<float> pk1(9 * 16 * cout);
            for (int o = 0; o < cout; ++o)
                for (int c = 0; c < 16; ++c)
                    for (int t = 0; t < 9; ++t) pk1[(t * cout + o) * 16 + c] = w2[((size_t)o * 16 + c) * 9 + t];
            if ((rc = upload(h, pk1, &d.l1_wt))) return rc;
            for (int o = 0; o < cout; ++o) d.tail_b[o] = b2[o];
        }
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------
// roll shifts (spec.py roll_shifts / full_roll_shifts; models.py:192-193, 489-491, 1094)
// ------------------------------------------------------------------------------------------------
static int window_offset(const VariantSpec& vs, int level, int L) {
    const int C = vs.match_ch[level];
    return vs.centre ? (int)((double)C / 2 - (double)L / 2) : 0;
}
static int mod(int a, int m) { int r = a % m; return r < 0 ? r + m : r; }

// ------------------------------------------------------------------------------------------------
// plan construction
// ------------------------------------------------------------------------------------------------
static ConvParams conv_params(const PackedConv& pc, const float* in, int in_ld, int B, int H, int W, int OH, int OW,
                              int stride, int pad_t, int pad_l, int act) {
    ConvParams p{};
    p.in = in; p.in_ld = in_ld; p.B = B; p.H = H; p.W = W; p.Cin = pc.cinp; p.OH = OH; p.OW = OW;
    p.KH = pc.KH; p.KW = pc.KW; p.stride = stride; p.pad_t = pad_t; p.pad_l = pad_l;
    p.wpk = pc.w; p.w_hi = pc.w_hi; p.w_lo = pc.w_lo; p.Kpad = pc.Kpad; p.Npad = round_up(pc.N, conv_igemm_npad()); p.nchunks = pc.nchunks; p.bias = pc.bias; p.N = pc.N; p.act = act;
    p.gate = nullptr; p.resid = nullptr; p.resid_ld = 0; p.ndst = 0; p.mode = MODE_CONV; p.deconv_cout = 0;
    p.M = B * OH * OW;
    p.in_bytes = (unsigned)((size_t)B * H * W * in_ld * sizeof(float));
    p.gate_bytes = (unsigned)((size_t)B * pc.cinp * sizeof(float));
    p.w_plane_bytes = (unsigned)((size_t)p.Npad * pc.Kpad * sizeof(unsigned short));
    p.wino_w = pc.wino; p.wino_n16 = pc.wino_n16; p.wino_bytes = (unsigned)pc.wino_bytes;
    p.wino4_w = pc.wino4; p.wino4_bytes = (unsigned)pc.wino4_bytes;
    return p;
}

struct EncOut { Tensor vol; Tensor tap[16]; };

// dsts for tap blocks: concat tensors the project GEMM also writes into (level index 0..4 -> block TAP_BLOCK[i])
struct TapDst { Tensor t[2]; int coff[2]; int n = 0; };

static void plan_encoder(ccvpe_handle_s* h, Plan& pl, const EncoderW& ew, bool is_grd, int B, int H, int W, bool circular,
                         const TapDst* tapdst, EncOut& out, const std::string& tag) {
    int lo, hi;
    static_pad(3, 2, lo, hi);
    int ch = conv_out(H, 3, 2), cw = conv_out(W, 3, 2);
    Tensor cur = pl.alloc(B, ch, cw, 32);
    {
        StemParams sp{};
        sp.B = B; sp.H = H; sp.W = W; sp.OH = ch; sp.OW = cw; sp.pad_t = lo; sp.pad_l = lo; sp.circular = circular;
        sp.w = ew.stem_w; sp.bias = ew.stem_b;
        Tensor o = cur;
        pl.add(tag + ".stem", {o}, [sp, o, is_grd](const Ctx& c) {
            StemParams q = sp; q.in = is_grd ? c.grd : c.sat; q.out = c.ptr(o);
            launch_stem(q, c.stream);
        }, 2.0 * B * ch * cw * 32 * 27, 4.0 * B * (3.0 * H * W + 32.0 * ch * cw));
    }
    for (int i = 0; i < 16; ++i) {
        const BlockSpec& b = B0[i];
        const BlockW& bw = ew.blk[i];
        const int mid = b.cin * b.e;
        const std::string bn = tag + ".b" + std::to_string(i);
        Tensor xin = cur;
        Tensor e = xin;
        static_pad(b.k, b.s, lo, hi);
        const int oh = conv_out(ch, b.k, b.s), ow = conv_out(cw, b.k, b.s);
        MbFrontParams mp{};
        mp.B = B; mp.H = ch; mp.W = cw; mp.Cin = b.cin; mp.cinp = bw.exp_cinp; mp.mid = mid;
        mp.we = bw.exp_lin; mp.be = bw.expand.bias; mp.wd = bw.dw_w; mp.bd = bw.dw_b;
        mp.k = b.k; mp.s = b.s; mp.pad_t = lo; mp.pad_l = lo; mp.circular = circular; mp.OH = oh; mp.OW = ow;
        // small-spatial blocks: the whole expanded image of 16 channels lives in LDS (kernels_mbimg.hip); CCVPE_FUSE_MBCONV=0 / CCVPE_MBCONV_IMAGE=0 turn it off
        const bool image_off = getenv("CCVPE_MBCONV_IMAGE") && std::atoi(getenv("CCVPE_MBCONV_IMAGE")) == 0;   // read per plan: tests toggle it
        const bool image = b.e != 1 && bw.exp_lin != nullptr && h->fuse_mbconv != 0 && !image_off && mbconv_image_supported(mp);
        const bool fused = image || (b.e != 1 && bw.exp_lin != nullptr && mbconv_front_supported(b.k, b.s, b.cin, mid) &&
                           (h->fuse_mbconv == 2 || (h->fuse_mbconv == 1 && mbconv_front_profitable(b.k))));
        Tensor d = pl.alloc(B, oh, ow, mid);
        const int S = image ? mbconv_image_strips(mp) : fused ? mbconv_front_tiles(b.k, b.s, oh, ow) : depthwise_strip_lanes(B, oh, ow, mid, b.k, b.s);
        Tensor pool = pl.alloc(B, 1, S, mid);
        if (fused) {
            pl.add(bn + ".expand_dw", {xin, d, pool}, [=](const Ctx& c) {
                MbFrontParams q = mp; q.x = c.ptr(xin); q.out = c.ptr(d); q.pool = c.ptr(pool);
                if (image) launch_mbconv_image(q, c.stream); else launch_mbconv_front(q, c.stream);
            }, 2.0 * B * ch * cw * b.cin * mid + 2.0 * B * oh * ow * mid * b.k * b.k, 4.0 * B * ((double)ch * cw * b.cin + (double)oh * ow * mid));
        } else {
        if (b.e != 1) {
            e = pl.alloc(B, ch, cw, mid);
            const PackedConv* pc = &bw.expand;
            const int hh = ch, ww = cw;
            pl.add_conv(bn + ".expand", {xin, e}, B * hh * ww, pc->N, pc->Kpad, [=](const Ctx& c, int tile) {
                ConvParams p = conv_params(*pc, c.ptr(xin), xin.C, B, hh, ww, hh, ww, 1, 0, 0, ACT_SWISH);
                p.dst[0] = {c.ptr(e), mid, 0}; p.ndst = 1;
                c.launch_conv(p, tile);
            }, 2.0 * B * ch * cw * b.cin * mid, 4.0 * B * ch * cw * (b.cin + mid));
            pl.ops.back().is_pw = true;
        }
        {
            DwParams dp{};
            dp.B = B; dp.H = ch; dp.W = cw; dp.C = mid; dp.OH = oh; dp.OW = ow; dp.k = b.k; dp.stride = b.s;
            dp.pad_t = lo; dp.pad_l = lo; dp.circular = circular; dp.w = bw.dw_w; dp.bias = bw.dw_b; dp.S = S;
            pl.add(bn + ".dw", {e, d, pool}, [=](const Ctx& c) {
                DwParams q = dp; q.in = c.ptr(e); q.out = c.ptr(d); q.pool_partial = c.ptr(pool);
                launch_depthwise(q, c.stream);
            }, 2.0 * B * oh * ow * mid * b.k * b.k, 4.0 * B * mid * ((double)ch * cw + (double)oh * ow));
        }
        }
        Tensor gate = pl.alloc(B, 1, 1, mid);
        const int SC = std::max(1, std::min(16, S / 32));
        Tensor pooled = pl.alloc(B, 1, SC, mid);
        Tensor sqt = pl.alloc(B, 1, 1, 64);
        {
            SeParams sp{};
            sp.B = B; sp.S = S; sp.C = mid; sp.SQ = bw.sq; sp.inv_hw = 1.f / (float)(oh * ow); sp.SC = SC;
            sp.w1 = bw.se_w1; sp.b1 = bw.se_b1; sp.w2 = bw.se_w2; sp.b2 = bw.se_b2;
            pl.add(bn + ".se", {pool, gate, pooled, sqt}, [=](const Ctx& c) {
                SeParams q = sp; q.pool_partial = c.ptr(pool); q.gate = c.ptr(gate); q.pooled = c.ptr(pooled); q.sq = c.ptr(sqt);
                launch_se(q, c.stream);
            }, 4.0 * B * mid * bw.sq, 4.0 * B * S * mid);
        }
        Tensor o = pl.alloc(B, oh, ow, b.cout);
        {
            const PackedConv* pc = &bw.project;
            const bool skip = (b.s == 1 && b.cin == b.cout);
            TapDst td;
            if (tapdst) for (int t = 0; t < 5; ++t) if (TAP_BLOCK[t] == i) td = tapdst[t];
            std::vector<Tensor> uses = {d, gate, o};
            if (skip) uses.push_back(xin);
            for (int t = 0; t < td.n; ++t) uses.push_back(td.t[t]);
            pl.add_conv(bn + ".project", uses, B * oh * ow, pc->N, pc->Kpad, [=](const Ctx& c, int tile) {
                ConvParams p = conv_params(*pc, c.ptr(d), mid, B, oh, ow, oh, ow, 1, 0, 0, ACT_NONE);
                p.gate = c.ptr(gate);
                if (skip) { p.resid = c.ptr(xin); p.resid_ld = xin.C; }
                p.dst[0] = {c.ptr(o), o.C, 0}; p.ndst = 1;
                for (int t = 0; t < td.n; ++t) p.dst[p.ndst++] = c.dst(td.t[t], td.coff[t]);
                c.launch_conv(p, tile);
            }, 2.0 * B * oh * ow * mid * b.cout, 4.0 * B * oh * ow * (mid + b.cout * (1 + td.n)));
            pl.ops.back().is_pw = true;
        }
        out.tap[i] = o;
        pl.taps[tag + "_block" + std::to_string(i)] = {o, 0, o.C};
        cur = o; ch = oh; cw = ow;
    }
    Tensor vol = pl.alloc(B, ch, cw, 1280);
    {
        const PackedConv* pc = &ew.head;
        Tensor x = cur;
        const int hh = ch, ww = cw;
        pl.add_conv(tag + ".head", {x, vol}, B * hh * ww, pc->N, pc->Kpad, [=](const Ctx& c, int tile) {
            ConvParams p = conv_params(*pc, c.ptr(x), x.C, B, hh, ww, hh, ww, 1, 0, 0, ACT_SWISH);
            p.dst[0] = {c.ptr(vol), 1280, 0}; p.ndst = 1;
            c.launch_conv(p, tile);
        }, 2.0 * B * ch * cw * 320 * 1280, 4.0 * B * ch * cw * 1600);
        pl.ops.back().is_pw = true;
    }
    out.vol = vol;
    pl.taps[tag + "_volume"] = {vol, 0, 1280};
}

// Aerial cache: everything the decoders need from the aerial image, NHWC fp32, batch-major:
// [descriptor map B x 8x8 x D | block15 B x 16^2 x 320 | block10 B x 32^2 x 112 | block4 B x 64^2 x 40 |
//  block2 B x 128^2 x 24 | block0 B x 256^2 x 16]
static const int TAP_HW[5] = {256, 1024, 4096, 16384, 65536};
static const int TAP_C[5] = {320, 112, 40, 24, 16};
static size_t cache_layout(const VariantSpec& vs, int B, size_t off[6]) {
    size_t o = 0;
    off[0] = o; o += (size_t)B * 64 * vs.sat_desc;
    for (int t = 0; t < 5; ++t) { off[t + 1] = o; o += (size_t)B * TAP_HW[t] * TAP_C[t]; }
    return o;
}

static int build_aerial_plan(ccvpe_handle_s* h, Plan& pl, int B);
extern "C" int ccvpe_max_micro_batch(int32_t variant, float ori_noise, int32_t grd_h, int32_t grd_w);

static int build_plan(ccvpe_handle_s* h, Plan& pl, int B, int gh, int gw, int mode = 0) {
    if (mode == 1) return build_aerial_plan(h, pl, B);
    const bool cached = mode == 2;
    pl.mode = mode;
    const VariantSpec& vs = h->vs;
    pl.B = B; pl.gh = gh; pl.gw = gw; pl.debug = h->debug;
    pl.scratch = pl.alloc(1, 1, 1, (int)Plan::SPLITK_FLOATS);
    pl.two_streams = h->two_streams && !h->debug;
    if (pl.two_streams) pl.scratch2 = pl.alloc(1, 1, 1, (int)Plan::SPLITK_FLOATS);
    pl.use_graph = !cached && (h->graph_mode == 1 || (h->graph_mode < 0 && B <= 4));
    if (pl.use_graph) {
        pl.io_grd = pl.alloc(B, 3, gh, gw);
        pl.io_sat = pl.alloc(B, 3, CCVPE_SAT_HW, CCVPE_SAT_HW);
        pl.io_logits = pl.alloc(B, 1, CCVPE_OUT_HW, CCVPE_OUT_HW);
        pl.io_heat = pl.alloc(B, 1, CCVPE_OUT_HW, CCVPE_OUT_HW);
        pl.io_ori = pl.alloc(B, 2, CCVPE_OUT_HW, CCVPE_OUT_HW);
        for (int k = 0; k < 6; ++k) pl.io_ms[k] = pl.alloc(B, h->rolls[k], 8 << k, 8 << k);
    }

    // ---- geometry of the ground feature volume ----
    int fh = conv_out(gh, 3, 2), fw = conv_out(gw, 3, 2);
    for (int i = 0; i < 16; ++i) { fh = conv_out(fh, B0[i].k, B0[i].s); fw = conv_out(fw, B0[i].k, B0[i].s); }
    if (fh != vs.feat_h)
        return fail(CCVPE_EINVAL, "ground image %dx%d gives a %d-row feature volume, the descriptor heads expect %d rows", gh, gw, fh, vs.feat_h);
    int L[6];
    for (int k = 0; k < 6; ++k) {
        L[k] = fw * vs.head_ch[k];
        if (L[k] > vs.match_ch[k])
            return fail(CCVPE_EINVAL, "descriptor length %d exceeds aerial channels %d at level %d", L[k], vs.match_ch[k], k + 1);
    }

    // ---- decoder concat buffers (allocated first: the aerial encoder's tap epilogues write into them) ----
    const int D = vs.sat_desc;
    const int rfull = vs.n_rolls;
    const int rpad = score_pad(rfull);
    Tensor loc_in[6], ori_in6;          // deconv inputs: [score pad 8 | C]
    Tensor loc_cat[6], ori_cat[6];      // deconv out + skip (level index j = 0..5 <-> decoder level 6-j)
    for (int j = 0; j < 6; ++j) {
        const int hw_in = 8 << j;
        loc_in[j] = pl.alloc(B, hw_in, hw_in, 8 + vs.match_ch[j]);
        loc_cat[j] = pl.alloc(B, hw_in * 2, hw_in * 2, vs.loc[j].dout + vs.loc[j].skip);
        ori_cat[j] = pl.alloc(B, hw_in * 2, hw_in * 2, vs.ori[j].dout + vs.ori[j].skip);
        // bf16x3 mode: tensors consumed only by convolutions live as pre-split bf16 planes (same bytes), so the
        // consumers' K loops carry no fp32->bf16 conversion; level 1 (j == 5) feeds the fp32 tail and stays fp32
        if (h->cfg.reserved[0] == 1 && j < 5 && !getenv("CCVPE_NO_SPLIT_PLANES")) { loc_cat[j].split = true; ori_cat[j].split = true; }
    }
    ori_in6 = pl.alloc(B, 8, 8, rpad + D);

    // ---- encoders ----
    EncOut genc, senc;
    plan_encoder(h, pl, h->grd_enc, true, B, gh, gw, h->cfg.circular_padding != 0, nullptr, genc, "grd");
    TapDst td[5];
    for (int t = 0; t < 5; ++t) {
        td[t].n = 2;
        td[t].t[0] = loc_cat[t]; td[t].coff[0] = vs.loc[t].dout;
        td[t].t[1] = ori_cat[t]; td[t].coff[1] = vs.ori[t].dout;
    }
    size_t coff[6];
    cache_layout(vs, B, coff);
    if (!cached) {
        plan_encoder(h, pl, h->sat_enc, false, B, CCVPE_SAT_HW, CCVPE_SAT_HW, false, td, senc, "sat");
    } else {
        for (int t = 0; t < 5; ++t) {   // cached encoder taps -> skip halves of the decoder concat buffers
            Tensor lc = loc_cat[t], oc = ori_cat[t];
            const int lcoff = vs.loc[t].dout, ocoff = vs.ori[t].dout;
            const size_t src_off = coff[t + 1];
            const int C = TAP_C[t];
            const long long P = (long long)B * TAP_HW[t];
            pl.add("sat.cached_tap" + std::to_string(TAP_BLOCK[t]), {lc, oc}, [=](const Ctx& c) {
                launch_scatter_channels(c.cache_in + src_off, C, P, c.dst(lc, lcoff), c.dst(oc, ocoff), 2, c.stream);
            }, 0, 4.0 * P * C * 3);
        }
    }

    // ---- ground descriptors ----
    int ntot = 0, ltot = 0, hoff[6], loff[6];
    for (int k = 0; k < 6; ++k) { hoff[k] = ntot; ntot += vs.head_ch[k]; loff[k] = ltot; ltot += round_up(L[k], 4); }
    Tensor ghead = pl.alloc(B, fh, fw, ntot);
    Tensor desc = pl.alloc(B, 1, 1, ltot);
    {
        const PackedConv* pc = &h->grd_heads;
        Tensor x = genc.vol;
        pl.add_conv("grd.heads", {x, ghead}, B * fh * fw, pc->N, pc->Kpad, [=](const Ctx& c, int tile) {
            ConvParams p = conv_params(*pc, c.ptr(x), 1280, B, fh, fw, fh, fw, 1, 0, 0, ACT_NONE);
            p.dst[0] = {c.ptr(ghead), ntot, 0}; p.ndst = 1;
            c.launch_conv(p, tile);
        }, 2.0 * B * fh * fw * 1280 * ntot, 4.0 * B * fh * fw * (1280 + ntot));
        GrdDescParams gp{};
        gp.B = B; gp.Hf = fh; gp.Wf = fw; gp.Ntot = ntot; gp.nlev = 6; gp.Ltot = ltot;
        for (int k = 0; k < 6; ++k) { gp.c[k] = vs.head_ch[k]; gp.off[k] = hoff[k]; gp.wh[k] = h->grd_wh[k]; gp.b2[k] = h->grd_b2[k]; gp.loff[k] = loff[k]; }
        pl.add("grd.desc", {ghead, desc}, [=](const Ctx& c) {
            GrdDescParams q = gp; q.y = c.ptr(ghead); q.desc = c.ptr(desc);
            launch_grd_desc(q, c.stream);
        }, 2.0 * B * fh * fw * ntot, 4.0 * B * fh * fw * ntot);
        for (int k = 0; k < 6; ++k) pl.taps["grd_desc" + std::to_string(k + 1)] = {desc, loff[k], L[k]};
    }

    // ---- aerial descriptor map: conv k2 s2 over the 1280x16x16 volume ----
    Tensor dmap = pl.alloc(B, 8, 8, D);
    if (cached) {
        Tensor dm = dmap;
        const long long P = (long long)B * 64;
        pl.add("sat.cached_descmap", {dm}, [=](const Ctx& c) {
            launch_scatter_channels(c.cache_in, D, P, c.dst(dm), Dst{nullptr, 0, 0, 0, 0}, 1, c.stream);
        }, 0, 8.0 * P * D);
        pl.taps["sat_descriptor_map"] = {dmap, 0, D};
    } else {
        const PackedConv* pc = &h->sat_desc;
        Tensor x = senc.vol;
        pl.add_conv("sat.descmap", {x, dmap}, B * 8 * 8, pc->N, pc->Kpad, [=](const Ctx& c, int tile) {
            ConvParams p = conv_params(*pc, c.ptr(x), 1280, B, 16, 16, 8, 8, 2, 0, 0, ACT_NONE);
            p.dst[0] = {c.ptr(dmap), D, 0}; p.ndst = 1;
            c.launch_conv(p, tile);
        }, 2.0 * B * 64 * 5120.0 * D, 4.0 * (B * 256 * 1280.0 + 5120.0 * D));
        pl.taps["sat_descriptor_map"] = {dmap, 0, D};
    }

    // ---- decoders ----
    auto plan_level = [&](const DecoderW& dw, const DecLevel* lv, int j, Tensor din, Tensor cat, const std::string& tag) -> Tensor {
        const int hin = 8 << j, hout = hin * 2;
        const DecLevel& l = lv[j];
        {
            const PackedConv* pc = &dw.deconv[j];
            const int cout = l.dout;
            pl.add_conv(tag + ".deconv", {din, cat}, B * hin * hin, pc->N, pc->Kpad, [=](const Ctx& c, int tile) {
                ConvParams p = conv_params(*pc, c.ptr(din), din.C, B, hin, hin, hin, hin, 1, 0, 0, ACT_NONE);
                p.mode = MODE_DECONV; p.deconv_cout = cout;
                p.dst[0] = c.dst(cat); p.ndst = 1;
                c.launch_conv(p, tile);
            }, 2.0 * B * hin * hin * (double)l.din * 4 * l.dout, 4.0 * B * hin * hin * ((double)din.C + 4.0 * l.dout));
            pl.ops.back().is_pw = !din.split;
        }
        Tensor mid = pl.alloc(B, hout, hout, l.mid);
        mid.split = cat.split;   // bf16x3 mode: conv_a -> conv_b hand-off stays in split bf16 form
        {
            const PackedConv* pc = &dw.conva[j];
            pl.add_conv(tag + ".conv_a", {cat, mid}, B * hout * hout, pc->N, pc->Kpad, [=](const Ctx& c, int tile) {
                ConvParams p = conv_params(*pc, c.ptr(cat), cat.C, B, hout, hout, hout, hout, 1, 1, 1, ACT_RELU);
                p.in_split = cat.split; p.in_plane_bytes = (unsigned)(cat.numel() * 2);
                p.dst[0] = c.dst(mid); p.ndst = 1;
                c.launch_conv(p, tile);
            }, 2.0 * B * hout * hout * 9.0 * cat.C * l.mid, 4.0 * B * hout * hout * ((double)cat.C + l.mid));
            pl.ops.back().bf16x3_only = cat.split;
            pl.ops.back().conv_cin = cat.C;
            pl.ops.back().wino_ok = pc->wino != nullptr && !cat.split && h->wino && h->cfg.reserved[0] == 0;
            pl.ops.back().wino4_ok = pl.ops.back().wino_ok && pc->wino4 != nullptr;
        }
        if (j == 5) return mid;   // tail conv handled by the caller
        Tensor o = pl.alloc(B, hout, hout, l.out);
        {
            const PackedConv* pc = &dw.convb[j];
            pl.add_conv(tag + ".conv_b", {mid, o}, B * hout * hout, pc->N, pc->Kpad, [=](const Ctx& c, int tile) {
                ConvParams p = conv_params(*pc, c.ptr(mid), mid.C, B, hout, hout, hout, hout, 1, 1, 1, ACT_NONE);
                p.in_split = mid.split; p.in_plane_bytes = (unsigned)(mid.numel() * 2);
                p.dst[0] = {c.ptr(o), o.C, 0}; p.ndst = 1;
                c.launch_conv(p, tile);
            }, 2.0 * B * hout * hout * 9.0 * l.mid * l.out, 4.0 * B * hout * hout * ((double)l.mid + l.out));
            pl.ops.back().bf16x3_only = mid.split;
            pl.ops.back().conv_cin = mid.C;
            pl.ops.back().wino_ok = pc->wino != nullptr && !mid.split && h->wino && h->cfg.reserved[0] == 0;
            pl.ops.back().wino4_ok = pl.ops.back().wino_ok && pc->wino4 != nullptr;
        }
        return o;
    };

    // fused last level: deconv1 + conv1[0] + ReLU + conv1[2] (+ normalize) in one launch
    auto plan_level1_fused = [&](const DecoderW& dw, Tensor din, int cin_real, int cout, bool is_ori, Tensor raw, const std::string& tag) {
        Level1Params lp{};
        lp.x_ld = din.C; lp.cx = dw.l1_cx; lp.cxp = dw.l1_cxp; lp.B = B; lp.H = CCVPE_OUT_HW; lp.W = CCVPE_OUT_HW;
        lp.wd = dw.l1_wd; lp.bd = dw.l1_bd; lp.wa = dw.l1_wa; lp.ba = dw.l1_ba; lp.wt = dw.l1_wt;
        lp.bt[0] = dw.tail_b[0]; lp.bt[1] = dw.tail_b[1]; lp.cout = cout; lp.normalize = is_ori ? 1 : 0;
        const bool has_raw = raw.id >= 0;
        std::vector<Tensor> uses = {din};
        if (has_raw) uses.push_back(raw);
        const double px = (double)B * CCVPE_OUT_HW * CCVPE_OUT_HW;
        pl.add(tag + ".fused", uses, [=](const Ctx& c) {
            Level1Params q = lp;
            q.x = c.ptr(din);
            q.out = is_ori ? c.out.ori : c.out.logits_flattened;
            q.raw = has_raw ? c.ptr(raw) : nullptr;
            launch_level1(q, c.stream);
        }, px / 4 * 2.0 * cin_real * 64 + px * 2.0 * 144 * 16 + px * 2.0 * 144 * cout, 4.0 * (px / 4 * lp.cx + px * cout));
    };

    Tensor x = dmap;
    Tensor loc_mid;
    for (int k = 0; k < 6; ++k) {   // matching level k+1 feeds decoder level 6-k
        MatchParams mp{};
        const int hw = (8 << k) * (8 << k);
        const int C = vs.match_ch[k];
        mp.x_ld = x.C; mp.B = B; mp.HW = hw; mp.C = C; mp.g_ld = ltot; mp.L = L[k];
        const int off = window_offset(vs, k, L[k]);
        const bool prior = h->cfg.variant == CCVPE_VARIANT_VIGOR_ORI_PRIOR;
        const int n = prior ? (int)(h->cfg.ori_noise / 18.f) : 0;
        if (k == 0 || !prior) {
            mp.R = rfull;
            for (int r = 0; r < rfull; ++r) mp.shift[r] = mod(off + r * vs.step[k], C);
            mp.inmax = 0;
            if (prior) { for (int i = -n; i <= n; ++i) mp.inmax |= 1u << mod(i, rfull); }
            else mp.inmax = rfull >= 32 ? 0xffffffffu : ((1u << rfull) - 1u);
        } else {
            mp.R = 2 * n + 1;
            for (int r = 0; r < mp.R; ++r) mp.shift[r] = mod(off + (r - n) * vs.step[k], C);
            mp.inmax = (mp.R >= 32) ? 0xffffffffu : ((1u << mp.R) - 1u);
        }
        mp.rpad = rpad;
        mp.P = match_pixels_per_block(hw, C);
        mp.cat_max_ld = 8 + C;
        mp.cat_all_ld = rpad + C;
        Tensor xin = x, lin = loc_in[k];
        Tensor ggs = pl.alloc(B, 1, 1, (int)match_scratch_floats(C));
        const bool first = (k == 0);
        const int goff = loff[k];
        const int R = mp.R;
        std::vector<Tensor> uses = {xin, desc, lin, ggs};
        if (first) uses.push_back(ori_in6);
        pl.add("match" + std::to_string(k + 1), uses, [=](const Ctx& c) {
            MatchParams q = mp;
            q.x = c.ptr(xin); q.g = c.ptr(desc) + goff;
            q.ms = c.out.matching_score[k];
            q.cat_max = c.ptr(lin);
            q.cat_all = first ? c.ptr(ori_in6) : nullptr;
            q.gg_scratch = c.ptr(ggs);
            launch_match(q, c.stream);
        }, 4.0 * B * hw * (double)R * L[k], 4.0 * B * hw * (2.0 * C + R + 8));
        pl.taps["loc_in" + std::to_string(6 - k)] = {lin, 0, lin.C};
        if (k == 5 && h->fuse_level1) { plan_level1_fused(h->loc, lin, vs.loc[5].din, 1, false, Tensor{}, "loc1"); break; }
        Tensor o = plan_level(h->loc, vs.loc, k, lin, loc_cat[k], "loc" + std::to_string(6 - k));
        if (k < 5) { pl.taps["loc_level" + std::to_string(6 - k)] = {o, 0, o.C}; x = o; }
        else loc_mid = o;
    }
    if (!h->fuse_level1) {
        Tensor m = loc_mid;
        const float* tw = h->loc.tail_w;
        const float tb = h->loc.tail_b[0];
        pl.add("loc1.tail", {m}, [=](const Ctx& c) {
            TailConvParams p{};
            p.in = c.ptr(m); p.B = B; p.H = CCVPE_OUT_HW; p.W = CCVPE_OUT_HW; p.w = tw; p.bias[0] = tb; p.cout = 1;
            p.normalize = 0; p.out = c.out.logits_flattened; p.raw = nullptr;
            launch_tail_conv(p, c.stream);
        }, 2.0 * B * 262144.0 * 144, 4.0 * B * 262144.0 * 17);
    }
    {
        Tensor part = pl.alloc(B, 1, 64, 2);
        pl.add("softmax", {part}, [=](const Ctx& c) {
            SoftmaxParams p{};
            p.logits = c.out.logits_flattened; p.B = B; p.n = CCVPE_OUT_HW * CCVPE_OUT_HW; p.partial = c.ptr(part); p.chunks = 64;
            p.out = c.out.heatmap;
            launch_softmax(p, c.stream);
        }, 0, 4.0 * B * 262144.0 * 3);
    }
    // orientation decoder
    {
        Tensor xo = ori_in6;
        Tensor ori_mid;
        Tensor raw;
        if (h->debug) { raw = pl.alloc(B, 2, CCVPE_OUT_HW, CCVPE_OUT_HW); pl.taps["ori_level1_nchw"] = {raw, 0, -1}; }
        bool fused_done = false;
        for (int j = 0; j < 6; ++j) {
            if (j == 5 && h->fuse_level1) { plan_level1_fused(h->ori, xo, vs.ori[5].din, 2, true, raw, "ori1"); fused_done = true; break; }
            Tensor o = plan_level(h->ori, vs.ori, j, xo, ori_cat[j], "ori" + std::to_string(6 - j));
            if (j < 5) { pl.taps["ori_level" + std::to_string(6 - j)] = {o, 0, o.C}; xo = o; }
            else ori_mid = o;
        }
        if (!fused_done) {
        Tensor m = ori_mid;
        const float* tw = h->ori.tail_w;
        const float tb0 = h->ori.tail_b[0], tb1 = h->ori.tail_b[1];
        const bool dbg = h->debug;
        std::vector<Tensor> uses = {m};
        if (dbg) uses.push_back(raw);
        pl.add("ori1.tail", uses, [=](const Ctx& c) {
            TailConvParams p{};
            p.in = c.ptr(m); p.B = B; p.H = CCVPE_OUT_HW; p.W = CCVPE_OUT_HW; p.w = tw; p.bias[0] = tb0; p.bias[1] = tb1; p.cout = 2;
            p.normalize = 1; p.out = c.out.ori; p.raw = dbg ? c.ptr(raw) : nullptr;
            launch_tail_conv(p, c.stream);
        }, 2.0 * B * 262144.0 * 288, 4.0 * B * 262144.0 * 18);
        }
    }
    // the ground / aerial inputs and the 2 x 512 x 512 orientation output are addressed the same way
    pl.max_tensor_bytes = std::max(pl.max_tensor_bytes, (size_t)B * 3 * std::max(gh * gw, CCVPE_SAT_HW * CCVPE_SAT_HW) * sizeof(float));
    if (pl.max_tensor_bytes >= ((size_t)1 << 31))
        return fail(CCVPE_EINVAL, "micro-batch %d needs a %d x %d x %d x %d tensor of %zu bytes; the kernels address tensors with 32-bit byte offsets (< 2 GiB): "
                    "use a smaller micro_batch (ccvpe_max_micro_batch)", B, pl.max_dims[0], pl.max_dims[1], pl.max_dims[2], pl.max_dims[3], pl.max_tensor_bytes);
    pl.schedule();
    pl.assign();
    return 0;
}

static int build_aerial_plan(ccvpe_handle_s* h, Plan& pl, int B) {
    const VariantSpec& vs = h->vs;
    pl.B = B; pl.gh = 0; pl.gw = 0; pl.mode = 1; pl.debug = false;
    pl.scratch = pl.alloc(1, 1, 1, (int)Plan::SPLITK_FLOATS);
    EncOut senc;
    plan_encoder(h, pl, h->sat_enc, false, B, CCVPE_SAT_HW, CCVPE_SAT_HW, false, nullptr, senc, "sat");
    size_t coff[6];
    cache_layout(vs, B, coff);
    const int D = vs.sat_desc;
    pl.tune_cache = pl.alloc(B, 8, 8, D);
    {
        const PackedConv* pc = &h->sat_desc;
        Tensor x = senc.vol;
        pl.add_conv("sat.descmap", {x}, B * 8 * 8, pc->N, pc->Kpad, [=](const Ctx& c, int tile) {
            ConvParams p = conv_params(*pc, c.ptr(x), 1280, B, 16, 16, 8, 8, 2, 0, 0, ACT_NONE);
            p.dst[0] = {c.cache_out, D, 0}; p.ndst = 1;
            c.launch_conv(p, tile);
        }, 2.0 * B * 64 * 5120.0 * D, 4.0 * (B * 256 * 1280.0 + 5120.0 * D));
    }
    for (int t = 0; t < 5; ++t) {
        Tensor tp = senc.tap[TAP_BLOCK[t]];
        const size_t o = coff[t + 1];
        const size_t n = (size_t)B * TAP_HW[t] * TAP_C[t];
        pl.add("sat.tap_to_cache" + std::to_string(TAP_BLOCK[t]), {tp}, [=](const Ctx& c) {
            (void)hipMemcpyAsync(c.cache_out + o, c.ptr(tp), n * sizeof(float), hipMemcpyDeviceToDevice, c.stream);
        }, 0, 8.0 * n);
    }
    pl.assign();
    return 0;
}



// Integer checksum of a tensor's bytes (diagnostics: ccvpe_debug_dump_plan)
__global__ __launch_bounds__(256) void checksum_kernel(const uint32_t* __restrict__ p, size_t n, unsigned long long* out) {
    unsigned long long s = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) s += (unsigned long long)p[i] * (unsigned long long)((i & 1023) + 1);
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
    if ((threadIdx.x & 63) == 0) atomicAdd(out, s);
}


// ---- packed-weight cache --------------------------------------------------------------------------
// ccvpe_finalize_weights folds BatchNorm, repacks ~60 M parameters into the kernels' layouts and runs the Winograd weight
// transforms in double precision: seconds per handle.  Its result is a set of device buffers plus plain-data descriptor
// structs that point into them.  ccvpe_save_packed writes both to a file; ccvpe_load_packed recreates the buffers and
// re-bases every pointer of the descriptors (old device address -> new), so a later process skips the state_dict
// ingestion and the packing entirely.  The caller keys the file (ccvpe_amd/models.py: sha256 of the state_dict bytes,
// variant, precision, library build digest); the header carries variant / precision / struct sizes and is checked.
struct PackedHeader {
    char magic[8];                 // "CCVPEPK2"
    int32_t variant, precision, circular, fuse_level1;
    uint64_t n_allocs, sz_encoder, sz_decoder, sz_conv;
};
static void packed_state_io(ccvpe_handle_s* h, const std::function<void(void*, size_t)>& io) {
    io(&h->grd_enc, sizeof(EncoderW)); io(&h->sat_enc, sizeof(EncoderW));
    io(&h->grd_heads, sizeof(PackedConv)); io(&h->sat_desc, sizeof(PackedConv));
    io(h->grd_wh, sizeof(h->grd_wh)); io(h->grd_b2, sizeof(h->grd_b2));
    io(&h->loc, sizeof(DecoderW)); io(&h->ori, sizeof(DecoderW));
}

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
extern "C" {

const char* ccvpe_last_error(void) { return g_err.c_str(); }
const char* ccvpe_version(void) { return "ccvpe-hip 0.1 (gfx950, fp32 MFMA)"; }

int ccvpe_create(const ccvpe_config* cfg, ccvpe_handle* out) {
    if (!cfg || !out) return fail(CCVPE_EINVAL, "null argument");
    if (cfg->variant < 0 || cfg->variant > 3) return fail(CCVPE_EINVAL, "unknown variant %d", cfg->variant);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(CCVPE_EHIP, "no HIP device visible: libccvpe_hip has no CPU fallback");
    if (cfg->device < 0 || cfg->device >= ndev) return fail(CCVPE_EINVAL, "device %d out of range (%d visible)", cfg->device, ndev);
    HIPCHK(hipSetDevice(cfg->device));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, cfg->device));
    if (std::string(prop.gcnArchName).find("gfx950") == std::string::npos)
        return fail(CCVPE_EHIP, "device %d is %s; this library carries gfx950 code objects only", cfg->device, prop.gcnArchName);
    auto* h = new ccvpe_handle_s();
    h->cfg = *cfg;
    if (h->cfg.micro_batch <= 0) h->cfg.micro_batch = 32;
    h->vs = make_variant(cfg->variant);
    if (const char* e = getenv("CCVPE_AUTOTUNE")) h->autotune = std::atoi(e) != 0;
    if (const char* e = getenv("CCVPE_GRAPH")) h->graph_mode = std::atoi(e) != 0;
    if (const char* e = getenv("CCVPE_FUSE_L1")) h->fuse_level1 = std::atoi(e) != 0;
    if (const char* e = getenv("CCVPE_WINOGRAD")) h->wino = std::atoi(e) != 0;
    if (const char* e = getenv("CCVPE_STREAMS")) h->two_streams = std::atoi(e) >= 2;
    if (const char* e = getenv("CCVPE_FUSE_MBCONV")) h->fuse_mbconv = std::atoi(e);
    if (const char* e = getenv("CCVPE_DIAG_SYNC_BEFORE")) h->diag_sync = e;
    if (const char* e = getenv("CCVPE_DIAG_SNAP")) h->diag_snap = e;
    if (const char* e = getenv("CCVPE_PRECISION")) h->cfg.reserved[0] = (std::string(e) == "bf16x3") ? 1 : 0;
    if (h->cfg.reserved[0] != 0 && h->cfg.reserved[0] != 1) { delete h; return fail(CCVPE_EINVAL, "unknown precision mode %d", cfg->reserved[0]); }
    const int n = (int)(cfg->ori_noise / 18.f);
    for (int k = 0; k < 6; ++k)
        h->rolls[k] = (cfg->variant == CCVPE_VARIANT_VIGOR_ORI_PRIOR && k > 0) ? 2 * n + 1 : h->vs.n_rolls;
    if (cfg->variant == CCVPE_VARIANT_VIGOR_ORI_PRIOR && (n < 0 || 2 * n + 1 > 32)) {
        delete h;
        return fail(CCVPE_EINVAL, "ori_noise %.1f out of range", cfg->ori_noise);
    }
    build_expect(h);
    *out = h;
    return 0;
}

int ccvpe_max_micro_batch(int32_t variant, float ori_noise, int32_t grd_h, int32_t grd_w) {
    if (variant < 0 || variant > 3) return fail(CCVPE_EINVAL, "unknown variant %d", variant);
    // a device-less stand-in handle: build_plan only sizes tensors and records launches, it never touches HIP.
    // fuse_mbconv = 0 sizes the unfused (largest) form of every MBConv block, so the bound holds for every plan.
    ccvpe_handle_s tmp;
    tmp.cfg.variant = variant; tmp.cfg.ori_noise = ori_noise; tmp.cfg.micro_batch = 1;
    tmp.vs = make_variant(variant);
    tmp.fuse_mbconv = 0; tmp.two_streams = false; tmp.graph_mode = 0;
    const int n = (int)(ori_noise / 18.f);
    for (int k = 0; k < 6; ++k) tmp.rolls[k] = (variant == CCVPE_VARIANT_VIGOR_ORI_PRIOR && k > 0) ? 2 * n + 1 : tmp.vs.n_rolls;
    int lo = 0, hi = 1024;   // invariant: lo fits (0 = nothing fits / bad geometry), hi does not
    while (hi - lo > 1) {
        const int mid = (lo + hi) / 2;
        Plan pl;
        if (build_plan(&tmp, pl, mid, grd_h, grd_w) == 0) lo = mid; else hi = mid;
    }
    if (lo == 0) return fail(CCVPE_EINVAL, "ground size %d x %d is not valid for variant %d: %s", grd_h, grd_w, variant, g_err.c_str());
    return lo;
}

int ccvpe_destroy(ccvpe_handle h) {
    if (!h) return 0;
    (void)hipSetDevice(h->cfg.device);
    for (void* p : h->dev_allocs) (void)hipFree(p);
    if (h->arena) (void)hipFree(h->arena);
    h->plans.clear();
    for (int k = 0; k < 2; ++k) if (h->snap[k]) (void)hipFree(h->snap[k]);
    if (h->capture_stream) (void)hipStreamDestroy(h->capture_stream);
    if (h->aux_stream) (void)hipStreamDestroy(h->aux_stream);
    delete h;
    return 0;
}

int ccvpe_skip_weight(ccvpe_handle h, const char* key) {
    if (!h || !key) return fail(CCVPE_EINVAL, "null argument");
    if (!h->expect.count(key)) return fail(CCVPE_EKEY, "unexpected state_dict key '%s'", key);
    h->skipped.insert(key);
    return 0;
}

int ccvpe_set_weight(ccvpe_handle h, const char* key, const float* data, const int64_t* shape, int32_t ndim) {
    if (!h || !key || !data) return fail(CCVPE_EINVAL, "null argument");
    auto it = h->expect.find(key);
    if (it == h->expect.end()) return fail(CCVPE_EKEY, "unexpected state_dict key '%s'", key);
    const auto& es = it->second;
    bool ok = (int)es.size() == ndim;
    size_t n = 1;
    for (int i = 0; ok && i < ndim; ++i) { ok = es[i] == shape[i]; n *= (size_t)shape[i]; }
    if (!ok) return fail(CCVPE_EINVAL, "shape mismatch for '%s'", key);
    HIPCHK(hipSetDevice(h->cfg.device));
    std::vector<float> v(n);
    HIPCHK(hipMemcpy(v.data(), data, n * sizeof(float), hipMemcpyDefault));
    h->host[key] = std::move(v);
    h->finalized = false;
    return 0;
}

int ccvpe_finalize_weights(ccvpe_handle h) {
    if (!h) return fail(CCVPE_EINVAL, "null handle");
    HIPCHK(hipSetDevice(h->cfg.device));
    for (auto& kv : h->expect) {
        const std::string& k = kv.first;
        const bool optional = k.find("num_batches_tracked") != std::string::npos || k.find("._fc.") != std::string::npos;
        if (!h->host.count(k) && !(optional || h->skipped.count(k)))
            return fail(CCVPE_EKEY, "missing state_dict key '%s'", k.c_str());
        if (!h->host.count(k) && !optional) return fail(CCVPE_EKEY, "key '%s' was skipped but is required", k.c_str());
    }
    // drop previous device copies (re-finalize after a new load_state_dict)
    for (void* p : h->dev_allocs) (void)hipFree(p);
    h->dev_allocs.clear();
    h->dev_alloc_bytes.clear();
    h->plans.clear();
    h->last_plan = nullptr;
    int rc;
    if ((rc = build_encoder(h, h->grd_enc, "grd_efficientnet"))) return rc;
    if ((rc = build_encoder(h, h->sat_enc, "sat_efficientnet"))) return rc;
    {   // ground descriptor heads: one 1280 -> sum(c_k) pointwise GEMM, then per-level row weights
        int ntot = 0;
        for (int k = 0; k < 6; ++k) ntot += h->vs.head_ch[k];
        std::vector<float> bias(ntot);
        std::vector<const std::vector<float>*> ws(6);
        std::vector<int> lvl(ntot), loc(ntot);
        int o = 0;
        for (int k = 0; k < 6; ++k) {
            std::string p = "grd_feature_to_descriptor" + std::to_string(k + 1);
            ws[k] = &h->host[p + ".0.weight"];
            const auto& b = h->host[p + ".0.bias"];
            for (int c = 0; c < h->vs.head_ch[k]; ++c, ++o) { bias[o] = b[c]; lvl[o] = k; loc[o] = c; }
            if ((rc = upload(h, h->host[p + ".2.weight"], &h->grd_wh[k]))) return rc;
            h->grd_b2[k] = h->host[p + ".2.bias"][0];
        }
        if ((rc = pack_conv(h, h->grd_heads, ntot, 1, 1280, 1280, identity_map(1280),
                            [&](int n, int, int c) { return (*ws[lvl[n]])[(size_t)loc[n] * 1280 + c]; }, bias, 1, 1))) return rc;
    }
    {   // Linear(5120, D) == conv k2 s2: flat index ch*4 + dy*2 + dx (models.py:400-402, 471-482)
        const auto& w = h->host["sat_feature_to_descriptors.1.weight"];
        const int D = h->vs.sat_desc;
        if ((rc = pack_conv(h, h->sat_desc, D, 4, 1280, 1280, identity_map(1280),
                            [&](int n, int t, int c) { return w[(size_t)n * 5120 + c * 4 + t]; },
                            h->host["sat_feature_to_descriptors.1.bias"], 2, 2))) return rc;
    }
    if ((rc = build_decoder(h, h->loc, h->vs.loc, "", 1, true))) return rc;
    if ((rc = build_decoder(h, h->ori, h->vs.ori, "_ori", h->vs.n_rolls, false))) return rc;
    h->host.clear();
    if (!level1_supported(h->loc.l1_cxp) || !level1_supported(h->ori.l1_cxp)) h->fuse_level1 = false;   // > 64 input channels
    h->finalized = true;
    return 0;
}


int ccvpe_save_packed(ccvpe_handle h, const char* path) {
    if (!h || !path) return fail(CCVPE_EINVAL, "null argument");
    if (!h->finalized) return fail(CCVPE_ESTATE, "ccvpe_finalize_weights has not been called");
    static_assert(std::is_trivially_copyable<EncoderW>::value && std::is_trivially_copyable<DecoderW>::value && std::is_trivially_copyable<PackedConv>::value,
                  "descriptor structs are written as plain bytes");
    HIPCHK(hipSetDevice(h->cfg.device));
    const std::string tmp = std::string(path) + ".tmp";
    FILE* f = std::fopen(tmp.c_str(), "wb");
    if (!f) return fail(CCVPE_EINVAL, "cannot open %s for writing", tmp.c_str());
    PackedHeader hd{};
    std::memcpy(hd.magic, "CCVPEPK2", 8);
    hd.variant = h->cfg.variant; hd.precision = h->cfg.reserved[0]; hd.circular = h->cfg.circular_padding; hd.fuse_level1 = h->fuse_level1 ? 1 : 0;
    hd.n_allocs = h->dev_allocs.size(); hd.sz_encoder = sizeof(EncoderW); hd.sz_decoder = sizeof(DecoderW); hd.sz_conv = sizeof(PackedConv);
    bool ok = std::fwrite(&hd, sizeof(hd), 1, f) == 1;
    packed_state_io(h, [&](void* p, size_t n) { ok = ok && std::fwrite(p, 1, n, f) == n; });
    std::vector<char> buf;
    for (size_t i = 0; ok && i < h->dev_allocs.size(); ++i) {
        const uint64_t old = (uint64_t)(uintptr_t)h->dev_allocs[i], bytes = h->dev_alloc_bytes[i];
        buf.resize(bytes);
        if (hipMemcpy(buf.data(), h->dev_allocs[i], bytes, hipMemcpyDeviceToHost) != hipSuccess) { ok = false; break; }
        ok = std::fwrite(&old, 8, 1, f) == 1 && std::fwrite(&bytes, 8, 1, f) == 1 && std::fwrite(buf.data(), 1, bytes, f) == bytes;
    }
    ok = (std::fclose(f) == 0) && ok;
    if (!ok || std::rename(tmp.c_str(), path) != 0) { std::remove(tmp.c_str()); return fail(CCVPE_EINVAL, "writing %s failed", path); }
    return 0;
}

int ccvpe_load_packed(ccvpe_handle h, const char* path) {
    if (!h || !path) return fail(CCVPE_EINVAL, "null argument");
    HIPCHK(hipSetDevice(h->cfg.device));
    FILE* f = std::fopen(path, "rb");
    if (!f) return fail(CCVPE_EINVAL, "cannot open %s", path);
    PackedHeader hd{};
    auto bad = [&](const char* why) { std::fclose(f); return fail(CCVPE_EINVAL, "%s: %s", path, why); };
    if (std::fread(&hd, sizeof(hd), 1, f) != 1 || std::memcmp(hd.magic, "CCVPEPK2", 8) != 0) return bad("not a packed-weight file of this library version");
    if (hd.variant != h->cfg.variant || hd.precision != h->cfg.reserved[0] || hd.circular != h->cfg.circular_padding) return bad("packed for a different variant / precision / padding mode");
    if (hd.sz_encoder != sizeof(EncoderW) || hd.sz_decoder != sizeof(DecoderW) || hd.sz_conv != sizeof(PackedConv)) return bad("descriptor layout mismatch");
    for (void* p : h->dev_allocs) (void)hipFree(p);
    h->dev_allocs.clear(); h->dev_alloc_bytes.clear(); h->plans.clear(); h->last_plan = nullptr; h->finalized = false;
    bool ok = true;
    packed_state_io(h, [&](void* p, size_t n) { ok = ok && std::fread(p, 1, n, f) == n; });
    std::map<uint64_t, uint64_t> remap;   // old device address -> new
    std::vector<char> buf;
    for (uint64_t i = 0; ok && i < hd.n_allocs; ++i) {
        uint64_t old = 0, bytes = 0;
        if (std::fread(&old, 8, 1, f) != 1 || std::fread(&bytes, 8, 1, f) != 1 || bytes > ((uint64_t)1 << 33)) { ok = false; break; }
        buf.resize(bytes);
        if (std::fread(buf.data(), 1, bytes, f) != bytes) { ok = false; break; }
        void* d = nullptr;
        if (hipMalloc(&d, bytes) != hipSuccess) { ok = false; break; }
        h->dev_allocs.push_back(d); h->dev_alloc_bytes.push_back(bytes);
        if (hipMemcpy(d, buf.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) { ok = false; break; }
        remap[old] = (uint64_t)(uintptr_t)d;
    }
    std::fclose(f);
    if (!ok) { g_err = std::string(path) + ": truncated or unreadable packed-weight file"; return CCVPE_EINVAL; }
    // re-base the pointers: every 8-byte aligned word of the descriptor structs that equals an old buffer address
    size_t patched = 0;
    packed_state_io(h, [&](void* p, size_t n) {
        uint64_t* w = reinterpret_cast<uint64_t*>(p);
        for (size_t i = 0; i + 8 <= n; i += 8, ++w) {
            auto it = remap.find(*w);
            if (*w != 0 && it != remap.end()) { *w = it->second; ++patched; }
        }
    });
    if (patched < hd.n_allocs / 2) return fail(CCVPE_EINVAL, "%s: descriptor / buffer table mismatch", path);
    h->host.clear();
    h->fuse_level1 = hd.fuse_level1 != 0 && h->fuse_level1;
    h->finalized = true;
    return 0;
}

int ccvpe_output_channels(ccvpe_handle h, int32_t level) {
    if (!h || level < 0 || level > 5) return fail(CCVPE_EINVAL, "bad level");
    return h->rolls[level];
}

// Per-layer tile selection by measurement: every implicit-GEMM launch of the plan is timed with each
// candidate tile (hipEvents, on the plan's own buffers - timing does not depend on the data) and the
// fastest is kept.  Runs once per (batch, ground size) plan, before its first forward.
static int autotune_plan(ccvpe_handle h, Plan& pl) {
    Ctx c;
    c.arena = h->arena; c.off = &pl.off; c.stream = nullptr;
    c.splitk_scratch = c.ptr(pl.scratch); c.splitk_floats = Plan::SPLITK_FLOATS;
    if (pl.tune_cache.id >= 0) c.cache_out = c.ptr(pl.tune_cache);
    // the candidates run on the null stream inside the shared arena: earlier forwards of this handle may still be in flight
    // on a non-blocking caller stream or on the internal second stream (neither is ordered with the null stream)
    HIPCHK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    HIPCHK(hipMemset(h->arena, 0, pl.total * sizeof(float)));
    const int nt = conv_igemm_num_tiles();
    for (auto& op : pl.ops) {
        if (!op.tile) continue;
        ConvParams q{};
        q.M = op.gemm_m; q.N = op.gemm_n;
        int best = 0;
        float best_ms = 1e30f;
        const int nkt = op.gemm_kpad / 32;
        for (int t = 1; t <= nt; ++t) {
            if (conv_igemm_tile_util(q, t) < 0.45) continue;
            if (conv_igemm_tile_is_bf16x3(t) && (h->cfg.reserved[0] != 1 || getenv("CCVPE_TUNE_NO_BF16X3"))) continue;
            if (const char* only = getenv("CCVPE_TUNE_BF16_ONLY"))   // diagnostic: keep only bf16x3 tiles whose name contains the string
                if (conv_igemm_tile_is_bf16x3(t) && !std::strstr(conv_igemm_tile_name(t), only)) continue;
            if (op.bf16x3_only && !conv_igemm_tile_is_bf16x3(t)) continue;
            if (conv_igemm_tile_is_wino(t) && !op.wino_ok) continue;
            if (conv_igemm_tile_is_wino4(t) && !op.wino4_ok) continue;
            static const bool prefer_pw = getenv("CCVPE_TUNE_PREFER_PW") != nullptr;   // test hook: pointwise tiles wherever they apply
            if (prefer_pw && op.is_pw && !op.bf16x3_only && !conv_igemm_tile_is_pw(t) && op.gemm_kpad <= 512) continue;
            if (conv_igemm_tile_is_pw(t)) {
                ConvParams qq{}; qq.M = 16; qq.N = 1 << 20;
                const int bn = (int)(((long long)qq.N) / conv_igemm_tile_blocks(qq, t));   // the tile's column width
                if (!op.is_pw || op.bf16x3_only || !conv_pw_fits(bn, op.gemm_kpad) || getenv("CCVPE_NO_PW")) continue;
            }
            const long long blocks = conv_igemm_tile_blocks(q, t);
            static const bool no_split = getenv("CCVPE_TUNE_SPLITK") && std::atoi(getenv("CCVPE_TUNE_SPLITK")) == 0;
            // the persistent Winograd grids also try odd split factors: 160 work items on 256 resident workgroups (conv6.0) are
            // 3 rounds of quarter items with split 4 but 2 rounds of thirds with split 3
            static const int SPLITS[] = {1, 255, 2, 3, 4, 5, 6, 8, 12, 16};   // 255: F(4x4) tail split (kernels_wino4.hip); before the rest, whose limits end the loop
            for (int split : SPLITS) {
                if (split > 1 && no_split) break;
                if (split == 255 && !conv_igemm_tile_is_wino4(t)) continue;
                if (split > 1 && (split & (split - 1)) && !conv_igemm_tile_is_wino(t)) continue;
                if (split > 1 && conv_igemm_tile_is_pw(t)) break;   // the pointwise persistent tiles keep K whole
                if (split > 1 && split != 255) {   // split-K only where the grid underfills the chip and K is deep enough
                    // (the persistent Winograd grid also splits when the tile count is an awkward multiple of the
                    // 512 resident workgroups: 640 tiles = 1.25 per workgroup, 4 x 640 quarter-tiles = 5 each)
                    const bool wino = conv_igemm_tile_is_wino(t);
                    if (blocks >= (wino ? 2048 : 512) || blocks * split > (wino ? 8192 : 2048) || nkt < 4 * split) break;
                    if ((size_t)split * op.gemm_m * op.gemm_n > Plan::SPLITK_FLOATS) break;
                }
                const int cfg = t | (split << 8);
                *op.tile = cfg;
                op.fn(c);   // warm-up (also sets the dynamic-LDS attribute on first use)
                if (split == 255 && (conv_igemm_last_tile() >> 8) != 255) continue;   // tail split not applicable to this grid
                float ms = 1e30f;
                for (int trial = 0; trial < 3; ++trial) {   // min of three timed pairs: one noisy sample must not pick the tile
                    HIPCHK(hipEventRecord(e0, nullptr));
                    op.fn(c);
                    op.fn(c);
                    HIPCHK(hipEventRecord(e1, nullptr));
                    HIPCHK(hipEventSynchronize(e1));
                    float t = 0.f;
                    HIPCHK(hipEventElapsedTime(&t, e0, e1));
                    ms = std::min(ms, t);
                }
                if (ms < best_ms) { best_ms = ms; best = cfg; }
            }
        }
        *op.tile = best;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    HIPCHK(hipDeviceSynchronize());   // ... and the forward that follows may be issued on such a stream
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(CCVPE_EHIP, "autotune launch failed: %s", hipGetErrorString(e));
    return 0;
}

static int get_plan(ccvpe_handle h, int B, int gh, int gw, Plan** out, int mode = 0) {
    for (auto& p : h->plans)
        if (p->B == B && p->gh == gh && p->gw == gw && p->mode == mode && (mode == 1 || p->debug == h->debug)) { *out = p.get(); return 0; }
    auto pl = std::make_unique<Plan>();
    int rc = build_plan(h, *pl, B, gh, gw, mode);
    if (rc) return rc;
    if (pl->total > h->arena_floats) {
        // growing the arena is the only synchronising step; it happens on the first call per shape
        HIPCHK(hipDeviceSynchronize());
        if (h->arena) HIPCHK(hipFree(h->arena));
        h->arena = nullptr;
        for (auto& q : h->plans)   // captured graphs point into the old arena
            if (q->exec) { (void)hipGraphExecDestroy(q->exec); q->exec = nullptr; q->runs = 0; }
        void* d = nullptr;
        hipError_t e = hipMalloc(&d, pl->total * sizeof(float));
        if (e != hipSuccess) { h->arena_floats = 0; return fail(CCVPE_ENOMEM, "workspace of %zu bytes: %s", pl->total * sizeof(float), hipGetErrorString(e)); }
        h->arena = (float*)d;
        h->arena_floats = pl->total;
    }
    if (h->autotune) {
        int rc2 = autotune_plan(h, *pl);
        if (rc2) return rc2;
    }
    *out = pl.get();
    h->plans.push_back(std::move(pl));
    return 0;
}

size_t ccvpe_workspace_bytes(ccvpe_handle h, int32_t batch, int32_t grd_h, int32_t grd_w) {
    if (!h || !h->finalized || batch <= 0) { fail(CCVPE_ESTATE, "handle not ready"); return 0; }
    Plan pl;
    const int mb = std::min(batch, h->cfg.micro_batch);
    if (build_plan(h, pl, mb, grd_h, grd_w)) return 0;
    return pl.total * sizeof(float);
}

// Issue a plan's ops: in order on one stream, or on two streams with event edges for the cross-stream dependencies.
static int run_ops(ccvpe_handle h, Plan& pl, const Ctx& base, hipStream_t s0) {
    if (!pl.two_streams || h->serial_issue) {
        Ctx c = base;
        c.stream = s0;
        for (auto& op : pl.ops) op.fn(c);
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
    c[1].splitk_scratch = c[1].ptr(pl.scratch2);
    const size_t n = pl.ops.size();
    HIPCHK(hipEventRecord(pl.events[n], st[0]));            // fork: the second stream starts after the caller's prior work
    HIPCHK(hipStreamWaitEvent(st[1], pl.events[n], 0));
    for (size_t i = 0; i < n; ++i) {
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
    if (!h || !grd || (!sat && !cache) || !out) return fail(CCVPE_EINVAL, "null argument");
    if (cache && batch > h->cfg.micro_batch) return fail(CCVPE_EINVAL, "cached forward needs batch <= micro_batch (%d)", h->cfg.micro_batch);
    if (!h->finalized) return fail(CCVPE_ESTATE, "ccvpe_finalize_weights has not been called");
    if (batch <= 0) return fail(CCVPE_EINVAL, "batch must be positive");
    if (!out->logits_flattened || !out->heatmap || !out->ori) return fail(CCVPE_EINVAL, "null output buffer");
    for (int k = 0; k < 6; ++k) if (!out->matching_score[k]) return fail(CCVPE_EINVAL, "null matching_score[%d]", k);
    HIPCHK(hipSetDevice(h->cfg.device));
    if (profile) h->prof.clear();
    int mbmax = h->cfg.micro_batch;
    {   // never build a plan with a tensor of 2 GiB or more (32-bit byte offsets): larger batches loop
        auto it = h->mb_cap.find({gh, gw});
        if (it == h->mb_cap.end()) {
            const std::string keep = g_err;
            const int cap = ccvpe_max_micro_batch(h->cfg.variant, h->cfg.ori_noise, gh, gw);
            g_err = keep;
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
        c.splitk_scratch = c.ptr(pl->scratch); c.splitk_floats = Plan::SPLITK_FLOATS;
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
            HIPCHK(hipMemcpyAsync((void*)c.grd, ugrd, (size_t)mb * 3 * gh * gw * sizeof(float), hipMemcpyDeviceToDevice, stream));
            HIPCHK(hipMemcpyAsync((void*)c.sat, usat, (size_t)mb * 3 * CCVPE_SAT_HW * CCVPE_SAT_HW * sizeof(float), hipMemcpyDeviceToDevice, stream));
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
            HIPCHK(hipMemcpyAsync(user.logits_flattened, c.out.logits_flattened, (size_t)mb * npx * sizeof(float), hipMemcpyDeviceToDevice, stream));
            HIPCHK(hipMemcpyAsync(user.heatmap, c.out.heatmap, (size_t)mb * npx * sizeof(float), hipMemcpyDeviceToDevice, stream));
            HIPCHK(hipMemcpyAsync(user.ori, c.out.ori, (size_t)mb * 2 * npx * sizeof(float), hipMemcpyDeviceToDevice, stream));
            for (int k = 0; k < 6; ++k) {
                const size_t hw = (size_t)(8 << k) * (8 << k);
                HIPCHK(hipMemcpyAsync(user.matching_score[k], c.out.matching_score[k], (size_t)mb * h->rolls[k] * hw * sizeof(float), hipMemcpyDeviceToDevice, stream));
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
        if (c.conv_errors) return fail(CCVPE_EINVAL, "%d convolution launches were refused (unsupported geometry)", c.conv_errors);
        done += mb;
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(CCVPE_EHIP, "kernel launch failed: %s", hipGetErrorString(e));
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
    if (!h || i < 0 || i >= (int)h->prof.size()) return fail(CCVPE_EINVAL, "row out of range");
    const auto& r = h->prof[i];
    if (name_buf && name_cap) { std::strncpy(name_buf, r.name.c_str(), name_cap - 1); name_buf[name_cap - 1] = 0; }
    if (ms) *ms = r.ms;
    if (flops) *flops = r.flops;
    if (bytes) *bytes = r.bytes;
    return 0;
}

int ccvpe_profile_row_issued(ccvpe_handle h, int32_t i, double* issued_flops) {
    if (!h || i < 0 || i >= (int)h->prof.size() || !issued_flops) return fail(CCVPE_EINVAL, "row out of range");
    *issued_flops = h->prof[i].issued;
    return 0;
}

int ccvpe_postprocess(ccvpe_handle h, const float* heatmap, const float* ori, int32_t batch, ccvpe_pose* poses, void* stream) {
    if (!h || !heatmap || !ori || !poses || batch <= 0) return fail(CCVPE_EINVAL, "bad argument");
    HIPCHK(hipSetDevice(h->cfg.device));
    static_assert(sizeof(ccvpe_pose) == sizeof(PoseOut), "pose layout");
    launch_postprocess(heatmap, ori, batch, CCVPE_OUT_HW * CCVPE_OUT_HW, reinterpret_cast<PoseOut*>(poses), (hipStream_t)stream);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(CCVPE_EHIP, "postprocess launch failed: %s", hipGetErrorString(e));
    return 0;
}

int ccvpe_eval_metrics(ccvpe_handle h, const ccvpe_pose* poses, const float* heatmap, int32_t batch, const int32_t* gt_index,
                       const float* gt_cos_sin, const double* meter_per_pixel, const double* heading_deg, ccvpe_metrics* out, void* stream) {
    if (!h || !poses || !heatmap || !gt_index || !meter_per_pixel || !out || batch <= 0) return fail(CCVPE_EINVAL, "bad argument");
    HIPCHK(hipSetDevice(h->cfg.device));
    static_assert(sizeof(ccvpe_metrics) == sizeof(MetricsOut), "metrics layout");
    launch_metrics(reinterpret_cast<const PoseOut*>(poses), heatmap, batch, CCVPE_OUT_HW, CCVPE_OUT_HW * CCVPE_OUT_HW, gt_index, gt_cos_sin,
                   meter_per_pixel, heading_deg, reinterpret_cast<MetricsOut*>(out), (hipStream_t)stream);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(CCVPE_EHIP, "metrics launch failed: %s", hipGetErrorString(e));
    return 0;
}

int ccvpe_set_streams(ccvpe_handle h, int32_t n_streams) {
    if (!h) return fail(CCVPE_EINVAL, "null handle");
    if (n_streams != 1 && n_streams != 2) return fail(CCVPE_EINVAL, "n_streams must be 1 or 2");
    const bool serial = n_streams == 1;
    if (serial != h->serial_issue)   // captured graphs embed the issue order: drop them
        for (auto& q : h->plans)
            if (q->exec) { (void)hipGraphExecDestroy(q->exec); q->exec = nullptr; q->runs = 0; }
    h->serial_issue = serial;
    return 0;
}

int ccvpe_set_debug(ccvpe_handle h, int32_t enable) {
    if (!h) return fail(CCVPE_EINVAL, "null handle");
    h->debug = enable != 0;
    return 0;
}

int ccvpe_read_tap(ccvpe_handle h, const char* name, float* host_dst, size_t capacity, size_t* n_out, int32_t shape_out[4]) {
    if (!h || !name || !host_dst) return fail(CCVPE_EINVAL, "null argument");
    HIPCHK(hipSetDevice(h->cfg.device));
    Plan* pl = nullptr;
    for (auto it = h->plans.rbegin(); it != h->plans.rend(); ++it)
        if ((*it)->debug) { pl = it->get(); break; }
    if (!pl) return fail(CCVPE_ESTATE, "no debug plan: call ccvpe_set_debug(h, 1) before forward");
    auto it = pl->taps.find(name);
    if (it == pl->taps.end()) return fail(CCVPE_EKEY, "unknown tap '%s'", name);
    const TapInfo& ti = it->second;
    const Tensor& t = ti.t;
    float* base = h->arena + pl->off[t.id];
    HIPCHK(hipDeviceSynchronize());
    if (ti.C < 0) {   // already NCHW
        const size_t n = (size_t)t.B * t.H * t.W * t.C;
        if (n > capacity) return fail(CCVPE_EINVAL, "tap needs %zu floats", n);
        HIPCHK(hipMemcpy(host_dst, base, n * sizeof(float), hipMemcpyDeviceToHost));
        if (n_out) *n_out = n;
        if (shape_out) { shape_out[0] = t.B; shape_out[1] = t.H; shape_out[2] = t.W; shape_out[3] = t.C; }
        return 0;
    }
    const int hw = t.H * t.W;
    const size_t n = (size_t)t.B * ti.C * hw;
    if (n > capacity) return fail(CCVPE_EINVAL, "tap needs %zu floats", n);
    float* tmp = nullptr;
    HIPCHK(hipMalloc((void**)&tmp, n * sizeof(float)));
    launch_nhwc_to_nchw(base, t.C, ti.coff, ti.C, t.B, hw, tmp, nullptr);
    hipError_t e = hipMemcpy(host_dst, tmp, n * sizeof(float), hipMemcpyDeviceToHost);
    (void)hipFree(tmp);
    if (e != hipSuccess) return fail(CCVPE_EHIP, "tap copy failed: %s", hipGetErrorString(e));
    if (n_out) *n_out = n;
    if (shape_out) { shape_out[0] = t.B; shape_out[1] = ti.C; shape_out[2] = t.H; shape_out[3] = t.W; }
    return 0;
}

size_t ccvpe_aerial_cache_bytes(ccvpe_handle h, int32_t batch) {
    if (!h || batch <= 0) { fail(CCVPE_EINVAL, "bad argument"); return 0; }
    size_t off[6];
    return cache_layout(h->vs, batch, off) * sizeof(float);
}

int ccvpe_encode_aerial(ccvpe_handle h, const float* sat, int32_t batch, void* cache, void* stream) {
    if (!h || !sat || !cache || batch <= 0) return fail(CCVPE_EINVAL, "bad argument");
    if (!h->finalized) return fail(CCVPE_ESTATE, "ccvpe_finalize_weights has not been called");
    if (batch > h->cfg.micro_batch) return fail(CCVPE_EINVAL, "aerial encode needs batch <= micro_batch (%d)", h->cfg.micro_batch);
    HIPCHK(hipSetDevice(h->cfg.device));
    Plan* pl; int rc = get_plan(h, batch, 0, 0, &pl, 1);
    if (rc) return rc;
    Ctx c;
    c.arena = h->arena; c.off = &pl->off; c.stream = (hipStream_t)stream;
    c.splitk_scratch = c.ptr(pl->scratch); c.splitk_floats = Plan::SPLITK_FLOATS;
    c.sat = sat; c.cache_out = (float*)cache;
    for (auto& op : pl->ops) op.fn(c);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(CCVPE_EHIP, "kernel launch failed: %s", hipGetErrorString(e));
    return 0;
}

int ccvpe_forward_cached(ccvpe_handle h, const float* grd, int32_t grd_h, int32_t grd_w, const void* cache, int32_t batch,
                         const ccvpe_outputs* out, void* stream) {
    if (!cache) return fail(CCVPE_EINVAL, "null cache");
    return run_forward(h, grd, grd_h, grd_w, nullptr, batch, out, (hipStream_t)stream, false, (const float*)cache);
}

int ccvpe_preprocess(const uint8_t* hwc, int32_t batch, int32_t H, int32_t W, const int32_t* shift, int32_t crop_w,
                     const float mean[3], const float stdv[3], float* out_nchw, void* stream) {
    if (!hwc || !out_nchw || !mean || !stdv) return fail(CCVPE_EINVAL, "null argument");
    if (batch <= 0 || H <= 0 || W <= 0 || crop_w <= 0 || crop_w > W) return fail(CCVPE_EINVAL, "bad geometry");
    PreprocParams p{};
    p.in = hwc; p.B = batch; p.H = H; p.W = W; p.crop_w = crop_w; p.shift = shift; p.out = out_nchw;
    for (int c = 0; c < 3; ++c) { p.mean[c] = mean[c]; p.stdv[c] = stdv[c]; }
    launch_preprocess(p, (hipStream_t)stream);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(CCVPE_EHIP, "preprocess launch failed: %s", hipGetErrorString(e));
    return 0;
}


int ccvpe_preprocess_resize(const uint8_t* hwc, int32_t batch, int32_t in_h, int32_t in_w, int32_t out_h, int32_t out_w,
                            const int32_t* shift, int32_t crop_w, const float mean[3], const float stdv[3], uint8_t* scratch,
                            float* out_nchw, void* stream) {
    if (!hwc || !out_nchw || !mean || !stdv) return fail(CCVPE_EINVAL, "null argument");
    if (batch <= 0 || in_h <= 0 || in_w <= 0 || out_h <= 0 || out_w <= 0 || crop_w <= 0 || crop_w > out_w) return fail(CCVPE_EINVAL, "bad geometry");
    if (in_w != out_w && !scratch) return fail(CCVPE_EINVAL, "scratch of batch*in_h*out_w*3 bytes is required when the width changes");
    if ((double)batch * in_h * std::max(in_w, out_w) * 3 >= 2147483647.0 * 2) return fail(CCVPE_EINVAL, "image batch too large");
    ResizeParams p{};
    p.in = hwc; p.B = batch; p.IH = in_h; p.IW = in_w; p.OH = out_h; p.OW = out_w; p.crop_w = crop_w; p.tmp = scratch; p.shift = shift; p.out = out_nchw;
    for (int c = 0; c < 3; ++c) { p.mean[c] = mean[c]; p.stdv[c] = stdv[c]; }
    if (launch_resize(p, (hipStream_t)stream) != 0) return fail(CCVPE_EINVAL, "down-scaling factors above 8 are not supported (%dx%d -> %dx%d)", in_h, in_w, out_h, out_w);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(CCVPE_EHIP, "resize launch failed: %s", hipGetErrorString(e));
    return 0;
}

int ccvpe_debug_dump_plan(ccvpe_handle h, const char* path) {
    if (!h || !path) return fail(CCVPE_EINVAL, "null argument");
    Plan* pl = h->last_plan;
    if (!pl) return fail(CCVPE_ESTATE, "no forward has run on this handle");
    HIPCHK(hipSetDevice(h->cfg.device));
    HIPCHK(hipDeviceSynchronize());
    const size_t n = pl->size.size();
    unsigned long long* d = nullptr;
    HIPCHK(hipMalloc((void**)&d, n * sizeof(unsigned long long)));
    HIPCHK(hipMemset(d, 0, n * sizeof(unsigned long long)));
    for (size_t id = 0; id < n; ++id) {
        if (pl->size[id] == 0) continue;
        const int blocks = (int)std::min<size_t>((pl->size[id] + 255) / 256, 2048);
        hipLaunchKernelGGL(checksum_kernel, dim3(blocks), dim3(256), 0, nullptr, reinterpret_cast<const uint32_t*>(h->arena + pl->off[id]), pl->size[id], d + id);
    }
    std::vector<unsigned long long> sums(n);
    hipError_t e = hipMemcpy(sums.data(), d, n * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(CCVPE_EHIP, "checksum copy failed: %s", hipGetErrorString(e));
    FILE* f = std::fopen(path, "w");
    if (!f) return fail(CCVPE_EINVAL, "cannot open %s", path);
    std::fprintf(f, "# plan B=%d grd=%dx%d mode=%d two_streams=%d tensors=%zu arena_floats=%zu\n", pl->B, pl->gh, pl->gw, pl->mode, (int)pl->two_streams, n, pl->total);
    for (size_t i = 0; i < pl->ops.size(); ++i) {
        const Op& op = pl->ops[i];
        std::fprintf(f, "op %zu %s stream=%d wait=%d signal=%d tile=%s", i, op.name.c_str(), op.stream, op.wait_on.empty() ? -1 : op.wait_on[0], (int)op.signal,
                     op.tile ? conv_igemm_tile_name(*op.tile & 0xff) : "-");
        if (op.tile && (*op.tile >> 8) == 255) std::fprintf(f, "_tailsplit");
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
                    hipLaunchKernelGGL(checksum_kernel, dim3(blocks), dim3(256), 0, nullptr, reinterpret_cast<const uint32_t*>(h->snap[which] + e.second), pl->size[e.first], dd);
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
    if (!in || !w || !out) return fail(CCVPE_EINVAL, "null argument");
    if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cin % 8 || Cout <= 0 || KH <= 0 || KW <= 0 || stride <= 0 || pad < 0)
        return fail(CCVPE_EINVAL, "bad conv geometry (Cin must be a multiple of 8)");
    const int OH = (H + 2 * pad - KH) / stride + 1, OW = (W + 2 * pad - KW) / stride + 1;
    if (OH <= 0 || OW <= 0) return fail(CCVPE_EINVAL, "empty output");
    if ((double)B * H * W * Cin >= 2147483647.0 || (double)B * OH * OW * Cout >= 2147483647.0)
        return fail(CCVPE_EINVAL, "tensor exceeds 2^31 elements");
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
    if (((tile >> 8) & 0xff) > 1) {   // tile word = id | (split-K << 8): give the launch a slab
        const int sk = (tile >> 8) & 0xff;
        const size_t fl = (size_t)(sk == 255 ? 8 : sk) * p.M * p.N;   // 255 = F(4x4) tail split: its slab is a fraction of 8 full ones
        void* d = nullptr;
        if (hipMalloc(&d, fl * sizeof(float)) != hipSuccess) { cleanup(); return fail(CCVPE_ENOMEM, "split-K slab"); }
        tmp.dev_allocs.push_back(d);
        tmp.dev_alloc_bytes.push_back(fl * sizeof(float));
        p.partial = (float*)d; p.partial_floats = fl;
    }
    if (conv_igemm_tile_is_wino(tile) && !conv_wino_tile_supported(p, tile)) { cleanup(); return fail(CCVPE_EINVAL, "layer is not Winograd-shaped (3x3, stride 1, pad 1, W %% 16 == 0, H %% 16 == 0, output channels a multiple of 4; F(4x4): >= 40 of them)"); }
    if (launch_conv_igemm(p, tile, st) != 0) { cleanup(); return fail(CCVPE_EINVAL, "unsupported conv geometry (KH*KW <= 16, Cin %% 8 == 0)"); }
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
    if (e != hipSuccess) return fail(CCVPE_EHIP, "conv launch failed: %s", hipGetErrorString(e));
    if (e2 != hipSuccess) return fail(CCVPE_EHIP, "conv execution failed: %s", hipGetErrorString(e2));
    return 0;
}

}  // extern "C"
