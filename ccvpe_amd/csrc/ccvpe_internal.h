// Internal declarations shared by the translation units of the host side of libccvpe_hip.so:
//   ccvpe_weights.hip  state_dict ingestion (BN folding, weight packing, Winograd transforms), packed-weight cache
//   ccvpe_plan.hip     the execution plan: CVM_*.forward restated as a static launch list over NHWC tensors
//   ccvpe_tune.hip     per-layer tile selection by measurement, the persistent tuning table, plan lookup
//   ccvpe_api.hip      the C ABI (include/ccvpe.h): handle lifetime, forward orchestration, diagnostics
#pragma once
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
// errors (defined in ccvpe_api.hip)
// ------------------------------------------------------------------------------------------------
std::string& ccvpe_err();                       // thread-local message behind ccvpe_last_error()
int ccvpe_fail(int code, const char* fmt, ...);
#define HIPCHK(expr)                                                                           \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess) return ccvpe_fail(CCVPE_EHIP, "%s failed: %s", #expr, hipGetErrorString(_e)); \
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
// Width the transposed conv of a decoder level occupies in its concat buffer (round 4).  Channel-slice writes into the buffers are slow
// when a pixel's piece is not a whole number of 64-byte lines (tools/ubench_strided_write.py: 40 of 56 channels 3.4 TB/s against 5.3
// dense): the localisation decoder's level 2 (40 + 16 channels) is therefore laid out as [40 | 8 zero | 16] - 64 channels, pieces of
// 192 and 64 bytes; the transposed conv has eight zero output columns (it writes the zeros), conv2.0 eight zero input columns.
// CCVPE_PAD_CONCAT=0: the reference's widths.  Every other level of every variant already ends on a line.
static int deconv_width(const DecLevel& l) {
    static const bool pad = !(getenv("CCVPE_PAD_CONCAT") && std::atoi(getenv("CCVPE_PAD_CONCAT")) == 0);
    return (pad && l.skip > 0 && (l.dout * 4) % 64 != 0) ? round_up(l.dout, 16) : l.dout;
}

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
    float* wino4x = nullptr;      // xi-split F(4x4,3x3) weights (kernels_wino4x.hip): layers of 24 .. 128 output channels
    size_t wino4x_bytes = 0;
    int wino4x_cfg = -1;
    float* proj = nullptr;        // fragment-order copy of a gated deep-K 1x1 layer for kernels_proj.hip
    size_t proj_bytes = 0;
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
    float* wino_v = nullptr;           // pre-transformed input of the split Winograd form (kernels_wino4p.hip), per stream
    size_t wino_v_floats = 0;
    unsigned* tickets = nullptr;       // the plan's last-arriver counters (ticket.h); every launch that draws tickets owns a range
    const float* cache_in = nullptr;   // aerial cache consumed by a "cached" plan
    float* cache_out = nullptr;        // aerial cache produced by an "encode" plan
    float* ptr(const Tensor& t) const { return arena + (*off)[t.id]; }
    Dst dst(const Tensor& t, int coff = 0) const { return Dst{ptr(t), t.C, coff, t.split ? 1 : 0, t.numel()}; }
    mutable int conv_errors = 0;   // launches refused by launch_conv_igemm (unsupported geometry)
    mutable size_t conv_tick_off = 0;   // ticket range of the convolution launch being issued (set by Plan::add_conv's wrapper)
    void launch_conv(ConvParams& p, int cfg) const {
        p.tickets = tickets ? tickets + conv_tick_off : nullptr;
        p.partial = splitk_scratch;
        p.partial_floats = splitk_floats;
        p.wino4_v = wino_v;
        p.wino4_v_floats = wino_v_floats;
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
    bool wino4x_ok = false;       // ... with the xi-split F(4x4,3x3) weights packed
    bool is_pw = false;           // 1x1 conv / k2s2 transposed conv: the pointwise persistent tiles may serve it
    bool proj_ok = false;         // gated project conv with the fragment-order weights packed (kernels_proj.hip)
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
    // last-arriver tickets (ticket.h): one device allocation per plan, outside the arena (the autotuner fills the arena with random
    // numbers, and plans share it), zeroed once when the plan is built - every kernel leaves its counters at zero
    size_t ticket_words = 0;
    unsigned* tickets = nullptr;
    size_t alloc_tickets(size_t n) { const size_t o = ticket_words; ticket_words += (n + 15) & ~(size_t)15; return o; }
    bool tiles_checked = false;   // the first issue compared every tiled launch with its plan entry (ccvpe_api.hip: check_issued_tile)
    // Two-stream execution: the aerial encoder and the orientation decoder are issued on a second stream, so the
    // ramp-up / drain of the ~330 short kernels of one chain is filled by the other chain.  Dependencies come from
    // the ops' tensor lists (any two ops that touch the same tensor stay ordered), and a two-stream plan gives every
    // tensor its own memory (lifetime-based reuse would add hidden dependencies between the streams).
    bool two_streams = false;
    Tensor scratch2;              // split-K slab scratch of the second stream
    Tensor vscratch, vscratch2;   // V = B^T d B of the split Winograd F(4x4) form, one per stream (whole-plan lifetime)
    std::vector<hipEvent_t> events;   // one per signalling op + fork + join, created on first use
    std::vector<int> issue_order;     // two-stream plans: the order the launches are handed to the two streams (schedule())
    ~Plan() {
        if (exec) (void)hipGraphExecDestroy(exec);
        for (hipEvent_t e : events) if (e) (void)hipEventDestroy(e);
        if (tickets) (void)hipFree(tickets);
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
        // Issue order (round 4).  In plan order every launch of the ground encoder is issued before the first launch of the aerial encoder:
        // the host (eagerly: ~3-4 us per launch; a replayed hipGraph submits its nodes in creation order just the same) feeds the second
        // stream only after ~60 launches of the first, and at batch 1 - where the GPU finishes a launch in 5-20 us - the aerial chain, the
        // one the decoders wait for, started ~450 us late (rocprofv3 timeline of a replayed frame).  The launches are therefore issued
        // interleaved: always the stream that is behind (by the number of launches issued), never ahead of a cross-stream dependency.
        // Any interleaving that keeps each stream's own order and issues a producer before its cross-stream consumer is valid: the
        // streams' memory pools are recycled in stream order, cross-stream tensors keep private memory (assign()).
        issue_order.clear();
        {
            std::vector<int> q[2];
            for (int i = 0; i < (int)ops.size(); ++i) q[ops[i].stream].push_back(i);
            size_t p[2] = {0, 0};
            std::vector<bool> issued(ops.size(), false);
            static const bool plan_order = getenv("CCVPE_ISSUE_ORDER") && std::atoi(getenv("CCVPE_ISSUE_ORDER")) == 0;   // A/B switch: plan order
            while (!plan_order && (p[0] < q[0].size() || p[1] < q[1].size())) {
                auto ready = [&](int s) {
                    if (p[s] >= q[s].size()) return false;
                    for (int d : ops[q[s][p[s]]].wait_on) if (!issued[d]) return false;
                    return true;
                };
                const bool r0 = ready(0), r1 = ready(1);
                int s;
                if (r0 && r1) s = (p[1] * q[0].size() < p[0] * q[1].size()) ? 1 : 0;   // the stream that has issued the smaller share of its launches
                else if (r0 || r1) s = r0 ? 0 : 1;
                else { issue_order.clear(); break; }                                        // (cannot happen: the dependency graph is acyclic)
                const int i = q[s][p[s]++];
                issued[i] = true;
                issue_order.push_back(i);
            }
            if (issue_order.size() != ops.size()) { issue_order.resize(ops.size()); for (int i = 0; i < (int)ops.size(); ++i) issue_order[i] = i; }
        }
        if (getenv("CCVPE_LOG_SCHEDULE"))
            for (int i = 0; i < (int)ops.size(); ++i) {
                std::fprintf(stderr, "op %3d s%d %-28s wait=%d uses=", i, ops[i].stream, ops[i].name.c_str(), ops[i].wait_on.empty() ? -1 : ops[i].wait_on[0]);
                for (int id : ops[i].uses) std::fprintf(stderr, "%d ", id);
                std::fprintf(stderr, "\n");
            }
    }
    static constexpr size_t SPLITK_FLOATS = 32u << 20;   // 128 MiB: 16 slabs of M*N <= 2M outputs
    static constexpr size_t WINO_V_FLOATS = 64u << 20;   // 256 MiB: upper bound of a V scratch (kernels_wino4p.hip)
    void set_scratch(Ctx& c, int stream) const {         // per-stream scratch pointers of a launch context
        const Tensor& sk = stream ? scratch2 : scratch;
        const Tensor& vv = stream ? vscratch2 : vscratch;
        c.splitk_scratch = c.ptr(sk); c.splitk_floats = SPLITK_FLOATS;
        c.wino_v = vv.id >= 0 ? c.ptr(vv) : nullptr; c.wino_v_floats = vv.id >= 0 ? (size_t)vv.C : 0;
    }

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
        const size_t toff = alloc_tickets(CONV_TICKETS);   // counters of a self-reducing split-K launch (ticket.h)
        add(name, std::move(uses), [fn, tp, toff](const Ctx& c) { c.conv_tick_off = toff; fn(c, *tp); }, flops, bytes);
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
        pin(scratch); pin(scratch2); pin(vscratch); pin(vscratch2); pin(tune_cache);
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
    // CCVPE_WINOGRAD=0 keeps the decoder 3x3 layers on the implicit GEMM.  The Winograd kernels serve fp32 plans only: bf16x3 plans keep the
    // decoder tensors as split bf16 planes, which only the bf16x3 tiles read
    bool wino = true;
    int graph_mode = 0;           // 1: plans replay a captured hipGraph (CCVPE_GRAPH=1; opt-in since round 4, see build_plan), 0 eager launches
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
    // tuning table (ccvpe_tune.hip): launch key -> (tile name, split-K code); launches found here are not measured again
    std::map<std::string, std::pair<std::string, int>> tuning;
    int tuned_plans = 0;          // plans this handle has (partly) tuned by measurement (ccvpe_tuning_generation)
    bool tuning_lookup = true;    // false while a CCVPE_TUNE_* / CCVPE_NO_PW diagnostic switch is set
    float* arena = nullptr;
    size_t arena_floats = 0;
    void* post_scratch = nullptr;   // launch_postprocess: partial pairs and ticket counters for post_batch samples
    int post_batch = 0;
    // profiling rows of the last ccvpe_profile_forward
    struct Row { std::string name; float ms; double flops, bytes, issued; };
    std::vector<Row> prof;
};


// ---- ccvpe_weights.hip ----
void build_expect(ccvpe_handle_s* h);
int pack_conv(ccvpe_handle_s* h, PackedConv& pc, int N, int taps, int cin, int cinp, const std::vector<int>& cmap,
              const std::function<float(int, int, int)>& get, const std::vector<float>& bias, int KH, int KW);
std::vector<int> identity_map(int n);
static inline int score_pad(int nscore) { return round_up(nscore, 8); }

// ---- ccvpe_plan.hip ----
ConvParams conv_params(const PackedConv& pc, const float* in, int in_ld, int B, int H, int W, int OH, int OW,
                       int stride, int pad_t, int pad_l, int act);
size_t cache_layout(const VariantSpec& vs, int B, size_t off[6]);
int build_plan(ccvpe_handle_s* h, Plan& pl, int B, int gh, int gw, int mode = 0);

// ---- ccvpe_tune.hip ----
int autotune_plan(ccvpe_handle_s* h, Plan& pl, const std::vector<bool>* known = nullptr);
int get_plan(ccvpe_handle_s* h, int B, int gh, int gw, Plan** out, int mode = 0);
std::string tuning_key(ccvpe_handle_s* h, const Plan& pl, const Op& op);
