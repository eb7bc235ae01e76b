// Per-layer tile selection of libccvpe_hip.so: every tiled launch of a plan is timed with each candidate (tile, split-K)
// on the plan's own buffers; the choices are kept in a tuning table that can be exported / imported as text, so a later
// process (or another rank) runs exactly the same launches without measuring again.  No reference counterpart.
#include "ccvpe_internal.h"


// ---- tuning table -----------------------------------------------------------------------------------------------
// One entry per tiled launch, keyed by everything that shapes it: variant, roll count, padding mode, precision, batch, launch
// name, GEMM shape and which kernel families may serve it - so the decoder launches of a full forward, of a cached-aerial
// forward and of a debug plan share their entries, and a switch that changes a layer's shape or eligibility gives another
// key.  Tiles are stored by NAME: a table survives library builds that renumber or add tiles.
// Keys are at most TUNING_KEY_MAX characters (the text form reads them with a bounded %s); a longer one - an op name nobody
// has written yet - yields the empty key, which is never stored or looked up: such a launch is measured in every process.
static constexpr size_t TUNING_KEY_MAX = 255;
std::string tuning_key(ccvpe_handle_s* h, const Plan& pl, const Op& op) {
    char head[64], tail[64];
    std::snprintf(head, sizeof(head), "v%d_r%d_c%d_p%d_b%d|", h->cfg.variant, h->rolls[1], h->cfg.circular_padding, h->cfg.reserved[0], pl.B);
    std::snprintf(tail, sizeof(tail), "|%dx%dx%d|%d%d%d%d%s%s", op.gemm_m, op.gemm_n, op.gemm_kpad, op.wino_ok ? 1 : 0, op.wino4_ok ? 1 : 0,
                  op.is_pw ? 1 : 0, op.bf16x3_only ? 1 : 0, op.wino4x_ok ? "x" : "", op.proj_ok ? "p" : "");
    std::string key = std::string(head) + op.name + tail;
    if (key.size() > TUNING_KEY_MAX || key.find_first_of(" \t\n") != std::string::npos) return std::string();
    return key;
}

static int tile_by_name(const std::string& name) {
    if (name == "auto") return 0;
    for (int t = 1; t <= conv_igemm_num_tiles(); ++t)
        if (name == conv_igemm_tile_name(t)) return t;
    return -1;
}

// Applies the table to the plan's launches; returns how many launches it does not cover (those keep TILE_AUTO and are measured).
static int apply_tuning(ccvpe_handle_s* h, Plan& pl, std::vector<bool>& known) {
    int missing = 0;
    known.assign(pl.ops.size(), false);
    for (size_t i = 0; i < pl.ops.size(); ++i) {
        Op& op = pl.ops[i];
        if (!op.tile) continue;
        int t = -1, split = 0;
        if (h->tuning_lookup) {
            const std::string key = tuning_key(h, pl, op);
            auto e = key.empty() ? h->tuning.end() : h->tuning.find(key);
            if (e != h->tuning.end()) { t = tile_by_name(e->second.first); split = e->second.second; }
        }
        if (t < 0) { ++missing; continue; }
        *op.tile = t | (split << 8);
        known[i] = true;
    }
    return missing;
}

static void record_tuning(ccvpe_handle_s* h, const Plan& pl) {
    for (const auto& op : pl.ops) {
        if (!op.tile) continue;
        const int cfg = *op.tile;
        const std::string key = tuning_key(h, pl, op);
        if (!key.empty()) h->tuning[key] = {(cfg & 0xff) ? conv_igemm_tile_name(cfg & 0xff) : "auto", (cfg >> 8) & 0xff};
    }
}

extern "C" {

/* text form: one "op <key> <tile name> <split code>" line per launch; '#' lines are comments.  All or nothing: the text is parsed
   into a temporary table that replaces / extends the handle's entries only when every line was understood, so a failing call leaves
   the handle exactly as it was.  A tile name this build does not know is kept (a table survives builds that add or drop tiles): the
   launch it names is then measured like an unknown one. */
int ccvpe_import_tuning(ccvpe_handle h, const char* text) {
    if (!h || !text) return ccvpe_fail(CCVPE_EINVAL, "null argument");
    std::map<std::string, std::pair<std::string, int>> parsed;
    const char* p = text;
    while (*p) {
        const char* e = std::strchr(p, '\n');
        const std::string line(p, e ? (size_t)(e - p) : std::strlen(p));
        p = e ? e + 1 : p + line.size();
        char a[TUNING_KEY_MAX + 2], b[66];
        int split = 0;
        // one more character than the limits allow is read, so an over-long field is seen (and refused) instead of being cut
        if (std::sscanf(line.c_str(), "op %256s %65s %d", a, b, &split) == 3 && std::strlen(a) <= TUNING_KEY_MAX && std::strlen(b) <= 64 && split >= 0 && split <= 255)
            parsed[a] = {b, split};
        else if (!line.empty() && line[0] != '#') return ccvpe_fail(CCVPE_EINVAL, "tuning table: cannot parse '%.200s'", line.c_str());
    }
    for (auto& kv : parsed) h->tuning[kv.first] = kv.second;
    return (int)parsed.size();
}

int ccvpe_export_tuning(ccvpe_handle h, char* buf, size_t capacity, size_t* needed) {
    if (!h) return ccvpe_fail(CCVPE_EINVAL, "null handle");
    std::string out;
    for (const auto& op : h->tuning) out += "op " + op.first + " " + op.second.first + " " + std::to_string(op.second.second) + "\n";
    if (needed) *needed = out.size() + 1;
    if (!buf || capacity < out.size() + 1) return buf ? ccvpe_fail(CCVPE_EINVAL, "tuning table needs %zu bytes", out.size() + 1) : 0;
    std::memcpy(buf, out.c_str(), out.size() + 1);
    return 0;
}

int ccvpe_tuning_generation(ccvpe_handle h) { return h ? h->tuned_plans : ccvpe_fail(CCVPE_EINVAL, "null handle"); }

}  // extern "C"

// Per-layer tile selection by measurement: every tiled launch of the plan is timed with each candidate tile (hipEvents) on the
// plan's own buffers, filled with unit-variance random data - on zeros the chip holds a ~19 % higher clock and pipe-bound
// candidates are mis-ranked against memory-bound ones (MI355X_MICROARCH.md, DVFS) - and the fastest is kept.  Runs once per
// plan that the tuning table does not know, before its first forward.
int autotune_plan(ccvpe_handle_s* h, Plan& pl, const std::vector<bool>* known) {
    Ctx c;
    c.arena = h->arena; c.off = &pl.off; c.stream = nullptr;
    c.tickets = pl.tickets;
    pl.set_scratch(c, 0);
    if (pl.tune_cache.id >= 0) c.cache_out = c.ptr(pl.tune_cache);
    // the candidates run on the null stream inside the shared arena: earlier forwards of this handle may still be in flight
    // on a non-blocking caller stream or on the internal second stream (neither is ordered with the null stream)
    HIPCHK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    launch_fill_random(h->arena, pl.total, 0x9e3779b9u, nullptr);
    const int nt = conv_igemm_num_tiles();
    for (size_t oi = 0; oi < pl.ops.size(); ++oi) {
        Op& op = pl.ops[oi];
        if (!op.tile || (known && (*known)[oi])) continue;   // launches the tuning table covers are not measured
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
            if (conv_igemm_tile_is_wino4x(t)) { if (!op.wino4x_ok || conv_igemm_tile_wino4x_cfg(t) != conv_wino4x_config(op.gemm_n)) continue; }
            else if (conv_igemm_tile_is_wino4(t) && !op.wino4_ok) continue;
            static const bool prefer_pw = getenv("CCVPE_TUNE_PREFER_PW") != nullptr;   // test hook: pointwise tiles wherever they apply
            if (prefer_pw && op.is_pw && !op.bf16x3_only && !conv_igemm_tile_is_pw(t) && op.gemm_kpad <= 512) continue;
            static const bool prefer_proj = getenv("CCVPE_TUNE_PREFER_PROJ") != nullptr;   // test hook: the deep-K project GEMM wherever it applies
            if (prefer_proj && op.proj_ok && !conv_igemm_tile_is_proj(t)) continue;
            static const bool prefer_lat = prefer_proj && std::strcmp(getenv("CCVPE_TUNE_PREFER_PROJ"), "lat") == 0;   // ... its latency form only
            if (prefer_lat && op.proj_ok && op.gemm_m <= 4096 && conv_igemm_tile_proj_rt(t) < 100) continue;
            if (conv_igemm_tile_is_proj(t)) {
                const int rt = conv_igemm_tile_proj_rt(t);   // row tiles per workgroup; >= 100: the latency form (small M only)
                if (!op.proj_ok || !conv_proj_has(rt, op.gemm_n) || getenv("CCVPE_NO_PW")) continue;
                if (rt >= 100 && (op.gemm_m > 4096 || op.gemm_kpad > 10240)) continue;   // (conv_proj_supported has the exact rule)
                // the multi-row forms (conv_projl_r2 / r4) win the level-6 transposed convs by 1 us when timed alone (17.7 against 18.8 us) and lose
                // in the frame, where the two decoders run that layer at the same time: 2 x 256 sixteen-wave workgroups, one per CU - 31.7 us
                // each in the traced frame against ~24 for the four-wave implicit GEMM.  CCVPE_TUNE_LAT_ROWS=1 times them all the same.
                static const bool lat_rows = getenv("CCVPE_TUNE_LAT_ROWS") != nullptr;
                if (rt > 104 && (!lat_rows || op.gemm_m > 1024)) continue;
            } else if (conv_igemm_tile_is_pw(t)) {
                ConvParams qq{}; qq.M = 16; qq.N = 1 << 20;
                const int bn = (int)(((long long)qq.N) / conv_igemm_tile_blocks(qq, t));   // the tile's column width
                if (!op.is_pw || op.bf16x3_only || !conv_pw_fits(bn, op.gemm_kpad) || getenv("CCVPE_NO_PW")) continue;
            }
            const long long blocks = conv_igemm_tile_blocks(q, t);
            static const bool no_split = getenv("CCVPE_TUNE_SPLITK") && std::atoi(getenv("CCVPE_TUNE_SPLITK")) == 0;
            // the persistent Winograd grids also try odd split factors: 160 work items on 256 resident workgroups (conv6.0) are
            // 3 rounds of quarter items with split 4 but 2 rounds of thirds with split 3
            static const int SPLITS[] = {1, 255, 2, 3, 4, 5, 6, 8, 12, 16};   // 255: F(4x4) tail split (kernels_wino4.hip); before the rest, whose limits end the loop
            // (20 / 24 / 32 slices were timed for the batch-1 decoder layers in round 4 - two workgroups per CU instead of one: never picked)
            for (int split : SPLITS) {
                if (split > 1 && no_split) break;
                if (split == 255 && (!conv_igemm_tile_is_wino4(t) || conv_igemm_tile_is_wino4x(t))) continue;
                if (split > 1 && (split & (split - 1)) && !conv_igemm_tile_is_wino(t)) continue;
                // the latency form can split K (self-reducing, layers without a gate) but never wins: a sixteen-wave workgroup is alone on its CU
                // and lives ~6 us whatever its share of K, so S times as many workgroups are S times as many rounds (tools/time_lat_gemm.py:
                // level-6 transposed conv 18 / 25 / 41 / 73 us at S = 1 / 2 / 4 / 8).  CCVPE_TUNE_LAT_SPLIT=1 times them all the same.
                static const bool lat_split_on = getenv("CCVPE_TUNE_LAT_SPLIT") != nullptr;
                const bool lat_split = lat_split_on && conv_igemm_tile_proj_rt(t) >= 100 && op.gemm_m <= 256;
                if (split > 1 && conv_igemm_tile_is_pw(t) && !lat_split) break;   // the pointwise persistent tiles keep K whole
                if (split > 1 && split != 255) {   // split-K only where the grid underfills the chip and K is deep enough
                    // (the persistent Winograd grid also splits when the tile count is an awkward multiple of the
                    // 512 resident workgroups: 640 tiles = 1.25 per workgroup, 4 x 640 quarter-tiles = 5 each)
                    const bool wino = conv_igemm_tile_is_wino(t);
                    if (blocks >= (wino ? 2048 : 512) || blocks * split > (wino ? 8192 : 2048) || nkt < 4 * split) break;
                    if ((size_t)split * op.gemm_m * op.gemm_n > Plan::SPLITK_FLOATS) break;
                }
                for (int fuse = (split > 1 && conv_igemm_tile_is_pw(t)) ? 1 : 0; fuse < 2; ++fuse) {   // a split launch: with the reduce launch, and reducing itself (ticket.h) where the kernel can
                if (fuse && (split <= 1 || split == 255 || !conv_igemm_tile_can_fuse_split(t) || pl.tickets == nullptr || getenv("CCVPE_TUNE_NO_FUSED_SPLIT"))) break;
                const int cfg = t | ((fuse ? split + SPLIT_FUSED : split) << 8);
                *op.tile = cfg;
                op.fn(c);   // warm-up (also sets the dynamic-LDS attribute on first use)
                const int ran = conv_igemm_last_tile();   // (tile | split code << 8 of the launch just issued; reading it clears it)
                if ((ran & 0xff) != t) break;   // the launch did not take this tile (launch_conv_igemm fell back to its own pick): nothing to time under this name
                if (split == 255 && (ran >> 8) != 255) break;   // tail split not applicable to this grid
                if (split > 1 && conv_igemm_tile_is_pw(t) && ((ran >> 8) & 0xff) <= 1) break;   // (a gated layer: the launch kept K whole)
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
                static const char* verbose = getenv("CCVPE_TUNE_VERBOSE");   // dev: print every candidate of the launches whose name contains the string
                if (verbose && op.name.find(verbose) != std::string::npos)
                    std::fprintf(stderr, "tune %-28s %-28s split %3d%s: %8.1f us\n", op.name.c_str(), conv_igemm_tile_name(t), split, fuse ? " self-reducing" : "", 500.0 * ms);
                if (ms < best_ms) { best_ms = ms; best = cfg; }
                }
            }
        }
        *op.tile = best;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    HIPCHK(hipDeviceSynchronize());   // ... and the forward that follows may be issued on such a stream
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return ccvpe_fail(CCVPE_EHIP, "autotune launch failed: %s", hipGetErrorString(e));
    return 0;
}

int get_plan(ccvpe_handle_s* h, int B, int gh, int gw, Plan** out, int mode) {
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
        if (e != hipSuccess) { h->arena_floats = 0; return ccvpe_fail(CCVPE_ENOMEM, "workspace of %zu bytes: %s", pl->total * sizeof(float), hipGetErrorString(e)); }
        h->arena = (float*)d;
        h->arena_floats = pl->total;
    }
    if (pl->ticket_words) {   // the plan's ticket counters: zero now, and every kernel that draws tickets leaves them at zero
        HIPCHK(hipMalloc((void**)&pl->tickets, pl->ticket_words * sizeof(unsigned)));
        HIPCHK(hipMemset(pl->tickets, 0, pl->ticket_words * sizeof(unsigned)));
    }
    std::vector<bool> known;
    if (apply_tuning(h, *pl, known) > 0 && h->autotune) {
        int rc2 = autotune_plan(h, *pl, &known);
        if (rc2) return rc2;
        // choices made under a candidate-filter switch (CCVPE_TUNE_PREFER_*, CCVPE_NO_PW, CCVPE_TUNE_SPLITK ...) stay in this plan:
        // the key does not name the switch, so recording them would hand the filtered tiles to every later default process
        // through ccvpe_export_tuning / the user cache
        if (h->tuning_lookup) {
            record_tuning(h, *pl);
            h->tuned_plans++;
        }
    }
    *out = pl.get();
    h->plans.push_back(std::move(pl));
    return 0;
}
