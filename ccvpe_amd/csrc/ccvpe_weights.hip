// State_dict ingestion of libccvpe_hip.so: the 818-key layout of the reference modules (models.py:50-148, 347-446,
// efficientnet_pytorch/model.py), BatchNorm folding, repacking into the kernels' layouts (implicit-GEMM K order, Winograd
// F(2x2) / F(4x4) weight transforms in double precision), and the packed-weight cache.
#include "ccvpe_internal.h"

#include <unistd.h>

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
void build_expect(ccvpe_handle_s* h) {
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

// Environment switches the packer branches on: ONE list, read by pack_conv's callers' cache key (ccvpe_pack_switches) - a packed-weight
// file written under one setting must never be loaded under another (ccvpe_load_packed restores every descriptor from the file).
static const char* const PACK_SWITCHES[] = {"CCVPE_NO_PROJ", "CCVPE_WINO4_MIN_N", "CCVPE_NO_WINO4", "CCVPE_NO_WINO4X", "CCVPE_PAD_CONCAT"};
extern "C" const char* ccvpe_pack_switches(void) {
    static thread_local std::string s;
    s.clear();
    for (const char* k : PACK_SWITCHES)
        if (const char* v = getenv(k)) s += std::string(k) + "=" + v + ";";
    return s.c_str();
}

// Generic packer: rows n < N, k = tap*cinp + cmap(c).  `get(n, tap, c)` returns the (already scaled) weight.
int pack_conv(ccvpe_handle_s* h, PackedConv& pc, int N, int taps, int cin, int cinp, const std::vector<int>& cmap,
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
    if (((KH == 1 && KW == 1 && taps == 1 && cin == cinp && conv_proj_wanted(N, cin)) || conv_proj_lat_wanted(taps, KH, KW, cinp)) && !getenv("CCVPE_NO_PROJ")) {
        // fragment-order copy for kernels_proj.hip; columns follow the packed channel positions (cmap: the [8 | C] layout of the
        // decoder's transposed convs has zero columns where the score padding sits)
        std::vector<int> inv(cinp, -1);
        for (int c = 0; c < cin; ++c) inv[cmap[c]] = c;
        std::vector<float> up;
        conv_proj_pack(N, cinp, [&](int n, int k) { const int t = k / cinp, c = inv[k - t * cinp]; return c >= 0 ? get(n, t, c) : 0.f; }, up, taps);
        pc.proj_bytes = up.size() * sizeof(float);
        if ((rc = upload(h, up, &pc.proj))) return rc;
    }
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
        // xi-split F(4x4,3x3) for the narrow layers (one n-block of <= 128 channels; kernels_wino4x.hip)
        if (N >= 24 && conv_wino4x_config(N) >= 0 && !getenv("CCVPE_NO_WINO4X")) {
            std::vector<float> ux;
            conv_wino4x_pack(N, cin, get, ux, &pc.wino4x_cfg);
            pc.wino4x_bytes = ux.size() * sizeof(float);
            if ((rc = upload(h, ux, &pc.wino4x))) return rc;
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
std::vector<int> identity_map(int n) { std::vector<int> m(n); for (int i = 0; i < n; ++i) m[i] = i; return m; }

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


static int build_decoder(ccvpe_handle_s* h, DecoderW& d, const DecLevel* lv, const std::string& sfx, int nscore_l6, bool every_level_scored) {
    int rc;
    for (int j = 0; j < 6; ++j) {
        std::string n = std::to_string(6 - j);
        {   // ConvTranspose2d weight [cin][cout][2][2] -> rows n = (dy*2+dx)*cout + o
            const auto& w = h->host["deconv" + n + sfx + ".weight"];
            const auto& b = h->host["deconv" + n + sfx + ".bias"];
            const int cin = lv[j].din, cout = lv[j].dout, cw = deconv_width(lv[j]);   // cw >= cout: zero columns (ccvpe_internal.h)
            int nscore = 0;
            if (every_level_scored) nscore = 1;
            else if (j == 0) nscore = nscore_l6;
            const int spad = score_pad(nscore);
            const int cinp = spad + (cin - nscore);
            std::vector<int> cmap(cin);
            for (int c = 0; c < cin; ++c) cmap[c] = c < nscore ? c : c - nscore + spad;
            std::vector<float> bias(4 * cw, 0.f);
            for (int qd = 0; qd < 4; ++qd) for (int o = 0; o < cout; ++o) bias[qd * cw + o] = b[o];
            if ((rc = pack_conv(h, d.deconv[j], 4 * cw, 1, cin, cinp, cmap,
                                [&](int nn, int, int c) { int qd = nn / cw, o = nn % cw; return o < cout ? w[((size_t)c * cout + o) * 4 + qd] : 0.f; },
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
            const int dout = lv[j].dout, cw = deconv_width(lv[j]);
            const int cin = dout + lv[j].skip, cinw = cw + lv[j].skip, cout = lv[j].mid;   // cinw: with the zero columns behind the transposed conv's channels
            if ((rc = pack_conv(h, d.conva[j], cout, 9, cinw, cinw, identity_map(cinw),
                                [&](int nn, int t, int c) {
                                    if (c >= dout && c < cw) return 0.f;
                                    const int cr = c < dout ? c : c - (cw - dout);
                                    return w[((size_t)nn * cin + cr) * 9 + t];
                                },
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

// ---- packed-weight cache --------------------------------------------------------------------------
// ccvpe_finalize_weights folds BatchNorm, repacks ~60 M parameters into the kernels' layouts and runs the Winograd weight
// transforms in double precision: seconds per handle.  Its result is a set of device buffers plus plain-data descriptor
// structs that point into them.  ccvpe_save_packed writes both to a file; ccvpe_load_packed recreates the buffers and
// re-bases every pointer of the descriptors (old device address -> new), so a later process skips the state_dict
// ingestion and the packing entirely.  The caller keys the file (ccvpe_amd/models.py: sha256 of the state_dict bytes,
// variant, precision, library build digest); the header carries variant / precision / struct sizes and is checked.
struct PackedHeader {
    char magic[8];                 // "CCVPEPK3"
    int32_t variant, precision, circular, fuse_level1;
    uint64_t n_allocs, sz_encoder, sz_decoder, sz_conv;
    uint64_t n_relocs;             // (descriptor word index, buffer index) pairs behind the descriptor structs
};
static void packed_state_io(ccvpe_handle_s* h, const std::function<void(void*, size_t)>& io) {
    io(&h->grd_enc, sizeof(EncoderW)); io(&h->sat_enc, sizeof(EncoderW));
    io(&h->grd_heads, sizeof(PackedConv)); io(&h->sat_desc, sizeof(PackedConv));
    io(h->grd_wh, sizeof(h->grd_wh)); io(h->grd_b2, sizeof(h->grd_b2));
    io(&h->loc, sizeof(DecoderW)); io(&h->ori, sizeof(DecoderW));
}

extern "C" {


int ccvpe_skip_weight(ccvpe_handle h, const char* key) {
    if (!h || !key) return ccvpe_fail(CCVPE_EINVAL, "null argument");
    if (!h->expect.count(key)) return ccvpe_fail(CCVPE_EKEY, "unexpected state_dict key '%s'", key);
    h->skipped.insert(key);
    return 0;
}

int ccvpe_set_weight(ccvpe_handle h, const char* key, const float* data, const int64_t* shape, int32_t ndim) {
    if (!h || !key || !data) return ccvpe_fail(CCVPE_EINVAL, "null argument");
    auto it = h->expect.find(key);
    if (it == h->expect.end()) return ccvpe_fail(CCVPE_EKEY, "unexpected state_dict key '%s'", key);
    const auto& es = it->second;
    bool ok = (int)es.size() == ndim;
    size_t n = 1;
    for (int i = 0; ok && i < ndim; ++i) { ok = es[i] == shape[i]; n *= (size_t)shape[i]; }
    if (!ok) return ccvpe_fail(CCVPE_EINVAL, "shape mismatch for '%s'", key);
    HIPCHK(hipSetDevice(h->cfg.device));
    std::vector<float> v(n);
    HIPCHK(hipMemcpy(v.data(), data, n * sizeof(float), hipMemcpyDefault));
    h->host[key] = std::move(v);
    h->finalized = false;
    return 0;
}

int ccvpe_finalize_weights(ccvpe_handle h) {
    if (!h) return ccvpe_fail(CCVPE_EINVAL, "null handle");
    HIPCHK(hipSetDevice(h->cfg.device));
    for (auto& kv : h->expect) {
        const std::string& k = kv.first;
        const bool optional = k.find("num_batches_tracked") != std::string::npos || k.find("._fc.") != std::string::npos;
        if (!h->host.count(k) && !(optional || h->skipped.count(k)))
            return ccvpe_fail(CCVPE_EKEY, "missing state_dict key '%s'", k.c_str());
        if (!h->host.count(k) && !optional) return ccvpe_fail(CCVPE_EKEY, "key '%s' was skipped but is required", k.c_str());
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
    if (!h || !path) return ccvpe_fail(CCVPE_EINVAL, "null argument");
    if (!h->finalized) return ccvpe_fail(CCVPE_ESTATE, "ccvpe_finalize_weights has not been called");
    static_assert(std::is_trivially_copyable<EncoderW>::value && std::is_trivially_copyable<DecoderW>::value && std::is_trivially_copyable<PackedConv>::value,
                  "descriptor structs are written as plain bytes");
    HIPCHK(hipSetDevice(h->cfg.device));
    // several ranks may miss the cache at once and save the same key: every writer has its own temporary file and the
    // finished file is renamed into place, so a reader only ever sees a complete file (whose content is the same whoever wins)
    char suffix[64];
    std::snprintf(suffix, sizeof(suffix), ".tmp.%ld.%llx", (long)getpid(), (unsigned long long)(uintptr_t)h ^ (unsigned long long)std::rand());
    const std::string tmp = std::string(path) + suffix;
    FILE* f = std::fopen(tmp.c_str(), "wb");
    if (!f) return ccvpe_fail(CCVPE_EINVAL, "cannot open %s for writing", tmp.c_str());
    // relocation table: every 8-byte word of the descriptor structs that holds the address of one of the handle's device
    // buffers, as (word index in the descriptor stream, buffer index) - the loader patches exactly these words
    std::map<uint64_t, uint64_t> index_of;
    for (size_t i = 0; i < h->dev_allocs.size(); ++i) index_of[(uint64_t)(uintptr_t)h->dev_allocs[i]] = i;
    std::vector<uint64_t> relocs;
    uint64_t word0 = 0;
    packed_state_io(h, [&](void* p, size_t n) {
        const uint64_t* w = reinterpret_cast<const uint64_t*>(p);
        for (size_t i = 0; i + 8 <= n; i += 8, ++w) {
            auto it = *w ? index_of.find(*w) : index_of.end();
            if (it != index_of.end()) { relocs.push_back(word0 + i / 8); relocs.push_back(it->second); }
        }
        word0 += n / 8;
    });
    PackedHeader hd{};
    std::memcpy(hd.magic, "CCVPEPK3", 8);
    hd.variant = h->cfg.variant; hd.precision = h->cfg.reserved[0]; hd.circular = h->cfg.circular_padding; hd.fuse_level1 = h->fuse_level1 ? 1 : 0;
    hd.n_allocs = h->dev_allocs.size(); hd.sz_encoder = sizeof(EncoderW); hd.sz_decoder = sizeof(DecoderW); hd.sz_conv = sizeof(PackedConv);
    hd.n_relocs = relocs.size() / 2;
    bool ok = std::fwrite(&hd, sizeof(hd), 1, f) == 1;
    packed_state_io(h, [&](void* p, size_t n) { ok = ok && std::fwrite(p, 1, n, f) == n; });
    ok = ok && (relocs.empty() || std::fwrite(relocs.data(), 8, relocs.size(), f) == relocs.size());
    std::vector<char> buf;
    for (size_t i = 0; ok && i < h->dev_allocs.size(); ++i) {
        const uint64_t bytes = h->dev_alloc_bytes[i];
        buf.resize(bytes);
        if (hipMemcpy(buf.data(), h->dev_allocs[i], bytes, hipMemcpyDeviceToHost) != hipSuccess) { ok = false; break; }
        ok = std::fwrite(&bytes, 8, 1, f) == 1 && std::fwrite(buf.data(), 1, bytes, f) == bytes;
    }
    ok = (std::fclose(f) == 0) && ok;
    if (!ok || std::rename(tmp.c_str(), path) != 0) { std::remove(tmp.c_str()); return ccvpe_fail(CCVPE_EINVAL, "writing %s failed", path); }
    return 0;
}

int ccvpe_load_packed(ccvpe_handle h, const char* path) {
    if (!h || !path) return ccvpe_fail(CCVPE_EINVAL, "null argument");
    HIPCHK(hipSetDevice(h->cfg.device));
    FILE* f = std::fopen(path, "rb");
    if (!f) return ccvpe_fail(CCVPE_EINVAL, "cannot open %s", path);
    PackedHeader hd{};
    auto bad = [&](const char* why) { std::fclose(f); return ccvpe_fail(CCVPE_EINVAL, "%s: %s", path, why); };
    if (std::fread(&hd, sizeof(hd), 1, f) != 1 || std::memcmp(hd.magic, "CCVPEPK3", 8) != 0) return bad("not a packed-weight file of this library version");
    if (hd.variant != h->cfg.variant || hd.precision != h->cfg.reserved[0] || hd.circular != h->cfg.circular_padding) return bad("packed for a different variant / precision / padding mode");
    if (hd.sz_encoder != sizeof(EncoderW) || hd.sz_decoder != sizeof(DecoderW) || hd.sz_conv != sizeof(PackedConv)) return bad("descriptor layout mismatch");
    if (hd.n_allocs == 0 || hd.n_allocs > 100000 || hd.n_relocs > 100000) return bad("implausible buffer / relocation counts");
    for (void* p : h->dev_allocs) (void)hipFree(p);
    h->dev_allocs.clear(); h->dev_alloc_bytes.clear(); h->plans.clear(); h->last_plan = nullptr; h->finalized = false;
    bool ok = true;
    uint64_t nwords = 0;
    packed_state_io(h, [&](void* p, size_t n) { ok = ok && std::fread(p, 1, n, f) == n; nwords += n / 8; });
    std::vector<uint64_t> relocs(hd.n_relocs * 2);
    ok = ok && (relocs.empty() || std::fread(relocs.data(), 8, relocs.size(), f) == relocs.size());
    for (size_t i = 0; ok && i < relocs.size(); i += 2) ok = relocs[i] < nwords && relocs[i + 1] < hd.n_allocs;
    std::vector<char> buf;
    for (uint64_t i = 0; ok && i < hd.n_allocs; ++i) {
        uint64_t bytes = 0;
        if (std::fread(&bytes, 8, 1, f) != 1 || bytes == 0 || bytes > ((uint64_t)1 << 33)) { ok = false; break; }
        buf.resize(bytes);
        if (std::fread(buf.data(), 1, bytes, f) != bytes) { ok = false; break; }
        void* d = nullptr;
        if (hipMalloc(&d, bytes) != hipSuccess) { ok = false; break; }
        h->dev_allocs.push_back(d); h->dev_alloc_bytes.push_back(bytes);
        if (hipMemcpy(d, buf.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) { ok = false; break; }
    }
    ok = ok && std::fgetc(f) == EOF;   // nothing may follow the last buffer
    std::fclose(f);
    if (!ok) {   // the descriptors may hold another process's addresses now: leave the handle empty, never half loaded
        for (void* p : h->dev_allocs) (void)hipFree(p);
        h->dev_allocs.clear(); h->dev_alloc_bytes.clear();
        packed_state_io(h, [&](void* p, size_t n) { std::memset(p, 0, n); });
        ccvpe_err() = std::string(path) + ": truncated, inconsistent or unreadable packed-weight file";
        return CCVPE_EINVAL;
    }
    // re-base exactly the recorded pointer words
    std::map<uint64_t, uint64_t> patch;   // word index -> new address
    for (size_t i = 0; i < relocs.size(); i += 2) patch[relocs[i]] = (uint64_t)(uintptr_t)h->dev_allocs[relocs[i + 1]];
    uint64_t word0 = 0;
    packed_state_io(h, [&](void* p, size_t n) {
        uint64_t* w = reinterpret_cast<uint64_t*>(p);
        for (auto it = patch.lower_bound(word0); it != patch.end() && it->first < word0 + n / 8; ++it) w[it->first - word0] = it->second;
        word0 += n / 8;
    });
    // sizes the kernels index with must agree with the variant (a foreign or damaged file must not drive addressing)
    const int want_desc = h->vs.sat_desc;
    if (h->sat_desc.N != want_desc || h->grd_enc.head.N != 1280 || h->sat_enc.head.N != 1280 || h->loc.conva[0].N != h->vs.loc[0].mid || h->ori.conva[0].N != h->vs.ori[0].mid) {
        for (void* p : h->dev_allocs) (void)hipFree(p);
        h->dev_allocs.clear(); h->dev_alloc_bytes.clear();
        packed_state_io(h, [&](void* p, size_t n) { std::memset(p, 0, n); });
        return ccvpe_fail(CCVPE_EINVAL, "%s: layer sizes do not match this variant", path);
    }
    h->host.clear();
    h->fuse_level1 = hd.fuse_level1 != 0 && h->fuse_level1;
    h->finalized = true;
    return 0;
}

}  // extern "C"
