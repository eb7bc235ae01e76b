// The execution plan of libccvpe_hip.so: CVM_*.forward (reference models.py:150-343, 448-652, 752-950, 1051-1244) restated
// as a static list of kernel launches over NHWC tensors; concatenations are channel-offset writes into pre-allocated
// buffers, the encoder taps are written by the producing GEMM's epilogue.
#include "ccvpe_internal.h"

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
ConvParams conv_params(const PackedConv& pc, const float* in, int in_ld, int B, int H, int W, int OH, int OW,
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
    p.wino4x_w = pc.wino4x; p.wino4x_bytes = (unsigned)pc.wino4x_bytes; p.wino4x_cfg = pc.wino4x_cfg;
    p.proj_w = pc.proj; p.proj_bytes = (unsigned)pc.proj_bytes;
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
    // stem + the depthwise conv of block 0 in one launch (block 0 has no expand conv; kernels_encoder.hip): the half-resolution 32-channel
    // stem output never reaches HBM.  CCVPE_STEM_DW=0: two launches (read per plan: tests toggle it)
    const bool stem_dw = !(getenv("CCVPE_STEM_DW") && std::atoi(getenv("CCVPE_STEM_DW")) == 0) && B0[0].e == 1 && B0[0].k == 3 && B0[0].s == 1 && B0[0].cin == 32 &&
                         (size_t)B * 3 * H * W * sizeof(float) < ((size_t)1 << 31);   // (the kernel addresses its input through one 32-bit buffer descriptor)
    StemParams stem_sp{};
    stem_sp.B = B; stem_sp.H = H; stem_sp.W = W; stem_sp.OH = ch; stem_sp.OW = cw; stem_sp.pad_t = lo; stem_sp.pad_l = lo; stem_sp.circular = circular;
    stem_sp.w = ew.stem_w; stem_sp.bias = ew.stem_b;
    Tensor cur = stem_dw ? Tensor{} : pl.alloc(B, ch, cw, 32);
    if (!stem_dw) {
        const StemParams sp = stem_sp;
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
        // latency plans: the work items a front may spread to = this encoder's share of the 256 CUs - the other encoder's fronts run beside
        // it on the other stream (both at 128: 1.20 ms per VIGOR frame, both at 256: 1.25; the aerial encoder is the longer chain).  The aerial
        // encoder takes 144 whatever it runs beside (its launches must not depend on the ground image: the aerial-only plan of
        // ccvpe_encode_aerial returns the bits of the full forward), the ground encoder its share by input pixels against a 512 x 512 tile.
        {
            const double share = (double)H * W / ((double)H * W + (double)CCVPE_SAT_HW * CCVPE_SAT_HW);
            const int dflt = is_grd ? std::max(64, std::min(128, (int)(256.0 * share / 16.0 + 0.5) * 16)) : 144;
            mp.spread = getenv("CCVPE_FRONT_SPREAD") ? std::atoi(getenv("CCVPE_FRONT_SPREAD")) : dflt;   // (read per plan: tests toggle it)
        }
        // small-spatial blocks: the whole expanded image of 16 channels lives in LDS (kernels_mbimg.hip); CCVPE_FUSE_MBCONV=0 / CCVPE_MBCONV_IMAGE=0 turn it off
        const bool image_off = getenv("CCVPE_MBCONV_IMAGE") && std::atoi(getenv("CCVPE_MBCONV_IMAGE")) == 0;   // read per plan: tests toggle it
        const bool image = b.e != 1 && bw.exp_lin != nullptr && h->fuse_mbconv != 0 && !image_off && mbconv_image_supported(mp);
        const bool fused = image || (b.e != 1 && bw.exp_lin != nullptr && mbconv_front_supported(b.k, b.s, b.cin, mid) &&
                           (h->fuse_mbconv == 2 || (h->fuse_mbconv == 1 && mbconv_front_profitable(b.k))));
        Tensor d = pl.alloc(B, oh, ow, mid);
        const bool with_stem = stem_dw && i == 0;
        // Squeeze-excite by ticket (round 4, ticket.h; model.py:113-118): the front kernel's last-arriving workgroup of a sample reduces the
        // pooling partials and computes the gates - no launch of its own.  CCVPE_SE_TICKET=0 keeps the separate launches (read per plan:
        // tests toggle it); kernels that take no ticket (plain depthwise, the workgroup form of the tile kernel) keep them too.
        const bool ticket_on = !(getenv("CCVPE_SE_TICKET") && std::atoi(getenv("CCVPE_SE_TICKET")) == 0) && mid <= 1152 && bw.sq <= 64;
        const int trows = !ticket_on ? 0 : with_stem ? stem_dw_tiles(oh, ow) : image ? mbconv_image_ticket_rows(mp) : fused ? mbconv_front_ticket_rows(mp) : 0;
        // Opt-in (CCVPE_SE_PROLOGUE=1; measured, not the default): latency plans (batch <= 4, blocks of <= 4096 rows) without a ticket either -
        // the image-resident front leaves its per-item squeeze rows and the latency-form project GEMM computes the gates in its prologue,
        // every wave those of its own K slice (kernels_proj.hip).  The front kernel loses its ~10 us combining step (block 12: 24.8 -> 15.5 us)
        // but the project GEMM gains as much (9.7 -> 20.9 us: rows -> squeezed vector -> gates is the same chain of dependent round trips,
        // now in front of its MFMAs, and its 128-register waves cannot keep all four kinds of requests in flight at once): 1.41 against
        // 1.28 ms per batch-1 frame.
        bool sep = false;
        if (trows > 0 && image && bw.project.proj != nullptr && h->cfg.reserved[0] == 0 && getenv("CCVPE_SE_PROLOGUE") && std::atoi(getenv("CCVPE_SE_PROLOGUE")) == 1) {
            ConvParams q = conv_params(bw.project, nullptr, mid, B, oh, ow, oh, ow, 1, 0, 0, ACT_NONE);
            q.M = B * oh * ow;
            q.se_rows = reinterpret_cast<const float*>(1); q.se_nrows = trows * (mid / 16); q.se_sq = bw.sq;
            sep = conv_proj_supported(q, 101);
        }
        const bool ticket = trows > 0 && !sep;
        const int S = ticket ? trows : with_stem ? stem_dw_tiles(oh, ow) : image ? mbconv_image_strips(mp) : fused ? mbconv_front_tiles(b.k, b.s, oh, ow) : depthwise_strip_lanes(B, oh, ow, mid, b.k, b.s);
        Tensor pool = pl.alloc(B, 1, S, mid);
        Tensor gate = pl.alloc(B, 1, 1, mid);
        SeTicket se{};
        size_t tick_off = 0;
        if (sep) {   // rows only: SeTicket::counter stays null, sqpart set (mbconv_image_kernel: no ticket, no combining step)
            se.per_sample = trows * (mid / 16);
            se.S = trows; se.C = mid; se.SQ = bw.sq; se.inv_hw = 1.f / (float)(oh * ow);
            se.w1 = bw.se_w1;
        }
        if (ticket) {
            tick_off = pl.alloc_tickets((size_t)B);
            se.per_sample = with_stem ? S : image ? S * (mid / 16) : S * mbconv_front_ticket_split(mp);   // tiles | (strip, 16-channel chunk) items | workgroups of a sample
            se.S = S; se.C = mid; se.SQ = bw.sq; se.inv_hw = 1.f / (float)(oh * ow);
            se.w1 = bw.se_w1; se.b1 = bw.se_b1; se.w2 = bw.se_w2; se.b2 = bw.se_b2;
            se.spec = (long long)B * se.per_sample <= 256 ? 1 : 0;   // latency plans: one workgroup per item, the chip not even filled once
        }
        // the image-resident kernel runs the squeeze conv per item (one row of SQ floats each); the others hand over pooling partial rows
        const bool parts = (ticket && image) || sep;
        Tensor sqp = parts ? pl.alloc(B, 1, se.per_sample, bw.sq) : Tensor{};
        auto fill_se = [=](const Ctx& c) {
            SeTicket t = se;
            if (ticket) { t.counter = c.tickets + tick_off; t.pool = c.ptr(pool); t.gate = c.ptr(gate); }
            if (parts) t.sqpart = c.ptr(sqp);
            return t;
        };
        const double se_flops = ticket ? 4.0 * B * mid * bw.sq : 0.0;
        if (with_stem) {
            StemDwParams sd{};
            sd.st = stem_sp; sd.wd = bw.dw_w; sd.bd = bw.dw_b;
            std::vector<Tensor> uses = {d, pool};
            if (ticket) uses.push_back(gate);
            pl.add(tag + ".stem_b0dw", uses, [=](const Ctx& c) {
                StemDwParams q = sd; q.st.in = is_grd ? c.grd : c.sat; q.out = c.ptr(d); q.pool_partial = c.ptr(pool); q.se = fill_se(c);
                launch_stem_dw(q, c.stream);
            }, 2.0 * B * ch * cw * 32 * 27 + 2.0 * B * oh * ow * mid * 9 + se_flops, 4.0 * B * (3.0 * H * W + 32.0 * oh * ow));
        } else if (fused) {
            std::vector<Tensor> uses = {xin, d, pool};
            if (ticket) uses.push_back(gate);
            if (parts) uses.push_back(sqp);
            (void)sep;
            pl.add(bn + ".expand_dw", uses, [=](const Ctx& c) {
                MbFrontParams q = mp; q.x = c.ptr(xin); q.out = c.ptr(d); q.pool = c.ptr(pool); q.se = fill_se(c);
                if (image) launch_mbconv_image(q, c.stream); else launch_mbconv_front(q, c.stream);
            }, 2.0 * B * ch * cw * b.cin * mid + 2.0 * B * oh * ow * mid * b.k * b.k + se_flops, 4.0 * B * ((double)ch * cw * b.cin + (double)oh * ow * mid));
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
        const int SC = std::max(1, std::min(16, S / 32));
        Tensor pooled = pl.alloc(B, 1, SC, mid);
        Tensor sqt = pl.alloc(B, 1, 1, 64);
        if (!ticket && !sep) {
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
            std::vector<Tensor> uses = {d, sep ? sqp : gate, o};
            if (skip) uses.push_back(xin);
            for (int t = 0; t < td.n; ++t) uses.push_back(td.t[t]);
            const int se_nrows = se.per_sample, se_sq = bw.sq;
            const float se_inv = se.inv_hw;
            const float *se_b1 = bw.se_b1, *se_w2 = bw.se_w2, *se_b2 = bw.se_b2;
            pl.add_conv(bn + (sep ? ".se_project" : ".project"), uses, B * oh * ow, pc->N, pc->Kpad, [=](const Ctx& c, int tile) {
                ConvParams p = conv_params(*pc, c.ptr(d), mid, B, oh, ow, oh, ow, 1, 0, 0, ACT_NONE);
                if (sep) {
                    p.se_rows = c.ptr(sqp); p.se_nrows = se_nrows; p.se_sq = se_sq; p.se_inv_hw = se_inv; p.se_b1 = se_b1; p.se_w2 = se_w2; p.se_b2 = se_b2;
                    tile = conv_proj_lat_tile();   // the only kernel that computes the gates itself
                } else {
                    p.gate = c.ptr(gate);
                }
                if (skip) { p.resid = c.ptr(xin); p.resid_ld = xin.C; }
                p.dst[0] = {c.ptr(o), o.C, 0}; p.ndst = 1;
                for (int t = 0; t < td.n; ++t) p.dst[p.ndst++] = c.dst(td.t[t], td.coff[t]);
                c.launch_conv(p, tile);
            }, 2.0 * B * oh * ow * mid * b.cout + (sep ? 4.0 * B * mid * bw.sq : 0.0), 4.0 * B * oh * ow * (mid + b.cout * (1 + td.n)));
            pl.ops.back().is_pw = true;
            pl.ops.back().proj_ok = pc->proj != nullptr && h->cfg.reserved[0] == 0;
            if (sep) pl.ops.back().tile.reset();   // not a tuning candidate: one kernel serves it (the launch above names its tile)
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
size_t cache_layout(const VariantSpec& vs, int B, size_t off[6]) {
    size_t o = 0;
    off[0] = o; o += (size_t)B * 64 * vs.sat_desc;
    for (int t = 0; t < 5; ++t) { off[t + 1] = o; o += (size_t)B * TAP_HW[t] * TAP_C[t]; }
    return o;
}

static int build_aerial_plan(ccvpe_handle_s* h, Plan& pl, int B);
extern "C" int ccvpe_max_micro_batch(int32_t variant, float ori_noise, int32_t grd_h, int32_t grd_w);

int build_plan(ccvpe_handle_s* h, Plan& pl, int B, int gh, int gw, int mode) {
    if (mode == 1) return build_aerial_plan(h, pl, B);
    const bool cached = mode == 2;
    pl.mode = mode;
    const VariantSpec& vs = h->vs;
    pl.B = B; pl.gh = gh; pl.gw = gw; pl.debug = h->debug;
    pl.scratch = pl.alloc(1, 1, 1, (int)Plan::SPLITK_FLOATS);
    pl.two_streams = h->two_streams && !h->debug;
    if (pl.two_streams) pl.scratch2 = pl.alloc(1, 1, 1, (int)Plan::SPLITK_FLOATS);
    if (h->wino && h->cfg.reserved[0] == 0 && getenv("CCVPE_WINO4P")) {
        // split Winograd form (kernels_wino4p.hip), opt-in: measured equal to or slower than the fused form on every decoder layer
        // (profiles/r03_wino4_forms.md), so plans neither reserve its scratch nor time its tiles by default.  Room for the largest
        // pre-transformed layer input (36 KiB per 16 x 16 pixel block and 16-channel group) that stays below 256 MiB
        auto vneed = [&](const DecLevel* lv) {
            size_t best = 0;
            for (int j = 0; j < 5; ++j) {
                const size_t hout = (size_t)16 << j, mbl = (size_t)B * (hout / 16) * (hout / 16);
                for (int cin : {deconv_width(lv[j]) + lv[j].skip, lv[j].mid}) {
                    const size_t f = mbl * ((cin + 15) / 16) * 9216;
                    if (f <= Plan::WINO_V_FLOATS) best = std::max(best, f);
                }
            }
            return best;
        };
        const size_t v0 = std::max(vneed(vs.loc), vneed(vs.ori)), v1 = vneed(vs.ori);   // stream 0's also serves a serial issue of both chains
        if (v0) pl.vscratch = pl.alloc(1, 1, 1, (int)v0);
        if (pl.two_streams && v1) pl.vscratch2 = pl.alloc(1, 1, 1, (int)v1);
    }
    // hipGraph replay: opt-in since round 4 (CCVPE_GRAPH=1).  Rounds 2-3 replayed plans of <= 4 samples: with ~330 launches per frame the
    // host could not keep up.  A batch-1 frame is 137 launches now; issued eagerly (interleaved over the two streams, Plan::schedule) the
    // GPU starts on the first while the host still hands over the rest, and a synchronised frame takes 1.31 ms against 1.44 ms replayed
    // (the replay's set-up precedes its first kernel); back to back both run at the GPU's pace.
    pl.use_graph = !cached && h->graph_mode == 1;
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
        return ccvpe_fail(CCVPE_EINVAL, "ground image %dx%d gives a %d-row feature volume, the descriptor heads expect %d rows", gh, gw, fh, vs.feat_h);
    int L[6];
    for (int k = 0; k < 6; ++k) {
        L[k] = fw * vs.head_ch[k];
        if (L[k] > vs.match_ch[k])
            return ccvpe_fail(CCVPE_EINVAL, "descriptor length %d exceeds aerial channels %d at level %d", L[k], vs.match_ch[k], k + 1);
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
        loc_cat[j] = pl.alloc(B, hw_in * 2, hw_in * 2, deconv_width(vs.loc[j]) + vs.loc[j].skip);
        ori_cat[j] = pl.alloc(B, hw_in * 2, hw_in * 2, deconv_width(vs.ori[j]) + vs.ori[j].skip);
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
        td[t].t[0] = loc_cat[t]; td[t].coff[0] = deconv_width(vs.loc[t]);
        td[t].t[1] = ori_cat[t]; td[t].coff[1] = deconv_width(vs.ori[t]);
    }
    size_t coff[6];
    cache_layout(vs, B, coff);
    if (!cached) {
        plan_encoder(h, pl, h->sat_enc, false, B, CCVPE_SAT_HW, CCVPE_SAT_HW, false, td, senc, "sat");
    } else {
        for (int t = 0; t < 5; ++t) {   // cached encoder taps -> skip halves of the decoder concat buffers
            Tensor lc = loc_cat[t], oc = ori_cat[t];
            const int lcoff = deconv_width(vs.loc[t]), ocoff = deconv_width(vs.ori[t]);
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
        pl.ops.back().is_pw = true;
        pl.ops.back().proj_ok = pc->proj != nullptr && h->cfg.reserved[0] == 0;
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
        pl.ops.back().proj_ok = pc->proj != nullptr && h->cfg.reserved[0] == 0;   // (k2s2: the latency form of kernels_proj.hip only)
        pl.taps["sat_descriptor_map"] = {dmap, 0, D};
    }

    // ---- decoders ----
    auto plan_level = [&](const DecoderW& dw, const DecLevel* lv, int j, Tensor din, Tensor cat, const std::string& tag) -> Tensor {
        const int hin = 8 << j, hout = hin * 2;
        const DecLevel& l = lv[j];
        {
            const PackedConv* pc = &dw.deconv[j];
            const int cout = deconv_width(l);   // (columns past l.dout: zero weights and biases, written as zeros)
            pl.add_conv(tag + ".deconv", {din, cat}, B * hin * hin, pc->N, pc->Kpad, [=](const Ctx& c, int tile) {
                ConvParams p = conv_params(*pc, c.ptr(din), din.C, B, hin, hin, hin, hin, 1, 0, 0, ACT_NONE);
                p.mode = MODE_DECONV; p.deconv_cout = cout;
                p.dst[0] = c.dst(cat); p.ndst = 1;
                c.launch_conv(p, tile);
            }, 2.0 * B * hin * hin * (double)l.din * 4 * l.dout, 4.0 * B * hin * hin * ((double)din.C + 4.0 * l.dout));
            pl.ops.back().is_pw = !din.split;
            pl.ops.back().proj_ok = pc->proj != nullptr && !din.split && h->cfg.reserved[0] == 0;
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
            pl.ops.back().wino4x_ok = pl.ops.back().wino_ok && pc->wino4x != nullptr;
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
            pl.ops.back().wino4x_ok = pl.ops.back().wino_ok && pc->wino4x != nullptr;
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
    // The preparation launch of every matching level (rolled descriptor / Gm, Mk) depends on the ground descriptor only: the six of them are
    // moved in front of the first matching level, where the localisation stream otherwise waits for the aerial encoder (batch 1: ~40 us off the
    // critical path).  CCVPE_MATCH_PREP_EARLY=0: inside each level's launch as before.
    const bool prep_early = !(getenv("CCVPE_MATCH_PREP_EARLY") && std::atoi(getenv("CCVPE_MATCH_PREP_EARLY")) == 0);
    size_t first_match_op = (size_t)-1;
    std::vector<std::function<MatchParams(const Ctx&)>> prep_fills;
    std::vector<Tensor> prep_uses;
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
        mp.no_wide = (getenv("CCVPE_MATCH_WIDE") && std::atoi(getenv("CCVPE_MATCH_WIDE")) == 0) ? 1 : 0;   // (read per plan: tests toggle it)
        mp.cat_max_ld = 8 + C;
        mp.cat_all_ld = rpad + C;
        Tensor xin = x, lin = loc_in[k];
        Tensor ggs = pl.alloc(B, 1, 1, (int)match_scratch_floats(C));
        const bool first = (k == 0);
        const int goff = loff[k];
        const int R = mp.R;
        std::vector<Tensor> uses = {xin, desc, lin, ggs};
        if (first) uses.push_back(ori_in6);
        mp.prep_done = prep_early ? 1 : 0;
        auto fill = [=](const Ctx& c) {        // the same parameters for the preparation and the main launch: both pick the same form
            MatchParams q = mp;
            q.x = c.ptr(xin); q.g = c.ptr(desc) + goff;
            q.ms = c.out.matching_score[k];
            q.cat_max = c.ptr(lin);
            q.cat_all = first ? c.ptr(ori_in6) : nullptr;
            q.gg_scratch = c.ptr(ggs);
            return q;
        };
        if (first_match_op == (size_t)-1) first_match_op = pl.ops.size();
        if (prep_early) {
            prep_fills.push_back(fill);
            if (prep_uses.empty()) prep_uses.push_back(desc);
            prep_uses.push_back(ggs);
        }
        pl.add("match" + std::to_string(k + 1), uses, [=](const Ctx& c) { launch_match(fill(c), c.stream); },
               4.0 * B * hw * (double)R * L[k], 4.0 * B * hw * (2.0 * C + R + 8));
        pl.taps["loc_in" + std::to_string(6 - k)] = {lin, 0, lin.C};
        if (k == 5 && h->fuse_level1) { plan_level1_fused(h->loc, lin, vs.loc[5].din, 1, false, Tensor{}, "loc1"); break; }
        Tensor o = plan_level(h->loc, vs.loc, k, lin, loc_cat[k], "loc" + std::to_string(6 - k));
        if (k < 5) { pl.taps["loc_level" + std::to_string(6 - k)] = {o, 0, o.C}; x = o; }
        else loc_mid = o;
    }
    if (!prep_fills.empty()) {   // the preparation of all levels: ONE launch (kernels_match.hip), in front of the first matching level
        const size_t n0 = pl.ops.size();
        pl.add("match.prep", prep_uses, [prep_fills](const Ctx& c) {
            MatchParams ps[6];
            const int n = (int)std::min<size_t>(prep_fills.size(), 6);
            for (int i = 0; i < n; ++i) ps[i] = prep_fills[i](c);
            launch_match_prep_all(ps, n, c.stream);
        }, 0.0, 0.0);
        std::rotate(pl.ops.begin() + first_match_op, pl.ops.begin() + n0, pl.ops.end());
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
        return ccvpe_fail(CCVPE_EINVAL, "micro-batch %d needs a %d x %d x %d x %d tensor of %zu bytes; the kernels address tensors with 32-bit byte offsets (< 2 GiB): "
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
        pl.ops.back().proj_ok = pc->proj != nullptr && h->cfg.reserved[0] == 0;   // (the same candidates - the same tuning-table entry - as the full plan's launch)
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



