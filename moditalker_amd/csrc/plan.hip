// Host side of libmtv_hip.so: UNet block list -> launch plan, weight repacking, workspace,
// hipGraph capture of the denoising step, and the C ABI of include/mtv_hip.h.
//
// Structure follows what MToV/models/ddpm/unet.py:710-975 constructs and :995-1117 executes, but
// the plan is MI355X-first: one token-major channels-last buffer per activation, planes batched in
// every launch, skip concatenations expressed as two-source K loops, timestep FiLM for all
// ResBlocks in one GEMV, the whole step replayed as one hipGraph with a device-side step counter.
#include "plan_internal.h"

static thread_local std::string g_err;
int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

// =====================================================================================
// structure (mirrors unet.py:710-975 for dims=2, resblock_updown=True, legacy attention)
// =====================================================================================
static int build_structure(mtv_ctx* c) {
    const mtv_config& f = c->cfg;
    const int mc = f.model_channels;
    auto has_attn = [&](int ds) {
        for (int i = 0; i < f.n_attention_resolutions; ++i)
            if (f.attention_resolutions[i] == ds) return true;
        return false;
    };
    int film = 0;
    auto res_layer = [&](int cin, int cout, int updown, const std::string& pre) {
        Layer l;
        l.type = 1;
        l.res = ResDesc{cin, cout, updown, film};
        l.c = cout;
        l.pre = pre;
        film += f.use_scale_shift_norm ? 2 * cout : cout;
        return l;
    };
    auto attn_layer = [&](int ch, const std::string& pre) {
        Layer l;
        l.type = 2;
        l.res = ResDesc{0, 0, 0, 0};
        l.c = ch;
        l.pre = pre;
        return l;
    };
    std::vector<int> chans;
    {
        Stage s;
        Layer l;
        l.type = 0;
        l.res = ResDesc{16, mc, 0, 0};
        l.c = mc;
        l.pre = "input_blocks.0.0.";
        s.layers.push_back(l);
        s.tap = "in0";
        c->inputs.push_back(s);
        chans.push_back(mc);
    }
    int ch = mc, ds = 1;
    for (int level = 0; level < f.n_levels; ++level) {
        const int mult = f.channel_mult[level];
        for (int k = 0; k < f.num_res_blocks; ++k) {
            const int idx = (int)c->inputs.size();
            Stage s;
            const std::string pre = "input_blocks." + std::to_string(idx) + ".";
            s.layers.push_back(res_layer(ch, mult * mc, 0, pre + "0."));
            ch = mult * mc;
            if (has_attn(ds)) s.layers.push_back(attn_layer(ch, pre + "1."));
            s.attn1_c = ch;
            s.attn1_pre = "input_attns." + std::to_string(idx) + ".";
            s.tap = "in" + std::to_string(idx);
            c->inputs.push_back(s);
            chans.push_back(ch);
        }
        if (level != f.n_levels - 1) {
            const int idx = (int)c->inputs.size();
            Stage s;
            s.layers.push_back(res_layer(ch, ch, 1, "input_blocks." + std::to_string(idx) + ".0."));
            s.attn1_c = ch;
            s.attn1_pre = "input_attns." + std::to_string(idx) + ".";
            s.tap = "in" + std::to_string(idx);
            c->inputs.push_back(s);
            chans.push_back(ch);
            ds *= 2;
        }
    }
    c->middle.layers.push_back(res_layer(ch, ch, 0, "middle_block.0."));
    c->middle.layers.push_back(attn_layer(ch, "middle_block.1."));
    c->middle.layers.push_back(res_layer(ch, ch, 0, "middle_block.2."));
    c->middle.attn1_c = ch;
    c->middle.attn1_pre = "mid_attn.";
    c->middle.tap = "mid";
    for (int level = f.n_levels - 1; level >= 0; --level) {
        const int mult = f.channel_mult[level];
        for (int i = 0; i <= f.num_res_blocks; ++i) {
            const int ich = chans.back();
            chans.pop_back();
            const int idx = (int)c->outputs.size();
            Stage s;
            const std::string pre = "output_blocks." + std::to_string(idx) + ".";
            int j = 0;
            s.layers.push_back(res_layer(ch + ich, mc * mult, 0, pre + std::to_string(j++) + "."));
            ch = mc * mult;
            if (has_attn(ds)) s.layers.push_back(attn_layer(ch, pre + std::to_string(j++) + "."));
            if (level && i == f.num_res_blocks) {
                s.layers.push_back(res_layer(ch, ch, 2, pre + std::to_string(j++) + "."));
                ds /= 2;
            }
            s.attn1_c = ch;
            s.attn1_pre = "output_attns." + std::to_string(idx) + ".";
            s.tap = "out" + std::to_string(idx);
            c->outputs.push_back(s);
        }
    }
    c->film_total = film;
    int nres = 0, nattn = 0;
    auto count = [&](const Stage& s) {
        for (auto& l : s.layers) {
            if (l.type == 1) ++nres;
            if (l.type == 2) ++nattn;
        }
        if (s.attn1_c) ++nattn;
    };
    for (auto& s : c->inputs) count(s);
    count(c->middle);
    for (auto& s : c->outputs) count(s);
    c->n_sites = 2 * nres + nattn + 1;
    return MTV_OK;
}

// im2col gather tables: source token per (tap, output token) within one batch element, -1 = pad.
static void plane_geom(const Level& l, int p, int& h, int& w, int& off) {
    if (p == 0) { h = l.r; w = l.r; off = 0; }
    else if (p == 1) { h = l.t; w = l.r; off = l.b1; }
    else { h = l.t; w = l.r; off = l.b2; }
}

static std::vector<int> make_gather3(const Level& dst, const Level& src, bool up) {
    std::vector<int> g((size_t)9 * dst.L, -1);
    for (int p = 0; p < 3; ++p) {
        int h, w, off, hs, ws, offs;
        plane_geom(dst, p, h, w, off);
        plane_geom(src, p, hs, ws, offs);
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x)
                for (int ky = 0; ky < 3; ++ky)
                    for (int kx = 0; kx < 3; ++kx) {
                        const int yy = y + ky - 1, xx = x + kx - 1;
                        if (yy < 0 || yy >= h || xx < 0 || xx >= w) continue;
                        const int sy = up ? yy >> 1 : yy, sx = up ? xx >> 1 : xx;
                        g[(size_t)(ky * 3 + kx) * dst.L + off + y * w + x] = offs + sy * ws + sx;
                    }
    }
    return g;
}

static std::vector<int> make_gather_up1(const Level& dst, const Level& src) {
    std::vector<int> g((size_t)dst.L, -1);
    for (int p = 0; p < 3; ++p) {
        int h, w, off, hs, ws, offs;
        plane_geom(dst, p, h, w, off);
        plane_geom(src, p, hs, ws, offs);
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x) g[off + y * w + x] = offs + (y >> 1) * ws + (x >> 1);
    }
    return g;
}

static std::vector<Level> make_levels(int res, int frames, int n_levels) {
    std::vector<Level> lv;
    for (int l = 0; l < n_levels; ++l) {
        Level v;
        v.r = res >> l;
        v.t = frames >> l;
        v.b1 = v.r * v.r;
        v.b2 = v.b1 + v.t * v.r;
        v.L = v.b2 + v.t * v.r;
        lv.push_back(v);
    }
    return lv;
}

// Does geo_source() (what the kernels evaluate) reproduce a gather table (the straightforward construction
// above)?  `src` = the level the table reads from (the next level for the upsampling tables).
static bool table_matches_formula(const std::vector<int>& h, int ntaps, bool upm, const Level& lv, const Level& src) {
    for (int t = 0; t < ntaps; ++t)
        for (int tok = 0; tok < lv.L; ++tok) {
            const int g = ntaps == 9 ? geo_source(lv.r, lv.t, tok, t / 3, t % 3, upm) : geo_source(lv.r, lv.t, tok, 1, 1, upm);
            const int want = h[(size_t)t * lv.L + tok];
            if ((g < 0 ? -1 : (g & 0x0FFFFFFF)) != want) return false;
            if (g >= 0) {                            // plane bits must equal the source segment of the token
                const int st = g & 0x0FFFFFFF, pl = st >= src.b2 ? 2 : (st >= src.b1 ? 1 : 0);
                if ((g >> 28) != pl) return false;
            }
        }
    return true;
}

// =====================================================================================
// plan builder
// =====================================================================================
static int g_deep_mode = -1;       // mtv_debug_deep: -1 = MTV_DEEP / default (on), 0 = every conv on k_conv, 1 = on
static int g_resident_override = 0; // mtv_debug_resident_cus: > 0 = contexts created afterwards plan for that many co-resident CUs
static int g_deep_opts = -1;       // mtv_debug_deep_options: -1 = the MTV_DEEP_* environment / defaults, else a bit mask (MTV_DEEP_OPT_*)
static bool deep_opt(int bit, const char* env, bool dflt) {
    if (g_deep_opts >= 0) return (g_deep_opts & bit) != 0;
    const char* e = getenv(env);
    return e ? atoi(e) != 0 : dflt;
}
static bool deep_forced_off();     // (a forced legacy tile -- testing aids below -- keeps every conv on the kernel under test)

namespace {

struct Builder {
    mtv_ctx* c;
    Plan* plan;
    int B;
    int mode;
    const mtv_config& f;
    int emb;
    float* film_out;     // FORWARD: [maxB][film_total]; sampler step: [film_total] shared by every clip of the call
    int film_stride;     // floats between batch elements of film_out (0 in a sampler step)
    std::string err;
    std::map<const float*, std::shared_ptr<ConvOp>> producer;   // tensor -> the conv that writes it
    std::map<const float*, StatSink> fin_producer;                 // plain copy of a deep tensor -> where its statistics targets go
    std::map<const float*, std::shared_ptr<DeepOp>> deep_producer; // deep tensor (slab 0) -> the conv that writes it
    std::map<const float*, std::shared_ptr<DeepAttnOp>> attn_producer;   // ... or the fused attention block
    std::map<const float*, Tens> fin_cache;                        // deep tensor (slab 0) -> its plain copy, made once

    Builder(mtv_ctx* ctx, Plan* p, int batch, int md)
        : c(ctx), plan(p), B(batch), mode(md), f(ctx->cfg), emb(ctx->emb_dim), film_out(nullptr), film_stride(0) {}

    void push(const std::string& name, std::function<hipError_t(hipStream_t)> fn, double flops = 0.0, double bytes = 0.0) {
        Op op;
        op.run = std::move(fn);
        op.name = name;
        op.flops = flops;
        op.bytes = bytes;
        plan->ops.push_back(std::move(op));
    }

    // ---- weights ----
    // conv weight (3x3 or 1x1 / conv1d) repacked to [tap][Cin][ld]; `dst`/`ld` given when it shares a fused buffer
    void wconv(const std::string& key, int N, int C, int kh, int kw, bool conv1d, float* dst, int ld) {
        std::vector<int64_t> shape;
        if (conv1d) shape = {N, C, 1};
        else shape = {N, C, kh, kw};
        c->slot(key, shape, ROLE_CONV, dst, ld);
    }
    static int pad64(int n) { return (n + 63) / 64 * 64; }

    void account_conv(const ConvArgs& a) {
        if (!c->accounting) return;
        const double m = (double)a.Lout;   // per batch element
        c->work.flops_conv3x3 += a.ntaps == 9 ? 2.0 * m * a.N * 9.0 * a.Cmain : 0.0;
        c->work.flops_1x1 += (a.ntaps == 1 ? 2.0 * m * a.N * a.Cmain : 0.0) + 2.0 * m * a.N * a.Cskip;
        const double wbytes = 4.0 * ((double)a.ntaps * a.Cmain + a.Cskip) * a.N + 4.0 * a.N;
        if (a.ntaps == 9) {
            c->work.bytes_weights_conv += wbytes;
            // activations of the fused conv/GN path: read the tapped source once, write the output once
            c->work.bytes_act_conv_path += 4.0 * ((double)a.Lsrc * a.Cmain + (double)a.Lout * a.N + (double)a.Lskip * a.Cskip +
                                                  (a.res ? (double)a.Lout * a.N : 0.0));
        } else {
            c->work.bytes_weights_other += wbytes;
        }
    }

    static std::string conv_tag(const ConvArgs& a, const ConvTile& t) {
        char tag[80];
        snprintf(tag, sizeof tag, "[%dx%d k%d t%d,%d,%d,%d%s]", a.Lout, a.N, a.ntaps * a.Cmain + a.Cskip, t.MT, t.NT, t.NW, t.KS, t.XM ? "x" : "");   // (NW 32: k_conv_lds, NW 64: k_lin)
        return tag;
    }

    void add_conv(ConvArgs a0, const std::string& name, int lvl_out) {
        a0.B = B;
        a0.seg_out = c->lv[lvl_out].seg();
        a0.stat_cstride = (unsigned)c->stats_copy_doubles;
        if (mode == MODE_STEP0 || mode == MODE_STEP1)      // (timing experiment MTV_DEBUG_STATDUP, launch_conv: where the other parity's arena lies)
            a0.dbg_stat_dup = (mode == MODE_STEP0 ? 1 : -1) * (long long)(c->stats_bytes / sizeof(double));
        static const bool geo_env = []() { const char* e = getenv("MTV_GEO"); return !e || atoi(e) != 0; }();
        if (c->geo_ok && geo_env) {                 // gather tables -> arithmetic (checked equal in mtv_create)
            const int lo = lvl_out;
            a0.geo_r = c->lv[lo].r;
            a0.geo_t = c->lv[lo].t;
            if (a0.gather && a0.ntaps == 9) a0.geo_main = a0.gather == c->g3[lo] ? 1 : (a0.gather == c->gup3[lo] ? 2 : 0);
            if (a0.gather_skip) a0.geo_skip = a0.gather_skip == c->gup1[lo] ? 2 : 0;
        }
        const int nchunks = a0.ntaps * (a0.Cmain / 16) + a0.Cskip / 16;
        if (a0.gn.sums) {      // reciprocals the kernel multiplies by (GroupNorm: 1/gs, 1/(tokens x gs) per plane / overall)
            const SegInfo& sg = a0.seg_src;
            const double gs = (double)a0.gn.gs;
            a0.gn.inv_gs = 1.0f / (float)a0.gn.gs;
            a0.gn.inv_n[0] = 1.0 / ((double)sg.b1 * gs);
            a0.gn.inv_n[1] = 1.0 / ((double)(sg.b2 - sg.b1) * gs);
            a0.gn.inv_n[2] = 1.0 / ((double)(sg.L - sg.b2) * gs);
            a0.gn.inv_n[3] = 1.0 / ((double)sg.L * gs);
        }
        if (x3_wanted((long)B * a0.Lout))          // split-bf16 copy of the weights for the large-token-count kernel k_conv_x3
            a0.W3 = c->w3_for(a0.W, a0.ntaps * a0.Cmain + a0.Cskip, a0.ldw, &a0.w3_plane);
        account_conv(a0);
        static const bool stamps_env = getenv("MTV_STAMPS") != nullptr;      // diagnostic build only (mtv_debug_stamps)
        if (stamps_env) a0.dbg = reinterpret_cast<unsigned long long*>(c->buf("dbg." + name + "." + std::to_string(mode) + "." + std::to_string(B), 128));
        auto op = std::make_shared<ConvOp>();
        op->a = a0;
        op->t = conv_pick_tile(B, a0.Lout, a0.N, nchunks, a0.Cmain, a0.gn.sums != nullptr);
        force_lds_tile(a0, &op->t);
        op->base_name = std::string(a0.ntaps == 9 ? "conv3:" : "conv1:") + name;
        op->op_index = (int)plan->ops.size();
        producer[a0.out] = op;
        plan->convs.push_back(op);
        const double K = (double)a0.ntaps * a0.Cmain + a0.Cskip;
        const double flops = 2.0 * B * a0.Lout * a0.N * K;
        const double bytes = 4.0 * (K * a0.N + a0.N) +
                             4.0 * B * ((double)a0.Lsrc * a0.Cmain + (double)a0.Lskip * a0.Cskip + (double)a0.Lout * a0.N + (a0.res ? (double)a0.Lout * a0.N : 0.0));
        push(op->base_name + conv_tag(op->a, op->t), [op](hipStream_t s) { return launch_conv(op->a, op->t, s); }, flops, bytes);
    }


    // ---- deep levels (deep.hip): K-sliced convs whose consumers add the partial slabs and compute the GroupNorm statistics ----
    bool deep_on(int lvl) const {
        static const bool env = []() { const char* e = getenv("MTV_DEEP"); return !e || atoi(e) != 0; }();
        const bool on = g_deep_mode < 0 ? env : g_deep_mode != 0;
        return on && !deep_forced_off() && B <= 2 && c->lv[lvl].L <= 128;
    }
    int deep_clips() const { return c->cfg.max_batch < 2 ? c->cfg.max_batch : 2; }     // clips per slab: the deep path only runs with B <= 2
    static DeepSrc dsrc(const Tens& t) { return DeepSrc{t.p, t.slab, t.ks, t.C}; }
    static std::string deep_tag(const DeepArgs& a, const DeepTile& t) {
        char tag[96];
        snprintf(tag, sizeof tag, "[%dx%d k%d d%d,%d,%d,%d]", a.Lout, a.N, a.ntaps * a.Cmain + a.Cskip, t.RT, t.NT, a.KS, a.nrg);
        return tag;
    }
    // common fields of a deep conv at output level `lvl`
    DeepArgs deep_args(int lvl, int ntaps, int N, const float* bias) {
        DeepArgs a{};
        a.ntaps = ntaps;
        a.r = c->lv[lvl].r;
        a.t = c->lv[lvl].t;
        a.B = B;
        a.Lout = a.Lsrc = a.Lres = c->lv[lvl].L;
        a.N = N;
        a.bias = bias;
        a.zeros = c->buf("deep.zeros", 8192);
        return a;
    }
    // emit a configured deep conv (deep_configure() succeeded): output slabs, deep-layout weights, the op
    Tens emit_deep(DeepArgs a, DeepTile t, const float* W, int ldw, const std::string& name, int lvl_out) {
        Tens out;
        out.lvl = lvl_out;
        out.C = a.N;
        out.ks = a.KS;
        out.slab = (unsigned)((size_t)deep_clips() * c->lv[lvl_out].L * a.N);
        // (8 slabs whatever this plan picked: plans of other batch sizes share the buffer and may slice differently)
        out.p = c->buf("act.deep." + name, (size_t)8 * out.slab);
        c->taps[name] = {lvl_out, a.N};
        c->bufs["tap." + name] = out.p;
        c->tap_slabs[name] = {out.ks, out.slab};
        a.out = out.p;
        a.out_slab_stride = out.slab;
        a.W = c->wdeep_for(W, ldw, a, t.NT);
        {   // row table: one per (geometry, slicing) -- shared by every conv of the context that has the same
            char key[128];
            snprintf(key, sizeof key, "deep.rowtab %d %d %d %d %d %d %d %d %d %d", a.r, a.t, a.up_main + 2 * a.pool_main, a.ntaps, a.nrg, t.RT, a.CSm, a.CSs, a.Cskip ? 1 : 0, a.Lout);
            if (!c->bufs.count(key)) {
                const std::vector<int> tab = deep_rowtab(a, t);
                float* d = c->buf(key, tab.size());
                if (d && hipMemcpy(d, tab.data(), tab.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) {
                    c->bufs.erase(key);          // (the zero-filled allocation stays owned by the context; nobody may find it under this key)
                    c->buf_floats.erase(key);
                    d = nullptr;
                }
                if (!d) { err = "row table upload failed at " + name; return Tens{}; }
            }
            a.rowtab = reinterpret_cast<const int*>(c->bufs[key]);
        }
        if (!a.W || !out.p || !a.zeros) { err = "deep conv allocation failed at " + name; return Tens{}; }
        if (c->accounting) {
            const double m = (double)a.Lout;
            c->work.flops_conv3x3 += a.ntaps == 9 ? 2.0 * m * a.N * 9.0 * a.Cmain : 0.0;
            c->work.flops_1x1 += (a.ntaps == 1 ? 2.0 * m * a.N * a.Cmain : 0.0) + 2.0 * m * a.N * a.Cskip;
            const double wbytes = 4.0 * ((double)a.ntaps * a.Cmain + a.Cskip) * a.N + 4.0 * a.N;
            if (a.ntaps == 9) {
                c->work.bytes_weights_conv += wbytes;
                c->work.bytes_act_conv_path += 4.0 * ((double)a.Lsrc * a.Cmain + (double)a.Lout * a.N + (double)a.Lout * a.Cskip + (a.res.p ? (double)a.Lout * a.N : 0.0));
            } else {
                c->work.bytes_weights_other += wbytes;
            }
        }
        auto op = std::make_shared<DeepOp>();
        op->a = a;
        op->t = t;
        const double K = (double)a.ntaps * a.Cmain + a.Cskip;
        const double flops = 2.0 * B * a.Lout * a.N * K;
        const double bytes = 4.0 * (K * a.N + a.N) + 4.0 * B * ((double)a.Lsrc * a.Cmain + (double)a.Lout * a.Cskip + (double)a.Lout * a.N + (a.res.p ? (double)a.Lout * a.N : 0.0));
        push(std::string(a.ntaps == 9 ? "conv3:" : "conv1:") + name + deep_tag(a, t), [op](hipStream_t s) { return launch_deep_conv(op->a, op->t, s); }, flops, bytes);
        if (out.ks > 1) deep_producer[out.p] = op;
        return out;
    }
    // a plain copy of a deep tensor for a consumer outside the deep region (k_conv, k_pool_down): made once per tensor
    Tens materialize(const Tens& x, const std::string& name) {
        if (x.ks <= 1) return x;
        auto it = fin_cache.find(x.p);
        if (it != fin_cache.end()) return it->second;
        Tens o;
        o.lvl = x.lvl;
        o.C = x.C;
        o.p = c->act(name + ".fin", x.lvl, x.C);
        // Default: a separate k_deep_finalize pass (4.6 us + a kernel boundary).  MTV_DEEP_INLAUNCH=1: completion inside the
        // producing kernel instead (its last-arriving K slice re-reads the slabs and writes the plain tensor + statistics; consumers
        // emitted so far keep reading the slabs) -- parity-green, and measured no faster: the store -> drain -> ticket -> re-read
        // chain adds 4-4.5 us to the producing launch (conv3 at 32 tokens 8.3 -> 10.1 us, at 128 tokens 15.1 -> 17.1, the
        // attention block 10.6 -> 14.8), the same three dependent round trips as k_conv's split-K completion.
        // Round 5 default: completion inside the producing kernel by data-tagged granules (DeepFin::tagged: no drain, no ticket wait -- K slice 0
        // polls the other slices' partial tiles; profiles/r05_tagged_fin_ab.txt).  MTV_DEEP_FIN_PASS=1 / MTV_DEEP_OPT_FIN_PASS: round 4's pass.
        const bool ticket_inlaunch = deep_opt(MTV_DEEP_OPT_INLAUNCH, "MTV_DEEP_INLAUNCH", false);
        const bool tagged_inlaunch = !ticket_inlaunch && !deep_opt(MTV_DEEP_OPT_FIN_PASS, "MTV_DEEP_FIN_PASS", false);
        DeepFin* fn = nullptr;
        std::shared_ptr<void> owner;
        long ntile = 0, nwg = 0;
        int nslices = 0, lastN = 0;
        bool tile_ok = true;
        if (ticket_inlaunch || tagged_inlaunch) {
            auto dp = deep_producer.find(x.p);
            auto ap = attn_producer.find(x.p);
            if (dp != deep_producer.end()) {
                fn = &dp->second->a.fin; owner = dp->second; ntile = (long)B * dp->second->a.nrg * dp->second->a.tiles_n;
                nslices = dp->second->a.KS; nwg = ntile * nslices; lastN = dp->second->a.N; tile_ok = dp->second->t.NT == 1;
            } else if (ap != attn_producer.end()) {
                fn = &ap->second->a.fin; owner = ap->second; ntile = (long)B * ap->second->a.nqg * ap->second->a.ncg;
                nslices = ap->second->a.nhg; nwg = ntile * nslices; lastN = ap->second->a.C;
            }
            // (the polling slice waits for workgroups of the same launch: all of them must be resident together)
            // -> grid <= the CUs this context may count on (mtv_ctx::resident_cus); otherwise the finalize pass
            if (fn && tagged_inlaunch && (!tile_ok || nwg > std::min(256, c->resident_cus) || nslices < 2 || nslices > 8)) fn = nullptr;
        }
        if (fn) {
            fn->out = o.p;
            fn->nstat = 0;
            fn->stat_cstride = (unsigned)c->stats_copy_doubles;
            fn->seg = c->lv[x.lvl].seg();
            if (tagged_inlaunch) {
                const size_t gf = (size_t)(nslices - 1) * B * c->lv[x.lvl].L * lastN * 2;        // 8-byte granules
                const std::string okey = name + ".B" + std::to_string(B) + ".S" + std::to_string(nslices);
                fn->tagged = 1;
                fn->gran = c->buf("deep.fin.gran." + okey, gf);                                  // (per op: tags are only unique per counter)
                fn->gran_bytes = (unsigned)(gf * 4);
                fn->ecnt = reinterpret_cast<unsigned long long*>(c->buf("deep.fin.ecnt." + okey, (size_t)ntile * 2));
                fn->fault = c->fault_d;
                if (!fn->gran || !fn->ecnt || !fn->fault) { err = "granule scratch allocation failed at " + name; return Tens{}; }
            } else {
                fn->tickets = reinterpret_cast<int*>(c->buf("deep.tickets." + name + ".B" + std::to_string(B), (size_t)ntile));
                if (!fn->tickets) { err = "ticket allocation failed at " + name; return Tens{}; }
            }
            fin_producer[o.p] = StatSink{fn->stat, &fn->nstat, owner};
            fin_cache[x.p] = o;
            return o;
        }
        auto op = std::make_shared<FinOp>();
        op->a.src = dsrc(x);
        op->a.out = o.p;
        op->a.B = B;
        op->a.L = c->lv[x.lvl].L;
        op->a.seg = c->lv[x.lvl].seg();
        op->a.stat_cstride = (unsigned)c->stats_copy_doubles;
        fin_producer[o.p] = StatSink{op->a.stat, &op->a.nstat, op};
        fin_cache[x.p] = o;
        char tag[64];
        snprintf(tag, sizeof tag, "[%dx%d ks%d]", op->a.L, x.C, x.ks);
        push("fin:" + name + tag, [op](hipStream_t s) { return launch_deep_finalize(op->a, s); }, 0.0, 4.0 * B * (double)op->a.L * x.C * (x.ks + 1));
        return o;
    }

    // GroupNorm site over the channel concatenation of `parts`: the statistics are accumulated by the
    // epilogues of the convs that produce the parts; a standalone pass is only the fallback.
    void add_stats(const std::vector<Tens>& parts, int lvl, double* site) {
        int ct = 0;
        for (auto& p : parts) ct += p.C;
        bool fused = true;
        for (auto& p : parts) {
            auto it = producer.find(p.p);
            auto fi = fin_producer.find(p.p);
            if (fi != fin_producer.end()) { if (*fi->second.nstat >= 2) fused = false; }
            else if (it == producer.end() || it->second->a.nstat >= 2 || it->second->a.out_cm) fused = false;
        }
        if (fused) {
            int coff = 0;
            for (auto& p : parts) {
                auto fi = fin_producer.find(p.p);
                if (fi != fin_producer.end()) {
                    fi->second.stat[(*fi->second.nstat)++] = StatOut{site, ct / 32, coff, 1.0f / (float)(ct / 32)};
                } else {
                    ConvArgs& pa = producer[p.p]->a;
                    pa.stat[pa.nstat++] = StatOut{site, ct / 32, coff, 1.0f / (float)(ct / 32)};
                }
                coff += p.C;
            }
            return;
        }
        add_stats_pass(parts, lvl, site);
    }

    void add_stats_pass(const std::vector<Tens>& parts, int lvl, double* site) {
        StatsArgs a{};
        a.nparts = (int)parts.size();
        int ct = 0;
        for (int i = 0; i < a.nparts; ++i) {
            a.src[i] = parts[i].p;
            a.C[i] = parts[i].C;
            ct += parts[i].C;
        }
        a.Ctot = ct;
        a.gs = ct / 32;
        a.B = B;
        a.seg = c->lv[lvl].seg();
        a.sums = site;
        push("gn_stats", [a](hipStream_t s) { return launch_gn_stats(a, s); });
    }

    float* gnvec(const std::string& key, int C) { return c->wcopy(key, {C}); }

    // ---- ResBlock (unet.py:93-207) ----
    Tens resblock(const std::vector<Tens>& x, const Layer& ly, const std::string& nm) {
        const ResDesc& r = ly.res;
        const int lvl_in = x[0].lvl;
        const int lvl_out = r.updown == 1 ? lvl_in + 1 : (r.updown == 2 ? lvl_in - 1 : lvl_in);
        const Level& Lo = c->lv[lvl_out];
        const Level& Li = c->lv[lvl_in];
        const std::string P = ly.pre;
        int cin = 0;
        for (auto& t : x) cin += t.C;
        if (cin != r.cin) { err = "resblock channel mismatch at " + P; return Tens{}; }
        if ((cin % 32) || (r.cout % 32) || (cin % 16) || (r.cout % 16)) { err = "channels must be multiples of 32 at " + P; return Tens{}; }

        float* g1 = gnvec(P + "in_layers.0.weight", cin);
        float* b1 = gnvec(P + "in_layers.0.bias", cin);
        float* cb1 = c->wcopy(P + "in_layers.2.bias", {r.cout});
        const int ld1 = pad64(r.cout);
        float* W1 = c->buf("w." + P + "in_layers.2.weight", (size_t)9 * cin * ld1);
        wconv(P + "in_layers.2.weight", r.cout, cin, 3, 3, false, W1, ld1);
        const int fdim = f.use_scale_shift_norm ? 2 * r.cout : r.cout;
        // emb_layers.1 lives inside the concatenated FiLM matrix
        c->slot(P + "emb_layers.1.weight", {fdim, emb}, ROLE_COPY, c->bufs["w.film"] + (size_t)r.film_off * emb, 0);
        c->slot(P + "emb_layers.1.bias", {fdim}, ROLE_COPY, c->bufs["w.film_bias"] + r.film_off, 0);
        float* g2 = gnvec(P + "out_layers.0.weight", r.cout);
        float* b2 = gnvec(P + "out_layers.0.bias", r.cout);
        float* cb2 = c->wcopy(P + "out_layers.3.bias", {r.cout});
        const bool has_skip_conv = cin != r.cout;
        const int ld2 = pad64(r.cout);
        float* W2 = c->buf("w." + P + "out_layers.3.weight", (size_t)(9 * r.cout + (has_skip_conv ? cin : 0)) * ld2);
        wconv(P + "out_layers.3.weight", r.cout, r.cout, 3, 3, false, W2, ld2);
        float* sb = nullptr;
        if (has_skip_conv) {
            wconv(P + "skip_connection.weight", r.cout, cin, 1, 1, false, W2 + (size_t)9 * r.cout * ld2, ld2);
            sb = c->wcopy(P + "skip_connection.bias", {r.cout});
        }

        if (deep_on(lvl_out) && !(r.updown == 1 && x.size() != 1) && !(has_skip_conv && r.updown) && (has_skip_conv || x.size() == 1)) {
            // ---- deep level: both convs K-sliced, partial slabs summed by their consumers (deep.hip)
            const bool down = r.updown == 1, up = r.updown == 2;
            // ResBlock(down=True) (unet.py:179-184, Downsample = AvgPool2d :594): round 5 folds both 2x2 means into the two convs -- conv1
            // stages the finer level's rows, applies GroupNorm / SiLU there and pools in LDS; conv2's residual is the pooled raw input, shared
            // out over its K slices.  No k_pool_down launch, no plain copy of the input, no statistics site.  MTV_DEEP_POOL_FOLD=0: round 4's form.
            static const bool fold_env = []() { const char* e = getenv("MTV_DEEP_POOL_FOLD"); return !e || atoi(e) != 0; }();
            const bool fold = down && fold_env && x.size() == 1 && !has_skip_conv && Li.r == 2 * Lo.r && Li.t == 2 * Lo.t;     // (even planes only)
            DeepArgs a1 = deep_args(lvl_out, 9, r.cout, cb1);
            a1.Cmain = cin;
            if (fold) {
                a1.main[0] = dsrc(x[0]);
                a1.pool_main = 1;
                a1.Lsrc = Li.L;
                a1.gn = 1; a1.act = 1; a1.gs = cin / 32; a1.gamma = g1; a1.beta = b1;
            } else if (down) {
                a1.main[0] = DeepSrc{a1.zeros, 0, 1, cin};                 // (placeholder for the pooled, already normalised input)
            } else {
                for (size_t i = 0; i < x.size(); ++i) a1.main[i] = dsrc(x[i]);
                a1.up_main = up ? 1 : 0;
                a1.Lsrc = Li.L;
                a1.gn = 1; a1.act = 1; a1.gs = cin / 32; a1.gamma = g1; a1.beta = b1;
            }
            if (!f.use_scale_shift_norm) { a1.bias_b = film_out + r.film_off; a1.bias_b_stride = film_stride; }
            DeepArgs a2 = deep_args(lvl_out, 9, r.cout, cb2);
            a2.Cmain = r.cout;
            a2.main[0] = DeepSrc{a2.zeros, 0, 8, r.cout};                  // (placeholder for h1)
            a2.gn = 1; a2.act = 1; a2.gs = r.cout / 32; a2.gamma = g2; a2.beta = b2;
            if (f.use_scale_shift_norm) { a2.film = film_out + r.film_off; a2.film_stride = film_stride; }
            if (has_skip_conv) {
                for (size_t i = 0; i < x.size(); ++i) a2.skip[i] = dsrc(x[i]);
                a2.Cskip = cin;
                a2.bias2 = sb;
            } else if (fold) {
                a2.res = dsrc(x[0]);
                a2.pool_res = 1;
                a2.Lres = Li.L;
            } else if (down) {
                a2.res = DeepSrc{a2.zeros, 0, 1, r.cout};                  // (placeholder for the pooled x)
            } else {
                a2.res = dsrc(x[0]);
                a2.up_res = up ? 1 : 0;
                a2.Lres = Li.L;
            }
            DeepTile t1{}, t2{};
            if (deep_configure(a1, &t1) && deep_configure(a2, &t2)) {
                if (down && !fold) {
                    const Tens x0 = materialize(x[0], nm + ".x");         // k_pool_down reads a plain tensor + its statistics
                    double* site1 = c->new_site();
                    add_stats({x0}, lvl_in, site1);
                    Tens pa, px;
                    pa.lvl = lvl_out; pa.C = cin; pa.p = c->act(nm + ".pool_act", lvl_out, cin);
                    px.lvl = lvl_out; px.C = cin; px.p = c->act(nm + ".pool_x", lvl_out, cin);
                    PoolArgs pl{};
                    pl.x = x0.p; pl.out_act = pa.p; pl.out_x = px.p; pl.sums = site1; pl.cstride = (unsigned)c->stats_copy_doubles; pl.gamma = g1; pl.beta = b1;
                    pl.B = B; pl.C = cin; pl.gs = cin / 32; pl.seg_src = Li.seg(); pl.seg_dst = Lo.seg(); pl.r_dst = Lo.r; pl.t_dst = Lo.t;
                    push("pool_down", [pl](hipStream_t s) { return launch_pool_down(pl, s); });
                    a1.main[0] = dsrc(pa);
                    a2.res = dsrc(px);
                }
                const Tens h1d = emit_deep(a1, t1, W1, ld1, nm + ".h1", lvl_out);
                if (!err.empty()) return Tens{};
                a2.main[0] = dsrc(h1d);
                return emit_deep(a2, t2, W2, ld2, nm + ".out", lvl_out);
            }
        }
        {   // k_conv / k_pool_down read plain tensors: inputs that come out of the deep region are summed up first
            std::vector<Tens> xl;
            for (size_t i = 0; i < x.size(); ++i) xl.push_back(materialize(x[i], nm + ".in" + std::to_string(i)));
            return resblock_legacy(xl, ly, nm, W1, ld1, W2, ld2, g1, b1, cb1, g2, b2, cb2, sb);
        }
    }

    Tens resblock_legacy(const std::vector<Tens>& x, const Layer& ly, const std::string& nm, float* W1, int ld1, float* W2, int ld2, float* g1, float* b1,
                         float* cb1, float* g2, float* b2, float* cb2, float* sb) {
        const ResDesc& r = ly.res;
        const int lvl_in = x[0].lvl;
        const int lvl_out = r.updown == 1 ? lvl_in + 1 : (r.updown == 2 ? lvl_in - 1 : lvl_in);
        const Level& Lo = c->lv[lvl_out];
        const Level& Li = c->lv[lvl_in];
        const std::string P = ly.pre;
        int cin = 0;
        for (auto& t : x) cin += t.C;
        const bool has_skip_conv = cin != r.cout;
        // GN1 statistics of x (per plane)
        double* site1 = c->new_site();
        add_stats(x, lvl_in, site1);

        Tens h1;
        h1.lvl = lvl_out;
        h1.C = r.cout;
        h1.p = c->act(nm + ".h1", lvl_out, r.cout);
        Tens px;   // pooled x (down only)
        ConvArgs a{};
        a.ntaps = 9;
        a.Lout = Lo.L;
        a.N = r.cout;
        a.W = W1;
        a.ldw = ld1;
        a.bias = cb1;
        a.out = h1.p;
        a.Cmain = cin;
        if (!f.use_scale_shift_norm) {        // h = h + emb_out (unet.py:205)
            a.bias_b = film_out + r.film_off;
            a.bias_b_stride = film_stride;
        }
        if (r.updown == 1) {
            if (x.size() != 1) { err = "down block with concat input"; return Tens{}; }
            Tens pa;
            pa.lvl = lvl_out; pa.C = cin; pa.p = c->act(nm + ".pool_act", lvl_out, cin);
            px.lvl = lvl_out; px.C = cin; px.p = c->act(nm + ".pool_x", lvl_out, cin);
            PoolArgs pl{};
            pl.x = x[0].p; pl.out_act = pa.p; pl.out_x = px.p; pl.sums = site1; pl.cstride = (unsigned)c->stats_copy_doubles; pl.gamma = g1; pl.beta = b1;
            pl.B = B; pl.C = cin; pl.gs = cin / 32; pl.seg_src = Li.seg(); pl.seg_dst = Lo.seg(); pl.r_dst = Lo.r; pl.t_dst = Lo.t;
            push("pool_down", [pl](hipStream_t s) { return launch_pool_down(pl, s); });
            a.nmain = 1; a.src[0] = pa.p; a.C[0] = cin;
            a.gather = c->g3[lvl_out];
            a.Lsrc = Lo.L;
            a.seg_src = Lo.seg();
        } else {
            a.nmain = (int)x.size();
            for (int i = 0; i < a.nmain; ++i) { a.src[i] = x[i].p; a.C[i] = x[i].C; }
            a.gather = r.updown == 2 ? c->gup3[lvl_out] : c->g3[lvl_out];
            a.Lsrc = Li.L;
            a.seg_src = Li.seg();
            a.gn = GnIn{site1, g1, b1, nullptr, 0, cin / 32, 0, 1, (unsigned)c->stats_copy_doubles};
        }
        add_conv(a, nm + ".conv1", lvl_out);

        // GN2 statistics of h1
        double* site2 = c->new_site();
        add_stats({h1}, lvl_out, site2);

        Tens out;
        out.lvl = lvl_out;
        out.C = r.cout;
        out.p = c->act(nm + ".out", lvl_out, r.cout);
        ConvArgs d{};
        d.ntaps = 9;
        d.Lout = Lo.L;
        d.Lsrc = Lo.L;
        d.N = r.cout;
        d.W = W2;
        d.ldw = ld2;
        d.bias = cb2;
        d.out = out.p;
        d.nmain = 1;
        d.src[0] = h1.p;
        d.C[0] = r.cout;
        d.Cmain = r.cout;
        d.gather = c->g3[lvl_out];
        d.seg_src = Lo.seg();
        d.gn = GnIn{site2, g2, b2, f.use_scale_shift_norm ? film_out + r.film_off : nullptr, film_stride, r.cout / 32, 0, 1, (unsigned)c->stats_copy_doubles};
        if (has_skip_conv) {
            if (r.updown) { err = "up/down block with skip conv"; return Tens{}; }
            d.nskip = (int)x.size();
            for (int i = 0; i < d.nskip; ++i) { d.src[2 + i] = x[i].p; d.C[2 + i] = x[i].C; }   // slots 2..3 = skip parts
            d.Cskip = cin;
            d.Lskip = Li.L;
            d.bias2 = sb;
        } else {
            if (x.size() != 1) { err = "identity skip with concat input"; return Tens{}; }
            if (r.updown == 1) { d.res = px.p; d.Lskip = Lo.L; }
            else if (r.updown == 2) { d.res = x[0].p; d.Lskip = Li.L; d.gather_skip = c->gup1[lvl_out]; }
            else { d.res = x[0].p; d.Lskip = Lo.L; }
        }
        add_conv(d, nm + ".conv2", lvl_out);
        return out;
    }

    // ---- AttentionBlock / AttentionBlock1D (unet.py:210-300) ----
    AttnArgs attn_args(const float* qkv, float* att, const Level& L, int C, int H, int d, bool whole) {
        AttnArgs t{};
        t.qkv = qkv; t.out = att; t.B = B; t.L = L.L; t.C = C; t.H = H;
        t.scale = 1.0f / std::sqrt(std::sqrt((float)d));
        if (whole) {
            t.nseg = 1; t.seg_start[0] = 0; t.seg_len[0] = L.L;
        } else {
            t.nseg = 3;
            t.seg_start[0] = 0; t.seg_len[0] = L.b1;
            t.seg_start[1] = L.b1; t.seg_len[1] = L.b2 - L.b1;
            t.seg_start[2] = L.b2; t.seg_len[2] = L.L - L.b2;
        }
        t.blk_prefix[0] = 0;
        for (int i = 0; i < t.nseg; ++i) t.blk_prefix[i + 1] = t.blk_prefix[i] + (t.seg_len[i] + 63) / 64;
        return t;
    }
    void push_attention(AttnArgs t, const std::string& nm, const Level& L, int C, int d, bool whole) {
        const int H = t.H;
        if (c->accounting)
            for (int i = 0; i < t.nseg; ++i) c->work.flops_attn_core += 4.0 * H * (double)t.seg_len[i] * t.seg_len[i] * d;
        double aflops = 0.0;
        for (int i = 0; i < t.nseg; ++i) aflops += 4.0 * B * H * (double)t.seg_len[i] * t.seg_len[i] * d;
        char tag[64];
        snprintf(tag, sizeof tag, "[L%d d%d %s]", L.L, d, whole ? "1d" : "2d");      // (which core runs is decided at launch: launch_attention)
        static const bool att_stamps = getenv("MTV_STAMPS") != nullptr;       // diagnostic build only (mtv_debug_stamps)
        if (att_stamps) {
            t.dbg = reinterpret_cast<unsigned long long*>(c->buf("dbg.attn." + nm + "." + std::to_string(mode) + "." + std::to_string(B), 128));
            plan->attn_dbg.emplace_back("attn:" + nm + tag, t.dbg);
        }
        push("attn:" + nm + tag, [t](hipStream_t s) { return launch_attention(t, s); }, aflops, 4.0 * B * L.L * 4.0 * C);
    }

    Tens attention(const Tens& x0, const std::string& P, bool whole, const std::string& nm) {
        const int C = x0.C, lvl = x0.lvl;
        const Level& L = c->lv[lvl];
        const int H = f.num_heads;
        if (C % H || (C / H) % 4 || C % 32) { err = "attention channels/heads unsupported at " + P; return Tens{}; }
        const int d = C / H;
        if (!(d == 4 || d == 8 || d == 16 || d == 32 || d == 48 || d == 64 || d == 128)) { err = "head dim unsupported at " + P; return Tens{}; }
        float* gw = gnvec(P + "norm.weight", C);
        float* gb = gnvec(P + "norm.bias", C);
        const int ldq = pad64(3 * C);
        float* Wq = c->buf("w." + P + "qkv.weight", (size_t)C * ldq);
        wconv(P + "qkv.weight", 3 * C, C, 1, 1, true, Wq, ldq);
        float* Wq_nk = c->buf("wnk." + P + "qkv.weight", (size_t)3 * C * C);      // the same weights as stored, [3C][C]: k_lin's operand
        c->slots[c->slot_index[P + "qkv.weight"]].dst2 = Wq_nk;
        float* Wq_pk = (C & 15) ? nullptr : c->buf("wpk." + P + "qkv.weight", (size_t)3 * C * C);      // ... and lane-linear: k_conv_pw's operand
        c->slots[c->slot_index[P + "qkv.weight"]].dst3 = Wq_pk;
        float* bq = c->wcopy(P + "qkv.bias", {3 * C});
        const int ldp = pad64(C);
        float* Wp = c->buf("w." + P + "proj_out.weight", (size_t)C * ldp);
        wconv(P + "proj_out.weight", C, C, 1, 1, true, Wp, ldp);
        float* Wp_nk = c->buf("wnk." + P + "proj_out.weight", (size_t)C * C);
        c->slots[c->slot_index[P + "proj_out.weight"]].dst2 = Wp_nk;
        float* Wp_pk = (C & 15) ? nullptr : c->buf("wpk." + P + "proj_out.weight", (size_t)C * C);
        c->slots[c->slot_index[P + "proj_out.weight"]].dst3 = Wp_pk;
        float* bp = c->wcopy(P + "proj_out.bias", {C});

        // ---- deep level (deep.hip): attention core + proj_out in ONE launch (k_deep_attn; the head groups are the K slices of the
        // projection, one output slab each).  It reads every channel of a head, so qkv must be one plain tensor: by default the
        // block's input is summed up once (k_deep_finalize, which also leaves the GroupNorm statistics) and qkv stays on k_conv
        // -- a K-sliced qkv conv (MTV_DEEP_QKV=1) stages ALL tokens of its channel slice per column tile and then needs its own
        // finalize pass: measured slower at both deep levels.
        DeepAttnArgs da{};
        da.B = B; da.L = L.L; da.C = C; da.H = H; da.r = L.r; da.t = L.t; da.whole = whole ? 1 : 0;
        da.scale = 1.0f / std::sqrt(std::sqrt((float)d));
        da.Wp = Wp; da.ldw = ldp; da.bias = bp;
        const bool fuse_env = !deep_opt(MTV_DEEP_OPT_NO_FUSED_ATTN, "MTV_DEEP_NO_ATTN", false);
        const bool deep_qkv_env = deep_opt(MTV_DEEP_OPT_SLICED_QKV, "MTV_DEEP_QKV", false);
        // ---- the whole block in ONE launch (block.hip, k_deep_block): a cluster of workgroups per (clip, head) reads the input's slabs,
        // computes its GroupNorm statistics, the head's q | k | v (K-sliced inside the cluster, two in-launch hand-offs), the attention and
        // the head's share of proj_out -> output slab `head`.  No finalize pass, no statistics site, no plain qkv tensor.
        // (the dataflow variants of round 4 describe the three-launch form: any of them selects it)
        const bool block_env = !deep_opt(MTV_DEEP_OPT_NO_BLOCK, "MTV_DEEP_NO_BLOCK", false) && fuse_env && !deep_qkv_env &&
                               !deep_opt(MTV_DEEP_OPT_UNSLICED_QKV, "MTV_DEEP_QKV1", false) && !deep_opt(MTV_DEEP_OPT_INLAUNCH, "MTV_DEEP_INLAUNCH", false);
        if (deep_on(lvl) && block_env) {
            DeepBlockArgs ba{};
            ba.x = dsrc(x0);
            ba.B = B; ba.L = L.L; ba.C = C; ba.H = H; ba.r = L.r; ba.t = L.t; ba.whole = whole ? 1 : 0;
            ba.scale = 1.0f / std::sqrt(std::sqrt((float)d));
            ba.gamma = gw; ba.beta = gb; ba.gs = C / 32;
            ba.Wq = Wq_nk; ba.bq = bq; ba.Wp = Wp_nk; ba.bp = bp;
            static const int force_cl = []() { const char* e = getenv("MTV_BLOCK_CL"); return e ? atoi(e) : 0; }();
            // Where it pays (profiles/r05_deep_block.txt: the block in one launch against fin + qkv + k_deep_attn of round 4, same graph
            // chains): 32 tokens 15.7 us against 21.1, [128 x 256] 21.2 against 26.8 -- but [128 x 512] 33 against 27.5: there the K slices'
            // partial q | k | v rows are 12.6 MB of hand-off traffic per block.  MTV_BLOCK_MAX_L / MTV_BLOCK_MAX_LC override the rule.
            static const int max_l = []() { const char* e = getenv("MTV_BLOCK_MAX_L"); return e ? atoi(e) : 32; }();
            static const long max_lc = []() { const char* e = getenv("MTV_BLOCK_MAX_LC"); return e ? atol(e) : 128L * 256; }();
            const bool pays = L.L <= max_l || (long)L.L * C <= max_lc || deep_opt(MTV_DEEP_OPT_BLOCK_ALL, "MTV_DEEP_BLOCK_ALL", false);
            static const int force_rq = []() { const char* e = getenv("MTV_BLOCK_RQ"); return e ? atoi(e) : 0; }();
            const int max_wgs = std::min(128, c->resident_cus / 2);        // all workgroups of the launch resident together, on half of what the launch may use
            if (pays && (deep_block_configure(ba, force_cl, force_rq, max_wgs) || ((force_cl || force_rq) && deep_block_configure(ba, 0, 0, max_wgs)))) {
                Tens out;
                out.lvl = lvl; out.C = C; out.ks = H;
                out.slab = (unsigned)((size_t)deep_clips() * L.L * C);
                out.p = c->buf("act.deep." + nm + ".out", (size_t)8 * out.slab);
                // Scratch and counters belong to THIS op (the same op of the context's other plans -- forward / step parities -- shares them:
                // launches are serial).  Never shared between ops: a granule is valid when its tag equals the reader's epoch, and two ops
                // count their epochs separately -- in a shared buffer op B would accept what op A wrote at the same count.
                const std::string okey = nm + ".B" + std::to_string(B) + ".CL" + std::to_string(ba.CL) + ".K" + std::to_string(ba.KSN);
                const size_t pf = deep_block_part_floats(ba), qf = deep_block_qkv_floats(ba);
                ba.part = c->buf("deep.block.part." + okey, pf);
                ba.qkv = c->buf("deep.block.qkv." + okey, qf);
                if (ba.RQ > 1) ba.stg = c->buf("deep.block.stg." + okey + ".RQ" + std::to_string(ba.RQ), deep_block_stg_floats(ba));
                // entry tickets: 64-bit, monotonic (never reset), per (clip, head)
                ba.cnt = reinterpret_cast<unsigned long long*>(c->buf("deep.block.cnt." + okey, (size_t)B * H * 2 * 2));
                ba.fault = c->fault_d;
                ba.out = out.p;
                ba.out_slab_stride = out.slab;
                if (!out.p || !ba.part || !ba.qkv || !ba.cnt || !ba.fault || (ba.RQ > 1 && !ba.stg)) { err = "deep block allocation failed at " + nm; return Tens{}; }
                c->taps[nm + ".out"] = {lvl, C};
                c->bufs["tap." + nm + ".out"] = out.p;
                c->tap_slabs[nm + ".out"] = {out.ks, out.slab};
                double aflops = 0.0;
                {
                    const int segs[3] = {L.b1, L.b2 - L.b1, L.L - L.b2};
                    if (whole) aflops = 4.0 * B * H * (double)L.L * L.L * d;
                    else for (int sg : segs) aflops += 4.0 * B * H * (double)sg * sg * d;
                }
                if (c->accounting) {
                    c->work.flops_attn_core += aflops / B;
                    c->work.flops_1x1 += 2.0 * (double)L.L * C * 3.0 * C + 2.0 * (double)L.L * C * C;
                    c->work.bytes_weights_other += 4.0 * (4.0 * (double)C * C + 4.0 * C);
                }
                char tag[96];
                snprintf(tag, sizeof tag, "[L%d d%d %s blk cl%d,cs%d,rq%d]", L.L, d, whole ? "1d" : "2d", ba.CL, ba.CS, ba.RQ);
                auto bop = std::make_shared<DeepBlockArgs>(ba);
                push("attn:" + nm + tag, [bop](hipStream_t s) { return launch_deep_block(*bop, s); },
                     aflops + 2.0 * B * (double)L.L * C * 4.0 * C, 4.0 * B * L.L * (2.0 * C) + 4.0 * (4.0 * (double)C * C + 4.0 * C));
                return out;
            }
        }
        const bool fused = deep_on(lvl) && fuse_env && deep_attn_configure(da);
        auto emit_fused = [&](const float* qkv_plain, const Tens& xres) -> Tens {
            Tens out;
            out.lvl = lvl; out.C = C; out.ks = da.nhg;
            out.slab = (unsigned)((size_t)deep_clips() * L.L * C);
            out.p = c->buf("act.deep." + nm + ".out", (size_t)8 * out.slab);
            if (!out.p) { err = "deep attention allocation failed at " + nm; return Tens{}; }
            c->taps[nm + ".out"] = {lvl, C};
            c->bufs["tap." + nm + ".out"] = out.p;
            c->tap_slabs[nm + ".out"] = {out.ks, out.slab};
            da.qkv = qkv_plain;
            da.res = dsrc(xres);
            da.out = out.p;
            da.out_slab_stride = out.slab;
            double aflops = 0.0;
            {
                const int segs[3] = {L.b1, L.b2 - L.b1, L.L - L.b2};
                if (whole) aflops = 4.0 * B * H * (double)L.L * L.L * d;
                else for (int sg : segs) aflops += 4.0 * B * H * (double)sg * sg * d;
            }
            if (c->accounting) {
                c->work.flops_attn_core += aflops / B;
                c->work.flops_1x1 += 2.0 * (double)L.L * C * C;
                c->work.bytes_weights_other += 4.0 * (double)C * C + 4.0 * C;
            }
            char tag[96];
            snprintf(tag, sizeof tag, "[L%d d%d %s +proj h%d,c%d]", L.L, d, whole ? "1d" : "2d", da.HPW, da.NC);
            auto aop = std::make_shared<DeepAttnOp>();
            aop->a = da;
            push("attn:" + nm + tag, [aop](hipStream_t s) { return launch_deep_attn(aop->a, s); }, aflops + 2.0 * B * (double)L.L * C * C,
                 4.0 * B * L.L * (3.0 * C + 2.0 * C) + 4.0 * (double)C * C);
            if (out.ks > 1) attn_producer[out.p] = aop;
            return out;
        };
        if (deep_on(lvl) && deep_qkv_env) {
            DeepArgs dq = deep_args(lvl, 1, 3 * C, bq);
            dq.Cmain = C;
            dq.main[0] = dsrc(x0);
            dq.gn = 1; dq.whole = whole ? 1 : 0; dq.act = 0; dq.gs = C / 32; dq.gamma = gw; dq.beta = gb;
            DeepArgs dp = deep_args(lvl, 1, C, bp);
            dp.Cmain = C;
            dp.main[0] = DeepSrc{dp.zeros, 0, 1, C};                       // (placeholder for the attention output)
            dp.res = dsrc(x0);
            DeepTile tq{}, tp{};
            if (deep_configure(dq, &tq) && (fused || deep_configure(dp, &tp))) {
                const Tens qkvd = emit_deep(dq, tq, Wq, ldq, nm + ".qkv", lvl);
                if (!err.empty()) return Tens{};
                const Tens qkvp = materialize(qkvd, nm + ".qkv");
                if (fused) return emit_fused(qkvp.p, x0);
                float* attd = c->act(nm + ".att", lvl, C);
                AttnArgs t = attn_args(qkvp.p, attd, L, C, H, d, whole);
                push_attention(t, nm, L, C, d, whole);
                Tens ad;
                ad.lvl = lvl; ad.C = C; ad.p = attd;
                dp.main[0] = dsrc(ad);
                return emit_deep(dp, tp, Wp, ldp, nm + ".out", lvl);
            }
        }
        const Tens x = materialize(x0, nm + ".in");       // k_conv reads a plain tensor and the statistics its producer left
        if (fused) {
            // qkv from the plain input with the whole K per workgroup (no slices: the output is one plain tensor, which is what
            // the attention needs), GroupNorm statistics computed in the kernel -- where all channels of a row group fit in LDS
            // (32 tokens; 128 tokens per plane group, not with whole-L statistics)
            // (off by default: measured 9.5 us against k_conv's 7.9 at 32 tokens -- 96 workgroups, each staging the whole input)
            const bool ks1_env = deep_opt(MTV_DEEP_OPT_UNSLICED_QKV, "MTV_DEEP_QKV1", false);
            DeepArgs dq = deep_args(lvl, 1, 3 * C, bq);
            dq.Cmain = C;
            dq.main[0] = dsrc(x);
            dq.gn = 1; dq.whole = whole ? 1 : 0; dq.act = 0; dq.gs = C / 32; dq.gamma = gw; dq.beta = gb;
            DeepTile tq{};
            if (ks1_env && deep_configure(dq, &tq, 1)) {
                const Tens qkvd = emit_deep(dq, tq, Wq, ldq, nm + ".qkv", lvl);
                if (!err.empty()) return Tens{};
                return emit_fused(qkvd.p, x);
            }
        }
        double* site = c->new_site();
        add_stats({x}, lvl, site);
        float* qkv = c->act(nm + ".qkv", lvl, 3 * C);
        ConvArgs a{};
        a.ntaps = 1; a.Lout = L.L; a.Lsrc = L.L; a.N = 3 * C; a.W = Wq; a.Wnk = Wq_nk; a.Wpk = Wq_pk; a.ldw = ldq; a.bias = bq; a.out = qkv;
        a.nmain = 1; a.src[0] = x.p; a.C[0] = C; a.Cmain = C; a.seg_src = L.seg();
        a.gn = GnIn{site, gw, gb, nullptr, 0, C / 32, whole ? 1 : 0, 0, (unsigned)c->stats_copy_doubles};
        add_conv(a, nm + ".qkv", lvl);
        if (fused) return emit_fused(qkv, x);

        float* att = c->act(nm + ".att", lvl, C);
        push_attention(attn_args(qkv, att, L, C, H, d, whole), nm, L, C, d, whole);

        Tens out;
        out.lvl = lvl; out.C = C; out.p = c->act(nm + ".out", lvl, C);
        ConvArgs p{};
        p.ntaps = 1; p.Lout = L.L; p.Lsrc = L.L; p.Lskip = L.L; p.N = C; p.W = Wp; p.Wnk = Wp_nk; p.Wpk = Wp_pk; p.ldw = ldp; p.bias = bp; p.out = out.p;
        p.nmain = 1; p.src[0] = att; p.C[0] = C; p.Cmain = C; p.seg_src = L.seg();
        p.res = x.p;
        add_conv(p, nm + ".proj", lvl);
        return out;
    }

    Tens run_stage(const Stage& st, std::vector<Tens> x, const std::string& nm) {
        Tens cur;
        for (size_t j = 0; j < st.layers.size(); ++j) {
            const Layer& ly = st.layers[j];
            const std::string lnm = nm + "." + std::to_string(j);
            if (ly.type == 0) {
                // stem conv (unet.py:714): 16 -> model_channels, no norm
                const int ld = pad64(ly.c);
                float* W = c->buf("w." + ly.pre + "weight", (size_t)9 * 16 * ld);
                wconv(ly.pre + "weight", ly.c, 16, 3, 3, false, W, ld);
                float* bias = c->wcopy(ly.pre + "bias", {ly.c});
                const Level& L = c->lv[0];
                cur.lvl = 0; cur.C = ly.c; cur.p = c->act(lnm + ".out", 0, ly.c);
                ConvArgs a{};
                a.ntaps = 9; a.Lout = L.L; a.Lsrc = L.L; a.N = ly.c; a.W = W; a.ldw = ld; a.bias = bias; a.out = cur.p;
                a.nmain = 1; a.src[0] = x[0].p; a.C[0] = 16; a.Cmain = 16; a.gather = c->g3[0]; a.seg_src = L.seg();
                add_conv(a, lnm, 0);
            } else if (ly.type == 1) {
                cur = resblock(x, ly, lnm);
            } else {
                cur = attention(x[0], ly.pre, false, lnm);
            }
            if (!err.empty()) return Tens{};
            x = {cur};
        }
        if (st.attn1_c) cur = attention(cur, st.attn1_pre, true, nm + ".a1");
        c->taps[st.tap] = {cur.lvl, cur.C};
        c->bufs["tap." + st.tap] = cur.p;
        if (cur.ks > 1) c->tap_slabs[st.tap] = {cur.ks, cur.slab};
        else c->tap_slabs.erase(st.tap);
        return cur;
    }

    int build() {
        const int mc = f.model_channels;
        // ---- timestep embedding path (unet.py:1011-1012 and every ResBlock's emb_layers) ----
        c->buf("w.film", (size_t)c->film_total * emb);
        c->buf("w.film_bias", (size_t)c->film_total);
        float* tsin = c->buf("emb.sin", (size_t)f.max_batch * mc);
        float* e0 = c->buf("emb.e0", (size_t)f.max_batch * emb);
        float* e1 = c->buf("emb.e1", (size_t)f.max_batch * emb);
        float* W0 = c->wcopy("time_embed.0.weight", {emb, mc});
        float* B0 = c->wcopy("time_embed.0.bias", {emb});
        float* W2 = c->wcopy("time_embed.2.weight", {emb, emb});
        float* B2 = c->wcopy("time_embed.2.bias", {emb});
        c->site_parity = mode == MODE_STEP1 ? 1 : 0;
        if (mode != MODE_FORWARD) {
            // sampler step: the FiLM row of this step was copied into emb.film_step by the previous step's head
            // conv (or by k_ddim_init), the statistics arena was zeroed by it, the packed input written by it
            film_out = c->buf("emb.film_step", (size_t)c->film_total);
            film_stride = 0;
        } else {
            film_out = c->buf("emb.film", (size_t)f.max_batch * c->film_total);
            film_stride = c->film_total;
            double* st = c->stats;
            const size_t nb = c->stats_bytes;
            push("memset_stats", [st, nb](hipStream_t s) { return hipMemsetAsync(st, 0, nb, s); });
            const int64_t* tb = c->tbuf;
            const float* fr = c->freqs;
            const int Bn = B, half = mc / 2;
            push("time_sinusoid", [tb, fr, tsin, Bn, half](hipStream_t s) { return launch_time_sinusoid(tb, fr, tsin, Bn, half, s); });
            LinearArgs l0{tsin, W0, B0, e0, B, mc, emb, emb, 0};
            push("time_embed.0", [l0](hipStream_t s) { return launch_linear(l0, s); });
            LinearArgs l2{e0, W2, B2, e1, B, emb, emb, emb, 1};
            push("time_embed.2", [l2](hipStream_t s) { return launch_linear(l2, s); });
            LinearArgs lf{e1, c->bufs["w.film"], c->bufs["w.film_bias"], film_out, B, emb, c->film_total, c->film_total, 1};
            push("film", [lf](hipStream_t s) { return launch_linear(lf, s); });
            if (c->accounting) {
                c->work.flops_linear += 2.0 * ((double)mc * emb + (double)emb * emb + (double)emb * c->film_total);
                c->work.bytes_weights_other += 4.0 * ((double)mc * emb + (double)emb * emb + (double)emb * c->film_total);
            }
        }
        // ---- input assembly (unet.py:1022-1025) ----
        Tens h0;
        h0.lvl = 0; h0.C = 16; h0.p = c->act("h0", 0, 16);
        if (mode == MODE_FORWARD) {
            const float *xi = c->xin, *ci = c->condin, *ii = c->icin;
            float* o = h0.p;
            const int Bn = B, L = c->lv[0].L, RR = c->lv[0].b1;
            push("pack_input", [xi, ci, ii, o, Bn, L, RR](hipStream_t s) { return launch_pack_input(xi, ci, ii, RR, o, Bn, L, RR, s); });
        }
        c->site_cursor = 0;
        std::vector<Tens> skips;
        Tens cur = h0;
        for (size_t i = 0; i < c->inputs.size(); ++i) {
            cur = run_stage(c->inputs[i], {cur}, "in" + std::to_string(i));
            if (!err.empty()) return fail(MTV_ERR_INVALID, err);
            skips.push_back(cur);
        }
        cur = run_stage(c->middle, {cur}, "mid");
        if (!err.empty()) return fail(MTV_ERR_INVALID, err);
        for (size_t i = 0; i < c->outputs.size(); ++i) {
            Tens sk = skips.back();
            skips.pop_back();
            if (sk.lvl != cur.lvl) return fail(MTV_ERR_INVALID, "skip level mismatch");
            cur = run_stage(c->outputs[i], {cur, sk}, "out" + std::to_string(i));
            if (!err.empty()) return fail(MTV_ERR_INVALID, err);
        }
        // ---- head (unet.py:971-975, 1103-1112): GN + SiLU + conv3x3 -> eps in the external layout ----
        {
            float* gw = gnvec("out.0.weight", cur.C);
            float* gb = gnvec("out.0.bias", cur.C);
            const int ld = pad64(f.out_channels);
            float* W = c->buf("w.out.2.weight", (size_t)9 * cur.C * ld);
            wconv("out.2.weight", f.out_channels, cur.C, 3, 3, false, W, ld);
            float* bias = c->wcopy("out.2.bias", {f.out_channels});
            // (a geometry whose level 0 is a deep level -- <= 128 tokens at B <= 2 -- ends the last stage on a slab tensor: k_conv
            //  reads ONE plain tensor and the statistics its producer left)
            cur = materialize(cur, "head.in");
            if (!err.empty()) return fail(MTV_ERR_INVALID, err);
            double* site = c->new_site();
            add_stats({cur}, 0, site);
            const Level& L = c->lv[0];
            ConvArgs a{};
            a.ntaps = 9; a.Lout = L.L; a.Lsrc = L.L; a.N = f.out_channels; a.W = W; a.ldw = ld; a.bias = bias;
            a.out = c->eps; a.out_cm = 1;
            a.nmain = 1; a.src[0] = cur.p; a.C[0] = cur.C; a.Cmain = cur.C; a.gather = c->g3[0]; a.seg_src = L.seg();
            a.gn = GnIn{site, gw, gb, nullptr, 0, cur.C / 32, 0, 1, (unsigned)c->stats_copy_doubles};
            if (mode != MODE_FORWARD) {             // DDIM update + step hand-over
                a.ddim = c->d_fuse + (mode == MODE_STEP1 ? 1 : 0);
                a.step_counter = c->d_counter;
            }
            add_conv(a, "head", 0);
        }
        if (c->site_cursor > c->n_sites) return fail(MTV_ERR_INVALID, "GN site arena overflow");
        { const int rcs = finish_split_k(c, plan); if (rcs != MTV_OK) return rcs; }
        if (c->accounting) c->work.n_launches = (int)plan->ops.size();
        return MTV_OK;
    }
};

}  // namespace

// Empirical tile selection: every distinct conv shape of the plan is timed once over the valid
// (MT, NT, NW, KS) candidates with its real arguments (MTV_AUTOTUNE=0 keeps the analytic pick).
// testing aid: MTV_FORCE_LDS="WM,WN" (or mtv_debug_force_lds) runs every eligible conv of plans built afterwards on the
// LDS-tiled kernel k_conv_lds<WM, WN>; a conv that turns out not to be eligible at launch falls back (launch_conv)
static int g_force_wm = -1, g_force_wn = 0;
static int g_force_b3[3] = {-1, 0, 1};       // MTV_FORCE_B3="MT,NT[,KS]" (or mtv_debug_force_b3): every eligible conv on the split-bf16 kernel k_conv_x3<MT, NT>
static int g_force_pw[3] = {-1, 0, 1};         // MTV_FORCE_PW="MT,NTW" (or mtv_debug_force_pw): every eligible 1x1 conv on k_conv_pw<MT, NTW>
static int g_force_win[3] = {-1, 0, 1};     // MTV_FORCE_WIN="MT,NT[,KS]" (or mtv_debug_force_win / _ks): every eligible 3x3 conv on k_conv_win<MT, NT>
static int g_force_lin[3] = {-1, 0, 0};     // MTV_FORCE_LIN="MT,NT,NWV" (or mtv_debug_force_lin): every eligible 1x1 conv on k_lin<MT, NT, NWV>
static void parse_force_b3() {
    if (g_force_b3[0] != -1) return;
    g_force_b3[0] = 0;
    if (const char* e = getenv("MTV_FORCE_B3")) {
        int x = 0, y = 0, z = 1;
        const int n = sscanf(e, "%d,%d,%d", &x, &y, &z);
        if (n >= 2 && x3_tile_exists(x, y) && (z == 1 || z == 2 || z == 4 || z == 8)) { g_force_b3[0] = x; g_force_b3[1] = y; g_force_b3[2] = z; }
    }
}
// Convs of at least X3_MIN_ROWS tokens (all clips together) get a split-bf16 weight copy + activation scratch and are offered to
// the tuner on k_conv_x3; below that the elementwise pass and the few 128-row tiles cannot pay (profiles/r03_conv_x3_bench.txt).
// A forced tile (tests) lifts the bound.
bool x3_wanted(long rows) {
    parse_force_b3();
    return rows >= X3_MIN_ROWS || g_force_b3[0] > 0;
}
static bool deep_forced_off() {
    parse_force_b3();
    return g_force_wm > 0 || g_force_lin[0] > 0 || g_force_b3[0] > 0 || getenv("MTV_FORCE_TILE") || getenv("MTV_FORCE_LDS") || getenv("MTV_FORCE_LIN");     // (not MTV_FORCE_WIN:
                                                                                                                                                           //  k_conv_win serves the large levels)
}
void force_lds_tile(const ConvArgs& a, ConvTile* t) {
    if (g_force_lin[0] == -1) {
        g_force_lin[0] = 0;
        if (const char* e = getenv("MTV_FORCE_LIN")) {
            int x = 0, y = 0, z = 0;
            if (sscanf(e, "%d,%d,%d", &x, &y, &z) == 3 && (x == 1 || x == 2) && (y == 1 || y == 2 || y == 4) && (z == 1 || z == 2 || z == 4)) { g_force_lin[0] = x; g_force_lin[1] = y; g_force_lin[2] = z; }
        }
    }
    if (g_force_lin[0] > 0 && conv_lin_eligible(a)) { *t = ConvTile{g_force_lin[0], g_force_lin[1], 64, g_force_lin[2], 0}; return; }
    if (g_force_win[0] == -1) {
        g_force_win[0] = 0;
        if (const char* e = getenv("MTV_FORCE_WIN")) {
            int x = 0, y = 0, z = 1;
            if (sscanf(e, "%d,%d,%d", &x, &y, &z) >= 2 && (x == 1 || x == 2) && (y == 2 || y == 4) && (z == 1 || z == 2 || z == 4)) { g_force_win[0] = x; g_force_win[1] = y; g_force_win[2] = z; }
        }
    }
    if (g_force_win[0] > 0) {
        static const int xm = getenv("MTV_FORCE_WIN_XM") ? atoi(getenv("MTV_FORCE_WIN_XM")) != 0 : 0;      // (with the XCD-aware block order)
        // (K slices where the conv can take them -- no fused skip conv, whole chunks per slice -- else the unsliced tile)
        const int ks = conv_win_eligible(a, g_force_win[0], g_force_win[1], g_force_win[2]) ? g_force_win[2] : 1;
        if (conv_win_eligible(a, g_force_win[0], g_force_win[1], ks)) {
            *t = ConvTile{g_force_win[0], g_force_win[1], 80, ks, xm};
            return;
        }
    }
    if (g_force_pw[0] == -1) {
        g_force_pw[0] = 0;
        if (const char* e = getenv("MTV_FORCE_PW")) {
            int x = 0, y = 0;
            int z = 1;
            const int nf = sscanf(e, "%d,%d,%d", &x, &y, &z);
            if (nf >= 2 && (x == 1 || x == 2) && (y == 1 || y == 2) && (z == 1 || z == 2 || z == 4 || z == 6)) { g_force_pw[0] = x; g_force_pw[1] = y; g_force_pw[2] = z; }
        }
    }
    if (g_force_pw[0] > 0 && conv_pw_eligible(a, g_force_pw[0], g_force_pw[1], g_force_pw[2])) { *t = ConvTile{g_force_pw[0], g_force_pw[1], 96, g_force_pw[2], 0}; return; }
    parse_force_b3();
    if (g_force_b3[0] > 0 && conv_x3_eligible(a) && conv_x3_smem_bytes(a, ConvTile{g_force_b3[0], g_force_b3[1], 48, 1, 0}) <= CONV_X3_MAX_LDS) {
        const int nch32 = a.ntaps * (a.Cmain / 32) + a.Cskip / 32;
        *t = ConvTile{g_force_b3[0], g_force_b3[1], 48, nch32 >= 6 * g_force_b3[2] ? g_force_b3[2] : 1, 0};     // (K slices of at least 6 chunks)
        return;
    }
    if (g_force_wm == -1) {
        g_force_wm = 0;
        if (const char* e = getenv("MTV_FORCE_LDS")) {
            int x = 0, y = 0;
            if (sscanf(e, "%d,%d", &x, &y) == 2 && (x == 2 || x == 4) && (y == 2 || y == 4 || y == 8)) { g_force_wm = x; g_force_wn = y; }
        }
    }
    if (g_force_wm > 0 && conv_lds_eligible(a)) *t = ConvTile{g_force_wm, g_force_wn, 32, 1, 0};
}

// One slab shared by every cross-workgroup split-K conv of a plan (they run back to back), sized so the auto-tuner may
// try up to 16 K slices wherever that stays under 64 MB; plus the arrival counters of the in-launch split-K completion:
// one per 16x16 output tile (the finest tiling), zeroed once here -- every launch leaves them at zero again.
int finish_split_k(mtv_ctx* c, Plan* plan) {
    const int B = plan->B;
    // UNet plans of one batch size (forward, step 0 / 1) have identical convs and share slab + counters; the two
    // autoencoder plans differ, so each gets its own (keyed by mode): mtv_ctx::buf() refuses a second, larger request
    const std::string tag = std::to_string(B) + (plan->mode >= MODE_AE_DECODE ? ".ae.m" + std::to_string(plan->mode) : "");
    size_t need = 0;
    for (auto& op : plan->convs) {
        const size_t one = (size_t)B * op->a.Lout * op->a.N;
        size_t ks = 16;
        while (ks > 1 && ks * one * 4 > ((size_t)64 << 20)) ks /= 2;
        if (op->t.NW != 64 && op->t.NW != 96 && (size_t)op->t.KS > ks) ks = op->t.KS;      // (k_lin's / k_conv_pw's KS is a wave count)
        if (ks < 2) continue;                       // never split: needs no slab (the autoencoder's 16384-token GEMMs)
        need = ks * one > need ? ks * one : need;
    }
    float* slab = c->buf("slab.B" + tag, need);
    if (!slab) return fail(MTV_ERR_HIP, "slab allocation failed: " + std::string(mtv_last_error()));
    // what the tuner may use = what is really allocated for THIS plan (autotune validates K slices against it)
    plan->slab_floats = c->buf_floats["slab.B" + tag];
    for (auto& op : plan->convs) op->a.slab = slab;
    // ... and one scratch for the split activations of the convs that may run on the split-bf16 kernels (those with a W3 copy)
    size_t x3need = 0;
    for (auto& op : plan->convs)
        if (op->a.W3) x3need = std::max(x3need, conv_x3_scratch_bytes(B, op->a.Lsrc, op->a.Cmain, op->a.Lskip, op->a.Cskip));
    if (x3need) {
        float* x3 = c->buf("x3.B" + tag, (x3need + 3) / 4);
        if (!x3) return fail(MTV_ERR_HIP, "split-activation scratch allocation failed: " + std::string(mtv_last_error()));
        for (auto& op : plan->convs)
            if (op->a.W3) op->a.x3 = x3;
    }
    size_t nt = 0;
    for (auto& op : plan->convs) nt += (size_t)B * ((op->a.Lout + 15) / 16) * ((op->a.N + 15) / 16);
    int* tk = (int*)c->buf("tickets.B" + tag, nt);
    if (!tk) return fail(MTV_ERR_HIP, "ticket allocation failed: " + std::string(mtv_last_error()));
    for (auto& op : plan->convs) {
        op->a.tickets = tk;
        tk += (size_t)B * ((op->a.Lout + 15) / 16) * ((op->a.N + 15) / 16);
    }
    return MTV_OK;
}

// Tile table.  moditalker_amd/csrc/tune_gfx950.txt (committed, next to the library) holds the measured best tile
// of every conv shape of the BASELINE configurations, so a fresh process reproduces the same launch plan without
// timing anything; shapes it does not list are tuned on first use.  MTV_TUNE_CACHE=<file> replaces it (and is
// appended to: that is how the committed table is regenerated, tools/make_tune_table.sh).
static std::string default_tune_path() {
    Dl_info di;
    if (dladdr((const void*)&default_tune_path, &di) && di.dli_fname) {
        std::string p(di.dli_fname);
        const size_t k = p.rfind('/');
        return (k == std::string::npos ? std::string(".") : p.substr(0, k)) + "/tune_gfx950.txt";
    }
    return "tune_gfx950.txt";
}

static void tune_cache_load(mtv_ctx* c) {
    if (c->tune_cache_loaded) return;
    c->tune_cache_loaded = true;
    const char* env = getenv("MTV_TUNE_CACHE");
    const std::string path = env ? std::string(env) : default_tune_path();
    if (FILE* f = fopen(path.c_str(), "r")) {
        char line[256];
        while (fgets(line, sizeof line, f)) {
            if (line[0] == '#') continue;
            char* bar = strchr(line, '|');
            if (!bar) continue;
            *bar = 0;
            ConvTile t{};
            if (sscanf(bar + 1, "%d %d %d %d %d", &t.MT, &t.NT, &t.NW, &t.KS, &t.XM) == 5) c->tune_cache[line] = t;
        }
        fclose(f);
    }
}

static void tune_cache_append(const char* key, const ConvTile& t) {
    const char* path = getenv("MTV_TUNE_CACHE");
    if (!path) return;
    if (FILE* f = fopen(path, "a")) {
        fprintf(f, "%s|%d %d %d %d %d\n", key, t.MT, t.NT, t.NW, t.KS, t.XM);
        fclose(f);
    }
}

int autotune(mtv_ctx* c, Plan* p, hipStream_t s) {
    if (p->tuned) return MTV_OK;
    p->tuned = true;
    tune_cache_load(c);
    const char* env = getenv("MTV_AUTOTUNE");
    if ((env && atoi(env) == 0) || getenv("MTV_FORCE_TILE") || g_force_wm > 0 || g_force_lin[0] > 0 || g_force_b3[0] > 0 || g_force_win[0] > 0 || g_force_pw[0] > 0) return MTV_OK;
    static const int cand[][2] = {{4, 4}, {2, 4}, {1, 4}, {2, 2}, {1, 2}, {1, 1}};
    struct Events {             // destroyed on every exit path (HIPCHK returns early)
        hipEvent_t e0 = nullptr, e1 = nullptr;
        ~Events() {
            if (e0) (void)hipEventDestroy(e0);
            if (e1) (void)hipEventDestroy(e1);
        }
    } ev;
    HIPCHK(hipEventCreate(&ev.e0));
    HIPCHK(hipEventCreate(&ev.e1));
    hipEvent_t e0 = ev.e0, e1 = ev.e1;
    static const int nsamp = []() { const char* e = getenv("MTV_TUNE_SAMPLES"); const int v = e ? atoi(e) : 5; return v < 1 ? 1 : (v > 15 ? 15 : v); }();
    const size_t slab_cap = p->slab_floats;
    if (!c->flush) {
        c->flush_bytes = (size_t)320 << 20;
        int rcm = c->dmalloc((void**)&c->flush, c->flush_bytes);
        if (rcm != MTV_OK) return rcm;
    }
    for (auto& op : p->convs) {
        ConvArgs a = op->a;
        a.ddim = nullptr;       // the tuner times the conv itself, not the step hand-over
        auto run = [&](const ConvTile& tt) -> hipError_t { return op->tune_launch ? op->tune_launch(a, tt, s) : launch_conv(a, tt, s); };
        char key[176];
        snprintf(key, sizeof key, "B%d L%d/%d/%d N%d t%d C%d+%d gn%d f%d r%d s%d cm%d", a.B, a.Lout, a.Lsrc, a.Lskip, a.N, a.ntaps, a.Cmain, a.Cskip,
                 a.gn.sums ? 1 : 0, a.gn.film ? 1 : 0, a.res ? 1 : 0, a.nstat, a.out_cm);
        // Two 1x1 convs can share a shape key and differ in whether the lean kernel (k_lin: identity rows, [N][K] weight copy) can run
        // them.  Whatever order they are met in, each has its own entry: "<key> l" for the one k_lin can run (falling back to a plain
        // "<key>" of an older table), the plain key for the other -- unless that holds a k_lin tile, then "<key> x".
        const bool lin_ok_conv = conv_lin_eligible(a);
        bool lin_fallback = false;      // the entry found is the plain "<key>" of an older table, possibly the non-k_lin twin's
        auto it = c->tune_cache.end();
        if (lin_ok_conv) {
            char kl[192];
            snprintf(kl, sizeof kl, "%s l", key);
            it = c->tune_cache.find(kl);
            if (it == c->tune_cache.end()) { it = c->tune_cache.find(key); lin_fallback = it != c->tune_cache.end(); }
            if (it == c->tune_cache.end()) snprintf(key, sizeof key, "%s", kl);     // a new measurement goes under the "l" key
        } else {
            it = c->tune_cache.find(key);
            if (it != c->tune_cache.end() && it->second.NW == 64) {
                strncat(key, " x", sizeof key - strlen(key) - 1);
                it = c->tune_cache.find(key);
            }
        }
        if (it != c->tune_cache.end()) {             // an entry read from MTV_TUNE_CACHE is only trusted if it is launchable
            const ConvTile& t = it->second;
            const int nchunks = a.ntaps * (a.Cmain / 16) + a.Cskip / 16;
            const bool tiled_ok = t.NW == 32 && (t.MT == 2 || t.MT == 4) && (t.NT == 2 || t.NT == 4 || t.NT == 8) && t.KS == 1 && t.XM == 0 && conv_lds_eligible(a);
            const bool lin_ok = t.NW == 64 && (t.MT == 1 || t.MT == 2) && (t.NT == 1 || t.NT == 2 || t.NT == 4) && (t.KS == 1 || t.KS == 2 || t.KS == 4) && t.XM == 0 && conv_lin_eligible(a);
            const bool b3_ok = t.NW == 48 && !(a.B == 1 && a.Lout <= 2048) && x3_tile_exists(t.MT, t.NT) && (t.KS == 1 || t.KS == 2 || t.KS == 4 || t.KS == 8) && t.XM == 0 && t.KS * 6 <= a.ntaps * (a.Cmain / 32) + a.Cskip / 32 && conv_x3_eligible(a) && a.x3 && conv_x3_smem_bytes(a, t) <= CONV_X3_MAX_LDS;
            const bool win_ok = t.NW == 80 && (t.MT == 1 || t.MT == 2) && (t.NT == 2 || t.NT == 4) && (t.KS == 1 || t.KS == 2 || t.KS == 4) && (t.XM == 0 || t.XM == 1) && conv_win_eligible(a, t.MT, t.NT, t.KS);
            const bool pw_ok = t.NW == 96 && (t.KS == 1 || t.KS == 2 || t.KS == 4 || t.KS == 6) && (t.XM == 0 || t.XM == 1) && conv_pw_eligible(a, t.MT, t.NT, t.KS);
            const bool shape_ok = tiled_ok || lin_ok || b3_ok || win_ok || pw_ok ||
                                  ((t.MT == 1 || t.MT == 2 || t.MT == 4) && (t.NT == 1 || t.NT == 2 || t.NT == 4) &&
                                   (t.NW == 1 || t.NW == 2 || t.NW == 4 || t.NW == 8 || t.NW == 16) && !(t.NW == 16 && t.MT * t.NT >= 8) &&
                                   t.KS >= 1 && t.KS <= 16 && (t.KS & (t.KS - 1)) == 0 && (t.XM == 0 || t.XM == 1));
            if (!shape_ok || (!tiled_ok && !lin_ok && !b3_ok && !win_ok && !pw_ok && t.NW * t.KS > nchunks) || (!lin_ok && !pw_ok && t.KS > 1 && ((size_t)t.KS * a.B * a.Lout * a.N > slab_cap || (a.N & 3))) ||
                (!b3_ok && !win_ok && !pw_ok && conv_smem_bytes(a, t) > 120 * 1024)) {
                // (a plain entry that a lin-eligible conv only borrowed stays: it may be its twin's; the new measurement goes under "<key> l")
                if (lin_fallback) { char kl[192]; snprintf(kl, sizeof kl, "%s l", key); snprintf(key, sizeof key, "%s", kl); }
                else c->tune_cache.erase(it);
                it = c->tune_cache.end();
            }
        }
        if (it == c->tune_cache.end()) {
            const int nchunks = a.ntaps * (a.Cmain / 16) + a.Cskip / 16;
            const double wbytes = 4.0 * ((double)a.ntaps * a.Cmain + a.Cskip) * a.N;
            const double abytes = 4.0 * a.B * ((double)a.Lsrc * a.Cmain + (double)a.Lskip * a.Cskip + (double)a.Lout * a.N);
            ConvTile best = op->t;
            float best_ms = 1e30f;
            for (auto& mn : cand) {
                const int MT = mn[0], NT = mn[1];
                if (NT > 1 && NT * 8 >= a.N) continue;
                if (MT > 1 && 16 * (MT / 2) >= a.Lout) continue;
                const long tiles = (long)a.B * ((a.Lout + 16 * MT - 1) / (16 * MT)) * ((a.N + 16 * NT - 1) / (16 * NT));
                for (int NW = 1; NW <= 16; NW *= 2) {
                    if (NW == 16 && MT * NT >= 8) continue;
                    for (int KS = 1; KS <= 16; KS *= 2) {
                        if (NW * KS > nchunks) continue;
                        if (KS > 1 && ((size_t)KS * a.B * a.Lout * a.N > slab_cap || (a.N & 3))) continue;
                        const long waves = tiles * NW * KS;
                        if (waves > 16384 || (waves < 512 && tiles * KS < 64 && NW * KS * 2 <= nchunks && NW < 16)) continue;
                        for (int XM = 0; XM < 2; ++XM) {
                            if (XM == 1 && (wbytes < 2.0 * abytes || (long)((a.N + 16 * NT - 1) / (16 * NT)) * KS < 8)) continue;
                            const ConvTile t{MT, NT, NW, KS, XM};
                            if (conv_smem_bytes(a, t) > 120 * 1024) continue;
                            // cold timing: the caches (32 MB L2 + 256 MB MALL) are flushed before every timed
                            // launch, because in the real step a layer's weights were last touched 0.5 GB ago
                            // (median of `nsamp` samples: a single slow sample -- clock ramp, a neighbour's
                            // traffic -- must not decide the plan)
                            float samp[16];
                            HIPCHK(run(t));
                            for (int w = 0; w < nsamp; ++w) {
                                HIPCHK(hipMemsetAsync(c->flush, w, c->flush_bytes, s));
                                HIPCHK(hipEventRecord(e0, s));
                                HIPCHK(run(t));
                                HIPCHK(hipEventRecord(e1, s));
                                HIPCHK(hipEventSynchronize(e1));
                                HIPCHK(hipEventElapsedTime(&samp[w], e0, e1));
                            }
                            std::sort(samp, samp + nsamp);
                            const float med = samp[nsamp / 2];
                            if (med < best_ms) {
                                best_ms = med;
                                best = t;
                            }
                        }
                    }
                }
            }
            // the LDS-tiled kernel for large token counts (>= 2048 rows in all: fewer would leave most CUs idle)
            if ((long)a.B * a.Lout >= 2048 && conv_lds_eligible(a)) {
                // (x 8: 256-column tiles.  Chosen for the autoencoder's widest GEMMs (N = 1536 / 3072 at 16384 tokens); never
                // for a UNet conv, although the GroupNorm / FiLM / SiLU transform is repeated per column tile)
                static const int tl[][2] = {{4, 4}, {2, 4}, {4, 2}, {2, 2}, {2, 8}, {4, 8}};
                for (auto& mn : tl) {
                    const ConvTile t{mn[0], mn[1], 32, 1, 0};
                    if (mn[1] == 8 && a.N < 256) continue;
                    if ((long)a.B * ((a.Lout + 32 * t.MT - 1) / (32 * t.MT)) * ((a.N + 32 * t.NT - 1) / (32 * t.NT)) < 128) continue;
                    if (conv_smem_bytes(a, t) > 120 * 1024) continue;
                    float samp[16];
                    HIPCHK(run(t));
                    for (int w = 0; w < nsamp; ++w) {
                        HIPCHK(hipMemsetAsync(c->flush, w, c->flush_bytes, s));
                        HIPCHK(hipEventRecord(e0, s));
                        HIPCHK(run(t));
                        HIPCHK(hipEventRecord(e1, s));
                        HIPCHK(hipEventSynchronize(e1));
                        HIPCHK(hipEventElapsedTime(&samp[w], e0, e1));
                    }
                    std::sort(samp, samp + nsamp);
                    if (samp[nsamp / 2] < best_ms) {
                        best_ms = samp[nsamp / 2];
                        best = t;
                    }
                }
            }
            // the split-bf16 kernels (conv_x3.hip) for large token counts: the timed launch is the elementwise pass + the GEMM
            // (not for the one-clip step at R = 32, the metric's workload: it stays on the exact-f32 instruction throughout)
            if ((long)a.B * a.Lout >= X3_MIN_ROWS && !(a.B == 1 && a.Lout <= 2048) && conv_x3_eligible(a) && a.x3) {
                static const int tb[][2] = {{4, 2}, {2, 2}, {4, 1}, {2, 1}, {8, 2}, {8, 1}, {4, 4}};
                const int nch32 = a.ntaps * (a.Cmain / 32) + a.Cskip / 32;
                for (auto& mn : tb)
                    for (int KS = 1; KS <= 8; KS *= 2) {
                        const ConvTile t{mn[0], mn[1], 48, KS, 0};
                        if (64 * t.NT > a.N && t.NT > 1) continue;
                        const long ntile = (long)a.B * ((a.Lout + 32 * t.MT - 1) / (32 * t.MT)) * ((a.N + 64 * t.NT - 1) / (64 * t.NT));
                        if (ntile * KS < 64) continue;
                        // K slices only while the tiles alone leave CUs idle, and never fewer than 6 chunks per slice
                        if (KS > 1 && (ntile * (KS / 2) >= 256 || nch32 / KS < 6 || (size_t)KS * a.B * a.Lout * a.N > slab_cap)) continue;
                        if (conv_x3_smem_bytes(a, t) > CONV_X3_MAX_LDS) continue;
                        float samp[16];
                        HIPCHK(run(t));
                        for (int w = 0; w < nsamp; ++w) {
                            HIPCHK(hipMemsetAsync(c->flush, w, c->flush_bytes, s));
                            HIPCHK(hipEventRecord(e0, s));
                            HIPCHK(run(t));
                            HIPCHK(hipEventRecord(e1, s));
                            HIPCHK(hipEventSynchronize(e1));
                            HIPCHK(hipEventElapsedTime(&samp[w], e0, e1));
                        }
                        std::sort(samp, samp + nsamp);
                        if (samp[nsamp / 2] < best_ms) {
                            best_ms = samp[nsamp / 2];
                            best = t;
                        }
                    }
            }
            // the window-staged 3x3 kernel (deep.hip, k_conv_win): GroupNorm / FiLM / SiLU once per element, all taps from LDS
            if (a.ntaps == 9 && (long)a.B * a.Lout >= 256) {
                static const int tw[][2] = {{1, 4}, {1, 2}, {2, 2}, {2, 4}};
                // block order by rule, not by timing (it moves L2 misses, not time: profiles/r04_conv_win_xcd_order.txt): one XCD per
                // weight column tile where the weights are the larger operand
                const int XM = wbytes >= abytes ? 1 : 0;
                for (auto& mn : tw)
                  for (int KS = 1; KS <= 4; KS *= 2) {
                    // K slices (round 6) only while the tiles alone leave CUs idle, slices of at least 3 x 9 chunks, within the slab
                    if (!conv_win_eligible(a, mn[0], mn[1], KS)) continue;
                    const ConvTile t{mn[0], mn[1], 80, KS, XM};
                    const long ntile = (long)a.B * ((a.Lout + 16 * t.MT - 1) / (16 * t.MT)) * (a.N / (16 * t.NT));
                    if (ntile * KS < 64) continue;
                    if (KS > 1 && (ntile * KS > 256 || a.Cmain / 16 / KS < 3 || (size_t)KS * a.B * a.Lout * a.N > slab_cap)) continue;
                    float samp[16];
                    HIPCHK(run(t));
                    for (int w = 0; w < nsamp; ++w) {
                        HIPCHK(hipMemsetAsync(c->flush, w, c->flush_bytes, s));
                        HIPCHK(hipEventRecord(e0, s));
                        HIPCHK(run(t));
                        HIPCHK(hipEventRecord(e1, s));
                        HIPCHK(hipEventSynchronize(e1));
                        HIPCHK(hipEventElapsedTime(&samp[w], e0, e1));
                    }
                    std::sort(samp, samp + nsamp);
                    if (samp[nsamp / 2] < best_ms) {
                        best_ms = samp[nsamp / 2];
                        best = t;
                    }
                }
            }
            // the 1x1 kernel of the large levels (deep.hip, k_conv_pw): rows normalised once into LDS, 8 waves side by side along N
            if (a.ntaps == 1 && (long)a.B * a.Lout >= 256) {
                for (int MT = 1; MT <= 2; ++MT)
                    for (int NTW = 1; NTW <= 2; ++NTW)
                      for (int wc : {1, 6, 4, 2}) {
                      for (int XM = 0; XM < 2; ++XM) {
                        if (!conv_pw_eligible(a, MT, NTW, wc)) continue;
                        const ConvTile t{MT, NTW, 96, wc, XM};
                        const int cols = 16 * NTW * (wc == 1 ? 8 : wc);
                        if (wc != 1 && a.N % cols) continue;                 // (narrower column tiles only where they divide N: whole tiles, a full grid)
                        if ((long)a.B * ((a.Lout + 16 * MT - 1) / (16 * MT)) * ((a.N + cols - 1) / cols) < 48) continue;
                        // (the XCD-aware block order only where the column groups tile the 8 XCDs and the weights dwarf the rows: measured -0.5 ... -0.8 us per
                        // launch at [128 x 1536, K = 512], +0.3 ... +0.8 at 512 tokens -- profiles/r06_conv_pw_xcd_order_ab.txt)
                        if (XM == 1 && (((a.N + cols - 1) / cols) % 8 || (long)a.N < 8L * a.B * a.Lout)) continue;
                        float samp[16];
                        HIPCHK(run(t));
                        for (int w = 0; w < nsamp; ++w) {
                            HIPCHK(hipMemsetAsync(c->flush, w, c->flush_bytes, s));
                            HIPCHK(hipEventRecord(e0, s));
                            HIPCHK(run(t));
                            HIPCHK(hipEventRecord(e1, s));
                            HIPCHK(hipEventSynchronize(e1));
                            HIPCHK(hipEventElapsedTime(&samp[w], e0, e1));
                        }
                        std::sort(samp, samp + nsamp);
                        if (samp[nsamp / 2] < best_ms) {
                            best_ms = samp[nsamp / 2];
                            best = t;
                        }
                      }
                    }
            }
            // the lean 1x1 kernel (lin.hip): wave tile 16 MT x 16 NT, NWV waves side by side along N, whole K per wave
            // (offered only with MTV_TUNE_LIN=1: in the step's graph its picks measured slower than k_conv's / k_conv_pw's on the same
            // shapes -- profiles/r04_per_op_rocprof.txt -- although they win the tuner's cold, isolated timing)
            static const bool tune_lin = getenv("MTV_TUNE_LIN") != nullptr;
            if (tune_lin && conv_lin_eligible(a)) {
                for (int MT = 1; MT <= 2; ++MT)
                    for (int NT = 1; NT <= 4; NT *= 2)
                        for (int NWV = 1; NWV <= 4; NWV *= 2) {
                            if (MT == 2 && a.Lout < 32) continue;
                            if (16 * NT * NWV > a.N && NWV > 1) continue;           // wider than the layer
                            const ConvTile t{MT, NT, 64, NWV, 0};
                            float samp[16];
                            HIPCHK(run(t));
                            for (int w = 0; w < nsamp; ++w) {
                                HIPCHK(hipMemsetAsync(c->flush, w, c->flush_bytes, s));
                                HIPCHK(hipEventRecord(e0, s));
                                HIPCHK(run(t));
                                HIPCHK(hipEventRecord(e1, s));
                                HIPCHK(hipEventSynchronize(e1));
                                HIPCHK(hipEventElapsedTime(&samp[w], e0, e1));
                            }
                            std::sort(samp, samp + nsamp);
                            if (samp[nsamp / 2] < best_ms) {
                                best_ms = samp[nsamp / 2];
                                best = t;
                            }
                        }
            }
            it = c->tune_cache.emplace(key, best).first;
            tune_cache_append(key, best);
        }
        op->t = it->second;
        p->ops[op->op_index].name = op->base_name + Builder::conv_tag(op->a, op->t);
    }
    return MTV_OK;
}

static int get_plan(mtv_ctx* c, int B, int mode, Plan** out) {
    auto it = c->plans.find({B, mode});
    if (it != c->plans.end()) {
        *out = it->second.get();
        return MTV_OK;
    }
    std::unique_ptr<Plan> p(new Plan());
    p->B = B;
    p->mode = mode;
    Builder b(c, p.get(), B, mode);
    int rc = b.build();
    if (rc != MTV_OK) return rc;
    *out = p.get();
    c->plans[{B, mode}] = std::move(p);
    return MTV_OK;
}

int run_ops(mtv_ctx* c, Plan* p, hipStream_t s) {
    for (auto& op : p->ops) {
        hipError_t e = op.run(s);
        if (e != hipSuccess) return fail(MTV_ERR_HIP, "launch " + op.name + ": " + hipGetErrorString(e));
    }
    return MTV_OK;
}

// =====================================================================================
// C ABI
// =====================================================================================
extern "C" {

const char* mtv_last_error(void) { return g_err.c_str(); }
int mtv_version(void) { return 1; }

}  // extern "C"

int ctx_init_common(mtv_ctx* c) {
    HIPCHK(hipGetDevice(&c->device));
    if (!c->cap_stream) HIPCHK(hipStreamCreateWithFlags(&c->cap_stream, hipStreamNonBlocking));
    HIPCHK(conv_init_attrs());     // dynamic LDS above 64 KB must be opted into once per kernel (never under capture)
    HIPCHK(attn_init_attrs());
    HIPCHK(deep_init_attrs());
    HIPCHK(deep_block_init_attrs());
    {   // Residency of the in-launch hand-offs (ADVICE r5 / VERDICT r5 item 6): their polls wait for workgroups of the SAME launch, which is only
        // live when the whole grid is resident together -- one 512-thread workgroup of those kernels per CU, so the bound is the CU count the
        // launch can use: a CPX / DPX partition reports fewer CUs here; a CU-masked stream is checked per call (check_stream_residency).
        int ncu = 0;
        HIPCHK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, c->device));
        static const int env_cus = []() { const char* e = getenv("MTV_RESIDENT_CUS"); return e ? atoi(e) : 0; }();
        const int want = g_resident_override > 0 ? g_resident_override : env_cus;
        c->resident_cus = want > 0 ? std::min(want, ncu > 0 ? ncu : want) : (ncu > 0 ? ncu : 256);
    }
    if (!c->fault_h) {     // device-side fault word (k_deep_block: a hand-off wait that timed out), host-mapped: checked without a copy
        HIPCHK(hipHostMalloc((void**)&c->fault_h, 64, hipHostMallocMapped));
        *c->fault_h = 0;
        HIPCHK(hipHostGetDevicePointer((void**)&c->fault_d, c->fault_h, 0));
    }
    return MTV_OK;
}

extern "C" {

int mtv_create(const mtv_config* cfg, mtv_ctx** out) {
    if (!cfg || !out) return fail(MTV_ERR_INVALID, "null argument");
    if (cfg->n_levels < 1 || cfg->n_levels > MTV_MAX_LEVELS) return fail(MTV_ERR_INVALID, "n_levels out of range");
    if (cfg->model_channels % 32) return fail(MTV_ERR_INVALID, "model_channels must be a multiple of 32 (GroupNorm32 + 16-channel K chunks)");
    if (cfg->max_batch < 1) return fail(MTV_ERR_INVALID, "max_batch < 1");
    if (cfg->out_channels < 1 || cfg->out_channels > 64) return fail(MTV_ERR_INVALID, "out_channels out of range");
    const int sh = cfg->n_levels - 1;
    if (cfg->res % (1 << sh) || cfg->frames % (1 << sh) || (cfg->frames >> sh) < 1 || (cfg->res >> sh) < 1)
        return fail(MTV_ERR_INVALID, "res/frames must survive len(channel_mult)-1 halvings (SURVEY.md fact 5)");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(MTV_ERR_HIP, "no HIP device available");
    std::unique_ptr<mtv_ctx> c(new mtv_ctx());
    c->cfg = *cfg;
    HIPCHK(hipGetDevice(&c->device));
    c->emb_dim = 4 * cfg->model_channels;
    c->lv = make_levels(cfg->res, cfg->frames, cfg->n_levels);
    int rc = build_structure(c.get());
    if (rc != MTV_OK) return rc;
    // gather tables
    for (int l = 0; l < cfg->n_levels; ++l) {
        auto up = [&](const std::vector<int>& h, int** d) -> int {
            void* p = nullptr;
            int r = c->dmalloc(&p, h.size() * sizeof(int));
            if (r != MTV_OK) return r;
            if (hipMemcpy(p, h.data(), h.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) return fail(MTV_ERR_HIP, "gather upload");
            *d = (int*)p;
            return MTV_OK;
        };
        // the kernels compute these indices arithmetically (geo_source); the tables stay as the reference
        // the formula is checked against here, and as the fallback if it ever disagreed
        const Level& lv = c->lv[l];
        int* d = nullptr;
        {
            const std::vector<int> h = make_gather3(lv, lv, false);
            c->geo_ok = c->geo_ok && table_matches_formula(h, 9, false, lv, lv);
            if ((rc = up(h, &d)) != MTV_OK) return rc;
        }
        c->g3.push_back(d);
        d = nullptr;
        int* d1 = nullptr;
        if (l + 1 < cfg->n_levels) {
            const std::vector<int> h3 = make_gather3(lv, c->lv[l + 1], true), h1 = make_gather_up1(lv, c->lv[l + 1]);
            c->geo_ok = c->geo_ok && table_matches_formula(h3, 9, true, lv, c->lv[l + 1]) && table_matches_formula(h1, 1, true, lv, c->lv[l + 1]);
            if ((rc = up(h3, &d)) != MTV_OK) return rc;
            if ((rc = up(h1, &d1)) != MTV_OK) return rc;
        }
        c->gup3.push_back(d);
        c->gup1.push_back(d1);
    }
    // statistics arena, staging buffers, sampler state
    c->stats_copy_doubles = (size_t)c->n_sites * cfg->max_batch * 192;
    c->stats_bytes = c->stats_copy_doubles * STAT_COPIES * sizeof(double);
    if ((rc = c->dmalloc((void**)&c->stats, 2 * c->stats_bytes)) != MTV_OK) return rc;      // one arena per step parity
    const int L = c->lv[0].L, RR = c->lv[0].b1, mb = cfg->max_batch;
    if ((rc = c->dmalloc((void**)&c->xin, (size_t)mb * 4 * L * 4)) != MTV_OK) return rc;
    if ((rc = c->dmalloc((void**)&c->condin, (size_t)mb * 8 * L * 4)) != MTV_OK) return rc;
    if ((rc = c->dmalloc((void**)&c->icin, (size_t)mb * 4 * RR * 4)) != MTV_OK) return rc;
    if ((rc = c->dmalloc((void**)&c->eps, (size_t)mb * cfg->out_channels * L * 4)) != MTV_OK) return rc;
    if ((rc = c->dmalloc((void**)&c->tbuf, (size_t)mb * 8)) != MTV_OK) return rc;
    if ((rc = c->dmalloc((void**)&c->d_counter, 16)) != MTV_OK) return rc;
    HIPCHK(hipMemset(c->d_counter, 0, 16));
    if ((rc = c->dmalloc((void**)&c->d_fuse, 2 * sizeof(DdimFuse))) != MTV_OK) return rc;
    HIPCHK(hipMemset(c->d_fuse, 0, 2 * sizeof(DdimFuse)));
    {
        // timestep_embedding frequencies, fp32 like the reference (diffusionmodules.py:118-121)
        const int half = cfg->model_channels / 2;
        std::vector<float> fr(half);
        const float nl = (float)(-std::log(10000.0));
        for (int k = 0; k < half; ++k) {
            float v = nl * (float)k;
            v = v / (float)half;
            fr[k] = expf(v);
        }
        if ((rc = c->dmalloc((void**)&c->freqs, half * 4)) != MTV_OK) return rc;
        HIPCHK(hipMemcpy(c->freqs, fr.data(), half * 4, hipMemcpyHostToDevice));
    }
    if ((rc = ctx_init_common(c.get())) != MTV_OK) return rc;
    // build the batch-1 plan now: registers every weight slot and fills the work accounting
    c->accounting = true;
    Plan* p1 = nullptr;
    rc = get_plan(c.get(), 1, MODE_FORWARD, &p1);
    c->accounting = false;
    if (rc != MTV_OK) return rc;
    Plan* ps = nullptr;
    if ((rc = get_plan(c.get(), 1, MODE_STEP0, &ps)) != MTV_OK) return rc;
    c->work.n_launches_step = (int)ps->ops.size();
    *out = c.release();
    return MTV_OK;
}

int mtv_destroy(mtv_ctx* c) {
    delete c;        // ~mtv_ctx binds the context's device, drains it and frees everything
    return MTV_OK;
}

int mtv_num_weights(const mtv_ctx* c) { return c ? (int)c->slots.size() : 0; }

int mtv_weight_info(const mtv_ctx* c, int index, char* key_out, int key_cap, int* ndim_out, int64_t shape_out[4]) {
    if (!c || index < 0 || index >= (int)c->slots.size()) return fail(MTV_ERR_INVALID, "weight index out of range");
    const WSlot& s = c->slots[index];
    if (key_out && key_cap > 0) {
        std::strncpy(key_out, s.key.c_str(), key_cap - 1);
        key_out[key_cap - 1] = 0;
    }
    if (ndim_out) *ndim_out = (int)s.shape.size();
    if (shape_out)
        for (size_t i = 0; i < 4; ++i) shape_out[i] = i < s.shape.size() ? s.shape[i] : 1;
    return MTV_OK;
}

int mtv_weights_missing(const mtv_ctx* c) {
    int n = 0;
    for (auto& s : c->slots) n += s.loaded ? 0 : 1;
    return n;
}

int mtv_load_weight(mtv_ctx* c, const char* key, const float* data, int ndim, const int64_t* shape) {
    if (!c || !key || !data || !shape) return fail(MTV_ERR_INVALID, "null argument");
    std::string k(key);
    const std::string pre = "diffusion_model.";
    if (k.compare(0, pre.size(), pre) == 0) k = k.substr(pre.size());
    if (k.compare(0, 17, "output_bg_blocks.") == 0 || k.compare(0, 16, "output_bg_attns.") == 0) return MTV_IGNORED;
    auto it = c->slot_index.find(k);
    if (it == c->slot_index.end()) return fail(MTV_ERR_WEIGHT, "unknown weight key: " + k);
    WSlot& s = c->slots[it->second];
    bool ok = ndim == (int)s.shape.size();
    size_t n = 1;
    for (int i = 0; ok && i < ndim; ++i) {
        ok = shape[i] == s.shape[i];
        n *= (size_t)shape[i];
    }
    if (!ok) {
        std::string m = "shape mismatch for " + k + ": expected [";
        for (auto d : s.shape) m += std::to_string(d) + ",";
        m += "] got [";
        for (int i = 0; i < ndim; ++i) m += std::to_string(shape[i]) + ",";
        return fail(MTV_ERR_WEIGHT, m + "]");
    }
    HIPCHK(hipSetDevice(c->device));
    if (s.role == ROLE_COPY) {
        HIPCHK(hipMemcpy(s.dst, data, n * sizeof(float), hipMemcpyDefault));
    } else if (s.role == ROLE_REPEAT || s.role == ROLE_QKV_HEADS || s.role == ROLE_KV_HEADS) {
        if (c->staging_floats < n) {
            if (c->staging) (void)hipFree(c->staging);
            c->staging = nullptr;
            c->staging_floats = 0;
            HIPCHK(hipMalloc((void**)&c->staging, n * sizeof(float)));
            c->staging_floats = n;
        }
        HIPCHK(hipMemcpy(c->staging, data, n * sizeof(float), hipMemcpyDefault));
        if (s.role == ROLE_REPEAT) HIPCHK(launch_repeat(c->staging, s.dst, (int)n, s.aux, nullptr));
        else if (s.role == ROLE_KV_HEADS)
            HIPCHK(launch_repack_heads(c->staging, s.dst, (int)s.shape[0] / (s.aux / 2), s.aux / 2, (int)s.shape[1], s.ld, 2, s.aux & 1, nullptr));
        else HIPCHK(launch_repack_qkv(c->staging, s.dst, (int)s.shape[0] / (3 * s.aux), s.aux, (int)s.shape[1], s.ld, nullptr));
        HIPCHK(hipStreamSynchronize(nullptr));
    } else {
        if (c->staging_floats < n) {
            if (c->staging) (void)hipFree(c->staging);
            c->staging = nullptr;
            c->staging_floats = 0;
            HIPCHK(hipMalloc((void**)&c->staging, n * sizeof(float)));
            c->staging_floats = n;
        }
        HIPCHK(hipMemcpy(c->staging, data, n * sizeof(float), hipMemcpyDefault));
        const int N = (int)s.shape[0], C = (int)s.shape[1];
        const int ntaps = (int)(n / ((size_t)N * C));
        HIPCHK(launch_repack_conv(c->staging, s.dst, N, C, ntaps, s.ld, nullptr));
        if (s.dst2) HIPCHK(hipMemcpyAsync(s.dst2, c->staging, n * sizeof(float), hipMemcpyDeviceToDevice, nullptr));
        if (s.dst3) HIPCHK(launch_repack_pw(c->staging, s.dst3, N, C, nullptr));
        HIPCHK(hipStreamSynchronize(nullptr));
    }
    s.loaded = true;
    for (auto& kv : c->w3) kv.second.dirty = true;      // (rebuilt before the next run: check_ready)
    for (auto& kv : c->wdeep) kv.second.dirty = true;
    return MTV_OK;
}

}  // extern "C"

// test hook (mtv_debug_arm_fault): what a timed-out poll does -- the same system-scope store to the host-mapped fault word
__global__ void k_debug_raise_fault(int* fault) { __hip_atomic_store(fault, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

static const char* fault_text() {
    return "an in-launch hand-off timed out (a poll inside k_deep_block, or the tagged completion of a deep tensor inside k_deep_conv / k_deep_attn, waited "
           "2^21 retries for workgroups of its own launch): results of that call and of every call since are invalid -- destroy this context.  The launch was "
           "not co-resident (CU-masked queue, partitioned device, another stream's kernel on the same CUs?): MTV_RESIDENT_CUS=<n> plans for n CUs, "
           "MTV_DEEP_NO_BLOCK=1 + MTV_DEEP_FIN_PASS=1 select the forms without in-launch hand-offs";
}

// A stream created with a CU mask (hipExtStreamCreateWithCUMask) runs its launches on fewer CUs than the device reports: a plan whose hand-off
// grids were sized for more must not be replayed there (its polls would depend on dispatch order for progress).
static int check_stream_residency(mtv_ctx* c, hipStream_t s) {
    uint32_t mask[32] = {0};
    if (hipExtStreamGetCUMask(s, 32, mask) != hipSuccess) { (void)hipGetLastError(); return MTV_OK; }     // (no mask to read: the device's CU count stands)
    int n = 0;
    for (uint32_t m : mask) n += __builtin_popcount(m);
    if (n > 0 && n < c->resident_cus)
        return fail(MTV_ERR_STATE, "the stream's CU mask leaves " + std::to_string(n) + " CUs but this context planned its in-launch hand-offs for " +
                                   std::to_string(c->resident_cus) + ": create the context with MTV_RESIDENT_CUS=" + std::to_string(n) + " (or mtv_debug_resident_cus) first");
    return MTV_OK;
}

int check_ready(mtv_ctx* c, int batch) {
    if (!c) return fail(MTV_ERR_INVALID, "null context");
    if (c->fault_h && *(volatile int*)c->fault_h)
        return fail(MTV_ERR_STATE, fault_text());
    if (batch < 1 || batch > c->cfg.max_batch) return fail(MTV_ERR_STATE, "batch outside [1, max_batch]");
    const int miss = mtv_weights_missing(c);
    if (miss) {
        std::string first;
        for (auto& s : c->slots)
            if (!s.loaded) { first = s.key; break; }
        return fail(MTV_ERR_WEIGHT, std::to_string(miss) + " weights not loaded (first: " + first + ")");
    }
    bool any = false;
    for (auto& kv : c->w3)
        if (kv.second.dirty) {
            HIPCHK(launch_split_w3(kv.first, kv.second.p, kv.second.plane_bytes, 0, kv.second.K, kv.second.ld, nullptr));
            kv.second.dirty = false;
            any = true;
        }
    for (auto& kv : c->wdeep)
        if (kv.second.dirty) {
            HIPCHK(launch_deep_repack(kv.second.W, kv.second.ldw, kv.second.p, kv.second.lay, kv.second.NT, nullptr));
            kv.second.dirty = false;
            any = true;
        }
    if (any) HIPCHK(hipStreamSynchronize(nullptr));
    return MTV_OK;
}

extern "C" {

static int stage_inputs(mtv_ctx* c, const float* x, const float* cond, const float* image_cond, int ic_len, int B, hipStream_t s) {
    const int L = c->lv[0].L, RR = c->lv[0].b1;
    if (ic_len < RR) return fail(MTV_ERR_INVALID, "image_cond has fewer than R*R tokens");
    if (x && x != c->xin) HIPCHK(hipMemcpyAsync(c->xin, x, (size_t)B * 4 * L * 4, hipMemcpyDeviceToDevice, s));
    HIPCHK(hipMemcpyAsync(c->condin, cond, (size_t)B * 8 * L * 4, hipMemcpyDeviceToDevice, s));
    HIPCHK(hipMemcpy2DAsync(c->icin, (size_t)RR * 4, image_cond, (size_t)ic_len * 4, (size_t)RR * 4, (size_t)B * 4, hipMemcpyDeviceToDevice, s));
    return MTV_OK;
}

}  // extern "C"

int capture(mtv_ctx* c, Plan* p, hipGraphExec_t* out) {
    hipGraph_t g = nullptr;
    HIPCHK(hipStreamBeginCapture(c->cap_stream, hipStreamCaptureModeThreadLocal));
    int rc = run_ops(c, p, c->cap_stream);
    hipError_t e = hipStreamEndCapture(c->cap_stream, &g);
    if (rc != MTV_OK) {
        if (g) (void)hipGraphDestroy(g);
        return rc;
    }
    if (e != hipSuccess) return fail(MTV_ERR_HIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
    e = hipGraphInstantiate(out, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (e != hipSuccess) return fail(MTV_ERR_HIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(e));
    return MTV_OK;
}

extern "C" {

int mtv_forward(mtv_ctx* c, const float* x, const float* cond, const float* image_cond, int image_cond_len,
                const int64_t* timesteps, float* eps_out, int batch, void* stream) {
    int rc = check_ready(c, batch);
    if (rc != MTV_OK) return rc;
    if (!x || !cond || !image_cond || !timesteps || !eps_out) return fail(MTV_ERR_INVALID, "null tensor pointer");
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(hipSetDevice(c->device));
    if ((rc = check_stream_residency(c, s)) != MTV_OK) return rc;
    Plan* p = nullptr;
    if ((rc = get_plan(c, batch, MODE_FORWARD, &p)) != MTV_OK) return rc;
    if ((rc = autotune(c, p, s)) != MTV_OK) return rc;
    if ((rc = stage_inputs(c, x, cond, image_cond, image_cond_len, batch, s)) != MTV_OK) return rc;
    HIPCHK(hipMemcpyAsync(c->tbuf, timesteps, (size_t)batch * 8, hipMemcpyDeviceToDevice, s));
    if (c->eager) {
        if ((rc = run_ops(c, p, s)) != MTV_OK) return rc;
    } else {
        if (!p->g_forward && (rc = capture(c, p, &p->g_forward)) != MTV_OK) return rc;
        HIPCHK(hipGraphLaunch(p->g_forward, s));
    }
    HIPCHK(hipMemcpyAsync(eps_out, c->eps, (size_t)batch * c->cfg.out_channels * c->lv[0].L * 4, hipMemcpyDeviceToDevice, s));
    if (c->debug_fault_armed) { c->debug_fault_armed = false; hipLaunchKernelGGL(k_debug_raise_fault, dim3(1), dim3(1), 0, s, c->fault_d); HIPCHK(hipGetLastError()); }
    return MTV_OK;
}

// Sampler set-up shared by mtv_ddim_sample and mtv_profile_step: uploads the step table and the head conv's
// hand-over records through pinned staging (no stream synchronisation), computes the FiLM row of every step
// (time-embedding MLP + all emb_layers, unet.py:1011-1012 and :193, batched over the steps), zeroes both statistics
// arenas, packs the UNet input and resets the step counter.
static int sampler_setup(mtv_ctx* c, int batch, const float* noise, const mtv_ddim_step* steps, int n_steps, hipStream_t s) {
    static_assert(sizeof(DdimStep) == sizeof(mtv_ddim_step), "step layout");
    const int mc = c->cfg.model_channels, emb = c->emb_dim;
    if (n_steps > c->d_steps_cap) {     // tables grow geometrically; growing does not invalidate captured graphs
        HIPCHK(hipStreamSynchronize(s)); // (the head conv reaches them through the DdimFuse records, not through kernel args)
        c->free_step_tables();
        int cap = 256;
        while (cap < n_steps) cap *= 2;
        HIPCHK(hipMalloc((void**)&c->d_steps, (size_t)cap * sizeof(DdimStep)));
        HIPCHK(hipMalloc((void**)&c->film_tab, (size_t)cap * c->film_total * 4));
        HIPCHK(hipMalloc((void**)&c->sin_steps, (size_t)cap * mc * 4));
        HIPCHK(hipMalloc((void**)&c->e0_steps, (size_t)cap * emb * 4));
        HIPCHK(hipMalloc((void**)&c->e1_steps, (size_t)cap * emb * 4));
        c->d_steps_cap = cap;
    }
    // pinned staging, double buffered: slot k is reused only after the copy issued from it two calls ago has run
    const size_t need = 2 * sizeof(DdimFuse) + (size_t)n_steps * sizeof(DdimStep);
    if (need > c->pin_bytes) {
        HIPCHK(hipStreamSynchronize(s));
        size_t cap = 64 << 10;
        while (cap < need) cap *= 2;
        for (int i = 0; i < 2; ++i) {
            if (c->h_pin[i]) (void)hipHostFree(c->h_pin[i]);
            c->h_pin[i] = nullptr;
            HIPCHK(hipHostMalloc((void**)&c->h_pin[i], cap, hipHostMallocDefault));
            if (!c->pin_ev[i]) HIPCHK(hipEventCreateWithFlags(&c->pin_ev[i], hipEventDisableTiming));
        }
        c->pin_bytes = cap;
    }
    const int slot = (int)(c->pin_turn++ & 1);
    HIPCHK(hipEventSynchronize(c->pin_ev[slot]));
    DdimFuse* hf = reinterpret_cast<DdimFuse*>(c->h_pin[slot]);
    const size_t arena_floats = c->stats_bytes / 4;
    for (int par = 0; par < 2; ++par) {
        DdimFuse f{};
        f.x = c->xin;
        f.h0 = c->bufs["act.h0"];
        f.noise = noise;
        f.steps = c->d_steps;
        f.counter = c->d_counter;
        f.done = c->d_counter + 1;
        f.n_per_draw = (long long)batch * 4 * c->lv[0].L;
        f.film_tab = c->film_tab;
        f.film_out = c->bufs["emb.film_step"];
        f.film_total = c->film_total;
        f.n_steps = n_steps;
        f.zero_arena = reinterpret_cast<float*>(c->stats) + (size_t)(1 - par) * arena_floats;
        f.zero_vec4 = (long long)(arena_floats / 4);
        hf[par] = f;
    }
    std::memcpy(c->h_pin[slot] + 2 * sizeof(DdimFuse), steps, (size_t)n_steps * sizeof(DdimStep));
    HIPCHK(hipMemcpyAsync(c->d_fuse, hf, 2 * sizeof(DdimFuse), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(c->d_steps, c->h_pin[slot] + 2 * sizeof(DdimFuse), (size_t)n_steps * sizeof(DdimStep), hipMemcpyHostToDevice, s));
    HIPCHK(hipEventRecord(c->pin_ev[slot], s));
    // FiLM rows of all steps: sinusoid -> time_embed.0 -> SiLU -> time_embed.2 -> SiLU -> [all emb_layers]
    HIPCHK(launch_step_sinusoid(c->d_steps, n_steps, c->freqs, c->sin_steps, mc / 2, s));
    LinearArgs l0{c->sin_steps, c->bufs["w.time_embed.0.weight"], c->bufs["w.time_embed.0.bias"], c->e0_steps, n_steps, mc, emb, emb, 0};
    HIPCHK(launch_linear_rows(l0, s));
    LinearArgs l2{c->e0_steps, c->bufs["w.time_embed.2.weight"], c->bufs["w.time_embed.2.bias"], c->e1_steps, n_steps, emb, emb, emb, 1};
    HIPCHK(launch_linear_rows(l2, s));
    LinearArgs lf{c->e1_steps, c->bufs["w.film"], c->bufs["w.film_bias"], c->film_tab, n_steps, emb, c->film_total, c->film_total, 1};
    HIPCHK(launch_linear_rows(lf, s));
    HIPCHK(hipMemsetAsync(c->stats, 0, 2 * c->stats_bytes, s));
    HIPCHK(launch_pack_input(c->xin, c->condin, c->icin, c->lv[0].b1, c->bufs["act.h0"], batch, c->lv[0].L, c->lv[0].b1, s));
    HIPCHK(launch_ddim_init(c->d_fuse, s));
    return MTV_OK;
}

int mtv_ddim_sample(mtv_ctx* c, float* x_io, const float* cond, const float* image_cond, int image_cond_len,
                    const float* noise, int n_noise, const mtv_ddim_step* steps, int n_steps, int batch, void* stream) {
    int rc = check_ready(c, batch);
    if (rc != MTV_OK) return rc;
    if (!x_io || !cond || !image_cond || !steps || n_steps < 1) return fail(MTV_ERR_INVALID, "null/empty argument");
    if (c->cfg.out_channels != 4) return fail(MTV_ERR_INVALID, "DDIM loop needs out_channels == 4 (eps has the shape of x)");
    for (int i = 0; i < n_steps; ++i) {
        if (steps[i].noise_index >= n_noise) return fail(MTV_ERR_INVALID, "noise_index out of range");
        if (steps[i].noise_index >= 0 && !noise) return fail(MTV_ERR_INVALID, "noise required");
        if (steps[i].t < 0) return fail(MTV_ERR_INVALID, "negative timestep");
    }
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(hipSetDevice(c->device));
    if ((rc = check_stream_residency(c, s)) != MTV_OK) return rc;
    Plan* p[2] = {nullptr, nullptr};
    for (int par = 0; par < 2; ++par) {
        if ((rc = get_plan(c, batch, par ? MODE_STEP1 : MODE_STEP0, &p[par])) != MTV_OK) return rc;
        if ((rc = autotune(c, p[par], s)) != MTV_OK) return rc;
    }
    if ((rc = stage_inputs(c, x_io, cond, image_cond, image_cond_len, batch, s)) != MTV_OK) return rc;
    if ((rc = sampler_setup(c, batch, noise, steps, n_steps, s)) != MTV_OK) return rc;
    // the loop: ONE graph replay per step (even / odd steps alternate between the two statistics arenas); the
    // step index, its coefficients, its FiLM row and its noise slab are all resolved on the device -- so M consecutive steps can just as
    // well be ONE graph (round 5: M = 8 by default, MTV_STEPS_PER_GRAPH; the hand-over between two graph launches is a few microseconds of
    // idle GPU per launch: profiles/r05_steps_per_graph_ab.txt)
    static const int M = []() { const char* e = getenv("MTV_STEPS_PER_GRAPH"); const int v = e ? atoi(e) : 8; return v < 1 ? 1 : (v > 32 ? 32 : v & ~1 ? v & ~1 : 1); }();
    int i0 = 0;
    if (!c->eager) {
        // every graph this context can need is captured by its FIRST call (the tail of a later call must never pay a capture)
        for (int par = 0; par < 2; ++par)
            if (!p[par]->g_forward && (rc = capture(c, p[par], &p[par]->g_forward)) != MTV_OK) return rc;
        if (M >= 2 && n_steps >= M) {            // (captured by the first call that can use it: a 4-step call never pays for ~1.2k kernel nodes -- ADVICE r5)
            auto mit = c->multi.find({batch, M});
            if (mit == c->multi.end()) mit = c->multi.emplace(std::make_pair(batch, M), (hipGraphExec_t) nullptr).first;
            hipGraphExec_t& gm = mit->second;
            if (!gm) {
                hipGraph_t g = nullptr;
                HIPCHK(hipStreamBeginCapture(c->cap_stream, hipStreamCaptureModeThreadLocal));
                int rcc = MTV_OK;
                for (int k = 0; k < M && rcc == MTV_OK; ++k) rcc = run_ops(c, p[k & 1], c->cap_stream);
                const hipError_t e = hipStreamEndCapture(c->cap_stream, &g);
                if (rcc != MTV_OK) { if (g) (void)hipGraphDestroy(g); c->multi.erase(mit); return rcc; }
                if (e != hipSuccess) { c->multi.erase(mit); return fail(MTV_ERR_HIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(e)); }
                const hipError_t e2 = hipGraphInstantiate(&gm, g, nullptr, nullptr, 0);
                (void)hipGraphDestroy(g);
                if (e2 != hipSuccess) { c->multi.erase(mit); return fail(MTV_ERR_HIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(e2)); }
            }
            for (; n_steps - i0 >= M; i0 += M) HIPCHK(hipGraphLaunch(gm, s));     // (M is even: the next step is an even one again)
        }
    }
    for (int i = i0; i < n_steps; ++i) {
        Plan* q = p[i & 1];
        if (c->eager) {
            if ((rc = run_ops(c, q, s)) != MTV_OK) return rc;
        } else {
            if (!q->g_forward && (rc = capture(c, q, &q->g_forward)) != MTV_OK) return rc;
            HIPCHK(hipGraphLaunch(q->g_forward, s));
        }
    }
    HIPCHK(hipMemcpyAsync(x_io, c->xin, (size_t)batch * 4 * c->lv[0].L * 4, hipMemcpyDeviceToDevice, s));
    if (c->debug_fault_armed) { c->debug_fault_armed = false; hipLaunchKernelGGL(k_debug_raise_fault, dim3(1), dim3(1), 0, s, c->fault_d); HIPCHK(hipGetLastError()); }
    return MTV_OK;
}

/* include/mtv_hip.h: the fault word of the in-launch hand-offs, readable at any time; meaningful for a call once the stream it ran on has drained */
int mtv_check_fault(mtv_ctx* c) {
    if (!c) return fail(MTV_ERR_INVALID, "null context");
    if (c->fault_h && *(volatile int*)c->fault_h) return fail(MTV_ERR_STATE, fault_text());
    return MTV_OK;
}

int mtv_debug_arm_fault(mtv_ctx* c) {
    if (!c || !c->fault_d) return fail(MTV_ERR_INVALID, "null context");
    c->debug_fault_armed = true;
    return MTV_OK;
}

int mtv_debug_resident_cus(int n) {
    if (n < 0 || n > 4096) return fail(MTV_ERR_INVALID, "resident CUs must be 0 (the device's count / MTV_RESIDENT_CUS) or a positive count");
    g_resident_override = n;
    return MTV_OK;
}

int mtv_resident_cus(const mtv_ctx* c) { return c ? c->resident_cus : fail(MTV_ERR_INVALID, "null context"); }

static int profile_plan(mtv_ctx* c, int batch, int mode, int iters, mtv_op_time* out, int cap, int* n_out, hipStream_t s) {
    int rc = check_ready(c, batch);
    if (rc != MTV_OK) return rc;
    if (iters < 1 || !n_out) return fail(MTV_ERR_INVALID, "bad argument");
    HIPCHK(hipSetDevice(c->device));
    Plan* p = nullptr;
    if ((rc = get_plan(c, batch, mode, &p)) != MTV_OK) return rc;
    if ((rc = autotune(c, p, s)) != MTV_OK) return rc;
    const int n = (int)p->ops.size();
    *n_out = n;
    if (!out) return MTV_OK;
    if (cap < n) return fail(MTV_ERR_INVALID, "profile table too small");
    // one event between consecutive launches (n+1 in all): launch i is charged the interval between the
    // event before it and the event after it, so the events' own cost is paid once per launch, not twice
    std::vector<hipEvent_t> ev((size_t)n + 1);
    for (auto& e : ev) HIPCHK(hipEventCreate(&e));
    std::vector<double> acc(n, 0.0);
    const mtv_ddim_step one{500, 0, 1.0f, 0.0f, 1.0f, 0.0f, 0.0f, -1};    // a 1-entry step table for the step plan
    for (int it = 0; it < iters; ++it) {
        if (mode != MODE_FORWARD && (rc = sampler_setup(c, batch, nullptr, &one, 1, s)) != MTV_OK) return rc;
        HIPCHK(hipEventRecord(ev[0], s));
        for (int i = 0; i < n; ++i) {
            hipError_t e = p->ops[i].run(s);
            if (e != hipSuccess) return fail(MTV_ERR_HIP, "launch " + p->ops[i].name + ": " + hipGetErrorString(e));
            HIPCHK(hipEventRecord(ev[i + 1], s));
        }
        HIPCHK(hipStreamSynchronize(s));
        for (int i = 0; i < n; ++i) {
            float ms = 0.f;
            HIPCHK(hipEventElapsedTime(&ms, ev[i], ev[i + 1]));
            acc[i] += ms;
        }
    }
    for (auto& e : ev) (void)hipEventDestroy(e);
    for (int i = 0; i < n; ++i) {
        std::memset(&out[i], 0, sizeof(mtv_op_time));
        std::strncpy(out[i].name, p->ops[i].name.c_str(), sizeof(out[i].name) - 1);
        out[i].ms = (float)(acc[i] / iters);
        out[i].flops = p->ops[i].flops;
        out[i].bytes = p->ops[i].bytes;
    }
    return MTV_OK;
}

int mtv_profile_forward(mtv_ctx* c, int batch, int iters, mtv_op_time* out, int cap, int* n_out, void* stream) {
    return profile_plan(c, batch, MODE_FORWARD, iters, out, cap, n_out, (hipStream_t)stream);
}

int mtv_profile_step(mtv_ctx* c, int batch, int iters, mtv_op_time* out, int cap, int* n_out, void* stream) {
    return profile_plan(c, batch, MODE_STEP0, iters, out, cap, n_out, (hipStream_t)stream);
}

// Diagnostic (needs the -DMTV_ABLATE=64 build of conv.hip and MTV_STAMPS=1 in the environment at plan-build time):
// runs one sampler step with plain launches, in plan order like the real chain, and writes the in-kernel phase
// timestamps (s_memtime ticks) of four sampled workgroups of every conv to `path`.
int mtv_debug_stamps(mtv_ctx* c, int batch, const char* path, void* stream) {
    int rc = check_ready(c, batch);
    if (rc != MTV_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(hipSetDevice(c->device));
    Plan* p = nullptr;
    if ((rc = get_plan(c, batch, MODE_STEP0, &p)) != MTV_OK) return rc;
    if ((rc = autotune(c, p, s)) != MTV_OK) return rc;
    const mtv_ddim_step one{500, 0, 1.0f, 0.0f, 1.0f, 0.0f, 0.0f, -1};
    for (int it = 0; it < 3; ++it) {
        if ((rc = sampler_setup(c, batch, nullptr, &one, 1, s)) != MTV_OK) return rc;
        for (auto& op : p->convs)
            if (op->a.dbg) HIPCHK(hipMemsetAsync(op->a.dbg, 0, 512, s));
        for (auto& ad : p->attn_dbg) HIPCHK(hipMemsetAsync(ad.second, 0, 512, s));
        if ((rc = run_ops(c, p, s)) != MTV_OK) return rc;
        HIPCHK(hipStreamSynchronize(s));
    }
    FILE* f = fopen(path, "w");
    if (!f) return fail(MTV_ERR_INVALID, "cannot open stamp file");
    fprintf(f, "# per conv: name, then 4 sampled blocks (first, second, middle, last) x stamps[0..15]: 7=entry 0=decoded 8=tables built 9=barrier passed 10=first chunk requested 1=ring issued 2=prologue done 3=K loop done 4=reduced 5=epilogue stored 6=statistics done (0 = path not taken)\n");
    for (auto& op : p->convs) {
        if (!op->a.dbg) continue;
        unsigned long long h[64];
        HIPCHK(hipMemcpy(h, op->a.dbg, sizeof h, hipMemcpyDeviceToHost));
        fprintf(f, "%s", p->ops[op->op_index].name.c_str());
        for (int i = 0; i < 64; ++i) fprintf(f, " %llu", h[i]);
        fprintf(f, "\n");
    }
    // attention launches (kernels.hip built with -DMTV_ATT_STAMP): 0=entry 1=decoded 2=first tile in LDS 3=key loop done
    // 4=key parts merged 5=output stored; 8/9/10 = ticks summed over the key blocks: block math / LDS store of the next
    // tile (incl. the wait for its loads) / barrier
    for (auto& ad : p->attn_dbg) {
        unsigned long long h[64];
        HIPCHK(hipMemcpy(h, ad.second, sizeof h, hipMemcpyDeviceToHost));
        fprintf(f, "%s", ad.first.c_str());
        for (int i = 0; i < 64; ++i) fprintf(f, " %llu", h[i]);
        fprintf(f, "\n");
    }
    fclose(f);
    return MTV_OK;
}

int mtv_debug_tap(mtv_ctx* c, const char* name, float* dst, int64_t cap, int* tokens_out, int* channels_out) {
    if (!c || !name) return fail(MTV_ERR_INVALID, "null argument");
    auto it = c->taps.find(name);
    if (it == c->taps.end()) return fail(MTV_ERR_INVALID, std::string("unknown tap: ") + name);
    const int L = c->lv[it->second.first].L, C = it->second.second;
    if (tokens_out) *tokens_out = L;
    if (channels_out) *channels_out = C;
    if (dst) {
        int B = (int)(cap / ((int64_t)L * C));
        if (B < 1) return fail(MTV_ERR_INVALID, "tap destination too small");
        if (B > c->cfg.max_batch) B = c->cfg.max_batch;
        HIPCHK(hipDeviceSynchronize());
        const float* src = c->bufs["tap." + std::string(name)];
        auto sl = c->tap_slabs.find(name);
        if (sl == c->tap_slabs.end() || sl->second.first <= 1) {
            HIPCHK(hipMemcpy(dst, src, (size_t)B * L * C * 4, hipMemcpyDefault));
        } else {
            // a tensor of the deep levels is the sum of its K-slice slabs (deep.hip): added up here in slab order, as its consumers do
            if (B > 2) B = 2;                              // (a slab holds at most two clips: the deep path only runs with B <= 2)
            const size_t n = (size_t)B * L * C;
            std::vector<float> acc(n), one(n);
            HIPCHK(hipMemcpy(acc.data(), src, n * 4, hipMemcpyDeviceToHost));
            for (int k = 1; k < sl->second.first; ++k) {
                HIPCHK(hipMemcpy(one.data(), src + (size_t)k * sl->second.second, n * 4, hipMemcpyDeviceToHost));
                for (size_t e = 0; e < n; ++e) acc[e] += one[e];
            }
            HIPCHK(hipMemcpy(dst, acc.data(), n * 4, hipMemcpyDefault));
        }
    }
    return MTV_OK;
}

int mtv_get_work(const mtv_ctx* c, mtv_work* out) {
    if (!c || !out) return fail(MTV_ERR_INVALID, "null argument");
    *out = c->work;
    return MTV_OK;
}

int mtv_selftest_geometry(int res, int frames, int n_levels) {
    if (res <= 0 || frames <= 0 || n_levels <= 0 || n_levels > 8 || (res >> (n_levels - 1)) <= 0 || (frames >> (n_levels - 1)) <= 0)
        return fail(MTV_ERR_INVALID, "selftest_geometry: bad geometry");
    const std::vector<Level> lv = make_levels(res, frames, n_levels);
    for (int l = 0; l < n_levels; ++l) {
        if (!table_matches_formula(make_gather3(lv[l], lv[l], false), 9, false, lv[l], lv[l])) return l + 1;
        if (l + 1 < n_levels) {
            if (!table_matches_formula(make_gather3(lv[l], lv[l + 1], true), 9, true, lv[l], lv[l + 1])) return l + 1;
            if (!table_matches_formula(make_gather_up1(lv[l], lv[l + 1]), 1, true, lv[l], lv[l + 1])) return l + 1;
        }
    }
    return 0;
}

/* Host-only self-test of the deep levels' row tables (deep.hip, deep_rowtab): for every level of at most 128 tokens, every row grouping
 * and tap count, and the upsampling variant, each table entry must name the source token the explicitly constructed im2col tables
 * name (or the zero row where they pad).  0 = all agree, else 1 + the level that does not. */
int mtv_selftest_win(int res, int frames, int n_levels) {
    if (res <= 0 || frames <= 0 || n_levels <= 0 || n_levels > 8 || (res >> (n_levels - 1)) <= 0 || (frames >> (n_levels - 1)) <= 0)
        return fail(MTV_ERR_INVALID, "selftest_win: bad geometry");
    const std::vector<Level> lv = make_levels(res, frames, n_levels);
    for (int l = 0; l < n_levels; ++l)
        for (int up = 0; up < 2; ++up) {
            if (up && l + 1 >= n_levels) continue;
            const Level& src = up ? lv[l + 1] : lv[l];
            if (conv_win_selftest(lv[l].r, lv[l].t, up != 0, lv[l].L, src.L) != 0) return l + 1;
        }
    return 0;
}

int mtv_selftest_deep(int res, int frames, int n_levels) {
    if (res <= 0 || frames <= 0 || n_levels <= 0 || n_levels > 8 || (res >> (n_levels - 1)) <= 0 || (frames >> (n_levels - 1)) <= 0)
        return fail(MTV_ERR_INVALID, "selftest_deep: bad geometry");
    const std::vector<Level> lv = make_levels(res, frames, n_levels);
    for (int l = 0; l < n_levels; ++l) {
        if (lv[l].L > 128) continue;
        for (int up = 0; up < 2; ++up) {
            if (up && l + 1 >= n_levels) continue;
            const Level& src = up ? lv[l + 1] : lv[l];
            const std::vector<int> g3 = make_gather3(lv[l], src, up != 0);
            const std::vector<int> g1 = up ? make_gather_up1(lv[l], src) : std::vector<int>();
            for (int ntaps : {9, 1})
                for (int nrg = 1; nrg <= 2; ++nrg) {
                    DeepArgs a{};
                    a.ntaps = ntaps; a.r = lv[l].r; a.t = lv[l].t; a.up_main = up; a.B = 1; a.Lout = lv[l].L; a.Lsrc = src.L;
                    a.N = 64; a.Cmain = 64; a.Cskip = 32; a.KS = 1; a.CSm = 64; a.CSs = 32; a.nrg = nrg;
                    DeepTile t{};
                    if (!deep_tile_for(a, &t)) return l + 1;
                    const std::vector<int> tab = deep_rowtab(a, t);
                    const int ROWS = 16 * t.RT, SM = a.CSm + 8, SS = a.CSs + 8;
                    int srows = 0;
                    for (int rg = 0; rg < nrg; ++rg) {
                        const int ssz = nrg == 2 ? (rg ? src.L - src.b1 : src.b1) : src.L;
                        srows = ssz > srows ? ssz : srows;
                    }
                    for (int rg = 0; rg < nrg; ++rg) {
                        const int tok0 = (nrg == 2 && rg) ? lv[l].b1 : 0, ntok = nrg == 2 ? (rg ? lv[l].L - lv[l].b1 : lv[l].b1) : lv[l].L;
                        const int stok0 = (nrg == 2 && rg) ? src.b1 : 0;
                        const int* tb = tab.data() + (size_t)rg * (ntaps + 1) * ROWS;
                        for (int tap = 0; tap < ntaps; ++tap)
                            for (int ri = 0; ri < ROWS; ++ri) {
                                int want = -1;
                                if (ri < ntok) want = ntaps == 9 ? g3[(size_t)tap * lv[l].L + tok0 + ri] : (up ? g1[tok0 + ri] : tok0 + ri);
                                const int exp_off = want < 0 ? srows * SM : (want - stok0) * SM;
                                if (tb[tap * ROWS + ri] != exp_off) return l + 1;
                                if (want >= 0 && (want - stok0 < 0 || want - stok0 >= srows)) return l + 1;     // a tap never leaves its row group
                            }
                        for (int ri = 0; ri < ROWS; ++ri)
                            if (tb[ntaps * ROWS + ri] != (ri < ntok ? ri * SS : ROWS * SS)) return l + 1;
                    }
                }
        }
        // ResBlock(down=True) folded into the conv (DeepArgs::pool_main, round 5): the source lives on level l - 1, the taps read the POOLED
        // rows of this level's own grid, parked behind the source slice (lds_pool = (source rows of the largest group + 1) SM)
        if (l > 0) {
            const Level& src = lv[l - 1];
            const std::vector<int> g3 = make_gather3(lv[l], lv[l], false);
            for (int nrg = 1; nrg <= 2; ++nrg) {
                DeepArgs a{};
                a.ntaps = 9; a.r = lv[l].r; a.t = lv[l].t; a.pool_main = 1; a.B = 1; a.Lout = lv[l].L; a.Lsrc = src.L;
                a.N = 64; a.Cmain = 64; a.KS = 1; a.CSm = 64; a.nrg = nrg;
                if (src.r != 2 * lv[l].r || src.t != 2 * lv[l].t) continue;      // (odd geometries never take the folded path: plan.hip)
                DeepTile t{};
                if (!deep_tile_for(a, &t)) return l + 1;
                const std::vector<int> tab = deep_rowtab(a, t);
                const int ROWS = 16 * t.RT, SM = a.CSm + 8;
                int srows = 0;
                for (int rg = 0; rg < nrg; ++rg) {
                    const int ssz = nrg == 2 ? (rg ? src.L - src.b1 : src.b1) : src.L;
                    srows = ssz > srows ? ssz : srows;
                }
                const int lds_pool = (srows + 1) * SM;
                for (int rg = 0; rg < nrg; ++rg) {
                    const int tok0 = (nrg == 2 && rg) ? lv[l].b1 : 0, ntok = nrg == 2 ? (rg ? lv[l].L - lv[l].b1 : lv[l].b1) : lv[l].L;
                    const int* tb = tab.data() + (size_t)rg * 10 * ROWS;
                    for (int tap = 0; tap < 9; ++tap)
                        for (int ri = 0; ri < ROWS; ++ri) {
                            int want = -1;
                            if (ri < ntok) want = g3[(size_t)tap * lv[l].L + tok0 + ri];
                            if (want >= 0 && (want - tok0 < 0 || want - tok0 >= ntok)) return l + 1;
                            if (tb[tap * ROWS + ri] != lds_pool + (want < 0 ? ntok : want - tok0) * SM) return l + 1;
                        }
                }
            }
        }
    }
    return 0;
}

int mtv_selftest_block(int tokens, int channels, int heads, int batch) {
    if (tokens <= 0 || channels <= 0 || heads <= 0 || batch <= 0 || channels % 32) return fail(MTV_ERR_INVALID, "selftest_block: bad arguments");
    for (int ks : {1, 2, 4, 8}) {
        DeepBlockArgs a{};
        a.B = batch; a.L = tokens; a.C = channels; a.H = heads; a.gs = channels / 32; a.x.ks = ks; a.r = 1; a.t = 1;
        if (!deep_block_configure(a, 0, 0)) return 1;
        const int d = channels / heads, NQ = 3 * d, LP = (tokens + 15) / 16 * 16;
        if (!deep_block_launchable(a)) return 2;                                 // exactly what launch_deep_block checks
        if (a.CS % a.gs) return 3;
        if (a.ncols * a.ncp != channels || a.ncols > 256 || (a.ncols & 15) || a.nqt * 16 != LP) return 4;
        // stage 2: every row pair below L dealt to exactly one workgroup
        std::vector<int> seen(LP / 2, 0);
        const int pp = a.rows_per / 2;
        for (int j = 0; j < a.CL; ++j)
            for (int prl = 0; prl < pp; ++prl) {
                const int pr = j * pp + prl;
                if (2 * pr >= tokens) continue;
                if (pr >= LP / 2) return 5;
                ++seen[pr];
            }
        for (int pr = 0; 2 * pr < tokens; ++pr)
            if (seen[pr] != 1) return 5;
        // stage 3: every (query tile, column part) exactly once
        std::vector<int> item(a.nqt * a.ncp, 0);
        for (int j = 0; j < a.CL; ++j)
            for (int it = j; it < a.nqt * a.ncp; it += a.CL) ++item[(it % a.nqt) * a.ncp + it / a.nqt];
        for (int v : item)
            if (v != 1) return 6;
        // scratch: the largest offset each stage forms lies inside the allocation (bytes)
        const size_t part_b = deep_block_part_floats(a) * 4, qkv_b = deep_block_qkv_floats(a) * 4, stg_b = deep_block_stg_floats(a) * 4;
        const size_t bh = (size_t)batch * heads - 1;
        const size_t pmax = (((bh * a.KSN + (a.KSN - 1)) * (LP / 2) + (LP / 2 - 1)) * NQ + (NQ - 1)) * 16 + 16;
        const size_t qmax = ((bh * tokens + (tokens - 1)) * NQ + (NQ - 1)) * 8 + 8;
        const size_t smax = (((bh * a.KSN + (a.KSN - 1)) * a.RQ + (a.RQ - 1)) * 192 + 191) * 16 + 16;
        if (pmax > part_b || qmax > qkv_b || (a.RQ > 1 && smax > stg_b) || part_b >= 0x7FFFFFFFull || qkv_b >= 0x7FFFFFFFull) return 7;
    }
    return 0;
}

int mtv_debug_gather_index(int res, int frames, int tok, int ky, int kx, int up) {
    if (res <= 0 || frames <= 0 || tok < 0 || tok >= res * res + 2 * frames * res || ky < 0 || ky > 2 || kx < 0 || kx > 2)
        return fail(MTV_ERR_INVALID, "debug_gather_index: bad arguments") - 1;   // (-2: distinct from "padding")
    return geo_source(res, frames, tok, ky, kx, up != 0);
}

int mtv_debug_deep(int mode) {
    if (mode < -1 || mode > 1) return fail(MTV_ERR_INVALID, "mode must be -1 (default), 0 (off) or 1 (on)");
    g_deep_mode = mode;
    return MTV_OK;
}

int mtv_debug_deep_options(int mask) {
    if (mask < -1 || mask > 127) return fail(MTV_ERR_INVALID, "mask must be -1 (environment / defaults) or a combination of MTV_DEEP_OPT_*");
    g_deep_opts = mask;
    return MTV_OK;
}

int mtv_debug_attention_qb(int mode) {
    if (mode < -1 || mode > 1) return fail(MTV_ERR_INVALID, "mode must be -1 (default), 0 (off) or 1 (on)");
    g_attn_qb_force = mode;
    return MTV_OK;
}

int mtv_debug_force_b3(int mt, int nt, int ks) {
    if (mt == 0) { g_force_b3[0] = 0; return MTV_OK; }
    if (!(ks == 1 || ks == 2 || ks == 4 || ks == 8)) return fail(MTV_ERR_INVALID, "K slices must be 1, 2, 4 or 8");
    if (!x3_tile_exists(mt, nt)) return fail(MTV_ERR_INVALID, "no k_conv_x3<MT, NT> of that shape (4,2 8,2 4,4 2,2 4,1 2,1 8,1)");
    g_force_b3[0] = mt; g_force_b3[1] = nt; g_force_b3[2] = ks;
    return MTV_OK;
}

int mtv_debug_force_lin(int mt, int nt, int nwv) {
    if (mt == 0) { g_force_lin[0] = 0; return MTV_OK; }
    if (!((mt == 1 || mt == 2) && (nt == 1 || nt == 2 || nt == 4) && (nwv == 1 || nwv == 2 || nwv == 4))) return fail(MTV_ERR_INVALID, "k_lin tile must be {1,2} x {1,2,4} x {1,2,4}");
    g_force_lin[0] = mt; g_force_lin[1] = nt; g_force_lin[2] = nwv;
    return MTV_OK;
}

int mtv_debug_force_pw(int mt, int ntw) { return mtv_debug_force_pw_waves(mt, ntw, 1); }

int mtv_debug_force_pw_waves(int mt, int ntw, int waves) {
    if (mt == 0) { g_force_pw[0] = 0; return MTV_OK; }
    if (waves == 8) waves = 1;
    if (!((mt == 1 || mt == 2) && (ntw == 1 || ntw == 2)) || !(waves == 1 || (ntw == 1 && (waves == 2 || waves == 4 || waves == 6))))
        return fail(MTV_ERR_INVALID, "k_conv_pw tile must be {1, 2} x {1, 2} with 8 multiplying waves, or {1, 2} x 1 with 6 / 4 / 2");
    g_force_pw[0] = mt; g_force_pw[1] = ntw; g_force_pw[2] = waves;
    return MTV_OK;
}

int mtv_debug_force_win_ks(int mt, int nt, int ks) {
    if (mt == 0) { g_force_win[0] = 0; return MTV_OK; }
    if (!((mt == 1 || mt == 2) && (nt == 2 || nt == 4))) return fail(MTV_ERR_INVALID, "k_conv_win tile must be 1x2, 1x4, 2x2 or 2x4");
    if (ks != 1 && ks != 2 && ks != 4) return fail(MTV_ERR_INVALID, "k_conv_win runs 1, 2 or 4 K slices");
    g_force_win[0] = mt; g_force_win[1] = nt; g_force_win[2] = ks;
    return MTV_OK;
}
int mtv_debug_force_win(int mt, int nt) { return mtv_debug_force_win_ks(mt, nt, 1); }

int mtv_debug_force_lds(int wm, int wn) {
    if (wm == 0) { g_force_wm = 0; return MTV_OK; }
    if (!((wm == 2 || wm == 4) && (wn == 2 || wn == 4 || wn == 8))) return fail(MTV_ERR_INVALID, "wave tile must be 2 or 4 by 2, 4 or 8");
    g_force_wm = wm;
    g_force_wn = wn;
    return MTV_OK;
}

int mtv_set_eager(mtv_ctx* c, int eager) {
    if (!c) return fail(MTV_ERR_INVALID, "null context");
    c->eager = eager != 0;
    return MTV_OK;
}

}  // extern "C"
