// pdengine: network construction, weight loading and the forward pass (host orchestration of the
// gfx950 kernels).  Mirrors the topology the reference builds in
//   ldm/modules/diffusionmodules/openaimodel.py:442-736 (UNetModel.__init__) and
//   cldm/cldm.py:48-297 (ControlNet.__init__),
// and the control flow of cldm/cldm.py:23-45 / :302-325 / :369-382.
#include "engine.h"

#include <climits>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>

static thread_local char g_err[1024] = "";
void pd_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
const char* pd_err_buf() { return g_err; }

static inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

// ------------------------------------------------------------------------------------ construction
void* pd_engine::dmalloc(size_t bytes) {
    void* p = nullptr;
    if (bytes == 0) bytes = 16;
    if (hipMalloc(&p, bytes) != hipSuccess) { alloc_failed = true; return nullptr; }
    // the engine stream is non-blocking w.r.t. the null stream: finish the clear before anyone uses p
    hipMemset(p, 0, bytes);
    hipDeviceSynchronize();
    owned.push_back(p);
    weight_bytes += bytes;
    return p;
}

void pd_engine::make_mat(WMat& m, int n_rows, int k_in, int taps, int cin, bool bias, bool geglu) {
    const int EB = (int)dt_size(T);
    const int BKE = 128 / EB;
    m.taps = taps;
    m.cin = cin;
    m.cin_pad = round_up(cin, 8);
    m.K = taps * m.cin_pad;
    m.Kpad = round_up(m.K, BKE);
    m.geglu = geglu;
    m.fan_in = k_in;
    if (geglu) {
        const int half = n_rows / 2;
        m.N = (half + 79) / 80 * 160;
        m.Nout = half;
    } else {
        m.N = round_up(n_rows, 4);
        m.Nout = n_rows;
    }
    m.w = dmalloc((size_t)m.N * m.Kpad * EB);
    m.bias = bias ? reinterpret_cast<float*>(dmalloc((size_t)(m.N + 4) * sizeof(float))) : nullptr;
}

void pd_engine::reg_mat(const std::string& name, std::vector<int64_t> shape, WMat* m, int row_off, bool conv) {
    Param p;
    p.name = name;
    p.shape = std::move(shape);
    p.kind = 1;
    p.mat = m;
    p.row_off = row_off;
    p.conv = conv;
    p.init = 'w';
    p.group = reg_group;
    index[name] = (int)params.size();
    params.push_back(std::move(p));
}

void pd_engine::reg_vec(const std::string& name, int n, float** dst, char init) {
    *dst = reinterpret_cast<float*>(dmalloc((size_t)(n + 4) * sizeof(float)));
    Param p;
    p.name = name;
    p.shape = {n};
    p.kind = 0;
    p.vdst = *dst;
    p.init = init;
    p.group = reg_group;
    index[name] = (int)params.size();
    params.push_back(std::move(p));
}

void pd_engine::reg_bias(const std::string& name, WMat* m, int off, int n, bool geglu) {
    Param p;
    p.name = name;
    p.shape = {n};
    p.kind = 0;
    p.vdst = m->bias + off;
    p.geglu_vec = geglu;
    p.geglu_half = geglu ? n / 2 : 0;
    p.init = 'b';
    p.group = reg_group;
    index[name] = (int)params.size();
    params.push_back(std::move(p));
}

void pd_engine::build_conv(const std::string& prefix, ConvW& c, int cin, int cout, int k, int stride) {
    c.cin = cin;
    c.cout = cout;
    c.k = k;
    c.stride = stride;
    make_mat(c.m, cout, cin * k * k, k * k, cin, true);
    reg_mat(prefix + "weight", {cout, cin, k, k}, &c.m, 0, true);
    reg_bias(prefix + "bias", &c.m, 0, cout);
}

void pd_engine::build_res(const std::string& prefix, ResW& r, int cin, int cout, NetW& net) {
    const int temb = cfg.model_channels * 4;
    r.cin = cin;
    r.cout = cout;
    reg_vec(prefix + "in_layers.0.weight", cin, &r.gn1_g, 'g');
    reg_vec(prefix + "in_layers.0.bias", cin, &r.gn1_b, 'e');
    build_conv(prefix + "in_layers.2.", r.conv1, cin, cout, 3, 1);
    make_mat(r.emb, cout, temb, 1, temb, true);
    reg_mat(prefix + "emb_layers.1.weight", {cout, temb}, &r.emb, 0, false);
    reg_bias(prefix + "emb_layers.1.bias", &r.emb, 0, cout);
    reg_vec(prefix + "out_layers.0.weight", cout, &r.gn2_g, 'g');
    reg_vec(prefix + "out_layers.0.bias", cout, &r.gn2_b, 'e');
    build_conv(prefix + "out_layers.3.", r.conv2, cout, cout, 3, 1);
    r.has_skip = cin != cout;
    if (r.has_skip) build_conv(prefix + "skip_connection.", r.skip, cin, cout, 1, 1);
    r.emb_slot = net.n_emb++;
    net.res_list.push_back(&r);
}

void pd_engine::build_st(const std::string& prefix, STW& s, int ch, NetW& net) {
    const int ctx = cfg.context_dim;
    s.C = ch;
    reg_vec(prefix + "norm.weight", ch, &s.gn_g, 'g');
    reg_vec(prefix + "norm.bias", ch, &s.gn_b, 'e');
    build_conv(prefix + "proj_in.", s.proj_in, ch, ch, 1, 1);
    const std::string t = prefix + "transformer_blocks.0.";
    make_mat(s.qkv, 3 * ch, ch, 1, ch, false);
    reg_mat(t + "attn1.to_q.weight", {ch, ch}, &s.qkv, 0, false);
    reg_mat(t + "attn1.to_k.weight", {ch, ch}, &s.qkv, ch, false);
    reg_mat(t + "attn1.to_v.weight", {ch, ch}, &s.qkv, 2 * ch, false);
    make_mat(s.out1, ch, ch, 1, ch, true);
    reg_mat(t + "attn1.to_out.0.weight", {ch, ch}, &s.out1, 0, false);
    reg_bias(t + "attn1.to_out.0.bias", &s.out1, 0, ch);
    make_mat(s.ff1, 8 * ch, ch, 1, ch, true, true);
    reg_mat(t + "ff.net.0.proj.weight", {8 * ch, ch}, &s.ff1, 0, false);
    reg_bias(t + "ff.net.0.proj.bias", &s.ff1, 0, 8 * ch, true);
    make_mat(s.ff2, ch, 4 * ch, 1, 4 * ch, true);
    reg_mat(t + "ff.net.2.weight", {ch, 4 * ch}, &s.ff2, 0, false);
    reg_bias(t + "ff.net.2.bias", &s.ff2, 0, ch);
    make_mat(s.q2, ch, ch, 1, ch, false);
    reg_mat(t + "attn2.to_q.weight", {ch, ch}, &s.q2, 0, false);
    make_mat(s.kv2, 2 * ch, ctx, 1, ctx, false);
    reg_mat(t + "attn2.to_k.weight", {ch, ctx}, &s.kv2, 0, false);
    reg_mat(t + "attn2.to_v.weight", {ch, ctx}, &s.kv2, ch, false);
    make_mat(s.out2, ch, ch, 1, ch, true);
    reg_mat(t + "attn2.to_out.0.weight", {ch, ch}, &s.out2, 0, false);
    reg_bias(t + "attn2.to_out.0.bias", &s.out2, 0, ch);
    for (int i = 0; i < 3; ++i) {
        const std::string n = t + "norm" + std::to_string(i + 1);
        reg_vec(n + ".weight", ch, &s.ln_g[i], 'g');
        reg_vec(n + ".bias", ch, &s.ln_b[i], 'e');
    }
    build_conv(prefix + "proj_out.", s.proj_out, ch, ch, 1, 1);
    s.kv_slot = net.n_kv++;
    net.st_list.push_back(&s);
}

static bool has_attn(const pd_config& c, int ds) {
    for (int i = 0; i < c.num_attn_res; ++i)
        if (c.attention_resolutions[i] == ds) return true;
    return false;
}

void pd_engine::build_encoder(const std::string& prefix, NetW& net) {
    const int mc = cfg.model_channels, temb = mc * 4;
    make_mat(net.te0, temb, mc, 1, mc, true);
    reg_mat(prefix + "time_embed.0.weight", {temb, mc}, &net.te0, 0, false);
    reg_bias(prefix + "time_embed.0.bias", &net.te0, 0, temb);
    make_mat(net.te2, temb, temb, 1, temb, true);
    reg_mat(prefix + "time_embed.2.weight", {temb, temb}, &net.te2, 0, false);
    reg_bias(prefix + "time_embed.2.bias", &net.te2, 0, temb);
    const int nblocks = 1 + cfg.num_levels * cfg.num_res_blocks + (cfg.num_levels - 1);
    net.enc.resize(nblocks);  // never resized again: ResW/STW addresses are recorded
    int bi = 0;
    {
        EncBlock& b = net.enc[bi];
        b.kind = 0;
        b.cout = mc;
        b.ds = 1;
        build_conv(prefix + "input_blocks.0.0.", b.conv, cfg.in_channels, mc, 3, 1);
        ++bi;
    }
    int ch = mc, ds = 1;
    for (int level = 0; level < cfg.num_levels; ++level) {
        const int mult = cfg.channel_mult[level];
        for (int nr = 0; nr < cfg.num_res_blocks; ++nr) {
            EncBlock& b = net.enc[bi];
            const std::string p = prefix + "input_blocks." + std::to_string(bi) + ".";
            b.kind = 1;
            b.cout = mult * mc;
            b.ds = ds;
            build_res(p + "0.", b.res, ch, mult * mc, net);
            ch = mult * mc;
            b.attn = has_attn(cfg, ds);
            if (b.attn) build_st(p + "1.", b.st, ch, net);
            ++bi;
        }
        if (level != cfg.num_levels - 1) {
            EncBlock& b = net.enc[bi];
            b.kind = 2;
            b.cout = ch;
            b.ds = ds;
            build_conv(prefix + "input_blocks." + std::to_string(bi) + ".0.op.", b.conv, ch, ch, 3, 2);
            ds *= 2;
            ++bi;
        }
    }
}

void pd_engine::build_middle(const std::string& prefix, NetW& net) {
    const int ch = cfg.model_channels * cfg.channel_mult[cfg.num_levels - 1];
    build_res(prefix + "middle_block.0.", net.mid0, ch, ch, net);
    build_st(prefix + "middle_block.1.", net.mid1, ch, net);
    build_res(prefix + "middle_block.2.", net.mid2, ch, ch, net);
}

int pd_engine::build() {
    if (cfg.precision < PD_PREC_BF16 || cfg.precision > PD_PREC_F16X2) {
        pd_set_error("unknown precision %d (PD_PREC_BF16 / PD_PREC_F32 / PD_PREC_F16 / PD_PREC_F16X2)", cfg.precision);
        return 1;
    }
    f32 = cfg.precision == PD_PREC_F32 || cfg.precision == PD_PREC_F16X2;
    T = f32 ? DT_F32 : cfg.precision == PD_PREC_F16 ? DT_F16 : DT_BF16;
    P = cfg.precision == PD_PREC_F16X2 ? PREC_F16X2 : T;
    S = (f32 || cfg.stream_f32) ? DT_F32 : T;
    if (cfg.model_channels % 32 || cfg.num_levels < 1 || cfg.num_levels > PD_MAX_LEVELS || cfg.num_heads < 1) {
        pd_set_error("unsupported config: model_channels must be a multiple of 32, 1..%d levels", PD_MAX_LEVELS);
        return 1;
    }
    const int mc = cfg.model_channels;
    // ---- UNet (model.diffusion_model.*)
    {
        const std::string P = "model.diffusion_model.";
        NetW& n = unet;
        build_encoder(P, n);
        build_middle(P, n);
        std::vector<int> chans;
        for (auto& b : n.enc) chans.push_back(b.cout);
        int ch = chans.back();
        int ds = 1 << (cfg.num_levels - 1);
        n.dec.resize(cfg.num_levels * (cfg.num_res_blocks + 1));
        int di = 0;
        for (int level = cfg.num_levels - 1; level >= 0; --level) {
            const int mult = cfg.channel_mult[level];
            for (int i = 0; i <= cfg.num_res_blocks; ++i) {
                DecBlock& b = n.dec[di];
                const std::string p = P + "output_blocks." + std::to_string(di) + ".";
                const int ich = chans.back();
                chans.pop_back();
                b.skip_c = ich;
                b.cout = mc * mult;
                build_res(p + "0.", b.res, ch + ich, mc * mult, n);
                ch = mc * mult;
                int j = 1;
                b.attn = has_attn(cfg, ds);
                if (b.attn) {
                    build_st(p + "1.", b.st, ch, n);
                    j = 2;
                }
                if (level && i == cfg.num_res_blocks) {
                    b.up = true;
                    build_conv(p + std::to_string(j) + ".conv.", b.upconv, ch, ch, 3, 1);
                    ds /= 2;
                }
                ++di;
            }
        }
        reg_vec(P + "out.0.weight", mc, &n.out_g, 'g');
        reg_vec(P + "out.0.bias", mc, &n.out_b, 'e');
        build_conv(P + "out.2.", n.outconv, mc, cfg.out_channels, 3, 1);
    }
    // ---- ControlNet (control_model.*)
    {
        const std::string P = "control_model.";
        NetW& n = cnet;
        build_encoder(P, n);
        n.zero.resize(n.enc.size());
        for (size_t i = 0; i < n.enc.size(); ++i)
            build_conv(P + "zero_convs." + std::to_string(i) + ".0.", n.zero[i], n.enc[i].cout, n.enc[i].cout, 1, 1);
        const int* w = cfg.hint_widths;
        for (int which = 0; which < 2; ++which) {
            const int cin0 = which == 0 ? cfg.hint_channels : cfg.query_channels;
            const char* nm = which == 0 ? "input_hint_block." : "input_cond_block.";
            std::vector<ConvW>& chain = which == 0 ? n.hint_pair : n.hint_query;
            const int ci[8] = {cin0, w[0], w[1], w[2], w[3], w[4], w[5], w[6]};
            const int co[8] = {w[0], w[1], w[2], w[3], w[4], w[5], w[6], mc};
            const int st[8] = {1, 1, 2, 1, 2, 1, 2, 1};
            chain.resize(8);
            for (int l = 0; l < 8; ++l) build_conv(P + nm + std::to_string(2 * l) + ".", chain[l], ci[l], co[l], 3, st[l]);
        }
        build_middle(P, n);
        const int ch = mc * cfg.channel_mult[cfg.num_levels - 1];
        build_conv(P + "middle_block_out.0.", n.mid_out, ch, ch, 1, 1);
    }
    build_vae();
    build_text();
    if ((int)cnet.enc.size() + 1 != PD_NUM_CONTROL && verbose)
        fprintf(stderr, "[pdengine] note: %zu control tensors (reference SD1.5 has 13)\n", cnet.enc.size() + 1);
    if ((int)cnet.enc.size() + 1 > PD_NUM_CONTROL) {
        pd_set_error("config yields %zu control tensors; at most %d supported", cnet.enc.size() + 1, PD_NUM_CONTROL);
        return 1;
    }
    {
        // initial workspace (per-op hooks, split-K slabs before the first session); sessions grow it
        void* p = nullptr;
        if (hipMalloc(&p, 256u << 20) == hipSuccess) { arena.base = reinterpret_cast<char*>(p); arena.cap = 256u << 20; }
    }
    gn_partial_cap = 8u << 20;
    gn_partial = reinterpret_cast<double*>(dmalloc(gn_partial_cap));
    gn_partial2 = reinterpret_cast<double*>(dmalloc(gn_partial_cap));
    tile_cnt = reinterpret_cast<int*>(dmalloc(kTileCnt * sizeof(int)));    // dmalloc zero-fills
    tile_cnt2 = reinterpret_cast<int*>(dmalloc(kTileCnt * sizeof(int)));
    if (hipStreamCreateWithFlags(&stream2, hipStreamNonBlocking) != hipSuccess) stream2 = nullptr;
    if (hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ev_join, hipEventDisableTiming) != hipSuccess) {
        pd_set_error("hipEventCreate failed");
        return 1;
    }
    if (alloc_failed) {
        pd_set_error("hipMalloc failed while building the engine (%zu bytes allocated so far)", weight_bytes);
        return 1;
    }
    return 0;
}

int pd_engine::check_arena() {
    if (!arena.overflow) return 0;
    pd_set_error("workspace overflow: %zu bytes needed, %zu available (internal sizing error; nothing was launched past it)",
                 arena.peak, arena.cap);
    return 1;
}

// ------------------------------------------------------------------------------------ weights
static inline uint16_t host_f2bf(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);  // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
// fp32 -> fp16, round to nearest even (overflow -> inf, subnormals kept), NaN stays NaN
static inline uint16_t host_f2h(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    const uint16_t sign = (uint16_t)((u >> 16) & 0x8000u);
    u &= 0x7fffffffu;
    if (u > 0x7f800000u) return (uint16_t)(sign | 0x7e00u);
    if (u >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);          // >= 65520 rounds to inf
    if (u < 0x38800000u) {                                            // subnormal half (or zero)
        if (u < 0x33000000u) return sign;                             // < 2^-25 rounds to zero
        const int shift = 113 - (int)(u >> 23);                       // 1..24
        uint32_t man = (u & 0x7fffffu) | 0x800000u;
        const uint32_t half = 1u << (shift + 12), rest = man & ((half << 1) - 1);
        man >>= shift + 13;
        if (rest > half || (rest == half && (man & 1u))) ++man;
        return (uint16_t)(sign | man);
    }
    uint32_t r = u - 0x38000000u;                                     // rebias exponent
    const uint32_t rest = r & 0x1fffu;
    r >>= 13;
    if (rest > 0x1000u || (rest == 0x1000u && (r & 1u))) ++r;
    return (uint16_t)(sign | r);
}
static inline float host_h2f(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000) << 16;
    uint32_t exp = (h >> 10) & 0x1f, man = h & 0x3ff, u;
    if (exp == 0) {
        if (man == 0) u = sign;
        else {
            exp = 127 - 15 + 1;
            while (!(man & 0x400)) { man <<= 1; --exp; }
            man &= 0x3ff;
            u = sign | (exp << 23) | (man << 13);
        }
    } else if (exp == 31) u = sign | 0x7f800000u | (man << 13);
    else u = sign | ((exp - 15 + 127) << 23) | (man << 13);
    float f;
    memcpy(&f, &u, 4);
    return f;
}

int pd_engine::load(const char* name, const void* data, const int64_t* shape, int ndim, int dtype) {
    auto it = index.find(name);
    if (it == index.end()) {
        pd_set_error("pd_load_weights: unknown tensor '%s'", name);
        return 2;
    }
    Param& p = params[it->second];
    if ((int)p.shape.size() != ndim) {
        pd_set_error("pd_load_weights: '%s' expects %zu dims, got %d", name, p.shape.size(), ndim);
        return 3;
    }
    size_t n = 1;
    for (int i = 0; i < ndim; ++i) {
        if (p.shape[i] != shape[i]) {
            pd_set_error("pd_load_weights: '%s' shape mismatch at dim %d: expected %lld, got %lld", name, i,
                         (long long)p.shape[i], (long long)shape[i]);
            return 3;
        }
        n *= (size_t)shape[i];
    }
    std::vector<float> src(n);
    if (dtype == PD_DT_F32) memcpy(src.data(), data, n * 4);
    else if (dtype == PD_DT_F16) {
        const uint16_t* h = reinterpret_cast<const uint16_t*>(data);
        for (size_t i = 0; i < n; ++i) src[i] = host_h2f(h[i]);
    } else if (dtype == PD_DT_BF16) {
        const uint16_t* h = reinterpret_cast<const uint16_t*>(data);
        for (size_t i = 0; i < n; ++i) {
            uint32_t u = (uint32_t)h[i] << 16;
            memcpy(&src[i], &u, 4);
        }
    } else {
        pd_set_error("pd_load_weights: unknown dtype %d", dtype);
        return 4;
    }
    HIP_OK(hipSetDevice(device));
    if (p.kind == 0) {
        PD_TRY(upload_vec(p.vdst, src.data(), (int)n, p.geglu_vec, p.geglu_half));
        p.loaded = true;
        ln_dirty = true;
        return 0;
    }
    WMat& m = *p.mat;
    const int rows = (int)shape[0];
    const int cin = p.conv ? (int)shape[1] : (int)(n / rows);
    const int kk = p.conv ? (int)(shape[2] * shape[3]) : 1;
    if (cin != m.cin || kk != m.taps) {
        pd_set_error("pd_load_weights: '%s' inner shape does not match the layer (cin %d vs %d, taps %d vs %d)", name, cin,
                     m.cin, kk, m.taps);
        return 3;
    }
    PD_TRY(upload_rows(m, p.row_off, src.data(), rows, p.conv));
    p.loaded = true;
    ln_dirty = true;
    sd3_fp8_dirty = true;
    return 0;
}

int pd_engine::upload_vec(float* dst_dev, const float* src, int n, bool geglu, int half) {
    if (geglu) {
        // permute like the weight rows: 80 x-columns then 80 gate-columns per 160-row block
        const int nb = (half + 79) / 80;
        std::vector<float> dst((size_t)nb * 160, 0.f);
        for (int j = 0; j < half; ++j) {
            dst[(size_t)(j / 80) * 160 + j % 80] = src[j];
            dst[(size_t)(j / 80) * 160 + 80 + j % 80] = src[half + j];
        }
        HIP_OK(hipMemcpy(dst_dev, dst.data(), dst.size() * 4, hipMemcpyHostToDevice));
    } else {
        HIP_OK(hipMemcpy(dst_dev, src, (size_t)n * 4, hipMemcpyHostToDevice));
    }
    return 0;
}

// src: [rows][cin] (linear) or [rows][cin][taps] (conv, OIHW) fp32 -> device rows [taps][cin_pad] in the compute type
int pd_engine::upload_rows(WMat& m, int row_off, const float* src, int rows, bool conv) {
    const int EB = (int)dt_size(T);
    const int cin = m.cin, kk = conv ? m.taps : 1;
    std::vector<char> img((size_t)rows * m.Kpad * EB, 0);
    for (int r = 0; r < rows; ++r) {
        char* drow = img.data() + (size_t)r * m.Kpad * EB;
        for (int tp = 0; tp < kk; ++tp)
            for (int c = 0; c < cin; ++c) {
                const float v = conv ? src[((size_t)r * cin + c) * kk + tp] : src[(size_t)r * cin + c];
                const size_t k = (size_t)tp * m.cin_pad + c;
                if (EB == 4) reinterpret_cast<float*>(drow)[k] = v;
                else reinterpret_cast<uint16_t*>(drow)[k] = T == DT_F16 ? host_f2h(v) : host_f2bf(v);
            }
    }
    const size_t rowb = (size_t)m.Kpad * EB;
    if (m.geglu) {
        // destination rows are interleaved in 80 + 80 blocks (x | gate), see gemm.hip
        const int half = m.Nout;
        std::vector<char> perm((size_t)m.N * rowb, 0);
        for (int r = 0; r < rows; ++r) {
            const int j = r < half ? r : r - half;
            const size_t drow = (size_t)(j / 80) * 160 + (r < half ? 0 : 80) + j % 80;
            memcpy(perm.data() + drow * rowb, img.data() + (size_t)r * rowb, rowb);
        }
        HIP_OK(hipMemcpy(m.w, perm.data(), perm.size(), hipMemcpyHostToDevice));
    } else {
        HIP_OK(hipMemcpy(reinterpret_cast<char*>(m.w) + (size_t)row_off * rowb, img.data(), (size_t)rows * rowb,
                         hipMemcpyHostToDevice));
    }
    return 0;
}

int pd_engine::init_random(uint64_t seed) {
    HIP_OK(hipSetDevice(device));
    uint64_t k = 0;
    for (Param& p : params) {
        ++k;
        const uint64_t sd = seed * 0x9E3779B97F4A7C15ull + k;
        if (p.kind == 0) {
            long long n = 1;
            for (int64_t d : p.shape) n *= d;
            if (p.geglu_vec) n = ((p.geglu_half + 79) / 80) * 160;
            float scale = 0.02f, shift = 0.f;
            if (p.init == 'g') { scale = 0.1f; shift = 1.f; }
            if (p.init == 'e') scale = 0.1f;
            if (launch_fill_random(p.vdst, DT_F32, n, scale, shift, sd, stream)) return 1;
        } else {
            WMat& m = *p.mat;
            // fill the whole (padded) row block: pad columns only ever multiply zero-padded inputs
            const long long rows = m.geglu ? m.N : p.shape[0];
            const float scale = 1.0f / sqrtf((float)m.fan_in);
            char* dst = reinterpret_cast<char*>(m.w) + (size_t)(m.geglu ? 0 : p.row_off) * m.Kpad * dt_size(T);
            if (launch_fill_random(dst, T, rows * m.Kpad, scale, 0.f, sd, stream)) return 1;
        }
        p.loaded = true;
    }
    HIP_OK(hipStreamSynchronize(stream));
    ln_dirty = true;
    sd3_fp8_dirty = true;
    return 0;
}

// LayerNorm folded into its consumer (attention.py:271-275: x + attn1(norm1(x)), + attn2(norm2(x)), + ff(norm3(x))):
// LN(x) . W^T = rstd * (x . (W diag(gamma))^T - mean * colsum) + beta . W^T, so the consumer GEMM reads the residual stream
// itself and the normalised tensor is never written.  Built once per weight change.
bool pd_engine::st_tail_on(const STW& s, int rows_per_sample) const {
    return opt_st_fuse && S == T && st_tail_eligible(P, s.C, cfg.num_heads, rows_per_sample, cfg.context_len);
}

int pd_engine::fold_layernorms() {
    const bool on = opt_ln_fuse < 0 ? !f32 : opt_ln_fuse != 0;
    const bool tail = opt_st_fuse && S == T && (P == DT_F16 || P == DT_BF16);
    if ((!on && !tail) || !ln_dirty) return 0;
    auto fold = [&](WMat& m, const float* g, const float* b) -> int {
        if (!m.w_ln) {
            m.w_ln = dmalloc((size_t)m.N * m.Kpad * dt_size(T));
            m.colsum = reinterpret_cast<float*>(dmalloc((size_t)(m.N + 4) * sizeof(float)));
            m.bias_ln = reinterpret_cast<float*>(dmalloc((size_t)(m.N + 4) * sizeof(float)));
            if (!m.w_ln || !m.colsum || !m.bias_ln) { pd_set_error("allocation of folded LayerNorm weights failed"); return 1; }
        }
        if (launch_ln_fold(m.w, m.w_ln, T, m.N, m.K, m.Kpad, g, b, m.bias, m.colsum, m.bias_ln, stream)) {
            pd_set_error("LayerNorm fold launch failed");
            return 1;
        }
        return 0;
    };
    for (int which = 0; which < 2; ++which) {
        NetW& net = which ? cnet : unet;
        for (STW* st : net.st_list) {
            // the shape gate of the fused tail that does not depend on the call (320 channels, 8 heads of 40, <= 96 context keys)
            const bool fused = tail && st_tail_eligible(P, st->C, cfg.num_heads, 128, cfg.context_len);
            if (on || fused) PD_TRY(fold(st->qkv, st->ln_g[0], st->ln_b[0]));   // norm1 -> to_q/k/v
            if (on || fused) PD_TRY(fold(st->q2, st->ln_g[1], st->ln_b[1])); // norm2 -> attn2.to_q
            if (!fused) continue;   // (norm3 stays a kernel in front of the GEGLU tile of gemm.hip)
            PD_TRY(fold(st->ff1, st->ln_g[2], st->ln_b[2]));                 // norm3 -> ff.net.0 (fused tail only)
            if (!st->tail_w) {
                st->tail_w = dmalloc(st_tail_weight_bytes());
                st->tail_vec = reinterpret_cast<float*>(dmalloc(st_tail_vec_floats() * sizeof(float)));
                st->front_w = dmalloc(st_front_weight_bytes());
                st->front_vec = reinterpret_cast<float*>(dmalloc(st_front_vec_floats() * sizeof(float)));
                if (!st->tail_w || !st->tail_vec || !st->front_w || !st->front_vec) { pd_set_error("allocation of the fused transformer weights failed"); return 1; }
            }
            if (launch_st_front_pack(st->proj_in.m.w, st->qkv.w_ln, st->proj_in.m.Kpad, st->proj_in.m.bias, st->qkv.bias_ln, st->front_w, st->front_vec,
                                     stream)) {
                pd_set_error("fused transformer-front weight packing failed");
                return 1;
            }
            if (launch_st_tail_pack(st->out1.w, st->q2.w_ln, st->out2.w, st->ff1.w_ln, st->ff2.w, st->proj_out.m.w, st->out1.Kpad, st->ff2.Kpad,
                                    st->tail_w, stream) ||
                launch_st_tail_vec(st->out1.bias, st->q2.bias_ln, st->out2.bias, st->ff1.bias_ln, st->ff2.bias, st->proj_out.m.bias, st->tail_vec,
                                   stream)) {
                pd_set_error("fused transformer-tail weight packing failed");
                return 1;
            }
        }
    }
    HIP_OK(hipStreamSynchronize(stream));
    ln_dirty = false;
    return 0;
}

// ------------------------------------------------------------------------------------ profiling
hipEvent_t pd_engine::next_event() {
    if (ev_used == ev_pool.size()) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        ev_pool.push_back(e);
    }
    return ev_pool[ev_used++];
}
void pd_engine::prof_begin(ProfRec& r, int klass, double flops) {
    r.klass = klass;
    r.flops = flops;
    r.a = next_event();
    if (r.a) (void)hipEventRecord(r.a, stream);
}
void pd_engine::prof_end(ProfRec& r) {
    r.b = next_event();
    if (r.b) (void)hipEventRecord(r.b, stream);
    prof.push_back(r);
}

// ------------------------------------------------------------------------------------ primitive ops
Act pd_engine::new_act(int B, int H, int W, int C, int dt) {
    Act a;
    a.B = B; a.H = H; a.W = W; a.C = C; a.dt = dt;
    a.p = arena.alloc(a.bytes());
    return a;
}

int pd_engine::gemm(const WMat& m, const Act& in, Act& out, int stride, int ups, int act, float scale, const Act* R,
                    const float* rowvec, int rowvec_stride, bool a_silu, void* VT, int vt_begin, int vt_ld, int ldc_override,
                    const float* gn_coef, bool gn_silu, const LnStats* ln_in, LnStats* ln_out) {
    if (in.C != m.cin_pad) {
        pd_set_error("gemm: input has %d channels, layer expects %d", in.C, m.cin_pad);
        return 1;
    }
    GemmParams p{};
    p.A = in.p; p.W = m.w; p.bias = m.bias; p.C = out.p;
    p.R = R ? R->p : nullptr;
    p.rowvec = rowvec;
    p.VT = VT;
    p.M = (int)out.rows(); p.N = m.N; p.K = m.K; p.Kpad = m.Kpad;
    p.lda = in.C;
    p.ldc = ldc_override ? ldc_override : out.C;
    p.ldr = R ? R->C : 0;
    p.a_dt = in.dt; p.c_dt = out.dt; p.r_dt = R ? R->dt : DT_F32;
    p.taps = m.taps; p.Cin = m.cin_pad;
    p.Hin = in.H; p.Win = in.W; p.Hout = out.H; p.Wout = out.W; p.stride = stride; p.ups = ups;
    p.rows_per_sample = out.H * out.W;
    p.rowvec_stride = rowvec_stride;
    p.act = m.geglu ? 2 : act;
    p.a_silu = a_silu ? 1 : 0;
    p.out_scale = scale;
    p.vt_begin = VT ? vt_begin : INT_MAX;
    p.vt_ld = vt_ld;
    p.Nout = m.Nout;
    p.splitk = 1;
    p.slab = nullptr;
    p.gn_coef = gn_coef;
    p.gn_silu = gn_silu ? 1 : 0;
    p.gate = gx.gate; p.gate_stride = gx.gate_stride;
    p.c_sample_rows = gx.c_sample_rows; p.c_row_off = gx.c_row_off; p.vt_tok_off = gx.vt_tok_off;
    p.a_sample_rows = gx.a_sample_rows; p.a_row_off = gx.a_row_off;
    p.c_scale = gx.c_scale;
    if (out.dt == DT_FP8 && !gx.c_scale) { pd_set_error("internal: fp8 output without row scales"); return 1; }
    const bool fp8 = in.dt == DT_FP8;
    if (fp8) {   // e4m3 operands with per-row scales: the layer's quantised copy
        if (!m.w8 || !gx.a_scale || m.taps != 1 || m.geglu) { pd_set_error("internal: fp8 GEMM without quantised weights / row scales"); return 1; }
        p.W = m.w8; p.Kpad = m.Kpad8; p.w_scale = m.wscale; p.a_scale = gx.a_scale;
    }
    const bool plain = !gx.gate && !gx.c_sample_rows && !gx.a_sample_rows && !gx.c_scale;
    gx = GemmExtra{};
    SlabDefer* defer = gx_defer;
    gx_defer = nullptr;
    if (defer) defer->active = false;
    // a handful of fp32 rows against a wide weight matrix: stream the weights once (gemm.hip's tiles would spend a 128-row
    // tile on <= 4 rows and run at a third of the HBM rate)
    if (opt_gemv && plain && p.M <= 4 && m.taps == 1 && in.dt == DT_F32 && out.dt == DT_F32 && !R && !rowvec && !VT && act == 0 &&
        !m.geglu && scale == 1.f && !ln_in && !ln_out && m.K % 8 == 0 && m.K <= 2048 && m.N >= 4096) {
        if (arena.dry) return 0;
        PD_TRY(check_arena());
        ++launches;
        ProfRec rec{};
        if (profiling) {
            prof_begin(rec, 1, 2.0 * (double)p.M * (double)m.Nout * (double)m.cin);
            rec.M = p.M; rec.N = p.N; rec.K = p.K; rec.taps = 10;
        }
        const int r = launch_gemv(reinterpret_cast<const float*>(in.p), in.C, m.w, T, m.Kpad, m.bias, reinterpret_cast<float*>(out.p), p.ldc,
                                  p.M, m.N, m.K, a_silu ? 1 : 0, stream);
        if (profiling) prof_end(rec);
        if (r) { pd_set_error("gemv launch failed"); return 1; }
        return 0;
    }
    if (ln_in) {   // LayerNorm of `in` folded into this layer: raw A, folded weights, statistics from the producer
        if (!m.w_ln) { pd_set_error("internal: folded LayerNorm weights missing"); return 1; }
        p.W = m.w_ln;
        p.bias = m.bias_ln;
        p.ln_stats = ln_in->stats;
        p.ln_parts = ln_in->parts;
        p.ln_colsum = m.colsum;
        p.ln_C = ln_in->C;
        p.ln_eps = 1e-5f;
    }
    // conv3x3 with enough 16x16 patches to fill the chip: LDS-patch kernel (conv_patch.hip)
    const int ptiles = opt_patch ? conv_patch_tiles(p, P) : 0;
    bool use_patch = ptiles >= ncu * 3 / 4;   // (192 of 256 CUs: a launch that fills three quarters of the chip takes the patch kernel unsplit)
    // 16x16-level convs: too few 16x16 patches for the chip, but the patch kernel still beats the generic gather when the
    // channel chunks are split across 2-4 slices (fp32 slabs + the same deterministic finalize pass as the GEMM's split-K)
    int patch_split = 1;
    if (!use_patch && opt_patch_split && ptiles >= opt_patch_split_tiles && !gn_coef && m.N % 4 == 0) {
        const int chunks = p.Cin / (f32 ? 32 : 64);
        int sk = (opt_patch_split_fill + ptiles - 1) / ptiles;
        if (sk > chunks / opt_patch_split_min) sk = chunks / opt_patch_split_min;   // at least this many channel chunks per slice
        if (sk > 4) sk = 4;
        while (sk >= 2 && (sk - 1) * ((chunks + sk - 1) / sk) >= chunks) --sk;       // no empty slice
        if (sk >= 2) { use_patch = true; patch_split = sk; }
    }
    if (gn_coef && !use_patch) {
        pd_set_error("internal: fused GroupNorm requested for a conv that is not patch-eligible");
        return 1;
    }
    // otherwise split K when the tile grid cannot fill the chip (8x8 / 16x16 levels, time-embedding GEMMs)
    bool use_ring = false;
    int ring_tile = 0;
    if (!use_patch) {
        const size_t mk = arena.mark();
        const int tiles = gemm_tiles(p.M, m.N);
        const int ktiles = fp8 ? m.Kpad8 / 128 : m.Kpad / (128 / (int)dt_size(T));
        int splitk = 1;
        // small-M linear layers (the 8x8 level's M = 1024: 64 tiles of 128 x 160): 64 x 80 ring tiles fill the chip without split-K slabs and
        // a finalize pass (every CU then streams 1 / 16 of the weight matrix once instead of 1 / 8 of a K slice + the slab traffic)
        const bool small_ring = opt_ring > 0 && opt_ring_small && m.taps == 1 && !fp8 && !m.geglu && in.dt == T && ktiles <= opt_ring && tiles * opt_ring_small <= ncu &&
                                ((p.M + 63) / 64) * ((m.N + 79) / 80) >= ncu / 2 && ring_gemm_eligible(p, P);
        // linear layers with a short K and at least half a chip of tiles: one 8-wave block per CU instead of split-K
        const bool dense8 = opt_dense_k > 0 && m.taps == 1 && in.dt == T && !fp8 && !m.geglu && ktiles <= opt_dense_k && tiles >= opt_dense_tiles;
        const int kmin = fp8 ? 8 : 16, kper = fp8 ? 4 : 8;   // an e4m3 K step carries twice the K of a 2-byte one
        if (!small_ring && !dense8 && !m.geglu && !VT && tiles < opt_splitk_tiles && tiles <= kTileCnt && ktiles >= kmin && m.N % 4 == 0) {
            splitk = (2 * ncu + tiles - 1) / tiles;   // about two blocks per CU
            if (splitk > ktiles / kper) splitk = ktiles / kper;
            if (splitk > opt_splitk_max) splitk = opt_splitk_max;
            if (splitk < 1) splitk = 1;
        }
        if (splitk > 1) {
            p.slab = arena.alloc((size_t)splitk * p.M * m.N * sizeof(float));
            if (!arena.dry && arena.top > arena.cap) { splitk = 1; p.slab = nullptr; arena.overflow = false; }  // no room (op hooks outside a session): unsplit
        }
        p.splitk = splitk;
        p.tile_cnt = (splitk > 1 && opt_splitk_fused) ? tile_cnt : nullptr;
        // split-K conv3x3 with at least 1024 rows (the 8x8 level at batch 8): 256-row tiles halve the weight bytes each slice streams
        // (every M tile reads the whole [160 x K / splitk] weight panel) when 256-row tiles x slices still give every CU a block
        const bool sk256 = opt_splitk_big && splitk > 1 && !p.tile_cnt && m.taps == 9 && !f32 && p.M >= 1024 &&
                           ((p.M + 255) / 256) * ((m.N + 159) / 160) * splitk >= ncu;
        // the consumer sums the slabs itself (SlabDefer): plain bias / time-embedding epilogue only
        if (defer && defer->allow && splitk > 1 && !p.tile_cnt && p.slab && plain && act == 0 && scale == 1.f && !R && !VT && !ln_in && !ln_out && !m.geglu) {
            p.defer_finalize = 1;
            defer->active = true;
        }
        // 256-row tiles when they still give every CU a block (1 block of 8 waves per CU)
        p.big_tile = ((opt_bigtile && splitk == 1 && ((p.M + 255) / 256) * ((m.N + 159) / 160) >= ncu) || sk256) ? 1 : 0;
        if (dense8 && tiles < opt_splitk_tiles) p.big_tile = 2;
        // 256 x 320 tiles for linear layers that still give (almost) every CU a block
        {
            const int t3 = ((p.M + 255) / 256) * ((m.N + 319) / 320);
            const int rounds = (t3 + ncu - 1) / ncu;
            if (opt_wide && splitk == 1 && m.taps == 1 && in.dt == T && t3 >= ncu && t3 * 100 >= rounds * ncu * 85) p.big_tile = 3;
        }
        // short-K linear layers are HBM-bound (K <= 1280: 1.5-2.5x their traffic floor): what they need is loads and
        // stores of one tile overlapping the MFMAs of others, i.e. many waves per CU rather than a big tile -- the
        // 128 x 160 tile on 8 waves at <= 128 VGPRs runs 2 blocks = 16 waves per CU (+0.6 % end-to-end, interleaved A/B)
        if (opt_short_k > 0 && splitk == 1 && m.taps == 1 && in.dt == T && !m.geglu && ktiles <= opt_short_k) p.big_tile = 2;
        // widths that are multiples of 192 but not of 160 (MMDiT hidden size 1536 and its 3x / 4x): the 256 x 192 tile
        if (opt_tile192 && !f32 && P != PREC_F16X2 && splitk == 1 && m.taps == 1 && (in.dt == T || fp8) && !m.geglu && m.N % 192 == 0 && m.N % 160 != 0 &&
            ((p.M + 255) / 256) * (m.N / 192) >= ncu * 3 / 4)
            p.big_tile = 4;
        if (P == PREC_F16X2 && m.geglu && p.big_tile == 3) p.big_tile = 1;   // the 256 x 320 GEGLU tile spills with the split-operand fragments
        if (!p.defer_finalize) arena.release(mk);  // stream-ordered: the slab is dead once this GEMM's finalize pass has run
        // short reductions over 2-byte operands: the persistent ring kernel (gemm_ring.hip); 256-row tiles when they give
        // (almost) every CU one
        if (opt_ring > 0 && splitk == 1 && !fp8 && ktiles <= opt_ring && (p.act != 2 || opt_ring_geglu) && ring_gemm_eligible(p, P)) {
            use_ring = true;
            ring_tile = opt_ring_tile >= 0 ? opt_ring_tile : small_ring ? 4 : (((p.M + 255) / 256) * ((m.N + 159) / 160) >= ncu * 7 / 8 ? 1 : 0);
            // ping-pong form (two wave groups half a K step apart; bit-identical): -10..-14 % on long reductions and -3..-5 % on one
            // 256-row tile per CU; +6..+13 % where a block walks several short tiles (the groups' epilogues serialise) -- tools/micro/ring_pp.hip
            if (opt_ring_pp && ring_tile < 2 && p.act != 2 && !small_ring) {
                const int bm = ring_tile ? 256 : 128;
                const int nblk = ((p.M + bm - 1) / bm) * ((m.N + 159) / 160);
                if (ktiles >= 40 || (ring_tile == 1 && nblk <= ncu)) ring_tile += 2;
            }
            p.big_tile = 0;
        }
        if (ln_out && splitk == 1 && !m.geglu && !VT) {   // this launch's own epilogue leaves the row statistics
            // the tile launch_prec (gemm.hip) takes for this launch: 256 x 320 only for linear layers over operands of the compute
            // type, otherwise big_tile 3 falls back to 256 x 160; launch_one rejects a part count that disagrees with its tile
            const bool wide = p.big_tile == 3 && m.taps == 1 && in.dt == T;
            const int bn_cols = wide ? 320 : p.big_tile == 4 ? 192 : 160;
            ln_out->parts = ((m.N + bn_cols - 1) / bn_cols) * 2;   // 2 waves across N in every non-GEGLU tile
            if (use_ring && ring_tile == 4) ln_out->parts = (m.N + 79) / 80;   // ... but one in the 64 x 80 ring tile
            if (ln_out->cap_parts && ln_out->parts > ln_out->cap_parts) {
                pd_set_error("internal: %d LayerNorm statistics partials per row, buffer holds %d", ln_out->parts, ln_out->cap_parts);
                return 1;
            }
            ln_out->C = m.Nout;
            p.stats_out = ln_out->stats;
            p.stats_parts = ln_out->parts;
        }
    }
    if (use_patch && patch_split > 1) {
        const size_t mk = arena.mark();
        p.slab = arena.alloc((size_t)patch_split * p.M * m.N * sizeof(float));
        p.splitk = patch_split;
        if (!arena.dry && arena.top > arena.cap) { pd_set_error("no workspace for the conv split-K slabs"); return 1; }
        if (defer && defer->allow && plain && act == 0 && scale == 1.f && !R && !VT && !ln_in && !ln_out) {
            p.defer_finalize = 1;
            defer->active = true;
        } else {
            arena.release(mk);   // stream-ordered: dead once the finalize pass has run
        }
    }
    if (arena.dry) return 0;
    PD_TRY(check_arena());
    if (f32 && in.dt != DT_F32) {
        pd_set_error("gemm: fp32 mode needs fp32 activations");
        return 1;
    }
    const int prec = fp8 ? PREC_FP8 : P;
    ++launches;
    ProfRec rec{};
    if (profiling) {
        // algorithmic flops: 2 * M * N * K with the logical (unpadded) channel counts
        const double n_log = m.geglu ? 2.0 * m.Nout : (double)m.Nout;
        prof_begin(rec, m.taps == 9 ? 0 : 1, 2.0 * (double)p.M * n_log * (double)m.taps * (double)m.cin);
        rec.M = p.M; rec.N = p.N; rec.K = p.K; rec.taps = p.taps * 10 + p.stride + (p.ups ? 5 : 0);
    }
    hipEvent_t mid = nullptr;
    if (profiling && !use_patch) { mid = next_event(); rec.klass = m.taps == 9 ? 0 : 1; }
    if (profiling && use_patch) rec.klass = 3;
    // second-generation (wave-specialised) patch kernel where it measures faster: many blocks per CU (its longer prologue
    // amortises) or the split-K 16x16 level; the 2-round 64x64 launches stay on the first generation (152 vs 142 us)
    const bool patch2 = use_patch && opt_patch2 && !f32 && !gn_coef && (patch_split > 1 || ptiles >= opt_patch2_tiles);
    // fourth generation (4 waves per block, one per SIMD, 32x32x16 MFMAs, LDS-DMA operands): every unsplit 2-byte launch (-4..-8 % at
    // batch 8, -17..-19 % at batch 1 against the faster of the first two; split-K launches tie and stay on the second)
    const bool patch4 = use_patch && opt_patch4 && patch_split == 1 && conv_patch4_eligible(p, P);
    if (verbose >= 2)   // dispatch trace (option "verbose" 2): which kernel family / tile / split a launch takes
        fprintf(stderr, "[pdengine] gemm M %d N %d K %d taps %d stride %d ups %d: %s splitk %d big_tile %d ring_tile %d act %d R %d ln_in %d ln_out %d\n", p.M, p.N, p.K, p.taps,
                p.stride, p.ups, use_patch ? (patch4 ? "patch4" : patch2 ? "patch2" : "patch1") : use_ring ? "ring" : "igemm", p.splitk, p.big_tile, ring_tile, p.act,
                p.R ? 1 : 0, p.ln_stats ? 1 : 0, p.stats_out ? 1 : 0);
    if (use_patch ? (patch4 ? launch_conv_patch4(p, P, stream) : patch2 ? launch_conv_patch2(p, P, stream) : launch_conv_patch(p, P, stream))
                  : use_ring ? launch_ring_gemm(p, prec, ring_tile, stream) : launch_gemm(p, prec, stream, mid)) {
        pd_set_error("gemm launch failed: %s", hipGetErrorString(hipGetLastError()));
        return 1;
    }
    if (use_ring) ++ring_launches;
    if (defer && defer->active) {
        defer->slabs = reinterpret_cast<const float*>(p.slab); defer->nslab = p.splitk;
        defer->bias = p.bias; defer->rowvec = p.rowvec; defer->rowvec_stride = p.rowvec_stride;
    }
    if (profiling && mid && use_ring) HIP_OK(hipEventRecord(mid, stream));
    if (profiling) {
        if (mid) { rec.b = mid; prof.push_back(rec); }   // bracket = the contraction kernel only (no split-K finalize)
        else prof_end(rec);
    }
    if (ln_out && !p.stats_out) {   // split-K / patch launches finish in another kernel: one statistics pass over the output
        ln_out->parts = 1;
        ln_out->C = m.Nout;
        ++launches;
        if (launch_row_stats(out.p, out.dt, ln_out->stats, (int)out.rows(), out.C, stream)) {
            pd_set_error("row statistics launch failed");
            return 1;
        }
    }
    return 0;
}

int pd_engine::conv(const ConvW& c, const Act& in, Act& out, int act, float scale, const Act* R, const float* rowvec,
                    int rowvec_stride, int ups) {
    return gemm(c.m, in, out, c.stride, ups, act, scale, R, rowvec, rowvec_stride, false, nullptr, 0, 0);
}

int pd_engine::gn_stats(const Act& x, int& nchunk) {
    const int HW = x.H * x.W;
    nchunk = HW / 8;   // enough blocks to fill the chip at the 8x8 / 16x16 levels too
    if (nchunk < 1) nchunk = 1;
    if (nchunk > 64) nchunk = 64;
    while ((size_t)x.B * nchunk * 32 * 2 * sizeof(double) > gn_partial_cap && nchunk > 1) nchunk /= 2;
    if (arena.dry) return 0;
    PD_TRY(check_arena());
    ++launches;
    if (launch_gn_stats(x.p, x.dt, gn_partial, x.B, HW, x.C, 32, nchunk, stream)) {
        pd_set_error("groupnorm stats launch failed (C=%d)", x.C);
        return 1;
    }
    return 0;
}

int pd_engine::groupnorm(const Act& x, Act& y, const float* g, const float* b, float eps, bool silu, const SlabDefer* from_slabs) {
    if (from_slabs && from_slabs->active) {   // x was never written: the producing GEMM's split-K slabs are the input
        if (!(opt_gn_single && gn_fused_bundle(x.dt, x.H * x.W, x.C, 32))) { pd_set_error("internal: deferred split-K slabs without the single-kernel GroupNorm"); return 1; }
        if (arena.dry) return 0;
        PD_TRY(check_arena());
        ++launches;
        ++gn_from_slabs;
        if (launch_gn_fused_slabs(from_slabs->slabs, from_slabs->nslab, from_slabs->bias, from_slabs->rowvec, from_slabs->rowvec_stride, x.dt, y.p, y.dt, g, b,
                                  x.B, x.H * x.W, x.C, 32, eps, silu ? 1 : 0, stream)) {
            pd_set_error("groupnorm (single kernel, split-K slabs) launch failed (C=%d)", x.C);
            return 1;
        }
        return 0;
    }
    if (opt_gn_single && gn_fused_bundle(x.dt, x.H * x.W, x.C, 32)) {   // slab fits in LDS: one kernel, one read
        if (arena.dry) return 0;
        PD_TRY(check_arena());
        ++launches;
        if (launch_gn_fused(x.p, x.dt, y.p, y.dt, g, b, x.B, x.H * x.W, x.C, 32, eps, silu ? 1 : 0, stream)) {
            pd_set_error("groupnorm (single kernel) launch failed (C=%d)", x.C);
            return 1;
        }
        return 0;
    }
    int nchunk = 1;
    PD_TRY(gn_stats(x, nchunk));
    if (arena.dry) return 0;
    ++launches;
    if (launch_gn_apply(x.p, x.dt, y.p, y.dt, gn_partial, g, b, x.B, x.H * x.W, x.C, 32, nchunk, eps, silu ? 1 : 0, stream)) {
        pd_set_error("groupnorm launch failed (C=%d)", x.C);
        return 1;
    }
    return 0;
}

// conv3x3(act(GroupNorm(x))): when the conv runs on the LDS-patch kernel the normalisation (+SiLU) is applied while
// the input patch is staged, so the normalised tensor is never written to HBM; otherwise GroupNorm runs as its own pass.
int pd_engine::conv_gn(const ConvW& c, const Act& x, Act& out, const float* g, const float* b, float eps, bool silu,
                       const Act* R, const float* rowvec, int rowvec_stride, SlabDefer* out_defer, const SlabDefer* in_slabs) {
    GemmParams q{};
    q.M = (int)out.rows(); q.N = c.m.N; q.K = c.m.K; q.taps = c.m.taps; q.Cin = c.m.cin_pad; q.stride = c.stride;
    q.Hin = x.H; q.Win = x.W; q.Hout = out.H; q.Wout = out.W; q.a_dt = x.dt; q.vt_begin = INT_MAX; q.splitk = 1;
    const bool fuse = opt_gn_fuse && opt_patch && x.C == c.m.cin_pad && conv_patch_tiles(q, P) >= ncu * 3 / 4;
    if (!fuse) {
        const size_t mk = arena.mark();
        Act a = new_act(x.B, x.H, x.W, x.C, T);
        PD_TRY(groupnorm(x, a, g, b, eps, silu, in_slabs));
        gx_defer = out_defer;
        PD_TRY(conv(c, a, out, 0, 1.f, R, rowvec, rowvec_stride));
        if (!(out_defer && out_defer->active)) arena.release(mk);   // deferred: the slabs sit above `a`; the caller's mark frees both
        return 0;
    }
    if (in_slabs && in_slabs->active) { pd_set_error("internal: deferred split-K slabs in front of a GroupNorm-fused conv"); return 1; }
    int nchunk = 1;
    PD_TRY(gn_stats(x, nchunk));
    const size_t mk = arena.mark();
    float* coef = reinterpret_cast<float*>(arena.alloc((size_t)x.B * x.C * 2 * sizeof(float)));
    if (!arena.dry) {
        ++launches;
        if (launch_gn_coef(gn_partial, g, b, coef, x.B, x.H * x.W, x.C, 32, nchunk, eps, stream)) {
            pd_set_error("groupnorm coefficient launch failed");
            return 1;
        }
    }
    PD_TRY(gemm(c.m, x, out, c.stride, 0, 0, 1.f, R, rowvec, rowvec_stride, false, nullptr, 0, 0, 0, coef, silu));
    arena.release(mk);
    return 0;
}

int pd_engine::layernorm(const Act& x, Act& y, const float* g, const float* b) {
    if (arena.dry) return 0;
    PD_TRY(check_arena());
    ++launches;
    if (launch_layernorm(x.p, x.dt, y.p, y.dt, g, b, (int)x.rows(), x.C, 1e-5f, stream)) {
        pd_set_error("layernorm launch failed (C=%d)", x.C);
        return 1;
    }
    return 0;
}

int pd_engine::attention(const void* Q, int ldq, const void* K, int ldk, const void* VT, int vt_ld, void* O, int ldo, int B,
                         int Nq, int Nk, int C, int heads, bool causal, long long q_bs, long long k_bs, long long o_bs) {
    if (heads <= 0) heads = cfg.num_heads;
    if (arena.dry) return 0;
    PD_TRY(check_arena());
    AttnParams p{};
    p.Q = Q; p.K = K; p.VT = VT; p.O = O;
    p.ldq = ldq; p.ldk = ldk; p.ldo = ldo; p.vt_ld = vt_ld;
    p.q_bs = q_bs ? q_bs : (long long)Nq * ldq;   // explicit strides: queries / keys that are row ranges of a larger buffer
    p.k_bs = k_bs ? k_bs : (long long)Nk * ldk;
    p.vt_bs = (long long)C * vt_ld;
    p.o_bs = o_bs ? o_bs : (long long)Nq * ldo;
    p.Nq = Nq; p.Nk = Nk; p.heads = heads; p.dh = C / heads;
    p.causal = causal ? 1 : 0;
    p.scale = (float)(1.0 / std::sqrt((double)p.dh));
    p.B = B;
    p.legacy = opt_attn_legacy ? 1 : 0;
    ++launches;
    ProfRec rec{};
    if (profiling) {
        prof_begin(rec, 2, 4.0 * (double)B * heads * (double)Nq * (double)Nk * (double)p.dh);
        rec.M = Nq; rec.N = Nk; rec.K = p.dh; rec.taps = B;
    }
    const int r = launch_attention(p, P, stream);
    if (profiling) prof_end(rec);
    if (r) {
        pd_set_error(r == 2 ? "attention: unsupported head dim %d" : "attention launch failed (dh %d)", p.dh);
        return 1;
    }
    return 0;
}

// ResBlock._forward, openaimodel.py:254-274
int pd_engine::resblock(const ResW& r, const Act& x, Act& out, const float* embrow, int emb_stride) {
    out = new_act(x.B, x.H, x.W, r.cout, S);
    const size_t mk = arena.mark();
    Act h = new_act(x.B, x.H, x.W, r.cout, T);
    // conv1's only reader is norm2: when conv1 runs split-K and norm2 is the single-kernel GroupNorm (16x16 / 8x8 levels), that
    // kernel sums the slabs itself -- no finalize pass, h is never written
    SlabDefer d;
    d.allow = opt_slab_gn && !opt_gn_fuse && opt_gn_single && h.dt == T && gn_fused_bundle(T, x.H * x.W, r.cout, 32) > 0;
    PD_TRY(conv_gn(r.conv1, x, h, r.gn1_g, r.gn1_b, r.eps, true, nullptr, embrow, emb_stride, &d));
    Act skip = x;
    if (r.has_skip) {
        skip = new_act(x.B, x.H, x.W, r.cout, S);
        PD_TRY(conv(r.skip, x, skip));
    }
    PD_TRY(conv_gn(r.conv2, h, out, r.gn2_g, r.gn2_b, r.eps, true, &skip, nullptr, 0, nullptr, &d));
    arena.release(mk);
    return 0;
}

// SpatialTransformer.forward + BasicTransformerBlock._forward, attention.py:321-340, :271-275
//
// out_B > x.B (the shared front of a CFG batch, forward_eps): x holds the x.B samples both halves of the batch have in common -- the two
// halves of cat([x] * 2) (ddim_hacked.py:189) are the same numbers until the first cross-attention reads their different contexts --
// so GroupNorm, proj_in, norm1, attn1 (and, per layer, attn1.to_out, norm2, attn2.to_q) run once on x.B samples and the block's
// output has out_B = 2 x.B: sample j of it continues from shared sample j % x.B with context j.
int pd_engine::transformer(const STW& s, const Act& x, Act& out, const KVSlot& kv, int out_B) {
    int B = x.B;
    const int H = x.H, W = x.W, C = s.C, N = H * W;
    if (out_B <= 0) out_B = B;
    if (out_B % B) { pd_set_error("internal: transformer output batch %d over %d shared samples", out_B, B); return 1; }
    const int reps = out_B / B;
    out = new_act(out_B, H, W, C, S);
    const size_t mk = arena.mark();
    const bool fused = st_tail_on(s, N) && s.tail_w && s.front_w && kv.P;
    const int npad = round_up(N, 8);
    Act h, qk, vt, ln;
    LnStats st0, st1;
    const bool fuse = (opt_ln_fuse < 0 ? !f32 : opt_ln_fuse != 0) && s.qkv.w_ln != nullptr;
    if (fused) {
        // GroupNorm statistics -> per-(sample, channel) {scale, shift}; the apply, proj_in, norm1 and to_q/k/v are one kernel
        int nchunk = 1;
        PD_TRY(gn_stats(x, nchunk));
        float* coef = reinterpret_cast<float*>(arena.alloc((size_t)B * C * 2 * sizeof(float)));
        h = new_act(B, H, W, C, S);
        qk = new_act(B, H, W, 2 * C, T);
        vt = new_act(B, C, 1, npad, T);
        if (!arena.dry) {
            PD_TRY(check_arena());
            launches += 2;
            if (launch_gn_coef(gn_partial, s.gn_g, s.gn_b, coef, B, N, C, 32, nchunk, 1e-6f, stream)) {
                pd_set_error("groupnorm coefficient launch failed");
                return 1;
            }
            ProfRec rec{};
            if (profiling) {
                prof_begin(rec, 4, st_front_flops((long long)B * N));
                rec.M = B * N; rec.N = 4 * C; rec.K = C; rec.taps = 1;
            }
            const int r = launch_st_front(x.p, coef, h.p, qk.p, vt.p, s.front_w, s.front_vec, (long long)B * N, N, npad, P, stream);
            if (profiling) prof_end(rec);
            if (r) { pd_set_error("fused transformer-front launch failed: %s", hipGetErrorString(hipGetLastError())); return 1; }
        }
    } else {
    Act a = new_act(B, H, W, C, T);
    PD_TRY(groupnorm(x, a, s.gn_g, s.gn_b, 1e-6f, false));
    // LayerNorm statistics travel from the epilogue that writes h / h1 / h2 to the GEMM that consumes norm1/2/3 of it
    // (fold_layernorms): one partial per 80-column wave range of the narrowest tile
    if (fuse) {
        st0.cap_parts = st1.cap_parts = ((C + 159) / 160) * 2;
        const size_t cap = (size_t)B * N * (size_t)st0.cap_parts * 2 * sizeof(float);
        st0.stats = reinterpret_cast<float*>(arena.alloc(cap));
        st1.stats = reinterpret_cast<float*>(arena.alloc(cap));
    }
    h = new_act(B, H, W, C, S);
    PD_TRY(gemm(s.proj_in.m, a, h, 1, 0, 0, 1.f, nullptr, nullptr, 0, false, nullptr, 0, 0, 0, nullptr, false, nullptr, fuse ? &st0 : nullptr));
    // self-attention: fused QKV projection; V stored transposed for the attention kernel
    if (!fuse) {
        ln = new_act(B, H, W, C, T);
        PD_TRY(layernorm(h, ln, s.ln_g[0], s.ln_b[0]));
    }
    qk = new_act(B, H, W, 2 * C, T);
    vt = new_act(B, C, 1, npad, T);
    PD_TRY(gemm(s.qkv, fuse ? h : ln, qk, 1, 0, 0, 1.f, nullptr, nullptr, 0, false, vt.p, 2 * C, npad, 0, nullptr, false, fuse ? &st0 : nullptr));
    }
    Act att = new_act(B, H, W, C, T);
    const size_t eb = dt_size(T);
    PD_TRY(attention(qk.p, 2 * C, reinterpret_cast<char*>(qk.p) + (size_t)C * eb, 2 * C, vt.p, npad, att.p, C, B, N, N, C));
    if (fused) {
        // everything after the self-attention product in one kernel (st_tail.hip): no h1 / q2 / h2 / norm3 / GEGLU / h3 round trips
        if (!arena.dry) {
            PD_TRY(check_arena());
            ++launches;
            ProfRec rec{};
            if (profiling) {
                prof_begin(rec, 4, st_tail_flops((long long)out_B * N, cfg.context_len));
                rec.M = out_B * N; rec.N = C; rec.K = C; rec.taps = 0;
            }
            const int r = launch_st_tail(att.p, h.p, x.p, out.p, s.tail_w, s.tail_vec, kv.P, (long long)out_B * N, N, cfg.context_len, S,
                                         (float)(1.0 / std::sqrt((double)(C / cfg.num_heads))), P, stream, (long long)B * N);
            if (profiling) prof_end(rec);
            if (r) { pd_set_error("fused transformer-tail launch failed: %s", hipGetErrorString(hipGetLastError())); return 1; }
        }
        arena.release(mk);
        return 0;
    }
    Act h1 = new_act(B, H, W, C, S);
    PD_TRY(gemm(s.out1, att, h1, 1, 0, 0, 1.f, &h, nullptr, 0, false, nullptr, 0, 0, 0, nullptr, false, nullptr, fuse ? &st1 : nullptr));
    // cross-attention against the hoisted context K / V^T
    if (!fuse) PD_TRY(layernorm(h1, ln, s.ln_g[1], s.ln_b[1]));
    Act q2 = new_act(B, H, W, C, T);
    PD_TRY(gemm(s.q2, fuse ? h1 : ln, q2, 1, 0, 0, 1.f, nullptr, nullptr, 0, false, nullptr, 0, 0, 0, nullptr, false, fuse ? &st1 : nullptr));
    const int L = cfg.context_len, lpad = round_up(L, 8);
    Act xr = x;   // the block's residual at the output batch
    if (reps > 1) {
        // the shared queries meet each half's own context keys: one launch per half; from here on the batch is out_B
        att = new_act(out_B, H, W, C, T);
        for (int r = 0; r < reps; ++r)
            PD_TRY(attention(q2.p, C, reinterpret_cast<const char*>(kv.K) + (size_t)r * B * L * C * eb, C,
                             reinterpret_cast<const char*>(kv.VT) + (size_t)r * B * C * lpad * eb, lpad,
                             reinterpret_cast<char*>(att.p) + (size_t)r * B * N * C * eb, C, B, N, L, C));
        PD_TRY(repeat_act(h1, reps, h1));
        PD_TRY(repeat_act(x, reps, xr));
    } else
    PD_TRY(attention(q2.p, C, kv.K, C, kv.VT, lpad, att.p, C, B, N, L, C));
    B = out_B;
    Act h2 = new_act(B, H, W, C, S);
    PD_TRY(gemm(s.out2, att, h2, 1, 0, 0, 1.f, &h1, nullptr, 0, false, nullptr, 0, 0));
    // GEGLU feed-forward (norm3 as a kernel: see fold_layernorms)
    if (fuse || reps > 1) ln = new_act(B, H, W, C, T);
    PD_TRY(layernorm(h2, ln, s.ln_g[2], s.ln_b[2]));
    Act g = new_act(B, H, W, 4 * C, T);
    PD_TRY(gemm(s.ff1, ln, g, 1, 0, 0, 1.f, nullptr, nullptr, 0, false, nullptr, 0, 0));
    Act h3 = new_act(B, H, W, C, S);
    PD_TRY(gemm(s.ff2, g, h3, 1, 0, 0, 1.f, &h2, nullptr, 0, false, nullptr, 0, 0));
    PD_TRY(conv(s.proj_out, h3, out, 0, 1.f, &xr));
    arena.release(mk);
    return 0;
}

// `reps` copies of a tensor along the batch axis (the per-layer transformer path at the end of a shared CFG front)
int pd_engine::repeat_act(const Act& src, int reps, Act& dst) {
    Act d = new_act(src.B * reps, src.H, src.W, src.C, src.dt);
    if (!arena.dry) {
        PD_TRY(check_arena());
        for (int r = 0; r < reps; ++r)
            HIP_OK(hipMemcpyAsync(reinterpret_cast<char*>(d.p) + (size_t)r * src.bytes(), src.p, src.bytes(), hipMemcpyDeviceToDevice, stream));
    }
    dst = d;
    return 0;
}

// ------------------------------------------------------------------------------------ networks
static const float* emb_ptr(const std::vector<float*>& tabs, const ResW& r, int row) {
    return tabs[r.emb_slot] + (size_t)row * r.cout;
}

// ControlNet.forward, cldm/cldm.py:302-325 (hint embedders hoisted into the session: they do not
// depend on the timestep); outputs go to ses.control[] already multiplied by control_scales (:379).
int pd_engine::run_controlnet(const Act& x_in_full, int emb_row, int emb_stride, const float* scales) {
    NetW& n = cnet;
    Act h;
    // shared CFG front (forward_eps): until the first transformer block the network runs on the samples the two halves have in common
    int Bf = x_in_full.B;
    Act x_in = x_in_full, hint = ses.hint;
    if (ses.share_c) { x_in.B = Bf / 2; hint.B = Bf / 2; }
    // guess mode under guidance ((D) pipeline :1220-1224, :1248-1253): the ControlNet sees the conditional half only and the unconditional
    // half gets zero residuals.  The network runs on views of the second half of the batch (latents, guided_hint, context K / V^T, control
    // tensors); join_controlnet zeroes the first half of every control tensor.
    const int s0 = ses.cn_cond_only ? Bf / 2 : 0;   // first sample of the views
    if (s0) {
        Bf -= s0;
        x_in = batch_view(x_in_full, s0, Bf);
        hint = batch_view(ses.hint, s0, Bf);
    }
    auto kvs = [&](const STW& st) {
        KVSlot k = ses.kv_c[st.kv_slot];
        if (s0) {
            const size_t eb = dt_size(T);
            const int L = cfg.context_len, lpad = round_up(L, 8);
            k.K = reinterpret_cast<char*>(k.K) + (size_t)s0 * L * st.C * eb;
            k.VT = reinterpret_cast<char*>(k.VT) + (size_t)s0 * st.C * lpad * eb;
            if (k.P) k.P = reinterpret_cast<char*>(k.P) + st_tail_kv_bytes(s0);
        }
        return k;
    };
    auto ctl = [&](int i) { return s0 ? batch_view(ses.control[i], s0, Bf) : ses.control[i]; };
    for (size_t i = 0; i < n.enc.size(); ++i) {
        EncBlock& b = n.enc[i];
        Act o;
        if (b.kind == 0) {
            o = new_act(x_in.B, x_in.H, x_in.W, b.cout, S);
            PD_TRY(conv(b.conv, x_in, o, 0, 1.f, &hint));  // h = conv_in(x) + guided_hint, :315-317
        } else if (b.kind == 2) {
            o = new_act(h.B, h.H / 2, h.W / 2, b.cout, S);
            PD_TRY(conv(b.conv, h, o));
        } else {
            PD_TRY(resblock(b.res, h, o, emb_ptr(ses.emb_c, b.res, emb_row), emb_stride ? b.res.cout : 0));
            if (b.attn) {
                Act o2;
                PD_TRY(transformer(b.st, o, o2, kvs(b.st), Bf));
                o = o2;
            }
        }
        h = o;
        Act c = ctl((int)i);
        if (c.B != h.B) { pd_set_error("internal: control tensor %d sized for batch %d, block output has %d", (int)i, c.B, h.B); return 1; }
        PD_TRY(conv(n.zero[i], h, c, 0, scales ? scales[i] : 1.f));
    }
    Act m0, m1, m2;
    PD_TRY(resblock(n.mid0, h, m0, emb_ptr(ses.emb_c, n.mid0, emb_row), emb_stride ? n.mid0.cout : 0));
    PD_TRY(transformer(n.mid1, m0, m1, kvs(n.mid1), Bf));
    PD_TRY(resblock(n.mid2, m1, m2, emb_ptr(ses.emb_c, n.mid2, emb_row), emb_stride ? n.mid2.cout : 0));
    const int last = (int)n.enc.size();
    Act cl = ctl(last);
    PD_TRY(conv(n.mid_out, m2, cl, 0, scales ? scales[last] : 1.f));
    return 0;
}

// samples [first, first + count) of a batched tensor
Act pd_engine::batch_view(const Act& a, int first, int count) {
    Act v = a;
    v.p = reinterpret_cast<char*>(a.p) + (size_t)first * a.H * a.W * a.C * dt_size(a.dt);
    v.B = count;
    return v;
}

// ControlledUnetModel.forward, cldm/cldm.py:23-45
int pd_engine::run_unet(const Act& x_in_full, int emb_row, int emb_stride, bool only_mid, Act& eps) {
    NetW& n = unet;
    std::vector<Act> hs;
    Act h;
    const int Bf = x_in_full.B;
    Act x_in = x_in_full;
    if (ses.share_u) x_in.B = Bf / 2;   // shared CFG front, as in run_controlnet; the skip tensors of the front stay at half batch
    for (size_t i = 0; i < n.enc.size(); ++i) {
        EncBlock& b = n.enc[i];
        Act o;
        if (b.kind == 0) {
            o = new_act(x_in.B, x_in.H, x_in.W, b.cout, S);
            PD_TRY(conv(b.conv, x_in, o));
        } else if (b.kind == 2) {
            o = new_act(h.B, h.H / 2, h.W / 2, b.cout, S);
            PD_TRY(conv(b.conv, h, o));
        } else {
            PD_TRY(resblock(b.res, h, o, emb_ptr(ses.emb_u, b.res, emb_row), emb_stride ? b.res.cout : 0));
            if (b.attn) {
                Act o2;
                PD_TRY(transformer(b.st, o, o2, ses.kv_u[b.st.kv_slot], Bf));
                o = o2;
            }
        }
        h = o;
        hs.push_back(h);
    }
    Act m0, m1, m2;
    PD_TRY(resblock(n.mid0, h, m0, emb_ptr(ses.emb_u, n.mid0, emb_row), emb_stride ? n.mid0.cout : 0));
    PD_TRY(transformer(n.mid1, m0, m1, ses.kv_u[n.mid1.kv_slot], Bf));
    PD_TRY(resblock(n.mid2, m1, m2, emb_ptr(ses.emb_u, n.mid2, emb_row), emb_stride ? n.mid2.cout : 0));
    h = m2;
    PD_TRY(join_controlnet());   // the decoder is the first consumer of the control tensors
    const int nctl = (int)cnet.enc.size();  // index of the middle control tensor
    for (size_t i = 0; i < n.dec.size(); ++i) {
        DecBlock& b = n.dec[i];
        Act skip = hs.back();
        hs.pop_back();
        // h = cat([h (+ mid control, :35), skip + control.pop() (:41)], dim=1)
        Act cat = new_act(h.B, h.H, h.W, h.C + skip.C, S);
        const Act* a_add = i == 0 ? &ses.control[nctl] : nullptr;
        const Act* b_add = only_mid ? nullptr : &ses.control[nctl - 1 - (int)i];
        if (!arena.dry) {
            PD_TRY(check_arena());
            ++launches;
            if (launch_concat_add(h.p, a_add ? a_add->p : nullptr, skip.p, b_add ? b_add->p : nullptr, cat.p, S, h.rows(), h.C,
                                  skip.C, stream, skip.rows(), b_add ? b_add->rows() : 0)) {
                pd_set_error("concat launch failed");
                return 1;
            }
        }
        Act o;
        PD_TRY(resblock(b.res, cat, o, emb_ptr(ses.emb_u, b.res, emb_row), emb_stride ? b.res.cout : 0));
        if (b.attn) {
            Act o2;
            PD_TRY(transformer(b.st, o, o2, ses.kv_u[b.st.kv_slot]));
            o = o2;
        }
        if (b.up) {
            Act o3 = new_act(o.B, o.H * 2, o.W * 2, o.C, S);
            PD_TRY(conv(b.upconv, o, o3, 0, 1.f, nullptr, nullptr, 0, /*ups=*/1));  // Upsample: nearest x2 then conv, :115-117
            o = o3;
        }
        h = o;
    }
    eps = new_act(h.B, h.H, h.W, round_up(cfg.out_channels, 4), DT_F32);
    PD_TRY(conv_gn(n.outconv, h, eps, n.out_g, n.out_b, 1e-5f, true, nullptr, nullptr, 0));
    return 0;
}

void pd_engine::swap_context() {
    std::swap(arena, arena2);
    std::swap(stream, stream2);
    std::swap(gn_partial, gn_partial2);
    std::swap(tile_cnt, tile_cnt2);
}

// Called by the UNet right before its decoder: wait for the ControlNet stream, then apply the guess-mode zeroing.
int pd_engine::join_controlnet() {
    const pd_sample_args& a = ses.a;
    if (arena.dry) return 0;
    if (cn_pending) {
        HIP_OK(hipStreamWaitEvent(stream, ev_join, 0));
        cn_pending = false;
    }
    if (a.guess_mode && a.use_cfg) {
        // (D) pipeline :1248-1253: the unconditional half gets zero residuals
        for (size_t i = 0; i <= cnet.enc.size(); ++i)
            HIP_OK(hipMemsetAsync(ses.control[i].p, 0, ses.control[i].bytes() / 2, stream));
    }
    return 0;
}

// ControlLDM.apply_model, cldm/cldm.py:369-382
int pd_engine::forward_eps(int emb_row, int emb_stride, const float* scales, Act& eps) {
    const pd_sample_args& a = ses.a;
    const int Bf = ses.Bf;
    Act x_in;
    x_in.p = ses.x_in; x_in.B = Bf; x_in.H = a.h; x_in.W = a.w; x_in.C = 8; x_in.dt = DT_F32;
    // Shared CFG front.  p_sample_ddim runs the networks on cat([x] * 2), cat([t] * 2) and [unconditional ; conditional] conditioning
    // (ddim_hacked.py:189-192): sample j and sample B + j enter with the same latent and the same timestep, and, unless the caller
    // gave the unconditional half its own example pair / query, the same guided_hint.  Until the first cross-attention reads the two
    // contexts, both halves therefore carry the same numbers, and the engine computes them once: conv_in, the first ResBlock, and the
    // first SpatialTransformer up to attn2.to_q run on B samples (pd_engine::transformer's out_B), their skip / control tensors stay at
    // B samples and are read twice.  Exact; what differs is only which tile shapes a half-size launch picks.  Needs the time embedding of a
    // step to be one row for the whole batch (emb_stride 0: the sampler's own calls).  Guess mode zeroes the unconditional half of every
    // control tensor (join_controlnet) and keeps the ControlNet unshared.
    const bool share = opt_cfg_share && a.use_cfg && emb_stride == 0 && Bf % 2 == 0;
    ses.share_u = share;
    ses.share_c = share && ses.hint_shared && !a.guess_mode;
    ses.cn_cond_only = opt_cfg_share && a.guess_mode && a.use_cfg && emb_stride == 0 && Bf % 2 == 0;
    // the 13 control tensors outlive the ControlNet pass
    int hh = a.h, ww = a.w;
    bool front = ses.share_c;
    for (size_t i = 0; i < cnet.enc.size(); ++i) {
        if (cnet.enc[i].kind == 2) { hh /= 2; ww /= 2; }
        if (cnet.enc[i].attn) front = false;
        ses.control[i] = new_act(front ? Bf / 2 : Bf, hh, ww, cnet.enc[i].cout, S);
    }
    ses.control[cnet.enc.size()] = new_act(Bf, hh, ww, cnet.enc.back().cout, S);
    // ControlNet and the UNet encoder + middle block are independent: ControlNet is enqueued on a second stream with its
    // own workspace so that its kernels overlap the encoder's (different kernels meet on a CU in different phases: MFMA
    // main loops next to VALU-heavy epilogues / softmax).  The decoder waits for it (join_controlnet).
    const bool two = opt_two_streams && stream2 != nullptr;
    if (two && !arena.dry) {
        HIP_OK(hipEventRecord(ev_fork, stream));
        HIP_OK(hipStreamWaitEvent(stream2, ev_fork, 0));
    }
    if (two) swap_context();
    const size_t mk = arena.mark();
    int rc = run_controlnet(x_in, emb_row, emb_stride, scales);
    arena.release(mk);
    if (two) {
        if (!rc && !arena.dry && hipEventRecord(ev_join, stream) != hipSuccess) rc = 1;   // `stream` is stream2 here
        swap_context();
    }
    if (rc) return rc;
    cn_pending = two;
    PD_TRY(run_unet(x_in, emb_row, emb_stride, a.only_mid_control != 0, eps));
    return 0;
}
