// pdengine: sampling sessions (DDIM loop state, hoisted loop invariants) and the C ABI.
// Follows DDIMSampler.{make_schedule, sample, ddim_sampling, p_sample_ddim}, cldm/ddim_hacked.py:23-234.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>

#include "engine.h"

const char* pd_err_buf();
static inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

// ------------------------------------------------------------------------------------ schedule
// make_beta_schedule('linear') util.py:21-25 + register_schedule ddpm.py:138-159 (alphas_cumprod kept
// as float32) + make_ddim_timesteps util.py:46-60 + make_ddim_sampling_parameters util.py:63-74.
// custom_desc (optional): `steps` timesteps in sampling order (descending), replacing the uniform grid -- the DDIM parameters
// are derived from the list the same way (a_prev of an entry = alphas_cumprod at the next entry, alphas_cumprod[0] after the last)
int pd_engine::make_schedule(int steps, float eta, std::vector<int64_t>& ts, std::vector<float>& al, std::vector<float>& ap,
                             std::vector<float>& sg, std::vector<float>& s1m, const int64_t* custom_desc) {
    const int T_ = cfg.timesteps;
    if (steps < 1 || steps > T_) {
        pd_set_error("steps must be in [1, %d]", T_);
        return 1;
    }
    const double s0 = std::sqrt(cfg.linear_start), s1 = std::sqrt(cfg.linear_end);
    std::vector<float> ac(T_);
    double cp = 1.0;
    const double st = T_ > 1 ? (s1 - s0) / (double)(T_ - 1) : 0.0;
    for (int i = 0; i < T_; ++i) {
        const double b = i == T_ - 1 ? s1 : s0 + st * (double)i;
        cp *= 1.0 - b * b;
        ac[i] = (float)cp;
    }
    const int c = T_ / steps;
    ts.clear();
    if (custom_desc) {
        for (int i = steps - 1; i >= 0; --i) {
            const int64_t t = custom_desc[i];
            // (equal neighbours are legal: make_ddim_timesteps('quad') repeats small timesteps, util.py:50)
            if (t < 0 || t >= T_ || (!ts.empty() && t < ts.back())) {
                pd_set_error("custom timesteps must be descending and inside [0, %d)", T_);
                return 1;
            }
            ts.push_back(t);
        }
    } else {
        for (int t = 0; t < T_; t += c) ts.push_back(t + 1);
    }
    const int n = (int)ts.size();
    if (ts.back() >= T_) {
        pd_set_error("ddim timestep %lld out of range for %d steps (same IndexError as the reference, util.py:65)",
                     (long long)ts.back(), steps);
        return 1;
    }
    al.resize(n); ap.resize(n); sg.resize(n); s1m.resize(n);
    for (int i = 0; i < n; ++i) {
        al[i] = ac[ts[i]];
        ap[i] = i == 0 ? ac[0] : ac[ts[i - 1]];
        const double a = al[i], p = ap[i];
        sg[i] = (float)((double)eta * std::sqrt((1.0 - p) / (1.0 - a) * (1.0 - a / p)));
        s1m[i] = std::sqrt(1.0f - al[i]);
    }
    return 0;
}

// timestep_embedding, util.py:154-174: [cos, sin], freqs = exp(-ln(10000) * i / half) in float32
void pd_host_timestep_embedding(const int64_t* t, int n, int dim, std::vector<float>& out) {
    const int half = dim / 2;
    out.assign((size_t)n * dim, 0.f);
    const float lg = -std::log(10000.0f);
    for (int r = 0; r < n; ++r)
        for (int i = 0; i < half; ++i) {
            const float f = std::exp(lg * (float)i / (float)half);
            const float a = (float)t[r] * f;
            out[(size_t)r * dim + i] = std::cos(a);
            out[(size_t)r * dim + half + i] = std::sin(a);
        }
}

// time_embed MLP + every ResBlock's emb_layers for n timesteps at once (they do not depend on the
// latents, so the sampler hoists them out of the step loop).
int pd_engine::compute_emb(NetW& net, std::vector<float*>& tabs, const int64_t* t, int n, int row0) {
    const int mc = cfg.model_channels, temb = mc * 4;
    const size_t mk = arena.mark();
    Act te = new_act(n, 1, 1, mc, DT_F32);
    if (!arena.dry) {
        std::vector<float> host;
        pd_host_timestep_embedding(t, n, mc, host);
        HIP_OK(hipMemcpyAsync(te.p, host.data(), host.size() * 4, hipMemcpyHostToDevice, stream));
        HIP_OK(hipStreamSynchronize(stream));
    }
    Act e1 = new_act(n, 1, 1, temb, DT_F32), e2 = new_act(n, 1, 1, temb, DT_F32);
    PD_TRY(gemm(net.te0, te, e1, 1, 0, /*silu*/ 1, 1.f, nullptr, nullptr, 0, false, nullptr, 0, 0));
    PD_TRY(gemm(net.te2, e1, e2, 1, 0, 0, 1.f, nullptr, nullptr, 0, false, nullptr, 0, 0));
    for (ResW* r : net.res_list) {
        Act o;
        o.B = n; o.H = 1; o.W = 1; o.C = r->cout; o.dt = DT_F32;
        o.p = tabs[r->emb_slot] + (size_t)row0 * r->cout;
        PD_TRY(gemm(r->emb, e2, o, 1, 0, 0, 1.f, nullptr, nullptr, 0, /*a_silu*/ true, nullptr, 0, 0));
    }
    arena.release(mk);
    return 0;
}

// Allocates the session state from the arena and computes everything that does not change across
// steps: NHWC copies of the inputs, guided_hint (cldm/cldm.py:306-308), context K/V of every
// cross-attention, projected time embeddings of every step.
int pd_engine::session_setup(const pd_sample_args& a, const int64_t* t_rows, int n_rows, bool per_sample_t, bool want_per_step) {
    (void)per_sample_t;
    Session& s = ses;
    const int B = a.batch, Bf = s.Bf, HW = a.h * a.w, C = cfg.in_channels;
    const int L = cfg.context_len, D = cfg.context_dim, Dp = round_up(D, 8), lpad = round_up(L, 8);
    const int IH = a.h * 8, IW = a.w * 8;
    const bool dev = a.mem == PD_MEM_DEVICE;
    arena.top = 0;
    arena.overflow = false;
    s.x_state = reinterpret_cast<float*>(arena.alloc((size_t)B * HW * 8 * 4));
    s.x_in = reinterpret_cast<float*>(arena.alloc((size_t)Bf * HW * 8 * 4));
    s.pred_x0 = reinterpret_cast<float*>(arena.alloc((size_t)B * HW * C * 4));
    s.eps_g = reinterpret_cast<float*>(arena.alloc((size_t)B * HW * C * 4));
    s.per_step = want_per_step ? reinterpret_cast<float*>(arena.alloc((size_t)(s.S + 1) * B * C * HW * 4)) : nullptr;
    s.noise = (a.noise && a.eta > 0.f) ? reinterpret_cast<float*>(arena.alloc((size_t)s.S * B * C * HW * 4)) : nullptr;
    s.ctx = arena.alloc((size_t)Bf * L * Dp * dt_size(T));
    s.hint = new_act(Bf, a.h, a.w, cfg.model_channels, S);
    s.kv_u.resize(unet.n_kv);
    s.kv_c.resize(cnet.n_kv);
    for (int which = 0; which < 2; ++which) {
        NetW& net = which ? cnet : unet;
        std::vector<KVSlot>& kv = which ? s.kv_c : s.kv_u;
        for (STW* st : net.st_list) {
            kv[st->kv_slot].K = arena.alloc((size_t)Bf * L * st->C * dt_size(T));
            kv[st->kv_slot].VT = arena.alloc((size_t)Bf * st->C * lpad * dt_size(T));
            kv[st->kv_slot].P = (st_tail_on(*st, 128) && st->tail_w) ? arena.alloc(st_tail_kv_bytes(Bf)) : nullptr;
        }
    }
    s.emb_rows = n_rows + 1;  // last row: scratch for pd_sample_eps_at
    s.emb_u.assign(unet.n_emb, nullptr);
    s.emb_c.assign(cnet.n_emb, nullptr);
    for (ResW* r : unet.res_list) s.emb_u[r->emb_slot] = reinterpret_cast<float*>(arena.alloc((size_t)s.emb_rows * r->cout * 4));
    for (ResW* r : cnet.res_list) s.emb_c[r->emb_slot] = reinterpret_cast<float*>(arena.alloc((size_t)s.emb_rows * r->cout * 4));
    s.session_top = arena.top;

    // ---- inputs -> device, NHWC, compute type
    const size_t mk = arena.mark();
    auto stage = [&](const float* src, size_t n) -> const float* {  // host pointer -> device temp
        float* d = reinterpret_cast<float*>(arena.alloc(n * 4));
        if (!arena.dry && !dev) {
            if (hipMemcpyAsync(d, src, n * 4, hipMemcpyHostToDevice, stream) != hipSuccess) return nullptr;
            hipStreamSynchronize(stream);
        }
        return dev ? src : d;
    };
    {
        const float* xT = stage(a.x_T, (size_t)B * C * HW);
        if (!xT && !arena.dry) { pd_set_error("copy of x_T failed"); return 1; }
        if (!arena.dry) {
            if (launch_nchw_to_nhwc(xT, s.x_state, DT_F32, B, C, a.h, a.w, 8, stream)) return 1;
            if (launch_fill_x_in(s.x_state, s.x_in, B, Bf / B, C, 8, HW, stream)) return 1;
            launches += 2;
            if (s.per_step) HIP_OK(hipMemcpyAsync(s.per_step, xT, (size_t)B * C * HW * 4, hipMemcpyDeviceToDevice, stream));
        }
        if (s.noise && !arena.dry)
            HIP_OK(hipMemcpyAsync(s.noise, a.noise, (size_t)s.S * B * C * HW * 4, dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, stream));
    }
    {
        // context rows: [uncond ; cond] (ddim_hacked.py:191 puts the unconditional half first)
        const size_t per = (size_t)B * L * D;
        const float* cc = stage(a.ctx_cond, per);
        const float* cu = a.use_cfg ? stage(a.ctx_uncond, per) : nullptr;
        if (!arena.dry) {
            char* dst = reinterpret_cast<char*>(s.ctx);
            const size_t half = (size_t)B * L * Dp * dt_size(T);
            if (a.use_cfg) {
                if (launch_cast_rows(cu, dst, T, (long long)B * L, D, Dp, stream)) return 1;
                if (launch_cast_rows(cc, dst + half, T, (long long)B * L, D, Dp, stream)) return 1;
            } else if (launch_cast_rows(cc, dst, T, (long long)B * L, D, Dp, stream)) return 1;
            launches += 2;
        }
    }
    Act ctx;
    ctx.p = s.ctx; ctx.B = Bf; ctx.H = L; ctx.W = 1; ctx.C = Dp; ctx.dt = T;
    // ---- hoisted context K / V^T of every cross-attention (attention.py:168-169)
    for (int which = 0; which < 2; ++which) {
        NetW& net = which ? cnet : unet;
        std::vector<KVSlot>& kv = which ? s.kv_c : s.kv_u;
        for (STW* st : net.st_list) {
            Act k;
            k.p = kv[st->kv_slot].K; k.B = Bf; k.H = L; k.W = 1; k.C = st->C; k.dt = T;
            PD_TRY(gemm(st->kv2, ctx, k, 1, 0, 0, 1.f, nullptr, nullptr, 0, false, kv[st->kv_slot].VT, st->C, lpad));
            if (kv[st->kv_slot].P && !arena.dry) {   // the same K / V^T in the fused tail's fragment order
                ++launches;
                if (launch_st_tail_kv_pack(k.p, kv[st->kv_slot].VT, kv[st->kv_slot].P, Bf, L, lpad, stream)) {
                    pd_set_error("context K/V packing for the fused transformer tail failed");
                    return 1;
                }
            }
        }
    }
    // ---- guided_hint = input_hint_block(pair) + input_cond_block(query), cldm/cldm.py:306-308
    {
        Act prev_out;
        s.hint_shared = a.use_cfg != 0 && (!a.pair_uncond || a.pair_uncond == a.pair) && (!a.query_uncond || a.query_uncond == a.query);
        // both halves of the CFG batch were given the same example pair and query: the hint embedders run on B samples and guided_hint's
        // second half is a copy of the first (option cfg_share; forward_eps)
        const int hb = (s.hint_shared && opt_cfg_share) ? B : Bf;
        Act hint_out = s.hint;
        hint_out.B = hb;
        for (int which = 0; which < 2; ++which) {
            const int cin = which == 0 ? cfg.hint_channels : cfg.query_channels;
            const float* src_c = which == 0 ? a.pair : a.query;
            const float* src_u = which == 0 ? (a.pair_uncond ? a.pair_uncond : a.pair) : (a.query_uncond ? a.query_uncond : a.query);
            const size_t per = (size_t)B * cin * IH * IW;
            Act img = new_act(hb, IH, IW, 8, T);
            const float* dc = stage(src_c, per);
            const float* du = (a.use_cfg && src_u != src_c) ? stage(src_u, per) : dc;
            if (!arena.dry) {
                char* dst = reinterpret_cast<char*>(img.p);
                const size_t half = (size_t)B * IH * IW * 8 * dt_size(T);
                if (hb != Bf) {
                    if (launch_nchw_to_nhwc(dc, dst, T, B, cin, IH, IW, 8, stream)) return 1;
                } else if (a.use_cfg) {
                    if (launch_nchw_to_nhwc(du, dst, T, B, cin, IH, IW, 8, stream)) return 1;
                    if (launch_nchw_to_nhwc(dc, dst + half, T, B, cin, IH, IW, 8, stream)) return 1;
                } else if (launch_nchw_to_nhwc(dc, dst, T, B, cin, IH, IW, 8, stream)) return 1;
                launches += 2;
            }
            std::vector<ConvW>& chain = which == 0 ? cnet.hint_pair : cnet.hint_query;
            Act cur = img;
            for (int l = 0; l < 8; ++l) {
                const ConvW& c = chain[l];
                const int Ho = c.stride == 2 ? (cur.H + 1) / 2 : cur.H, Wo = c.stride == 2 ? (cur.W + 1) / 2 : cur.W;
                if (l < 7) {
                    Act o = new_act(hb, Ho, Wo, round_up(c.cout, 8), T);
                    if (o.C != c.cout) { pd_set_error("hint widths must be multiples of 8"); return 1; }
                    PD_TRY(conv(c, cur, o, /*silu*/ 1));
                    cur = o;
                } else if (which == 0) {
                    prev_out = new_act(hb, Ho, Wo, c.cout, S);
                    PD_TRY(conv(c, cur, prev_out));
                } else {
                    PD_TRY(conv(c, cur, hint_out, 0, 1.f, &prev_out));
                }
            }
        }
        if (hb != Bf && !arena.dry)
            HIP_OK(hipMemcpyAsync(reinterpret_cast<char*>(s.hint.p) + hint_out.bytes(), s.hint.p, hint_out.bytes(), hipMemcpyDeviceToDevice, stream));
    }
    // ---- projected time embeddings for all rows
    PD_TRY(compute_emb(unet, s.emb_u, t_rows, n_rows, 0));
    PD_TRY(compute_emb(cnet, s.emb_c, t_rows, n_rows, 0));
    arena.release(mk);
    if (arena.top != s.session_top) arena.top = s.session_top;
    return 0;
}

int pd_engine::ensure_arena(int Bf, int h, int w, int rows, bool per_step) {
    (void)Bf; (void)h; (void)w;
    // dry run of setup + one forward to size both workspaces (main and ControlNet context)
    Arena saved = arena, saved2 = arena2;
    arena.base = nullptr; arena.cap = 0; arena.top = 0; arena.peak = 0; arena.dry = true;
    arena2.base = nullptr; arena2.cap = 0; arena2.top = 0; arena2.peak = 0; arena2.dry = true;
    std::vector<int64_t> t(rows, 1);
    int r = session_setup(ses.a, t.data(), rows, false, per_step);
    Act eps;
    if (!r) r = forward_eps(0, 0, nullptr, eps);
    const size_t need = arena.peak + (64u << 20), need2 = arena2.peak + (64u << 20);
    arena = saved;
    arena2 = saved2;
    arena.dry = false;
    arena2.dry = false;
    cn_pending = false;
    if (r) return r;
    auto grow = [&](Arena& a, size_t want, const char* what) -> int {
        if (want <= a.cap) return 0;
        if (a.base) HIP_OK(hipFree(a.base));
        a.base = nullptr;
        a.cap = 0;
        void* p = nullptr;
        if (hipMalloc(&p, want) != hipSuccess) {
            pd_set_error("%s workspace allocation of %.2f GiB failed", what, (double)want / (1 << 30));
            return 1;
        }
        a.base = reinterpret_cast<char*>(p);
        a.cap = want;
        if (verbose) fprintf(stderr, "[pdengine] %s workspace %.2f GiB\n", what, (double)want / (1 << 30));
        return 0;
    };
    if (need > arena.cap || need2 > arena2.cap) {
        clear_graphs();   // captured loops point into the old workspaces
        HIP_OK(hipStreamSynchronize(stream));
        if (stream2) HIP_OK(hipStreamSynchronize(stream2));
        PD_TRY(grow(arena, need, "main"));
        PD_TRY(grow(arena2, need2, "controlnet"));
    }
    arena.top = 0; arena.peak = 0;
    arena2.top = 0; arena2.peak = 0;
    return 0;
}

static int check_args(pd_engine* e, const pd_sample_args* a) {
    if (!a || a->batch < 1 || a->h < 1 || a->w < 1) { pd_set_error("bad sample args (batch/h/w)"); return 1; }
    const int need = 1 << (e->cfg.num_levels - 1);
    if (a->h % need || a->w % need) {
        pd_set_error("latent size %dx%d must be a multiple of %d (the UNet halves the resolution %d times)", a->h, a->w, need,
                     e->cfg.num_levels - 1);
        return 1;
    }
    if (!a->x_T || !a->ctx_cond || !a->pair || !a->query) { pd_set_error("x_T, ctx_cond, pair and query are required"); return 1; }
    if (a->use_cfg && !a->ctx_uncond) { pd_set_error("use_cfg needs ctx_uncond"); return 1; }
    if (a->eta > 0.f && !a->noise) {
        // the reference always draws noise when eta > 0 (ddim_hacked.py:230); running the eta > 0 coefficients without
        // it would deflate the sample variance silently
        pd_set_error("eta > 0 needs the noise draws (pd_sample_args.noise: [steps][B, in_ch, h, w])");
        return 1;
    }
    for (auto& p : e->params)
        if (p.group == 0 && !p.loaded) { pd_set_error("weights not loaded: '%s' (and possibly more)", p.name.c_str()); return 1; }
    return 0;
}

int pd_engine::begin(const pd_sample_args* a, bool want_per_step) {
    PD_TRY(check_args(this, a));
    HIP_OK(hipSetDevice(device));
    PD_TRY(fold_layernorms());
    ses.active = false;
    ses.a = *a;
    ses.Bf = a->use_cfg ? 2 * a->batch : a->batch;
    ses.custom_ts.clear();
    if (a->timesteps) ses.custom_ts.assign(a->timesteps, a->timesteps + a->steps);   // always HOST memory
    ses.a.timesteps = nullptr;
    PD_TRY(make_schedule(a->steps, a->eta, ses.timesteps, ses.alphas, ses.alphas_prev, ses.sigmas, ses.sqrt_1m,
                         ses.custom_ts.empty() ? nullptr : ses.custom_ts.data()));
    ses.S = (int)ses.timesteps.size();
    const int S_ = ses.S;
    const int nc = (int)cnet.enc.size() + 1;
    ses.scales_step.assign((size_t)S_ * PD_NUM_CONTROL, 1.0f);
    for (int i = 0; i < S_; ++i)
        for (int j = 0; j < nc; ++j) {
            float v = a->control_scales ? a->control_scales[j] : 1.0f;
            if (a->control_scales_step) v = a->control_scales_step[(size_t)i * PD_NUM_CONTROL + j];
            ses.scales_step[(size_t)i * PD_NUM_CONTROL + j] = v;
        }
    std::vector<int64_t> trows(S_);
    for (int i = 0; i < S_; ++i) trows[i] = ses.timesteps[S_ - 1 - i];  // sampling order = flipped (ddim_hacked.py:144)
    PD_TRY(ensure_arena(ses.Bf, a->h, a->w, S_, want_per_step));
    PD_TRY(session_setup(*a, trows.data(), S_, false, want_per_step));
    ses.active = true;
    return 0;
}

int pd_engine::step(int i) {
    Session& s = ses;
    if (!s.active) { pd_set_error("no active sampling session"); return 1; }
    if (i < 0 || i >= s.S) { pd_set_error("step %d out of range [0,%d)", i, s.S); return 1; }
    const pd_sample_args& a = s.a;
    const int index = s.S - 1 - i;
    const size_t mk = arena.mark();
    Act eps;
    PD_TRY(forward_eps(i, 0, &s.scales_step[(size_t)i * PD_NUM_CONTROL], eps));
    // p_sample_ddim, ddim_hacked.py:206-233, scalars in float32 as torch.full materialises them
    DdimCoef k;
    const float a_t = s.alphas[index], a_prev = s.alphas_prev[index], sig = s.sigmas[index];
    k.sqrt_one_minus_at = s.sqrt_1m[index];
    k.sqrt_at = std::sqrt(a_t);
    k.sqrt_a_prev = std::sqrt(a_prev);
    k.dir_coef = std::sqrt(1.0f - a_prev - sig * sig);
    k.sigma = sig;
    k.cfg_scale = a.cfg_scale;
    const int C = cfg.in_channels, HW = a.h * a.w;
    const float* nz = s.noise ? s.noise + (size_t)i * a.batch * C * HW : nullptr;
    ++launches;
    if (launch_cfg_ddim(eps.p, eps.dt, eps.C, s.x_state, s.pred_x0, s.eps_g, s.x_in, nz, a.batch, HW, C, 8, a.use_cfg, k,
                        a.temperature, 1, stream)) {
        pd_set_error("ddim update launch failed");
        return 1;
    }
    if (s.per_step) {
        ++launches;
        if (launch_nhwc_to_nchw(s.x_state, DT_F32, s.per_step + (size_t)(i + 1) * a.batch * C * HW, a.batch, C, a.h, a.w, 8, 1.f, stream))
            return 1;
    }
    arena.release(mk);
    return 0;
}

// ------------------------------------------------------------------------------------ captured step loop
// The S steps of a call differ only in kernel ARGUMENTS (time-embedding row, DDIM coefficients, control scales, noise
// slice); every buffer they touch is carved out of the two workspaces in a deterministic order by begin().  So the
// whole loop of a call -- ControlNet on the second stream included, through the fork / join events -- is captured once
// into a hipGraph and replayed by later calls whose arguments hash to the same key.  Pays where the loop is
// launch-bound on the host (small batches); the inputs are staged by begin() outside the graph.
void pd_engine::clear_graphs() {
    for (auto& g : graphs) {
        if (g.exec) hipGraphExecDestroy(g.exec);
        if (g.graph) hipGraphDestroy(g.graph);
    }
    graphs.clear();
}

static inline void hash_mix(uint64_t& h, const void* p, size_t n) {
    const unsigned char* b = reinterpret_cast<const unsigned char*>(p);
    for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; }
}

int pd_engine::run_steps_graph() {
    const pd_sample_args& a = ses.a;
    uint64_t key = 1469598103934665603ull;
    const int32_t ints[] = {a.batch, a.h, a.w, a.steps, a.use_cfg, a.guess_mode, a.only_mid_control, ses.noise ? 1 : 0,
                            ses.per_step ? 1 : 0, opt_two_streams ? 1 : 0, ses.S, opt_cfg_share ? 1 : 0};
    const float flts[] = {a.eta, a.cfg_scale, a.temperature};
    const void* ptrs[] = {arena.base, arena2.base, ses.x_state, ses.per_step};
    hash_mix(key, ints, sizeof(ints));
    hash_mix(key, flts, sizeof(flts));
    hash_mix(key, ptrs, sizeof(ptrs));
    hash_mix(key, ses.scales_step.data(), ses.scales_step.size() * sizeof(float));
    hash_mix(key, ses.timesteps.data(), ses.timesteps.size() * sizeof(int64_t));
    for (auto& g : graphs)
        if (g.key == key) {
            HIP_OK(hipGraphLaunch(g.exec, stream));
            return 0;
        }
    GraphEntry ge{key, nullptr, nullptr};
    if (hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal) != hipSuccess) {
        (void)hipGetLastError();
        for (int i = 0; i < ses.S; ++i) PD_TRY(step(i));   // capture unavailable: run eagerly
        return 0;
    }
    int rc = 0;
    for (int i = 0; i < ses.S && !rc; ++i) rc = step(i);
    const hipError_t ec = hipStreamEndCapture(stream, &ge.graph);
    if (rc || ec != hipSuccess || !ge.graph) {
        if (ge.graph) hipGraphDestroy(ge.graph);
        if (!rc) pd_set_error("hipStreamEndCapture failed: %s", hipGetErrorString(ec));
        return 1;
    }
    if (hipGraphInstantiate(&ge.exec, ge.graph, nullptr, nullptr, 0) != hipSuccess) {
        hipGraphDestroy(ge.graph);
        pd_set_error("hipGraphInstantiate failed: %s", hipGetErrorString(hipGetLastError()));
        return 1;
    }
    if (graphs.size() >= 8) clear_graphs();
    graphs.push_back(ge);
    HIP_OK(hipGraphLaunch(ge.exec, stream));
    return 0;
}

// ------------------------------------------------------------------------------------ C ABI
extern "C" {

const char* pd_last_error(void) { return pd_err_buf(); }
int pd_abi_version(void) { return PD_ABI_VERSION; }

int pd_engine_create(const pd_config* cfg, int device_id, pd_engine** out) {
    if (!cfg || !out) { pd_set_error("null argument"); return 1; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
        pd_set_error("no HIP device visible: the pdengine compute path needs an MI355X (there is no CPU fallback)");
        return 10;
    }
    if (device_id < 0 || device_id >= ndev) { pd_set_error("device %d out of range (%d visible)", device_id, ndev); return 1; }
    if (cfg->text_layers > 0) {
        const int th = cfg->text_heads, dh = th > 0 && cfg->context_dim % th == 0 ? cfg->context_dim / th : 0;
        if (dh != 8 && dh != 16 && dh != 32 && dh != 40 && dh != 64 && dh != 80 && dh != 160) {
            pd_set_error("text transformer: context_dim %d / text_heads %d is not a supported head size", cfg->context_dim, th);
            return 1;
        }
        if (cfg->text_vocab < 1 || cfg->text_ff < 8 || cfg->text_ff % 8) { pd_set_error("text transformer: bad vocab / ff size"); return 1; }
    }
    HIP_OK(hipSetDevice(device_id));
    pd_engine* e = new pd_engine();
    e->cfg = *cfg;
    e->device = device_id;
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device_id) == hipSuccess && prop.multiProcessorCount >= 8) {
            e->ncu = prop.multiProcessorCount;
            e->opt_splitk_tiles = e->opt_splitk_tiles * e->ncu / 256;   // the defaults are written for 256 CUs
        }
    }
    if (hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking) != hipSuccess) {
        pd_set_error("hipStreamCreate failed");
        delete e;
        return 1;
    }
    if (e->build()) { pd_engine_destroy(e); return 1; }
    *out = e;
    return 0;
}

void pd_engine_destroy(pd_engine* e) {
    if (!e) return;
    hipSetDevice(e->device);
    if (e->stream) hipStreamSynchronize(e->stream);
    (void)pd_comm_destroy(e);
    e->clear_graphs();
    if (e->stream2) { hipStreamSynchronize(e->stream2); hipStreamDestroy(e->stream2); }
    for (hipEvent_t ev : e->sd3_ev) hipEventDestroy(ev);
    if (e->ev_fork) hipEventDestroy(e->ev_fork);
    if (e->ev_join) hipEventDestroy(e->ev_join);
    if (e->arena2.base) hipFree(e->arena2.base);
    for (void* p : e->owned)
        if (p) hipFree(p);
    if (e->arena.base) hipFree(e->arena.base);
    if (e->stream) hipStreamDestroy(e->stream);
    delete e;
}

int pd_param_count(pd_engine* e) { return e ? (int)e->params.size() : 0; }

int pd_param_info(pd_engine* e, int i, const char** name, int32_t* ndim, int64_t shape[4]) {
    if (!e || i < 0 || i >= (int)e->params.size()) { pd_set_error("param index out of range"); return 1; }
    const Param& p = e->params[i];
    if (name) *name = p.name.c_str();
    if (ndim) *ndim = (int32_t)p.shape.size();
    if (shape)
        for (size_t k = 0; k < 4; ++k) shape[k] = k < p.shape.size() ? p.shape[k] : 1;
    return 0;
}

int pd_load_weights(pd_engine* e, const char* name, const void* data, const int64_t* shape, int32_t ndim, int32_t dtype) {
    if (!e || !name || !data || !shape) { pd_set_error("null argument"); return 1; }
    return e->load(name, data, shape, ndim, dtype);
}

int pd_init_random_weights(pd_engine* e, uint64_t seed) {
    if (!e) { pd_set_error("null engine"); return 1; }
    return e->init_random(seed);
}

int pd_weights_missing(pd_engine* e) {
    int n = 0;
    if (e)
        for (auto& p : e->params) n += (p.group == 0 && !p.loaded) ? 1 : 0;
    return n;
}

int pd_make_schedule(pd_engine* e, int32_t steps, float eta, int64_t* timesteps, float* alphas, float* alphas_prev,
                     float* sigmas, float* sqrt_one_minus_alphas) {
    if (!e) { pd_set_error("null engine"); return 1; }
    std::vector<int64_t> ts;
    std::vector<float> a, ap, sg, s1;
    PD_TRY(e->make_schedule(steps, eta, ts, a, ap, sg, s1));
    for (size_t i = 0; i < ts.size(); ++i) {
        if (timesteps) timesteps[i] = ts[i];
        if (alphas) alphas[i] = a[i];
        if (alphas_prev) alphas_prev[i] = ap[i];
        if (sigmas) sigmas[i] = sg[i];
        if (sqrt_one_minus_alphas) sqrt_one_minus_alphas[i] = s1[i];
    }
    return 0;
}

int pd_control_shape(pd_engine* e, int index, int32_t h, int32_t w, int32_t* C, int32_t* H, int32_t* W) {
    if (!e) { pd_set_error("null engine"); return 1; }
    const int n = (int)e->cnet.enc.size();
    if (index < 0 || index > n) { pd_set_error("control index out of range"); return 1; }
    int hh = h, ww = w;
    for (int i = 0; i <= (index < n ? index : n - 1); ++i)
        if (e->cnet.enc[i].kind == 2) { hh = (hh + 1) / 2; ww = (ww + 1) / 2; }
    if (C) *C = index < n ? e->cnet.enc[index].cout : e->cnet.enc[n - 1].cout;
    if (H) *H = hh;
    if (W) *W = ww;
    return 0;
}

static int copy_out(pd_engine* e, const float* dev_src, float* dst, size_t n, int mem) {
    HIP_OK(hipMemcpyAsync(dst, dev_src, n * 4, mem == PD_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, e->stream));
    HIP_OK(hipStreamSynchronize(e->stream));
    return 0;
}

int pd_eps(pd_engine* e, const float* x, const int64_t* t, const float* ctx, const float* pair, const float* query,
           const float* scales, int32_t Bf, int32_t h, int32_t w, int32_t mem, float* eps_out, float* residuals_out) {
    if (!e || !x || !t || !ctx || !pair || !query || !eps_out) { pd_set_error("null argument"); return 1; }
    pd_sample_args a{};
    a.batch = Bf; a.h = h; a.w = w; a.steps = 1; a.eta = 0.f; a.cfg_scale = 1.f; a.use_cfg = 0; a.temperature = 1.f;
    a.mem = mem; a.x_T = x; a.ctx_cond = ctx; a.pair = pair; a.query = query; a.control_scales = scales;
    PD_TRY(check_args(e, &a));
    HIP_OK(hipSetDevice(e->device));
    PD_TRY(e->fold_layernorms());
    std::vector<int64_t> th(Bf);
    if (mem == PD_MEM_DEVICE) HIP_OK(hipMemcpy(th.data(), t, (size_t)Bf * 8, hipMemcpyDeviceToHost));
    else memcpy(th.data(), t, (size_t)Bf * 8);
    Session& s = e->ses;
    s.active = false;
    s.a = a;
    s.Bf = Bf;
    s.S = 1;
    PD_TRY(e->ensure_arena(Bf, h, w, Bf, false));
    PD_TRY(e->session_setup(a, th.data(), Bf, true, false));
    const size_t mk = e->arena.mark();
    Act eps;
    PD_TRY(e->forward_eps(0, 1, scales, eps));
    // outputs back to NCHW fp32
    // one NCHW fp32 staging buffer, reused for every tensor read back (copy_out synchronises before the next use);
    // it is not part of the workspace dry run, so it is a plain allocation of its own
    const int C = e->cfg.out_channels, HW = h * w;
    size_t nmax = (size_t)Bf * C * HW;
    if (residuals_out)
        for (size_t i = 0; i <= e->cnet.enc.size(); ++i) nmax = std::max(nmax, (size_t)s.control[i].rows() * s.control[i].C);
    float* tmp = nullptr;
    HIP_OK(hipMalloc(reinterpret_cast<void**>(&tmp), nmax * 4));
    auto to_nchw = [&](const void* src, int dt, int b_, int c_, int h_, int w_, int cpad) -> int {
        if (launch_nhwc_to_nchw(src, dt, tmp, b_, c_, h_, w_, cpad, 1.f, e->stream)) { pd_set_error("pd_eps: layout conversion launch failed"); return 1; }
        return 0;
    };
    int rc = to_nchw(eps.p, eps.dt, Bf, C, h, w, eps.C);
    if (!rc) rc = copy_out(e, tmp, eps_out, (size_t)Bf * C * HW, mem);
    if (!rc && residuals_out) {
        size_t off = 0;
        for (size_t i = 0; i <= e->cnet.enc.size() && !rc; ++i) {
            const Act& c = s.control[i];
            const size_t n = (size_t)c.rows() * c.C;
            rc = to_nchw(c.p, c.dt, c.B, c.C, c.H, c.W, c.C);
            if (!rc) rc = copy_out(e, tmp, residuals_out + off, n, mem);
            off += n;
        }
    }
    (void)hipFree(tmp);
    e->arena.release(mk);
    return rc;
}

int pd_sample_begin(pd_engine* e, const pd_sample_args* args) {
    if (!e) { pd_set_error("null engine"); return 1; }
    return e->begin(args, false);
}

int pd_sample_step(pd_engine* e, int32_t i) {
    if (!e) { pd_set_error("null engine"); return 1; }
    return e->step(i);
}

int pd_sample_get(pd_engine* e, int32_t what, int32_t mem, float* out) {
    if (!e || !out) { pd_set_error("null argument"); return 1; }
    Session& s = e->ses;
    if (!s.active) { pd_set_error("no active sampling session"); return 1; }
    const int B = s.a.batch, C = e->cfg.in_channels, h = s.a.h, w = s.a.w;
    const size_t n = (size_t)B * C * h * w;
    const size_t mk = e->arena.mark();
    float* tmp = reinterpret_cast<float*>(e->arena.alloc(n * 4));
    PD_TRY(e->check_arena());
    const float* src = what == PD_GET_LATENTS ? s.x_state : what == PD_GET_PRED_X0 ? s.pred_x0 : s.eps_g;
    const int cpad = what == PD_GET_LATENTS ? 8 : C;
    if (launch_nhwc_to_nchw(src, DT_F32, tmp, B, C, h, w, cpad, 1.f, e->stream)) return 1;
    PD_TRY(copy_out(e, tmp, out, n, mem));
    e->arena.release(mk);
    return 0;
}

int pd_sample_set_guidance(pd_engine* e, float scale) {
    if (!e) { pd_set_error("null engine"); return 1; }
    if (!e->ses.active) { pd_set_error("no active sampling session"); return 1; }
    e->ses.a.cfg_scale = scale;
    return 0;
}

int pd_sample_set_latents(pd_engine* e, int32_t mem, const float* latents) {
    if (!e || !latents) { pd_set_error("null argument"); return 1; }
    Session& s = e->ses;
    if (!s.active) { pd_set_error("no active sampling session"); return 1; }
    const int B = s.a.batch, C = e->cfg.in_channels, h = s.a.h, w = s.a.w;
    const size_t n = (size_t)B * C * h * w;
    const size_t mk = e->arena.mark();
    const float* src = latents;
    if (mem != PD_MEM_DEVICE) {
        float* tmp = reinterpret_cast<float*>(e->arena.alloc(n * 4));
        PD_TRY(e->check_arena());
        HIP_OK(hipMemcpyAsync(tmp, latents, n * 4, hipMemcpyHostToDevice, e->stream));
        HIP_OK(hipStreamSynchronize(e->stream));
        src = tmp;
    }
    if (launch_nchw_to_nhwc(src, s.x_state, DT_F32, B, C, h, w, 8, e->stream)) return 1;
    if (launch_fill_x_in(s.x_state, s.x_in, B, s.Bf / B, C, 8, h * w, e->stream)) return 1;
    HIP_OK(hipStreamSynchronize(e->stream));
    e->arena.release(mk);
    return 0;
}

int pd_sample_eps_at(pd_engine* e, int64_t t, const float* scales13) {
    if (!e) { pd_set_error("null engine"); return 1; }
    Session& s = e->ses;
    if (!s.active) { pd_set_error("no active sampling session"); return 1; }
    const int row = s.emb_rows - 1;
    PD_TRY(e->compute_emb(e->unet, s.emb_u, &t, 1, row));
    PD_TRY(e->compute_emb(e->cnet, s.emb_c, &t, 1, row));
    const size_t mk = e->arena.mark();
    Act eps;
    PD_TRY(e->forward_eps(row, 0, scales13, eps));
    DdimCoef k{};
    k.cfg_scale = s.a.cfg_scale;
    const int C = e->cfg.in_channels, HW = s.a.h * s.a.w;
    if (launch_cfg_ddim(eps.p, eps.dt, eps.C, s.x_state, s.pred_x0, s.eps_g, s.x_in, nullptr, s.a.batch, HW, C, 8, s.a.use_cfg,
                        k, 1.f, 0, e->stream))
        return 1;
    e->arena.release(mk);
    return 0;
}

int pd_sample_end(pd_engine* e) {
    if (!e) { pd_set_error("null engine"); return 1; }
    HIP_OK(hipStreamSynchronize(e->stream));
    e->ses.active = false;
    return 0;
}

int pd_ddim_sample(pd_engine* e, const pd_sample_args* args, int32_t mem_out, float* latents_out, float* per_step_out) {
    if (!e || !latents_out) { pd_set_error("null argument"); return 1; }
    PD_TRY(e->begin(args, per_step_out != nullptr));
    if (e->opt_graph && !e->profiling) {
        PD_TRY(e->run_steps_graph());
    } else {
        for (int i = 0; i < e->ses.S; ++i) PD_TRY(e->step(i));
    }
    PD_TRY(pd_sample_get(e, PD_GET_LATENTS, mem_out, latents_out));
    if (per_step_out) {
        const size_t n = (size_t)(e->ses.S + 1) * args->batch * e->cfg.in_channels * args->h * args->w;
        PD_TRY(copy_out(e, e->ses.per_step, per_step_out, n, mem_out));
    }
    return pd_sample_end(e);
}

int pd_synchronize(pd_engine* e) {
    if (!e) { pd_set_error("null engine"); return 1; }
    HIP_OK(hipStreamSynchronize(e->stream));
    if (e->stream2) HIP_OK(hipStreamSynchronize(e->stream2));
    return 0;
}

void* pd_stream(pd_engine* e) { return e ? (void*)e->stream : nullptr; }

// Orders the engine's streams after the work already enqueued on `producer` (the stream that wrote the PD_MEM_DEVICE
// buffers about to be handed over; NULL = the legacy default stream).  The engine's streams are non-blocking, so without
// this nothing orders their reads after the producer's kernels.
int pd_wait_stream(pd_engine* e, void* producer) {
    if (!e) { pd_set_error("null engine"); return 1; }
    HIP_OK(hipSetDevice(e->device));
    hipEvent_t ev = nullptr;
    HIP_OK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    hipError_t r = hipEventRecord(ev, reinterpret_cast<hipStream_t>(producer));
    if (r == hipSuccess) r = hipStreamWaitEvent(e->stream, ev, 0);
    if (r == hipSuccess && e->stream2) r = hipStreamWaitEvent(e->stream2, ev, 0);
    (void)hipEventDestroy(ev);   // destruction is deferred until the recorded work has completed
    if (r != hipSuccess) { pd_set_error("pd_wait_stream: %s", hipGetErrorString(r)); return 1; }
    return 0;
}

int pd_set_option(pd_engine* e, const char* key, int64_t value) {
    if (!e || !key) { pd_set_error("null argument"); return 1; }
    if (!strcmp(key, "verbose")) { e->verbose = (int)value; return 0; }
    e->clear_graphs();   // every other knob changes what a step launches
    if (!strcmp(key, "gn_single")) { e->opt_gn_single = value != 0; return 0; }
    if (!strcmp(key, "gn_reg")) { g_gn_reg = value != 0; return 0; }
    if (!strcmp(key, "graph")) { e->opt_graph = value != 0; return 0; }
    if (!strcmp(key, "conv_patch")) { e->opt_patch = value != 0; return 0; }
    if (!strcmp(key, "conv_patch2")) { e->opt_patch2 = value != 0; return 0; }
    if (!strcmp(key, "conv_patch2_tiles")) { e->opt_patch2_tiles = (int)value; return 0; }
    if (!strcmp(key, "splitk_fused")) { e->opt_splitk_fused = value != 0; return 0; }
    if (!strcmp(key, "splitk_big")) { e->opt_splitk_big = (int)value; return 0; }
    if (!strcmp(key, "splitk_max")) { e->opt_splitk_max = (int)value; return 0; }
    if (!strcmp(key, "splitk_tiles")) { e->opt_splitk_tiles = (int)value; return 0; }
    if (!strcmp(key, "attn_legacy")) { e->opt_attn_legacy = value != 0; return 0; }
    if (!strcmp(key, "gn_fuse")) { e->opt_gn_fuse = value != 0; return 0; }
    if (!strcmp(key, "ln_fuse")) { e->opt_ln_fuse = (int)value; e->ln_dirty = true; return 0; }
    if (!strcmp(key, "st_fuse")) { e->opt_st_fuse = value != 0; e->ln_dirty = true; return 0; }
    if (!strcmp(key, "two_streams")) { e->opt_two_streams = value != 0; return 0; }
    if (!strcmp(key, "cfg_share")) { e->opt_cfg_share = value != 0; return 0; }
    if (!strcmp(key, "wide_tile")) { e->opt_wide = value != 0; return 0; }
    if (!strcmp(key, "slab_gn")) { e->opt_slab_gn = (int)value; return 0; }
    if (!strcmp(key, "ring")) { e->opt_ring = (int)value; return 0; }
    if (!strcmp(key, "ring_tile")) { e->opt_ring_tile = (int)value; return 0; }
    if (!strcmp(key, "ring_geglu")) { e->opt_ring_geglu = (int)value; return 0; }
    if (!strcmp(key, "ring_small")) { e->opt_ring_small = (int)value; return 0; }
    if (!strcmp(key, "short_k")) { e->opt_short_k = (int)value; return 0; }
    if (!strcmp(key, "patch_split")) { e->opt_patch_split = value != 0; return 0; }
    if (!strcmp(key, "ring_pp")) { e->opt_ring_pp = (int)value; return 0; }
    if (!strcmp(key, "patch4")) { e->opt_patch4 = value != 0; return 0; }
    if (!strcmp(key, "patch_split_fill")) { e->opt_patch_split_fill = (int)value; return 0; }
    if (!strcmp(key, "patch_split_min")) { if (value < 1) { pd_set_error("patch_split_min must be >= 1"); return 1; } e->opt_patch_split_min = (int)value; return 0; }
    if (!strcmp(key, "patch_split_tiles")) { e->opt_patch_split_tiles = (int)value; return 0; }
    if (!strcmp(key, "dense_tiles")) { e->opt_dense_tiles = (int)value; return 0; }
    if (!strcmp(key, "dense_k")) { e->opt_dense_k = (int)value; return 0; }
    if (!strcmp(key, "big_tile")) { e->opt_bigtile = value != 0; return 0; }
    if (!strcmp(key, "tile192")) { e->opt_tile192 = value != 0; return 0; }
    if (!strcmp(key, "gemv")) { e->opt_gemv = value != 0; return 0; }
    if (!strcmp(key, "sd3_fp8")) { e->opt_sd3_fp8 = (int)value; e->sd3_fp8_dirty = true; return 0; }
    if (!strcmp(key, "profile")) {
        (void)hipStreamSynchronize(e->stream);
        e->profiling = value != 0;
        e->prof.clear();
        e->ev_used = 0;
        if (e->profiling) {
            // an event pair around a kernel reads the kernel PLUS the marker packets between them: calibrate that on back-to-back event
            // pairs with NOTHING between them and take it off every bracket.  (Round 3 calibrated on a pair around an empty launch and
            // thereby subtracted that launch's own latency too: the per-launch averages came out 8 % under the kernel trace.)
            const int n = 64;
            std::vector<hipEvent_t> ev(2 * n);
            for (auto& x : ev) x = e->next_event();
            for (int i = 0; i < n; ++i) {
                (void)hipEventRecord(ev[2 * i], e->stream);
                (void)hipEventRecord(ev[2 * i + 1], e->stream);
            }
            (void)hipStreamSynchronize(e->stream);
            double tot = 0.0;
            int cnt = 0;
            for (int i = n / 2; i < n; ++i) {   // second half: warmed up
                float t = 0.f;
                if (ev[2 * i] && ev[2 * i + 1] && hipEventElapsedTime(&t, ev[2 * i], ev[2 * i + 1]) == hipSuccess) { tot += t; ++cnt; }
            }
            e->prof_overhead_ms = cnt ? (float)(tot / cnt) : 0.f;
            e->ev_used = 0;
        }
        return 0;
    }
    pd_set_error("unknown option '%s'", key);
    return 1;
}

int64_t pd_get_stat(pd_engine* e, const char* key) {
    if (!e || !key) return -1;
    if (!strcmp(key, "workspace_bytes")) return (int64_t)e->arena.cap;
    if (!strcmp(key, "weight_bytes")) return (int64_t)e->weight_bytes;
    if (!strcmp(key, "launches")) return (int64_t)e->launches;
    if (!strcmp(key, "gn_from_slabs")) return (int64_t)e->gn_from_slabs;
    if (!strcmp(key, "ring_launches")) return (int64_t)e->ring_launches;   // of which: gemm_ring.hip's persistent ring kernel
    if (!strcmp(key, "steps")) return (int64_t)e->ses.S;
    if (!strcmp(key, "cfg_shared")) return (int64_t)((e->ses.share_u ? 1 : 0) | (e->ses.share_c ? 2 : 0) | (e->ses.cn_cond_only ? 4 : 0));
    if (!strcmp(key, "event_overhead_ns")) return (int64_t)(e->prof_overhead_ms * 1e6f);
    return -1;
}

// Per-launch HIP-event timing collected while option "profile" is on.  klass: 0 igemm conv3x3,
// 1 igemm conv1x1 / linear, 2 attention, 3 conv3x3 LDS-patch kernel, -1 all.
int pd_profile_read(pd_engine* e, int32_t klass, double* total_ms, int64_t* n_launches, double* flops) {
    if (!e) { pd_set_error("null engine"); return 1; }
    HIP_OK(hipStreamSynchronize(e->stream));
    if (e->stream2) HIP_OK(hipStreamSynchronize(e->stream2));
    double ms = 0.0, fl = 0.0;
    int64_t n = 0;
    for (auto& r : e->prof) {
        if (klass >= 0 && r.klass != klass) continue;
        float t = 0.f;
        if (r.a && r.b && hipEventElapsedTime(&t, r.a, r.b) == hipSuccess) {
            t -= e->prof_overhead_ms;            // the empty-bracket time (see option "profile")
            ms += t > 0.f ? t : 0.f;
            fl += r.flops;
            ++n;
        }
    }
    if (total_ms) *total_ms = ms;
    if (n_launches) *n_launches = n;
    if (flops) *flops = fl;
    return 0;
}

// Debug aid: one CSV row per profiled launch (klass, M, N, K, taps-code, ms, flops).
int pd_profile_dump(pd_engine* e, const char* path) {
    if (!e || !path) { pd_set_error("null argument"); return 1; }
    HIP_OK(hipStreamSynchronize(e->stream));
    if (e->stream2) HIP_OK(hipStreamSynchronize(e->stream2));
    FILE* f = fopen(path, "w");
    if (!f) { pd_set_error("cannot open %s", path); return 1; }
    fprintf(f, "klass,M,N,K,taps,ms,flops\n");
    for (auto& r : e->prof) {
        float t = 0.f;
        if (r.a && r.b && hipEventElapsedTime(&t, r.a, r.b) == hipSuccess) {
            t -= e->prof_overhead_ms;
            fprintf(f, "%d,%d,%d,%d,%d,%.6f,%.0f\n", r.klass, r.M, r.N, r.K, r.taps, t > 0.f ? t : 0.f, r.flops);
        }
    }
    fclose(f);
    return 0;
}

int pd_bench_conv3x3(pd_engine* e, int32_t Bf, int32_t H, int32_t W, int32_t Cin, int32_t Cout, int32_t iters, float* ms) {
    if (!e || !ms || iters < 1) { pd_set_error("bad argument"); return 1; }
    HIP_OK(hipSetDevice(e->device));
    ConvW c;
    c.cin = Cin; c.cout = Cout; c.k = 3; c.stride = 1;
    const size_t owned0 = e->owned.size();
    e->make_mat(c.m, Cout, Cin * 9, 9, Cin, true);
    if (!c.m.w || !c.m.bias) { pd_set_error("allocation failed"); return 1; }
    const size_t eb = dt_size(e->T);
    void *in = nullptr, *out = nullptr;
    const size_t nin = (size_t)Bf * H * W * c.m.cin_pad, nout = (size_t)Bf * H * W * Cout;
    HIP_OK(hipMalloc(&in, nin * eb));
    HIP_OK(hipMalloc(&out, nout * eb));
    launch_fill_random(in, e->T, (long long)nin, 1.f, 0.f, 1, e->stream);
    launch_fill_random(c.m.w, e->T, (long long)c.m.N * c.m.Kpad, 1.0f / std::sqrt((float)Cin * 9), 0.f, 2, e->stream);
    Act a, o;
    a.p = in; a.B = Bf; a.H = H; a.W = W; a.C = c.m.cin_pad; a.dt = e->T;
    o.p = out; o.B = Bf; o.H = H; o.W = W; o.C = Cout; o.dt = e->T;
    int r = 0;
    for (int i = 0; i < 3 && !r; ++i) r = e->conv(c, a, o);
    hipEvent_t e0, e1;
    HIP_OK(hipEventCreate(&e0));
    HIP_OK(hipEventCreate(&e1));
    HIP_OK(hipEventRecord(e0, e->stream));
    for (int i = 0; i < iters && !r; ++i) r = e->conv(c, a, o);
    HIP_OK(hipEventRecord(e1, e->stream));
    HIP_OK(hipEventSynchronize(e1));
    float t = 0.f;
    HIP_OK(hipEventElapsedTime(&t, e0, e1));
    *ms = t / (float)iters;
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    hipFree(in);
    hipFree(out);
    while (e->owned.size() > owned0) {
        hipFree(e->owned.back());
        e->owned.pop_back();
    }
    return r;
}

int pd_bench_linear(pd_engine* e, int32_t M, int32_t K, int32_t N, int32_t residual, int32_t iters, float* ms) {
    if (!e || !ms || iters < 1 || M < 1 || K < 8 || N < 4) { pd_set_error("bad argument"); return 1; }
    HIP_OK(hipSetDevice(e->device));
    WMat m;
    const size_t owned0 = e->owned.size();
    const bool geglu = (residual & 2) != 0;     // bit 1: N counts the 2x-wide GEGLU projection (output N/2 columns)
    residual &= 1;
    e->make_mat(m, N, K, 1, K, true, geglu);
    if (!m.w || !m.bias) { pd_set_error("allocation failed"); return 1; }
    if (geglu) N = m.Nout;
    const size_t eb = dt_size(e->T);
    void *in = nullptr, *out = nullptr, *res = nullptr;
    HIP_OK(hipMalloc(&in, (size_t)M * m.cin_pad * eb));
    HIP_OK(hipMalloc(&out, (size_t)M * N * eb));
    if (residual) HIP_OK(hipMalloc(&res, (size_t)M * N * eb));
    launch_fill_random(in, e->T, (long long)M * m.cin_pad, 1.f, 0.f, 1, e->stream);
    if (res) launch_fill_random(res, e->T, (long long)M * N, 1.f, 0.f, 3, e->stream);
    launch_fill_random(m.w, e->T, (long long)m.N * m.Kpad, 1.0f / std::sqrt((float)K), 0.f, 2, e->stream);
    Act a, o, rr;
    a.p = in; a.B = 1; a.H = M; a.W = 1; a.C = m.cin_pad; a.dt = e->T;
    o.p = out; o.B = 1; o.H = M; o.W = 1; o.C = N; o.dt = e->T;
    rr = o; rr.p = res;
    int r = 0;
    for (int i = 0; i < 3 && !r; ++i) r = e->gemm(m, a, o, 1, 0, 0, 1.f, res ? &rr : nullptr, nullptr, 0, false, nullptr, 0, 0);
    hipEvent_t e0, e1;
    HIP_OK(hipEventCreate(&e0));
    HIP_OK(hipEventCreate(&e1));
    HIP_OK(hipEventRecord(e0, e->stream));
    for (int i = 0; i < iters && !r; ++i) r = e->gemm(m, a, o, 1, 0, 0, 1.f, res ? &rr : nullptr, nullptr, 0, false, nullptr, 0, 0);
    HIP_OK(hipEventRecord(e1, e->stream));
    HIP_OK(hipEventSynchronize(e1));
    float t = 0.f;
    HIP_OK(hipEventElapsedTime(&t, e0, e1));
    *ms = t / (float)iters;
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    hipFree(in);
    hipFree(out);
    if (res) hipFree(res);
    while (e->owned.size() > owned0) {
        hipFree(e->owned.back());
        e->owned.pop_back();
    }
    return r;
}

}  // extern "C"
