// pdengine: first-stage KL-VAE decoder (SURVEY.md §8f "next" row N1), built from the same kernels as the loop.
//   LatentDiffusion.decode_first_stage   ldm/models/diffusion/ddpm.py:820-828   (z / scale_factor)
//   AutoencoderKL.decode                 ldm/models/autoencoder.py:89-92        (post_quant_conv, decoder)
//   Decoder.forward                      ldm/modules/diffusionmodules/model.py:619-653
//   ResnetBlock / AttnBlock / Upsample   model.py:82-141, 144-202, 45-65        (GroupNorm eps 1e-6, swish)
#include <climits>
#include <cmath>

#include "engine.h"

static inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

void pd_engine::build_vres(const std::string& prefix, ResW& r, int cin, int cout) {
    r.cin = cin;
    r.cout = cout;
    r.eps = 1e-6f;
    reg_vec(prefix + "norm1.weight", cin, &r.gn1_g, 'g');
    reg_vec(prefix + "norm1.bias", cin, &r.gn1_b, 'e');
    build_conv(prefix + "conv1.", r.conv1, cin, cout, 3, 1);
    reg_vec(prefix + "norm2.weight", cout, &r.gn2_g, 'g');
    reg_vec(prefix + "norm2.bias", cout, &r.gn2_b, 'e');
    build_conv(prefix + "conv2.", r.conv2, cout, cout, 3, 1);
    r.has_skip = cin != cout;
    if (r.has_skip) build_conv(prefix + "nin_shortcut.", r.skip, cin, cout, 1, 1);
}

void pd_engine::build_vae() {
    if (cfg.vae_ch <= 0) return;
    reg_group = 1;
    const std::string P = "first_stage_model.", D = P + "decoder.";
    const int nl = cfg.vae_num_levels;
    const int top = cfg.vae_ch * cfg.vae_ch_mult[nl - 1];
    VaeW& v = vae;
    v.top = top;
    build_conv(D + "conv_in.", v.conv_in, cfg.in_channels, top, 3, 1);
    build_vres(D + "mid.block_1.", v.mid1, top, top);
    reg_vec(D + "mid.attn_1.norm.weight", top, &v.attn_g, 'g');
    reg_vec(D + "mid.attn_1.norm.bias", top, &v.attn_b, 'e');
    make_mat(v.qkv, 3 * top, top, 1, top, true);
    const char* nm[3] = {"q", "k", "v"};
    for (int i = 0; i < 3; ++i) {
        reg_mat(D + "mid.attn_1." + nm[i] + ".weight", {top, top, 1, 1}, &v.qkv, i * top, true);
        reg_bias(D + "mid.attn_1." + nm[i] + ".bias", &v.qkv, i * top, top);
    }
    build_conv(D + "mid.attn_1.proj_out.", v.proj_out, top, top, 1, 1);
    build_vres(D + "mid.block_2.", v.mid2, top, top);
    // parameters are registered in module order up.0 .. up.N-1; execution runs the highest level first
    std::vector<std::vector<std::pair<int, int>>> io(nl);
    std::vector<int> chs(nl);
    int block_in = top;
    for (int lvl = nl - 1; lvl >= 0; --lvl) {
        const int block_out = cfg.vae_ch * cfg.vae_ch_mult[lvl];
        for (int j = 0; j <= cfg.vae_num_res_blocks; ++j) {
            io[lvl].push_back({block_in, block_out});
            block_in = block_out;
        }
        chs[lvl] = block_in;
    }
    v.levels.resize(nl);   // index = execution order
    for (int lvl = 0; lvl < nl; ++lvl) {
        VaeLevel& L = v.levels[nl - 1 - lvl];
        L.blocks.resize(io[lvl].size());   // never resized again
        L.ch = chs[lvl];
        for (size_t j = 0; j < io[lvl].size(); ++j)
            build_vres(D + "up." + std::to_string(lvl) + ".block." + std::to_string(j) + ".", L.blocks[j], io[lvl][j].first,
                       io[lvl][j].second);
        L.up = lvl != 0;
        if (L.up) build_conv(D + "up." + std::to_string(lvl) + ".upsample.conv.", L.upconv, L.ch, L.ch, 3, 1);
    }
    reg_vec(D + "norm_out.weight", cfg.vae_ch, &v.out_g, 'g');
    reg_vec(D + "norm_out.bias", cfg.vae_ch, &v.out_b, 'e');
    build_conv(D + "conv_out.", v.conv_out, cfg.vae_ch, cfg.vae_out_ch, 3, 1);
    build_conv(P + "post_quant_conv.", v.post_quant, cfg.in_channels, cfg.in_channels, 1, 1);
    v.built = true;
    reg_group = 0;
}

// AttnBlock.forward: one head over all C channels, N = H*W tokens.  Scores are materialised per sample (fp32
// [N,N]) exactly like the reference's bmm + softmax; at 512x512 that is 64 MiB per sample, once per image.
int pd_engine::vae_attention(const Act& x, Act& out) {
    VaeW& v = vae;
    const int B = x.B, H = x.H, W = x.W, C = v.top, N = H * W;
    out = new_act(B, H, W, C, S);
    const size_t mk = arena.mark();
    Act a = new_act(B, H, W, C, T);
    PD_TRY(groupnorm(x, a, v.attn_g, v.attn_b, 1e-6f, false));
    Act qk = new_act(B, H, W, 2 * C, T);
    const int npad = round_up(N, 8);
    Act vt = new_act(B, C, 1, npad, T);
    PD_TRY(gemm(v.qkv, a, qk, 1, 0, 0, 1.f, nullptr, nullptr, 0, false, vt.p, 2 * C, npad));
    Act att = new_act(B, H, W, C, T);
    float* sc = reinterpret_cast<float*>(arena.alloc((size_t)N * N * sizeof(float)));
    void* pr = arena.alloc((size_t)N * npad * dt_size(T));
    if (!arena.dry) {
        if (npad != N) HIP_OK(hipMemsetAsync(pr, 0, (size_t)N * npad * dt_size(T), stream));
        const size_t eb = dt_size(T);
        const int bke = 128 / (int)eb;
        for (int b = 0; b < B; ++b) {
            const char* qb = reinterpret_cast<const char*>(qk.p) + (size_t)b * N * 2 * C * eb;
            GemmParams p{};
            // scores = (q k^T) * C^-0.5 : A = q rows, "weights" = k rows of the same buffer (row stride 2C)
            p.A = qb; p.W = qb + (size_t)C * eb; p.C = sc;
            p.M = N; p.N = N; p.K = C; p.Kpad = round_up(C, bke); p.ldw = 2 * C;
            p.lda = 2 * C; p.ldc = N; p.a_dt = T; p.c_dt = DT_F32; p.r_dt = DT_F32;
            p.taps = 1; p.Cin = C; p.Hin = N; p.Win = 1; p.Hout = N; p.Wout = 1; p.stride = 1;
            p.rows_per_sample = N; p.out_scale = (float)(1.0 / std::sqrt((double)C));
            p.vt_begin = INT_MAX; p.Nout = N; p.splitk = 1; p.big_tile = N >= 1024 ? 1 : 0;
            if (launch_gemm(p, P, stream)) { pd_set_error("vae attention: score GEMM launch failed"); return 1; }
            if (launch_softmax_rows(sc, pr, T, N, N, stream)) { pd_set_error("vae attention: softmax launch failed"); return 1; }
            // out = P v : A = P [N, N] (row stride npad when padded), "weights" = V^T [C][npad]
            GemmParams o{};
            o.A = pr; o.W = reinterpret_cast<const char*>(vt.p) + (size_t)b * C * npad * eb;
            o.C = reinterpret_cast<char*>(att.p) + (size_t)b * N * C * eb;
            o.M = N; o.N = C; o.K = N; o.Kpad = round_up(N, bke); o.ldw = npad;
            o.lda = N; o.ldc = C; o.a_dt = T; o.c_dt = T; o.r_dt = DT_F32;
            o.taps = 1; o.Cin = N; o.Hin = N; o.Win = 1; o.Hout = N; o.Wout = 1; o.stride = 1;
            o.rows_per_sample = N; o.out_scale = 1.f; o.vt_begin = INT_MAX; o.Nout = C; o.splitk = 1;
            o.big_tile = N >= 1024 ? 1 : 0;
            if (launch_gemm(o, P, stream)) { pd_set_error("vae attention: value GEMM launch failed"); return 1; }
            launches += 3;
        }
    }
    PD_TRY(conv(v.proj_out, att, out, 0, 1.f, &x));
    arena.release(mk);
    return 0;
}

int pd_engine::vae_forward(const float* latents_dev, int B, int h, int w, float* out_dev) {
    VaeW& v = vae;
    Act z = new_act(B, h, w, 8, T);
    if (!arena.dry) {
        ++launches;
        if (launch_nchw_to_nhwc(latents_dev, z.p, T, B, cfg.in_channels, h, w, 8, stream, (float)(1.0 / cfg.scale_factor))) return 1;
    }
    Act zq = new_act(B, h, w, 8, T);   // post_quant_conv writes channels 0..in_ch-1; the pad channels must read as 0
    if (!arena.dry) HIP_OK(hipMemsetAsync(zq.p, 0, zq.bytes(), stream));
    PD_TRY(conv(v.post_quant, z, zq));
    Act hcur = new_act(B, h, w, v.top, S);
    PD_TRY(conv(v.conv_in, zq, hcur));
    Act t;
    PD_TRY(resblock(v.mid1, hcur, t, nullptr, 0));
    hcur = t;
    PD_TRY(vae_attention(hcur, t));
    hcur = t;
    PD_TRY(resblock(v.mid2, hcur, t, nullptr, 0));
    hcur = t;
    for (VaeLevel& L : v.levels) {
        for (ResW& r : L.blocks) {
            PD_TRY(resblock(r, hcur, t, nullptr, 0));
            hcur = t;
        }
        if (L.up) {
            Act u = new_act(hcur.B, hcur.H * 2, hcur.W * 2, hcur.C, S);
            PD_TRY(conv(L.upconv, hcur, u, 0, 1.f, nullptr, nullptr, 0, /*ups=*/1));   // nearest x2 then conv, model.py:62-64
            hcur = u;
        }
    }
    Act img = new_act(hcur.B, hcur.H, hcur.W, round_up(cfg.vae_out_ch, 4), DT_F32);
    PD_TRY(conv_gn(v.conv_out, hcur, img, v.out_g, v.out_b, 1e-6f, true, nullptr, nullptr, 0));
    if (!arena.dry) {
        ++launches;
        if (launch_nhwc_to_nchw(img.p, DT_F32, out_dev, B, cfg.vae_out_ch, img.H, img.W, img.C, 1.f, stream)) return 1;
    }
    return 0;
}

extern "C" int pd_vae_weights_missing(pd_engine* e) {
    int n = 0;
    if (e)
        for (auto& p : e->params) n += (p.group == 1 && !p.loaded) ? 1 : 0;
    return n;
}

extern "C" int pd_vae_decode(pd_engine* e, const float* latents, int32_t B, int32_t h, int32_t w, int32_t mem, float* images_out) {
    if (!e || !latents || !images_out || B < 1 || h < 1 || w < 1) { pd_set_error("bad argument"); return 1; }
    if (!e->vae.built) { pd_set_error("this engine was created without a VAE decoder (vae_ch = 0)"); return 1; }
    if (e->ses.active) { pd_set_error("pd_vae_decode: end the sampling session first"); return 1; }
    if ((h * w) % 64) { pd_set_error("pd_vae_decode: h*w must be a multiple of 64 (latents of 64x64-pixel multiples)"); return 1; }
    for (auto& p : e->params)
        if (p.group == 1 && !p.loaded) { pd_set_error("VAE weights not loaded: '%s' (and possibly more)", p.name.c_str()); return 1; }
    HIP_OK(hipSetDevice(e->device));
    const int H = 8 * h, W = 8 * w;
    const size_t n_in = (size_t)B * e->cfg.in_channels * h * w, n_out = (size_t)B * e->cfg.vae_out_ch * H * W;
    // the decoder runs in the ControlNet context's workspace (idle outside a sampling step) on the main stream
    std::swap(e->arena, e->arena2);
    Arena saved = e->arena;
    e->arena.base = nullptr; e->arena.cap = 0; e->arena.top = 0; e->arena.peak = 0; e->arena.dry = true;
    int r = e->vae_forward(nullptr, B, h, w, nullptr);
    const size_t need = e->arena.peak + (n_in + n_out) * sizeof(float) + (64u << 20);
    e->arena = saved;
    e->arena.dry = false;
    if (!r && need > e->arena.cap) {
        hipStreamSynchronize(e->stream);
        if (e->stream2) hipStreamSynchronize(e->stream2);
        e->clear_graphs();   // captured step loops point into this workspace
        if (e->arena.base) hipFree(e->arena.base);
        e->arena.base = nullptr; e->arena.cap = 0;
        void* p = nullptr;
        if (hipMalloc(&p, need) != hipSuccess) { pd_set_error("VAE workspace allocation of %.2f GiB failed", (double)need / (1 << 30)); r = 1; }
        else { e->arena.base = reinterpret_cast<char*>(p); e->arena.cap = need; }
    }
    if (!r) {
        e->arena.top = 0; e->arena.peak = 0;
        float* din = reinterpret_cast<float*>(e->arena.alloc(n_in * sizeof(float)));
        float* dout = reinterpret_cast<float*>(e->arena.alloc(n_out * sizeof(float)));
        const float* src = latents;
        if (mem != PD_MEM_DEVICE) {
            if (hipMemcpyAsync(din, latents, n_in * sizeof(float), hipMemcpyHostToDevice, e->stream) != hipSuccess ||
                hipStreamSynchronize(e->stream) != hipSuccess) { pd_set_error("latent upload failed"); r = 1; }
            src = din;
        }
        if (!r) r = e->vae_forward(src, B, h, w, dout);
        if (!r) {
            if (hipMemcpyAsync(images_out, dout, n_out * sizeof(float),
                               mem == PD_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, e->stream) != hipSuccess ||
                hipStreamSynchronize(e->stream) != hipSuccess) { pd_set_error("image read-back failed"); r = 1; }
        }
        e->arena.top = 0;
    }
    std::swap(e->arena, e->arena2);
    return r;
}
