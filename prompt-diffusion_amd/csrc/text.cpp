// pdengine: cond-stage CLIP text transformer (SURVEY.md §8f "next" row N3), built from the same kernels as the loop.
//   FrozenCLIPEmbedder.forward            ldm/modules/encoders/modules.py:118-128  (layer "last": last_hidden_state)
//   (D) PromptDiffusionPipeline.encode_prompt  pipeline_prompt_diffusion.py:308-487 (text_encoder(ids)[0])
// The module behind both is transformers' CLIPTextModel ("openai/clip-vit-large-patch14"): token + position
// embeddings, 12 pre-LN blocks (causal multi-head self-attention with biased q/k/v/out projections, quick-GELU MLP),
// final LayerNorm.  Weight names are the checkpoint's: cond_stage_model.transformer.text_model.*.
// Tokenisation (BPE vocabulary files) stays with the caller: the boundary takes token ids.
#include <climits>
#include <cmath>

#include "engine.h"

static inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

void pd_engine::build_text() {
    if (cfg.text_layers <= 0) return;
    reg_group = 2;
    const std::string P = "cond_stage_model.transformer.text_model.";
    const int C = cfg.context_dim, F = cfg.text_ff, L = cfg.context_len;
    TextW& t = text;
    make_mat(t.tok, cfg.text_vocab, C, 1, C, false);
    reg_mat(P + "embeddings.token_embedding.weight", {cfg.text_vocab, C}, &t.tok, 0, false);
    make_mat(t.pos, L, C, 1, C, false);
    reg_mat(P + "embeddings.position_embedding.weight", {L, C}, &t.pos, 0, false);
    t.layers.resize(cfg.text_layers);   // never resized again (Params point into it)
    for (int i = 0; i < cfg.text_layers; ++i) {
        TextLayerW& l = t.layers[i];
        const std::string Lp = P + "encoder.layers." + std::to_string(i) + ".";
        make_mat(l.qkv, 3 * C, C, 1, C, true);   // rows [0,C) q, [C,2C) k, [2C,3C) v -- the attention kernel's layout
        const char* nm[3] = {"k_proj", "v_proj", "q_proj"};   // module order of the checkpoint
        const int off[3] = {C, 2 * C, 0};
        for (int j = 0; j < 3; ++j) {
            reg_mat(Lp + "self_attn." + nm[j] + ".weight", {C, C}, &l.qkv, off[j], false);
            reg_bias(Lp + "self_attn." + nm[j] + ".bias", &l.qkv, off[j], C);
        }
        make_mat(l.out, C, C, 1, C, true);
        reg_mat(Lp + "self_attn.out_proj.weight", {C, C}, &l.out, 0, false);
        reg_bias(Lp + "self_attn.out_proj.bias", &l.out, 0, C);
        reg_vec(Lp + "layer_norm1.weight", C, &l.ln1_g, 'g');
        reg_vec(Lp + "layer_norm1.bias", C, &l.ln1_b, 'e');
        make_mat(l.fc1, F, C, 1, C, true);
        reg_mat(Lp + "mlp.fc1.weight", {F, C}, &l.fc1, 0, false);
        reg_bias(Lp + "mlp.fc1.bias", &l.fc1, 0, F);
        make_mat(l.fc2, C, F, 1, F, true);
        reg_mat(Lp + "mlp.fc2.weight", {C, F}, &l.fc2, 0, false);
        reg_bias(Lp + "mlp.fc2.bias", &l.fc2, 0, C);
        reg_vec(Lp + "layer_norm2.weight", C, &l.ln2_g, 'g');
        reg_vec(Lp + "layer_norm2.bias", C, &l.ln2_b, 'e');
    }
    reg_vec(P + "final_layer_norm.weight", C, &t.fln_g, 'g');
    reg_vec(P + "final_layer_norm.bias", C, &t.fln_b, 'e');
    t.built = true;
    reg_group = 0;
}

// tokens [B, L] int32 (device) -> last_hidden_state [B, L, C] fp32 (device)
int pd_engine::text_forward(const int* ids_dev, int B, float* out_dev, int clip_skip) {
    TextW& t = text;
    const int C = cfg.context_dim, F = cfg.text_ff, L = cfg.context_len, H = cfg.text_heads;
    const size_t eb = dt_size(T);
    Act x = new_act(B, L, 1, C, S);
    if (!arena.dry) {
        ++launches;
        if (launch_embed_tokens(ids_dev, t.tok.w, t.tok.Kpad, t.pos.w, t.pos.Kpad, T, x.p, S, B, L, C, cfg.text_vocab, stream)) {
            pd_set_error("text embedding launch failed");
            return 1;
        }
    }
    const int lpad = round_up(L, 8);
    const int n_run = (int)t.layers.size() - clip_skip;   // hidden_states[-(clip_skip+1)]: output of block n - clip_skip
    for (int li = 0; li < n_run; ++li) {
        TextLayerW& l = t.layers[li];
        const size_t mk = arena.mark();
        Act ln = new_act(B, L, 1, C, T);
        PD_TRY(layernorm(x, ln, l.ln1_g, l.ln1_b));
        Act qk = new_act(B, L, 1, 2 * C, T);
        Act vt = new_act(B, C, 1, lpad, T);
        if (!arena.dry && lpad != L) HIP_OK(hipMemsetAsync(vt.p, 0, vt.bytes(), stream));   // pad keys of V^T must read as 0
        PD_TRY(gemm(l.qkv, ln, qk, 1, 0, 0, 1.f, nullptr, nullptr, 0, false, vt.p, 2 * C, lpad));
        Act att = new_act(B, L, 1, C, T);
        PD_TRY(attention(qk.p, 2 * C, reinterpret_cast<char*>(qk.p) + (size_t)C * eb, 2 * C, vt.p, lpad, att.p, C, B, L, L, C, H,
                         /*causal=*/true));
        Act h1 = new_act(B, L, 1, C, S);
        PD_TRY(gemm(l.out, att, h1, 1, 0, 0, 1.f, &x, nullptr, 0, false, nullptr, 0, 0));
        PD_TRY(layernorm(h1, ln, l.ln2_g, l.ln2_b));
        Act f = new_act(B, L, 1, F, T);
        PD_TRY(gemm(l.fc1, ln, f, 1, 0, /*act=quick_gelu*/ 3, 1.f, nullptr, nullptr, 0, false, nullptr, 0, 0));
        Act h2 = new_act(B, L, 1, C, S);
        PD_TRY(gemm(l.fc2, f, h2, 1, 0, 0, 1.f, &h1, nullptr, 0, false, nullptr, 0, 0));
        // carry the block output down to the slot below this block's temporaries
        if (!arena.dry) HIP_OK(hipMemcpyAsync(x.p, h2.p, x.bytes(), hipMemcpyDeviceToDevice, stream));
        arena.release(mk);
    }
    Act out = new_act(B, L, 1, C, DT_F32);
    PD_TRY(layernorm(x, out, t.fln_g, t.fln_b));
    if (!arena.dry) HIP_OK(hipMemcpyAsync(out_dev, out.p, out.bytes(), hipMemcpyDeviceToDevice, stream));
    return 0;
}

extern "C" int pd_text_weights_missing(pd_engine* e) {
    int n = 0;
    if (e)
        for (auto& p : e->params) n += (p.group == 2 && !p.loaded) ? 1 : 0;
    return n;
}

extern "C" int pd_text_encode_ex(pd_engine* e, const int32_t* ids, int32_t B, int32_t mem, int32_t clip_skip, float* out) {
    if (!e || !ids || !out || B < 1) { pd_set_error("bad argument"); return 1; }
    if (clip_skip < 0 || (e->text.built && clip_skip > (int)e->text.layers.size())) {   // k = text_layers: hidden_states[0], the embeddings
        pd_set_error("clip_skip %d out of range [0, %d]", clip_skip, e->cfg.text_layers);
        return 1;
    }
    if (!e->text.built) { pd_set_error("this engine was created without a text transformer (text_layers = 0)"); return 1; }
    if (e->ses.active) { pd_set_error("pd_text_encode: end the sampling session first"); return 1; }
    for (auto& p : e->params)
        if (p.group == 2 && !p.loaded) { pd_set_error("text transformer weights not loaded: '%s' (and possibly more)", p.name.c_str()); return 1; }
    HIP_OK(hipSetDevice(e->device));
    const int L = e->cfg.context_len, C = e->cfg.context_dim;
    const size_t n_in = (size_t)B * L, n_out = (size_t)B * L * C;
    // like the VAE decoder: the ControlNet context's workspace (idle outside a sampling step), main stream
    std::swap(e->arena, e->arena2);
    Arena saved = e->arena;
    e->arena.base = nullptr; e->arena.cap = 0; e->arena.top = 0; e->arena.peak = 0; e->arena.dry = true;
    int r = e->text_forward(nullptr, B, nullptr, 0);
    const size_t need = e->arena.peak + n_in * sizeof(int) + n_out * sizeof(float) + (64u << 20);
    e->arena = saved;
    e->arena.dry = false;
    if (!r && need > e->arena.cap) {
        hipStreamSynchronize(e->stream);
        if (e->stream2) hipStreamSynchronize(e->stream2);
        e->clear_graphs();   // captured step loops point into this workspace
        if (e->arena.base) hipFree(e->arena.base);
        e->arena.base = nullptr; e->arena.cap = 0;
        void* p = nullptr;
        if (hipMalloc(&p, need) != hipSuccess) { pd_set_error("text workspace allocation of %.2f GiB failed", (double)need / (1 << 30)); r = 1; }
        else { e->arena.base = reinterpret_cast<char*>(p); e->arena.cap = need; }
    }
    if (!r) {
        e->arena.top = 0; e->arena.peak = 0;
        int* din = reinterpret_cast<int*>(e->arena.alloc(n_in * sizeof(int)));
        float* dout = reinterpret_cast<float*>(e->arena.alloc(n_out * sizeof(float)));
        if (hipMemcpyAsync(din, ids, n_in * sizeof(int), mem == PD_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice,
                           e->stream) != hipSuccess) { pd_set_error("token upload failed"); r = 1; }
        if (!r) r = e->text_forward(din, B, dout, clip_skip);
        if (!r) {
            if (hipMemcpyAsync(out, dout, n_out * sizeof(float), mem == PD_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost,
                               e->stream) != hipSuccess ||
                hipStreamSynchronize(e->stream) != hipSuccess) { pd_set_error("embedding read-back failed"); r = 1; }
        }
        e->arena.top = 0;
    }
    std::swap(e->arena, e->arena2);
    return r;
}

extern "C" int pd_text_encode(pd_engine* e, const int32_t* ids, int32_t B, int32_t mem, float* out) {
    return pd_text_encode_ex(e, ids, B, mem, 0, out);
}
