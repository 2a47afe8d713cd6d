// pdengine: SD3 / MMDiT variant of the hot path (SURVEY.md §8f "next" row N4), built from the same contraction kernels.
//   SD3PromptDiffusionModel.forward                        promptdiffusioncontrolnet_sd3.py:362-483
//   self.transformer(..., block_controlnet_hidden_states)   promptdiffusioncontrolnetpipeline_sd3.py:1226-1234
//   CFG + FlowMatchEuler scheduler.step                     promptdiffusioncontrolnetpipeline_sd3.py:1237-1243
// The modules behind those calls (SD3Transformer2DModel, JointTransformerBlock, AdaLayerNormZero / Continuous, PatchEmbed,
// CombinedTimestepTextProjEmbeddings) are diffusers' -- not vendored in the reference and not installed offline -- so this
// file follows the published MMDiT (Esser et al. 2024) under diffusers' state-dict names; parity is pinned to
// oracle/sd3_oracle.py only ("parity unpinned").
//
// Per block, 2-byte modes: 2 AdaLN passes + 2 QKV GEMMs writing one joint [B, N + S] q|k buffer and one V^T (row-remapped
// epilogue stores, no concat pass) + 2 attention launches (image queries, context queries; both over all N + S keys) +
// 2 gated out-projections (gate * (acc + bias) + residual in the GEMM epilogue) + per stream AdaLN, tanh-GELU GEMM, gated GEMM.
// All 2 * layers modulation vectors of a net come from ONE GEMM per evaluation (they depend on temb alone).
#include <climits>
#include <cmath>
#include <cstring>
#include <string>

#include "engine.h"

static inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

void pd_engine::build_sd3_net(const std::string& P, Sd3NetW& net, bool controlnet) {
    const int D = sd3.heads * sd3.head_dim, C = sd3.in_channels, ps = sd3.patch_size;
    net.layers = controlnet ? sd3.cn_layers : sd3.layers;
    net.pos_max = controlnet && sd3.cn_pos_embed_max_size ? sd3.cn_pos_embed_max_size : sd3.pos_embed_max_size;
    auto lin = [&](WMat& m, const std::string& name, int n, int k) {
        make_mat(m, n, k, 1, k, true);
        reg_mat(P + name + ".weight", {n, k}, &m, 0, false);
        reg_bias(P + name + ".bias", &m, 0, n);
    };
    auto patch = [&](WMat& m, const std::string& name) {   // Conv2d(C, D, ps, stride ps): tap-major rows like every conv
        make_mat(m, D, C * ps * ps, ps * ps, C, true);
        reg_mat(P + name + ".weight", {D, C, ps, ps}, &m, 0, true);
        reg_bias(P + name + ".bias", &m, 0, D);
    };
    patch(net.pe, "pos_embed.proj");
    reg_vec(P + "pos_embed.pos_embed", net.pos_max * net.pos_max * D, &net.pos, 'b');
    params.back().shape = {1, (int64_t)net.pos_max * net.pos_max, D};
    if (controlnet) {
        patch(net.pe_in, "pos_embed_input.proj");
        build_conv(P + "down_proj.", net.down_proj, 6, 3, 3, 1);   // encode_support_pair's Conv2d(6, 3, 3, padding=1), :114
    }
    const uint32_t dual_mask = controlnet ? sd3.cn_dual_mask : sd3.dual_mask;
    const int hd = sd3.head_dim;
    lin(net.t1, "time_text_embed.timestep_embedder.linear_1", D, 256);
    lin(net.t2, "time_text_embed.timestep_embedder.linear_2", D, D);
    lin(net.p1, "time_text_embed.text_embedder.linear_1", D, sd3.pooled_dim);
    lin(net.p2, "time_text_embed.text_embedder.linear_2", D, D);
    // joint_attention_dim = None (promptdiffusioncontrolnet_sd3.py:147-160): SD3SingleTransformerBlocks, no context stream
    net.single = controlnet && sd3.cn_single != 0;
    if (!net.single) lin(net.ctx_emb, "context_embedder", D, sd3.joint_dim);
    // modulation matrix: every norm1.linear / norm1_context.linear (+ norm_out.linear) of the net stacked row-wise
    net.blocks.resize(net.layers);   // never resized again (Params point into it)
    int rows = 0;
    for (int i = 0; i < net.layers; ++i) {
        Sd3BlockW& b = net.blocks[i];
        b.pre_only = !controlnet && i == net.layers - 1;
        b.single = net.single;
        b.dual = i < 32 && ((dual_mask >> i) & 1u) && !b.pre_only && !b.single;
        b.mod_off = rows; rows += (b.dual ? 9 : 6) * D;
        b.mod_c_off = rows; rows += b.single ? 0 : (b.pre_only ? 2 : 6) * D;
    }
    if (!controlnet) { net.norm_out_off = rows; rows += 2 * D; }
    net.mod_rows = rows;
    make_mat(net.mod, rows, D, 1, D, true);
    for (int i = 0; i < net.layers; ++i) {
        Sd3BlockW& b = net.blocks[i];
        const std::string Bp = P + "transformer_blocks." + std::to_string(i) + ".";
        reg_mat(Bp + "norm1.linear.weight", {(b.dual ? 9 : 6) * D, D}, &net.mod, b.mod_off, false);
        reg_bias(Bp + "norm1.linear.bias", &net.mod, b.mod_off, (b.dual ? 9 : 6) * D);
        const int nc = (b.pre_only ? 2 : 6) * D;
        if (!b.single) {
            reg_mat(Bp + "norm1_context.linear.weight", {nc, D}, &net.mod, b.mod_c_off, false);
            reg_bias(Bp + "norm1_context.linear.bias", &net.mod, b.mod_c_off, nc);
        }
        make_mat(b.qkv, 3 * D, D, 1, D, true);
        if (!b.single) make_mat(b.qkv_c, 3 * D, D, 1, D, true);
        const char* nm[3] = {"to_q", "to_k", "to_v"};
        const char* nmc[3] = {"add_q_proj", "add_k_proj", "add_v_proj"};
        for (int j = 0; j < 3; ++j) {
            reg_mat(Bp + "attn." + nm[j] + ".weight", {D, D}, &b.qkv, j * D, false);
            reg_bias(Bp + "attn." + nm[j] + ".bias", &b.qkv, j * D, D);
            if (b.single) continue;
            reg_mat(Bp + "attn." + nmc[j] + ".weight", {D, D}, &b.qkv_c, j * D, false);
            reg_bias(Bp + "attn." + nmc[j] + ".bias", &b.qkv_c, j * D, D);
        }
        lin(b.out, "transformer_blocks." + std::to_string(i) + ".attn.to_out.0", D, D);
        if (sd3.qk_norm) {
            reg_vec(Bp + "attn.norm_q.weight", hd, &b.nq, 'g');
            reg_vec(Bp + "attn.norm_k.weight", hd, &b.nk, 'g');
            if (!b.single) {
                reg_vec(Bp + "attn.norm_added_q.weight", hd, &b.naq, 'g');
                reg_vec(Bp + "attn.norm_added_k.weight", hd, &b.nak, 'g');
            }
        }
        if (b.dual) {
            make_mat(b.qkv2, 3 * D, D, 1, D, true);
            for (int j = 0; j < 3; ++j) {
                reg_mat(Bp + "attn2." + nm[j] + ".weight", {D, D}, &b.qkv2, j * D, false);
                reg_bias(Bp + "attn2." + nm[j] + ".bias", &b.qkv2, j * D, D);
            }
            lin(b.out2, "transformer_blocks." + std::to_string(i) + ".attn2.to_out.0", D, D);
            if (sd3.qk_norm) {
                reg_vec(Bp + "attn2.norm_q.weight", hd, &b.nq2, 'g');
                reg_vec(Bp + "attn2.norm_k.weight", hd, &b.nk2, 'g');
            }
        }
        lin(b.ff1, "transformer_blocks." + std::to_string(i) + ".ff.net.0.proj", 4 * D, D);
        lin(b.ff2, "transformer_blocks." + std::to_string(i) + ".ff.net.2", D, 4 * D);
        if (!b.pre_only && !b.single) {
            lin(b.out_c, "transformer_blocks." + std::to_string(i) + ".attn.to_add_out", D, D);
            lin(b.ffc1, "transformer_blocks." + std::to_string(i) + ".ff_context.net.0.proj", 4 * D, D);
            lin(b.ffc2, "transformer_blocks." + std::to_string(i) + ".ff_context.net.2", D, 4 * D);
        }
    }
    if (controlnet) {
        net.zero.resize(net.layers);
        for (int i = 0; i < net.layers; ++i) lin(net.zero[i], "controlnet_blocks." + std::to_string(i), D, D);
    } else {
        reg_mat(P + "norm_out.linear.weight", {2 * D, D}, &net.mod, net.norm_out_off, false);
        reg_bias(P + "norm_out.linear.bias", &net.mod, net.norm_out_off, 2 * D);
        lin(net.proj_out, "proj_out", ps * ps * sd3.out_channels, D);
    }
    net.built = true;
}

// patch embedding (+ ControlNet condition embeddings), temb, context embedding, modulation vectors
int pd_engine::sd3_embed(Sd3NetW& net, const Sd3Io& io, bool controlnet, Act& hs, Act& c, Act& modbuf) {
    const int D = sd3.heads * sd3.head_dim, ps = sd3.patch_size, h = io.H / ps, w = io.W / ps, N = h * w, B = io.B;
    hs = new_act(B, N, 1, D, S);
    c = new_act(B, net.single ? 0 : io.S, 1, D, S);   // a single-block net has no context stream
    modbuf = new_act(B, 1, 1, net.mod_rows, DT_F32);
    const size_t mk = arena.mark();
    WMat pe = net.pe;            // the GEMM sees the patch conv as a linear layer over the patchified rows
    pe.taps = 1; pe.cin = pe.cin_pad = net.pe.K;
    Act pos = new_act(B, N, 1, D, DT_F32);
    Act rows = new_act(B, N, 1, net.pe.K, T);
    if (!arena.dry) {
        PD_TRY(check_arena());
        launches += 2;
        if (launch_pos_crop(net.pos, reinterpret_cast<float*>(pos.p), B, h, w, net.pos_max, D, stream) ||
            launch_patchify(io.latents, rows.p, T, B, sd3.in_channels, io.H, io.W, ps, net.pe.cin_pad, net.pe.K, stream)) {
            pd_set_error("sd3: patch embedding launch failed (latent %dx%d, pos_embed_max_size %d)", io.H, io.W, net.pos_max);
            return 1;
        }
    }
    PD_TRY(gemm(pe, rows, hs, 1, 0, 0, 1.f, &pos, nullptr, 0, false, nullptr, 0, 0));
    if (controlnet) {   // hidden_states + pos_embed_input(cond) + pos_embed_input(example pair)   (:440)
        WMat pi = net.pe_in;
        pi.taps = 1; pi.cin = pi.cin_pad = net.pe_in.K;
        const float* src[2] = {io.cond, io.pair};
        for (int j = 0; j < 2; ++j) {
            if (!arena.dry) {
                ++launches;
                if (launch_patchify(src[j], rows.p, T, B, sd3.in_channels, io.H, io.W, ps, net.pe_in.cin_pad, net.pe_in.K, stream)) {
                    pd_set_error("sd3: condition patchify launch failed");
                    return 1;
                }
            }
            PD_TRY(gemm(pi, rows, hs, 1, 0, 0, 1.f, &hs, nullptr, 0, false, nullptr, 0, 0));   // in place: R == C element-wise
        }
    }
    // temb = MLP(sinusoid(t)) + MLP(pooled)
    Act sin_t = new_act(B, 1, 1, 256, DT_F32);
    Act pooled = new_act(B, 1, 1, round_up(sd3.pooled_dim, 8), DT_F32);
    Act u = new_act(B, 1, 1, D, DT_F32), temb = new_act(B, 1, 1, D, DT_F32);
    if (!arena.dry) {
        ++launches;
        if (launch_timestep_embedding(io.t_host, B, reinterpret_cast<float*>(sin_t.p), stream)) {   // timesteps travel as kernel arguments
            pd_set_error("sd3: timestep embedding launch failed (batch %d)", B);
            return 1;
        }
        HIP_OK(hipMemsetAsync(pooled.p, 0, pooled.bytes(), stream));
        const float* src = controlnet ? io.cn_pooled : io.pooled;   // ControlNet: null = zero pooled projections
        if (src)
            HIP_OK(hipMemcpy2DAsync(pooled.p, (size_t)pooled.C * 4, src, (size_t)sd3.pooled_dim * 4, (size_t)sd3.pooled_dim * 4, B,
                                    hipMemcpyDeviceToDevice, stream));
    }
    PD_TRY(gemm(net.t1, sin_t, u, 1, 0, /*SiLU*/ 1, 1.f, nullptr, nullptr, 0, false, nullptr, 0, 0));
    PD_TRY(gemm(net.t2, u, temb, 1, 0, 0, 1.f, nullptr, nullptr, 0, false, nullptr, 0, 0));
    PD_TRY(gemm(net.p1, pooled, u, 1, 0, 1, 1.f, nullptr, nullptr, 0, false, nullptr, 0, 0));
    PD_TRY(gemm(net.p2, u, temb, 1, 0, 0, 1.f, &temb, nullptr, 0, false, nullptr, 0, 0));
    // every AdaLN modulation of the net: Linear(SiLU(temb)) stacked
    PD_TRY(gemm(net.mod, temb, modbuf, 1, 0, 0, 1.f, nullptr, nullptr, 0, /*a_silu=*/true, nullptr, 0, 0));
    // context_embedder
    if (net.single) { arena.release(mk); return 0; }
    Act ctx = new_act(B, io.S, 1, round_up(sd3.joint_dim, 8), T);
    if (!arena.dry) {
        ++launches;
        if (launch_cast_rows(io.context, ctx.p, T, (long long)B * io.S, sd3.joint_dim, ctx.C, stream)) {
            pd_set_error("sd3: context cast launch failed");
            return 1;
        }
    }
    PD_TRY(gemm(net.ctx_emb, ctx, c, 1, 0, 0, 1.f, nullptr, nullptr, 0, false, nullptr, 0, 0));
    arena.release(mk);
    return 0;
}

// JointTransformerBlock.forward (diffusers attention.py; MMDiT block of Esser et al. Fig. 2b); x, c updated in place.
// qk: joint q|k buffer [B, N + S, 2D]; vt: joint V^T [B, D, pad(N + S)] with zeroed pad columns (both owned by the caller:
// the same pair serves every block of a network, so the pad is cleared once per evaluation).
// pre_add: x += *pre_add before anything else (folded into the first AdaLN pass).
int pd_engine::sd3_block(const Sd3BlockW& b, Act& x, Act& c, const Act& modbuf, const Act& qk, const Act& vt, const Act* pre_add) {
    const int D = x.C, B = x.B, N = x.H, Sx = b.single ? 0 : c.H, Nt = N + Sx, heads = sd3.heads;
    const bool ctx_stream = !b.single;   // SD3SingleTransformerBlock: the image half alone
    const size_t eb = dt_size(T);
    const float* mod = reinterpret_cast<const float*>(modbuf.p);
    const int ms = modbuf.C, vt_ld = vt.C;
    const size_t mk = arena.mark();
    // option sd3_fp8 (2-byte modes): the AdaLN outputs feeding the QKV and feed-forward-in projections are written as e4m3
    // with one scale per token, and those GEMMs run in PREC_FP8 against the layers' quantised weights
    const bool f8 = opt_sd3_fp8 && !f32, f8b = opt_sd3_fp8 >= 2 && !f32;
    const int NT = f8 ? DT_FP8 : T;
    // level 2: the feed-forward-out projection reads an e4m3 GELU output too.  No kernel sees a whole row of it before it
    // is written, so its row scale is a bound known beforehand: |GELU(W y + b)|_inf <= |y|_2 max_n |W_n|_2 + max |b|
    // (Cauchy-Schwarz; ~10x above the actual row maximum, which costs e4m3 nothing but subnormal range), emitted by the AdaLN
    // pass that produces y.  1.13 covers the e4m3 rounding of y and W (each <= 2^-4 relative).
    constexpr float kBoundMargin = 1.13f;
    auto adaln = [&](const Act& in, Act& out, int shift_off, int scale_off, float* row_scale, const void* add = nullptr,
                     float* bound_out = nullptr, const WMat* consumer = nullptr) -> int {
        if (arena.dry) return 0;
        PD_TRY(check_arena());
        ++launches;
        const float bmul = consumer ? kBoundMargin * consumer->wnorm_max / 448.0f : 0.f, badd = consumer ? consumer->bias_max / 448.0f : 0.f;
        if (launch_adaln(in.p, in.dt, out.p, out.dt, mod, ms, shift_off, scale_off, (int)in.rows(), in.H, D, 1e-6f, stream, row_scale, add,
                         bound_out, bmul, badd)) {
            pd_set_error("sd3: AdaLN launch failed (C=%d)", D);
            return 1;
        }
        return 0;
    };
    auto scales = [&](int rows) { return f8 ? reinterpret_cast<float*>(arena.alloc((size_t)rows * sizeof(float))) : nullptr; };
    Act xn = new_act(B, N, 1, D, NT), cn = new_act(B, Sx, 1, D, NT);
    float *xs = scales(B * N), *cs = ctx_stream ? scales(B * Sx) : nullptr;
    if (pre_add && pre_add->dt != x.dt) { pd_set_error("internal: residual dtype mismatch"); return 1; }
    PD_TRY(adaln(x, xn, b.mod_off, b.mod_off + D, xs, pre_add ? pre_add->p : nullptr));   // (shift_msa, scale_msa, gate_msa, shift_mlp, ...)
    if (!ctx_stream) {}
    else if (b.pre_only) PD_TRY(adaln(c, cn, b.mod_c_off + D, b.mod_c_off, cs));     // AdaLayerNormContinuous: (scale, shift)
    else PD_TRY(adaln(c, cn, b.mod_c_off, b.mod_c_off + D, cs));
    // dual_attention_layers: the second modulated copy of the block INPUT (chunks 6 / 7 of SD35AdaLayerNormZeroX) for attn2
    Act xn2;
    if (b.dual) {
        xn2 = new_act(B, N, 1, D, T);
        PD_TRY(adaln(x, xn2, b.mod_off + 6 * D, b.mod_off + 7 * D, nullptr));
    }
    auto qk_norm = [&](const Act& buf, int rows_per_sample, int n_first, const float* wq, const float* wk, const float* wq2, const float* wk2) -> int {
        if (!sd3.qk_norm || arena.dry) return 0;
        ++launches;
        if (launch_qk_rmsnorm(buf.p, T, (long long)B * rows_per_sample, rows_per_sample, n_first, heads, D / heads, wq, wk, wq2, wk2, 1e-6f, stream)) {
            pd_set_error("sd3: qk RMSNorm launch failed");
            return 1;
        }
        return 0;
    };
    // both QKV GEMMs store straight into the joint buffers
    {
        Act o = qk; o.H = N;
        gx.c_sample_rows = Nt; gx.c_row_off = 0; gx.vt_tok_off = 0; gx.a_scale = xs;
        PD_TRY(gemm(b.qkv, xn, o, 1, 0, 0, 1.f, nullptr, nullptr, 0, false, vt.p, 2 * D, vt_ld));
        if (ctx_stream) {
            o.H = Sx;
            gx.c_sample_rows = Nt; gx.c_row_off = N; gx.vt_tok_off = N; gx.a_scale = cs;
            PD_TRY(gemm(b.qkv_c, cn, o, 1, 0, 0, 1.f, nullptr, nullptr, 0, false, vt.p, 2 * D, vt_ld));
        }
    }
    PD_TRY(qk_norm(qk, Nt, N, b.nq, b.nk, b.naq, b.nak));   // per-head RMSNorm of q and k, own weights per stream
    // ONE attention launch over the joint sequence (image queries only in the context_pre_only block); the out-projections
    // read their token stream out of the joint output through the A-row remap
    const char* qkp = reinterpret_cast<const char*>(qk.p);
    const long long bs = (long long)Nt * 2 * D;
    const int Nq = b.pre_only ? N : Nt;
    Act att = new_act(B, Nt, 1, D, T);
    PD_TRY(attention(qkp, 2 * D, qkp + (size_t)D * eb, 2 * D, vt.p, vt_ld, att.p, D, B, Nq, Nt, D, heads, false, bs, bs, (long long)Nt * D));
    gx.gate = mod + b.mod_off + 2 * D; gx.gate_stride = ms; gx.a_sample_rows = Nt; gx.a_row_off = 0;
    PD_TRY(gemm(b.out, att, x, 1, 0, 0, 1.f, &x, nullptr, 0, false, nullptr, 0, 0));       // x += gate_msa * to_out(o_x)
    if (!b.pre_only && ctx_stream) {
        gx.gate = mod + b.mod_c_off + 2 * D; gx.gate_stride = ms; gx.a_sample_rows = Nt; gx.a_row_off = N;
        PD_TRY(gemm(b.out_c, att, c, 1, 0, 0, 1.f, &c, nullptr, 0, false, nullptr, 0, 0));
    }
    if (b.dual) {   // x += gate_msa2 * attn2(norm_hidden_states2): self-attention over the image tokens alone
        const int vl2 = round_up(N, 8);
        Act qk2 = new_act(B, N, 1, 2 * D, T), vt2 = new_act(B, D, 1, vl2, T), att2 = new_act(B, N, 1, D, T);
        if (!arena.dry && vl2 != N) HIP_OK(hipMemsetAsync(vt2.p, 0, vt2.bytes(), stream));
        PD_TRY(gemm(b.qkv2, xn2, qk2, 1, 0, 0, 1.f, nullptr, nullptr, 0, false, vt2.p, 2 * D, vl2));
        PD_TRY(qk_norm(qk2, N, N, b.nq2, b.nk2, nullptr, nullptr));
        const char* q2p = reinterpret_cast<const char*>(qk2.p);
        PD_TRY(attention(q2p, 2 * D, q2p + (size_t)D * eb, 2 * D, vt2.p, vl2, att2.p, D, B, N, N, D, heads));
        gx.gate = mod + b.mod_off + 8 * D; gx.gate_stride = ms;
        PD_TRY(gemm(b.out2, att2, x, 1, 0, 0, 1.f, &x, nullptr, 0, false, nullptr, 0, 0));
    }
    arena.release(mk);
    // feed-forward of each stream
    {
        Act n2 = new_act(B, N, 1, D, NT);
        float* ns = scales(B * N);
        float* fs = f8b ? scales(B * N) : nullptr;
        PD_TRY(adaln(x, n2, b.mod_off + 3 * D, b.mod_off + 4 * D, ns, nullptr, fs, f8b ? &b.ff1 : nullptr));
        Act f = new_act(B, N, 1, 4 * D, f8b ? DT_FP8 : T);
        gx.a_scale = ns; gx.c_scale = fs;
        PD_TRY(gemm(b.ff1, n2, f, 1, 0, /*tanh-GELU*/ 4, 1.f, nullptr, nullptr, 0, false, nullptr, 0, 0));
        gx.gate = mod + b.mod_off + 5 * D; gx.gate_stride = ms; gx.a_scale = fs;
        PD_TRY(gemm(b.ff2, f, x, 1, 0, 0, 1.f, &x, nullptr, 0, false, nullptr, 0, 0));
        arena.release(mk);
    }
    if (!b.pre_only && ctx_stream) {
        Act n2 = new_act(B, Sx, 1, D, NT);
        float* ns = scales(B * Sx);
        float* fs = f8b ? scales(B * Sx) : nullptr;
        PD_TRY(adaln(c, n2, b.mod_c_off + 3 * D, b.mod_c_off + 4 * D, ns, nullptr, fs, f8b ? &b.ffc1 : nullptr));
        Act f = new_act(B, Sx, 1, 4 * D, f8b ? DT_FP8 : T);
        gx.a_scale = ns; gx.c_scale = fs;
        PD_TRY(gemm(b.ffc1, n2, f, 1, 0, 4, 1.f, nullptr, nullptr, 0, false, nullptr, 0, 0));
        gx.gate = mod + b.mod_c_off + 5 * D; gx.gate_stride = ms; gx.a_scale = fs;
        PD_TRY(gemm(b.ffc2, f, c, 1, 0, 0, 1.f, &c, nullptr, 0, false, nullptr, 0, 0));
        arena.release(mk);
    }
    return 0;
}

// e4m3 copies (one scale per output row) of the layers option sd3_fp8 runs in PREC_FP8
int pd_engine::sd3_quantize() {
    if (!opt_sd3_fp8 || f32 || !sd3_fp8_dirty) return 0;
    float* red = nullptr;   // {max row norm, max |bias|} of one layer
    HIP_OK(hipMalloc(&red, 2 * sizeof(float)));
    int r = 0;
    for (Sd3NetW* net : {&sd3_tr, &sd3_cn}) {
        if (!net->built) continue;
        for (Sd3BlockW& b : net->blocks) {
            const bool l2 = opt_sd3_fp8 >= 2;
            const bool cs = !b.pre_only && !b.single;   // the block has a context feed-forward
            WMat* mats[6] = {&b.qkv, b.single ? nullptr : &b.qkv_c, &b.ff1, cs ? &b.ffc1 : nullptr, l2 ? &b.ff2 : nullptr,
                             (l2 && cs) ? &b.ffc2 : nullptr};
            for (int i = 0; i < 6 && !r; ++i) {
                WMat* m = mats[i];
                if (!m) continue;
                if (!m->w8) {
                    m->Kpad8 = round_up(m->K, 128);
                    m->w8 = dmalloc((size_t)m->N * m->Kpad8);
                    m->wscale = reinterpret_cast<float*>(dmalloc((size_t)(m->N + 4) * sizeof(float)));
                    if (!m->w8 || !m->wscale) { pd_set_error("allocation of fp8 weights failed"); r = 1; break; }
                }
                if (launch_quant_rows(m->w, T, m->Kpad, m->w8, m->Kpad8, m->wscale, m->N, m->K, stream)) {
                    pd_set_error("fp8 weight quantisation launch failed");
                    r = 1;
                    break;
                }
                if (l2 && (i == 2 || i == 3)) {   // feed-forward-in layers: the bound of their outputs (see sd3_block)
                    float host[2] = {0.f, 0.f};
                    if (hipMemsetAsync(red, 0, sizeof(host), stream) != hipSuccess ||
                        launch_rows_norm_max(m->w, T, m->Kpad, m->bias, m->Nout, m->K, red, stream) ||
                        hipMemcpyAsync(host, red, sizeof(host), hipMemcpyDeviceToHost, stream) != hipSuccess ||
                        hipStreamSynchronize(stream) != hipSuccess) {
                        pd_set_error("weight norm reduction failed");
                        r = 1;
                        break;
                    }
                    m->wnorm_max = host[0];
                    m->bias_max = host[1];
                }
            }
        }
    }
    hipFree(red);
    if (!r) sd3_fp8_dirty = false;
    return r;
}

// ControlNet (when io.cond) then transformer.  control_index >= 0: stop after the ControlNet and copy that residual out.
int pd_engine::sd3_forward(const Sd3Io& io, float* v_out, int control_index, float* control_out) {
    const int D = sd3.heads * sd3.head_dim, ps = sd3.patch_size, h = io.H / ps, w = io.W / ps, N = h * w, B = io.B;
    const size_t mk0 = arena.mark();
    std::vector<Act> control;
    // The ControlNet runs on the second stream with its own workspace (like the UNet path's, engine.cpp forward_eps): the
    // transformer needs residual j only after its block ~ j * layers / residuals, so all but the first ControlNet block overlap
    // transformer work -- the two streams fill each other's tails (attention runs 2.2 rounds of blocks) and small launches.
    const bool have_cn = io.cond && (io.scale != 0.f || control_index >= 0);
    const bool two = have_cn && opt_two_streams && stream2 != nullptr && control_index < 0;
    if (have_cn) {
        Sd3NetW& net = sd3_cn;
        for (int i = 0; i < net.layers; ++i) control.push_back(new_act(B, N, 1, D, control_index >= 0 ? DT_F32 : S));
        if (two && !arena.dry) {
            while ((int)sd3_ev.size() < net.layers) {
                hipEvent_t ev = nullptr;
                HIP_OK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
                sd3_ev.push_back(ev);
            }
            HIP_OK(hipEventRecord(ev_fork, stream));
            HIP_OK(hipStreamWaitEvent(stream2, ev_fork, 0));
        }
        if (two) swap_context();
        auto body = [&]() -> int {
            const size_t mk = arena.mark();
            Act hs, c, modbuf, qk, vt;
            PD_TRY(sd3_embed(net, io, true, hs, c, modbuf));
            {
                const int Nt = N + (net.single ? 0 : io.S), vt_ld = round_up(Nt, 8);
                qk = new_act(B, Nt, 1, 2 * D, T);
                vt = new_act(B, D, 1, vt_ld, T);
                if (!arena.dry && vt_ld != Nt) HIP_OK(hipMemsetAsync(vt.p, 0, vt.bytes(), stream));   // pad keys of V^T must read as 0
            }
            for (int i = 0; i < net.layers; ++i) {
                PD_TRY(sd3_block(net.blocks[i], hs, c, modbuf, qk, vt));
                // controlnet_blocks[i](hidden_states) * conditioning_scale   (:469-474)
                Act in = hs;
                if (hs.dt != T) {   // stream_f32: the zero Linear reads 2-byte operands
                    in = new_act(B, N, 1, D, T);
                    if (!arena.dry) {
                        ++launches;
                        if (launch_cast_rows(reinterpret_cast<const float*>(hs.p), in.p, T, (long long)B * N, D, D, stream)) {
                            pd_set_error("sd3: cast launch failed");
                            return 1;
                        }
                    }
                }
                PD_TRY(gemm(net.zero[i], in, control[i], 1, 0, 0, io.scale, nullptr, nullptr, 0, false, nullptr, 0, 0));
                if (two && !arena.dry) HIP_OK(hipEventRecord(sd3_ev[i], stream));   // residual i is ready (`stream` is stream2 here)
            }
            arena.release(mk);
            return 0;
        };
        const int rc = body();
        if (two) swap_context();
        if (rc) return rc;
    }
    if (control_index >= 0) {
        if (control_index >= (int)control.size()) { pd_set_error("sd3: control index %d out of range", control_index); return 1; }
        if (!arena.dry)
            HIP_OK(hipMemcpyAsync(control_out, control[control_index].p, control[control_index].bytes(), hipMemcpyDeviceToDevice, stream));
        arena.release(mk0);
        return 0;
    }
    Sd3NetW& net = sd3_tr;
    Act hs, c, modbuf, qk, vt;
    PD_TRY(sd3_embed(net, io, false, hs, c, modbuf));
    {
        const int Nt = N + io.S, vt_ld = round_up(Nt, 8);
        qk = new_act(B, Nt, 1, 2 * D, T);
        vt = new_act(B, D, 1, vt_ld, T);
        if (!arena.dry && vt_ld != Nt) HIP_OK(hipMemsetAsync(vt.p, 0, vt.bytes(), stream));   // pad keys of V^T must read as 0
    }
    // hidden_states + block_controlnet_hidden_states[int(i / interval_control)] after block i (every block but the
    // context_pre_only last one), interval_control = len(blocks) / len(residuals) as a FLOAT (SD3Transformer2DModel.forward;
    // it is the Flux transformer that takes the ceiling): 18 residuals on 24 blocks map 0,0,1,2,3,3,4,5,...  The add rides
    // on the next block's first AdaLN pass.
    const bool steer = !control.empty();
    auto ctl_index = [&](int i) { return (int)((double)i / ((double)net.layers / (double)control.size())); };
    const Act* pending = nullptr;
    int waited = -1;   // last ControlNet residual this stream has waited for
    for (int i = 0; i < net.layers; ++i) {
        PD_TRY(sd3_block(net.blocks[i], hs, c, modbuf, qk, vt, pending));
        pending = (steer && !net.blocks[i].pre_only) ? &control[ctl_index(i)] : nullptr;
        if (pending && two && !arena.dry && ctl_index(i) > waited) {
            waited = ctl_index(i);
            HIP_OK(hipStreamWaitEvent(stream, sd3_ev[waited], 0));
        }
    }
    // norm_out (AdaLayerNormContinuous: scale, shift) + proj_out + unpatchify
    Act nx = new_act(B, N, 1, D, T);
    Act out = new_act(B, N, 1, net.proj_out.N, DT_F32);
    if (!arena.dry) {
        PD_TRY(check_arena());
        ++launches;
        if (launch_adaln(hs.p, hs.dt, nx.p, T, reinterpret_cast<const float*>(modbuf.p), modbuf.C, net.norm_out_off + D, net.norm_out_off,
                         B * N, N, D, 1e-6f, stream)) {
            pd_set_error("sd3: norm_out launch failed");
            return 1;
        }
    }
    PD_TRY(gemm(net.proj_out, nx, out, 1, 0, 0, 1.f, nullptr, nullptr, 0, false, nullptr, 0, 0));
    if (!arena.dry) {
        ++launches;
        if (launch_unpatchify(out.p, DT_F32, out.C, v_out, B, sd3.out_channels, h, w, ps, stream)) {
            pd_set_error("sd3: unpatchify launch failed");
            return 1;
        }
    }
    arena.release(mk0);
    return 0;
}

// ------------------------------------------------------------------------------------------------ C ABI
extern "C" int pd_sd3_configure(pd_engine* e, const pd_sd3_config* c) {
    if (!e || !c) { pd_set_error("bad argument"); return 1; }
    if (e->sd3_tr.built) { pd_set_error("pd_sd3_configure: already configured"); return 1; }
    if (e->ses.active) { pd_set_error("pd_sd3_configure: end the sampling session first"); return 1; }
    const long long D = (long long)c->heads * c->head_dim;
    if (c->in_channels < 1 || c->out_channels < 1 || c->patch_size < 1 || c->patch_size > 4 || c->heads < 1 || c->layers < 1 ||
        c->cn_layers < 0 || c->joint_dim < 1 || c->pooled_dim < 1 || c->pos_embed_max_size < 1 || c->cn_pos_embed_max_size < 0) {
        pd_set_error("pd_sd3_configure: invalid configuration");
        return 1;
    }
    if (c->head_dim != 8 && c->head_dim != 16 && c->head_dim != 32 && c->head_dim != 64) {
        pd_set_error("pd_sd3_configure: head_dim %d not supported (8, 16, 32, 64)", c->head_dim);
        return 1;
    }
    if (D % 8 || D > 2048) { pd_set_error("pd_sd3_configure: hidden size %lld must be a multiple of 8 and <= 2048", D); return 1; }
    if ((c->cn_single != 0 && c->cn_single != 1) || (c->cn_single && c->cn_dual_mask)) {
        pd_set_error("pd_sd3_configure: cn_single must be 0 or 1, and single blocks have no second attention (cn_dual_mask %u)", c->cn_dual_mask);
        return 1;
    }
    if (c->cn_layers > c->layers) { pd_set_error("pd_sd3_configure: more ControlNet blocks (%d) than transformer blocks (%d)", c->cn_layers, c->layers); return 1; }
    HIP_OK(hipSetDevice(e->device));
    e->sd3 = *c;
    e->alloc_failed = false;
    e->reg_group = 3;
    e->build_sd3_net("transformer.", e->sd3_tr, false);
    if (c->cn_layers > 0) e->build_sd3_net("controlnet.", e->sd3_cn, true);
    e->reg_group = 0;
    if (e->alloc_failed) { pd_set_error("pd_sd3_configure: weight allocation failed"); return 1; }
    return 0;
}

extern "C" int pd_sd3_weights_missing(pd_engine* e) {
    int n = 0;
    if (e)
        for (auto& p : e->params) n += (p.group == 3 && !p.loaded) ? 1 : 0;
    return n;
}

namespace {
// One SD3 call: staged inputs in the workspace, a dry run to size it, the evaluation(s), read-back.
struct Sd3Call {
    pd_engine* e;
    const pd_sd3_args* a;
    int Bf;   // rows of context / pooled / timestep: B, or 2B under guidance
};

int sd3_check(pd_engine* e, const pd_sd3_args* a, bool need_cond) {
    if (!e || !a) { pd_set_error("bad argument"); return 1; }
    if (!e->sd3_tr.built) { pd_set_error("SD3 path not configured (pd_sd3_configure)"); return 1; }
    if (e->ses.active) { pd_set_error("pd_sd3: end the sampling session first"); return 1; }
    for (auto& p : e->params)
        if (p.group == 3 && !p.loaded) { pd_set_error("SD3 weights not loaded: '%s' (and possibly more)", p.name.c_str()); return 1; }
    const int ps = e->sd3.patch_size;
    if (a->batch > 32) { pd_set_error("pd_sd3: batch %d exceeds 32 per call (the guided loop doubles it; shard larger batches)", a->batch); return 1; }
    if (a->batch < 1 || a->height < ps || a->width < ps || a->height % ps || a->width % ps || a->context_len < 1) {
        pd_set_error("pd_sd3: bad shape (batch %d, latent %dx%d, patch %d, context_len %d)", a->batch, a->height, a->width, ps, a->context_len);
        return 1;
    }
    if (a->height / ps > e->sd3.pos_embed_max_size || a->width / ps > e->sd3.pos_embed_max_size ||
        (a->cond && (a->height / ps > e->sd3_cn.pos_max || a->width / ps > e->sd3_cn.pos_max))) {
        pd_set_error("pd_sd3: latent %dx%d exceeds pos_embed_max_size", a->height, a->width);
        return 1;
    }
    if (!a->latents || !a->context || !a->pooled) { pd_set_error("pd_sd3: latents, context and pooled are required"); return 1; }
    if ((a->cond != nullptr) != (a->pair != nullptr)) { pd_set_error("pd_sd3: cond and pair come together"); return 1; }
    if (a->cond && !e->sd3_cn.built) { pd_set_error("pd_sd3: this engine has no SD3 ControlNet (cn_layers = 0)"); return 1; }
    if (need_cond && !a->cond) { pd_set_error("pd_sd3_control: cond and pair are required"); return 1; }
    return 0;
}

// steps < 0: single evaluation (v_out or control residual); otherwise the Euler loop
int sd3_run(pd_engine* e, const pd_sd3_args* a, const float* sigmas, int steps, float guidance, const float* step_scales,
            int control_index, float* out) {
    HIP_OK(hipSetDevice(e->device));
    PD_TRY(e->sd3_quantize());
    const bool loop = steps >= 0;
    const bool cfg = loop && guidance > 1.0f;
    const int B = a->batch, Bf = cfg ? 2 * B : B;
    const int C = e->sd3.in_channels, Co = e->sd3.out_channels, H = a->height, W = a->width, S = a->context_len;
    const int ps = e->sd3.patch_size, D = e->sd3.heads * e->sd3.head_dim;
    if (loop && C != Co) { pd_set_error("pd_sd3_sample: in_channels != out_channels"); return 1; }
    const size_t n_lat = (size_t)B * C * H * W, n_ctx = (size_t)Bf * S * e->sd3.joint_dim, n_pool = (size_t)Bf * e->sd3.pooled_dim;
    const size_t n_v = (size_t)Bf * Co * H * W;
    const size_t n_out = control_index >= 0 ? (size_t)B * (H / ps) * (W / ps) * D : (loop ? n_lat : n_v);
    const hipMemcpyKind kin = a->mem == PD_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    const hipMemcpyKind kout = a->mem == PD_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
    // the sampling path's main workspace, sized by a dry run of the same call
    pd_engine::Sd3Io io{};
    io.B = Bf; io.H = H; io.W = W; io.S = S; io.scale = a->conditioning_scale;
    std::vector<float> t_host(Bf, 0.f);
    io.t_host = t_host.data();
    io.cond = a->cond; io.pair = a->pair;   // non-null markers for the dry run
    io.scale = 1.f;                         // ... which always counts the ControlNet (a per-step scale may switch it on)
    Arena saved = e->arena, saved2 = e->arena2;
    e->arena.base = nullptr; e->arena.cap = 0; e->arena.top = 0; e->arena.peak = 0; e->arena.dry = true; e->arena.overflow = false;
    e->arena2 = e->arena;
    int r = e->sd3_forward(io, nullptr, control_index, nullptr);
    const size_t staged = (n_lat * 8 + n_ctx + 2 * n_pool + n_v + n_out) * sizeof(float) + 16 * 256;
    const size_t need = e->arena.peak + staged + (64u << 20);
    const size_t need2 = e->arena2.peak ? e->arena2.peak + (16u << 20) : 0;   // the ControlNet stream's workspace
    e->arena = saved;
    e->arena2 = saved2;
    e->arena.dry = false;
    if (r) return r;
    if (need2 > e->arena2.cap) {
        hipStreamSynchronize(e->stream);
        if (e->stream2) hipStreamSynchronize(e->stream2);
        e->clear_graphs();
        if (e->arena2.base) hipFree(e->arena2.base);
        e->arena2.base = nullptr; e->arena2.cap = 0;
        void* p2 = nullptr;
        if (hipMalloc(&p2, need2) != hipSuccess) { pd_set_error("SD3 ControlNet workspace allocation of %.2f GiB failed", (double)need2 / (1 << 30)); return 1; }
        e->arena2.base = reinterpret_cast<char*>(p2); e->arena2.cap = need2;
    }
    e->arena2.top = 0; e->arena2.peak = 0; e->arena2.overflow = false;
    if (need > e->arena.cap) {
        hipStreamSynchronize(e->stream);
        if (e->stream2) hipStreamSynchronize(e->stream2);
        e->clear_graphs();
        if (e->arena.base) hipFree(e->arena.base);
        e->arena.base = nullptr; e->arena.cap = 0;
        void* p = nullptr;
        if (hipMalloc(&p, need) != hipSuccess) { pd_set_error("SD3 workspace allocation of %.2f GiB failed", (double)need / (1 << 30)); return 1; }
        e->arena.base = reinterpret_cast<char*>(p); e->arena.cap = need;
    }
    e->arena.top = 0; e->arena.peak = 0; e->arena.overflow = false;
    auto falloc = [&](size_t n) { return reinterpret_cast<float*>(e->arena.alloc(n * sizeof(float))); };
    float* x = falloc(n_lat);                       // latents (state of the loop)
    float* xin = cfg ? falloc(2 * n_lat) : x;       // doubled batch
    float* ctx = falloc(n_ctx);
    float* pool = falloc(n_pool);
    // the ControlNet's pooled projections: zeros (force_zeros_for_pooled_projection), the caller's, or the transformer's
    float* cn_pool = nullptr;
    // (pd_sd3_control is the model-level call: the ControlNet gets what the caller hands it, like SD3PromptDiffusionModel.forward)
    if (a->cond && (!e->sd3.cn_zero_pooled || control_index >= 0)) cn_pool = a->cn_pooled ? falloc(n_pool) : pool;
    float* cond = a->cond ? falloc(cfg ? 2 * n_lat : n_lat) : nullptr;
    float* pair = a->cond ? falloc(cfg ? 2 * n_lat : n_lat) : nullptr;
    float* v = falloc(n_v);
    float* res = control_index >= 0 ? falloc(n_out) : nullptr;
    hipStream_t st = e->stream;
    HIP_OK(hipMemcpyAsync(x, a->latents, n_lat * 4, kin, st));
    HIP_OK(hipMemcpyAsync(ctx, a->context, n_ctx * 4, kin, st));
    HIP_OK(hipMemcpyAsync(pool, a->pooled, n_pool * 4, kin, st));
    if (cn_pool && cn_pool != pool) HIP_OK(hipMemcpyAsync(cn_pool, a->cn_pooled, n_pool * 4, kin, st));
    if (a->cond) {
        for (int d = 0; d < (cfg ? 2 : 1); ++d) {
            HIP_OK(hipMemcpyAsync(cond + d * n_lat, a->cond, n_lat * 4, kin, st));
            HIP_OK(hipMemcpyAsync(pair + d * n_lat, a->pair, n_lat * 4, kin, st));
        }
    }
    io.scale = a->conditioning_scale;
    io.latents = xin; io.context = ctx; io.pooled = pool; io.cn_pooled = cn_pool; io.cond = cond; io.pair = pair;
    if (!loop) {
        for (int b = 0; b < Bf; ++b) t_host[b] = a->timestep ? a->timestep[b] : 0.f;
        r = e->sd3_forward(io, v, control_index, res);
        if (!r) {
            HIP_OK(hipMemcpyAsync(out, control_index >= 0 ? res : v, n_out * 4, kout, st));
            HIP_OK(hipStreamSynchronize(st));
        }
    } else {
        for (int i = 0; i < steps && !r; ++i) {
            if (cfg) {
                HIP_OK(hipMemcpyAsync(xin, x, n_lat * 4, hipMemcpyDeviceToDevice, st));
                HIP_OK(hipMemcpyAsync(xin + n_lat, x, n_lat * 4, hipMemcpyDeviceToDevice, st));
            }
            for (int b = 0; b < Bf; ++b) t_host[b] = sigmas[i] * 1000.0f;   // timestep = sigma * num_train_timesteps
            if (step_scales) io.scale = step_scales[i];
            r = e->sd3_forward(io, v, -1, nullptr);
            if (r) break;
            ++e->launches;
            if (launch_cfg_euler(v, x, B, (long long)n_lat, guidance, sigmas[i + 1] - sigmas[i], cfg ? 1 : 0, st)) {
                pd_set_error("sd3: Euler step launch failed");
                r = 1;
            }
        }
        if (!r) {
            HIP_OK(hipMemcpyAsync(out, x, n_lat * 4, kout, st));
            HIP_OK(hipStreamSynchronize(st));
        }
    }
    if (!r) r = e->check_arena();
    if (e->stream2 && hipStreamSynchronize(e->stream2) != hipSuccess && !r) { pd_set_error("second stream failed"); r = 1; }
    if (e->arena2.overflow && !r) { pd_set_error("SD3 ControlNet workspace overflow"); r = 1; }
    e->arena.top = 0;
    e->arena2.top = 0;
    return r;
}
}  // namespace

extern "C" int pd_sd3_forward(pd_engine* e, const pd_sd3_args* a, float* v_out) {
    PD_TRY(sd3_check(e, a, false));
    if (!v_out || !a->timestep) { pd_set_error("pd_sd3_forward: v_out and timestep are required"); return 1; }
    return sd3_run(e, a, nullptr, -1, 0.f, nullptr, -1, v_out);
}

extern "C" int pd_sd3_control(pd_engine* e, const pd_sd3_args* a, int32_t index, float* out) {
    PD_TRY(sd3_check(e, a, true));
    if (!out || !a->timestep || index < 0 || index >= e->sd3.cn_layers) { pd_set_error("pd_sd3_control: bad argument"); return 1; }
    return sd3_run(e, a, nullptr, -1, 0.f, nullptr, index, out);
}

extern "C" int pd_sd3_sample(pd_engine* e, const pd_sd3_args* a, const float* sigmas, int32_t steps, float guidance,
                             const float* step_scales, float* latents_out) {
    PD_TRY(sd3_check(e, a, false));
    if (!sigmas || steps < 1 || !latents_out) { pd_set_error("pd_sd3_sample: sigmas, steps >= 1 and latents_out are required"); return 1; }
    return sd3_run(e, a, sigmas, steps, guidance, step_scales, -1, latents_out);
}

// SD3PromptDiffusionModel.down_proj, promptdiffusioncontrolnet_sd3.py:114,189-194: Conv2d(6, 3, 3, padding=1) on the example pair
extern "C" int pd_sd3_down_proj(pd_engine* e, const float* pair, int32_t B, int32_t H, int32_t W, int32_t mem, float* out) {
    if (!e || !pair || !out || B < 1 || H < 1 || W < 1) { pd_set_error("pd_sd3_down_proj: bad argument"); return 1; }
    if (!e->sd3_cn.built) { pd_set_error("pd_sd3_down_proj: this engine has no SD3 ControlNet"); return 1; }
    for (auto& p : e->params)
        if (p.group == 3 && !p.loaded && p.name.find("down_proj") != std::string::npos) { pd_set_error("weights not loaded: '%s'", p.name.c_str()); return 1; }
    HIP_OK(hipSetDevice(e->device));
    const ConvW& c = e->sd3_cn.down_proj;
    const size_t n_in = (size_t)B * 6 * H * W, n_out = (size_t)B * 3 * H * W, px = (size_t)B * H * W;
    const int T = e->T;
    float *din = nullptr, *dout = nullptr;
    void *xin = nullptr, *y = nullptr;
    int r = 1;
    do {
        if (hipMalloc(&din, n_in * 4) != hipSuccess || hipMalloc(&dout, n_out * 4) != hipSuccess ||
            hipMalloc(&xin, px * 8 * dt_size(T)) != hipSuccess || hipMalloc(&y, px * 4 * 4) != hipSuccess) { pd_set_error("pd_sd3_down_proj: allocation failed"); break; }
        if (hipMemcpyAsync(din, pair, n_in * 4, mem == PD_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, e->stream) != hipSuccess) break;
        if (launch_nchw_to_nhwc(din, xin, T, B, 6, H, W, 8, e->stream)) break;
        Act a, o;
        a.p = xin; a.B = B; a.H = H; a.W = W; a.C = 8; a.dt = T;
        o.p = y; o.B = B; o.H = H; o.W = W; o.C = 4; o.dt = DT_F32;
        const Arena saved = e->arena;
        e->arena = Arena{};     // no split-K workspace: the conv runs as one plain launch
        const int rc = e->conv(c, a, o);
        e->arena = saved;
        if (rc) break;
        if (launch_nhwc_to_nchw(y, DT_F32, dout, B, 3, H, W, 4, 1.f, e->stream)) break;
        if (hipMemcpyAsync(out, dout, n_out * 4, mem == PD_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, e->stream) != hipSuccess) break;
        if (hipStreamSynchronize(e->stream) != hipSuccess) break;
        r = 0;
    } while (0);
    if (r) pd_set_error("pd_sd3_down_proj failed: %s", hipGetErrorString(hipGetLastError()));
    hipFree(din); hipFree(dout); hipFree(xin); hipFree(y);
    return r;
}
