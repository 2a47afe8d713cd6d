"""CPU restatement (NumPy fp32) of the SD3 hot path of the reference: SD3PromptDiffusionModel.forward
(/root/reference/promptdiffusioncontrolnet_sd3.py:362-483) + the MMDiT it steers + the flow-matching Euler loop of
promptdiffusioncontrolnetpipeline_sd3.py:1192-1245.  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED.  The block arithmetic (SD3Transformer2DModel, JointTransformerBlock, AdaLayerNormZero / Continuous,
PatchEmbed, CombinedTimestepTextProjEmbeddings, FlowMatchEulerDiscreteScheduler) lives in diffusers >= 0.33.0.dev0, which is
neither vendored in the reference tree nor installed here, and no SD3 weights or fixtures exist offline.  What follows
restates the published MMDiT (Esser et al. 2024, "Scaling Rectified Flow Transformers for High-Resolution Image Synthesis",
Fig. 2 / Sec. 4) under the reference's own call sites and parameter names:
  * controlnet forward  promptdiffusioncontrolnet_sd3.py:431-476: pos_embed(latents) + pos_embed_input(cond) +
    pos_embed_input(example pair), joint blocks, one zero Linear per block, * conditioning_scale
  * transformer forward consumed as self.transformer(hidden_states, timestep, encoder_hidden_states, pooled_projections,
    block_controlnet_hidden_states) (pipeline :1226-1234); residual i is added after block i with
    interval_control = len(blocks) / len(residuals) (a float; index int(i / interval_control))
  * CFG uncond + s (text - uncond) (:1237-1239) and latents + (sigma_next - sigma) * v (scheduler.step, :1243)
The engine path (csrc/sd3.cpp) is tested against THIS file only (tests/test_sd3_gpu.py)."""
import math

import numpy as np

F32 = np.float32


def linear(x, w, b=None):
    y = x @ w.T
    return y if b is None else y + b


def quant_e4m3(x):
    """Round-to-nearest-even onto the OCP e4m3fn grid (3 mantissa bits, normal range 2^-6 .. 448, subnormal step 2^-9),
    saturating at +-448: what v_cvt_pk_fp8_f32 does to a clamped input."""
    x = np.clip(np.asarray(x, np.float64), -448.0, 448.0)
    e = np.floor(np.log2(np.maximum(np.abs(x), 2.0 ** -6)))
    ulp = 2.0 ** (e - 3)
    return (np.round(x / ulp) * ulp).astype(F32)


def linear_fp8(x, w, b=None):
    """The engine's PREC_FP8 linear layer: e4m3 operands with one scale per activation row and per weight row
    (max |row| / 448), exact products, fp32 accumulation; scales applied to the sum."""
    def rows(t):
        amax = np.abs(t).max(-1, keepdims=True)
        sc = np.where(amax > 0, amax * F32(1.0 / 448.0), F32(1.0)).astype(F32)
        return quant_e4m3(t * (F32(1.0) / sc)), sc
    xq, xs = rows(x)
    wq, ws = rows(w)
    y = (xq @ wq.T) * xs * ws[:, 0]
    return (y if b is None else y + b).astype(F32)


def ff_fp8_bounded(xn, w1, b1, w2, b2):
    """The engine's sd3_fp8 level 2 feed-forward: net.0 as linear_fp8, its GELU output stored as e4m3 under the row scale
    (1.13 |xn_row|_2 max_n |W1_n|_2 + max |b1|) / 448 (a Cauchy-Schwarz bound, known before the GEMM runs), net.2 on e4m3
    operands with that scale and per-output-channel weight scales."""
    h = gelu_tanh(linear_fp8(xn, w1, b1))
    wn = F32(np.sqrt((w1.astype(np.float64) ** 2).sum(-1)).max())
    bm = F32(np.abs(b1).max())
    rn = np.sqrt((xn.astype(np.float64) ** 2).sum(-1, keepdims=True)).astype(F32)
    sc = (rn * (F32(1.13) * wn / F32(448.0)) + bm / F32(448.0)).astype(F32)
    hq = quant_e4m3(h * (F32(1.0) / sc))
    amax = np.abs(w2).max(-1, keepdims=True)
    ws = np.where(amax > 0, amax * F32(1.0 / 448.0), F32(1.0)).astype(F32)
    wq = quant_e4m3(w2 * (F32(1.0) / ws))
    return ((hq @ wq.T) * sc * ws[:, 0] + b2).astype(F32)


def silu(x):
    return x / (1.0 + np.exp(-x))


def gelu_tanh(x):
    return (0.5 * x * (1.0 + np.tanh(F32(math.sqrt(2.0 / math.pi)) * (x + F32(0.044715) * x * x * x)))).astype(F32)


def layer_norm_noaffine(x, eps=1e-6):
    mu = x.mean(-1, keepdims=True)
    var = ((x - mu) ** 2).mean(-1, keepdims=True)
    return ((x - mu) / np.sqrt(var + F32(eps))).astype(F32)


def timestep_embedding(t, dim=256, max_period=10000.0):
    """Timesteps(num_channels=256, flip_sin_to_cos=True, downscale_freq_shift=0): [cos, sin]."""
    half = dim // 2
    freqs = np.exp(-math.log(max_period) * np.arange(half, dtype=F32) / F32(half)).astype(F32)
    a = np.asarray(t, F32)[:, None] * freqs[None]
    return np.concatenate([np.cos(a), np.sin(a)], axis=-1).astype(F32)


def sincos_pos_embed(dim, grid, base_size, interpolation_scale=1.0):
    """2-D sin/cos table [grid*grid, dim] (get_2d_sincos_pos_embed: w-axis first half, h-axis second half)."""
    def one(pos, d):
        omega = 1.0 / 10000 ** (np.arange(d // 2, dtype=np.float64) / (d / 2.0))
        out = pos.reshape(-1)[:, None] * omega[None]
        return np.concatenate([np.sin(out), np.cos(out)], axis=1)
    gh = np.arange(grid, dtype=np.float64) / (grid / base_size) / interpolation_scale
    gw = np.arange(grid, dtype=np.float64) / (grid / base_size) / interpolation_scale
    mw, mh = np.meshgrid(gw, gh)             # w varies fastest
    return np.concatenate([one(mw, dim // 2), one(mh, dim // 2)], axis=1).astype(F32)


def cropped_pos_embed(table, max_size, h, w):
    """PatchEmbed.cropped_pos_embed: centre crop of the [max, max, D] table to [h, w]."""
    D = table.shape[-1]
    t = table.reshape(max_size, max_size, D)
    top, left = (max_size - h) // 2, (max_size - w) // 2
    return t[top:top + h, left:left + w].reshape(h * w, D)


def patch_embed(x, w, b, p=2):
    """Conv2d(C, D, kernel p, stride p) then flatten(2).transpose(1, 2): [B, C, H, W] -> [B, (H/p)(W/p), D]."""
    B, C, H, W = x.shape
    xr = x.reshape(B, C, H // p, p, W // p, p).transpose(0, 2, 4, 1, 3, 5).reshape(B, (H // p) * (W // p), C * p * p)
    return linear(xr, w.reshape(w.shape[0], -1), b).astype(F32)


def attention(q, k, v, heads):
    B, Nq, D = q.shape
    dh = D // heads
    sp = lambda t: t.reshape(B, t.shape[1], heads, dh).transpose(0, 2, 1, 3)
    s = np.einsum("bhid,bhjd->bhij", sp(q), sp(k)) * F32(dh ** -0.5)
    s = s - s.max(-1, keepdims=True)
    e = np.exp(s)
    a = e / e.sum(-1, keepdims=True)
    return np.einsum("bhij,bhjd->bhid", a, sp(v)).transpose(0, 2, 1, 3).reshape(B, Nq, D).astype(F32)


def rms_norm_heads(t, w, heads, eps=1e-6):
    """RMSNorm(dim_head, eps=1e-6, elementwise_affine=True) on every head of a [B, N, heads * dim_head] projection (qk_norm="rms_norm")."""
    B, N, D = t.shape
    x = t.reshape(B, N, heads, D // heads)
    y = x / np.sqrt((x * x).mean(-1, keepdims=True) + F32(eps)) * w
    return y.reshape(B, N, D).astype(F32)


def joint_block(sd, pre, cfg, x, c, temb, context_pre_only, fp8=False, dual=False):
    """JointTransformerBlock: AdaLN-Zero on both streams, attention over [image ; context] tokens, gated residuals.
    fp8: the engine's sd3_fp8 option (0 / False, 1, 2) -- the projections fed by an AdaLN output (q/k/v of both streams,
    ff / ff_context net.0) take e4m3 operands (linear_fp8); level 2 also the feed-forward-out projections (ff_fp8_bounded)."""
    P = lambda n: sd[pre + n]
    fp8 = int(fp8)
    qlinear = linear_fp8 if fp8 else linear
    e = silu(temb)
    m = linear(e, P("norm1.linear.weight"), P("norm1.linear.bias"))
    if dual:    # SD35AdaLayerNormZeroX: 9 chunks, the last three modulate the image-only second attention (attn2)
        sh_a, sc_a, g_a, sh_m, sc_m, g_m, sh_a2, sc_a2, g_a2 = np.split(m, 9, axis=-1)
        xn_2 = layer_norm_noaffine(x) * (1 + sc_a2[:, None]) + sh_a2[:, None]
    else:
        sh_a, sc_a, g_a, sh_m, sc_m, g_m = np.split(m, 6, axis=-1)
    xn = layer_norm_noaffine(x) * (1 + sc_a[:, None]) + sh_a[:, None]
    qkn = getattr(cfg, "qk_norm", None) == "rms_norm"
    mc = linear(e, P("norm1_context.linear.weight"), P("norm1_context.linear.bias"))
    if context_pre_only:      # AdaLayerNormContinuous: chunk order (scale, shift)
        c_sc, c_sh = np.split(mc, 2, axis=-1)
        cn = layer_norm_noaffine(c) * (1 + c_sc[:, None]) + c_sh[:, None]
    else:
        c_sh_a, c_sc_a, c_g_a, c_sh_m, c_sc_m, c_g_m = np.split(mc, 6, axis=-1)
        cn = layer_norm_noaffine(c) * (1 + c_sc_a[:, None]) + c_sh_a[:, None]
    N = x.shape[1]
    qx, qc = qlinear(xn, P("attn.to_q.weight"), P("attn.to_q.bias")), qlinear(cn, P("attn.add_q_proj.weight"), P("attn.add_q_proj.bias"))
    kx, kc = qlinear(xn, P("attn.to_k.weight"), P("attn.to_k.bias")), qlinear(cn, P("attn.add_k_proj.weight"), P("attn.add_k_proj.bias"))
    if qkn:     # per-head RMSNorm of queries and keys, own weights per stream
        qx, kx = rms_norm_heads(qx, P("attn.norm_q.weight"), cfg.heads), rms_norm_heads(kx, P("attn.norm_k.weight"), cfg.heads)
        qc, kc = rms_norm_heads(qc, P("attn.norm_added_q.weight"), cfg.heads), rms_norm_heads(kc, P("attn.norm_added_k.weight"), cfg.heads)
    q, k = np.concatenate([qx, qc], axis=1), np.concatenate([kx, kc], axis=1)
    v = np.concatenate([qlinear(xn, P("attn.to_v.weight"), P("attn.to_v.bias")),
                        qlinear(cn, P("attn.add_v_proj.weight"), P("attn.add_v_proj.bias"))], axis=1)
    o = attention(q, k, v, cfg.heads)
    ox = linear(o[:, :N], P("attn.to_out.0.weight"), P("attn.to_out.0.bias"))
    x = x + g_a[:, None] * ox
    if dual:    # attn2: self-attention over the image tokens alone, on the second modulated copy of the block input
        q2, k2 = linear(xn_2, P("attn2.to_q.weight"), P("attn2.to_q.bias")), linear(xn_2, P("attn2.to_k.weight"), P("attn2.to_k.bias"))
        if qkn:
            q2, k2 = rms_norm_heads(q2, P("attn2.norm_q.weight"), cfg.heads), rms_norm_heads(k2, P("attn2.norm_k.weight"), cfg.heads)
        o2 = attention(q2, k2, linear(xn_2, P("attn2.to_v.weight"), P("attn2.to_v.bias")), cfg.heads)
        x = x + g_a2[:, None] * linear(o2, P("attn2.to_out.0.weight"), P("attn2.to_out.0.bias"))
    xn2 = layer_norm_noaffine(x) * (1 + sc_m[:, None]) + sh_m[:, None]
    if fp8 >= 2:
        ff = ff_fp8_bounded(xn2, P("ff.net.0.proj.weight"), P("ff.net.0.proj.bias"), P("ff.net.2.weight"), P("ff.net.2.bias"))
    else:
        ff = linear(gelu_tanh(qlinear(xn2, P("ff.net.0.proj.weight"), P("ff.net.0.proj.bias"))), P("ff.net.2.weight"), P("ff.net.2.bias"))
    x = (x + g_m[:, None] * ff).astype(F32)
    if context_pre_only:
        return None, x
    oc = linear(o[:, N:], P("attn.to_add_out.weight"), P("attn.to_add_out.bias"))
    c = c + c_g_a[:, None] * oc
    cn2 = layer_norm_noaffine(c) * (1 + c_sc_m[:, None]) + c_sh_m[:, None]
    if fp8 >= 2:
        ffc = ff_fp8_bounded(cn2, P("ff_context.net.0.proj.weight"), P("ff_context.net.0.proj.bias"),
                             P("ff_context.net.2.weight"), P("ff_context.net.2.bias"))
    else:
        ffc = linear(gelu_tanh(qlinear(cn2, P("ff_context.net.0.proj.weight"), P("ff_context.net.0.proj.bias"))),
                     P("ff_context.net.2.weight"), P("ff_context.net.2.bias"))
    c = (c + c_g_m[:, None] * ffc).astype(F32)
    return c, x


def single_block(sd, pre, cfg, x, temb):
    """SD3SingleTransformerBlock (the ControlNet's blocks when joint_attention_dim is None, promptdiffusioncontrolnet_sd3.py:147-160):
    the image half of the joint block -- AdaLN-Zero, self-attention over the image tokens, gated residual, LayerNorm + tanh-GELU MLP."""
    P = lambda n: sd[pre + n]
    e = silu(temb)
    sh_a, sc_a, g_a, sh_m, sc_m, g_m = np.split(linear(e, P("norm1.linear.weight"), P("norm1.linear.bias")), 6, axis=-1)
    xn = layer_norm_noaffine(x) * (1 + sc_a[:, None]) + sh_a[:, None]
    q, k = linear(xn, P("attn.to_q.weight"), P("attn.to_q.bias")), linear(xn, P("attn.to_k.weight"), P("attn.to_k.bias"))
    if getattr(cfg, "qk_norm", None) == "rms_norm":
        q, k = rms_norm_heads(q, P("attn.norm_q.weight"), cfg.heads), rms_norm_heads(k, P("attn.norm_k.weight"), cfg.heads)
    o = attention(q, k, linear(xn, P("attn.to_v.weight"), P("attn.to_v.bias")), cfg.heads)
    x = x + g_a[:, None] * linear(o, P("attn.to_out.0.weight"), P("attn.to_out.0.bias"))
    xn2 = layer_norm_noaffine(x) * (1 + sc_m[:, None]) + sh_m[:, None]
    ff = linear(gelu_tanh(linear(xn2, P("ff.net.0.proj.weight"), P("ff.net.0.proj.bias"))), P("ff.net.2.weight"), P("ff.net.2.bias"))
    return (x + g_m[:, None] * ff).astype(F32)


def time_text_embed(sd, pre, t, pooled):
    """CombinedTimestepTextProjEmbeddings: MLP(sinusoid(t)) + MLP(pooled)."""
    P = lambda n: sd[pre + n]
    te = linear(silu(linear(timestep_embedding(t), P("timestep_embedder.linear_1.weight"), P("timestep_embedder.linear_1.bias"))),
                P("timestep_embedder.linear_2.weight"), P("timestep_embedder.linear_2.bias"))
    pe = linear(silu(linear(pooled, P("text_embedder.linear_1.weight"), P("text_embedder.linear_1.bias"))),
                P("text_embedder.linear_2.weight"), P("text_embedder.linear_2.bias"))
    return (te + pe).astype(F32)


def controlnet_forward(sd, cfg, x, t, ctx, pooled, cond, pair, scale=1.0, prefix="controlnet.", fp8=False):
    """SD3PromptDiffusionModel.forward (promptdiffusioncontrolnet_sd3.py:431-476): list of cfg.cn_layers residuals."""
    P = lambda n: sd[prefix + n]
    B, C, H, W = x.shape
    h, w = H // cfg.patch, W // cfg.patch
    hs = patch_embed(x, P("pos_embed.proj.weight"), P("pos_embed.proj.bias"), cfg.patch)
    hs = hs + cropped_pos_embed(P("pos_embed.pos_embed")[0], cfg.cn_pos_embed_max_size or cfg.pos_embed_max_size, h, w)[None]
    temb = time_text_embed(sd, prefix + "time_text_embed.", t, pooled)
    single = bool(getattr(cfg, "cn_single_blocks", False))      # joint_attention_dim=None: no context stream at all (:147-160, :426-431)
    c = None if single else linear(ctx, P("context_embedder.weight"), P("context_embedder.bias"))
    pi_w, pi_b = P("pos_embed_input.proj.weight"), P("pos_embed_input.proj.bias")
    hs = (hs + patch_embed(cond, pi_w, pi_b, cfg.patch) + patch_embed(pair, pi_w, pi_b, cfg.patch)).astype(F32)   # :440
    res = []
    for i in range(cfg.cn_layers):
        if single:
            hs = single_block(sd, f"{prefix}transformer_blocks.{i}.", cfg, hs, temb)
        else:
            c, hs = joint_block(sd, f"{prefix}transformer_blocks.{i}.", cfg, hs, c, temb, False, fp8,
                                dual=i in tuple(getattr(cfg, "cn_dual_attention_layers", ())))
        res.append(hs)
    return [(linear(r, P(f"controlnet_blocks.{i}.weight"), P(f"controlnet_blocks.{i}.bias")) * F32(scale)).astype(F32)
            for i, r in enumerate(res)]


def transformer_forward(sd, cfg, x, t, ctx, pooled, control=None, prefix="transformer.", fp8=False):
    """SD3Transformer2DModel.forward as the pipeline calls it (:1226-1234): velocity [B, C, H, W]."""
    P = lambda n: sd[prefix + n]
    B, C, H, W = x.shape
    h, w = H // cfg.patch, W // cfg.patch
    hs = patch_embed(x, P("pos_embed.proj.weight"), P("pos_embed.proj.bias"), cfg.patch)
    hs = (hs + cropped_pos_embed(P("pos_embed.pos_embed")[0], cfg.pos_embed_max_size, h, w)[None]).astype(F32)
    temb = time_text_embed(sd, prefix + "time_text_embed.", t, pooled)
    c = linear(ctx, P("context_embedder.weight"), P("context_embedder.bias"))
    interval_control = cfg.layers / len(control) if control else 0.0     # float division, as SD3Transformer2DModel.forward does
    for i in range(cfg.layers):
        last = i == cfg.layers - 1
        c, hs = joint_block(sd, f"{prefix}transformer_blocks.{i}.", cfg, hs, c, temb, last, fp8,
                            dual=i in tuple(getattr(cfg, "dual_attention_layers", ())))
        if control and not last:
            hs = (hs + control[int(i / interval_control)]).astype(F32)
    m = linear(silu(temb), P("norm_out.linear.weight"), P("norm_out.linear.bias"))
    sc, sh = np.split(m, 2, axis=-1)
    hs = layer_norm_noaffine(hs) * (1 + sc[:, None]) + sh[:, None]
    out = linear(hs, P("proj_out.weight"), P("proj_out.bias"))            # [B, h*w, p*p*Cout]
    p, Co = cfg.patch, cfg.out_channels
    out = out.reshape(B, h, w, p, p, Co).transpose(0, 5, 1, 3, 2, 4).reshape(B, Co, h * p, w * p)   # nhwpqc -> nchpwq
    return out.astype(F32)


def down_proj(sd, pair, prefix="controlnet."):
    """encode_support_pair without a VAE (promptdiffusioncontrolnet_sd3.py:189-198): Conv2d(6, 3, 3, padding=1) over cat([cond, gt], 1)."""
    w, b = sd[prefix + "down_proj.weight"], sd[prefix + "down_proj.bias"]
    B, C, H, W = pair.shape
    xp = np.pad(pair, ((0, 0), (0, 0), (1, 1), (1, 1)))
    out = np.zeros((B, w.shape[0], H, W), F32)
    for ky in range(3):
        for kx in range(3):
            out += np.einsum("bchw,oc->bohw", xp[:, :, ky:ky + H, kx:kx + W], w[:, :, ky, kx])
    return (out + b[None, :, None, None]).astype(F32)


def flow_match_sigmas(steps, shift=3.0, num_train=1000):
    """FlowMatchEulerDiscreteScheduler.set_timesteps: sigmas from sigma_max to sigma_min, shifted, then 0."""
    smax, smin = 1.0, 1.0 / num_train
    shifted = lambda s: shift * s / (1 + (shift - 1) * s)
    ts = np.linspace(shifted(smax) * num_train, shifted(smin) * num_train, steps)
    s = ts / num_train
    s = shifted(s)
    return np.concatenate([s, [0.0]]).astype(F32)


def sample(sd, cfg, latents, ctx, ctx_neg, pooled, pooled_neg, cond, pair, steps, guidance, scale=1.0, shift=3.0,
           guidance_start=0.0, guidance_end=1.0, cn_pooled=None):
    """pipeline :1155-1168, :1192-1245: CFG-doubled batch ([negative, positive]), ControlNet (zero pooled projections under
    force_zeros_for_pooled_projection, scale * controlnet_keep[i]) then transformer, Euler step."""
    sig = flow_match_sigmas(steps, shift)
    x = latents
    B = x.shape[0]
    for i in range(steps):
        t = np.full((2 * B,), sig[i] * 1000.0, F32)
        xi = np.concatenate([x, x])
        cc, pp = np.concatenate([ctx_neg, ctx]), np.concatenate([pooled_neg, pooled])
        keep = 1.0 - float(i / steps < guidance_start or (i + 1) / steps > guidance_end)
        cp = np.zeros_like(pp) if cfg.force_zeros_for_pooled_projection else (pp if cn_pooled is None else cn_pooled)
        ctl = controlnet_forward(sd, cfg, xi, t, cc, cp, np.concatenate([cond, cond]), np.concatenate([pair, pair]), scale * keep)
        v = transformer_forward(sd, cfg, xi, t, cc, pp, ctl)
        v = v[:B] + F32(guidance) * (v[B:] - v[:B])
        x = (x + (sig[i + 1] - sig[i]) * v).astype(F32)
    return x
