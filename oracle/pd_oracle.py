"""CPU oracle for the Prompt-Diffusion DDIM hot path — TEST INFRASTRUCTURE ONLY.

A NumPy (fp32, NCHW) restatement of the reference's (L) path, function by function,
each citing the reference file:line it follows.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module; the product (``prompt-diffusion_amd/``) never does and fails loudly when the
HIP library is missing.

Pinning: the reference has no tests or golden vectors of its own, so
this oracle is pinned against outputs of the reference itself, produced in the build
container by ``tests/golden/make_golden.py`` (which imports the reference's unmodified
``cldm.cldm`` / ``cldm.ddim_hacked`` modules) and committed under ``tests/golden/``;
``tests/test_oracle_golden.py`` replays them.

Weights are a dict {reference state-dict name: ndarray} (``model.diffusion_model.*``,
``control_model.*``); the network topology is derived from the same hyper-parameters
as ``models/cldm_v15.yaml:30-62``.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence

import numpy as np

F32 = np.float32


# ----------------------------------------------------------------------------- schedule
def make_beta_schedule(n_timestep=1000, linear_start=0.00085, linear_end=0.0120):
    """'linear' branch of ldm/modules/diffusionmodules/util.py:21-25 (float64)."""
    return np.linspace(linear_start ** 0.5, linear_end ** 0.5, n_timestep, dtype=np.float64) ** 2


def register_schedule(n_timestep=1000, linear_start=0.00085, linear_end=0.0120):
    """ldm/models/diffusion/ddpm.py:138-159: alphas_cumprod stored as float32."""
    betas = make_beta_schedule(n_timestep, linear_start, linear_end)
    alphas_cumprod = np.cumprod(1.0 - betas, axis=0)
    return betas.astype(F32), alphas_cumprod.astype(F32)


def make_ddim_timesteps(num_ddim_timesteps, num_ddpm_timesteps=1000):
    """'uniform' branch of util.py:46-60 (note the +1)."""
    c = num_ddpm_timesteps // num_ddim_timesteps
    return np.asarray(list(range(0, num_ddpm_timesteps, c))) + 1


def make_ddim_timesteps_quad(num_ddim_timesteps, num_ddpm_timesteps=1000):
    """'quad' branch of util.py:49-50 (+1 like the uniform one, :56)."""
    return ((np.linspace(0, np.sqrt(num_ddpm_timesteps * .8), num_ddim_timesteps)) ** 2).astype(int) + 1


def make_ddim_sampling_parameters(alphacums, ddim_timesteps, eta):
    """util.py:63-74; alphacums is the float32 buffer (ddim_hacked.py:42 passes .cpu())."""
    alphas = alphacums[ddim_timesteps]
    alphas_prev = np.asarray([alphacums[0]] + alphacums[ddim_timesteps[:-1]].tolist())
    sigmas = eta * np.sqrt((1 - alphas_prev) / (1 - alphas) * (1 - alphas / alphas_prev))
    return sigmas, alphas, alphas_prev


def make_schedule(S, eta=0.0, n_timestep=1000, linear_start=0.00085, linear_end=0.0120, timesteps=None):
    """DDIMSampler.make_schedule, cldm/ddim_hacked.py:23-52.

    Returns dict with ddim_timesteps (int), and the four per-index scalars exactly as
    p_sample_ddim consumes them through torch.full (float32): ddim_alphas (f32 tensor
    slice), ddim_alphas_prev (float64 ndarray -> f32 by torch.full), ddim_sigmas,
    ddim_sqrt_one_minus_alphas (np.sqrt(1 - f32 alphas))."""
    _, ac = register_schedule(n_timestep, linear_start, linear_end)
    # timesteps (optional): a custom ascending grid in place of make_ddim_timesteps' uniform one -- the (D) pipeline's
    # `timesteps=` argument (pipeline_prompt_diffusion.py:101-142); the DDIM parameters derive from it the same way
    ts = make_ddim_timesteps(S, n_timestep) if timesteps is None else np.asarray(timesteps, dtype=np.int64)
    sigmas, alphas, alphas_prev = make_ddim_sampling_parameters(ac, ts, eta)
    return dict(ddim_timesteps=ts, ddim_alphas=alphas.astype(F32),
                ddim_alphas_prev=np.asarray(alphas_prev, dtype=np.float64).astype(F32),
                ddim_sigmas=np.asarray(sigmas).astype(F32),
                ddim_sqrt_one_minus_alphas=np.sqrt(1.0 - alphas).astype(F32))


# ----------------------------------------------------------------------------- primitives
def silu(x):
    return x / (1.0 + np.exp(-x))


def gelu(x):
    """Exact erf GELU (F.gelu default), attention.py:56."""
    from scipy.special import erf
    return (0.5 * x * (1.0 + erf(x.astype(np.float64) / math.sqrt(2.0)))).astype(F32)


def linear(x, w, b=None):
    y = x @ w.T
    return y if b is None else y + b


def conv2d(x, w, b=None, stride=1, padding=1):
    """nn.Conv2d (cross-correlation), im2col + sgemm.  x [B,C,H,W], w [O,C,kh,kw]."""
    B, C, H, W = x.shape
    O, _, kh, kw = w.shape
    if kh == 1 and kw == 1 and stride == 1:
        y = np.einsum("oc,bchw->bohw", w[:, :, 0, 0], x, optimize=True)
        return y if b is None else y + b[None, :, None, None]
    xp = np.pad(x, ((0, 0), (0, 0), (padding, padding), (padding, padding)))
    Ho = (H + 2 * padding - kh) // stride + 1
    Wo = (W + 2 * padding - kw) // stride + 1
    cols = np.empty((B, C, kh, kw, Ho, Wo), dtype=x.dtype)
    for i in range(kh):
        for j in range(kw):
            cols[:, :, i, j] = xp[:, :, i:i + stride * Ho:stride, j:j + stride * Wo:stride]
    cols = cols.reshape(B, C * kh * kw, Ho * Wo)
    y = np.matmul(w.reshape(O, -1)[None], cols).reshape(B, O, Ho, Wo)
    return y if b is None else y + b[None, :, None, None]


def group_norm(x, gamma, beta, groups=32, eps=1e-5):
    """nn.GroupNorm in fp32 (GroupNorm32, util.py:217-219); biased variance."""
    B, C, H, W = x.shape
    xg = x.reshape(B, groups, -1).astype(np.float64)
    mean = xg.mean(axis=2, keepdims=True)
    var = xg.var(axis=2, keepdims=True)
    y = ((xg - mean) / np.sqrt(var + eps)).astype(F32).reshape(B, C, H, W)
    return y * gamma[None, :, None, None] + beta[None, :, None, None]


def layer_norm(x, gamma, beta, eps=1e-5):
    x64 = x.astype(np.float64)
    mean = x64.mean(axis=-1, keepdims=True)
    var = x64.var(axis=-1, keepdims=True)
    return ((x64 - mean) / np.sqrt(var + eps)).astype(F32) * gamma + beta


def timestep_embedding(timesteps, dim, max_period=10000):
    """util.py:154-174: [cos, sin] order, freqs = exp(-ln(P) * arange(half)/half) in f32."""
    half = dim // 2
    freqs = np.exp(-math.log(max_period) * np.arange(half, dtype=F32) / F32(half)).astype(F32)
    args = np.asarray(timesteps, dtype=F32)[:, None] * freqs[None]
    return np.concatenate([np.cos(args), np.sin(args)], axis=-1).astype(F32)


# ----------------------------------------------------------------------------- blocks
class Net:
    """Parameter accessor for one network prefix."""

    def __init__(self, sd: Dict[str, np.ndarray], prefix: str):
        self.sd, self.prefix = sd, prefix

    def __call__(self, name):
        return np.asarray(self.sd[self.prefix + name], dtype=F32)

    def has(self, name):
        return (self.prefix + name) in self.sd


def cross_attention(p: Net, pre: str, x, context=None, heads=8):
    """CrossAttention.forward, attention.py:163-194 (fp32 logits, scale dh^-0.5)."""
    q = linear(x, p(pre + "to_q.weight"))
    ctx = x if context is None else context
    k = linear(ctx, p(pre + "to_k.weight"))
    v = linear(ctx, p(pre + "to_v.weight"))
    B, N, C = q.shape
    dh = C // heads

    def split(t):
        return t.reshape(t.shape[0], t.shape[1], heads, dh).transpose(0, 2, 1, 3)

    q, k, v = split(q), split(k), split(v)
    sim = np.matmul(q, k.transpose(0, 1, 3, 2)) * F32(dh ** -0.5)
    sim = sim - sim.max(axis=-1, keepdims=True)
    e = np.exp(sim)
    attn = e / e.sum(axis=-1, keepdims=True)
    out = np.matmul(attn, v).transpose(0, 2, 1, 3).reshape(B, N, C)
    return linear(out, p(pre + "to_out.0.weight"), p(pre + "to_out.0.bias"))


def feed_forward(p: Net, pre: str, x):
    """FeedForward(glu=True) = GEGLU -> Linear, attention.py:49-76."""
    h = linear(x, p(pre + "net.0.proj.weight"), p(pre + "net.0.proj.bias"))
    a, gate = np.split(h, 2, axis=-1)
    return linear(a * gelu(gate), p(pre + "net.2.weight"), p(pre + "net.2.bias"))


def basic_transformer_block(p: Net, pre: str, x, context, heads):
    """BasicTransformerBlock._forward, attention.py:271-275."""
    x = cross_attention(p, pre + "attn1.", layer_norm(x, p(pre + "norm1.weight"), p(pre + "norm1.bias")),
                        None, heads) + x
    x = cross_attention(p, pre + "attn2.", layer_norm(x, p(pre + "norm2.weight"), p(pre + "norm2.bias")),
                        context, heads) + x
    x = feed_forward(p, pre + "ff.", layer_norm(x, p(pre + "norm3.weight"), p(pre + "norm3.bias"))) + x
    return x


def spatial_transformer(p: Net, pre: str, x, context, heads=8):
    """SpatialTransformer.forward (use_linear=False, depth 1), attention.py:321-340; GN eps 1e-6 (:89)."""
    B, C, H, W = x.shape
    x_in = x
    h = group_norm(x, p(pre + "norm.weight"), p(pre + "norm.bias"), eps=1e-6)
    h = conv2d(h, p(pre + "proj_in.weight"), p(pre + "proj_in.bias"), padding=0)
    h = h.reshape(B, C, H * W).transpose(0, 2, 1)
    h = basic_transformer_block(p, pre + "transformer_blocks.0.", h, context, heads)
    h = h.transpose(0, 2, 1).reshape(B, C, H, W)
    h = conv2d(h, p(pre + "proj_out.weight"), p(pre + "proj_out.bias"), padding=0)
    return h + x_in


def resblock(p: Net, pre: str, x, emb):
    """ResBlock._forward (no up/down, no scale-shift), openaimodel.py:254-274."""
    h = conv2d(silu(group_norm(x, p(pre + "in_layers.0.weight"), p(pre + "in_layers.0.bias"))),
               p(pre + "in_layers.2.weight"), p(pre + "in_layers.2.bias"))
    emb_out = linear(silu(emb), p(pre + "emb_layers.1.weight"), p(pre + "emb_layers.1.bias"))
    h = h + emb_out[:, :, None, None]
    h = conv2d(silu(group_norm(h, p(pre + "out_layers.0.weight"), p(pre + "out_layers.0.bias"))),
               p(pre + "out_layers.3.weight"), p(pre + "out_layers.3.bias"))
    if p.has(pre + "skip_connection.weight"):
        x = conv2d(x, p(pre + "skip_connection.weight"), p(pre + "skip_connection.bias"), padding=0)
    return x + h


def time_embed(p: Net, t, model_channels):
    """time_embed(timestep_embedding(t)), openaimodel.py:526-531,767-768 / cldm.py:303-304."""
    e = timestep_embedding(t, model_channels)
    e = silu(linear(e, p("time_embed.0.weight"), p("time_embed.0.bias")))
    return linear(e, p("time_embed.2.weight"), p("time_embed.2.bias"))


def _input_block(p: Net, i: int, blk: dict, h, emb, ctx, heads):
    """One TimestepEmbedSequential of input_blocks (openaimodel.py:79-87)."""
    pre = f"input_blocks.{i}."
    if blk["kind"] == "conv_in":
        return conv2d(h, p(pre + "0.weight"), p(pre + "0.bias"))
    if blk["kind"] == "down":
        return conv2d(h, p(pre + "0.op.weight"), p(pre + "0.op.bias"), stride=2)  # :148-152
    h = resblock(p, pre + "0.", h, emb)
    if blk["attn"]:
        h = spatial_transformer(p, pre + "1.", h, ctx, heads)
    return h


def _middle(p: Net, h, emb, ctx, heads):
    h = resblock(p, "middle_block.0.", h, emb)
    h = spatial_transformer(p, "middle_block.1.", h, ctx, heads)
    return resblock(p, "middle_block.2.", h, emb)


def hint_block(p: Net, name: str, layers: Sequence[dict], x):
    """input_hint_block / input_cond_block, cldm/cldm.py:147-181: conv(+SiLU) x7, conv."""
    for l in layers:
        x = conv2d(x, p(f"{name}.{l['idx']}.weight"), p(f"{name}.{l['idx']}.bias"), stride=l["stride"])
        if l["silu"]:
            x = silu(x)
    return x


def controlnet_forward(sd, cfg, layouts, x, t, example_pair, query, context):
    """ControlNet.forward, cldm/cldm.py:302-325 -> list of 13 residuals."""
    p = Net(sd, "control_model.")
    emb = time_embed(p, t, cfg.model_channels)
    guided_hint = hint_block(p, "input_hint_block", layouts["hint_pair"], example_pair) + \
        hint_block(p, "input_cond_block", layouts["hint_query"], query)
    outs = []
    h = x
    for i, blk in enumerate(layouts["enc"]):
        h = _input_block(p, i, blk, h, emb, context, cfg.num_heads)
        if guided_hint is not None:
            h = h + guided_hint
            guided_hint = None
        outs.append(conv2d(h, p(f"zero_convs.{i}.0.weight"), p(f"zero_convs.{i}.0.bias"), padding=0))
    h = _middle(p, h, emb, context, cfg.num_heads)
    outs.append(conv2d(h, p("middle_block_out.0.weight"), p("middle_block_out.0.bias"), padding=0))
    return outs


def controlled_unet_forward(sd, cfg, layouts, x, t, context, control, only_mid_control=False):
    """ControlledUnetModel.forward, cldm/cldm.py:23-45."""
    p = Net(sd, "model.diffusion_model.")
    control = list(control) if control is not None else None
    emb = time_embed(p, t, cfg.model_channels)
    hs = []
    h = x
    for i, blk in enumerate(layouts["enc"]):
        h = _input_block(p, i, blk, h, emb, context, cfg.num_heads)
        hs.append(h)
    h = _middle(p, h, emb, context, cfg.num_heads)
    if control is not None:
        h = h + control.pop()
    for i, blk in enumerate(layouts["dec"]):
        if only_mid_control or control is None:
            h = np.concatenate([h, hs.pop()], axis=1)
        else:
            h = np.concatenate([h, hs.pop() + control.pop()], axis=1)
        pre = f"output_blocks.{i}."
        h = resblock(p, pre + "0.", h, emb)
        j = 1
        if blk["attn"]:
            h = spatial_transformer(p, pre + "1.", h, context, cfg.num_heads)
            j = 2
        if blk["up"]:
            h = np.repeat(np.repeat(h, 2, axis=2), 2, axis=3)  # F.interpolate nearest x2, :115
            h = conv2d(h, p(pre + f"{j}.conv.weight"), p(pre + f"{j}.conv.bias"))
    h = silu(group_norm(h, p("out.0.weight"), p("out.0.bias")))
    return conv2d(h, p("out.2.weight"), p("out.2.bias"))


def apply_model(sd, cfg, layouts, x, t, context, example_pair, query, control_scales=None,
                only_mid_control=False, return_control=False):
    """ControlLDM.apply_model, cldm/cldm.py:369-382."""
    control = controlnet_forward(sd, cfg, layouts, x, t, example_pair, query, context)
    scales = [1.0] * 13 if control_scales is None else control_scales
    control = [c * F32(s) for c, s in zip(control, scales)]
    eps = controlled_unet_forward(sd, cfg, layouts, x, t, context, control, only_mid_control)
    return (eps, control) if return_control else eps


# ----------------------------------------------------------------------------- sampler
def p_sample_ddim(sd, cfg, layouts, sched, x, cond, uncond, index, step, cfg_scale,
                  control_scales=None, noise=None, temperature=1.0, only_mid_control=False):
    """DDIMSampler.p_sample_ddim, cldm/ddim_hacked.py:181-234 (eps parameterisation).

    cond/uncond: dicts with 'c_crossattn' [B,L,D], 'example_pair', 'query'.  CFG batching
    happens whenever uncond is given (uncond first), :188-193."""
    B = x.shape[0]
    t = np.full((B,), step, dtype=np.int64)
    if uncond is not None:
        x_in = np.concatenate([x, x])
        t_in = np.concatenate([t, t])
        c_in = {k: np.concatenate([uncond[k], cond[k]]) for k in cond}
        out = apply_model(sd, cfg, layouts, x_in, t_in, c_in["c_crossattn"], c_in["example_pair"],
                          c_in["query"], control_scales, only_mid_control)
        e_u, e_c = out[:B], out[B:]
        e_t = e_u + F32(cfg_scale) * (e_c - e_u)
    else:
        e_t = apply_model(sd, cfg, layouts, x, t, cond["c_crossattn"], cond["example_pair"],
                          cond["query"], control_scales, only_mid_control)
    a_t = sched["ddim_alphas"][index]
    a_prev = sched["ddim_alphas_prev"][index]
    sigma_t = sched["ddim_sigmas"][index]
    sqrt_one_minus_at = sched["ddim_sqrt_one_minus_alphas"][index]
    pred_x0 = (x - sqrt_one_minus_at * e_t) / np.sqrt(a_t)
    dir_xt = np.sqrt(F32(1.0) - a_prev - sigma_t ** 2) * e_t
    nz = 0.0 if noise is None else sigma_t * noise * F32(temperature)
    x_prev = np.sqrt(a_prev) * pred_x0 + dir_xt + nz
    return x_prev.astype(F32), pred_x0.astype(F32), e_t.astype(F32)


def q_sample(cfg, x_start, t, noise):
    """DDPM.q_sample, ldm/models/diffusion/ddpm.py:356-359, on the fp32 buffers of register_schedule (:162-163)."""
    betas = np.linspace(cfg.linear_start ** 0.5, cfg.linear_end ** 0.5, cfg.timesteps, dtype=np.float64) ** 2
    ac = np.cumprod(1.0 - betas)
    sa = np.sqrt(ac).astype(F32)[t].reshape(-1, 1, 1, 1)
    sb = np.sqrt(1.0 - ac).astype(F32)[t].reshape(-1, 1, 1, 1)
    return (sa * x_start + sb * noise).astype(F32)


def ddim_sampling(sd, cfg, layouts, S, x_T, cond, uncond, cfg_scale, eta=0.0, control_scales=None,
                  noises=None, steps_limit: Optional[int] = None, mask=None, x0=None, q_noise=None, timesteps=None,
                  temperature=1.0, only_mid_control=False, ucg_schedule=None):
    """DDIMSampler.sample + ddim_sampling, cldm/ddim_hacked.py:55-178 with log_every_t=1.
    mask/x0: the inpainting blend of :154-157, with the q_sample noise of each step supplied (q_noise[i]).

    Returns (samples, x_inter list of S+1 latents, pred_x0 list)."""
    sched = make_schedule(S, eta, cfg.timesteps, cfg.linear_start, cfg.linear_end, timesteps)
    S = len(sched["ddim_timesteps"])
    img = x_T
    x_inter, preds = [img], [img]
    time_range = np.flip(sched["ddim_timesteps"])
    for i, step in enumerate(time_range):
        if steps_limit is not None and i >= steps_limit:
            break
        index = S - i - 1
        if mask is not None:
            img_orig = q_sample(cfg, x0, np.full((img.shape[0],), int(step), np.int64), q_noise[i])
            img = (img_orig * mask + (F32(1.0) - mask) * img).astype(F32)
        nz = None if noises is None else noises[i]
        if ucg_schedule is not None:        # ddim_hacked.py:159-161
            assert len(ucg_schedule) == len(time_range)
            cfg_scale = ucg_schedule[i]
        img, pred_x0, _ = p_sample_ddim(sd, cfg, layouts, sched, img, cond, uncond, index, int(step),
                                        cfg_scale, control_scales, nz, temperature, only_mid_control)
        x_inter.append(img)
        preds.append(pred_x0)
    return img, x_inter, preds


def ddim_encode(sd, cfg, layouts, sched, x0, cond, t_enc):
    """DDIMSampler.encode with unconditional_guidance_scale == 1, cldm/ddim_hacked.py:237-282.  The reference hands the loop
    index i -- not the DDIM timestep -- to apply_model (:256); restated as it runs."""
    alphas_next = sched["ddim_alphas"][:t_enc]
    alphas = sched["ddim_alphas_prev"][:t_enc]
    x_next = x0
    one = F32(1.0)
    for i in range(t_enc):
        t = np.full((x0.shape[0],), i, dtype=np.int64)
        noise_pred = apply_model(sd, cfg, layouts, x_next, t, cond["c_crossattn"], cond["example_pair"], cond["query"])
        xt_weighted = np.sqrt(alphas_next[i] / alphas[i]) * x_next
        weighted = np.sqrt(alphas_next[i]) * (np.sqrt(one / alphas_next[i] - one) - np.sqrt(one / alphas[i] - one)) * noise_pred
        x_next = (xt_weighted + weighted).astype(F32)
    return x_next


def stochastic_encode(sched, x0, t, noise):
    """DDIMSampler.stochastic_encode, cldm/ddim_hacked.py:284-299 (t indexes the DDIM grid)."""
    t = np.asarray(t, np.int64).reshape(-1)
    sa = np.sqrt(sched["ddim_alphas"])[t].reshape(-1, 1, 1, 1)
    sb = sched["ddim_sqrt_one_minus_alphas"][t].reshape(-1, 1, 1, 1)
    return (sa * x0 + sb * noise).astype(F32)


def ddim_decode(sd, cfg, layouts, sched, x_latent, cond, uncond, t_start, cfg_scale):
    """DDIMSampler.decode, cldm/ddim_hacked.py:301-318 (eta = 0)."""
    timesteps = sched["ddim_timesteps"][:t_start]
    x_dec = x_latent
    for i, step in enumerate(np.flip(timesteps)):
        index = len(timesteps) - i - 1
        x_dec, _, _ = p_sample_ddim(sd, cfg, layouts, sched, x_dec, cond, uncond, index, int(step), cfg_scale)
    return x_dec


def make_layouts(cfg, weights_mod):
    """Topology tables (shared with the host package's inventory, prompt-diffusion_amd/weights.py)."""
    return dict(enc=weights_mod.encoder_layout(cfg), dec=weights_mod.decoder_layout(cfg),
                hint_pair=weights_mod.hint_layout(cfg, cfg.hint_channels),
                hint_query=weights_mod.hint_layout(cfg, cfg.query_channels))


# ----------------------------------------------------------------------------- first-stage decoder
def vae_resnet_block(p: Net, pre: str, x):
    """ResnetBlock.forward with temb=None, ldm/modules/diffusionmodules/model.py:121-141 (GroupNorm eps 1e-6, :41-42)."""
    h = conv2d(silu(group_norm(x, p(pre + "norm1.weight"), p(pre + "norm1.bias"), eps=1e-6)),
               p(pre + "conv1.weight"), p(pre + "conv1.bias"))
    h = conv2d(silu(group_norm(h, p(pre + "norm2.weight"), p(pre + "norm2.bias"), eps=1e-6)),
               p(pre + "conv2.weight"), p(pre + "conv2.bias"))
    if p.has(pre + "nin_shortcut.weight"):
        x = conv2d(x, p(pre + "nin_shortcut.weight"), p(pre + "nin_shortcut.bias"), padding=0)
    return x + h


def vae_attn_block(p: Net, pre: str, x):
    """AttnBlock.forward (single head over all channels), model.py:171-202."""
    B, C, H, W = x.shape
    h = group_norm(x, p(pre + "norm.weight"), p(pre + "norm.bias"), eps=1e-6)
    q = conv2d(h, p(pre + "q.weight"), p(pre + "q.bias"), padding=0).reshape(B, C, H * W).transpose(0, 2, 1)
    k = conv2d(h, p(pre + "k.weight"), p(pre + "k.bias"), padding=0).reshape(B, C, H * W)
    v = conv2d(h, p(pre + "v.weight"), p(pre + "v.bias"), padding=0).reshape(B, C, H * W)
    w_ = np.matmul(q, k) * F32(int(C) ** (-0.5))
    w_ = w_ - w_.max(axis=2, keepdims=True)
    e = np.exp(w_)
    w_ = e / e.sum(axis=2, keepdims=True)
    h = np.matmul(v, w_.transpose(0, 2, 1)).reshape(B, C, H, W)
    h = conv2d(h, p(pre + "proj_out.weight"), p(pre + "proj_out.bias"), padding=0)
    return x + h


def vae_decode(sd, cfg, vae_layout, z):
    """LatentDiffusion.decode_first_stage (ddpm.py:820-828: z / scale_factor) -> AutoencoderKL.decode
    (autoencoder.py:89-92: post_quant_conv, decoder) -> Decoder.forward (model.py:619-653)."""
    p = Net(sd, "first_stage_model.")
    z = (F32(1.0) / F32(cfg.scale_factor) * z).astype(F32)
    z = conv2d(z, p("post_quant_conv.weight"), p("post_quant_conv.bias"), padding=0)
    h = conv2d(z, p("decoder.conv_in.weight"), p("decoder.conv_in.bias"))
    h = vae_resnet_block(p, "decoder.mid.block_1.", h)
    h = vae_attn_block(p, "decoder.mid.attn_1.", h)
    h = vae_resnet_block(p, "decoder.mid.block_2.", h)
    for lvl in vae_layout:  # execution order: highest level first
        for j in range(len(lvl["blocks"])):
            h = vae_resnet_block(p, f"decoder.up.{lvl['level']}.block.{j}.", h)
        if lvl["upsample"]:
            h = np.repeat(np.repeat(h, 2, axis=2), 2, axis=3)   # F.interpolate(scale_factor=2.0, mode="nearest"), :62
            h = conv2d(h, p(f"decoder.up.{lvl['level']}.upsample.conv.weight"), p(f"decoder.up.{lvl['level']}.upsample.conv.bias"))
    h = silu(group_norm(h, p("decoder.norm_out.weight"), p("decoder.norm_out.bias"), eps=1e-6))
    return conv2d(h, p("decoder.conv_out.weight"), p("decoder.conv_out.bias"))


# ---------------------------------------------------------------------------------------------------
# UniPC.  The reference takes this scheduler from diffusers (`README.md:49`), which is not in
# the reference tree: PARITY UNPINNED.  Independent fp64 restatement of the published algorithm (Zhao et al. 2023,
# UniP-2 / UniC-2, data prediction, B(h) = e^h - 1) with the order <= 2 coefficients written out in closed form --
# deliberately not the generic R/b solve the product scheduler uses, so that the two can check each other.
def unipc2_sample(eps_fn, x_T, alphas_cumprod, timesteps):
    """eps_fn(x, t) -> eps.  timesteps descending; an extra final point with sigma = 0 is appended.
    Order warm-up 1, 2, 2, ...; the last step is first order (lower_order_final)."""
    ac = np.asarray(alphas_cumprod, np.float64)[np.asarray(timesteps, np.int64)]
    alpha = np.concatenate([np.sqrt(ac), [1.0]])
    sigma = np.concatenate([np.sqrt(1.0 - ac), [0.0]])
    with np.errstate(divide="ignore"):
        lam = np.log(alpha) - np.log(sigma)
    n = len(timesteps)
    x = np.asarray(x_T, np.float64)
    m_prev = None        # x0 prediction at point i-1
    x_last = None        # sample the predictor started from at point i-1
    prev_order = 1
    for i in range(n):
        m = (x - sigma[i] * np.asarray(eps_fn(x, int(timesteps[i])), np.float64)) / alpha[i]
        if i > 0:        # UniC: redo the step (i-1 -> i) with the new evaluation at i
            h = lam[i] - lam[i - 1]
            phi1 = np.expm1(-h)
            base = sigma[i] / sigma[i - 1] * x_last - alpha[i] * phi1 * m_prev
            if prev_order == 1:
                x = base - alpha[i] * phi1 * 0.5 * (m - m_prev)
            else:
                r = (lam[i - 2] - lam[i - 1]) / h
                b1 = (phi1 / (-h) - 1.0) / phi1
                b2 = ((phi1 / (-h) - 1.0) / (-h) - 0.5) * 2.0 / phi1
                rho1 = (b1 - b2) / (1.0 - r)
                rho2 = b1 - rho1
                x = base - alpha[i] * phi1 * (rho1 * (m_pp - m_prev) / r + rho2 * (m - m_prev))
        order = 1 if (i == 0 or i == n - 1) else 2
        x_last = x
        if i == n - 1:
            x_next = m * 1.0                       # sigma -> 0: the data prediction itself
        else:
            h = lam[i + 1] - lam[i]
            phi1 = np.expm1(-h)
            x_next = sigma[i + 1] / sigma[i] * x - alpha[i + 1] * phi1 * m
            if order == 2:
                r = (lam[i - 1] - lam[i]) / h
                x_next = x_next - alpha[i + 1] * phi1 * 0.5 * (m_prev - m) / r
        m_pp, m_prev, prev_order = m_prev, m, order
        x = x_next
    return x


# ---------------------------------------------------------------------------------------------------
# CLIP text transformer: FrozenCLIPEmbedder.forward, ldm/modules/encoders/modules.py:118-128
# (layer "last": `outputs.last_hidden_state`), i.e. transformers' CLIPTextModel: token + position embeddings,
# pre-LN blocks with causal self-attention and quick-GELU MLP, final LayerNorm.  The model code lives in the
# `transformers` dependency (not in the reference tree); this restates its published architecture and is pinned against
# `transformers.CLIPTextModel` itself run in the build container (tests/golden/make_golden.py, clip_*.npz).
def clip_text_forward(sd, cfg, ids, prefix="cond_stage_model.transformer.text_model.", clip_skip=0):
    """clip_skip k: hidden_states[-(k+1)] + final_layer_norm (pipeline_prompt_diffusion.py:398-413) = the first
    text_layers - k blocks."""
    p = lambda n: sd[prefix + n]
    B, L = ids.shape
    C, H = cfg.context_dim, cfg.text_heads
    dh = C // H
    x = p("embeddings.token_embedding.weight")[ids] + p("embeddings.position_embedding.weight")[None, :L]
    causal = np.triu(np.full((L, L), -np.inf, F32), k=1)
    for i in range(cfg.text_layers - int(clip_skip or 0)):
        pre = f"encoder.layers.{i}."
        h = layer_norm(x, p(pre + "layer_norm1.weight"), p(pre + "layer_norm1.bias"), eps=1e-5)
        q = linear(h, p(pre + "self_attn.q_proj.weight"), p(pre + "self_attn.q_proj.bias")) * F32(dh ** -0.5)
        k = linear(h, p(pre + "self_attn.k_proj.weight"), p(pre + "self_attn.k_proj.bias"))
        v = linear(h, p(pre + "self_attn.v_proj.weight"), p(pre + "self_attn.v_proj.bias"))
        sp = lambda t: t.reshape(B, L, H, dh).transpose(0, 2, 1, 3)
        s_ = np.einsum("bhid,bhjd->bhij", sp(q), sp(k)) + causal
        s_ = s_ - s_.max(-1, keepdims=True)
        e = np.exp(s_)
        a = e / e.sum(-1, keepdims=True)
        o = np.einsum("bhij,bhjd->bhid", a, sp(v)).transpose(0, 2, 1, 3).reshape(B, L, C)
        x = x + linear(o, p(pre + "self_attn.out_proj.weight"), p(pre + "self_attn.out_proj.bias"))
        h = layer_norm(x, p(pre + "layer_norm2.weight"), p(pre + "layer_norm2.bias"), eps=1e-5)
        h = linear(h, p(pre + "mlp.fc1.weight"), p(pre + "mlp.fc1.bias"))
        h = h * (F32(1.0) / (F32(1.0) + np.exp(F32(-1.702) * h)))                 # quick_gelu
        x = x + linear(h, p(pre + "mlp.fc2.weight"), p(pre + "mlp.fc2.bias"))
    return layer_norm(x, p("final_layer_norm.weight"), p("final_layer_norm.bias"), eps=1e-5).astype(F32)
