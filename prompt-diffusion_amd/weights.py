"""Model configuration, parameter inventory and the synthetic-weight recipe.

The engine consumes weights under the reference's checkpoint names
(``model.diffusion_model.*`` for the SD1.5 UNet, ``control_model.*`` for the
Prompt-Diffusion ControlNet; reference ``tool_add_control.py:36-45``,
``cldm/model.py:12-21``).  ``param_spec`` enumerates every tensor the two
networks own, in the order the reference constructors create them
(``cldm/cldm.py:48-297`` and ``ldm/modules/diffusionmodules/openaimodel.py:442-736``),
so a checkpoint state-dict walk, the synthetic recipe and the engine's own
registry (``pd_param_name``) can be cross-checked by name and shape.

No trained checkpoint exists offline; ``synth_state_dict`` generates seeded
values, one independent Philox stream per tensor keyed by crc32(name), so any
subset can be regenerated on any host (the GPU box included) without torch.
"""
from __future__ import annotations

import dataclasses
import zlib
from typing import Dict, Iterator, List, Sequence, Tuple

import numpy as np

UNET_PREFIX = "model.diffusion_model."
CNET_PREFIX = "control_model."


@dataclasses.dataclass(frozen=True)
class ModelConfig:
    """Hyper-parameters of ``models/cldm_v15.yaml:30-62`` (defaults = SD1.5)."""

    in_channels: int = 4
    out_channels: int = 4
    hint_channels: int = 6      # example pair (two RGB images), cldm_v15.yaml:34
    query_channels: int = 3     # cldm/cldm.py:166
    model_channels: int = 320
    channel_mult: Tuple[int, ...] = (1, 2, 4, 4)
    num_res_blocks: int = 2
    attention_resolutions: Tuple[int, ...] = (4, 2, 1)
    num_heads: int = 8
    context_dim: int = 768
    context_len: int = 77
    hint_widths: Tuple[int, ...] = (16, 16, 32, 32, 96, 96, 256)  # cldm/cldm.py:147-163
    # schedule, cldm_v15.yaml:4-8
    timesteps: int = 1000
    linear_start: float = 0.00085
    linear_end: float = 0.0120

    # first-stage KL-VAE decoder, models/cldm_v15.yaml:64-85 (vae_ch = 0: not built)
    vae_ch: int = 128
    vae_ch_mult: Tuple[int, ...] = (1, 2, 4, 4)
    vae_num_res_blocks: int = 2
    vae_out_ch: int = 3
    scale_factor: float = 0.18215       # cldm_v15.yaml:17

    # cond-stage CLIP text transformer (FrozenCLIPEmbedder -> CLIPTextModel "openai/clip-vit-large-patch14",
    # ldm/modules/encoders/modules.py:88-131): width = context_dim, length = context_len (text_layers = 0: not built)
    text_vocab: int = 49408
    text_layers: int = 12
    text_heads: int = 12
    text_ff: int = 3072

    @property
    def time_embed_dim(self) -> int:
        return 4 * self.model_channels


SD15 = ModelConfig()
# A reduced network with the same topology (4 levels, attention at ds 1/2/4,
# 8 heads) used by fast parity tests: channels 64/128/256/256, dh 8/16/32.
TINY = ModelConfig(model_channels=64, context_dim=96, context_len=77,
                   hint_widths=(8, 8, 16, 16, 24, 24, 32), vae_ch=32,
                   text_vocab=1000, text_layers=2, text_heads=3, text_ff=192)

Spec = Tuple[str, Tuple[int, ...], str]  # (name, shape, kind)


def _res(prefix: str, cin: int, cout: int, temb: int) -> Iterator[Spec]:
    # ResBlock, openaimodel.py:200-240
    yield prefix + "in_layers.0.weight", (cin,), "gamma"
    yield prefix + "in_layers.0.bias", (cin,), "beta"
    yield prefix + "in_layers.2.weight", (cout, cin, 3, 3), "w"
    yield prefix + "in_layers.2.bias", (cout,), "b"
    yield prefix + "emb_layers.1.weight", (cout, temb), "w"
    yield prefix + "emb_layers.1.bias", (cout,), "b"
    yield prefix + "out_layers.0.weight", (cout,), "gamma"
    yield prefix + "out_layers.0.bias", (cout,), "beta"
    yield prefix + "out_layers.3.weight", (cout, cout, 3, 3), "w"
    yield prefix + "out_layers.3.bias", (cout,), "b"
    if cin != cout:
        yield prefix + "skip_connection.weight", (cout, cin, 1, 1), "w"
        yield prefix + "skip_connection.bias", (cout,), "b"


def _st(prefix: str, ch: int, ctx: int) -> Iterator[Spec]:
    # SpatialTransformer (depth 1, use_linear False), attention.py:287-319
    yield prefix + "norm.weight", (ch,), "gamma"
    yield prefix + "norm.bias", (ch,), "beta"
    yield prefix + "proj_in.weight", (ch, ch, 1, 1), "w"
    yield prefix + "proj_in.bias", (ch,), "b"
    t = prefix + "transformer_blocks.0."
    yield t + "attn1.to_q.weight", (ch, ch), "w"
    yield t + "attn1.to_k.weight", (ch, ch), "w"
    yield t + "attn1.to_v.weight", (ch, ch), "w"
    yield t + "attn1.to_out.0.weight", (ch, ch), "w"
    yield t + "attn1.to_out.0.bias", (ch,), "b"
    yield t + "ff.net.0.proj.weight", (8 * ch, ch), "w"
    yield t + "ff.net.0.proj.bias", (8 * ch,), "b"
    yield t + "ff.net.2.weight", (ch, 4 * ch), "w"
    yield t + "ff.net.2.bias", (ch,), "b"
    yield t + "attn2.to_q.weight", (ch, ch), "w"
    yield t + "attn2.to_k.weight", (ch, ctx), "w"
    yield t + "attn2.to_v.weight", (ch, ctx), "w"
    yield t + "attn2.to_out.0.weight", (ch, ch), "w"
    yield t + "attn2.to_out.0.bias", (ch,), "b"
    for n in ("norm1", "norm2", "norm3"):
        yield t + n + ".weight", (ch,), "gamma"
        yield t + n + ".bias", (ch,), "beta"
    yield prefix + "proj_out.weight", (ch, ch, 1, 1), "w"
    yield prefix + "proj_out.bias", (ch,), "b"


def _conv(prefix: str, cin: int, cout: int, k: int) -> Iterator[Spec]:
    yield prefix + "weight", (cout, cin, k, k), "w"
    yield prefix + "bias", (cout,), "b"


def encoder_layout(cfg: ModelConfig) -> List[dict]:
    """The 12 input blocks shared by UNet and ControlNet (openaimodel.py:542-621).

    Each entry: kind 'conv_in' | 'res' | 'down', cin, cout, attn(bool), ds.
    """
    mc = cfg.model_channels
    blocks = [dict(kind="conv_in", cin=cfg.in_channels, cout=mc, attn=False, ds=1)]
    ch, ds = mc, 1
    for level, mult in enumerate(cfg.channel_mult):
        for _ in range(cfg.num_res_blocks):
            blocks.append(dict(kind="res", cin=ch, cout=mult * mc,
                               attn=ds in cfg.attention_resolutions, ds=ds))
            ch = mult * mc
        if level != len(cfg.channel_mult) - 1:
            blocks.append(dict(kind="down", cin=ch, cout=ch, attn=False, ds=ds))
            ds *= 2
    return blocks


def decoder_layout(cfg: ModelConfig) -> List[dict]:
    """The 12 output blocks (openaimodel.py:662-724): res(ch+skip -> mult*mc), attn, up."""
    mc = cfg.model_channels
    enc = encoder_layout(cfg)
    chans = [b["cout"] for b in enc]
    ch = enc[-1]["cout"]
    ds = 2 ** (len(cfg.channel_mult) - 1)
    out = []
    for level, mult in list(enumerate(cfg.channel_mult))[::-1]:
        for i in range(cfg.num_res_blocks + 1):
            ich = chans.pop()
            blk = dict(cin=ch + ich, skip=ich, cout=mc * mult,
                       attn=ds in cfg.attention_resolutions, up=False, ds=ds)
            ch = mc * mult
            if level and i == cfg.num_res_blocks:
                blk["up"] = True
                ds //= 2
            out.append(blk)
    return out


def _encoder_spec(prefix: str, cfg: ModelConfig) -> Iterator[Spec]:
    temb = cfg.time_embed_dim
    yield prefix + "time_embed.0.weight", (temb, cfg.model_channels), "w"
    yield prefix + "time_embed.0.bias", (temb,), "b"
    yield prefix + "time_embed.2.weight", (temb, temb), "w"
    yield prefix + "time_embed.2.bias", (temb,), "b"
    for i, b in enumerate(encoder_layout(cfg)):
        p = f"{prefix}input_blocks.{i}."
        if b["kind"] == "conv_in":
            yield from _conv(p + "0.", b["cin"], b["cout"], 3)
        elif b["kind"] == "res":
            yield from _res(p + "0.", b["cin"], b["cout"], temb)
            if b["attn"]:
                yield from _st(p + "1.", b["cout"], cfg.context_dim)
        else:
            yield from _conv(p + "0.op.", b["cin"], b["cout"], 3)


def _middle_spec(prefix: str, cfg: ModelConfig) -> Iterator[Spec]:
    ch = cfg.model_channels * cfg.channel_mult[-1]
    temb = cfg.time_embed_dim
    yield from _res(prefix + "middle_block.0.", ch, ch, temb)
    yield from _st(prefix + "middle_block.1.", ch, cfg.context_dim)
    yield from _res(prefix + "middle_block.2.", ch, ch, temb)


def unet_spec(cfg: ModelConfig, prefix: str = UNET_PREFIX) -> List[Spec]:
    out = list(_encoder_spec(prefix, cfg)) + list(_middle_spec(prefix, cfg))
    temb = cfg.time_embed_dim
    for i, b in enumerate(decoder_layout(cfg)):
        p = f"{prefix}output_blocks.{i}."
        out += list(_res(p + "0.", b["cin"], b["cout"], temb))
        j = 1
        if b["attn"]:
            out += list(_st(p + "1.", b["cout"], cfg.context_dim))
            j = 2
        if b["up"]:
            out += list(_conv(p + f"{j}.conv.", b["cout"], b["cout"], 3))
    mc = cfg.model_channels
    out += [(prefix + "out.0.weight", (mc,), "gamma"), (prefix + "out.0.bias", (mc,), "beta")]
    out += list(_conv(prefix + "out.2.", mc, cfg.out_channels, 3))
    return out


def hint_layout(cfg: ModelConfig, cin: int) -> List[dict]:
    """input_hint_block / input_cond_block: 8 convs (cldm/cldm.py:147-181)."""
    w = cfg.hint_widths
    chain = [(cin, w[0], 1), (w[0], w[1], 1), (w[1], w[2], 2), (w[2], w[3], 1),
             (w[3], w[4], 2), (w[4], w[5], 1), (w[5], w[6], 2), (w[6], cfg.model_channels, 1)]
    return [dict(idx=2 * i, cin=a, cout=b, stride=s, silu=i < 7) for i, (a, b, s) in enumerate(chain)]


def controlnet_spec(cfg: ModelConfig, prefix: str = CNET_PREFIX) -> List[Spec]:
    out = list(_encoder_spec(prefix, cfg))
    enc = encoder_layout(cfg)
    for i, b in enumerate(enc):
        out += list(_conv(f"{prefix}zero_convs.{i}.0.", b["cout"], b["cout"], 1))
    for name, cin in (("input_hint_block", cfg.hint_channels), ("input_cond_block", cfg.query_channels)):
        for l in hint_layout(cfg, cin):
            out += list(_conv(f"{prefix}{name}.{l['idx']}.", l["cin"], l["cout"], 3))
    out += list(_middle_spec(prefix, cfg))
    ch = cfg.model_channels * cfg.channel_mult[-1]
    out += list(_conv(prefix + "middle_block_out.0.", ch, ch, 1))
    return out


def param_spec(cfg: ModelConfig) -> List[Spec]:
    return unet_spec(cfg) + controlnet_spec(cfg)


VAE_PREFIX = "first_stage_model."


def vae_layout(cfg: ModelConfig) -> List[dict]:
    """Decoder.up levels in EXECUTION order (highest i_level first), ldm/modules/diffusionmodules/model.py:588-606."""
    nres = len(cfg.vae_ch_mult)
    block_in = cfg.vae_ch * cfg.vae_ch_mult[-1]
    out = []
    for i_level in reversed(range(nres)):
        block_out = cfg.vae_ch * cfg.vae_ch_mult[i_level]
        blocks = []
        for _ in range(cfg.vae_num_res_blocks + 1):
            blocks.append((block_in, block_out))
            block_in = block_out
        out.append(dict(level=i_level, blocks=blocks, upsample=i_level != 0, ch=block_in))
    return out


def _vres(prefix: str, cin: int, cout: int) -> Iterator[Spec]:
    # ResnetBlock (temb_channels = 0), model.py:82-141
    yield prefix + "norm1.weight", (cin,), "gamma"
    yield prefix + "norm1.bias", (cin,), "beta"
    yield from _conv(prefix + "conv1.", cin, cout, 3)
    yield prefix + "norm2.weight", (cout,), "gamma"
    yield prefix + "norm2.bias", (cout,), "beta"
    yield from _conv(prefix + "conv2.", cout, cout, 3)
    if cin != cout:
        yield from _conv(prefix + "nin_shortcut.", cin, cout, 1)


def vae_spec(cfg: ModelConfig, prefix: str = VAE_PREFIX) -> List[Spec]:
    """post_quant_conv + Decoder parameters in the reference's registration order
    (ldm/models/autoencoder.py:34, model.py:546-653): decoder first (conv_in, mid, up.0..3, norm_out, conv_out),
    then post_quant_conv."""
    if cfg.vae_ch <= 0:
        return []
    d = prefix + "decoder."
    top = cfg.vae_ch * cfg.vae_ch_mult[-1]
    out: List[Spec] = list(_conv(d + "conv_in.", cfg.in_channels, top, 3))
    out += list(_vres(d + "mid.block_1.", top, top))
    out += [(d + "mid.attn_1.norm.weight", (top,), "gamma"), (d + "mid.attn_1.norm.bias", (top,), "beta")]
    for n in ("q", "k", "v", "proj_out"):
        out += list(_conv(d + f"mid.attn_1.{n}.", top, top, 1))
    out += list(_vres(d + "mid.block_2.", top, top))
    by_level = {l["level"]: l for l in vae_layout(cfg)}
    for lvl in range(len(cfg.vae_ch_mult)):          # module order: up.0 .. up.N-1
        l = by_level[lvl]
        for j, (ci, co) in enumerate(l["blocks"]):
            out += list(_vres(d + f"up.{lvl}.block.{j}.", ci, co))
        if l["upsample"]:
            out += list(_conv(d + f"up.{lvl}.upsample.conv.", l["ch"], l["ch"], 3))
    out += [(d + "norm_out.weight", (cfg.vae_ch,), "gamma"), (d + "norm_out.bias", (cfg.vae_ch,), "beta")]
    out += list(_conv(d + "conv_out.", cfg.vae_ch, cfg.vae_out_ch, 3))
    out += list(_conv(prefix + "post_quant_conv.", cfg.in_channels, cfg.in_channels, 1))
    return out


def synth_vae_state_dict(cfg: ModelConfig, seed: int = 1234) -> Dict[str, np.ndarray]:
    return {n: synth_tensor(n, s, k, seed) for n, s, k in vae_spec(cfg)}


TEXT_PREFIX = "cond_stage_model.transformer.text_model."


def text_spec(cfg: ModelConfig, prefix: str = TEXT_PREFIX) -> List[Spec]:
    """Parameters of the CLIP text transformer under the names the SD1.5 checkpoint stores them
    (`FrozenCLIPEmbedder.transformer` = transformers' `CLIPTextModel`, ldm/modules/encoders/modules.py:98), in module
    order: embeddings, encoder.layers.i.{self_attn.{k,v,q,out}_proj, layer_norm1, mlp.fc1/fc2, layer_norm2},
    final_layer_norm."""
    out: List[Spec] = []
    if cfg.text_layers <= 0:
        return out
    C, F = cfg.context_dim, cfg.text_ff
    out.append((prefix + "embeddings.token_embedding.weight", (cfg.text_vocab, C), "w"))
    out.append((prefix + "embeddings.position_embedding.weight", (cfg.context_len, C), "w"))
    for i in range(cfg.text_layers):
        L = f"{prefix}encoder.layers.{i}."
        for nm in ("k_proj", "v_proj", "q_proj", "out_proj"):
            out.append((L + f"self_attn.{nm}.weight", (C, C), "w"))
            out.append((L + f"self_attn.{nm}.bias", (C,), "b"))
        out += [(L + "layer_norm1.weight", (C,), "gamma"), (L + "layer_norm1.bias", (C,), "beta")]
        out += [(L + "mlp.fc1.weight", (F, C), "w"), (L + "mlp.fc1.bias", (F,), "b")]
        out += [(L + "mlp.fc2.weight", (C, F), "w"), (L + "mlp.fc2.bias", (C,), "b")]
        out += [(L + "layer_norm2.weight", (C,), "gamma"), (L + "layer_norm2.bias", (C,), "beta")]
    out += [(prefix + "final_layer_norm.weight", (C,), "gamma"), (prefix + "final_layer_norm.bias", (C,), "beta")]
    return out


def synth_text_state_dict(cfg: ModelConfig, seed: int = 1234) -> Dict[str, np.ndarray]:
    return {n: synth_tensor(n, s, k, seed) for n, s, k in text_spec(cfg)}


def synth_token_ids(cfg: ModelConfig, batch: int, seed: int = 7) -> np.ndarray:
    """Synthetic prompts: BOS, a run of random tokens, EOS padding to context_len (CLIPTokenizer pads with EOS for
    openai/clip-vit-large-patch14); int32 [batch, context_len]."""
    g = np.random.default_rng(seed)
    bos, eos = cfg.text_vocab - 2, cfg.text_vocab - 1
    ids = np.full((batch, cfg.context_len), eos, np.int32)
    ids[:, 0] = bos
    for b in range(batch):
        n = int(g.integers(3, cfg.context_len - 2))
        ids[b, 1:1 + n] = g.integers(0, cfg.text_vocab - 2, n)
    return ids


def num_params(cfg: ModelConfig) -> Tuple[int, int]:
    u = sum(int(np.prod(s)) for _, s, _ in unet_spec(cfg))
    c = sum(int(np.prod(s)) for _, s, _ in controlnet_spec(cfg))
    return u, c


def synth_tensor(name: str, shape: Sequence[int], kind: str, seed: int = 1234) -> np.ndarray:
    """Seeded value for one tensor (SURVEY.md §8d recipe).

    w: N(0, 1/fan_in); b: N(0, 0.02^2); gamma: 1 + N(0, 0.1^2); beta: N(0, 0.1^2).
    The reference's zero-initialised tensors (zero convs, ResBlock out conv,
    proj_out, hint-block last conv) get ordinary 'w'/'b' values,
    otherwise eps would be identically zero.
    """
    key = zlib.crc32(name.encode())
    rng = np.random.Generator(np.random.Philox(key=[seed, key]))
    x = rng.standard_normal(size=tuple(shape), dtype=np.float32)
    if kind == "w":
        fan_in = int(np.prod(shape[1:]))
        x *= np.float32(1.0 / np.sqrt(fan_in))
    elif kind == "b":
        x *= np.float32(0.02)
    elif kind == "gamma":
        x = np.float32(1.0) + np.float32(0.1) * x
    elif kind == "beta":
        x *= np.float32(0.1)
    else:
        raise ValueError(kind)
    return x


def synth_state_dict(cfg: ModelConfig, seed: int = 1234) -> Dict[str, np.ndarray]:
    return {n: synth_tensor(n, s, k, seed) for n, s, k in param_spec(cfg)}


def iter_synth(cfg: ModelConfig, seed: int = 1234):
    """Stream (name, array) pairs without holding the whole model in host memory."""
    for n, s, k in param_spec(cfg):
        yield n, synth_tensor(n, s, k, seed)


def synth_inputs(cfg: ModelConfig, batch: int, h: int, w: int, seed: int = 2023,
                 unit_range: bool = False) -> Dict[str, np.ndarray]:
    """Synthetic call inputs: x_T ~ N(0,1) seed 2023; ctx ~ N(0,1);
    pair/query ~ U(-1,1) ((L) convention) or U(0,1) ((D) convention)."""
    def g(tag):
        return np.random.Generator(np.random.Philox(key=[seed, zlib.crc32(tag.encode())]))
    lo = 0.0 if unit_range else -1.0
    return dict(
        x_T=g("x_T").standard_normal((batch, cfg.in_channels, h, w), dtype=np.float32),
        ctx_cond=g("ctx_cond").standard_normal((batch, cfg.context_len, cfg.context_dim), dtype=np.float32),
        ctx_uncond=g("ctx_uncond").standard_normal((batch, cfg.context_len, cfg.context_dim), dtype=np.float32),
        pair=g("pair").uniform(lo, 1.0, (batch, cfg.hint_channels, 8 * h, 8 * w)).astype(np.float32),
        query=g("query").uniform(lo, 1.0, (batch, cfg.query_channels, 8 * h, 8 * w)).astype(np.float32),
    )
