"""Host side of the SD3 / MMDiT variant of the path over the C ABI (pd_sd3_* in include/pdengine.h).

Mirrors, at the tensor level, what the reference's SD3 pipeline does per denoising step
(promptdiffusioncontrolnetpipeline_sd3.py:1192-1245): ControlNet (promptdiffusioncontrolnet_sd3.py:362-483), transformer
with ``block_controlnet_hidden_states``, classifier-free guidance, FlowMatchEuler step.  Text encoders (CLIP-L, CLIP-G, T5),
the VAE and ``down_proj`` / ``encode_support_pair`` stay with the caller: the boundary takes prompt embeddings and condition
LATENTS.  PARITY UNPINNED (diffusers is absent offline): checked against oracle/sd3_oracle.py only."""
import ctypes as C
import math
from dataclasses import dataclass
from typing import Dict, Optional

import numpy as np

from . import engine as E
from . import weights as W


@dataclass(frozen=True)
class SD3Config:
    in_channels: int = 16
    out_channels: int = 16
    patch: int = 2
    heads: int = 24
    head_dim: int = 64
    layers: int = 24
    cn_layers: int = 6              # the reference class defaults to 18 (promptdiffusioncontrolnet_sd3.py:95); SD3 ControlNet
                                    # checkpoints ship 6 or 12 blocks
    joint_dim: int = 4096
    pooled_dim: int = 2048
    pos_embed_max_size: int = 192
    cn_pos_embed_max_size: int = 0  # 0: same as the transformer's
    force_zeros_for_pooled_projection: bool = True   # the reference class's default (promptdiffusioncontrolnet_sd3.py:108)
    # SD3.5-style blocks (promptdiffusioncontrolnet_sd3.py:104-105, :140-141): per-head RMSNorm of queries / keys, and a second,
    # image-only attention in the listed blocks (own lists for the transformer and the ControlNet)
    qk_norm: Optional[str] = None            # None or "rms_norm"
    dual_attention_layers: tuple = ()        # transformer blocks with attn2
    cn_dual_attention_layers: tuple = ()     # ControlNet blocks with attn2 (the reference class's dual_attention_layers)
    cn_single_blocks: bool = False           # the ControlNet built with joint_attention_dim=None: SD3SingleTransformerBlock, no
                                             # context stream / context_embedder (promptdiffusioncontrolnet_sd3.py:147-160)

    @property
    def hidden(self) -> int:
        return self.heads * self.head_dim


SD3_MEDIUM = SD3Config()
SD3_TINY = SD3Config(in_channels=4, out_channels=4, heads=2, head_dim=64, layers=3, cn_layers=2, joint_dim=96, pooled_dim=40,
                     pos_embed_max_size=12, cn_pos_embed_max_size=10)


class pd_sd3_config(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("in_channels", "out_channels", "patch_size", "heads", "head_dim", "layers", "cn_layers",
                                         "joint_dim", "pooled_dim", "pos_embed_max_size", "cn_pos_embed_max_size",
                                         "cn_zero_pooled")] + \
               [("qk_norm", C.c_int32), ("dual_mask", C.c_uint32), ("cn_dual_mask", C.c_uint32), ("cn_single", C.c_int32)]


class pd_sd3_args(C.Structure):
    _fields_ = [("batch", C.c_int32), ("height", C.c_int32), ("width", C.c_int32), ("context_len", C.c_int32), ("mem", C.c_int32),
                ("conditioning_scale", C.c_float),
                ("latents", C.c_void_p), ("timestep", C.c_void_p), ("context", C.c_void_p), ("pooled", C.c_void_p),
                ("cond", C.c_void_p), ("pair", C.c_void_p), ("cn_pooled", C.c_void_p), ("reserved", C.c_int64 * 3)]


def flow_match_sigmas(steps: int, shift: float = 3.0, num_train_timesteps: int = 1000) -> np.ndarray:
    """FlowMatchEulerDiscreteScheduler.set_timesteps(num_inference_steps) of the scheduler the reference pipeline holds
    (promptdiffusioncontrolnetpipeline_sd3.py:1086-1088): sigmas[steps + 1], descending, last 0; timestep_i = 1000 sigma_i."""
    shifted = lambda s: shift * s / (1.0 + (shift - 1.0) * s)
    ts = np.linspace(shifted(1.0) * num_train_timesteps, shifted(1.0 / num_train_timesteps) * num_train_timesteps, steps)
    return np.concatenate([shifted(ts / num_train_timesteps), [0.0]]).astype(np.float32)


def sincos_pos_embed(dim: int, grid: int, base_size: int, interpolation_scale: float = 1.0) -> np.ndarray:
    """The 2-D sin/cos table PatchEmbed registers as `pos_embed` ([grid * grid, dim]; first half of the channels encodes the
    column, second half the row) -- real checkpoints carry it as a tensor; this is for synthetic weights."""
    def axis(pos, d):
        omega = 1.0 / 10000 ** (np.arange(d // 2, dtype=np.float64) / (d / 2.0))
        ang = pos.reshape(-1)[:, None] * omega[None]
        return np.concatenate([np.sin(ang), np.cos(ang)], axis=1)
    g = np.arange(grid, dtype=np.float64) / (grid / base_size) / interpolation_scale
    col, row = np.meshgrid(g, g)
    return np.concatenate([axis(col, dim // 2), axis(row, dim // 2)], axis=1).astype(np.float32)


def sd3_param_shapes(cfg: SD3Config) -> Dict[str, tuple]:
    """diffusers state-dict names and shapes of both networks ("transformer." / "controlnet." prefixes)."""
    D, out = cfg.hidden, {}

    def lin(name, n, k):
        out[name + ".weight"], out[name + ".bias"] = (n, k), (n,)

    for net, layers, cn in (("transformer.", cfg.layers, False), ("controlnet.", cfg.cn_layers, True)):
        if layers == 0:
            continue
        pm = cfg.cn_pos_embed_max_size if cn and cfg.cn_pos_embed_max_size else cfg.pos_embed_max_size
        out[net + "pos_embed.proj.weight"], out[net + "pos_embed.proj.bias"] = (D, cfg.in_channels, cfg.patch, cfg.patch), (D,)
        out[net + "pos_embed.pos_embed"] = (1, pm * pm, D)
        if cn:
            out[net + "pos_embed_input.proj.weight"] = (D, cfg.in_channels, cfg.patch, cfg.patch)
            out[net + "pos_embed_input.proj.bias"] = (D,)
            out[net + "down_proj.weight"], out[net + "down_proj.bias"] = (3, 6, 3, 3), (3,)      # promptdiffusioncontrolnet_sd3.py:114
        lin(net + "time_text_embed.timestep_embedder.linear_1", D, 256)
        lin(net + "time_text_embed.timestep_embedder.linear_2", D, D)
        lin(net + "time_text_embed.text_embedder.linear_1", D, cfg.pooled_dim)
        lin(net + "time_text_embed.text_embedder.linear_2", D, D)
        single = cn and cfg.cn_single_blocks
        if not single:
            lin(net + "context_embedder", D, cfg.joint_dim)
        for i in range(layers):
            b = f"{net}transformer_blocks.{i}."
            pre_only = (not cn) and i == layers - 1
            dual = (not single) and i in tuple(cfg.cn_dual_attention_layers if cn else cfg.dual_attention_layers)
            lin(b + "norm1.linear", (9 if dual else 6) * D, D)
            if not single:
                lin(b + "norm1_context.linear", (2 if pre_only else 6) * D, D)
            for n in ("to_q", "to_k", "to_v", "to_out.0") + (() if single else ("add_q_proj", "add_k_proj", "add_v_proj")):
                lin(b + "attn." + n, D, D)
            if cfg.qk_norm:
                for n in ("norm_q", "norm_k") + (() if single else ("norm_added_q", "norm_added_k")):
                    out[b + "attn." + n + ".weight"] = (cfg.head_dim,)
            if dual:
                for n in ("to_q", "to_k", "to_v", "to_out.0"):
                    lin(b + "attn2." + n, D, D)
                if cfg.qk_norm:
                    out[b + "attn2.norm_q.weight"], out[b + "attn2.norm_k.weight"] = (cfg.head_dim,), (cfg.head_dim,)
            lin(b + "ff.net.0.proj", 4 * D, D)
            lin(b + "ff.net.2", D, 4 * D)
            if not pre_only and not single:
                lin(b + "attn.to_add_out", D, D)
                lin(b + "ff_context.net.0.proj", 4 * D, D)
                lin(b + "ff_context.net.2", D, 4 * D)
            if cn:
                lin(f"{net}controlnet_blocks.{i}", D, D)
        if not cn:
            lin(net + "norm_out.linear", 2 * D, D)
            lin(net + "proj_out", cfg.patch * cfg.patch * cfg.out_channels, D)
    return out


def synth_sd3_state_dict(cfg: SD3Config, seed: int = 0) -> Dict[str, np.ndarray]:
    """Deterministic non-degenerate weights (the zero-initialised modules get small non-zero values so that every branch
    contributes); pos_embed.pos_embed is the real sin/cos table."""
    rng = np.random.default_rng(seed)
    sd = {}
    for name, shp in sd3_param_shapes(cfg).items():
        if name.endswith("pos_embed.pos_embed"):
            m = int(round(math.sqrt(shp[1])))
            sd[name] = sincos_pos_embed(shp[2], m, base_size=max(1, m // 2))[None].astype(np.float32)
        elif name.endswith(".bias"):
            sd[name] = (0.05 * rng.standard_normal(shp)).astype(np.float32)
        elif len(shp) == 1:                                   # RMSNorm weights of qk_norm
            sd[name] = (1.0 + 0.1 * rng.standard_normal(shp)).astype(np.float32)
        else:
            fan_in = int(np.prod(shp[1:]))
            scale = 0.5 if "norm1" in name or "norm_out" in name else 1.0     # keep the modulation gates moderate
            sd[name] = (scale * rng.standard_normal(shp) / math.sqrt(fan_in)).astype(np.float32)
    return sd


class SD3Engine:
    """The SD3 networks on one engine (one GPU, one stream).  NumPy arrays or CUDA torch tensors in, the same kind out."""

    def __init__(self, cfg: SD3Config = SD3_MEDIUM, device: int = 0, precision: str = "f16", stream_f32: bool = False,
                 lib_path: Optional[str] = None, fp8: bool = False):
        """fp8 (2-byte modes; BASELINE config #5 "fp8 MFMA"): the projections fed by an AdaLN output -- q/k/v of both
        streams and ff / ff_context net.0 -- take e4m3 operands with one scale per token and per output channel and run on
        the block-scaled K = 128 MFMA (twice the f16 rate); level 2 (= True) also the feed-forward-out projections, whose
        input (the GELU output) is stored as e4m3 under a norm bound.  fp8=1: the first group only.  Everything else stays
        in `precision`."""
        if cfg.qk_norm not in (None, "rms_norm"):
            raise NotImplementedError(f"qk_norm={cfg.qk_norm!r}: only None and 'rms_norm' are built (promptdiffusioncontrolnet_sd3.py:105)")
        for lst, n in ((cfg.dual_attention_layers, cfg.layers), (cfg.cn_dual_attention_layers, cfg.cn_layers)):
            if any(i < 0 or i >= min(n, 32) for i in lst):
                raise ValueError("dual_attention_layers index out of range")
        if cfg.cn_single_blocks and tuple(cfg.cn_dual_attention_layers):
            raise ValueError("SD3SingleTransformerBlock has no second attention: cn_dual_attention_layers must be empty with cn_single_blocks")
        if cfg.layers - 1 in tuple(cfg.dual_attention_layers):
            raise NotImplementedError("the context_pre_only last block cannot carry a second attention")
        self.cfg = cfg
        self.base = E.Engine(W.TINY, device=device, precision=precision, stream_f32=stream_f32, lib_path=lib_path)
        lib = self.base.lib
        lib.pd_sd3_configure.argtypes = [C.c_void_p, C.POINTER(pd_sd3_config)]
        lib.pd_sd3_weights_missing.argtypes = [C.c_void_p]
        lib.pd_sd3_forward.argtypes = [C.c_void_p, C.POINTER(pd_sd3_args), C.c_void_p]
        lib.pd_sd3_control.argtypes = [C.c_void_p, C.POINTER(pd_sd3_args), C.c_int32, C.c_void_p]
        lib.pd_sd3_sample.argtypes = [C.c_void_p, C.POINTER(pd_sd3_args), C.c_void_p, C.c_int32, C.c_float, C.c_void_p, C.c_void_p]
        lib.pd_sd3_down_proj.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
        c = pd_sd3_config(cfg.in_channels, cfg.out_channels, cfg.patch, cfg.heads, cfg.head_dim, cfg.layers, cfg.cn_layers,
                          cfg.joint_dim, cfg.pooled_dim, cfg.pos_embed_max_size, cfg.cn_pos_embed_max_size,
                          1 if cfg.force_zeros_for_pooled_projection else 0, 1 if cfg.qk_norm else 0,
                          sum(1 << i for i in cfg.dual_attention_layers), sum(1 << i for i in cfg.cn_dual_attention_layers),
                          1 if cfg.cn_single_blocks else 0)
        self.base._check(lib.pd_sd3_configure(self.base._h, C.byref(c)))
        self.fp8 = 2 if fp8 is True else int(fp8)
        if self.fp8:
            if precision in ("f32", "f16x2"):
                raise ValueError("fp8 needs a 2-byte engine precision (f16 / bf16)")
            self.base.set_option("sd3_fp8", self.fp8)

    def close(self):
        self.base.close()

    # ------------------------------------------------------------------ weights
    def load_state_dict(self, sd, strict: bool = True) -> None:
        known = {n for n, _ in self.base.param_names()}
        unexpected = []
        for name, arr in (sd.items() if isinstance(sd, dict) else sd):
            if not (name.startswith("transformer.") or name.startswith("controlnet.")):
                continue
            if name in known:
                self.base.load_tensor(name, arr)
            else:
                unexpected.append(name)   # e.g. attn.norm_q / attn2.* of an SD3.5 checkpoint: a different network
        if strict and unexpected:
            raise E.PdError(f"{len(unexpected)} unexpected SD3 tensors (first: {unexpected[0]}): the checkpoint describes blocks "
                            "this engine does not implement (qk_norm / dual_attention_layers?)")
        if strict and self.weights_missing():
            raise E.PdError(f"{self.weights_missing()} SD3 tensors missing after load_state_dict")

    def init_random_weights(self, seed: int = 1234) -> None:
        self.base.init_random_weights(seed)

    def encode_support_pair(self, cond, gt, vae=None):
        """SD3PromptDiffusionModel.encode_support_pair (promptdiffusioncontrolnet_sd3.py:189-198): down_proj (Conv2d 6 -> 3, 3x3)
        over cat([cond, gt], 1) on the engine; with a `vae` the caller's encoder turns the result into latents
        (``vae.encode(x).latent_dist.sample()``), as the reference does."""
        b0, b1 = E._Buf(cond), E._Buf(gt)
        if b0.mem != b1.mem or tuple(b0.owner.shape) != tuple(b1.owner.shape) or b0.owner.shape[1] != 3:
            raise ValueError("cond and gt must be [B, 3, H, W] in the same memory space")
        Bn, _, H, Wd = b0.owner.shape
        if b0.mem == E.PD_MEM_DEVICE:
            import torch
            pair = torch.cat([b0.owner, b1.owner], 1).contiguous()
            out = torch.empty((Bn, 3, H, Wd), dtype=torch.float32, device=pair.device)
            self.base._order_after_torch(b0.mem)
            self.base._check(self.base.lib.pd_sd3_down_proj(self.base._h, pair.data_ptr(), Bn, H, Wd, b0.mem, out.data_ptr()))
        else:
            pair = np.ascontiguousarray(np.concatenate([b0.owner, b1.owner], 1), np.float32)
            out = np.empty((Bn, 3, H, Wd), np.float32)
            self.base._check(self.base.lib.pd_sd3_down_proj(self.base._h, pair.ctypes.data, Bn, H, Wd, b0.mem, out.ctypes.data))
        if vae is not None:
            return vae.encode(out).latent_dist.sample()
        return out

    def weights_missing(self) -> int:
        return int(self.base.lib.pd_sd3_weights_missing(self.base._h))

    # ------------------------------------------------------------------ calls
    def _args(self, latents, context, pooled, cond, pair, scale, timestep=None, rows=None, cn_pooled=None):
        bufs = [E._Buf(x) for x in (latents, context, pooled, cond, pair, cn_pooled)]
        mems = {b.mem for b in bufs if b.mem is not None}
        if len(mems) != 1:
            raise ValueError("all tensors of one call must live in the same memory space (NumPy / CPU or one CUDA device)")
        mem = mems.pop()
        lat = bufs[0].owner
        B, _, H, Wd = lat.shape
        rows = rows or B
        if tuple(bufs[1].owner.shape[::2]) != (rows, self.cfg.joint_dim) or tuple(bufs[2].owner.shape) != (rows, self.cfg.pooled_dim):
            raise ValueError(f"context must be [{rows}, S, {self.cfg.joint_dim}] and pooled [{rows}, {self.cfg.pooled_dim}]")
        for b in bufs[3:5]:
            if b.owner is not None and tuple(b.owner.shape) != tuple(lat.shape):
                raise ValueError("cond / pair latents must have the shape of latents")
        if bufs[5].owner is not None and tuple(bufs[5].owner.shape) != (rows, self.cfg.pooled_dim):
            raise ValueError(f"controlnet_pooled_projections must be [{rows}, {self.cfg.pooled_dim}]")
        a = pd_sd3_args()
        a.batch, a.height, a.width, a.context_len, a.mem = B, H, Wd, bufs[1].owner.shape[1], mem
        a.conditioning_scale = float(scale)
        a.latents, a.context, a.pooled, a.cond, a.pair, a.cn_pooled = (b.ptr for b in bufs)
        keep = bufs
        if timestep is not None:
            t = np.ascontiguousarray(np.broadcast_to(np.asarray(E._to_host(timestep), np.float32), (B,)))
            a.timestep = t.ctypes.data
            keep = bufs + [t]
        self.base._order_after_torch(mem)
        return a, keep, mem, lat

    def _out(self, mem, like, shape):
        if mem == E.PD_MEM_DEVICE:
            import torch
            o = torch.empty(shape, dtype=torch.float32, device=like.device)
            return o, o.data_ptr()
        o = np.empty(shape, np.float32)
        return o, o.ctypes.data

    def forward(self, latents, timestep, context, pooled, cond=None, pair=None, conditioning_scale: float = 1.0,
                controlnet_pooled_projections=None):
        """One evaluation: transformer(latents, timestep, context, pooled | ControlNet(cond, pair) residuals) -> velocity.
        The ControlNet's pooled projections are zeros under force_zeros_for_pooled_projection (the reference default), else
        `controlnet_pooled_projections`, else `pooled` (pipeline :1164-1168)."""
        a, keep, mem, lat = self._args(latents, context, pooled, cond, pair, conditioning_scale, timestep,
                                       cn_pooled=controlnet_pooled_projections)
        out, ptr = self._out(mem, lat, (a.batch, self.cfg.out_channels, a.height, a.width))
        self.base._check(self.base.lib.pd_sd3_forward(self.base._h, C.byref(a), ptr))
        return out

    def controlnet(self, latents, timestep, context, pooled, cond, pair, conditioning_scale: float = 1.0):
        """SD3PromptDiffusionModel.forward: the list of cn_layers scaled residuals [B, N, hidden].  `pooled` is what the
        ControlNet itself is given (the pipeline passes zeros under force_zeros_for_pooled_projection)."""
        res = []
        for i in range(self.cfg.cn_layers):
            a, keep, mem, lat = self._args(latents, context, pooled, cond, pair, conditioning_scale, timestep, cn_pooled=pooled)
            n = (a.height // self.cfg.patch) * (a.width // self.cfg.patch)
            out, ptr = self._out(mem, lat, (a.batch, n, self.cfg.hidden))
            self.base._check(self.base.lib.pd_sd3_control(self.base._h, C.byref(a), i, ptr))
            res.append(out)
        return res

    def sample(self, latents, prompt_embeds, pooled_prompt_embeds, negative_prompt_embeds=None, negative_pooled_prompt_embeds=None,
               control_latents=None, pair_latents=None, num_inference_steps: int = 28, guidance_scale: float = 7.0,
               controlnet_conditioning_scale: float = 1.0, shift: float = 3.0, sigmas=None, control_guidance_start: float = 0.0,
               control_guidance_end: float = 1.0, controlnet_pooled_projections=None):
        """The denoising loop of the reference's __call__ (promptdiffusioncontrolnetpipeline_sd3.py:1192-1245) from initial
        noise `latents` to final latents.  guidance_scale > 1 needs the negative embeddings (batch [negative ; positive])."""
        cfg_on = guidance_scale > 1.0
        if cfg_on and (negative_prompt_embeds is None or negative_pooled_prompt_embeds is None):
            raise ValueError("guidance_scale > 1 needs negative_prompt_embeds and negative_pooled_prompt_embeds")
        if cfg_on:
            if E._is_torch(prompt_embeds):
                import torch
                ctx = torch.cat([negative_prompt_embeds, prompt_embeds], 0)
                pooled = torch.cat([negative_pooled_prompt_embeds, pooled_prompt_embeds], 0)
            else:
                ctx = np.concatenate([negative_prompt_embeds, prompt_embeds], 0)
                pooled = np.concatenate([negative_pooled_prompt_embeds, pooled_prompt_embeds], 0)
        else:
            ctx, pooled = prompt_embeds, pooled_prompt_embeds
        sig = flow_match_sigmas(num_inference_steps, shift) if sigmas is None else np.ascontiguousarray(sigmas, np.float32)
        if sig.shape != (num_inference_steps + 1,):
            raise ValueError("sigmas must hold num_inference_steps + 1 values")
        B = latents.shape[0]
        a, keep, mem, lat = self._args(latents, ctx, pooled, control_latents, pair_latents, controlnet_conditioning_scale,
                                       rows=2 * B if cfg_on else B, cn_pooled=controlnet_pooled_projections)
        # controlnet_keep (pipeline :1155-1162): the ControlNet acts on the steps inside [start, end] of the schedule
        n = num_inference_steps
        keep_steps = np.array([1.0 - float(i / n < control_guidance_start or (i + 1) / n > control_guidance_end) for i in range(n)],
                              np.float32)
        scales = np.ascontiguousarray(keep_steps * np.float32(controlnet_conditioning_scale))
        out, ptr = self._out(mem, lat, tuple(lat.shape))
        self.base._check(self.base.lib.pd_sd3_sample(self.base._h, C.byref(a), sig.ctypes.data, num_inference_steps,
                                                     float(guidance_scale), scales.ctypes.data, ptr))
        return out
