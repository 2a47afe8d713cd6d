"""Host-side mirror of the reference's SD3 pipeline surface over the engine's SD3 path.

`StableDiffusion3PromptDiffusionPipeline.__call__` keeps the keyword names, defaults, argument normalisation and error
behaviour of promptdiffusioncontrolnetpipeline_sd3.py:853-1283 for everything that concerns the denoising path; what the
reference does with modules outside the path is injected as callables (the boundary of DESIGN.md §7 N4):

  encode_prompt(prompt, prompt_2, prompt_3, negative_prompt, negative_prompt_2, negative_prompt_3, do_classifier_free_guidance,
                num_images_per_prompt, clip_skip, max_sequence_length)
        -> (prompt_embeds, negative_prompt_embeds, pooled_prompt_embeds, negative_pooled_prompt_embeds)   (:1059-1082)
  vae_encode(images [B, 3, H, W] in [-1, 1]) -> latents [B, 16, H/8, W/8]      (vae.encode(...).latent_dist.sample(), :1126)
  vae_decode(latents) -> images in [-1, 1]                                       (vae.decode, :1270)
  down_proj(pair [B, 6, H, W]) -> [B, 3, H, W]                                   (controlnet.down_proj, encode_support_pair :189-198)

The three steps of the loop body (controlnet, transformer, CFG + scheduler.step; :1192-1245) run inside the engine
(`SD3Engine.sample`); with `callback_on_step_end` the loop is driven step by step from here (`SD3Engine.forward` + the same
Euler update) so that the callback can replace the latents or the embeddings (:1247-1258).  Arrays are NumPy (torch tensors
are converted).  Not supported, raising NotImplementedError like the SD1.5 mirror does for options outside the path:
`ip_adapter_image(_embeds)`, `joint_attention_kwargs` (LoRA scale)."""
from typing import Any, Callable, Dict, List, Optional, Union

import numpy as np

from . import engine as E
from .sd3 import SD3Engine, flow_match_sigmas

_CALLBACK_TENSOR_INPUTS = ["latents", "prompt_embeds", "negative_prompt_embeds", "negative_pooled_prompt_embeds"]


def _np(x, dtype=np.float32):
    return None if x is None else np.ascontiguousarray(E._to_host(x), dtype=dtype)


class StableDiffusion3PromptDiffusionPipeline:
    _callback_tensor_inputs = _CALLBACK_TENSOR_INPUTS

    def __init__(self, engine: SD3Engine, encode_prompt: Optional[Callable] = None, vae_encode: Optional[Callable] = None,
                 vae_decode: Optional[Callable] = None, down_proj: Optional[Callable] = None, vae_scaling_factor: float = 1.5305,
                 vae_shift_factor: float = 0.0609, vae_scale_factor: int = 8, shift: float = 3.0):
        self.engine = engine
        self.encode_prompt, self.vae_encode, self.vae_decode, self.down_proj = encode_prompt, vae_encode, vae_decode, down_proj
        self.vae_scaling_factor, self.vae_shift_factor, self.vae_scale_factor = vae_scaling_factor, vae_shift_factor, vae_scale_factor
        self.shift = shift
        self._guidance_scale, self._clip_skip, self._num_timesteps, self._interrupt = 7.0, None, 0, False

    # ------------------------------------------------------------------ properties of the reference pipeline
    @property
    def guidance_scale(self):
        return self._guidance_scale

    @property
    def clip_skip(self):
        return self._clip_skip

    @property
    def do_classifier_free_guidance(self):
        return self._guidance_scale > 1

    @property
    def num_timesteps(self):
        return self._num_timesteps

    @property
    def interrupt(self):
        return self._interrupt

    # ------------------------------------------------------------------ check_inputs (:541-633)
    def check_inputs(self, prompt, prompt_2, prompt_3, height, width, negative_prompt=None, negative_prompt_2=None,
                     negative_prompt_3=None, prompt_embeds=None, negative_prompt_embeds=None, pooled_prompt_embeds=None,
                     negative_pooled_prompt_embeds=None, callback_on_step_end_tensor_inputs=None, max_sequence_length=None):
        if height % 8 != 0 or width % 8 != 0:
            raise ValueError(f"`height` and `width` have to be divisible by 8 but are {height} and {width}.")
        if callback_on_step_end_tensor_inputs is not None and not all(
                k in self._callback_tensor_inputs for k in callback_on_step_end_tensor_inputs):
            bad = [k for k in callback_on_step_end_tensor_inputs if k not in self._callback_tensor_inputs]
            raise ValueError(f"`callback_on_step_end_tensor_inputs` has to be in {self._callback_tensor_inputs}, but found {bad}")
        for name, p in (("prompt", prompt), ("prompt_2", prompt_2), ("prompt_3", prompt_3)):
            if p is not None and prompt_embeds is not None:
                raise ValueError(f"Cannot forward both `{name}`: {p} and `prompt_embeds`. Please make sure to only forward one of the two.")
        if prompt is None and prompt_embeds is None:
            raise ValueError("Provide either `prompt` or `prompt_embeds`. Cannot leave both `prompt` and `prompt_embeds` undefined.")
        for name, p in (("prompt", prompt), ("prompt_2", prompt_2), ("prompt_3", prompt_3)):
            if p is not None and not isinstance(p, (str, list)):
                raise ValueError(f"`{name}` has to be of type `str` or `list` but is {type(p)}")
        for name, p in (("negative_prompt", negative_prompt), ("negative_prompt_2", negative_prompt_2), ("negative_prompt_3", negative_prompt_3)):
            if p is not None and negative_prompt_embeds is not None:
                raise ValueError(f"Cannot forward both `{name}`: {p} and `negative_prompt_embeds`. Please make sure to only forward one of the two.")
        if prompt_embeds is not None and negative_prompt_embeds is not None and tuple(prompt_embeds.shape) != tuple(negative_prompt_embeds.shape):
            raise ValueError("`prompt_embeds` and `negative_prompt_embeds` must have the same shape when passed directly, but got: "
                             f"`prompt_embeds` {tuple(prompt_embeds.shape)} != `negative_prompt_embeds` {tuple(negative_prompt_embeds.shape)}.")
        if prompt_embeds is not None and pooled_prompt_embeds is None:
            raise ValueError("If `prompt_embeds` are provided, `pooled_prompt_embeds` also have to be passed.")
        if negative_prompt_embeds is not None and negative_pooled_prompt_embeds is None:
            raise ValueError("If `negative_prompt_embeds` are provided, `negative_pooled_prompt_embeds` also have to be passed.")
        if max_sequence_length is not None and max_sequence_length > 512:
            raise ValueError(f"`max_sequence_length` cannot be greater than 512 but is {max_sequence_length}")

    # ------------------------------------------------------------------ host pieces of steps 3 and 5
    def prepare_image(self, image, batch_size, num_images_per_prompt):
        """prepare_image (:666-698) without the CFG doubling (the engine doubles internally): torch tensors pass unchanged
        like in the reference; other arrays ([B, H, W, 3] or [B, 3, H, W] in [0, 1]) get VaeImageProcessor's normalisation to
        [-1, 1].  Resizing is the caller's."""
        as_is = E._is_torch(image)
        img = _np(image)
        if img.ndim == 3:
            img = img[None]
        if img.shape[-1] == 3 and img.shape[1] != 3:
            img = img.transpose(0, 3, 1, 2)
        if not as_is:
            img = 2.0 * img - 1.0
        repeat_by = batch_size if img.shape[0] == 1 else num_images_per_prompt
        return np.ascontiguousarray(np.repeat(img, repeat_by, axis=0), np.float32)

    def _to_latents(self, x, batch_size, num_images_per_prompt, shift, pair=False):
        """Control conditions: 16-channel arrays are taken as latents already; images go through (down_proj +) vae_encode and
        (x - shift_factor) * scaling_factor (:1112-1127)."""
        C = self.engine.cfg.in_channels
        if not pair:
            a = _np(x)
            if a.ndim == 4 and a.shape[1] == C:
                return np.ascontiguousarray(np.repeat(a, batch_size * num_images_per_prompt if a.shape[0] == 1 else num_images_per_prompt, axis=0))
            if self.vae_encode is None:
                raise ValueError(f"`control_image` must be [{C}]-channel latents unless the pipeline was given `vae_encode`")
            lat = _np(self.vae_encode(self.prepare_image(a, batch_size * num_images_per_prompt, num_images_per_prompt)))
            return ((lat - shift) * self.vae_scaling_factor).astype(np.float32)
        if not isinstance(x, (list, tuple)):
            a = _np(x)
            if a.ndim == 4 and a.shape[1] == C:
                return np.ascontiguousarray(np.repeat(a, batch_size * num_images_per_prompt if a.shape[0] == 1 else num_images_per_prompt, axis=0))
            raise ValueError("`control_image_pair` must be a list of two images (or pair latents)")
        if len(x) != 2:
            raise ValueError("`control_image_pair` must hold exactly two images")
        if self.vae_encode is None or self.down_proj is None:
            raise ValueError("image pairs need `vae_encode` and `down_proj`; pass pair latents otherwise")
        imgs = [self.prepare_image(i, batch_size * num_images_per_prompt, num_images_per_prompt) for i in x]
        lat = _np(self.vae_encode(_np(self.down_proj(np.concatenate(imgs, axis=1)))))      # encode_support_pair (:189-198)
        return ((lat - shift) * self.vae_scaling_factor).astype(np.float32)

    def prepare_latents(self, batch_size, num_channels_latents, height, width, generator, latents=None):
        if latents is not None:
            return _np(latents)
        shape = (batch_size, num_channels_latents, int(height) // self.vae_scale_factor, int(width) // self.vae_scale_factor)
        if isinstance(generator, list) and len(generator) != batch_size:
            raise ValueError(f"You have passed a list of generators of length {len(generator)}, but requested an effective batch"
                             f" size of {batch_size}. Make sure the batch size matches the length of the generators.")
        if generator is None:
            generator = np.random.default_rng()
        if isinstance(generator, list):
            return np.concatenate([self._randn(g, (1,) + shape[1:]) for g in generator], 0)
        return self._randn(generator, shape)

    @staticmethod
    def _randn(g, shape):
        if isinstance(g, np.random.Generator):
            return g.standard_normal(shape).astype(np.float32)
        import torch   # a torch.Generator, as the reference takes
        return torch.randn(shape, generator=g, device=g.device).float().cpu().numpy()

    def _sigmas(self, num_inference_steps, sigmas):
        """retrieve_timesteps(scheduler, num_inference_steps, sigmas=sigmas) on FlowMatchEulerDiscreteScheduler: custom sigmas
        are shifted like the default grid and terminated with 0."""
        if sigmas is None:
            return flow_match_sigmas(num_inference_steps, self.shift), num_inference_steps
        s = np.asarray(sigmas, np.float64)
        s = self.shift * s / (1.0 + (self.shift - 1.0) * s)
        return np.concatenate([s, [0.0]]).astype(np.float32), len(s)

    # ------------------------------------------------------------------ __call__ (:853-1283)
    def __call__(self, prompt: Union[str, List[str]] = None, prompt_2=None, prompt_3=None, height: Optional[int] = None,
                 width: Optional[int] = None, num_inference_steps: int = 28, sigmas: Optional[List[float]] = None,
                 guidance_scale: float = 7.0, control_guidance_start: Union[float, List[float]] = 0.0,
                 control_guidance_end: Union[float, List[float]] = 1.0, control_image=None, control_image_pair=None,
                 controlnet_conditioning_scale: Union[float, List[float]] = 1.0, controlnet_pooled_projections=None,
                 negative_prompt=None, negative_prompt_2=None, negative_prompt_3=None, num_images_per_prompt: Optional[int] = 1,
                 generator=None, latents=None, prompt_embeds=None, negative_prompt_embeds=None, pooled_prompt_embeds=None,
                 negative_pooled_prompt_embeds=None, ip_adapter_image=None, ip_adapter_image_embeds=None,
                 output_type: Optional[str] = "pil", return_dict: bool = True, joint_attention_kwargs: Optional[Dict[str, Any]] = None,
                 clip_skip: Optional[int] = None, callback_on_step_end: Optional[Callable] = None,
                 callback_on_step_end_tensor_inputs: List[str] = ["latents"], max_sequence_length: int = 256):
        if ip_adapter_image is not None or ip_adapter_image_embeds is not None:
            raise NotImplementedError("ip_adapter_image / ip_adapter_image_embeds are outside the engine's path")
        if joint_attention_kwargs:
            raise NotImplementedError("joint_attention_kwargs (LoRA scale) is outside the engine's path")
        cfg = self.engine.cfg
        if control_image is None or control_image_pair is None:
            raise ValueError("`control_image` and `control_image_pair` are required (the reference asserts a SD3PromptDiffusionModel)")
        # default size: the control latents / image decide (the reference overrides height, width from the control image, :1123)
        ci = _np(control_image)
        if ci.ndim == 3:
            ci = ci[None]
        if ci.shape[1] == cfg.in_channels:
            h_img, w_img = ci.shape[-2] * self.vae_scale_factor, ci.shape[-1] * self.vae_scale_factor
        elif ci.shape[1] == 3:
            h_img, w_img = ci.shape[-2:]
        else:
            h_img, w_img = ci.shape[1:3]
        height, width = height or h_img, width or w_img
        # align format for control guidance (:1016-1025); the single-ControlNet reference uses element 0
        if isinstance(control_guidance_start, list) or isinstance(control_guidance_end, list):
            if not isinstance(control_guidance_start, list):
                control_guidance_start = len(control_guidance_end) * [control_guidance_start]
            if not isinstance(control_guidance_end, list):
                control_guidance_end = len(control_guidance_start) * [control_guidance_end]
            control_guidance_start, control_guidance_end = control_guidance_start[0], control_guidance_end[0]
        if isinstance(controlnet_conditioning_scale, list):
            controlnet_conditioning_scale = controlnet_conditioning_scale[0]
        # 1. check inputs
        self.check_inputs(prompt, prompt_2, prompt_3, height, width, negative_prompt, negative_prompt_2, negative_prompt_3,
                          prompt_embeds, negative_prompt_embeds, pooled_prompt_embeds, negative_pooled_prompt_embeds,
                          callback_on_step_end_tensor_inputs, max_sequence_length)
        self._guidance_scale, self._clip_skip, self._interrupt = guidance_scale, clip_skip, False
        # 2. call parameters
        if prompt is not None and isinstance(prompt, str):
            batch_size = 1
        elif prompt is not None:
            batch_size = len(prompt)
        else:
            batch_size = prompt_embeds.shape[0]
        cfg_on = self.do_classifier_free_guidance
        if prompt is not None:
            if self.encode_prompt is None:
                raise ValueError("text prompts need the `encode_prompt` callable; pass `prompt_embeds` / `pooled_prompt_embeds` otherwise")
            prompt_embeds, negative_prompt_embeds, pooled_prompt_embeds, negative_pooled_prompt_embeds = self.encode_prompt(
                prompt=prompt, prompt_2=prompt_2, prompt_3=prompt_3, negative_prompt=negative_prompt, negative_prompt_2=negative_prompt_2,
                negative_prompt_3=negative_prompt_3, do_classifier_free_guidance=cfg_on, num_images_per_prompt=num_images_per_prompt,
                clip_skip=clip_skip, max_sequence_length=max_sequence_length)
            rep = 1
        else:
            rep = num_images_per_prompt          # encode_prompt repeats given embeddings per image (:397-399 of diffusers' encode_prompt)
        pe, ppe = np.repeat(_np(prompt_embeds), rep, 0), np.repeat(_np(pooled_prompt_embeds), rep, 0)
        if cfg_on:
            if negative_prompt_embeds is None or negative_pooled_prompt_embeds is None:
                raise ValueError("guidance_scale > 1 with `prompt_embeds` needs `negative_prompt_embeds` and `negative_pooled_prompt_embeds`")
            npe, nppe = np.repeat(_np(negative_prompt_embeds), rep, 0), np.repeat(_np(negative_pooled_prompt_embeds), rep, 0)
        else:
            npe = nppe = None
        B = batch_size * num_images_per_prompt
        # 3. control conditions (no shift factor under force_zeros_for_pooled_projection, :1084-1088)
        vshift = 0.0 if cfg.force_zeros_for_pooled_projection else self.vae_shift_factor
        pair_lat = self._to_latents(control_image_pair, batch_size, num_images_per_prompt, vshift, pair=True)
        cond_lat = self._to_latents(control_image, batch_size, num_images_per_prompt, vshift)
        if cond_lat.shape[0] != B or pair_lat.shape[0] != B:
            raise ValueError(f"control conditions have batch {cond_lat.shape[0]} / {pair_lat.shape[0]}, expected {B}")
        height, width = cond_lat.shape[-2] * self.vae_scale_factor, cond_lat.shape[-1] * self.vae_scale_factor
        # 4. timesteps
        sig, num_inference_steps = self._sigmas(num_inference_steps, sigmas)
        self._num_timesteps = num_inference_steps
        # 5. latents
        latents = self.prepare_latents(B, cfg.in_channels, height, width, generator, latents)
        if tuple(latents.shape) != tuple(cond_lat.shape):
            raise ValueError(f"latents {tuple(latents.shape)} and control latents {tuple(cond_lat.shape)} differ")
        cn_pooled = None
        if not cfg.force_zeros_for_pooled_projection and controlnet_pooled_projections is not None:
            cn_pooled = _np(controlnet_pooled_projections)
        # 8. denoising loop
        if callback_on_step_end is None:
            latents = self.engine.sample(latents, pe, ppe, npe, nppe, control_latents=cond_lat, pair_latents=pair_lat,
                                         num_inference_steps=num_inference_steps, guidance_scale=guidance_scale,
                                         controlnet_conditioning_scale=controlnet_conditioning_scale, sigmas=sig,
                                         control_guidance_start=control_guidance_start, control_guidance_end=control_guidance_end,
                                         controlnet_pooled_projections=cn_pooled)
        else:
            n = num_inference_steps
            for i in range(n):
                if self._interrupt:
                    continue
                keep = 1.0 - float(i / n < control_guidance_start or (i + 1) / n > control_guidance_end)
                t = np.full((2 * B if cfg_on else B,), sig[i] * 1000.0, np.float32)
                x_in = np.concatenate([latents, latents]) if cfg_on else latents
                ctx = np.concatenate([npe, pe]) if cfg_on else pe
                pooled = np.concatenate([nppe, ppe]) if cfg_on else ppe
                dup = (lambda a: np.concatenate([a, a])) if cfg_on else (lambda a: a)
                v = self.engine.forward(x_in, t, ctx, pooled, dup(cond_lat), dup(pair_lat), controlnet_conditioning_scale * keep,
                                        controlnet_pooled_projections=cn_pooled)
                if cfg_on:
                    v = v[:B] + np.float32(guidance_scale) * (v[B:] - v[:B])
                latents = (latents + (sig[i + 1] - sig[i]) * v).astype(np.float32)
                kw = {k: {"latents": latents, "prompt_embeds": pe, "negative_prompt_embeds": npe,
                          "negative_pooled_prompt_embeds": nppe}[k] for k in callback_on_step_end_tensor_inputs}
                out = callback_on_step_end(self, i, float(sig[i] * 1000.0), kw) or {}
                latents = _np(out.pop("latents", latents))
                pe = _np(out.pop("prompt_embeds", pe))
                npe = _np(out.pop("negative_prompt_embeds", npe)) if cfg_on else npe
                nppe = _np(out.pop("negative_pooled_prompt_embeds", nppe)) if cfg_on else nppe
        # post
        if output_type == "latent":
            image = latents
        else:
            if self.vae_decode is None:
                raise ValueError('output_type other than "latent" needs the `vae_decode` callable')
            image = _np(self.vae_decode(latents / self.vae_scaling_factor + self.vae_shift_factor))
            image = np.clip(image / 2 + 0.5, 0.0, 1.0)            # postprocess (:823-851)
            if output_type != "pt":
                image = image.transpose(0, 2, 3, 1)
            if output_type == "pil":
                from PIL import Image
                image = [Image.fromarray((im * 255).round().astype("uint8")) for im in image]
        if not return_dict:
            return (image,)
        return {"images": image}
