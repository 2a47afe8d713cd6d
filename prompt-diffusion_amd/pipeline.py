"""(D)-path host surface: ``PromptDiffusionPipeline.__call__(prompt, image, image_pair, ...)``.

Same keyword arguments, defaults, exception types and loop semantics as the reference's
``pipeline_prompt_diffusion.py:890-1321`` (diffusers naming): images in [0, 1], one float
``controlnet_conditioning_scale`` (or logspace(-1, 0, 13) x scale in guess mode,
promptdiffusioncontrolnet.py:371-375), ``controlnet_keep`` gating from ``control_guidance_start/end``
(:1196-1202), guess-mode zero residuals for the unconditional half (:1248-1253), ``[negative, positive]``
CFG order (:1108, :1269-1270).  The denoising loop itself runs in the HIP engine.

Either side of the hot path:
  * the text encoder: ``tokenizer=`` (a transformers CLIPTokenizer or a callable prompts -> ids [B, 77]) runs the CLIP text
    transformer inside the engine (``pd_text_encode``) when the checkpoint's ``cond_stage_model.*`` tensors are
    loaded; a ``text_encoder(list_of_prompts) -> [B, 77, 768]`` callable overrides it; with neither pass ``prompt_embeds`` /
    ``negative_prompt_embeds`` (the north star consumes the CLIP embedding as a fixed context tensor);
  * the VAE decoder: built into the engine (``pd_vae_decode``) when the checkpoint's ``first_stage_model.*``
    tensors are loaded; a ``vae_decode(latents / scaling_factor) -> images in [-1, 1]`` callable overrides it; with
    neither use ``output_type="latent"``.
"""
from __future__ import annotations

import dataclasses
import inspect
from typing import Any, Callable, Dict, List, Optional, Sequence, Union

import numpy as np

from . import engine as E


@dataclasses.dataclass
class StableDiffusionPipelineOutput:
    images: Any
    nsfw_content_detected: Optional[List[bool]] = None


def _is_pil(x) -> bool:
    return type(x).__module__.startswith("PIL")


def _to_numpy(x):
    if E._is_torch(x):
        return x.detach().float().cpu().numpy()
    return np.asarray(x)


class PromptDiffusionPipeline:
    _callback_tensor_inputs = ["latents", "prompt_embeds", "negative_prompt_embeds"]
    vae_scale_factor = 8
    vae_scaling_factor = 0.18215      # models/cldm_v15.yaml:17

    def __init__(self, engine: E.Engine, text_encoder: Optional[Callable] = None, vae_decode: Optional[Callable] = None,
                 scheduler: Any = None, tokenizer: Any = None):
        self.engine = engine
        self.tokenizer = tokenizer
        if text_encoder is None and tokenizer is not None:
            # cond stage inside the engine (pd_text_encode): the caller only supplies the tokenizer -- a
            # transformers CLIPTokenizer, or any callable prompts -> token ids [B, context_len]
            text_encoder = self._engine_text_encoder
        self.text_encoder = text_encoder
        self.vae_decode = vae_decode
        self.scheduler = scheduler          # None = the engine's fused DDIM (DDIMScheduler semantics of SD1.5)
        self._guidance_scale = 7.5

    # ------------------------------------------------------------------ properties of the reference
    @property
    def guidance_scale(self):
        return self._guidance_scale

    @property
    def do_classifier_free_guidance(self):
        return self._guidance_scale > 1     # pipeline_prompt_diffusion.py:878

    # ------------------------------------------------------------------ input validation
    def check_image(self, image, prompt, prompt_embeds):
        """pipeline_prompt_diffusion.py:722-757."""
        ok_single = _is_pil(image) or E._is_torch(image) or isinstance(image, np.ndarray)
        ok_list = isinstance(image, list) and len(image) > 0 and (
            _is_pil(image[0]) or E._is_torch(image[0]) or isinstance(image[0], np.ndarray))
        if not ok_single and not ok_list:
            raise TypeError("image must be passed and be one of PIL image, numpy array, torch tensor, list of PIL images, "
                            f"list of numpy arrays or list of torch tensors, but is {type(image)}")
        image_batch_size = 1 if _is_pil(image) else len(image)
        if prompt is not None and isinstance(prompt, str):
            prompt_batch_size = 1
        elif prompt is not None and isinstance(prompt, list):
            prompt_batch_size = len(prompt)
        else:
            prompt_batch_size = prompt_embeds.shape[0]
        if image_batch_size != 1 and image_batch_size != prompt_batch_size:
            raise ValueError("If image batch size is not 1, image batch size must be same as prompt batch size. "
                             f"image batch size: {image_batch_size}, prompt batch size: {prompt_batch_size}")

    def check_inputs(self, prompt, image, image_pair, callback_steps, negative_prompt=None, prompt_embeds=None,
                     negative_prompt_embeds=None, controlnet_conditioning_scale=1.0, control_guidance_start=0.0,
                     control_guidance_end=1.0, callback_on_step_end_tensor_inputs=None):
        """pipeline_prompt_diffusion.py:559-719 (single-ControlNet branches)."""
        if callback_steps is not None and (not isinstance(callback_steps, int) or callback_steps <= 0):
            raise ValueError(f"`callback_steps` has to be a positive integer but is {callback_steps} of type {type(callback_steps)}.")
        if callback_on_step_end_tensor_inputs is not None and not all(
                k in self._callback_tensor_inputs for k in callback_on_step_end_tensor_inputs):
            bad = [k for k in callback_on_step_end_tensor_inputs if k not in self._callback_tensor_inputs]
            raise ValueError(f"`callback_on_step_end_tensor_inputs` has to be in {self._callback_tensor_inputs}, but found {bad}")
        if prompt is not None and prompt_embeds is not None:
            raise ValueError(f"Cannot forward both `prompt`: {prompt} and `prompt_embeds`: {prompt_embeds}. Please make sure to"
                             " only forward one of the two.")
        elif prompt is None and prompt_embeds is None:
            raise ValueError("Provide either `prompt` or `prompt_embeds`. Cannot leave both `prompt` and `prompt_embeds` undefined.")
        elif prompt is not None and (not isinstance(prompt, str) and not isinstance(prompt, list)):
            raise ValueError(f"`prompt` has to be of type `str` or `list` but is {type(prompt)}")
        if negative_prompt is not None and negative_prompt_embeds is not None:
            raise ValueError(f"Cannot forward both `negative_prompt`: {negative_prompt} and `negative_prompt_embeds`:"
                             f" {negative_prompt_embeds}. Please make sure to only forward one of the two.")
        if prompt_embeds is not None and negative_prompt_embeds is not None:
            if tuple(prompt_embeds.shape) != tuple(negative_prompt_embeds.shape):
                raise ValueError("`prompt_embeds` and `negative_prompt_embeds` must have the same shape when passed directly, but"
                                 f" got: `prompt_embeds` {prompt_embeds.shape} != `negative_prompt_embeds` {negative_prompt_embeds.shape}.")
        self.check_image(image, prompt, prompt_embeds)
        if len(image_pair) == 2:
            for im in image_pair:
                self.check_image(im, prompt, prompt_embeds)
        else:
            raise ValueError(f"You have passed a list of images of length {len(image_pair)}.Make sure the list size equals to two.")
        if not isinstance(controlnet_conditioning_scale, float):
            raise TypeError("For single controlnet: `controlnet_conditioning_scale` must be type `float`.")
        starts = control_guidance_start if isinstance(control_guidance_start, (tuple, list)) else [control_guidance_start]
        ends = control_guidance_end if isinstance(control_guidance_end, (tuple, list)) else [control_guidance_end]
        if len(starts) != len(ends):
            raise ValueError(f"`control_guidance_start` has {len(starts)} elements, but `control_guidance_end` has {len(ends)} elements."
                             " Make sure to provide the same number of elements to each list.")
        for start, end in zip(starts, ends):
            if start >= end:
                raise ValueError(f"control guidance start: {start} cannot be larger or equal to control guidance end: {end}.")
            if start < 0.0:
                raise ValueError(f"control guidance start: {start} can't be smaller than 0.")
            if end > 1.0:
                raise ValueError(f"control guidance end: {end} can't be larger than 1.0.")

    # ------------------------------------------------------------------ host pre-processing
    def prepare_image(self, image, width, height, batch_size, num_images_per_prompt):
        """VaeImageProcessor(do_normalize=False).preprocess + repeat (pipeline :236-238, :760-788): [B,3,H,W] in [0,1]."""
        items = image if isinstance(image, list) else [image]
        outs = []
        for im in items:
            if _is_pil(im):
                if height is not None and width is not None and im.size != (width, height):
                    from PIL import Image
                    im = im.resize((width, height), resample=Image.LANCZOS)
                a = np.asarray(im.convert("RGB"), dtype=np.float32) / 255.0
                outs.append(a.transpose(2, 0, 1)[None])
            else:
                a = _to_numpy(im).astype(np.float32)
                if isinstance(im, np.ndarray):        # numpy images are [H,W,C] or [B,H,W,C] in [0,1]
                    a = a[None] if a.ndim == 3 else a
                    a = a.transpose(0, 3, 1, 2)
                else:                                 # torch tensors are [C,H,W] or [B,C,H,W]
                    a = a[None] if a.ndim == 3 else a
                outs.append(a)
        x = np.concatenate(outs, axis=0)
        repeat_by = batch_size if x.shape[0] == 1 else num_images_per_prompt
        return np.repeat(x, repeat_by, axis=0)

    def prepare_latents(self, batch_size, num_channels_latents, height, width, generator, latents=None):
        """pipeline :791-806; init_noise_sigma of DDIM is 1."""
        shape = (batch_size, num_channels_latents, height // self.vae_scale_factor, width // self.vae_scale_factor)
        if isinstance(generator, list) and len(generator) != batch_size:
            raise ValueError(f"You have passed a list of generators of length {len(generator)}, but requested an effective batch"
                             f" size of {batch_size}. Make sure the batch size matches the length of the generators.")
        if latents is None:
            latents = self._randn(shape, generator)
        else:
            latents = _to_numpy(latents).astype(np.float32)
        sigma = float(getattr(self.scheduler, "init_noise_sigma", 1.0)) if self.scheduler is not None else 1.0
        return latents * np.float32(sigma)

    @staticmethod
    def _randn(shape, generator):
        if generator is None:
            return np.random.standard_normal(shape).astype(np.float32)
        if isinstance(generator, np.random.Generator):
            return generator.standard_normal(shape, dtype=np.float32)
        import torch
        if isinstance(generator, list):
            return np.concatenate([torch.randn((1,) + tuple(shape[1:]), generator=g, device=g.device).cpu().numpy()
                                   for g in generator])
        return torch.randn(shape, generator=generator, device=generator.device).cpu().numpy()

    def _tokenize(self, prompts: List[str]):
        """pipeline :386-392 / FrozenCLIPEmbedder.forward (modules.py:119-121): pad / truncate to context_len."""
        L = self.engine.cfg.context_len
        tok = self.tokenizer
        if hasattr(tok, "model_max_length") or hasattr(tok, "pad_token_id"):      # transformers tokenizer
            ids = tok(prompts, padding="max_length", max_length=L, truncation=True, return_tensors="np")["input_ids"]
        else:
            ids = tok(prompts)
        ids = np.asarray(_to_numpy(ids), np.int32)
        if ids.ndim != 2 or ids.shape != (len(prompts), L):
            raise ValueError(f"tokenizer must return ids of shape [{len(prompts)}, {L}], got {ids.shape}")
        return ids

    def _engine_text_encoder(self, prompts: List[str], clip_skip: Optional[int] = None):
        if self.engine.text_weights_missing() != 0:
            raise ValueError("a string `prompt` needs the cond_stage_model.transformer.text_model.* weights in the engine "
                             "(or a text_encoder callable, or `prompt_embeds`)")
        return self.engine.text_encode(self._tokenize(prompts), clip_skip=clip_skip or 0)

    def _encode_text(self, prompts: List[str], clip_skip: Optional[int]):
        """text_encoder(ids)[0], or with clip_skip the hidden state of layer -(clip_skip+1) through final_layer_norm
        (pipeline :398-413).  The conditional prompt honours clip_skip; the reference encodes the negative prompt without
        it (pipeline :455-459)."""
        if not clip_skip:
            return _to_numpy(self.text_encoder(prompts))
        if "clip_skip" not in set(inspect.signature(self.text_encoder).parameters.keys()):
            raise ValueError("clip_skip needs a text_encoder that accepts `clip_skip=` (the engine's own does); "
                             "or pass prompt_embeds computed with it")
        return _to_numpy(self.text_encoder(prompts, clip_skip=clip_skip))

    def encode_prompt(self, prompt, num_images_per_prompt, do_cfg, negative_prompt=None, prompt_embeds=None,
                      negative_prompt_embeds=None):
        """pipeline :308-487 reduced to its tensor contract: returns (prompt_embeds, negative_prompt_embeds) [B*n, L, D]."""
        if prompt_embeds is None:
            if self.text_encoder is None:
                raise ValueError("a string `prompt` needs a text_encoder; this engine consumes the CLIP embedding as a fixed "
                                 "context tensor -- pass `prompt_embeds` (and `negative_prompt_embeds`)")
            plist = [prompt] if isinstance(prompt, str) else list(prompt)
            prompt_embeds = self._encode_text(plist, getattr(self, "_clip_skip", None))
        pe = np.repeat(_to_numpy(prompt_embeds).astype(np.float32), num_images_per_prompt, axis=0)
        ne = None
        if do_cfg:
            if negative_prompt_embeds is None:
                if self.text_encoder is None:
                    raise ValueError("classifier-free guidance needs `negative_prompt_embeds` when no text_encoder is attached")
                bs = pe.shape[0] // num_images_per_prompt
                if negative_prompt is None:
                    neg = [""] * bs
                elif isinstance(negative_prompt, str):
                    neg = [negative_prompt] * bs
                else:
                    if len(negative_prompt) != bs:
                        raise ValueError(f"`negative_prompt`: {negative_prompt} has batch size {len(negative_prompt)}, but `prompt`: "
                                         f"{prompt} has batch size {bs}.")
                    neg = list(negative_prompt)
                negative_prompt_embeds = _to_numpy(self.text_encoder(neg))
            ne = np.repeat(_to_numpy(negative_prompt_embeds).astype(np.float32), num_images_per_prompt, axis=0)
        return pe, ne

    # ------------------------------------------------------------------ the call
    def __call__(self, prompt: Union[str, List[str]] = None, image=None, image_pair: List = None,
                 height: Optional[int] = None, width: Optional[int] = None, num_inference_steps: int = 50,
                 timesteps: List[int] = None, guidance_scale: float = 7.5,
                 negative_prompt: Optional[Union[str, List[str]]] = None, num_images_per_prompt: Optional[int] = 1,
                 eta: float = 0.0, generator=None, latents=None, prompt_embeds=None, negative_prompt_embeds=None,
                 ip_adapter_image=None, output_type: Optional[str] = "pil", return_dict: bool = True,
                 cross_attention_kwargs: Optional[Dict[str, Any]] = None,
                 controlnet_conditioning_scale: Union[float, List[float]] = 1.0, guess_mode: bool = False,
                 control_guidance_start: Union[float, List[float]] = 0.0, control_guidance_end: Union[float, List[float]] = 1.0,
                 clip_skip: Optional[int] = None, callback_on_step_end: Optional[Callable] = None,
                 callback_on_step_end_tensor_inputs: List[str] = ["latents"], **kwargs):
        callback = kwargs.pop("callback", None)
        callback_steps = kwargs.pop("callback_steps", None)
        if ip_adapter_image is not None:
            raise NotImplementedError("ip_adapter_image is outside the hot path this engine replaces")
        if cross_attention_kwargs:
            raise NotImplementedError("cross_attention_kwargs (LoRA scale) is outside the hot path this engine replaces")
        self._clip_skip = clip_skip
        # 0/1. defaults + checks (pipeline :1033-1062)
        if not isinstance(control_guidance_start, list) and isinstance(control_guidance_end, list):
            control_guidance_start = len(control_guidance_end) * [control_guidance_start]
        elif not isinstance(control_guidance_end, list) and isinstance(control_guidance_start, list):
            control_guidance_end = len(control_guidance_start) * [control_guidance_end]
        elif not isinstance(control_guidance_start, list) and not isinstance(control_guidance_end, list):
            control_guidance_start, control_guidance_end = [control_guidance_start], [control_guidance_end]
        self.check_inputs(prompt, image, image_pair, callback_steps, negative_prompt, prompt_embeds, negative_prompt_embeds,
                          controlnet_conditioning_scale, control_guidance_start, control_guidance_end,
                          callback_on_step_end_tensor_inputs)
        self._guidance_scale = guidance_scale
        # 2. call parameters
        if prompt is not None and isinstance(prompt, str):
            batch_size = 1
        elif prompt is not None and isinstance(prompt, list):
            batch_size = len(prompt)
        else:
            batch_size = prompt_embeds.shape[0]
        do_cfg = self.do_classifier_free_guidance
        # 3. text context
        pe, ne = self.encode_prompt(prompt, num_images_per_prompt, do_cfg, negative_prompt, prompt_embeds, negative_prompt_embeds)
        # 3.1 / 4. images: [B,6,H,W] example pair and [B,3,H,W] query, both in [0,1]
        pair = np.concatenate([self.prepare_image(im, width, height, batch_size * num_images_per_prompt, num_images_per_prompt)
                               for im in image_pair], axis=1)
        query = self.prepare_image(image, width, height, batch_size * num_images_per_prompt, num_images_per_prompt)
        height, width = query.shape[-2:]
        B = batch_size * num_images_per_prompt
        # 6. latents
        x_T = self.prepare_latents(B, self.engine.cfg.in_channels, height, width, generator, latents)
        # 7.2 controlnet_keep gating and per-step scales (pipeline :1196-1202, :1229-1235; controlnet :371-378)
        custom_ts = None
        if self.scheduler is not None:      # plug-in scheduler: its own time grid (retrieve_timesteps, pipeline :101-142, :1164)
            if timesteps is not None:
                if "timesteps" not in set(inspect.signature(self.scheduler.set_timesteps).parameters.keys()):
                    raise ValueError(f"The current scheduler class {self.scheduler.__class__}'s `set_timesteps` does not support custom"
                                     f" timestep schedules. Please check whether you are using the correct scheduler.")
                self.scheduler.set_timesteps(timesteps=timesteps)
            else:
                self.scheduler.set_timesteps(num_inference_steps)
            n_steps = len(self.scheduler.timesteps)
        else:
            # the engine's own DDIM loop with diffusers' DDIMScheduler grid for SD1.5 (timestep_spacing "leading",
            # steps_offset 1): exactly num_inference_steps entries arange(S) * (T // S) + 1.  For S dividing T this IS the
            # LDM grid of make_ddim_timesteps and the fused default runs; otherwise -- and for a caller-supplied list -- the
            # grid goes to the engine as custom timesteps.
            T = self.engine.cfg.timesteps
            if timesteps is not None:
                custom_ts = [int(t) for t in _to_numpy(timesteps).reshape(-1)]
            elif T % num_inference_steps != 0:
                custom_ts = [int(t) for t in (np.arange(num_inference_steps) * (T // num_inference_steps))[::-1] + 1]
            n_steps = len(custom_ts) if custom_ts is not None else self.engine.num_ddim_steps(num_inference_steps)
        keep = [1.0 - float(i / n_steps < control_guidance_start[0] or (i + 1) / n_steps > control_guidance_end[0])
                for i in range(n_steps)]
        n_ctl = E.PD_NUM_CONTROL
        base = np.logspace(-1, 0, n_ctl).astype(np.float32) if guess_mode else np.ones(n_ctl, np.float32)
        scales_step = np.stack([base * np.float32(controlnet_conditioning_scale) * np.float32(k) for k in keep])
        noise = None
        if eta > 0.0:
            if self.scheduler is None:   # a plug-in scheduler draws its own noise (`generator` is forwarded to step())
                noise = self._randn((n_steps,) + x_T.shape, generator if not isinstance(generator, list) else None)
        kw = dict(x_T=x_T, ctx_cond=pe, ctx_uncond=ne, pair=pair, query=query, steps=num_inference_steps,
                  cfg_scale=float(guidance_scale), eta=float(eta), use_cfg=do_cfg, guess_mode=guess_mode,
                  control_scales_step=scales_step, noise=noise)
        if custom_ts is not None:
            kw["timesteps"] = custom_ts
        eng = self.engine
        if self.scheduler is None and callback_on_step_end is None and callback is None:
            lat = eng.ddim_sample(**kw)                                  # 8. the fused loop
        else:
            lat = self._stepwise(kw, scales_step, callback_on_step_end, callback_on_step_end_tensor_inputs, callback,
                                 callback_steps, pe, ne, eta, generator)
        # 9. post-processing (pipeline :1298-1321); safety checker is forced off there too
        if output_type == "latent":
            images = lat
        else:
            if self.vae_decode is not None:
                img = _to_numpy(self.vae_decode(lat / np.float32(self.vae_scaling_factor)))
            elif getattr(self.engine.cfg, "vae_ch", 0) > 0 and self.engine.vae_weights_missing() == 0:
                img = _to_numpy(self.engine.vae_decode(lat))      # divides by scaling_factor itself (ddpm.py:827)
            else:
                raise ValueError('output_type other than "latent" needs first-stage weights in the engine '
                                 "(first_stage_model.decoder.* / post_quant_conv.*) or a vae_decode callable")
            img = np.clip(img / 2 + 0.5, 0, 1).transpose(0, 2, 3, 1)      # denormalize, NHWC
            if output_type == "pil":
                from PIL import Image
                images = [Image.fromarray((im * 255).round().astype("uint8")) for im in img]
            else:
                images = img
        if not return_dict:
            return (images, None)
        return StableDiffusionPipelineOutput(images=images, nsfw_content_detected=None)

    # ------------------------------------------------------------------ per-step driver (callbacks / plug-in schedulers)
    def _stepwise(self, kw, scales_step, cb_end, cb_inputs, cb_legacy, cb_steps, pe, ne, eta, generator):
        eng = self.engine
        sched = self.scheduler
        if sched is not None:       # the engine only evaluates eps at the scheduler's timesteps: no DDIM tables needed
            kw = {k: v for k, v in kw.items() if k not in ("control_scales_step", "noise")}
            kw["eta"] = 0.0
        n = eng.sample_begin(**kw)
        if sched is not None:
            ts = [int(t) for t in sched.timesteps]       # set_timesteps ran in __call__
            extra = {}
            params = set(inspect.signature(sched.step).parameters.keys())
            if "eta" in params:
                extra["eta"] = eta
            if "generator" in params:
                extra["generator"] = generator
        elif kw.get("timesteps") is not None:
            ts = list(kw["timesteps"])
        else:
            ts = [int(t) for t in np.flip(eng.make_schedule(kw["steps"], kw["eta"])["ddim_timesteps"])]
        lat = None
        for i, t in enumerate(ts):
            if sched is None:
                eng.sample_step(i)
            else:
                # scheduler.scale_model_input is the identity for the DDIM / UniPC families used with SD1.5
                import torch
                noise_pred = eng.sample_eps_at(t, scales_step[min(i, len(scales_step) - 1)])
                cur = eng.sample_get(E.PD_GET_LATENTS)
                out = sched.step(torch.from_numpy(np.asarray(noise_pred)), t, torch.from_numpy(np.asarray(cur)), **extra,
                                 return_dict=False)[0]
                eng.sample_set_latents(out.numpy())
            if cb_end is not None:
                lat = eng.sample_get(E.PD_GET_LATENTS)
                cb_kwargs = {}
                for k in cb_inputs:
                    cb_kwargs[k] = {"latents": lat, "prompt_embeds": pe, "negative_prompt_embeds": ne}[k]
                outs = cb_end(self, i, t, cb_kwargs) or {}
                new = outs.pop("latents", None)
                if new is not None and new is not lat:
                    eng.sample_set_latents(_to_numpy(new))
            if cb_legacy is not None and i % (cb_steps or 1) == 0:
                cb_legacy(i, t, eng.sample_get(E.PD_GET_LATENTS))
        lat = eng.sample_get(E.PD_GET_LATENTS)
        eng.sample_end()
        return lat
