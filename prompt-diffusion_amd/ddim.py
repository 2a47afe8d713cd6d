"""(L)-path host surface: ``ControlLDM`` + ``DDIMSampler`` look-alikes over the HIP engine.

Mirrors the reference call (run_prompt_diffusion.ipynb cell 5; cldm/ddim_hacked.py:55-120):

    model = ControlLDM(engine)                      # owns control_scales / only_mid_control (cldm/cldm.py:330-335)
    sampler = DDIMSampler(model)
    samples, intermediates = sampler.sample(S, batch_size, shape, cond, eta=0.0, x_T=x_T,
                                            unconditional_guidance_scale=9.0,
                                            unconditional_conditioning=un_cond, log_every_t=...)

``cond`` / ``un_cond`` are the reference's dicts: ``{"c_crossattn": [ctx], "example_pair": [pair], "query": [query]}``
(lists of tensors, cldm/cldm.py:372-378).  Images are in [-1, 1] like the notebook feeds them.  The loop is driven through the
asynchronous per-step export (``pd_sample_step``), so ``callback`` / ``img_callback`` / ``log_every_t`` behave like the
reference's; the host synchronises only when it reads latents back.
Unsupported reference options raise ``NotImplementedError``.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np

from . import engine as E


def _cat(xs):
    """torch.cat(list, 1) for the reference's list-of-tensors conditioning (cldm/cldm.py:372,377)."""
    if len(xs) == 1:
        return xs[0]
    if E._is_torch(xs[0]):
        import torch
        return torch.cat(list(xs), 1)
    return np.concatenate(list(xs), 1)


def _np32(x):
    if E._is_torch(x):
        x = x.detach().cpu().numpy()
    return np.asarray(x, np.float32)


class ControlLDM:
    """The attributes of cldm.cldm.ControlLDM the sampler touches, backed by an Engine."""

    def __init__(self, engine: E.Engine):
        self.engine = engine
        self.control_scales = [1.0] * 13          # cldm/cldm.py:335
        self.only_mid_control = False             # cldm/cldm.py:334
        self.num_timesteps = engine.cfg.timesteps
        self.parameterization = "eps"
        # DDPM.register_schedule buffers q_sample reads (ddpm.py:138-163): fp64 schedule, stored fp32
        cfg = engine.cfg
        betas = np.linspace(cfg.linear_start ** 0.5, cfg.linear_end ** 0.5, cfg.timesteps, dtype=np.float64) ** 2
        ac = np.cumprod(1.0 - betas)
        self.sqrt_alphas_cumprod = np.sqrt(ac).astype(np.float32)
        self.sqrt_one_minus_alphas_cumprod = np.sqrt(1.0 - ac).astype(np.float32)

    def q_sample(self, x_start, t, noise=None):
        """DDPM.q_sample, ldm/models/diffusion/ddpm.py:356-359 (host arithmetic; used by the inpainting blend)."""
        x_start = _np32(x_start)
        if noise is None:
            noise = np.random.standard_normal(x_start.shape).astype(np.float32)
        t = np.asarray(t, np.int64).reshape(-1)
        sa = self.sqrt_alphas_cumprod[t].reshape(-1, 1, 1, 1)
        sb = self.sqrt_one_minus_alphas_cumprod[t].reshape(-1, 1, 1, 1)
        return (sa * x_start + sb * np.asarray(noise, np.float32)).astype(np.float32)

    def apply_model(self, x_noisy, t, cond, *args, **kwargs):
        """eps = apply_model(x, t, cond), cldm/cldm.py:369-382 (one HIP pass through ControlNet + UNet)."""
        assert isinstance(cond, dict)
        assert cond["example_pair"] is not None
        if self.only_mid_control:
            raise NotImplementedError("only_mid_control is supported through DDIMSampler.sample only")
        return self.engine.eps(x_noisy, t, _cat(cond["c_crossattn"]), _cat(cond["example_pair"]), cond["query"][0],
                               scales=self.control_scales)


class DDIMSampler:
    """cldm/ddim_hacked.py:10 DDIMSampler over the HIP engine (sampling only)."""

    def __init__(self, model: ControlLDM, schedule: str = "linear", **kwargs):
        self.model = model
        self.ddpm_num_timesteps = model.num_timesteps
        self.schedule = schedule

    def make_schedule(self, ddim_num_steps, ddim_discretize="uniform", ddim_eta=0.0, verbose=True):
        if ddim_discretize != "uniform":
            raise NotImplementedError('only ddim_discretize="uniform" (the hot path\'s setting) is built')
        s = self.model.engine.make_schedule(ddim_num_steps, ddim_eta)
        self.ddim_timesteps = s["ddim_timesteps"]
        self.ddim_alphas = s["ddim_alphas"]
        self.ddim_alphas_prev = s["ddim_alphas_prev"]
        self.ddim_sigmas = s["ddim_sigmas"]
        self.ddim_sqrt_one_minus_alphas = s["ddim_sqrt_one_minus_alphas"]

    def sample(self, S, batch_size, shape, conditioning=None, callback=None, normals_sequence=None, img_callback=None,
               quantize_x0=False, eta=0.0, mask=None, x0=None, temperature=1.0, noise_dropout=0.0, score_corrector=None,
               corrector_kwargs=None, verbose=True, x_T=None, log_every_t=100, unconditional_guidance_scale=1.0,
               unconditional_conditioning=None, dynamic_threshold=None, ucg_schedule=None, noise=None, **kwargs):
        if conditioning is None or not isinstance(conditioning, dict):
            raise NotImplementedError("conditioning must be the ControlLDM dict (c_crossattn / example_pair / query)")
        for name, val in (("score_corrector", score_corrector), ("dynamic_threshold", dynamic_threshold),
                          ("ucg_schedule", ucg_schedule), ("normals_sequence", normals_sequence)):
            if val is not None:
                raise NotImplementedError(f"{name} is not supported by the HIP sampler")
        if quantize_x0 or noise_dropout > 0.0:
            raise NotImplementedError("quantize_x0 / noise_dropout are not supported by the HIP sampler")
        ctmp = conditioning[list(conditioning.keys())[0]]
        while isinstance(ctmp, list):
            ctmp = ctmp[0]
        if ctmp.shape[0] != batch_size:   # the reference only prints (ddim_hacked.py:85-86)
            print(f"Warning: Got {ctmp.shape[0]} conditionings but batch-size is {batch_size}")
        C, H, W = shape
        if x_T is None:
            # torch.randn on the reference's device cannot be reproduced bit-exactly; draw on the host
            x_T = np.random.standard_normal((batch_size, C, H, W)).astype(np.float32)
        eng = self.model.engine
        n_steps = eng.num_ddim_steps(S)
        if eta > 0.0 and noise is None:
            noise = np.random.standard_normal((n_steps, batch_size, C, H, W)).astype(np.float32)
        self.make_schedule(S, ddim_eta=eta, verbose=verbose)
        uc = unconditional_conditioning
        kw = dict(x_T=x_T, ctx_cond=_cat(conditioning["c_crossattn"]), pair=_cat(conditioning["example_pair"]),
                  query=conditioning["query"][0], steps=S, cfg_scale=float(unconditional_guidance_scale), eta=float(eta),
                  use_cfg=uc is not None, temperature=float(temperature), control_scales=self.model.control_scales,
                  only_mid_control=self.model.only_mid_control, noise=noise if eta > 0.0 else None,
                  ctx_uncond=_cat(uc["c_crossattn"]) if uc is not None else None)
        if uc is not None:
            # the reference concatenates EVERY cond key (ddim_hacked.py:190-191); pass differing images through
            if uc["example_pair"][0] is not conditioning["example_pair"][0]:
                kw["pair_uncond"] = _cat(uc["example_pair"])
            if uc["query"][0] is not conditioning["query"][0]:
                kw["query_uncond"] = uc["query"][0]
        every = log_every_t
        # per-step export (asynchronous launches; the host only syncs when it reads latents back)
        n = eng.sample_begin(**kw)
        x_inter, preds = [x_T], [x_T]
        if mask is not None:
            assert x0 is not None                          # ddim_hacked.py:155
            mask_np = _np32(mask)
            x0_np = _np32(x0)
        for i in range(n):
            if mask is not None:
                # inpainting blend, ddim_hacked.py:154-157: keep the known region at this step's noise level
                ts = np.full((batch_size,), int(self.ddim_timesteps[n - i - 1]), np.int64)
                img_orig = _np32(self.model.q_sample(x0_np, ts))
                cur = _np32(eng.sample_get(E.PD_GET_LATENTS))
                eng.sample_set_latents(img_orig * mask_np + (np.float32(1.0) - mask_np) * cur)
            eng.sample_step(i)
            index = n - i - 1
            if callback:
                callback(i)
            if img_callback:
                img_callback(eng.sample_get(E.PD_GET_PRED_X0), i)
            if index % every == 0 or index == n - 1:
                x_inter.append(eng.sample_get(E.PD_GET_LATENTS))
                preds.append(eng.sample_get(E.PD_GET_PRED_X0))
        samples = eng.sample_get(E.PD_GET_LATENTS)
        eng.sample_end()
        return samples, {"x_inter": x_inter, "pred_x0": preds}
