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
``ucg_schedule`` (per-step guidance, :159-161), ``noise_dropout`` (:232), ``make_schedule(ddim_discretize="quad")`` (util.py:49-50) and
``encode`` / ``stochastic_encode`` / ``decode`` (:237-318) are built on the same per-step export; ``score_corrector``,
``quantize_x0`` and ``dynamic_threshold`` raise ``NotImplementedError``.
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
        """cldm/ddim_hacked.py:23-52.  'uniform' comes from the engine (pd_make_schedule); 'quad' (util.py:49-50) is the same
        arithmetic on the host over the quadratic timestep grid, which the engine then samples as a custom grid."""
        eng = self.model.engine
        self.ddim_eta = float(ddim_eta)
        self._custom_ts = None
        if ddim_discretize == "uniform":
            s = eng.make_schedule(ddim_num_steps, ddim_eta)
        elif ddim_discretize == "quad":
            cfg = eng.cfg
            ts = ((np.linspace(0, np.sqrt(cfg.timesteps * .8), ddim_num_steps)) ** 2).astype(int) + 1
            betas = np.linspace(cfg.linear_start ** 0.5, cfg.linear_end ** 0.5, cfg.timesteps, dtype=np.float64) ** 2
            ac = np.cumprod(1.0 - betas).astype(np.float32)             # ddpm.py:155-157 registers float32 buffers
            al = ac[ts]                                                  # util.py:62-66
            ap = np.asarray([ac[0]] + ac[ts[:-1]].tolist(), np.float32)
            sg = ddim_eta * np.sqrt((1 - ap.astype(np.float64)) / (1 - al.astype(np.float64)) * (1 - al.astype(np.float64) / ap.astype(np.float64)))
            s = dict(ddim_timesteps=ts.astype(np.int64), ddim_alphas=al, ddim_alphas_prev=ap, ddim_sigmas=sg.astype(np.float32),
                     ddim_sqrt_one_minus_alphas=np.sqrt(np.float32(1.0) - al))
            self._custom_ts = ts.astype(np.int64)
        else:
            raise NotImplementedError(f'There is no ddim discretization method called "{ddim_discretize}"')   # util.py:52
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
        for name, val in (("score_corrector", score_corrector), ("dynamic_threshold", dynamic_threshold)):
            if val is not None:
                raise NotImplementedError(f"{name} is not supported by the HIP sampler")
        if quantize_x0:
            raise NotImplementedError("quantize_x0 is not supported by the HIP sampler")
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
        if eta > 0.0 and noise_dropout > 0.0:
            # torch.nn.functional.dropout(noise, p), ddim_hacked.py:231-232: zero with probability p, survivors scaled by 1/(1-p)
            noise = _np32(noise)
            keep_mask = (np.random.random_sample(noise.shape) >= noise_dropout).astype(np.float32)
            noise = noise * keep_mask * np.float32(1.0 / (1.0 - noise_dropout))
        if ucg_schedule is not None:
            assert len(ucg_schedule) == n_steps                      # ddim_hacked.py:160
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
            if ucg_schedule is not None:
                eng.sample_set_guidance(float(ucg_schedule[i]))          # ddim_hacked.py:159-161
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

    # ------------------------------------------------------------------ encode / decode (ddim_hacked.py:237-318)
    def _session(self, x, cond, uc, scale, timesteps):
        """one engine session over a custom descending timestep list (context K/V, hint embedders and time embeddings hoisted once)"""
        kw = dict(x_T=x, ctx_cond=_cat(cond["c_crossattn"]), pair=_cat(cond["example_pair"]), query=cond["query"][0],
                  steps=len(timesteps), cfg_scale=float(scale), eta=self.ddim_eta, use_cfg=uc is not None, temperature=1.0,
                  control_scales=self.model.control_scales, only_mid_control=self.model.only_mid_control, timesteps=timesteps,
                  ctx_uncond=_cat(uc["c_crossattn"]) if uc is not None else None)
        if uc is not None:
            if uc["example_pair"][0] is not cond["example_pair"][0]:
                kw["pair_uncond"] = _cat(uc["example_pair"])
            if uc["query"][0] is not cond["query"][0]:
                kw["query_uncond"] = uc["query"][0]
        return self.model.engine.sample_begin(**kw)

    def encode(self, x0, c, t_enc, use_original_steps=False, return_intermediates=None, unconditional_guidance_scale=1.0,
               unconditional_conditioning=None, callback=None):
        """DDIM inversion exactly as ddim_hacked.py:237-282 runs it -- including its quirk of handing the LOOP INDEX i (not the
        DDIM timestep) to apply_model.  With guidance the reference concatenates the two conditioning dicts with torch.cat
        (:263), which raises TypeError for ControlLDM's dict conditioning; that path raises the same here."""
        if use_original_steps:
            raise NotImplementedError("use_original_steps needs the full 1000-step buffers of DDPM; not built")
        num_reference_steps = self.ddim_timesteps.shape[0]
        assert t_enc <= num_reference_steps
        if unconditional_guidance_scale != 1.0:
            assert unconditional_conditioning is not None
            raise TypeError("expected Tensor as element 0 in argument 0, but got dict")
        num_steps = t_enc
        alphas_next = self.ddim_alphas[:num_steps].astype(np.float32)
        alphas = self.ddim_alphas_prev[:num_steps].astype(np.float32)
        eng = self.model.engine
        x_next = _np32(x0)
        self._session(x_next, c, None, 1.0, [1])     # the session only hoists the conditioning; eps is asked for at explicit t
        intermediates, inter_steps = [], []
        one = np.float32(1.0)
        for i in range(num_steps):
            noise_pred = _np32(eng.sample_eps_at(i, self.model.control_scales))
            xt_weighted = np.sqrt(alphas_next[i] / alphas[i]) * x_next
            weighted_noise_pred = np.sqrt(alphas_next[i]) * (np.sqrt(one / alphas_next[i] - one) - np.sqrt(one / alphas[i] - one)) * noise_pred
            x_next = (xt_weighted + weighted_noise_pred).astype(np.float32)
            eng.sample_set_latents(x_next)
            if return_intermediates and i % (num_steps // return_intermediates) == 0 and i < num_steps - 1:
                intermediates.append(x_next)
                inter_steps.append(i)
            elif return_intermediates and i >= num_steps - 2:
                intermediates.append(x_next)
                inter_steps.append(i)
            if callback:
                callback(i)
        eng.sample_end()
        out = {"x_encoded": x_next, "intermediate_steps": inter_steps}
        if return_intermediates:
            out.update({"intermediates": intermediates})
        return x_next, out

    def stochastic_encode(self, x0, t, use_original_steps=False, noise=None):
        """ddim_hacked.py:284-299: q(x_t | x_0) on the DDIM grid; t indexes the DDIM schedule."""
        if use_original_steps:
            raise NotImplementedError("use_original_steps needs the full 1000-step buffers of DDPM; not built")
        x0 = _np32(x0)
        if noise is None:
            noise = np.random.standard_normal(x0.shape).astype(np.float32)
        t = np.asarray(t, np.int64).reshape(-1)
        sa = np.sqrt(self.ddim_alphas.astype(np.float32))[t].reshape(-1, 1, 1, 1)
        sb = self.ddim_sqrt_one_minus_alphas.astype(np.float32)[t].reshape(-1, 1, 1, 1)
        return (sa * x0 + sb * _np32(noise)).astype(np.float32)

    def decode(self, x_latent, cond, t_start, unconditional_guidance_scale=1.0, unconditional_conditioning=None,
               use_original_steps=False, callback=None):
        """ddim_hacked.py:301-318: p_sample_ddim over the first t_start timesteps of the current schedule, in the engine's own
        step loop (custom timestep grid)."""
        if use_original_steps:
            raise NotImplementedError("use_original_steps needs the full 1000-step buffers of DDPM; not built")
        if self.ddim_eta > 0.0:
            raise NotImplementedError("decode with eta > 0 draws fresh noise per step in the reference; pass eta = 0 to make_schedule")
        timesteps = np.asarray(self.ddim_timesteps)[:t_start]
        eng = self.model.engine
        n = self._session(_np32(x_latent), cond, unconditional_conditioning, unconditional_guidance_scale,
                          [int(t) for t in np.flip(timesteps)])
        for i in range(n):
            eng.sample_step(i)
            if callback:
                callback(i)
        x_dec = eng.sample_get(E.PD_GET_LATENTS)
        eng.sample_end()
        return x_dec
