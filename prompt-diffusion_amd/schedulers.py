"""Host-side scheduler plug-ins for `PromptDiffusionPipeline(scheduler=...)`.

The reference's README swaps the pipeline's scheduler for diffusers' `UniPCMultistepScheduler`
(`README.md:49`, `pipe.scheduler = UniPCMultistepScheduler.from_config(pipe.scheduler.config)`) and BASELINE config #4
(768x768, 20 steps) is quoted with it.  diffusers is not vendored in the reference tree, so this is a restatement of the
published algorithm -- UniPC, Zhao et al. 2023, "UniPC: A Unified Predictor-Corrector Framework for Fast Sampling of
Diffusion Models", Alg. 5-8 (multistep UniP-p / UniC-p, data prediction, B(h) = e^h - 1 "bh2" or h "bh1") -- behind the
scheduler interface the pipeline drives (`set_timesteps`, `timesteps`, `scale_model_input`, `step(..., return_dict=False)`,
`init_noise_sigma`).  PARITY UNPINNED against diffusers (no source, no fixtures); it is cross-checked against an
independent fp64 closed-form restatement in `oracle/` and against the analytic probability-flow solution for Gaussian
data (`tests/test_schedulers_cpu.py`).

All coefficient arithmetic is fp64 NumPy; the per-step update is O(latent size) host work (2 MB per step at 768x768,
bs 8) and stays on the host like the reference's scheduler does.
"""
from __future__ import annotations

import math
from typing import List, Optional

import numpy as np


def _to_np(x):
    if isinstance(x, np.ndarray):
        return x, None
    import torch
    if isinstance(x, torch.Tensor):
        return x.detach().cpu().numpy(), x
    return np.asarray(x), None


class UniPCMultistepScheduler:
    """UniPC multistep predictor-corrector, epsilon-prediction model, data-prediction (x0) form.

    Defaults follow what `from_config(<SD1.5 scheduler config>)` yields in the reference's README flow:
    scaled-linear betas 0.00085..0.012 over 1000 steps, solver_order 2, `bh2`, lower_order_final, final sigma 0.
    """
    order = 1
    init_noise_sigma = 1.0

    def __init__(self, num_train_timesteps: int = 1000, beta_start: float = 0.00085, beta_end: float = 0.012,
                 beta_schedule: str = "scaled_linear", solver_order: int = 2, prediction_type: str = "epsilon",
                 predict_x0: bool = True, solver_type: str = "bh2", lower_order_final: bool = True,
                 disable_corrector: Optional[List[int]] = None, timestep_spacing: str = "linspace", steps_offset: int = 0,
                 thresholding: bool = False):
        if beta_schedule == "scaled_linear":
            betas = np.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=np.float64) ** 2
        elif beta_schedule == "linear":
            betas = np.linspace(beta_start, beta_end, num_train_timesteps, dtype=np.float64)
        else:
            raise NotImplementedError(f"beta_schedule {beta_schedule!r}")
        if prediction_type != "epsilon":
            raise NotImplementedError("the Prompt-Diffusion UNet predicts epsilon (ddpm.py:71)")
        if not predict_x0:
            raise NotImplementedError("noise-prediction form of UniPC")
        if thresholding:
            raise NotImplementedError("dynamic thresholding is for pixel-space models")
        if solver_type not in ("bh1", "bh2"):
            raise ValueError("solver_type must be 'bh1' or 'bh2'")
        if solver_order < 1 or solver_order > 3:
            raise ValueError("solver_order must be 1, 2 or 3")
        if timestep_spacing not in ("linspace", "leading", "trailing"):
            raise ValueError(f"timestep_spacing {timestep_spacing!r}")
        self.num_train_timesteps = num_train_timesteps
        self.alphas_cumprod = np.cumprod(1.0 - betas)
        self.solver_order = solver_order
        self.solver_type = solver_type
        self.lower_order_final = lower_order_final
        self.disable_corrector = list(disable_corrector or [])
        self.timestep_spacing = timestep_spacing
        self.steps_offset = steps_offset
        self.timesteps = np.zeros((0,), np.int64)
        self._reset()

    def _reset(self):
        self.model_outputs = [None] * self.solver_order     # x0 predictions, oldest first
        self.timestep_list = [None] * self.solver_order
        self.lower_order_nums = 0
        self.last_sample = None
        self._step_index = 0
        self.this_order = 1

    # ------------------------------------------------------------------ schedule
    def set_timesteps(self, num_inference_steps: int, device=None):
        T, n = self.num_train_timesteps, int(num_inference_steps)
        if n < 1 or n > T:
            raise ValueError("num_inference_steps out of range")
        if self.timestep_spacing == "linspace":
            ts = np.linspace(0, T - 1, n + 1).round()[::-1][:-1]
        elif self.timestep_spacing == "leading":
            ts = (np.arange(0, n + 1) * (T // (n + 1))).round()[::-1][:-1] + self.steps_offset
        else:
            ts = np.arange(T, 0, -T / n).round() - 1
        self.timesteps = ts.astype(np.int64)
        a = self.alphas_cumprod[self.timesteps]
        # alpha_t = sqrt(abar), sigma_t = sqrt(1 - abar), lambda = log(alpha / sigma); the final point is sigma = 0
        self._alpha = np.concatenate([np.sqrt(a), [1.0]])
        self._sigma = np.concatenate([np.sqrt(1.0 - a), [0.0]])
        with np.errstate(divide="ignore"):
            self._lambda = np.log(self._alpha) - np.log(self._sigma)
        self.num_inference_steps = n
        self._reset()

    def scale_model_input(self, sample, timestep=None):
        return sample

    # ------------------------------------------------------------------ coefficients (fp64)
    def _coeffs(self, i0: int, order: int, hist_idx: List[int]):
        """Quantities shared by UniP and UniC for the update from schedule point i0 to i0+1.
        hist_idx: schedule indices of the older model outputs, newest first (length order-1)."""
        lam0, lam1 = self._lambda[i0], self._lambda[i0 + 1]
        h = lam1 - lam0
        rks = [(self._lambda[j] - lam0) / h for j in hist_idx] + [1.0]
        hh = -h
        h_phi_1 = math.expm1(hh) if np.isfinite(hh) else -1.0
        h_phi_k = (h_phi_1 / hh - 1.0) if np.isfinite(hh) else -1.0
        B_h = h_phi_1 if self.solver_type == "bh2" else hh
        R, b, fact = [], [], 1.0
        for i in range(1, order + 1):
            R.append(np.power(rks, i - 1))
            b.append(h_phi_k * fact / B_h)
            fact *= i + 1
            h_phi_k = (h_phi_k / hh - 1.0 / fact) if np.isfinite(hh) else -1.0 / fact
        return dict(h_phi_1=h_phi_1, B_h=B_h, rks=np.asarray(rks), R=np.stack(R), b=np.asarray(b),
                    alpha1=self._alpha[i0 + 1], sig_ratio=self._sigma[i0 + 1] / self._sigma[i0])

    def _d1s(self, m0, hist, rks):
        return [(m - m0) / r for m, r in zip(hist, rks[:-1])]

    def _uni_p(self, x, i0, order):
        m0 = self.model_outputs[-1]
        hist = [self.model_outputs[-(k + 1)] for k in range(1, order)]
        hidx = [self.timestep_list[-(k + 1)] for k in range(1, order)]
        c = self._coeffs(i0, order, hidx)
        x_t = c["sig_ratio"] * x - c["alpha1"] * c["h_phi_1"] * m0
        if order > 1:
            D1s = self._d1s(m0, hist, c["rks"])
            rhos = np.array([0.5]) if order == 2 else np.linalg.solve(c["R"][:-1, :-1], c["b"][:-1])
            x_t = x_t - c["alpha1"] * c["B_h"] * sum(r * d for r, d in zip(rhos, D1s))
        return x_t

    def _uni_c(self, m_t, x_last, i0, order):
        """Corrector for the step i0 -> i0+1 that the predictor already took; m_t = x0 prediction at i0+1.
        Runs BEFORE m_t is pushed, so model_outputs[-1] is the output at i0."""
        m0 = self.model_outputs[-1]
        hist = [self.model_outputs[-(k + 1)] for k in range(1, order)]
        hidx = [self.timestep_list[-(k + 1)] for k in range(1, order)]
        c = self._coeffs(i0, order, hidx)
        rhos = np.array([0.5]) if order == 1 else np.linalg.solve(c["R"], c["b"])
        corr = 0.0
        if order > 1:
            D1s = self._d1s(m0, hist, c["rks"])
            corr = sum(r * d for r, d in zip(rhos[:-1], D1s))
        x_t = c["sig_ratio"] * x_last - c["alpha1"] * c["h_phi_1"] * m0
        return x_t - c["alpha1"] * c["B_h"] * (corr + rhos[-1] * (m_t - m0))

    # ------------------------------------------------------------------ one step
    def step(self, model_output, timestep, sample, return_dict: bool = True, **_):
        if len(self.timesteps) == 0:
            raise ValueError("call set_timesteps first")
        eps, like = _to_np(model_output)
        x, like_x = _to_np(sample)
        like = like_x if like_x is not None else like
        out_dtype = x.dtype
        eps = eps.astype(np.float64)
        x = x.astype(np.float64)
        i = self._step_index
        if int(timestep) != int(self.timesteps[i]):
            raise ValueError(f"step {i} expects timestep {int(self.timesteps[i])}, got {int(timestep)}")
        m_t = (x - self._sigma[i] * eps) / self._alpha[i]              # epsilon -> x0 prediction
        use_corrector = i > 0 and (i - 1) not in self.disable_corrector and self.last_sample is not None
        if use_corrector:
            x = self._uni_c(m_t, self.last_sample, i - 1, self.this_order)
        self.model_outputs = self.model_outputs[1:] + [m_t]
        self.timestep_list = self.timestep_list[1:] + [i]
        order = min(self.solver_order, len(self.timesteps) - i) if self.lower_order_final else self.solver_order
        self.this_order = min(order, self.lower_order_nums + 1)        # warm-up: orders 1, 2, ...
        if not np.isfinite(self._lambda[i + 1]):
            self.this_order = 1                                        # the step onto sigma = 0 is x0 itself
        self.last_sample = x
        prev = self._uni_p(x, i, self.this_order)
        if self.lower_order_nums < self.solver_order:
            self.lower_order_nums += 1
        self._step_index += 1
        prev = prev.astype(out_dtype)
        if like is not None:
            import torch
            prev = torch.from_numpy(prev).to(like.device)
        if not return_dict:
            return (prev,)
        return {"prev_sample": prev}
