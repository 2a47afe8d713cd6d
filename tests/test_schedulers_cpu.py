"""UniPC scheduler plug-in: PARITY UNPINNED against diffusers (its source is not in the reference
tree).  What is checked instead: (1) the product scheduler against an independent closed-form fp64 restatement in the
oracle, (2) both against the analytic probability-flow solution for Gaussian data, incl. the order of accuracy,
(3) the scheduler interface the pipeline drives."""
import numpy as np
import pytest

from oracle import pd_oracle as O
from prompt_diffusion_amd.schedulers import UniPCMultistepScheduler


def gaussian_eps(s, ac):
    """Exact noise prediction when x0 ~ N(0, s^2 I): eps(x, t) = sigma_t x / (alpha_t^2 s^2 + sigma_t^2)."""
    def f(x, t):
        a2 = ac[int(t)]
        return np.sqrt(1.0 - a2) * x / (a2 * s * s + (1.0 - a2))
    return f


def run(sched, eps_fn, x_T, n):
    sched.set_timesteps(n)
    x = x_T
    for t in sched.timesteps:
        x = sched.step(eps_fn(x, t), t, x, return_dict=False)[0]
    return x


@pytest.mark.parametrize("spacing", ["linspace", "leading", "trailing"])
@pytest.mark.parametrize("n", [5, 12, 20])
def test_unipc_matches_closed_form_oracle(spacing, n):
    sc = UniPCMultistepScheduler(timestep_spacing=spacing, steps_offset=1 if spacing == "leading" else 0)
    g = np.random.default_rng(0)
    x_T = g.standard_normal((2, 4, 8, 8))
    # a non-linear "model" so that history terms matter
    eps_fn = lambda x, t: np.tanh(x) * (0.3 + t / 1000.0) + 0.1 * np.sin(3 * x)
    got = run(sc, eps_fn, x_T, n)
    ref = O.unipc2_sample(eps_fn, x_T, sc.alphas_cumprod, sc.timesteps)
    assert np.abs(got - ref).max() < 1e-11 * max(1.0, np.abs(ref).max())


def _err_at_250(sc, s, x_T, n):
    """relative error of the (corrected) sample at grid point t = 250 against the closed-form probability-flow solution
    x_t = x_T sqrt(v(t)/v(T)), v = alpha^2 s^2 + sigma^2, for data ~ N(0, s^2 I).  t = 250 is on the linspace grid for
    every n that is a multiple of 20, so halving the step keeps the comparison point fixed."""
    ac = sc.alphas_cumprod
    v = lambda t: ac[t] * s * s + (1.0 - ac[t])
    sc.set_timesteps(n)
    x, rec = x_T, None
    for t in sc.timesteps:
        x = sc.step(gaussian_eps(s, ac)(x, t), t, x, return_dict=False)[0]
        if int(t) == 250:
            rec = sc.last_sample
    exact = x_T * np.sqrt(v(250) / v(sc.timesteps[0]))
    return float(np.abs(rec - exact).max() / np.abs(exact).max())


def test_unipc_order_of_accuracy_on_gaussian_probability_flow():
    """UniPC-p is of order p+1 with the corrector and p without it (Zhao et al. 2023, Thm 3.1 / Cor. 3.2):
    halving the step must cut the error by ~2^(p+1) resp. ~2^p.  DDIM on the same grid is first order."""
    s = 0.7
    x_T = np.random.default_rng(1).standard_normal((1, 4, 16, 16))
    for kw, lo, hi in ((dict(), 7.0, 10.0), (dict(solver_order=1), 3.5, 4.5), (dict(solver_order=3), 10.0, 20.0),
                       (dict(disable_corrector=list(range(200))), 3.5, 4.6)):
        sc = UniPCMultistepScheduler(**kw)
        e = [_err_at_250(sc, s, x_T, n) for n in (20, 40, 80)]
        assert lo < e[0] / e[1] < hi and lo < e[1] / e[2] < hi, (kw, e)
    sc = UniPCMultistepScheduler()
    e20 = _err_at_250(sc, s, x_T, 20)
    assert e20 < 1e-4
    # first-order DDIM (ddim_hacked.py:218-233, eta 0) on the same 20-point grid
    ac = sc.alphas_cumprod
    sc.set_timesteps(20)
    ts = list(sc.timesteps)
    x = x_T
    for i, t in enumerate(ts):
        if int(t) == 250:
            break
        e = gaussian_eps(s, ac)(x, t)
        x0 = (x - np.sqrt(1 - ac[t]) * e) / np.sqrt(ac[t])
        x = np.sqrt(ac[ts[i + 1]]) * x0 + np.sqrt(1 - ac[ts[i + 1]]) * e
    exact = x_T * np.sqrt((ac[250] * s * s + 1 - ac[250]) / (ac[ts[0]] * s * s + 1 - ac[ts[0]]))
    assert float(np.abs(x - exact).max() / np.abs(exact).max()) > 100 * e20


def test_unipc_interface_and_dtypes():
    import torch
    sc = UniPCMultistepScheduler()
    with pytest.raises(ValueError):
        sc.step(np.zeros(3), 0, np.zeros(3))
    sc.set_timesteps(20)
    assert len(sc.timesteps) == 20 and sc.timesteps[0] == 999 and sc.timesteps[-1] == 50
    assert sc.init_noise_sigma == 1.0 and sc.scale_model_input("x", 3) == "x"
    x = torch.randn(1, 4, 8, 8)
    out = sc.step(torch.zeros_like(x), sc.timesteps[0], x, return_dict=False)[0]
    assert isinstance(out, torch.Tensor) and out.dtype == torch.float32 and out.shape == x.shape
    with pytest.raises(ValueError, match="expects timestep"):
        sc.step(torch.zeros_like(x), 7, out)
    d = UniPCMultistepScheduler(solver_order=1)
    d.set_timesteps(4)
    xn = np.ones((2, 3), np.float32)
    assert d.step(np.zeros_like(xn), d.timesteps[0], xn)["prev_sample"].dtype == np.float32
    for bad in (dict(prediction_type="v_prediction"), dict(predict_x0=False), dict(thresholding=True), dict(beta_schedule="cosine")):
        with pytest.raises(NotImplementedError):
            UniPCMultistepScheduler(**bad)
    with pytest.raises(ValueError):
        UniPCMultistepScheduler(solver_type="midpoint")


def test_unipc_variants_run_and_close_on_the_data_prediction():
    s = 0.7
    g = np.random.default_rng(2)
    x_T = g.standard_normal((1, 4, 8, 8))
    for kw in (dict(solver_order=3), dict(solver_type="bh1"), dict(solver_order=2, lower_order_final=False), dict(solver_order=1)):
        sc = UniPCMultistepScheduler(**kw)
        ac = sc.alphas_cumprod
        sc.set_timesteps(25)
        x = x_T
        for t in sc.timesteps:
            x_in, x = x, sc.step(gaussian_eps(s, ac)(x, t), t, x, return_dict=False)[0]
        assert np.isfinite(x).all()
        v = lambda t: ac[t] * s * s + (1.0 - ac[t])
        exact = x_T * np.sqrt(v(sc.timesteps[-1]) / v(sc.timesteps[0]))
        assert float(np.abs(sc.last_sample - exact).max() / np.abs(exact).max()) < 2e-2   # sanity only: the last steps of a t-uniform grid are coarse in lambda
        # the closing step onto sigma = 0 returns the data prediction made from the last model evaluation
        t = sc.timesteps[-1]
        m = (x_in - np.sqrt(1 - ac[t]) * gaussian_eps(s, ac)(x_in, t)) / np.sqrt(ac[t])
        assert np.abs(x - m).max() < 1e-12


def test_unipc_predictor_is_reference_dpm_solver_2m():
    """The multistep logic (history, step ratios, warm-up order, lower-order final step) pinned by the REFERENCE tree's own
    DPM-Solver++ (ldm/models/diffusion/dpm_solver/dpm_solver.py:319, :723-757, :1044-1070): UniPC's order-2 predictor with
    B(h) = e^h - 1 ('bh2') and the corrector switched off is DPM-Solver++(2M), solver_type 'dpm_solver'.  The fixture
    (tests/golden/make_golden.py --only dpm) ran the reference solver in fp64 on its own time-uniform grid with an analytic
    epsilon model; here the plug-in runs on that grid's (alpha, sigma, lambda) with the same closed form."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dpm_solver_2m.npz"))
    A = g["A"]

    def eps_model(x, tc):   # the generator's closed form (it receives (t - 1/N) * 1000 from model_wrapper and undoes it)
        s = 0.3 + 0.6 * tc
        return np.tanh(np.einsum("oc,bchw->bohw", A, x) * s) + 0.25 * x * (1.0 - s)
    for steps in (8, 5):
        t = g[f"s{steps}_t"]
        s = UniPCMultistepScheduler(disable_corrector=list(range(steps)))
        s.set_timesteps(steps)
        s._alpha, s._sigma, s._lambda = g[f"s{steps}_alpha"].copy(), g[f"s{steps}_sigma"].copy(), g[f"s{steps}_lambda"].copy()
        x = g["x_T"].copy()
        states = g[f"s{steps}_x"]
        assert states.shape[0] == steps + 1
        for i in range(steps):
            np.testing.assert_allclose(x, states[i], rtol=0, atol=1e-12 * np.abs(states[i]).max())
            x = s.step(eps_model(x, t[i]), int(s.timesteps[i]), x, return_dict=False)[0]
        np.testing.assert_allclose(x, states[steps], rtol=0, atol=1e-12 * np.abs(states[steps]).max())
