"""The rest of DDIMSampler's (L) surface through the engine, against fixtures the reference sampler itself produced
(tests/golden/make_golden.py --only extras; reduced network, fp32 engine): ucg_schedule (cldm/ddim_hacked.py:159-161),
make_schedule(ddim_discretize="quad") (util.py:49-50) + decode (:301-318), encode (:237-282, guidance 1; with guidance the
reference raises TypeError on ControlLDM's dict conditioning, mirrored), stochastic_encode (:284-299), noise_dropout (:231-232)."""
import os

import numpy as np
import pytest

from prompt_diffusion_amd import ddim as D
from prompt_diffusion_amd import engine as E
from prompt_diffusion_amd import weights as W

pytestmark = pytest.mark.gpu


def relerr(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


@pytest.fixture(scope="module")
def setup(golden_dir):
    g = np.load(os.path.join(golden_dir, "sampler_extras_tiny.npz"))
    e = E.Engine(W.TINY, precision="f32")
    for n, a in W.iter_synth(W.TINY):
        e.load_tensor(n, a)
    B, h, w = int(g["B"]), int(g["h"]), int(g["w"])
    inp = W.synth_inputs(W.TINY, B, h, w, seed=int(g["seed"]))
    cond = {"c_crossattn": [inp["ctx_cond"]], "example_pair": [inp["pair"]], "query": [inp["query"]]}
    uc = {"c_crossattn": [inp["ctx_uncond"]], "example_pair": [inp["pair"]], "query": [inp["query"]]}
    sampler = D.DDIMSampler(D.ControlLDM(e))
    yield g, sampler, inp, cond, uc, (B, h, w)
    e.close()


def test_ucg_schedule(setup):
    g, s, inp, cond, uc, (B, h, w) = setup
    _, inter = s.sample(5, B, (W.TINY.in_channels, h, w), cond, eta=0.0, x_T=inp["x_T"], unconditional_guidance_scale=7.5,
                        unconditional_conditioning=uc, log_every_t=1, verbose=False, ucg_schedule=list(g["ucg_schedule"]))
    assert len(inter["x_inter"]) == 6
    for i in range(6):
        assert relerr(inter["x_inter"][i], g["ucg_x_inter"][i]) < 2e-4, i
    with pytest.raises(AssertionError):      # ddim_hacked.py:160
        s.sample(5, B, (W.TINY.in_channels, h, w), cond, x_T=inp["x_T"], unconditional_conditioning=uc, ucg_schedule=[1.0, 2.0])


def test_quad_grid_and_decode(setup):
    g, s, inp, cond, uc, _ = setup
    s.make_schedule(6, ddim_discretize="quad", ddim_eta=0.0, verbose=False)
    np.testing.assert_array_equal(s.ddim_timesteps, g["quad_timesteps"])
    for k in ("ddim_alphas", "ddim_alphas_prev", "ddim_sigmas", "ddim_sqrt_one_minus_alphas"):
        np.testing.assert_allclose(getattr(s, k), g["quad_" + k], rtol=3e-7, atol=0, err_msg=k)
    x = s.decode(inp["x_T"], cond, 6, unconditional_guidance_scale=5.0, unconditional_conditioning=uc)
    assert relerr(x, g["quad_decode"]) < 2e-4
    x = s.decode(inp["x_T"], cond, 4, unconditional_guidance_scale=1.0, unconditional_conditioning=None)
    assert relerr(x, g["quad_decode_t4"]) < 2e-4
    with pytest.raises(NotImplementedError):     # util.py:52
        s.make_schedule(6, ddim_discretize="cosine")


def test_encode_and_stochastic_encode(setup):
    g, s, inp, cond, uc, _ = setup
    s.make_schedule(5, ddim_eta=0.0, verbose=False)
    x_enc, info = s.encode(g["enc_x0"], cond, 4, return_intermediates=2)
    assert relerr(x_enc, g["enc_out"]) < 2e-4
    assert list(info["intermediate_steps"]) == list(g["enc_inter_steps"])
    for a, b in zip(info["intermediates"], g["enc_inter"]):
        assert relerr(a, b) < 2e-4
    assert relerr(s.stochastic_encode(g["enc_x0"], g["senc_t"], noise=g["senc_noise"]), g["senc_out"]) < 1e-6
    with pytest.raises(TypeError):               # torch.cat of two dicts, ddim_hacked.py:263
        s.encode(g["enc_x0"], cond, 4, unconditional_guidance_scale=3.0, unconditional_conditioning=uc)


def test_noise_dropout_is_dropout_of_the_supplied_noise(setup):
    """F.dropout(noise, p) (ddim_hacked.py:231-232): the run with noise_dropout = p equals the run that is handed the
    already-dropped noise (same host generator state), and differs from the undropped one."""
    g, s, inp, cond, uc, (B, h, w) = setup
    shape = (W.TINY.in_channels, h, w)
    noise = np.random.default_rng(5).standard_normal((5, B) + shape).astype(np.float32)
    kw = dict(eta=0.8, x_T=inp["x_T"], unconditional_guidance_scale=3.0, unconditional_conditioning=uc, verbose=False)
    np.random.seed(123)
    a, _ = s.sample(5, B, shape, cond, noise=noise, noise_dropout=0.25, **kw)
    np.random.seed(123)
    keep = (np.random.random_sample(noise.shape) >= 0.25).astype(np.float32)
    b, _ = s.sample(5, B, shape, cond, noise=noise * keep * np.float32(1.0 / 0.75), **kw)
    c, _ = s.sample(5, B, shape, cond, noise=noise, **kw)
    np.testing.assert_array_equal(a, b)
    assert relerr(a, c) > 1e-3
    assert 0.70 < keep.mean() < 0.80
