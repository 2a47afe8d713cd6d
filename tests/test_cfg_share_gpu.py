"""The shared front of a CFG batch (option "cfg_share", pd_engine::forward_eps).

p_sample_ddim evaluates the networks on cat([x] * 2), cat([t] * 2) and [unconditional ; conditional] conditioning
(cldm/ddim_hacked.py:189-192): until the first cross-attention reads the two contexts, sample j and sample B + j carry the same
numbers.  The engine computes that front once per pair.  These tests pin that it is the same function: against the oracle (which
runs the doubled batch layer by layer like the reference), against the engine with the option off, and that the engine falls back
to the doubled batch whenever the halves do NOT share their inputs (own unconditional pair / query, guess mode, no guidance,
per-sample timesteps through pd_eps)."""
import numpy as np
import pytest

from oracle import pd_oracle as O
from prompt_diffusion_amd import engine as E
from prompt_diffusion_amd import weights as W

pytestmark = pytest.mark.gpu


def relerr(a, b):
    return float(np.abs(np.asarray(a) - b).max() / (np.abs(b).max() + 1e-30))


@pytest.fixture(scope="module")
def net():
    return W.synth_state_dict(W.TINY), O.make_layouts(W.TINY, W)


def _kw(inp, steps, scale):
    return dict(x_T=inp["x_T"], ctx_cond=inp["ctx_cond"], ctx_uncond=inp["ctx_uncond"], pair=inp["pair"], query=inp["query"],
                steps=steps, cfg_scale=scale)


# (oracle bound, shared-vs-doubled bound): the fp32-class modes are the reference's arithmetic up to summation order; in the 2-byte
# modes a half-size launch may pick another tile / split-K shape, i.e. another rounding pattern of the same sums
@pytest.mark.parametrize("prec,tol,tol_ab", [("f32", 2e-4, 2e-5), ("f16x2", 2e-4, 2e-5), ("f16", 2e-2, 1e-2), ("bf16", 1.5e-1, 8e-2)])
def test_shared_front_is_the_doubled_batch(net, prec, tol, tol_ab):
    sd, lay = net
    cfg, B, h, w, S = W.TINY, 2, 8, 8, 4
    inp = W.synth_inputs(cfg, B, h, w, seed=91)
    cond = dict(c_crossattn=inp["ctx_cond"], example_pair=inp["pair"], query=inp["query"])
    unc = dict(c_crossattn=inp["ctx_uncond"], example_pair=inp["pair"], query=inp["query"])
    ref, _, _ = O.ddim_sampling(sd, cfg, lay, S, inp["x_T"], cond, unc, 7.5)
    e = E.Engine(cfg, precision=prec)
    try:
        e.load_state_dict(sd)
        got = e.ddim_sample(**_kw(inp, S, 7.5))
        assert e.stat("cfg_shared") == 3          # UNet and ControlNet fronts both ran once per pair
        n_shared = e.stat("launches")
        e.set_option("cfg_share", 0)
        base = e.ddim_sample(**_kw(inp, S, 7.5))
        assert e.stat("cfg_shared") == 0
        n_doubled = e.stat("launches") - n_shared
        assert np.isfinite(got).all()
        assert relerr(got, ref) < tol, (prec, relerr(got, ref))
        assert relerr(base, ref) < tol, (prec, relerr(base, ref))
        assert relerr(got, base) < tol_ab, (prec, relerr(got, base))
        print(f"[cfg_share] {prec}: shared vs oracle {relerr(got, ref):.2e}, doubled vs oracle {relerr(base, ref):.2e}, "
              f"shared vs doubled {relerr(got, base):.2e}; launches {n_shared} / {n_doubled}")
    finally:
        e.close()


def test_shared_front_only_when_the_halves_share_their_inputs(net):
    sd, lay = net
    cfg, B, h, w, S = W.TINY, 2, 8, 8, 2
    inp = W.synth_inputs(cfg, B, h, w, seed=92)
    other = W.synth_inputs(cfg, B, h, w, seed=93)
    e = E.Engine(cfg, precision="f32")
    try:
        e.load_state_dict(sd)
        kw = _kw(inp, S, 5.0)
        # own unconditional example pair / query: the UNet's front is still shared (it sees x and t only), the ControlNet's is not
        cond = dict(c_crossattn=inp["ctx_cond"], example_pair=inp["pair"], query=inp["query"])
        unc = dict(c_crossattn=inp["ctx_uncond"], example_pair=other["pair"], query=other["query"])
        ref, _, _ = O.ddim_sampling(sd, cfg, lay, S, inp["x_T"], cond, unc, 5.0)
        got = e.ddim_sample(pair_uncond=other["pair"], query_uncond=other["query"], **kw)
        assert e.stat("cfg_shared") == 1
        assert relerr(got, ref) < 2e-4
        # only the query differs
        unc2 = dict(c_crossattn=inp["ctx_uncond"], example_pair=inp["pair"], query=other["query"])
        ref2, _, _ = O.ddim_sampling(sd, cfg, lay, S, inp["x_T"], cond, unc2, 5.0)
        got2 = e.ddim_sample(query_uncond=other["query"], **kw)
        assert e.stat("cfg_shared") == 1
        assert relerr(got2, ref2) < 2e-4
        # guess mode zeroes the unconditional half of every control tensor ((D) pipeline :1248-1253): the ControlNet's front is not shared;
        # it runs on the conditional half alone instead, like the reference's (:1220-1224; bit 2 of the stat), and the result equals the
        # doubled-batch evaluation with the zeroing afterwards (the option off)
        g1 = e.ddim_sample(guess_mode=True, **kw)
        assert e.stat("cfg_shared") == 5
        e.set_option("cfg_share", 0)
        g0 = e.ddim_sample(guess_mode=True, **kw)
        e.set_option("cfg_share", 1)
        assert relerr(g1, g0) < 2e-5
        # only_mid_control: the skip tensors of the front are still read at half batch
        unc3 = dict(c_crossattn=inp["ctx_uncond"], example_pair=inp["pair"], query=inp["query"])
        ref3, _, _ = O.ddim_sampling(sd, cfg, lay, S, inp["x_T"], cond, unc3, 5.0, only_mid_control=True)
        got3 = e.ddim_sample(only_mid_control=True, **kw)
        assert e.stat("cfg_shared") == 3
        assert relerr(got3, ref3) < 2e-4
        # no guidance: nothing to share
        e.ddim_sample(use_cfg=False, **kw)
        assert e.stat("cfg_shared") == 0
    finally:
        e.close()


def test_shared_front_stepwise_and_eps_at(net):
    """The stepwise session (callbacks, set_latents, eps_at of encode / decode) goes through the same forward."""
    sd, lay = net
    cfg, B, h, w, S = W.TINY, 1, 8, 8, 4
    inp = W.synth_inputs(cfg, B, h, w, seed=94)
    e = E.Engine(cfg, precision="f32")
    try:
        e.load_state_dict(sd)
        kw = _kw(inp, S, 7.5)
        fused = e.ddim_sample(**kw)
        e.sample_begin(**kw)
        for i in range(S):
            e.sample_step(i)
        step = e.sample_get()
        eps1 = e.sample_eps_at(301)
        assert e.stat("cfg_shared") == 3
        e.sample_end()
        assert np.array_equal(fused, step)
        e.set_option("cfg_share", 0)
        e.sample_begin(**kw)
        for i in range(S):
            e.sample_step(i)
        eps0 = e.sample_eps_at(301)
        e.sample_end()
        assert relerr(eps1, eps0) < 2e-5
    finally:
        e.close()


@pytest.mark.parametrize("prec", ["f16", "f32"])
def test_shared_front_sd15_first_block(prec):
    """SD1.5 widths at a 16 x 16 latent (N = 256 tokens: the 320-channel level takes the fused st_front / st_tail kernels in the
    2-byte modes, whose tail reads the shared att / h / x rows twice through `in_rows`)."""
    cfg, B, h, w = W.SD15, 2, 16, 16
    inp = W.synth_inputs(cfg, B, h, w, seed=95)
    e = E.Engine(cfg, precision=prec)
    try:
        e.init_random_weights(777)
        kw = _kw(inp, 50, 7.5)
        e.sample_begin(**kw)
        e.sample_step(0)
        a = e.sample_get()
        assert e.stat("cfg_shared") == 3
        e.sample_end()
        e.set_option("cfg_share", 0)
        e.sample_begin(**kw)
        e.sample_step(0)
        b = e.sample_get()
        e.sample_end()
        assert np.isfinite(a).all()
        err = relerr(a, b)
        print(f"[cfg_share] SD1.5 16x16 {prec}: shared vs doubled after one step {err:.2e}, bit-identical {np.array_equal(a, b)}")
        assert err < (5e-4 if prec == "f16" else 1e-5)
    finally:
        e.close()
