"""Branches of the hot path that the fixture runs do not reach, each against the oracle on the reduced network:
`only_mid_control` (cldm/cldm.py:38-39), `temperature` != 1 with eta > 0 (ddim_hacked.py:229-233), distinct
unconditional example pair / query (the (L) sampler batches whatever the unconditional dict holds, ddim_hacked.py:188-200),
and the fp32 residual stream option of the 2-byte modes (`stream_f32`)."""
import os

import numpy as np
import pytest

from oracle import pd_oracle as O
from prompt_diffusion_amd import engine as E
from prompt_diffusion_amd import weights as W

pytestmark = pytest.mark.gpu


def relerr(a, b):
    return float(np.abs(np.asarray(a) - b).max() / (np.abs(b).max() + 1e-30))


@pytest.fixture(scope="module")
def eng():
    e = E.Engine(W.TINY, precision="f32")
    e.load_state_dict(W.synth_state_dict(W.TINY))
    yield e
    e.close()


@pytest.fixture(scope="module")
def net():
    return W.synth_state_dict(W.TINY), O.make_layouts(W.TINY, W)


def _dicts(inp, pair_u=None, query_u=None):
    cond = dict(c_crossattn=inp["ctx_cond"], example_pair=inp["pair"], query=inp["query"])
    unc = dict(c_crossattn=inp["ctx_uncond"], example_pair=inp["pair"] if pair_u is None else pair_u,
               query=inp["query"] if query_u is None else query_u)
    return cond, unc


def test_only_mid_control(eng, net):
    sd, lay = net
    cfg, B, h, w, S = W.TINY, 1, 8, 8, 2
    inp = W.synth_inputs(cfg, B, h, w, seed=51)
    cond, unc = _dicts(inp)
    kw = dict(x_T=inp["x_T"], ctx_cond=inp["ctx_cond"], ctx_uncond=inp["ctx_uncond"], pair=inp["pair"], query=inp["query"],
              steps=S, cfg_scale=6.0)
    ref, _, _ = O.ddim_sampling(sd, cfg, lay, S, inp["x_T"], cond, unc, 6.0, only_mid_control=True)
    got = eng.ddim_sample(only_mid_control=True, **kw)
    assert relerr(got, ref) < 2e-4
    full = eng.ddim_sample(**kw)
    assert relerr(full, ref) > 1e-2      # the 12 skip residuals do matter on this network


def test_temperature_with_eta(eng, net):
    sd, lay = net
    cfg, B, h, w, S = W.TINY, 2, 8, 8, 4
    inp = W.synth_inputs(cfg, B, h, w, seed=52)
    cond, unc = _dicts(inp)
    noise = np.random.default_rng(5).standard_normal((S, B, 4, h, w)).astype(np.float32)
    ref, _, _ = O.ddim_sampling(sd, cfg, lay, S, inp["x_T"], cond, unc, 7.5, eta=0.8, noises=noise, temperature=0.6)
    kw = dict(x_T=inp["x_T"], ctx_cond=inp["ctx_cond"], ctx_uncond=inp["ctx_uncond"], pair=inp["pair"], query=inp["query"],
              steps=S, cfg_scale=7.5, eta=0.8, noise=noise)
    got = eng.ddim_sample(temperature=0.6, **kw)
    assert relerr(got, ref) < 2e-4
    assert relerr(eng.ddim_sample(temperature=1.0, **kw), ref) > 1e-3


def test_distinct_unconditional_pair_and_query(eng, net):
    sd, lay = net
    cfg, B, h, w, S = W.TINY, 2, 8, 8, 2
    inp = W.synth_inputs(cfg, B, h, w, seed=53)
    other = W.synth_inputs(cfg, B, h, w, seed=54)
    cond, unc = _dicts(inp, other["pair"], other["query"])
    ref, _, _ = O.ddim_sampling(sd, cfg, lay, S, inp["x_T"], cond, unc, 5.0)
    kw = dict(x_T=inp["x_T"], ctx_cond=inp["ctx_cond"], ctx_uncond=inp["ctx_uncond"], pair=inp["pair"], query=inp["query"],
              steps=S, cfg_scale=5.0)
    got = eng.ddim_sample(pair_uncond=other["pair"], query_uncond=other["query"], **kw)
    assert relerr(got, ref) < 2e-4
    assert relerr(eng.ddim_sample(**kw), ref) > 1e-3
    # only one of the two replaced
    cond, unc = _dicts(inp, other["pair"], None)
    ref2, _, _ = O.ddim_sampling(sd, cfg, lay, S, inp["x_T"], cond, unc, 5.0)
    assert relerr(eng.ddim_sample(pair_uncond=other["pair"], **kw), ref2) < 2e-4


@pytest.mark.parametrize("prec,tol_plain,tol_stream", [("f16", 5e-3, 3.5e-3), ("bf16", 5e-2, 3.5e-2)])
def test_stream_f32_option(golden_dir, prec, tol_plain, tol_stream):
    """pd_config.stream_f32: the residual stream (block outputs) stays fp32 while the MFMA operands are 2-byte.  Against the
    reference's own trajectory it must be at least as close as the plain mode (measured: 1.5-1.7x closer)."""
    g = np.load(os.path.join(golden_dir, "net_tiny_b2_16x16_s5.npz"))
    B, h, w, S = int(g["B"]), int(g["h"]), int(g["w"]), int(g["S"])
    inp = W.synth_inputs(W.TINY, B, h, w)
    errs = {}
    for sf in (False, True):
        e = E.Engine(W.TINY, precision=prec, stream_f32=sf)
        e.load_state_dict(W.synth_state_dict(W.TINY))
        out, inter = e.ddim_sample(x_T=inp["x_T"], ctx_cond=inp["ctx_cond"], ctx_uncond=inp["ctx_uncond"], pair=inp["pair"],
                                   query=inp["query"], steps=S, cfg_scale=float(g["cfg_scale"]), return_intermediates=True)
        errs[sf] = max(relerr(inter[i], g["x_inter"][i]) for i in range(1, S + 1))
        e.close()
    print(prec, "per-step latent error: plain %.2e, stream_f32 %.2e" % (errs[False], errs[True]))
    assert errs[False] < tol_plain and errs[True] < tol_stream and errs[True] < errs[False]


@pytest.mark.parametrize("tag", ["tiny_b2_16x16_s5", "tiny_b1_8x24_s4"])
def test_layernorm_fold_in_fp32_mode(golden_dir, tag):
    """Option ln_fuse: norm1 and norm2 of a transformer block folded into their consumer GEMMs (weights * gamma,
    rstd * (acc - mean * colsum) + beta . W^T in the epilogue, row statistics from the producing GEMM's epilogue or the
    row-statistics kernel).  It is the default of the 2-byte modes; forced on in the fp32 mode it must reproduce the
    reference's own run to the fp32 bound."""
    g = np.load(os.path.join(golden_dir, f"net_{tag}.npz"))
    B, h, w, S = int(g["B"]), int(g["h"]), int(g["w"]), int(g["S"])
    inp = W.synth_inputs(W.TINY, B, h, w)
    e = E.Engine(W.TINY, precision="f32")
    e.set_option("ln_fuse", 1)
    e.load_state_dict(W.synth_state_dict(W.TINY))
    n0 = e.stat("launches")
    out, inter = e.ddim_sample(x_T=inp["x_T"], ctx_cond=inp["ctx_cond"], ctx_uncond=inp["ctx_uncond"], pair=inp["pair"],
                               query=inp["query"], steps=S, cfg_scale=float(g["cfg_scale"]), return_intermediates=True)
    n_fused = e.stat("launches") - n0
    for i in range(S + 1):
        assert relerr(inter[i], g["x_inter"][i]) < 2e-4, i
    e.set_option("ln_fuse", 0)
    n0 = e.stat("launches")
    out2 = e.ddim_sample(x_T=inp["x_T"], ctx_cond=inp["ctx_cond"], ctx_uncond=inp["ctx_uncond"], pair=inp["pair"],
                         query=inp["query"], steps=S, cfg_scale=float(g["cfg_scale"]))
    assert relerr(out, out2) < 2e-4 and e.stat("launches") - n0 > n_fused     # the fold removes launches
    e.close()
