"""The oracle (oracle/pd_oracle.py) against fixtures produced by the reference itself
(tests/golden/make_golden.py).  CPU only.  Tolerances are fp32 round-off of two different
summation orders (torch/oneDNN vs NumPy/OpenBLAS): 2e-5 relative to the tensor's max."""
import json
import os

import numpy as np
import pytest

from oracle import pd_oracle as O
from prompt_diffusion_amd import weights as W


def relerr(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def test_schedule_matches_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "schedule.npz"))
    _, ac = O.register_schedule()
    np.testing.assert_array_equal(ac, g["alphas_cumprod"])
    for S, eta in ((5, 0.0), (50, 0.0), (20, 0.0), (50, 0.5), (10, 1.0)):
        s = O.make_schedule(S, eta)
        tag = f"S{S}_eta{eta}"
        np.testing.assert_array_equal(s["ddim_timesteps"], g[tag + "_timesteps"])
        for k in ("ddim_alphas", "ddim_alphas_prev", "ddim_sigmas", "ddim_sqrt_one_minus_alphas"):
            np.testing.assert_allclose(s[k], g[f"{tag}_{k}"], rtol=3e-7, atol=0, err_msg=f"{tag} {k}")
    assert list(O.make_schedule(5)["ddim_timesteps"]) == [1, 201, 401, 601, 801]


def test_timestep_embedding(golden_dir):
    g = np.load(os.path.join(golden_dir, "ops.npz"))
    for dim in (320, 64):
        got = O.timestep_embedding(g["temb_t"], dim)
        # arguments reach 999 rad; a 1-ulp difference in expf (torch vs NumPy) moves cos/sin by <= 999*6e-8
        np.testing.assert_allclose(got, g[f"temb_{dim}"], atol=1e-4)


def _op_params(golden_dir):
    with open(os.path.join(golden_dir, "ops_params.json")) as f:
        spec = json.load(f)
    return {n: W.synth_tensor(n, s, k, seed=99) for n, s, k in spec}


def test_ops_match_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "ops.npz"))
    sd = _op_params(golden_dir)
    for tag in ("res_a", "res_b"):
        y = O.resblock(O.Net(sd, tag + "."), "", g[tag + "_x"], g[tag + "_emb"])
        assert relerr(y, g[tag + "_y"]) < 2e-5, tag
    for tag, heads in (("st_a", 4), ("st_b", 8)):
        y = O.spatial_transformer(O.Net(sd, tag + "."), "", g[tag + "_x"], g[tag + "_ctx"], heads)
        assert relerr(y, g[tag + "_y"]) < 2e-5, tag
    y = O.cross_attention(O.Net(sd, "ca."), "", g["ca_x"], g["ca_ctx"], heads=2)
    assert relerr(y, g["ca_y"]) < 2e-5
    p = O.Net(sd, "down.")
    assert relerr(O.conv2d(g["down_x"], p("op.weight"), p("op.bias"), stride=2), g["down_y"]) < 2e-5
    p = O.Net(sd, "up.")
    up = np.repeat(np.repeat(g["down_x"], 2, axis=2), 2, axis=3)
    assert relerr(O.conv2d(up, p("conv.weight"), p("conv.bias")), g["up_y"]) < 2e-5


def test_param_inventory_matches_reference(golden_dir):
    with open(os.path.join(golden_dir, "state_dict_spec_sd15.json")) as f:
        ref = json.load(f)
    mine_u = [[n[len(W.UNET_PREFIX):], list(s)] for n, s, _ in W.unet_spec(W.SD15)]
    mine_c = [[n[len(W.CNET_PREFIX):], list(s)] for n, s, _ in W.controlnet_spec(W.SD15)]
    assert mine_u == ref["unet"]
    assert mine_c == ref["controlnet"]
    assert W.num_params(W.SD15) == (859520964, 362366032)   # run_prompt_diffusion.ipynb:70, SURVEY §6


def _replay(golden_dir, tag, cfg, steps_limit=None):
    g = np.load(os.path.join(golden_dir, f"net_{tag}.npz"))
    B, h, w, S = int(g["B"]), int(g["h"]), int(g["w"]), int(g["S"])
    sd = W.synth_state_dict(cfg)
    lay = O.make_layouts(cfg, W)
    inp = W.synth_inputs(cfg, B, h, w)
    cond = dict(c_crossattn=inp["ctx_cond"], example_pair=inp["pair"], query=inp["query"])
    unc = dict(c_crossattn=inp["ctx_uncond"], example_pair=inp["pair"], query=inp["query"])
    return g, sd, lay, inp, cond, unc, (B, h, w, S)


@pytest.mark.parametrize("tag", ["tiny_b2_16x16_s5", "tiny_b1_8x24_s4"])
def test_tiny_network_and_sampler(golden_dir, tag):
    cfg = W.TINY
    g, sd, lay, inp, cond, unc, (B, h, w, S) = _replay(golden_dir, tag, cfg)
    # apply_model on the CFG batch at the first step, plus the 13 control residuals
    x_in = np.concatenate([inp["x_T"]] * 2)
    t_in = np.full((2 * B,), int(g["first_step"]), dtype=np.int64)
    ctx = np.concatenate([inp["ctx_uncond"], inp["ctx_cond"]])
    pair = np.concatenate([inp["pair"]] * 2)
    qry = np.concatenate([inp["query"]] * 2)
    eps, control = O.apply_model(sd, cfg, lay, x_in, t_in, ctx, pair, qry, return_control=True)
    assert relerr(eps, g["eps"]) < 5e-5
    for i, c in enumerate(control):
        assert tuple(c.shape) == tuple(g[f"control_{i}_shape"])
        if f"control_{i}" in g:
            assert relerr(c, g[f"control_{i}"]) < 5e-5, i
        else:
            st = int(g[f"control_{i}_stride"])
            sub = c.reshape(-1)[::st][:4096]
            assert relerr(sub, g[f"control_{i}_sub"]) < 5e-5, i
    # whole DDIM trajectory (x_inter has S+1 latents, ddim_hacked.py:174-176 with log_every_t=1)
    _, x_inter, preds = O.ddim_sampling(sd, cfg, lay, S, inp["x_T"], cond, unc, float(g["cfg_scale"]),
                                        eta=float(g["eta"]))
    for i in range(S + 1):
        assert relerr(x_inter[i], g["x_inter"][i]) < 1e-4, i
        assert relerr(preds[i], g["pred_x0"][i]) < 1e-4, i


def test_inpainting_blend_matches_reference(golden_dir):
    """DDIMSampler.sample(mask=, x0=) (ddim_hacked.py:154-157 + DDPM.q_sample): reference run with the q_sample noise
    draws recorded; the oracle replays the blend with the same draws."""
    cfg = W.TINY
    g = np.load(os.path.join(golden_dir, "net_tiny_mask_b2_16x16_s5.npz"))
    B, h, w, S = int(g["B"]), int(g["h"]), int(g["w"]), int(g["S"])
    inp = W.synth_inputs(cfg, B, h, w, seed=int(g["seed"]))
    sd = W.synth_state_dict(cfg)
    lay = O.make_layouts(cfg, W)
    cond = dict(c_crossattn=inp["ctx_cond"], example_pair=inp["pair"], query=inp["query"])
    unc = dict(c_crossattn=inp["ctx_uncond"], example_pair=inp["pair"], query=inp["query"])
    samples, x_inter, _ = O.ddim_sampling(sd, cfg, lay, S, inp["x_T"], cond, unc, float(g["cfg_scale"]),
                                          mask=g["mask"], x0=g["x0"], q_noise=g["q_noise"])
    for i in range(S + 1):
        assert relerr(x_inter[i], g["x_inter"][i]) < 1e-4, i
    assert relerr(samples, g["samples"]) < 1e-4


def test_sampler_extras_match_reference(golden_dir):
    """ucg_schedule, the 'quad' grid + decode, encode and stochastic_encode (ddim_hacked.py:159-161, :237-318; util.py:49-50)
    of the reference sampler on the reduced network, replayed by the oracle."""
    cfg = W.TINY
    g = np.load(os.path.join(golden_dir, "sampler_extras_tiny.npz"))
    B, h, w = int(g["B"]), int(g["h"]), int(g["w"])
    inp = W.synth_inputs(cfg, B, h, w, seed=int(g["seed"]))
    sd = W.synth_state_dict(cfg)
    lay = O.make_layouts(cfg, W)
    cond = dict(c_crossattn=inp["ctx_cond"], example_pair=inp["pair"], query=inp["query"])
    unc = dict(c_crossattn=inp["ctx_uncond"], example_pair=inp["pair"], query=inp["query"])
    _, x_inter, _ = O.ddim_sampling(sd, cfg, lay, 5, inp["x_T"], cond, unc, 7.5, ucg_schedule=list(g["ucg_schedule"]))
    for i in range(6):
        assert relerr(x_inter[i], g["ucg_x_inter"][i]) < 1e-4, i
    ts = O.make_ddim_timesteps_quad(6)
    np.testing.assert_array_equal(ts, g["quad_timesteps"])
    sq = O.make_schedule(6, 0.0, timesteps=ts)
    for k in ("ddim_alphas", "ddim_alphas_prev", "ddim_sigmas", "ddim_sqrt_one_minus_alphas"):
        np.testing.assert_allclose(sq[k], g["quad_" + k], rtol=3e-7, atol=0, err_msg=k)
    assert relerr(O.ddim_decode(sd, cfg, lay, sq, inp["x_T"], cond, unc, 6, 5.0), g["quad_decode"]) < 1e-4
    assert relerr(O.ddim_decode(sd, cfg, lay, sq, inp["x_T"], cond, None, 4, 1.0), g["quad_decode_t4"]) < 1e-4
    su = O.make_schedule(5, 0.0)
    assert relerr(O.ddim_encode(sd, cfg, lay, su, g["enc_x0"], cond, 4), g["enc_out"]) < 1e-4
    assert relerr(O.stochastic_encode(su, g["enc_x0"], g["senc_t"], g["senc_noise"]), g["senc_out"]) < 1e-6


@pytest.mark.slow
def test_sd15_config1_first_and_last_step(golden_dir):
    """BASELINE config #1 (256x256, 5 DDIM steps, bs 1): replay steps 0 and 4 of the reference
    trajectory, each from the reference's own previous latent."""
    path = os.path.join(golden_dir, "net_sd15_b1_32x32_s5.npz")
    if not os.path.exists(path):
        pytest.skip("sd15 fixture not generated")
    cfg = W.SD15
    g, sd, lay, inp, cond, unc, (B, h, w, S) = _replay(golden_dir, "sd15_b1_32x32_s5", cfg)
    sched = O.make_schedule(S)
    tr = np.flip(sched["ddim_timesteps"])
    for i in (0, S - 1):
        x_prev, pred, _ = O.p_sample_ddim(sd, cfg, lay, sched, g["x_inter"][i], cond, unc, S - i - 1,
                                          int(tr[i]), float(g["cfg_scale"]))
        assert relerr(x_prev, g["x_inter"][i + 1]) < 1e-4, i
        assert relerr(pred, g["pred_x0"][i + 1]) < 1e-4, i


@pytest.mark.slow
def test_sd15_config1_real_images_first_step(golden_dir):
    """The same configuration on the README's example images (fixture net_sd15_b1_32x32_s5_house.npz: pixel arrays + the
    reference's trajectory): one eps evaluation and the first DDIM step of the oracle against the reference's."""
    path = os.path.join(golden_dir, "net_sd15_b1_32x32_s5_house.npz")
    if not os.path.exists(path):
        pytest.skip("real-image fixture not generated")
    g = np.load(path)
    cfg = W.SD15
    to_m11 = lambda u8: (u8.astype(np.float32) / 127.5 - 1.0).transpose(2, 0, 1)[None]
    inp = W.synth_inputs(cfg, 1, 32, 32)
    pair = np.concatenate([to_m11(g["image_a_u8"]), to_m11(g["image_b_u8"])], axis=1)
    query = to_m11(g["query_u8"])
    sd = W.synth_state_dict(cfg)
    lay = O.make_layouts(cfg, W)
    cond = dict(c_crossattn=inp["ctx_cond"], example_pair=pair, query=query)
    unc = dict(c_crossattn=inp["ctx_uncond"], example_pair=pair, query=query)
    S = int(g["S"])
    sched = O.make_schedule(S)
    tr = np.flip(sched["ddim_timesteps"])
    x_prev, _, _ = O.p_sample_ddim(sd, cfg, lay, sched, g["x_inter"][0], cond, unc, S - 1, int(tr[0]), float(g["cfg_scale"]))
    assert relerr(x_prev, g["x_inter"][1]) < 1e-4


@pytest.mark.parametrize("tag,cfg", [("tiny", W.TINY), ("sd15", W.SD15)])
def test_vae_decoder_matches_reference(golden_dir, tag, cfg):
    """decode_first_stage = z/scale_factor -> post_quant_conv -> Decoder."""
    g = np.load(os.path.join(golden_dir, "vae.npz"))
    sd = W.synth_vae_state_dict(cfg)
    if tag == "sd15":
        ref = json.loads(str(g["sd15_spec"]))
        mine = [[n[len(W.VAE_PREFIX):], list(s)] for n, s, _ in W.vae_spec(cfg)]
        assert mine == ref
        assert sum(int(np.prod(s)) for _, s in ref) == 49490199   # 49.5 M, SURVEY N1
    x = O.vae_decode(sd, cfg, W.vae_layout(cfg), g[tag + "_z"])
    assert x.shape == g[tag + "_x"].shape
    assert relerr(x, g[tag + "_x"]) < 5e-5


@pytest.mark.parametrize("tag,cfg", [("tiny_b3", W.TINY), ("sd15_b2", W.SD15)])
def test_clip_text_transformer_matches_transformers(golden_dir, tag, cfg):
    """Cond stage: fixtures are `transformers.CLIPTextModel(...).last_hidden_state` -- what
    FrozenCLIPEmbedder.forward returns (ldm/modules/encoders/modules.py:118-128) -- with the seeded weights."""
    g = np.load(os.path.join(golden_dir, f"clip_{tag}.npz"))
    sd = W.synth_text_state_dict(cfg)
    assert len(sd) == 2 + 16 * cfg.text_layers + 2
    ids = g["ids"]
    assert np.array_equal(ids, W.synth_token_ids(cfg, int(g["B"])))
    z = O.clip_text_forward(sd, cfg, ids)
    if "z" in g:
        assert relerr(z, g["z"]) < 5e-5
    else:
        assert relerr(z.reshape(-1)[::int(g["z_stride"])][:16384], g["z_sub"]) < 5e-5
    st = g["z_stats"]
    assert abs(float(np.abs(z).mean()) - st[1]) < 1e-4 * st[1]
    # clip_skip = 2 as the (D) pipeline derives it: hidden_states[-3] through final_layer_norm (:403-413)
    z2 = O.clip_text_forward(sd, cfg, ids, clip_skip=2)
    if "z_skip2" in g:
        assert relerr(z2, g["z_skip2"]) < 5e-5
    else:
        assert relerr(z2.reshape(-1)[::int(g["z_stride"])][:16384], g["z_skip2_sub"]) < 5e-5

