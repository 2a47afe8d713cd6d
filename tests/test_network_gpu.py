"""End-to-end parity of the HIP path (through the C ABI) with the golden fixtures generated from the
reference and with the NumPy oracle.

Tolerances (max |diff| / max |ref| per tensor); the north-star bound is 1e-3 per denoising step:
  fp32 engine mode  : 2e-4 per step on every fixture -- the reference's arithmetic, different summation order.
  f16x2 engine mode : 1e-3 per step on every fixture (split fp16 operands over fp32 storage; measured ~1e-5).
  fp16 engine mode  : the benchmarked default (the reference's own GPU dtype).  On the headline 50-step schedule
                      (fixtures net_sd15_b1_{32x32,64x64}_s50, produced by the reference sampler itself) every single
                      denoising step is within 1e-3 of the reference's step from the same latent.  On the 5- and 4-step
                      fixtures one step moves the latent ~8x further (the eps coefficient of the DDIM update), so the
                      same eps error shows as ~3.8e-3 there: bounds 5e-3 (eps 2.5e-3).
  bf16 engine mode  : eps within 3e-2; latents within 5e-2 per step on the 5-step configs (2^-9 operand rounding,
                      ~100 contraction layers deep, CFG 7.5 amplifies the uncorrelated part ~10x).
See DESIGN.md "Precision" for the error budget (tools/prec_probe.py reproduces it).
"""
import json
import os

import numpy as np
import pytest

from oracle import pd_oracle as O
from prompt_diffusion_amd import engine as E
from prompt_diffusion_amd import weights as W

pytestmark = pytest.mark.gpu


def relerr(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def _engine(cfg, prec, **kw):
    e = E.Engine(cfg, precision=prec, **kw)
    for n, a in W.iter_synth(cfg):
        e.load_tensor(n, a)
    assert e.weights_missing() == 0
    return e


@pytest.fixture(scope="module")
def tiny_f32():
    e = _engine(W.TINY, "f32")
    yield e
    e.close()


def test_param_registry_matches_reference_checkpoint(golden_dir):
    e = E.Engine(W.SD15, precision="bf16")
    with open(os.path.join(golden_dir, "state_dict_spec_sd15.json")) as f:
        ref = json.load(f)
    want = [(W.UNET_PREFIX + n, tuple(s)) for n, s in ref["unet"]] + [(W.CNET_PREFIX + n, tuple(s)) for n, s in ref["controlnet"]]
    names = e.param_names()
    assert names[:len(want)] == want
    # the optional first-stage decoder follows, under the checkpoint's first_stage_model.* names
    nv = len(W.vae_spec(W.SD15))
    assert names[len(want):len(want) + nv] == [(n, tuple(s)) for n, s, _ in W.vae_spec(W.SD15)]
    # ... and the cond-stage text transformer, under cond_stage_model.transformer.text_model.* in module order
    assert names[len(want) + nv:] == [(n, tuple(s)) for n, s, _ in W.text_spec(W.SD15)]
    e.close()


def test_schedule_matches_reference(golden_dir, tiny_f32):
    g = np.load(os.path.join(golden_dir, "schedule.npz"))
    for S, eta in ((5, 0.0), (50, 0.0), (20, 0.0), (50, 0.5), (10, 1.0)):
        s = tiny_f32.make_schedule(S, eta)
        tag = f"S{S}_eta{eta}"
        np.testing.assert_array_equal(s["ddim_timesteps"], g[tag + "_timesteps"])
        for k in ("ddim_alphas", "ddim_alphas_prev", "ddim_sigmas", "ddim_sqrt_one_minus_alphas"):
            np.testing.assert_allclose(s[k], g[f"{tag}_{k}"], rtol=3e-7, atol=0, err_msg=f"{tag} {k}")


def _cfg_inputs(cfg, g):
    B, h, w = int(g["B"]), int(g["h"]), int(g["w"])
    inp = W.synth_inputs(cfg, B, h, w)
    x_in = np.concatenate([inp["x_T"]] * 2)
    t_in = np.full((2 * B,), int(g["first_step"]), dtype=np.int64)
    ctx = np.concatenate([inp["ctx_uncond"], inp["ctx_cond"]])
    pair = np.concatenate([inp["pair"]] * 2)
    qry = np.concatenate([inp["query"]] * 2)
    return inp, x_in, t_in, ctx, pair, qry


@pytest.mark.parametrize("tag", ["tiny_b2_16x16_s5", "tiny_b1_8x24_s4"])
def test_tiny_apply_model_f32(golden_dir, tiny_f32, tag):
    g = np.load(os.path.join(golden_dir, f"net_{tag}.npz"))
    inp, x_in, t_in, ctx, pair, qry = _cfg_inputs(W.TINY, g)
    eps, control = tiny_f32.eps(x_in, t_in, ctx, pair, qry, return_control=True)
    for i, c in enumerate(control):
        assert tuple(c.shape) == tuple(g[f"control_{i}_shape"])
        if f"control_{i}" in g:
            assert relerr(c, g[f"control_{i}"]) < 1e-4, f"control {i}"
        else:
            st = int(g[f"control_{i}_stride"])
            assert relerr(c.reshape(-1)[::st][:4096], g[f"control_{i}_sub"]) < 1e-4, f"control {i}"
    assert relerr(eps, g["eps"]) < 1e-4


@pytest.mark.parametrize("tag", ["tiny_b2_16x16_s5", "tiny_b1_8x24_s4"])
def test_tiny_ddim_trajectory_f32(golden_dir, tiny_f32, tag):
    g = np.load(os.path.join(golden_dir, f"net_{tag}.npz"))
    B, h, w, S = int(g["B"]), int(g["h"]), int(g["w"]), int(g["S"])
    inp = W.synth_inputs(W.TINY, B, h, w)
    out, inter = tiny_f32.ddim_sample(x_T=inp["x_T"], ctx_cond=inp["ctx_cond"], ctx_uncond=inp["ctx_uncond"],
                                      pair=inp["pair"], query=inp["query"], steps=S, cfg_scale=float(g["cfg_scale"]),
                                      eta=0.0, return_intermediates=True)
    assert inter.shape == g["x_inter"].shape
    for i in range(S + 1):
        assert relerr(inter[i], g["x_inter"][i]) < 2e-4, f"step {i}"
    assert relerr(out, g["samples"]) < 2e-4


@pytest.mark.parametrize("prec,tol_eps,tol_step", [("bf16", 3e-2, 5e-2), ("f16", 2.5e-3, 5e-3), ("f16x2", 2e-4, 1e-3)])
@pytest.mark.parametrize("tag", ["tiny_b2_16x16_s5", "tiny_b1_8x24_s4"])
def test_tiny_reduced_precision(golden_dir, tag, prec, tol_eps, tol_step):
    e = _engine(W.TINY, prec)
    g = np.load(os.path.join(golden_dir, f"net_{tag}.npz"))
    inp, x_in, t_in, ctx, pair, qry = _cfg_inputs(W.TINY, g)
    eps = e.eps(x_in, t_in, ctx, pair, qry)
    assert relerr(eps, g["eps"]) < tol_eps
    B, h, w, S = int(g["B"]), int(g["h"]), int(g["w"]), int(g["S"])
    out, inter = e.ddim_sample(x_T=inp["x_T"], ctx_cond=inp["ctx_cond"], ctx_uncond=inp["ctx_uncond"],
                               pair=inp["pair"], query=inp["query"], steps=S, cfg_scale=float(g["cfg_scale"]),
                               return_intermediates=True)
    errs = [relerr(inter[i], g["x_inter"][i]) for i in range(S + 1)]
    print(prec, "per-step latent error", ["%.2e" % v for v in errs])
    assert np.isfinite(out).all() and max(errs) < tol_step
    e.close()


def test_fused_groupnorm_option_matches(golden_dir):
    """Option 'gn_fuse': GroupNorm(+SiLU) applied inside the patch conv's LDS staging (SD1.5 shapes are patch-eligible)."""
    g = np.load(os.path.join(golden_dir, "net_sd15_b1_32x32_s5.npz"))
    e = _engine(W.SD15, "f32")
    e.set_option("gn_fuse", 1)
    inp, x_in, t_in, ctx, pair, qry = _cfg_inputs(W.SD15, g)
    # batch 2 x 32x32 gives 8 patches x 2 n-tiles = 16 blocks (< 192): widen the batch so the patch kernel is chosen
    rep = 12
    eps = e.eps(np.tile(x_in, (rep, 1, 1, 1)), np.tile(t_in, rep), np.tile(ctx, (rep, 1, 1)), np.tile(pair, (rep, 1, 1, 1)),
                np.tile(qry, (rep, 1, 1, 1)))
    assert relerr(eps[:2], g["eps"]) < 2e-4 and relerr(eps[-2:], g["eps"]) < 2e-4
    e.close()


@pytest.mark.parametrize("prec,tol", [("f32", 1e-4), ("f16", 4e-3), ("bf16", 3e-2)])
@pytest.mark.parametrize("tag,cfg", [("tiny", W.TINY), ("sd15", W.SD15)])
def test_vae_decode_matches_reference(golden_dir, prec, tol, tag, cfg):
    """decode_first_stage through the engine vs the reference Decoder's own output."""
    g = np.load(os.path.join(golden_dir, "vae.npz"))
    e = E.Engine(cfg, precision=prec)
    for n, s_, k in W.vae_spec(cfg):
        e.load_tensor(n, W.synth_tensor(n, s_, k))
    assert e.vae_weights_missing() == 0 and e.weights_missing() > 0
    x = e.vae_decode(g[tag + "_z"])
    assert x.shape == g[tag + "_x"].shape
    assert relerr(x, g[tag + "_x"]) < tol
    e.close()


def test_stepwise_equals_fused(tiny_f32):
    inp = W.synth_inputs(W.TINY, 1, 8, 8)
    kw = dict(x_T=inp["x_T"], ctx_cond=inp["ctx_cond"], ctx_uncond=inp["ctx_uncond"], pair=inp["pair"],
              query=inp["query"], steps=4, cfg_scale=5.0)
    fused = tiny_f32.ddim_sample(**kw)
    n = tiny_f32.sample_begin(**kw)
    for i in range(n):
        tiny_f32.sample_step(i)
    step = tiny_f32.sample_get()
    tiny_f32.sample_end()
    np.testing.assert_array_equal(fused, step)


def test_eta_noise_and_scales_against_oracle(tiny_f32):
    """eta > 0 with caller-supplied noise, non-unit control scales: HIP path vs the oracle."""
    cfg = W.TINY
    B, h, w, S = 1, 8, 8, 4
    inp = W.synth_inputs(cfg, B, h, w, seed=7)
    rng = np.random.default_rng(3)
    noise = rng.standard_normal((S, B, 4, h, w)).astype(np.float32)
    scales = [0.825 ** (12 - i) for i in range(13)]
    sd = W.synth_state_dict(cfg)
    lay = O.make_layouts(cfg, W)
    cond = dict(c_crossattn=inp["ctx_cond"], example_pair=inp["pair"], query=inp["query"])
    unc = dict(c_crossattn=inp["ctx_uncond"], example_pair=inp["pair"], query=inp["query"])
    ref, x_inter, _ = O.ddim_sampling(sd, cfg, lay, S, inp["x_T"], cond, unc, 9.0, eta=0.7, control_scales=scales,
                                      noises=noise)
    got = tiny_f32.ddim_sample(x_T=inp["x_T"], ctx_cond=inp["ctx_cond"], ctx_uncond=inp["ctx_uncond"], pair=inp["pair"],
                               query=inp["query"], steps=S, cfg_scale=9.0, eta=0.7, control_scales=scales, noise=noise)
    assert relerr(got, ref) < 2e-4


def test_errors_are_reported_not_thrown(tiny_f32):
    inp = W.synth_inputs(W.TINY, 1, 8, 8)
    with pytest.raises(E.PdError, match="multiple of 8"):
        bad = W.synth_inputs(W.TINY, 1, 4, 4)
        tiny_f32.ddim_sample(x_T=bad["x_T"], ctx_cond=bad["ctx_cond"], ctx_uncond=bad["ctx_uncond"], pair=bad["pair"],
                             query=bad["query"], steps=2, cfg_scale=2.0)
    with pytest.raises(E.PdError, match="out of range"):
        tiny_f32.ddim_sample(x_T=inp["x_T"], ctx_cond=inp["ctx_cond"], ctx_uncond=inp["ctx_uncond"], pair=inp["pair"],
                             query=inp["query"], steps=3, cfg_scale=2.0)   # 1000//3 -> index 1000 (reference IndexError)
    with pytest.raises(E.PdError, match="needs the noise draws"):   # the reference always draws noise when eta > 0
        tiny_f32.ddim_sample(x_T=inp["x_T"], ctx_cond=inp["ctx_cond"], ctx_uncond=inp["ctx_uncond"], pair=inp["pair"],
                             query=inp["query"], steps=4, cfg_scale=2.0, eta=0.5)
    e = E.Engine(W.TINY, precision="f32")
    with pytest.raises(E.PdError, match="not loaded"):
        e.ddim_sample(x_T=inp["x_T"], ctx_cond=inp["ctx_cond"], ctx_uncond=inp["ctx_uncond"], pair=inp["pair"],
                      query=inp["query"], steps=2, cfg_scale=2.0)
    with pytest.raises(E.PdError, match="unknown tensor"):
        e.load_tensor("model.diffusion_model.nope", np.zeros(3, np.float32))
    e.close()


NORTH_STAR = 1e-3   # BASELINE.json: per-step latents within 1e-3 (relative) of the CPU reference


def _config1_errors(golden_dir, prec, wround=None):
    """BASELINE config #1 (SD1.5 + Prompt-Diffusion ControlNet, 256x256 = latent 32x32, 5 DDIM steps, bs 1, CFG 7.5) through the
    engine; returns (eps error of one apply_model, control-tensor errors, per-step latent errors) against the trajectory the
    reference itself produced on CPU.  wround="f16": weights pre-rounded to fp16 on the host (the fp32 engine then isolates the
    weight-rounding share of the 2-byte modes' error)."""
    g = np.load(os.path.join(golden_dir, "net_sd15_b1_32x32_s5.npz"))
    cfg = W.SD15
    e = E.Engine(cfg, precision=prec)
    for n, a in W.iter_synth(cfg):
        if wround == "f16" and a.ndim >= 2:
            a = a.astype(np.float16).astype(np.float32)
        e.load_tensor(n, a)
    inp, x_in, t_in, ctx, pair, qry = _cfg_inputs(cfg, g)
    eps, control = e.eps(x_in, t_in, ctx, pair, qry, return_control=True)
    cerr = []
    for i, c in enumerate(control):
        assert tuple(c.shape) == tuple(g[f"control_{i}_shape"])
        sub = c if f"control_{i}" in g else c.reshape(-1)[::int(g[f"control_{i}_stride"])][:4096]
        want = g[f"control_{i}"] if f"control_{i}" in g else g[f"control_{i}_sub"]
        cerr.append(relerr(sub, want))
    S = int(g["S"])
    out, inter = e.ddim_sample(x_T=inp["x_T"], ctx_cond=inp["ctx_cond"], ctx_uncond=inp["ctx_uncond"], pair=inp["pair"],
                               query=inp["query"], steps=S, cfg_scale=float(g["cfg_scale"]), return_intermediates=True)
    e.close()
    assert np.isfinite(out).all()
    errs = [relerr(inter[i], g["x_inter"][i]) for i in range(S + 1)]
    print(f"sd15 config #1 {prec}{' + f16-rounded weights' if wround else ''} per-step latent relerr:", ["%.2e" % v for v in errs])
    return relerr(eps, g["eps"]), cerr, errs


@pytest.mark.parametrize("prec", ["f32", "f16x2"])
def test_sd15_config1(golden_dir, prec):
    """BASELINE config #1 at the north star's tolerance, in the modes that carry fp32-class operands (f16x2 is the conforming
    mode of the 2-byte engine family: fp32 storage, every MFMA operand split into fp16 hi + lo)."""
    eps_err, cerr, errs = _config1_errors(golden_dir, prec)
    assert max(cerr) < 2e-4 and eps_err < 2e-4
    assert max(errs) < NORTH_STAR
    assert max(errs) < 2e-4     # in fact fp32-class


def test_sd15_config1_weight_rounding_floor(golden_dir):
    """Why no 2-byte mode can meet 1e-3 on THIS schedule: the fp32 engine fed weights rounded to fp16 -- every activation, every
    accumulation and the whole residual stream exact -- is already at 8-9e-4 per step (CFG 7.5 multiplies the part of the
    eps error that differs between the two passes; one 5-step DDIM step moves the latent by ~0.9 eps, against ~0.11 on the
    50-step schedule the metric is quoted on).  Any rounding of an activation comes on top."""
    _, _, errs = _config1_errors(golden_dir, "f32", wround="f16")
    assert 5e-4 < max(errs) < NORTH_STAR


@pytest.mark.parametrize("prec,tol_step,tol_eps", [("f16", 5e-3, 2.5e-3), ("bf16", 5e-2, 3e-2)])
def test_sd15_config1_two_byte_modes_measured_bounds(golden_dir, prec, tol_step, tol_eps):
    """The 2-byte modes on config #1, held to their measured bounds (fp16: weight rounding 8.5e-4 + activation rounding
    3.6e-3 in quadrature; bf16: 8x both).  These are NOT the north-star tolerance: see the strict-xfail test below."""
    eps_err, cerr, errs = _config1_errors(golden_dir, prec)
    assert max(cerr) < tol_eps and eps_err < tol_eps
    assert max(errs) < tol_step


@pytest.mark.xfail(strict=True, reason="fp16 storage cannot meet 1e-3 per step on the 5-step schedule with CFG 7.5: weight rounding alone "
                                       "is 8.5e-4 (test_sd15_config1_weight_rounding_floor), activation rounding adds 3.6e-3; the "
                                       "conforming mode is f16x2 (test_sd15_config1). On the metric's own 50-step schedule f16 is "
                                       "inside 1e-3 (test_sd15_headline_schedule_per_step).")
def test_sd15_config1_f16_at_north_star(golden_dir):
    _, _, errs = _config1_errors(golden_dir, "f16")
    assert max(errs) < NORTH_STAR


def _house_inputs(g, cfg):
    """BASELINE config #1 as written: the README's house_line -> house support pair and the new_01 query (pixel arrays in the fixture),
    (L) convention: [-1, 1], pair = [condition map | rgb image]; x_T and the contexts from the seeded recipe."""
    to_m11 = lambda u8: (u8.astype(np.float32) / 127.5 - 1.0).transpose(2, 0, 1)[None]
    inp = W.synth_inputs(cfg, 1, int(g["h"]), int(g["w"]))
    inp["pair"] = np.concatenate([to_m11(g["image_a_u8"]), to_m11(g["image_b_u8"])], axis=1)
    inp["query"] = to_m11(g["query_u8"])
    return inp


@pytest.mark.parametrize("prec,tol_eps,tol_step", [("f32", 2e-4, 2e-4), ("f16x2", 2e-4, 2e-4), ("f16", 2.5e-3, 5e-3)])
def test_sd15_config1_real_images(golden_dir, prec, tol_eps, tol_step):
    """BASELINE config #1 on the reference's own example images (images_to_try/house_line.png inverted, house.png, new_01.png
    inverted: README.md:37-40) at 256 x 256, 5 DDIM steps, bs 1, CFG 7.5, against the trajectory the reference produced on them on CPU
    (make_golden.py --only sd15_house).  Line drawings: large flat / saturated regions (most hint pixels are exactly -1 or +1), which the
    U(-1, 1) images of the other fixtures never contain.  f32 / f16x2: the north star's 1e-3 (in fact 2e-4); f16: its measured bound."""
    path = os.path.join(golden_dir, "net_sd15_b1_32x32_s5_house.npz")
    if not os.path.exists(path):
        pytest.skip("real-image fixture not generated")
    g = np.load(path)
    cfg = W.SD15
    a = g["image_a_u8"]
    assert a.shape == (256, 256, 3) and ((a == 0) | (a == 255)).mean() > 0.5   # a line drawing: mostly saturated pixels
    e = _engine(cfg, prec)
    inp = _house_inputs(g, cfg)
    x_in = np.concatenate([inp["x_T"]] * 2)
    t_in = np.full((2,), int(g["first_step"]), dtype=np.int64)
    ctx = np.concatenate([inp["ctx_uncond"], inp["ctx_cond"]])
    eps = e.eps(x_in, t_in, ctx, np.concatenate([inp["pair"]] * 2), np.concatenate([inp["query"]] * 2))
    assert relerr(eps, g["eps"]) < tol_eps, relerr(eps, g["eps"])
    S = int(g["S"])
    out, inter = e.ddim_sample(x_T=inp["x_T"], ctx_cond=inp["ctx_cond"], ctx_uncond=inp["ctx_uncond"], pair=inp["pair"],
                               query=inp["query"], steps=S, cfg_scale=float(g["cfg_scale"]), return_intermediates=True)
    e.close()
    errs = [relerr(inter[i], g["x_inter"][i]) for i in range(S + 1)]
    print(f"sd15 config #1 (README images) {prec} per-step latent relerr:", ["%.2e" % v for v in errs])
    assert np.isfinite(out).all() and max(errs) < tol_step
    if prec != "f16":
        assert max(errs) < NORTH_STAR


# one denoising step from the reference's own latent: what "per denoising step" means for the north-star bound
ONE_STEP_TOL = {"f32": 2e-4, "f16x2": 2e-4, "f16": 1e-3, "bf16": 1e-2}
TRAJ_TOL = {"f32": 1e-3, "f16x2": 1e-3, "f16": 1.5e-2, "bf16": 1.5e-1}   # accumulated over all 50 steps (chaotic growth included)


@pytest.mark.parametrize("prec", ["f32", "f16x2", "f16", "bf16"])
@pytest.mark.parametrize("tag", ["sd15_b1_32x32_s50", "sd15_b1_64x64_s50"])
def test_sd15_headline_schedule_per_step(golden_dir, tag, prec):
    """The headline schedule (50 DDIM steps, CFG 7.5) against trajectories the reference sampler produced on CPU: 256x256
    with every latent, 512x512 (BASELINE config #2's shape) with a subset.  For each kept pair of consecutive latents the
    engine is handed the reference's x_i and must reproduce x_{i+1}: the per-denoising-step error of the north star.  The
    256x256 case also runs the whole loop from x_T and bounds the accumulated deviation."""
    path = os.path.join(golden_dir, f"net_{tag}.npz")
    if not os.path.exists(path):
        pytest.skip(f"{tag} fixture not generated")
    if prec in ("f32", "f16x2") and "64x64" in tag:
        pytest.skip("covered at 32x32 (fp32-storage modes take minutes per 512x512 trajectory)")
    g = np.load(path)
    cfg = W.SD15
    B, h, w, S = int(g["B"]), int(g["h"]), int(g["w"]), int(g["S"])
    keep = [int(k) for k in g["keep"]]
    ref = {k: g["x_inter"][j] for j, k in enumerate(keep)}
    e = _engine(cfg, prec)
    inp = W.synth_inputs(cfg, B, h, w)
    kw = dict(x_T=inp["x_T"], ctx_cond=inp["ctx_cond"], ctx_uncond=inp["ctx_uncond"], pair=inp["pair"], query=inp["query"],
              steps=S, cfg_scale=float(g["cfg_scale"]))
    n = e.sample_begin(**kw)
    assert n == S
    pairs = [k for k in keep if k + 1 in ref]
    if prec in ("f32", "f16x2"):
        pairs = pairs[:3] + pairs[-2:]
    errs = []
    for i in pairs:
        e.sample_set_latents(ref[i])
        e.sample_step(i)
        errs.append(relerr(e.sample_get(), ref[i + 1]))
    e.sample_end()
    print(f"{tag} {prec}: one-step relerr max {max(errs):.2e} (first {errs[0]:.2e}, last {errs[-1]:.2e}) over {len(pairs)} steps")
    assert max(errs) < ONE_STEP_TOL[prec]
    if "32x32" in tag:
        out, inter = e.ddim_sample(return_intermediates=True, **kw)
        acc = [relerr(inter[k], ref[k]) for k in keep]
        print(f"{tag} {prec}: accumulated relerr after 10/25/50 steps {acc[10]:.2e} {acc[25]:.2e} {acc[50]:.2e}")
        assert np.isfinite(out).all() and max(acc) < TRAJ_TOL[prec]
        assert relerr(out, g["samples"]) < TRAJ_TOL[prec]
    e.close()


def test_sd15_headline_schedule_f16_margin_under_another_summation_order(golden_dir):
    """How thin the fp16 mode's margin on the metric's schedule is: letting the patch conv split K on more shapes
    (patch_split_min = 1, patch_split_tiles = 32) changes nothing but the order of fp32 partial sums -- 1e-6 in the fp32 mode -- yet moves
    single outputs by one fp16 ulp, and the first step's error against the reference's latent from 7.4e-4 to 1.02e-3.  The shipped order is
    inside the north star's 1e-3 on every step (test_sd15_headline_schedule_per_step); this order is held to 1.1e-3 at the first step
    and to 1e-3 from the second on -- the honest reading of the fp16 mode is 0.7-1.0e-3 at step 0 and <= 7.5e-4 afterwards, f16x2 (5e-6)
    being the conforming mode."""
    path = os.path.join(golden_dir, "net_sd15_b1_32x32_s50.npz")
    if not os.path.exists(path):
        pytest.skip("fixture not generated")
    g = np.load(path)
    cfg = W.SD15
    keep = [int(k) for k in g["keep"]]
    ref = {k: g["x_inter"][j] for j, k in enumerate(keep)}
    inp = W.synth_inputs(cfg, int(g["B"]), int(g["h"]), int(g["w"]))
    kw = dict(x_T=inp["x_T"], ctx_cond=inp["ctx_cond"], ctx_uncond=inp["ctx_uncond"], pair=inp["pair"], query=inp["query"],
              steps=int(g["S"]), cfg_scale=float(g["cfg_scale"]))
    e = _engine(cfg, "f16")
    try:
        e.set_option("patch_split_min", 1)
        e.set_option("patch_split_tiles", 32)
        e.sample_begin(**kw)
        errs = []
        for i in range(6):
            e.sample_set_latents(ref[i])
            e.sample_step(i)
            errs.append(relerr(e.sample_get(), ref[i + 1]))
        e.sample_end()
    finally:
        e.close()
    print("f16, other split order: one-step relerr of steps 0-5", ["%.2e" % v for v in errs])
    assert errs[0] < 1.1e-3 and max(errs[1:]) < NORTH_STAR


@pytest.mark.parametrize("prec", ["f16", "f32"])
def test_split_k_slabs_into_groupnorm_is_bit_identical(prec):
    """Option 'slab_gn' (default on): where a ResBlock's conv1 runs split-K and norm2 is the single-kernel GroupNorm (the
    16x16 / 8x8 levels), that kernel sums the fp32 slabs itself in splitk_finalize_kernel's order and rounding, so the finalize
    pass and the tensor between them disappear.  SD1.5 at 256x256 (latent 32x32: levels 32 / 16 / 8 / 4), batch 2, two guided
    steps: the latents must equal the unfused path's bit for bit, and the fused path must really be taken (fewer launches)."""
    cfg = W.SD15
    e = E.Engine(cfg, precision=prec)
    e.init_random_weights(7)
    inp = W.synth_inputs(cfg, 2, 32, 32, seed=11)
    kw = dict(x_T=inp["x_T"], ctx_cond=inp["ctx_cond"], ctx_uncond=inp["ctx_uncond"], pair=inp["pair"], query=inp["query"], steps=2, cfg_scale=7.5)
    try:
        e.set_option("slab_gn", 1)
        fused = e.ddim_sample(**kw)
        e.set_option("slab_gn", 0)
        plain = e.ddim_sample(**kw)
        n_fused, n_plain = e.stat("gn_from_slabs"), None
    finally:
        e.set_option("slab_gn", 1)
        e.close()
    assert np.isfinite(fused).all()
    assert n_fused > 0, "no ResBlock took the fused path at these sizes"
    np.testing.assert_array_equal(fused, plain)


def test_graph_replay_is_bit_identical(tiny_f32):
    """Option 'graph': the step loop of pd_ddim_sample captured in a hipGraph (both streams, fork/join events) and
    replayed on later calls with the same arguments.  Capture, replay, replay-with-new-inputs and a changed argument
    (new capture) must all equal the eager loop bit for bit."""
    e = tiny_f32
    inp = W.synth_inputs(W.TINY, 2, 16, 16, seed=3)
    inp2 = W.synth_inputs(W.TINY, 2, 16, 16, seed=4)
    kw = lambda i, s=5.0: dict(x_T=i["x_T"], ctx_cond=i["ctx_cond"], ctx_uncond=i["ctx_uncond"], pair=i["pair"], query=i["query"],
                               steps=4, cfg_scale=s)
    eager1, eager2, eager3 = e.ddim_sample(**kw(inp)), e.ddim_sample(**kw(inp2)), e.ddim_sample(**kw(inp, 3.0))
    try:
        e.set_option("graph", 1)
        np.testing.assert_array_equal(e.ddim_sample(**kw(inp)), eager1)     # capture + launch
        np.testing.assert_array_equal(e.ddim_sample(**kw(inp)), eager1)     # replay
        np.testing.assert_array_equal(e.ddim_sample(**kw(inp2)), eager2)    # replay, other inputs (staged by begin())
        np.testing.assert_array_equal(e.ddim_sample(**kw(inp, 3.0)), eager3)   # other guidance scale: new graph
        np.testing.assert_array_equal(e.ddim_sample(**kw(inp)), eager1)     # back to the first graph
        got, inter = e.ddim_sample(return_intermediates=True, **kw(inp))
        np.testing.assert_array_equal(got, eager1)
        np.testing.assert_array_equal(inter[-1], eager1)
    finally:
        e.set_option("graph", 0)
