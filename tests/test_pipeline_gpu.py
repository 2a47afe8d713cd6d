"""The reference-shaped host surfaces on the GPU: DDIMSampler.sample ((L), cldm/ddim_hacked.py:55) against the
reference's own trajectory, and PromptDiffusionPipeline.__call__ ((D), pipeline_prompt_diffusion.py:890) against
an oracle replay of the (D)-specific loop logic (guess mode, controlnet_keep window, [0,1] images)."""
import os

import numpy as np
import pytest

from oracle import pd_oracle as O
from prompt_diffusion_amd import engine as E
from prompt_diffusion_amd import weights as W
from prompt_diffusion_amd.ddim import ControlLDM, DDIMSampler
from prompt_diffusion_amd.pipeline import PromptDiffusionPipeline

pytestmark = pytest.mark.gpu


def relerr(a, b):
    return float(np.abs(np.asarray(a) - b).max() / (np.abs(b).max() + 1e-30))


@pytest.fixture(scope="module")
def eng():
    e = E.Engine(W.TINY, precision="f32")
    e.load_state_dict(W.synth_state_dict(W.TINY))
    yield e
    e.close()


def test_ddim_sampler_surface_matches_reference_run(golden_dir, eng):
    g = np.load(os.path.join(golden_dir, "net_tiny_b2_16x16_s5.npz"))
    B, h, w, S = int(g["B"]), int(g["h"]), int(g["w"]), int(g["S"])
    inp = W.synth_inputs(W.TINY, B, h, w)
    model = ControlLDM(eng)
    sampler = DDIMSampler(model)
    cond = {"c_crossattn": [inp["ctx_cond"]], "example_pair": [inp["pair"]], "query": [inp["query"]]}
    uc = {"c_crossattn": [inp["ctx_uncond"]], "example_pair": [inp["pair"]], "query": [inp["query"]]}
    calls, preds = [], []
    samples, inter = sampler.sample(S, B, (4, h, w), cond, eta=0.0, x_T=inp["x_T"], unconditional_guidance_scale=float(g["cfg_scale"]),
                                    unconditional_conditioning=uc, log_every_t=1, verbose=False,
                                    callback=calls.append, img_callback=lambda p, i: preds.append((i, p)))
    assert calls == list(range(S)) and [i for i, _ in preds] == list(range(S))
    assert len(inter["x_inter"]) == S + 1 and len(inter["pred_x0"]) == S + 1
    for i in range(S + 1):
        assert relerr(inter["x_inter"][i], g["x_inter"][i]) < 2e-4
        assert relerr(inter["pred_x0"][i], g["pred_x0"][i]) < 2e-4
    assert relerr(samples, g["samples"]) < 2e-4
    # default log_every_t=100: x_T, the first step (index == total-1) and the last (index 0), ddim_hacked.py:174
    _, inter2 = sampler.sample(S, B, (4, h, w), cond, eta=0.0, x_T=inp["x_T"], unconditional_guidance_scale=float(g["cfg_scale"]),
                               unconditional_conditioning=uc, verbose=False)
    assert len(inter2["x_inter"]) == 3
    # apply_model boundary
    t = np.full((B,), int(g["first_step"]), np.int64)
    eps = model.apply_model(inp["x_T"], t, cond)
    assert relerr(eps, g["eps"][B:]) < 2e-4


def test_ddim_sampler_inpainting_mask_matches_reference_run(golden_dir, eng):
    """sample(mask=, x0=): the blend of ddim_hacked.py:154-157 through the per-step export, against the reference's
    own run (q_sample noise draws replayed from the fixture)."""
    g = np.load(os.path.join(golden_dir, "net_tiny_mask_b2_16x16_s5.npz"))
    B, h, w, S = int(g["B"]), int(g["h"]), int(g["w"]), int(g["S"])
    inp = W.synth_inputs(W.TINY, B, h, w, seed=int(g["seed"]))
    model = ControlLDM(eng)
    draws = iter(g["q_noise"])
    model.q_sample = lambda x, t, noise=None: ControlLDM.q_sample(model, x, t, next(draws))
    sampler = DDIMSampler(model)
    cond = {"c_crossattn": [inp["ctx_cond"]], "example_pair": [inp["pair"]], "query": [inp["query"]]}
    uc = {"c_crossattn": [inp["ctx_uncond"]], "example_pair": [inp["pair"]], "query": [inp["query"]]}
    samples, inter = sampler.sample(S, B, (4, h, w), cond, eta=0.0, x_T=inp["x_T"], mask=g["mask"], x0=g["x0"],
                                    unconditional_guidance_scale=float(g["cfg_scale"]), unconditional_conditioning=uc,
                                    log_every_t=1, verbose=False)
    for i in range(S + 1):
        assert relerr(inter["x_inter"][i], g["x_inter"][i]) < 3e-4, i
    assert relerr(samples, g["samples"]) < 3e-4
    with pytest.raises(AssertionError):
        sampler.sample(S, B, (4, h, w), cond, eta=0.0, x_T=inp["x_T"], mask=g["mask"], unconditional_conditioning=uc)


def _oracle_pipeline(cfg, sd, lay, x_T, pe, ne, pair, query, S, gs, scale, guess, g_start, g_end, timesteps=None):
    """Replay of pipeline_prompt_diffusion.py:1196-1273 with the oracle's networks and a DDIM step."""
    sched = O.make_schedule(S, timesteps=None if timesteps is None else sorted(timesteps))
    n = len(sched["ddim_timesteps"])
    keep = [1.0 - float(i / n < g_start or (i + 1) / n > g_end) for i in range(n)]
    B = x_T.shape[0]
    x = x_T
    for i, step in enumerate(np.flip(sched["ddim_timesteps"])):
        index = n - i - 1
        base = np.logspace(-1, 0, 13).astype(np.float32) if guess else np.ones(13, np.float32)
        scales = base * np.float32(scale) * np.float32(keep[i])
        x_in = np.concatenate([x, x]); t_in = np.full((2 * B,), int(step), np.int64)
        ctx = np.concatenate([ne, pe]); pr = np.concatenate([pair, pair]); qr = np.concatenate([query, query])
        if guess:   # ControlNet on the conditional half only, zeros for the unconditional half (:1220-1224,:1248-1253)
            ctl = O.controlnet_forward(sd, cfg, lay, x, t_in[:B], pair, query, pe)
            ctl = [np.concatenate([np.zeros_like(c), c]) * s for c, s in zip(ctl, scales)]
        else:
            ctl = O.controlnet_forward(sd, cfg, lay, x_in, t_in, pr, qr, ctx)
            ctl = [c * s for c, s in zip(ctl, scales)]
        eps = O.controlled_unet_forward(sd, cfg, lay, x_in, t_in, ctx, ctl)
        e = eps[:B] + np.float32(gs) * (eps[B:] - eps[:B])
        a_t, a_prev = sched["ddim_alphas"][index], sched["ddim_alphas_prev"][index]
        pred = (x - sched["ddim_sqrt_one_minus_alphas"][index] * e) / np.sqrt(a_t)
        x = (np.sqrt(a_prev) * pred + np.sqrt(np.float32(1.0) - a_prev) * e).astype(np.float32)
    return x


@pytest.mark.parametrize("guess,g_end", [(False, 1.0), (True, 0.6)])
def test_pipeline_call_against_oracle(eng, guess, g_end):
    cfg = W.TINY
    B, hw, S = 2, 64, 4
    inp = W.synth_inputs(cfg, B, hw // 8, hw // 8, seed=11, unit_range=True)
    sd = W.synth_state_dict(cfg)
    lay = O.make_layouts(cfg, W)
    pipe = PromptDiffusionPipeline(eng)
    a, b = inp["pair"][:, :3], inp["pair"][:, 3:]
    seen = []
    out = pipe(prompt_embeds=inp["ctx_cond"], negative_prompt_embeds=inp["ctx_uncond"],
               image=inp["query"].transpose(0, 2, 3, 1), image_pair=[a.transpose(0, 2, 3, 1), b.transpose(0, 2, 3, 1)],
               num_inference_steps=S, guidance_scale=5.0, latents=inp["x_T"], output_type="latent",
               controlnet_conditioning_scale=0.8, guess_mode=guess, control_guidance_end=g_end,
               callback_on_step_end=(lambda p, i, t, kw: seen.append((i, int(t))) or {}) if guess else None)
    ref = _oracle_pipeline(cfg, sd, lay, inp["x_T"], inp["ctx_cond"], inp["ctx_uncond"], inp["pair"], inp["query"], S, 5.0, 0.8,
                           guess, 0.0, g_end)
    assert out.nsfw_content_detected is None
    assert relerr(out.images, ref) < 3e-4
    if guess:
        assert [i for i, _ in seen] == list(range(S)) and seen[0][1] == 751
    tup = pipe(prompt_embeds=inp["ctx_cond"], negative_prompt_embeds=inp["ctx_uncond"], image=inp["query"].transpose(0, 2, 3, 1),
               image_pair=[a.transpose(0, 2, 3, 1), b.transpose(0, 2, 3, 1)], num_inference_steps=S, guidance_scale=5.0,
               latents=inp["x_T"], output_type="latent", controlnet_conditioning_scale=0.8, guess_mode=guess,
               control_guidance_end=g_end, return_dict=False)
    assert isinstance(tup, tuple) and tup[1] is None
    np.testing.assert_allclose(np.asarray(tup[0]), np.asarray(out.images), rtol=0, atol=0)
    with pytest.raises(ValueError, match="first-stage weights|vae_decode"):
        pipe(prompt_embeds=inp["ctx_cond"], negative_prompt_embeds=inp["ctx_uncond"], image=inp["query"].transpose(0, 2, 3, 1),
             image_pair=[a.transpose(0, 2, 3, 1), b.transpose(0, 2, 3, 1)], num_inference_steps=S, latents=inp["x_T"])


def test_pipeline_custom_timesteps_and_leading_grid(eng):
    """`timesteps=` (retrieve_timesteps, pipeline_prompt_diffusion.py:101-142) through the engine's own DDIM loop, and the
    diffusers grid for a step count that does not divide 1000: exactly num_inference_steps entries arange(S)*(T//S)+1."""
    cfg = W.TINY
    B, hw = 1, 64
    inp = W.synth_inputs(cfg, B, hw // 8, hw // 8, seed=41, unit_range=True)
    sd = W.synth_state_dict(cfg)
    lay = O.make_layouts(cfg, W)
    pipe = PromptDiffusionPipeline(eng)
    a, b = inp["pair"][:, :3], inp["pair"][:, 3:]
    kw = dict(prompt_embeds=inp["ctx_cond"], negative_prompt_embeds=inp["ctx_uncond"], image=inp["query"].transpose(0, 2, 3, 1),
              image_pair=[a.transpose(0, 2, 3, 1), b.transpose(0, 2, 3, 1)], guidance_scale=4.0, latents=inp["x_T"],
              output_type="latent", control_guidance_end=0.7)
    ts = [961, 640, 333, 40, 2]
    seen = []
    out = pipe(timesteps=ts, callback_on_step_end=lambda p, i, t, k: seen.append(int(t)) or {}, **kw).images
    assert seen == ts
    fused = pipe(timesteps=ts, **kw).images      # no callback: the fused loop with the same grid
    np.testing.assert_array_equal(np.asarray(fused), np.asarray(out))
    ref = _oracle_pipeline(cfg, sd, lay, inp["x_T"], inp["ctx_cond"], inp["ctx_uncond"], inp["pair"], inp["query"], len(ts), 4.0, 1.0,
                           False, 0.0, 0.7, timesteps=ts)
    assert relerr(out, ref) < 3e-4
    seen.clear()
    out3 = pipe(num_inference_steps=3, callback_on_step_end=lambda p, i, t, k: seen.append(int(t)) or {}, **kw).images
    assert seen == [667, 334, 1]                 # 3 steps, not the 4 of range(0, 1000, 333)
    np.testing.assert_array_equal(np.asarray(out3), np.asarray(pipe(timesteps=[667, 334, 1], **kw).images))
    with pytest.raises(E.PdError, match="must be descending"):
        pipe(timesteps=[500, 600, 1], **kw)
    # repeated timesteps are legal (make_ddim_timesteps('quad') produces them, util.py:50)
    assert np.isfinite(np.asarray(pipe(timesteps=[667, 334, 334, 1], **kw).images)).all()
    from prompt_diffusion_amd.schedulers import UniPCMultistepScheduler
    with pytest.raises(ValueError, match="does not support custom"):
        PromptDiffusionPipeline(eng, scheduler=UniPCMultistepScheduler())(timesteps=ts, **kw)


def test_pipeline_decodes_images_with_engine_vae():
    """output_type="np": loop + first-stage decode inside the engine, against the oracle (f32 mode)."""
    cfg = W.TINY
    e = E.Engine(cfg, precision="f32")
    sd = W.synth_state_dict(cfg)
    vsd = W.synth_vae_state_dict(cfg)
    e.load_state_dict({**sd, **vsd})
    assert e.vae_weights_missing() == 0
    B, hw, S = 1, 64, 2
    inp = W.synth_inputs(cfg, B, hw // 8, hw // 8, seed=5, unit_range=True)
    lay = O.make_layouts(cfg, W)
    pipe = PromptDiffusionPipeline(e)
    a, b = inp["pair"][:, :3], inp["pair"][:, 3:]
    kw = dict(prompt_embeds=inp["ctx_cond"], negative_prompt_embeds=inp["ctx_uncond"], image=inp["query"].transpose(0, 2, 3, 1),
              image_pair=[a.transpose(0, 2, 3, 1), b.transpose(0, 2, 3, 1)], num_inference_steps=S, guidance_scale=3.0,
              latents=inp["x_T"])
    imgs = pipe(output_type="np", **kw).images
    lat = _oracle_pipeline(cfg, sd, lay, inp["x_T"], inp["ctx_cond"], inp["ctx_uncond"], inp["pair"], inp["query"], S, 3.0, 1.0,
                           False, 0.0, 1.0)
    ref = np.clip(O.vae_decode(vsd, cfg, W.vae_layout(cfg), lat) / 2 + 0.5, 0, 1).transpose(0, 2, 3, 1)
    assert imgs.shape == (B, hw, hw, 3)
    assert float(np.abs(imgs - ref).max()) < 2e-3
    pil = pipe(output_type="pil", **kw).images
    assert pil[0].size == (hw, hw)
    e.close()


def test_pipeline_with_unipc_scheduler_against_oracle_replay(eng):
    """Scheduler plug-in path (README.md:49 swaps in UniPC): the engine evaluates eps at the scheduler's own timesteps
    (pd_sample_eps_at) and the host scheduler advances the latents.  Replayed with the oracle network and the oracle's
    independent closed-form UniPC (parity with diffusers itself is unpinned, see schedulers.py); the control window
    is counted on the scheduler's grid (pipeline_prompt_diffusion.py:1196-1202)."""
    from prompt_diffusion_amd.schedulers import UniPCMultistepScheduler
    cfg = W.TINY
    B, hw, S, gs, scale, g_end = 1, 64, 5, 4.0, 0.9, 0.8
    inp = W.synth_inputs(cfg, B, hw // 8, hw // 8, seed=21, unit_range=True)
    sd = W.synth_state_dict(cfg)
    lay = O.make_layouts(cfg, W)
    sched = UniPCMultistepScheduler()
    pipe = PromptDiffusionPipeline(eng, scheduler=sched)
    a, b = inp["pair"][:, :3], inp["pair"][:, 3:]
    out = pipe(prompt_embeds=inp["ctx_cond"], negative_prompt_embeds=inp["ctx_uncond"], image=inp["query"].transpose(0, 2, 3, 1),
               image_pair=[a.transpose(0, 2, 3, 1), b.transpose(0, 2, 3, 1)], num_inference_steps=S, guidance_scale=gs,
               latents=inp["x_T"], output_type="latent", controlnet_conditioning_scale=scale, control_guidance_end=g_end).images
    ts = [int(t) for t in sched.timesteps]
    assert len(ts) == S and ts[0] == 999
    keep = {t: 1.0 - float((i + 1) / S > g_end) for i, t in enumerate(ts)}
    pe, ne, pair, query = inp["ctx_cond"], inp["ctx_uncond"], inp["pair"], inp["query"]

    def eps_fn(x, t):
        x = x.astype(np.float32)
        x_in = np.concatenate([x, x]); t_in = np.full((2 * B,), t, np.int64)
        ctx = np.concatenate([ne, pe]); pr = np.concatenate([pair, pair]); qr = np.concatenate([query, query])
        ctl = [c * np.float32(scale * keep[t]) for c in O.controlnet_forward(sd, cfg, lay, x_in, t_in, pr, qr, ctx)]
        eps = O.controlled_unet_forward(sd, cfg, lay, x_in, t_in, ctx, ctl)
        return eps[:B] + np.float32(gs) * (eps[B:] - eps[:B])

    ref = O.unipc2_sample(eps_fn, inp["x_T"], sched.alphas_cumprod, ts)
    assert relerr(out, ref) < 5e-4


def test_pipeline_from_prompts_with_engine_text_encoder():
    """String prompts -> tokenizer (caller's) -> pd_text_encode -> loop, all inside one engine (SURVEY N3 + hot path):
    must equal the same call fed with the oracle's CLIP embeddings of the same token ids."""
    cfg = W.TINY
    e = E.Engine(cfg, precision="f32")
    tsd = W.synth_text_state_dict(cfg)
    e.load_state_dict({**W.synth_state_dict(cfg), **tsd})
    vocab = {}

    def toy_tokenizer(prompts):     # BOS, one id per word, EOS padding -- stands in for CLIPTokenizer (BPE files are not available offline)
        ids = np.full((len(prompts), cfg.context_len), cfg.text_vocab - 1, np.int32)
        for b, p in enumerate(prompts):
            ids[b, 0] = cfg.text_vocab - 2
            for j, wd in enumerate(p.split()[:cfg.context_len - 2]):
                ids[b, 1 + j] = vocab.setdefault(wd, len(vocab) + 1)
        return ids
    pipe = PromptDiffusionPipeline(e, tokenizer=toy_tokenizer)
    B, hw, S = 2, 64, 4
    inp = W.synth_inputs(cfg, B, hw // 8, hw // 8, seed=17, unit_range=True)
    a, b = inp["pair"][:, :3], inp["pair"][:, 3:]
    prompts = ["a photo of a house, best quality", "a line drawing"]
    kw = dict(image=inp["query"].transpose(0, 2, 3, 1), image_pair=[a.transpose(0, 2, 3, 1), b.transpose(0, 2, 3, 1)],
              num_inference_steps=S, guidance_scale=4.0, latents=inp["x_T"], output_type="latent")
    try:
        out = pipe(prompt=prompts, negative_prompt="blurry", **kw).images
        pe = O.clip_text_forward(tsd, cfg, toy_tokenizer(prompts))
        ne = O.clip_text_forward(tsd, cfg, toy_tokenizer(["blurry"] * B))
        ref = pipe(prompt_embeds=pe, negative_prompt_embeds=ne, **kw).images
        assert relerr(out, ref) < 2e-4
        # clip_skip (pipeline_prompt_diffusion.py:398-413): the conditional prompt from hidden_states[-(k+1)] + final LN,
        # the negative prompt from the last layer
        out1 = pipe(prompt=prompts, negative_prompt="blurry", clip_skip=1, **kw).images
        pe1 = O.clip_text_forward(tsd, cfg, toy_tokenizer(prompts), clip_skip=1)
        assert relerr(pe1, pe) > 1e-2
        ref1 = pipe(prompt_embeds=pe1, negative_prompt_embeds=ne, **kw).images
        assert relerr(out1, ref1) < 2e-4 and relerr(out1, out) > 1e-3
        with pytest.raises(E.PdError, match="clip_skip"):
            pipe(prompt=prompts, clip_skip=cfg.text_layers + 1, **kw)
    finally:
        e.close()
