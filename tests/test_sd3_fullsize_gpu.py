"""SD3 / MMDiT path at SD3-medium's real sizes (width 1536, 24 + 6 blocks, 16 latent channels, 333 context tokens), where
the NumPy oracle no longer finishes in seconds: size-independent properties of the path and the benchmarked f16 mode against
the engine's own fp32-class mode (f16x2) on identical random weights.  The reduced-size oracle parity is test_sd3_gpu.py."""
import numpy as np
import pytest

from prompt_diffusion_amd import sd3

pytestmark = pytest.mark.gpu

CFG = sd3.SD3Config(pos_embed_max_size=96)      # SD3-medium with a 96 x 96 table (latents up to 192 x 192)
H, S = 64, 333                                  # 512 x 512 pixels, 77 CLIP + 256 T5 tokens


def relerr(a, b):
    return float(np.abs(np.asarray(a) - b).max() / (np.abs(b).max() + 1e-30))


def inputs(B, seed=0):
    import torch
    g = torch.Generator(device="cuda").manual_seed(seed)
    f = lambda *s: torch.randn(*s, device="cuda", generator=g)
    return dict(x=f(B, 16, H, H), ctx=f(B, S, CFG.joint_dim), pooled=f(B, CFG.pooled_dim), cond=f(B, 16, H, H), pair=f(B, 16, H, H),
                nctx=f(B, S, CFG.joint_dim), npooled=f(B, CFG.pooled_dim))


@pytest.fixture(scope="module")
def eng():
    e = sd3.SD3Engine(CFG, precision="f16")
    e.init_random_weights(11)
    yield e
    e.close()


def test_f16_against_the_fp32_class_mode(eng):
    """One evaluation (ControlNet + transformer, batch 2) in the benchmarked f16 mode vs f16x2 (fp32 storage, split-fp16
    operands: fp32-class results, test_sd3_gpu.py) on the same weights: 30 blocks of fp16 operand rounding."""
    i = inputs(2, seed=1)
    t = np.array([800.0, 300.0], np.float32)
    v16 = eng.forward(i["x"], t, i["ctx"], i["pooled"], i["cond"], i["pair"]).cpu().numpy()
    ref = sd3.SD3Engine(CFG, precision="f16x2")
    ref.init_random_weights(11)
    v32 = ref.forward(i["x"], t, i["ctx"], i["pooled"], i["cond"], i["pair"]).cpu().numpy()
    c16 = [c.cpu().numpy() for c in eng.controlnet(i["x"], t, i["ctx"], i["pooled"], i["cond"], i["pair"])]
    c32 = [c.cpu().numpy() for c in ref.controlnet(i["x"], t, i["ctx"], i["pooled"], i["cond"], i["pair"])]
    ref.close()
    assert np.isfinite(v16).all() and np.isfinite(v32).all()
    e_v, e_c = relerr(v16, v32), max(relerr(a, b) for a, b in zip(c16, c32))
    print("SD3-medium f16 vs f16x2: velocity %.2e, control residuals %.2e" % (e_v, e_c))
    assert e_v < 5e-3 and e_c < 5e-3      # measured 1.5e-3 / 8.5e-4


def test_properties_at_full_size(eng):
    i = inputs(1, seed=2)
    t = np.array([650.0], np.float32)
    plain = eng.forward(i["x"], t, i["ctx"], i["pooled"]).cpu().numpy()
    steered = eng.forward(i["x"], t, i["ctx"], i["pooled"], i["cond"], i["pair"]).cpu().numpy()
    assert relerr(steered, plain) > 1e-2                                     # random (non-zero) controlnet_blocks steer
    # conditioning_scale 0 == no ControlNet, and the residuals are linear in the scale
    off = eng.forward(i["x"], t, i["ctx"], i["pooled"], i["cond"], i["pair"], conditioning_scale=0.0).cpu().numpy()
    assert relerr(off, plain) < 1e-6
    c1 = eng.controlnet(i["x"], t, i["ctx"], i["pooled"], i["cond"], i["pair"], 1.0)
    c2 = eng.controlnet(i["x"], t, i["ctx"], i["pooled"], i["cond"], i["pair"], 0.25)
    for a, b in zip(c1, c2):
        assert relerr(b.cpu().numpy(), 0.25 * a.cpu().numpy()) < 1e-6
    # batch independence: sample 0 of a batch-2 call == the batch-1 call
    j = inputs(2, seed=3)
    both = eng.forward(j["x"], np.array([650.0, 120.0], np.float32), j["ctx"], j["pooled"], j["cond"], j["pair"]).cpu().numpy()
    one = eng.forward(j["x"][:1], t, j["ctx"][:1], j["pooled"][:1], j["cond"][:1], j["pair"][:1]).cpu().numpy()
    assert relerr(both[:1], one) < 5e-3      # other tile shapes / split-K at another M: fp32 summation order, then fp16 rounding


def test_sampling_loop_identities(eng):
    """pd_sd3_sample at full size: guidance with identical negative / positive embeddings cancels (v_n + g (v_p - v_n) = v),
    a single Euler step over [1, 0] is x - v(x, t = 1000), and the loop is deterministic."""
    i = inputs(1, seed=4)
    kw = dict(control_latents=i["cond"], pair_latents=i["pair"], num_inference_steps=3)
    a = eng.sample(i["x"], i["ctx"], i["pooled"], i["ctx"], i["pooled"], guidance_scale=6.0, **kw).cpu().numpy()
    b = eng.sample(i["x"], i["ctx"], i["pooled"], guidance_scale=1.0, **kw).cpu().numpy()
    print('guidance cancellation: %.2e' % relerr(a, b))
    assert np.isfinite(a).all() and relerr(a, b) < 1e-2        # batch 2 vs 1: other tile shapes, then (1 - g) v + g v in fp32
    assert np.array_equal(a, eng.sample(i["x"], i["ctx"], i["pooled"], i["ctx"], i["pooled"], guidance_scale=6.0, **kw).cpu().numpy())
    v = eng.forward(i["x"], np.array([1000.0], np.float32), i["ctx"], i["pooled"], i["cond"], i["pair"]).cpu().numpy()
    one = eng.sample(i["x"], i["ctx"], i["pooled"], control_latents=i["cond"], pair_latents=i["pair"], num_inference_steps=1,
                     guidance_scale=1.0).cpu().numpy()
    assert relerr(one, i["x"].cpu().numpy() - v) < 1e-6
    # guidance on with different negative embeddings does change the result
    c = eng.sample(i["x"], i["ctx"], i["pooled"], i["nctx"], i["npooled"], guidance_scale=6.0, **kw).cpu().numpy()
    assert relerr(c, b) > 1e-2


def test_fp8_option_at_the_benchmarked_size():
    """1024 x 1024 (4096 image + 333 context tokens, the size tools/sd3_bench.py and bench.py's SD3 leg time), batch 2 (config
    #5's per-GPU share): one evaluation
    with e4m3 operands (levels 1 and 2) against the plain f16 mode on the same random weights -- what fp8 operands cost one
    velocity evaluation at full depth -- plus finiteness and determinism of a guided 2-step sampling."""
    import torch
    cfg = sd3.SD3Config(pos_embed_max_size=96)
    g = torch.Generator(device="cuda").manual_seed(5)
    f = lambda *s: torch.randn(*s, device="cuda", generator=g)
    NB = 2      # BASELINE config #5: bs 16 over 8 GPUs = 2 images per GPU
    x, cond, pair = f(NB, 16, 128, 128), f(NB, 16, 128, 128), f(NB, 16, 128, 128)
    ctx, nctx, pooled, npooled = f(NB, S, cfg.joint_dim), f(NB, S, cfg.joint_dim), f(NB, cfg.pooled_dim), f(NB, cfg.pooled_dim)
    t = np.array([500.0, 120.0], np.float32)
    outs = {}
    for level in (0, 1, 2):
        e = sd3.SD3Engine(cfg, precision="f16", fp8=level)
        e.init_random_weights(11)
        outs[level] = e.forward(x, t, ctx, pooled, cond, pair).cpu().numpy()
        if level == 2:
            kw = dict(control_latents=cond, pair_latents=pair, num_inference_steps=2, guidance_scale=5.0)
            a = e.sample(x, ctx, pooled, nctx, npooled, **kw).cpu().numpy()
            b = e.sample(x, ctx, pooled, nctx, npooled, **kw).cpu().numpy()
            assert np.isfinite(a).all() and np.array_equal(a, b)      # two streams, same result every time
        e.close()
    e1, e2 = relerr(outs[1], outs[0]), relerr(outs[2], outs[0])
    print("SD3-medium at 4096 + 333 tokens, fp8 vs f16 velocity: level 1 %.2e, level 2 %.2e" % (e1, e2))
    assert np.isfinite(outs[2]).all() and e1 < 3e-2 and e2 < 3e-2      # measured 6.3e-3 / 8.8e-3
