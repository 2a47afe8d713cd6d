"""SD3 / MMDiT variant of the path against oracle/sd3_oracle.py on a reduced configuration.
PARITY UNPINNED: diffusers (which holds the block arithmetic) is absent offline, so the oracle restates the published
MMDiT under the reference's call sites (promptdiffusioncontrolnet_sd3.py:362-483, pipeline :1192-1245); these tests pin the
HIP path to that restatement -- fp32-class modes to summation-order noise, the 2-byte modes to their operand precision."""
import numpy as np
import pytest

from oracle import sd3_oracle as O
from prompt_diffusion_amd import engine as E
from prompt_diffusion_amd import sd3

pytestmark = pytest.mark.gpu

CFG = sd3.SD3_TINY
TOL = {"f32": 2e-4, "f16x2": 2e-4, "f16": 1.5e-2, "bf16": 1e-1}


def relerr(a, b):
    return float(np.abs(np.asarray(a) - b).max() / (np.abs(b).max() + 1e-30))


def inputs(B, H, W, S, seed=0, cfg=CFG):
    rng = np.random.default_rng(seed)
    f = lambda *s: rng.standard_normal(s).astype(np.float32)
    return dict(x=f(B, cfg.in_channels, H, W), ctx=f(B, S, cfg.joint_dim), pooled=f(B, cfg.pooled_dim),
                cond=f(B, cfg.in_channels, H, W), pair=f(B, cfg.in_channels, H, W),
                t=rng.uniform(20.0, 980.0, B).astype(np.float32))


@pytest.fixture(scope="module")
def sd():
    return sd3.synth_sd3_state_dict(CFG)


@pytest.fixture(scope="module", params=["f32", "f16x2", "f16", "bf16"])
def eng(request, sd):
    e = sd3.SD3Engine(CFG, precision=request.param)
    e.load_state_dict(sd)
    e.prec = request.param
    yield e
    e.close()


@pytest.mark.parametrize("B,H,W,S", [(2, 8, 12, 5), (1, 16, 16, 77), (3, 4, 6, 1)])
def test_transformer_alone(eng, sd, B, H, W, S):
    i = inputs(B, H, W, S, seed=B)
    ref = O.transformer_forward(sd, CFG, i["x"], i["t"], i["ctx"], i["pooled"], None)
    got = eng.forward(i["x"], i["t"], i["ctx"], i["pooled"])
    assert got.shape == ref.shape
    assert relerr(got, ref) < TOL[eng.prec]


@pytest.mark.parametrize("B,H,W,S", [(2, 8, 12, 5), (1, 16, 16, 77)])
def test_controlnet_residuals_and_steered_velocity(eng, sd, B, H, W, S):
    i = inputs(B, H, W, S, seed=10 + B)
    ctl = O.controlnet_forward(sd, CFG, i["x"], i["t"], i["ctx"], i["pooled"], i["cond"], i["pair"], 0.8)
    got_ctl = eng.controlnet(i["x"], i["t"], i["ctx"], i["pooled"], i["cond"], i["pair"], 0.8)   # model-level call: pooled as given
    assert len(got_ctl) == CFG.cn_layers
    for g, r in zip(got_ctl, ctl):
        assert relerr(g, r) < TOL[eng.prec]
    # the pipeline-level evaluation hands the ControlNet ZERO pooled projections (force_zeros_for_pooled_projection)
    ctl = O.controlnet_forward(sd, CFG, i["x"], i["t"], i["ctx"], np.zeros_like(i["pooled"]), i["cond"], i["pair"], 0.8)
    ref = O.transformer_forward(sd, CFG, i["x"], i["t"], i["ctx"], i["pooled"], ctl)
    got = eng.forward(i["x"], i["t"], i["ctx"], i["pooled"], i["cond"], i["pair"], 0.8)
    assert relerr(got, ref) < TOL[eng.prec]
    plain = O.transformer_forward(sd, CFG, i["x"], i["t"], i["ctx"], i["pooled"], None)
    assert relerr(plain, ref) > 10 * TOL[eng.prec] or eng.prec == "bf16"    # the residuals do steer this network


def test_non_divisible_residual_mapping():
    """5 transformer blocks, 4 ControlNet blocks: residual int(i / 1.25) after block i (0, 0, 1, 2) -- the float interval of
    SD3Transformer2DModel.forward, where a ceiling would give 0, 0, 1, 1."""
    import dataclasses
    cfg = dataclasses.replace(CFG, layers=5, cn_layers=4)
    w = sd3.synth_sd3_state_dict(cfg)
    e = sd3.SD3Engine(cfg, precision="f32")
    e.load_state_dict(w)
    i = inputs(2, 8, 8, 6, seed=81, cfg=cfg)
    ctl = O.controlnet_forward(w, cfg, i["x"], i["t"], i["ctx"], 0 * i["pooled"], i["cond"], i["pair"])
    ref = O.transformer_forward(w, cfg, i["x"], i["t"], i["ctx"], i["pooled"], ctl)
    assert relerr(e.forward(i["x"], i["t"], i["ctx"], i["pooled"], i["cond"], i["pair"]), ref) < 2e-4
    e.close()


def test_sampling_loop_with_guidance(eng, sd):
    B, H, W, S, steps = 1, 8, 8, 7, 3
    i, n = inputs(B, H, W, S, seed=21), inputs(B, H, W, S, seed=22)
    ref = O.sample(sd, CFG, i["x"], i["ctx"], n["ctx"], i["pooled"], n["pooled"], i["cond"], i["pair"], steps, 5.0, scale=0.7)
    got = eng.sample(i["x"], i["ctx"], i["pooled"], n["ctx"], n["pooled"], i["cond"], i["pair"], num_inference_steps=steps,
                     guidance_scale=5.0, controlnet_conditioning_scale=0.7)
    assert relerr(got, ref) < 3 * TOL[eng.prec]
    # guidance off: single batch, no negative embeddings
    ref1 = O.sample(sd, CFG, i["x"], i["ctx"], i["ctx"], i["pooled"], i["pooled"], i["cond"], i["pair"], steps, 1.0, scale=0.7)
    got1 = eng.sample(i["x"], i["ctx"], i["pooled"], control_latents=i["cond"], pair_latents=i["pair"], num_inference_steps=steps,
                      guidance_scale=1.0, controlnet_conditioning_scale=0.7)
    assert relerr(got1, ref1) < 3 * TOL[eng.prec]


def test_controlnet_pooled_projections_and_guidance_window(sd):
    """force_zeros_for_pooled_projection = False: the ControlNet gets controlnet_pooled_projections, or the transformer's
    (pipeline :1164-1168); control_guidance_start / end switch the ControlNet off outside their window (:1155-1162)."""
    import dataclasses
    cfg = dataclasses.replace(CFG, force_zeros_for_pooled_projection=False)
    e = sd3.SD3Engine(cfg, precision="f32")
    e.load_state_dict(sd)
    i, o = inputs(2, 8, 8, 6, seed=61), inputs(2, 8, 8, 6, seed=62)
    for cn_pooled in (None, o["pooled"]):
        ctl = O.controlnet_forward(sd, cfg, i["x"], i["t"], i["ctx"], i["pooled"] if cn_pooled is None else cn_pooled, i["cond"], i["pair"])
        ref = O.transformer_forward(sd, cfg, i["x"], i["t"], i["ctx"], i["pooled"], ctl)
        got = e.forward(i["x"], i["t"], i["ctx"], i["pooled"], i["cond"], i["pair"], controlnet_pooled_projections=cn_pooled)
        assert relerr(got, ref) < 2e-4
    e.close()
    e = sd3.SD3Engine(CFG, precision="f32")
    e.load_state_dict(sd)
    i, n = inputs(1, 8, 8, 5, seed=63), inputs(1, 8, 8, 5, seed=64)
    kw = dict(steps=4, guidance=4.0, scale=0.9)
    for lo, hi in ((0.0, 0.5), (0.25, 1.0), (0.3, 0.7)):
        ref = O.sample(sd, CFG, i["x"], i["ctx"], n["ctx"], i["pooled"], n["pooled"], i["cond"], i["pair"], guidance_start=lo,
                       guidance_end=hi, **kw)
        got = e.sample(i["x"], i["ctx"], i["pooled"], n["ctx"], n["pooled"], i["cond"], i["pair"], num_inference_steps=4,
                       guidance_scale=4.0, controlnet_conditioning_scale=0.9, control_guidance_start=lo, control_guidance_end=hi)
        assert relerr(got, ref) < 6e-4
    full = O.sample(sd, CFG, i["x"], i["ctx"], n["ctx"], i["pooled"], n["pooled"], i["cond"], i["pair"], **kw)
    assert relerr(full, ref) > 1e-3          # the window matters
    e.close()


def test_cuda_tensors_in_and_out(sd):
    import torch
    e = sd3.SD3Engine(CFG, precision="f32")
    e.load_state_dict(sd)
    i = inputs(2, 8, 8, 6, seed=31)
    ref = O.transformer_forward(sd, CFG, i["x"], i["t"], i["ctx"], i["pooled"],
                                O.controlnet_forward(sd, CFG, i["x"], i["t"], i["ctx"], 0 * i["pooled"], i["cond"], i["pair"]))
    d = {k: torch.from_numpy(v).cuda() for k, v in i.items()}
    got = e.forward(d["x"], d["t"], d["ctx"], d["pooled"], d["cond"], d["pair"])
    assert got.is_cuda and relerr(got.cpu().numpy(), ref) < 2e-4
    with pytest.raises(ValueError):
        e.forward(d["x"], i["t"], i["ctx"], d["pooled"])     # mixed memory spaces
    e.close()


def test_stream_f32_option(sd):
    """fp32 residual streams with 2-byte MFMA operands: closer to the oracle than the plain f16 mode."""
    i = inputs(2, 8, 12, 9, seed=41)
    ref = O.transformer_forward(sd, CFG, i["x"], i["t"], i["ctx"], i["pooled"],
                                O.controlnet_forward(sd, CFG, i["x"], i["t"], i["ctx"], 0 * i["pooled"], i["cond"], i["pair"]))
    errs = {}
    for sf in (False, True):
        e = sd3.SD3Engine(CFG, precision="f16", stream_f32=sf)
        e.load_state_dict(sd)
        errs[sf] = relerr(e.forward(i["x"], i["t"], i["ctx"], i["pooled"], i["cond"], i["pair"]), ref)
        e.close()
    print("f16 velocity error: plain %.2e, stream_f32 %.2e" % (errs[False], errs[True]))
    assert errs[True] < TOL["f16"] and errs[True] <= errs[False] * 1.05


def test_errors(sd):
    e = sd3.SD3Engine(CFG, precision="f32")
    i = inputs(1, 8, 8, 4)
    with pytest.raises(E.PdError, match="not loaded"):
        e.forward(i["x"], i["t"], i["ctx"], i["pooled"])
    e.load_state_dict(sd)
    with pytest.raises(E.PdError, match="bad shape"):
        e.forward(i["x"][:, :, :7], i["t"], i["ctx"], i["pooled"])                  # odd latent height
    big = inputs(1, 24, 8, 4)
    with pytest.raises(E.PdError, match="pos_embed_max_size"):
        e.forward(big["x"], big["t"], big["ctx"], big["pooled"], big["cond"], big["pair"])   # ControlNet table is 10 x 10 patches
    with pytest.raises(E.PdError, match="come together"):
        e.forward(i["x"], i["t"], i["ctx"], i["pooled"], cond=i["cond"])
    with pytest.raises(ValueError):
        e.sample(i["x"], i["ctx"], i["pooled"], guidance_scale=4.0)                  # guidance without negative embeddings
    # the UNet path of the same engine still works after pd_sd3_configure
    from prompt_diffusion_amd import weights as W
    e.base.load_state_dict(W.synth_state_dict(W.TINY))
    ui = W.synth_inputs(W.TINY, 1, 8, 8)
    out = e.base.ddim_sample(x_T=ui["x_T"], ctx_cond=ui["ctx_cond"], ctx_uncond=ui["ctx_uncond"], pair=ui["pair"], query=ui["query"],
                             steps=2, cfg_scale=7.5)
    assert np.isfinite(out).all()
    e.close()


# ------------------------------------------------------------------------------------------------ sd3_fp8 option
@pytest.mark.parametrize("M,K,N", [(300, 128, 192), (77, 200, 160), (1024, 1536, 384), (6144 + 5, 256, 1536)])
def test_fp8_linear_kernel_against_emulation(M, K, N):
    """The PREC_FP8 GEMM (block-scaled K = 128 MFMA, per-row operand scales in the epilogue) against the oracle's exact
    emulation of the same quantisation: identical e4m3 operands, so only the fp32 summation order differs.  Covers K that is
    not a multiple of 128 (zero-padded step), ragged M, and the 256 x 192 tile."""
    rng = np.random.default_rng(M + K)
    x = (rng.standard_normal((M, K)) * np.exp(rng.uniform(-3, 3, (M, 1)))).astype(np.float32)   # rows of very different scale
    w = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32) * 0.1
    e = E.Engine(__import__("prompt_diffusion_amd.weights", fromlist=["TINY"]).TINY, precision="f16")
    ref = O.linear_fp8(x, w, b)
    # (products of e4m3 values are exact; the MFMA sums 128 of them per step in fp32 -- heavy cancellation at these scales)
    assert relerr(e.op_linear_fp8(x, w, b), ref) < 1e-4
    assert relerr(e.op_linear_fp8(x, w, None, gelu_tanh=True), O.gelu_tanh(O.linear_fp8(x, w))) < 1e-4
    assert 5e-3 < relerr(ref, O.linear(x, w, b)) < 1e-1           # what e4m3 operands cost one layer
    e.close()


@pytest.mark.parametrize("prec,level", [("f16", 1), ("f16", 2), ("bf16", 2)])
def test_fp8_option_against_emulating_oracle(sd, prec, level):
    """Whole evaluation with option sd3_fp8: against the oracle that emulates the e4m3 operands of the same layers the error
    is the 2-byte mode's own (plus rounding-boundary flips of single e4m3 values); against the unquantised oracle it is what
    fp8 operands cost this (random-weight) network."""
    e = sd3.SD3Engine(CFG, precision=prec, fp8=level, stream_f32=(prec == "f16" and level == 2))   # one case on fp32 residual streams
    e.load_state_dict(sd)
    i = inputs(2, 8, 12, 9, seed=71)
    zero = np.zeros_like(i["pooled"])
    got = e.forward(i["x"], i["t"], i["ctx"], i["pooled"], i["cond"], i["pair"], 0.8)
    ctl8 = O.controlnet_forward(sd, CFG, i["x"], i["t"], i["ctx"], zero, i["cond"], i["pair"], 0.8, fp8=level)
    ref8 = O.transformer_forward(sd, CFG, i["x"], i["t"], i["ctx"], i["pooled"], ctl8, fp8=level)
    ctl = O.controlnet_forward(sd, CFG, i["x"], i["t"], i["ctx"], zero, i["cond"], i["pair"], 0.8)
    ref = O.transformer_forward(sd, CFG, i["x"], i["t"], i["ctx"], i["pooled"], ctl)
    e8, e32 = relerr(got, ref8), relerr(got, ref)
    print("%s + fp8 level %d: vs emulating oracle %.2e, vs fp32 oracle %.2e (oracle fp8 vs fp32 %.2e)" % (prec, level, e8, e32, relerr(ref8, ref)))
    assert e8 < 2 * TOL[prec] and e32 < 1.5e-1
    with pytest.raises(ValueError):
        sd3.SD3Engine(CFG, precision="f32", fp8=True)
    e.close()


@pytest.mark.parametrize("prec", ["f16x2", "f16"])
def test_mid_size_against_oracle(prec):
    """Oracle-vs-HIP beyond the toy widths: width 512 (8 heads x 64), 8 transformer + 3 ControlNet blocks, 16 latent channels,
    32 x 32 latents -> 256 image + 77 context tokens -- the largest configuration the NumPy oracle finishes in seconds.  It
    exercises the 256-row GEMM tiles, multi-tile attention over the joint 333-token sequence and the two-stream schedule that
    SD3-medium runs through, which the two-block toy config does not."""
    import dataclasses
    cfg = dataclasses.replace(sd3.SD3_MEDIUM, heads=8, head_dim=64, layers=8, cn_layers=3, joint_dim=512, pooled_dim=256,
                              pos_embed_max_size=32)
    w = sd3.synth_sd3_state_dict(cfg, seed=21)
    i = inputs(2, 32, 32, 77, seed=22, cfg=cfg)
    ctl = O.controlnet_forward(w, cfg, i["x"], i["t"], i["ctx"], np.zeros_like(i["pooled"]), i["cond"], i["pair"], 1.0)
    ref = O.transformer_forward(w, cfg, i["x"], i["t"], i["ctx"], i["pooled"], ctl)
    e = sd3.SD3Engine(cfg, precision=prec)
    e.load_state_dict(w)
    got = e.forward(i["x"], i["t"], i["ctx"], i["pooled"], i["cond"], i["pair"], 1.0)
    got_ctl = e.controlnet(i["x"], i["t"], i["ctx"], np.zeros_like(i["pooled"]), i["cond"], i["pair"], 1.0)
    e.close()
    tol = {"f16x2": 3e-4, "f16": 1.5e-2}[prec]
    for g, r in zip(got_ctl, ctl):
        assert relerr(g, r) < tol
    print("SD3 mid-size %s vs oracle: %.2e" % (prec, relerr(got, ref)))
    assert relerr(got, ref) < tol


@pytest.mark.parametrize("prec", ["f32", "f16"])
def test_sd35_style_blocks(prec):
    """qk_norm = "rms_norm" (per-head RMSNorm of queries and keys, own weights for the image and the context stream) and
    dual_attention_layers (attn2: a second, image-only attention on the 7th-9th modulation chunks) in both networks
    (promptdiffusioncontrolnet_sd3.py:104-105, :140-141), against the oracle's restatement."""
    import dataclasses
    cfg = dataclasses.replace(CFG, qk_norm="rms_norm", dual_attention_layers=(0, 1), cn_dual_attention_layers=(1,))
    w = sd3.synth_sd3_state_dict(cfg, seed=3)
    e = sd3.SD3Engine(cfg, precision=prec)
    e.load_state_dict(w)
    for B, H, Wd, S in ((2, 8, 12, 5), (1, 6, 4, 77)):     # the second one: 6 image tokens, V^T padded to 8
        i = inputs(B, H, Wd, S, seed=40 + B, cfg=cfg)
        ctl = O.controlnet_forward(w, cfg, i["x"], i["t"], i["ctx"], np.zeros_like(i["pooled"]), i["cond"], i["pair"], 0.9)
        ref = O.transformer_forward(w, cfg, i["x"], i["t"], i["ctx"], i["pooled"], ctl)
        got = e.forward(i["x"], i["t"], i["ctx"], i["pooled"], i["cond"], i["pair"], 0.9)
        assert relerr(got, ref) < TOL[prec], (B, relerr(got, ref))
        plain_cfg = dataclasses.replace(cfg, qk_norm=None)
        # the same weights without the RMSNorm give a different velocity: the norm is really applied
        ref_plain = O.transformer_forward(w, plain_cfg, i["x"], i["t"], i["ctx"], i["pooled"], None)
        ref_norm = O.transformer_forward(w, cfg, i["x"], i["t"], i["ctx"], i["pooled"], None)
        assert relerr(ref_plain, ref_norm) > 1e-2
    e.close()


@pytest.mark.parametrize("prec", ["f32", "f16"])
def test_single_block_controlnet(prec):
    """The ControlNet built with joint_attention_dim = None (promptdiffusioncontrolnet_sd3.py:147-160, :426-431): its blocks are
    SD3SingleTransformerBlocks that see the image tokens alone, it has no context_embedder, and the prompt does not reach it."""
    import dataclasses
    for cfg in (dataclasses.replace(CFG, cn_single_blocks=True), dataclasses.replace(CFG, cn_single_blocks=True, qk_norm="rms_norm")):
        w = sd3.synth_sd3_state_dict(cfg, seed=8)
        assert not any(k.startswith("controlnet.context_embedder") or "controlnet.transformer_blocks.0.ff_context" in k for k in w)
        e = sd3.SD3Engine(cfg, precision=prec)
        e.load_state_dict(w)
        for B, H, Wd, S in ((2, 8, 12, 5), (1, 6, 4, 77)):
            i = inputs(B, H, Wd, S, seed=60 + B, cfg=cfg)
            zp = np.zeros_like(i["pooled"])
            ctl = O.controlnet_forward(w, cfg, i["x"], i["t"], i["ctx"], zp, i["cond"], i["pair"], 0.8)
            got_ctl = e.controlnet(i["x"], i["t"], i["ctx"], zp, i["cond"], i["pair"], 0.8)
            for g, r in zip(got_ctl, ctl):
                assert relerr(g, r) < TOL[prec]
            # no context stream: another prompt gives the same residuals, bit for bit
            other = e.controlnet(i["x"], i["t"], i["ctx"][:, ::-1].copy() * 3.0, zp, i["cond"], i["pair"], 0.8)
            assert all(np.array_equal(a, b) for a, b in zip(got_ctl, other))
            ref = O.transformer_forward(w, cfg, i["x"], i["t"], i["ctx"], i["pooled"], ctl)
            got = e.forward(i["x"], i["t"], i["ctx"], i["pooled"], i["cond"], i["pair"], 0.8)
            assert relerr(got, ref) < TOL[prec], (B, relerr(got, ref))
        e.close()


def test_down_proj_and_strict_loading():
    """encode_support_pair's Conv2d(6, 3, 3, padding=1) on the engine (promptdiffusioncontrolnet_sd3.py:114, :189-198); strict
    loading refuses tensors of blocks the configuration does not have."""
    w = sd3.synth_sd3_state_dict(CFG, seed=5)
    e = sd3.SD3Engine(CFG, precision="f32")
    e.load_state_dict(w)
    rng = np.random.default_rng(6)
    cond, gt = rng.standard_normal((2, 3, 20, 12)).astype(np.float32), rng.standard_normal((2, 3, 20, 12)).astype(np.float32)
    got = e.encode_support_pair(cond, gt)
    ref = O.down_proj(w, np.concatenate([cond, gt], 1))
    assert got.shape == ref.shape and relerr(got, ref) < 2e-5

    class _Dist:
        def __init__(self, x): self.x = x
        def sample(self): return self.x[:, :, ::8, ::8] * 2.0
    class _Vae:
        def encode(self, x): return type("o", (), {"latent_dist": _Dist(x)})()
    assert np.array_equal(e.encode_support_pair(cond, gt, vae=_Vae()), got[:, :, ::8, ::8] * 2.0)
    bad = dict(w)
    bad["transformer.transformer_blocks.0.attn.norm_q.weight"] = np.ones(CFG.head_dim, np.float32)
    with pytest.raises(E.PdError, match="unexpected"):
        e.load_state_dict(bad)
    e.close()
