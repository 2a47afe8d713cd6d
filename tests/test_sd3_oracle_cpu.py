"""CPU checks of the SD3 oracle (oracle/sd3_oracle.py) and of the host logic beside it.  The oracle is "parity unpinned"
(diffusers absent offline), so what CAN be pinned here is pinned: every primitive against torch's CPU implementation of the
same operator, the architecture against the published parameter count of SD3-medium, and the semantics the reference itself
states (zero-initialised ControlNet modules leave the transformer untouched, promptdiffusioncontrolnet_sd3.py:34-37, :161-175)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import sd3_oracle as O
from prompt_diffusion_amd import sd3

CFG = sd3.SD3_TINY


def test_primitives_against_torch():
    rng = np.random.default_rng(0)
    x = rng.standard_normal((2, 4, 8, 12)).astype(np.float32)
    w = rng.standard_normal((32, 4, 2, 2)).astype(np.float32)
    b = rng.standard_normal(32).astype(np.float32)
    ref = F.conv2d(torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(b), stride=2).flatten(2).transpose(1, 2).numpy()
    assert np.allclose(O.patch_embed(x, w, b, 2), ref, atol=1e-5)
    h = rng.standard_normal((3, 7, 64)).astype(np.float32) * 3
    assert np.allclose(O.layer_norm_noaffine(h), F.layer_norm(torch.from_numpy(h), (64,), eps=1e-6).numpy(), atol=1e-5)
    assert np.allclose(O.gelu_tanh(h), F.gelu(torch.from_numpy(h), approximate="tanh").numpy(), atol=1e-6)
    assert np.allclose(O.silu(h), F.silu(torch.from_numpy(h)).numpy(), atol=1e-6)
    q, k, v = (rng.standard_normal((2, n, 128)).astype(np.float32) for n in (5, 9, 9))
    sp = lambda t: torch.from_numpy(t).reshape(2, -1, 2, 64).transpose(1, 2)
    ref = F.scaled_dot_product_attention(sp(q), sp(k), sp(v)).transpose(1, 2).reshape(2, 5, 128).numpy()
    assert np.allclose(O.attention(q, k, v, 2), ref, atol=1e-5)


def test_timestep_embedding_and_tables():
    e = O.timestep_embedding(np.array([0.0, 500.0], np.float32))
    assert e.shape == (2, 256)
    assert np.allclose(e[0, :128], 1.0) and np.allclose(e[0, 128:], 0.0)            # flip_sin_to_cos: [cos | sin]
    assert np.isclose(e[1, 0], np.cos(500.0), atol=1e-4) and np.isclose(e[1, 128], np.sin(500.0), atol=1e-4)
    assert np.isclose(e[1, 127], np.cos(500.0 * 10000 ** (-127 / 128)), atol=1e-5)  # downscale_freq_shift = 0
    t = O.sincos_pos_embed(16, 6, 3)
    assert t.shape == (36, 16) and np.array_equal(t, sd3.sincos_pos_embed(16, 6, 3))
    assert np.allclose(t[0, :4], 0) and np.allclose(t[0, 4:8], 1)                    # position (0, 0): sin 0, cos 0
    assert np.array_equal(t[1, 8:], t[0, 8:]) and not np.array_equal(t[1, :8], t[0, :8])   # moving along a row changes the column half only
    c = O.cropped_pos_embed(t, 6, 2, 4)
    assert np.array_equal(c, t.reshape(6, 6, 16)[2:4, 1:5].reshape(8, 16))


def test_flow_match_sigmas():
    for steps in (1, 3, 28):
        s = O.flow_match_sigmas(steps)
        assert np.array_equal(s, sd3.flow_match_sigmas(steps))
        assert s.shape == (steps + 1,) and s[0] == 1.0 and s[-1] == 0.0 and (np.diff(s) < 0).all()
    # shift 1: the plain linear grid between sigma_max = 1 and sigma_min = 1 / 1000
    assert np.allclose(O.flow_match_sigmas(4, shift=1.0)[:4], np.linspace(1.0, 0.001, 4))


def test_parameter_count_matches_published_sd3_medium():
    """The restated architecture under diffusers' names, at SD3-medium's published size: 24 blocks of width 1536 hold the
    model card's 2.03 B transformer parameters."""
    shapes = sd3.sd3_param_shapes(sd3.SD3_MEDIUM)
    n = sum(int(np.prod(s)) for k, s in shapes.items() if k.startswith("transformer.") and not k.endswith("pos_embed.pos_embed"))
    assert n == 2_028_328_000
    assert "transformer.transformer_blocks.23.attn.to_add_out.weight" not in shapes      # context_pre_only
    assert shapes["transformer.transformer_blocks.23.norm1_context.linear.weight"] == (2 * 1536, 1536)
    assert shapes["controlnet.transformer_blocks.5.norm1_context.linear.weight"] == (6 * 1536, 1536)
    assert shapes["transformer.pos_embed.pos_embed"] == (1, 192 * 192, 1536)


def test_zero_initialised_controlnet_is_a_no_op():
    sd = sd3.synth_sd3_state_dict(CFG)
    rng = np.random.default_rng(3)
    f = lambda *s: rng.standard_normal(s).astype(np.float32)
    x, ctx, pooled, cond, pair = f(2, 4, 8, 8), f(2, 6, CFG.joint_dim), f(2, CFG.pooled_dim), f(2, 4, 8, 8), f(2, 4, 8, 8)
    t = np.array([700.0, 100.0], np.float32)
    plain = O.transformer_forward(sd, CFG, x, t, ctx, pooled, None)
    z = dict(sd)
    for k in sd:
        if k.startswith("controlnet.controlnet_blocks."):
            z[k] = np.zeros_like(sd[k])
    ctl = O.controlnet_forward(z, CFG, x, t, ctx, pooled, cond, pair)
    assert all(np.abs(c).max() == 0 for c in ctl)
    assert np.array_equal(O.transformer_forward(z, CFG, x, t, ctx, pooled, ctl), plain)
    # and the non-zero ones steer; conditioning_scale is linear in the residuals
    c1 = O.controlnet_forward(sd, CFG, x, t, ctx, pooled, cond, pair, 1.0)
    c2 = O.controlnet_forward(sd, CFG, x, t, ctx, pooled, cond, pair, 0.5)
    assert all(np.allclose(a * 0.5, b, atol=1e-6) for a, b in zip(c1, c2))
    assert np.abs(O.transformer_forward(sd, CFG, x, t, ctx, pooled, c1) - plain).max() > 1e-2
    # residual int(i / (layers / len(residuals))) lands after block i -- one residual for all blocks == the same one repeated
    one = O.transformer_forward(sd, CFG, x, t, ctx, pooled, c1[:1])
    rep = O.transformer_forward(sd, CFG, x, t, ctx, pooled, [c1[0], c1[0], c1[0]])
    assert np.allclose(one, rep, atol=1e-6)


def test_residual_to_block_mapping_is_the_float_interval():
    """SD3Transformer2DModel.forward indexes the residuals with int(i / (len(blocks) / len(residuals))) -- a float interval,
    not the ceiling the Flux transformer takes.  5 blocks (the last context_pre_only) and 4 residuals: blocks 0..3 get
    residuals 0, 0, 1, 2 (the ceiling would give 0, 0, 1, 1), and residual 3 is never used."""
    import dataclasses
    cfg = dataclasses.replace(CFG, layers=5, cn_layers=4)
    sd = sd3.synth_sd3_state_dict(cfg)
    rng = np.random.default_rng(5)
    f = lambda *s: rng.standard_normal(s).astype(np.float32)
    x, ctx, pooled = f(1, 4, 8, 8), f(1, 3, cfg.joint_dim), f(1, cfg.pooled_dim)
    t = np.array([400.0], np.float32)
    ctl = [f(1, 16, cfg.hidden) for _ in range(4)]
    base = O.transformer_forward(sd, cfg, x, t, ctx, pooled, ctl)
    for j, used in ((3, False), (2, True), (1, True), (0, True)):
        alt = list(ctl)
        alt[j] = alt[j] + f(1, 16, cfg.hidden)      # (a constant shift would vanish in the next LayerNorm)
        changed = np.abs(O.transformer_forward(sd, cfg, x, t, ctx, pooled, alt) - base).max() > 1e-3
        assert changed == used, j
    # residual 2 enters after block 3 only: identical hidden states up to there means equal outputs when blocks 0-2 ignore it
    assert [int(i / (5 / 4)) for i in range(4)] == [0, 0, 1, 2]


def test_guidance_identities():
    sd = sd3.synth_sd3_state_dict(CFG)
    rng = np.random.default_rng(4)
    f = lambda *s: rng.standard_normal(s).astype(np.float32)
    x, ctx, pooled, cond, pair = f(1, 4, 8, 8), f(1, 3, CFG.joint_dim), f(1, CFG.pooled_dim), f(1, 4, 8, 8), f(1, 4, 8, 8)
    a = O.sample(sd, CFG, x, ctx, ctx, pooled, pooled, cond, pair, 2, 6.0)    # negative == positive: guidance cancels
    b = O.sample(sd, CFG, x, ctx, ctx, pooled, pooled, cond, pair, 2, 1.0)
    assert np.allclose(a, b, atol=1e-4)
    # one Euler step over the whole interval: x + (0 - 1) * v
    v = O.transformer_forward(sd, CFG, x, np.array([1000.0], np.float32), ctx, pooled,
                              O.controlnet_forward(sd, CFG, x, np.array([1000.0], np.float32), ctx, 0 * pooled, cond, pair))
    assert np.allclose(O.sample(sd, CFG, x, ctx, ctx, pooled, pooled, cond, pair, 1, 1.0), x - v, atol=1e-5)


def test_struct_layout_matches_header():
    """ctypes mirrors of pd_sd3_config / pd_sd3_args against the C header (sizes by the C compiler's rules)."""
    import ctypes as C
    assert C.sizeof(sd3.pd_sd3_config) == 16 * 4
    assert C.sizeof(sd3.pd_sd3_args) == 6 * 4 + 7 * 8 + 3 * 8
    assert sd3.pd_sd3_args.latents.offset == 24 and sd3.pd_sd3_args.pair.offset == 64 and sd3.pd_sd3_args.cn_pooled.offset == 72


def test_pipeline_host_logic_without_a_gpu():
    """The SD3 pipeline mirror's host pieces that need no engine: check_inputs errors, the sigma grid, image / latent
    preparation (promptdiffusioncontrolnetpipeline_sd3.py:541-698, :1129-1146)."""
    from prompt_diffusion_amd.pipeline_sd3 import StableDiffusion3PromptDiffusionPipeline as Pipe
    p = Pipe(None, shift=3.0)
    e = np.zeros((1, 4, 8), np.float32)
    with pytest.raises(ValueError, match="divisible by 8"):
        p.check_inputs(None, None, None, 100, 64, prompt_embeds=e, pooled_prompt_embeds=e)
    with pytest.raises(ValueError, match="Cannot forward both `prompt_2`"):
        p.check_inputs(None, "x", None, 64, 64, prompt_embeds=e, pooled_prompt_embeds=e)
    with pytest.raises(ValueError, match="has to be of type"):
        p.check_inputs(3, None, None, 64, 64)
    with pytest.raises(ValueError, match="negative_pooled_prompt_embeds"):
        p.check_inputs(None, None, None, 64, 64, prompt_embeds=e, pooled_prompt_embeds=e, negative_prompt_embeds=e)
    p.check_inputs("a", None, None, 64, 64, negative_prompt="b", max_sequence_length=512)
    sig, n = p._sigmas(5, None)
    assert n == 5 and np.array_equal(sig, sd3.flow_match_sigmas(5, 3.0))
    sig, n = p._sigmas(99, [1.0, 0.5])                       # custom sigmas decide the step count, get shifted, end in 0
    assert n == 2 and np.allclose(sig, [1.0, 3 * 0.5 / 2.0, 0.0])
    img = p.prepare_image(np.full((16, 16, 3), 0.75, np.float32), batch_size=3, num_images_per_prompt=1)
    assert img.shape == (3, 3, 16, 16) and np.allclose(img, 0.5)          # HWC in [0, 1] -> NCHW in [-1, 1], repeated per prompt
    t = p.prepare_image(torch.full((2, 3, 8, 8), -0.25), batch_size=2, num_images_per_prompt=2)
    assert t.shape == (4, 3, 8, 8) and np.allclose(t, -0.25)              # tensors pass unchanged, repeated per image
    a = p.prepare_latents(2, 4, 64, 32, np.random.default_rng(0))
    b = p.prepare_latents(2, 4, 64, 32, np.random.default_rng(0))
    assert a.shape == (2, 4, 8, 4) and np.array_equal(a, b)
    g = p.prepare_latents(1, 4, 64, 32, torch.Generator().manual_seed(3))
    assert np.array_equal(g, torch.randn((1, 4, 8, 4), generator=torch.Generator().manual_seed(3)).numpy())
    with pytest.raises(ValueError, match="list of generators"):
        p.prepare_latents(2, 4, 64, 32, [np.random.default_rng(0)])
