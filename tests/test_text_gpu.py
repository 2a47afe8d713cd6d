"""Cond-stage CLIP text transformer in the engine (pd_text_encode) against the fixtures generated from
transformers' CLIPTextModel -- the module FrozenCLIPEmbedder wraps (ldm/modules/encoders/modules.py:88-131) -- and the
oracle.  Tolerances: fp32 mode 1e-4 of the tensor's max; fp16 mode 4e-3; bf16 mode 3e-2 (12 pre-LN blocks of 2-byte GEMMs)."""
import os

import numpy as np
import pytest

from oracle import pd_oracle as O
from prompt_diffusion_amd import engine as E
from prompt_diffusion_amd import weights as W

pytestmark = pytest.mark.gpu


def relerr(a, b):
    return float(np.abs(np.asarray(a) - b).max() / (np.abs(b).max() + 1e-30))


@pytest.mark.parametrize("prec,tol", [("f32", 1e-4), ("f16", 4e-3), ("bf16", 3e-2)])
@pytest.mark.parametrize("tag,cfg", [("tiny_b3", W.TINY), ("sd15_b2", W.SD15)])
def test_text_encode_matches_transformers_fixture(golden_dir, tag, cfg, prec, tol):
    g = np.load(os.path.join(golden_dir, f"clip_{tag}.npz"))
    e = E.Engine(cfg, precision=prec)
    try:
        assert e.text_weights_missing() == 2 + 16 * cfg.text_layers + 2
        with pytest.raises(E.PdError, match="not loaded"):
            e.text_encode(g["ids"])
        e.load_state_dict(W.synth_text_state_dict(cfg), strict=False)
        assert e.text_weights_missing() == 0
        z = e.text_encode(g["ids"])
        assert z.shape == g["ids"].shape + (cfg.context_dim,) and z.dtype == np.float32
        if "z" in g:
            assert relerr(z, g["z"]) < tol
        else:
            assert relerr(z.reshape(-1)[::int(g["z_stride"])][:16384], g["z_sub"]) < tol
        # clip_skip = 2: hidden_states[-3] through final_layer_norm, as the (D) pipeline derives it (:403-413)
        z2s = e.text_encode(g["ids"], clip_skip=2)
        if "z_skip2" in g:
            assert relerr(z2s, g["z_skip2"]) < tol
        elif "z_skip2_sub" in g:
            assert relerr(z2s.reshape(-1)[::int(g["z_stride"])][:16384], g["z_skip2_sub"]) < tol
        # causality: a token's embedding must not depend on later tokens
        ids2 = g["ids"].copy()
        ids2[:, 40:] = 5
        z2 = e.text_encode(ids2)
        assert np.array_equal(z2[:, :40], z[:, :40])
        with pytest.raises(ValueError):
            e.text_encode(g["ids"][:, :10])
    finally:
        e.close()


def test_text_encode_ragged_batch_against_oracle():
    import torch   # imported before the engine exists: Engine() then brings torch's HIP runtime up first
    cfg = W.TINY
    e = E.Engine(cfg, precision="f32")
    try:
        sd = W.synth_text_state_dict(cfg)
        e.load_state_dict(sd, strict=False)
        ids = W.synth_token_ids(cfg, 5, seed=99)
        assert relerr(e.text_encode(ids), O.clip_text_forward(sd, cfg, ids)) < 1e-4
        zt = e.text_encode(torch.from_numpy(ids).cuda())
        assert zt.is_cuda and relerr(zt.cpu().numpy(), O.clip_text_forward(sd, cfg, ids)) < 1e-4
    finally:
        e.close()
