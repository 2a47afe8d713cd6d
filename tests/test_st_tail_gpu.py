"""The fused transformer tail (csrc/st_tail.hip) on the GPU.

pd_op_spatial_transformer runs one whole SpatialTransformer block of the SD1.5 network (C = 320, 8 heads x 40, context 77 x 768)
through the code path a sampling step takes: in the 2-byte modes everything after the self-attention product is ONE kernel
(attn1.to_out + residual, norm2, attn2 over the context keys, norm3, GEGLU feed-forward, proj_out + x); the fp32-storage modes
keep the per-layer kernels.  All four modes are held against oracle.spatial_transformer (attention.py:321-340, :271-275; the
oracle is pinned by the reference's own st_a / st_b fixtures in tests/test_oracle_golden.py) on the same seeded weights.

Tolerances (max-abs / max-abs of the block output, |y| ~ 5): the block chains 9 contractions, so the 2-byte modes are given twice
the single-kernel bound of tests/test_kernels_gpu.py (f16 5e-3, bf16 4e-2); the fp32-class modes 1e-4.
At BASELINE's full size (forward batch 16, 64 x 64 = 65536 tokens) the oracle cannot run: the fused kernel is compared with the
unfused per-layer path of the same engine, and checked for determinism and batch independence."""
import numpy as np
import pytest

from oracle import pd_oracle as O
from prompt_diffusion_amd import engine as E
from prompt_diffusion_amd import weights as W

pytestmark = pytest.mark.gpu
PRE = "model.diffusion_model."
BLK = "input_blocks.1.1."
TOL = {"f32": 1e-4, "f16x2": 1e-4, "f16": 5e-3, "bf16": 4e-2}


def relerr(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def block_weights():
    return {n: W.synth_tensor(n, s, k) for n, s, k in W.param_spec(W.SD15) if n.startswith(PRE + BLK)}


def make_engine(prec, sd, **kw):
    e = E.Engine(W.SD15, precision=prec, **kw)
    for n, a in sd.items():
        e.load_tensor(n, a)
    return e


def inputs(B, H, Wd, seed=5):
    r = np.random.default_rng(seed)
    return r.standard_normal((B, 320, H, Wd), dtype=np.float32), r.standard_normal((B, 77, 768), dtype=np.float32)


@pytest.mark.parametrize("prec", ["f32", "f16x2", "f16", "bf16"])
@pytest.mark.parametrize("B,H,Wd", [(2, 16, 16), (1, 8, 48), (3, 8, 8)])   # 8 x 8 = 64 tokens: not a multiple of 128 -> per-layer path
def test_block_matches_oracle(prec, B, H, Wd):
    sd = block_weights()
    x, ctx = inputs(B, H, Wd)
    ref = O.spatial_transformer(O.Net(sd, PRE), BLK, x, ctx, heads=8)
    e = make_engine(prec, sd)
    y = e.op_spatial_transformer(PRE + BLK, x, ctx)
    e.close()
    assert np.isfinite(y).all()
    assert relerr(y, ref) < TOL[prec], (prec, relerr(y, ref))


@pytest.mark.parametrize("prec", ["f16", "bf16"])
def test_fused_equals_per_layer_path(prec):
    """same engine, option st_fuse on / off: both within the mode's bound of the oracle and of each other"""
    sd = block_weights()
    x, ctx = inputs(2, 16, 16, seed=9)
    ref = O.spatial_transformer(O.Net(sd, PRE), BLK, x, ctx, heads=8)
    e = make_engine(prec, sd)
    launches0 = e.stat("launches")
    y1 = e.op_spatial_transformer(PRE + BLK, x, ctx)
    n_fused = e.stat("launches") - launches0
    e.set_option("st_fuse", 0)
    launches0 = e.stat("launches")
    y0 = e.op_spatial_transformer(PRE + BLK, x, ctx)
    n_plain = e.stat("launches") - launches0
    e.close()
    assert n_fused < n_plain - 6, (n_fused, n_plain)     # 12 launches became 5 (statistics, coefficients, front, attention, tail)
    assert relerr(y1, ref) < TOL[prec] and relerr(y0, ref) < TOL[prec]
    assert relerr(y1, y0) < TOL[prec]


def test_stream_f32_and_context_variants():
    """fp32 residual stream (option stream_f32) takes the per-layer path (the fused kernel carries the 2-byte stream type only)."""
    sd = block_weights()
    x, ctx = inputs(1, 16, 16, seed=3)
    ref = O.spatial_transformer(O.Net(sd, PRE), BLK, x, ctx, heads=8)
    e = make_engine("f16", sd, stream_f32=True)
    y = e.op_spatial_transformer(PRE + BLK, x, ctx)
    e.close()
    assert relerr(y, ref) < TOL["f16"]


def test_full_size_properties():
    """BASELINE's headline shape of this block: forward batch 16, 64 x 64 tokens (M = 65536)."""
    sd = block_weights()
    x, ctx = inputs(16, 64, 64, seed=11)
    e = make_engine("f16", sd)
    y = e.op_spatial_transformer(PRE + BLK, x, ctx)
    y2 = e.op_spatial_transformer(PRE + BLK, x, ctx)
    np.testing.assert_array_equal(y, y2)                                    # deterministic
    perm = np.arange(16)[::-1].copy()
    yp = e.op_spatial_transformer(PRE + BLK, x[perm], ctx[perm])
    np.testing.assert_array_equal(yp, y[perm])                              # samples are independent (no cross-sample state)
    e.set_option("st_fuse", 0)
    y0 = e.op_spatial_transformer(PRE + BLK, x, ctx)
    e.close()
    assert np.isfinite(y).all()
    assert relerr(y, y0) < TOL["f16"], relerr(y, y0)
