"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/*.h declares,
and fails loudly (no CPU fallback) when no GPU is visible."""
import ctypes as C
import os
import re

import pytest

from prompt_diffusion_amd import engine as E
from prompt_diffusion_amd import weights as W

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(pd_[a-z0-9_]+)\s*\(", txt)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(E.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return E.load_library()


def test_exports_every_declared_symbol(lib):
    names = _declared("pdengine.h") + _declared("pdengine_ops.h")
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/ but not exported"
    assert sorted(set(E.EXPORTS)) == sorted(set(names))
    assert lib.pd_abi_version() == 2


def test_struct_layout_matches_header(lib):
    # sizes are part of the ABI: 35 int32 (+4 pad) + 2 double + 2 int32 + (2 + 8 + 2) int32 + double + (4 text + 2 reserved) int32
    assert C.sizeof(E.pd_config) == 4 * 36 + 16 + 4 * 2 + 4 * 12 + 8 + 4 * 6
    assert C.sizeof(E.pd_sample_args) % 8 == 0


def test_no_cpu_fallback_without_gpu(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(E.PdError, match="no HIP device|MI355X"):
        E.Engine(W.TINY, precision="f32")


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "prompt-diffusion_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "pd_oracle" not in src and "from oracle" not in src and "import oracle" not in src, f


def test_sd3_rejects_unbuilt_block_variants():
    """promptdiffusioncontrolnet_sd3.py:104-105: qk_norm other than rms_norm, or a second attention in the context_pre_only last
    block, select blocks this engine does not build; they must be refused before any engine exists."""
    import pytest
    from prompt_diffusion_amd import sd3
    import dataclasses
    with pytest.raises(NotImplementedError):
        sd3.SD3Engine(dataclasses.replace(sd3.SD3_TINY, qk_norm="layer_norm"))
    with pytest.raises(NotImplementedError):
        sd3.SD3Engine(dataclasses.replace(sd3.SD3_TINY, dual_attention_layers=(sd3.SD3_TINY.layers - 1,)))
    with pytest.raises(ValueError):
        sd3.SD3Engine(dataclasses.replace(sd3.SD3_TINY, cn_dual_attention_layers=(7,)))
    # the SD3.5-style parameter inventory: RMSNorm weights per attention, attn2.* and 9 modulation chunks in the dual blocks
    cfg = dataclasses.replace(sd3.SD3_TINY, qk_norm="rms_norm", dual_attention_layers=(0,), cn_dual_attention_layers=(1,))
    sh = sd3.sd3_param_shapes(cfg)
    D = cfg.hidden
    assert sh["transformer.transformer_blocks.0.norm1.linear.weight"] == (9 * D, D)
    assert sh["transformer.transformer_blocks.1.norm1.linear.weight"] == (6 * D, D)
    assert sh["controlnet.transformer_blocks.1.attn2.to_out.0.weight"] == (D, D)
    assert sh["transformer.transformer_blocks.2.attn.norm_added_k.weight"] == (cfg.head_dim,)
    assert sh["controlnet.down_proj.weight"] == (3, 6, 3, 3)
    # joint_attention_dim = None for the ControlNet (:147-160): single blocks, no context parameters at all
    with pytest.raises(ValueError):
        sd3.SD3Engine(dataclasses.replace(sd3.SD3_TINY, cn_single_blocks=True, cn_dual_attention_layers=(0,)))
    sh = sd3.sd3_param_shapes(dataclasses.replace(sd3.SD3_TINY, cn_single_blocks=True, qk_norm="rms_norm"))
    cn = [k for k in sh if k.startswith("controlnet.")]
    assert not any("context" in k or "add_" in k or "norm_added" in k for k in cn)
    assert "controlnet.transformer_blocks.0.attn.norm_q.weight" in sh and "transformer.context_embedder.weight" in sh
