"""Host-side surface of PromptDiffusionPipeline.__call__: same exception types / conditions as the reference's
check_inputs (pipeline_prompt_diffusion.py:559-757), checked before anything touches the GPU.  CPU only."""
import numpy as np
import pytest

from prompt_diffusion_amd import weights as W
from prompt_diffusion_amd.pipeline import PromptDiffusionPipeline


class _NoEngine:
    cfg = W.TINY

    def __getattr__(self, name):
        raise AssertionError(f"engine.{name} touched before input validation finished")


@pytest.fixture
def pipe():
    return PromptDiffusionPipeline(_NoEngine())


def _imgs(b=1, hw=64):
    img = np.zeros((b, hw, hw, 3), np.float32)
    return img, [img.copy(), img.copy()]


def test_prompt_and_embeds_are_exclusive(pipe):
    img, pair = _imgs()
    emb = np.zeros((1, 77, 96), np.float32)
    with pytest.raises(ValueError, match="Cannot forward both `prompt`"):
        pipe(prompt="a", prompt_embeds=emb, image=img, image_pair=pair)
    with pytest.raises(ValueError, match="Provide either `prompt` or `prompt_embeds`"):
        pipe(image=img, image_pair=pair)
    with pytest.raises(ValueError, match="has to be of type `str` or `list`"):
        pipe(prompt=3, image=img, image_pair=pair)
    with pytest.raises(ValueError, match="Cannot forward both `negative_prompt`"):
        pipe(prompt_embeds=emb, negative_prompt="x", negative_prompt_embeds=emb, image=img, image_pair=pair)
    with pytest.raises(ValueError, match="must have the same shape"):
        pipe(prompt_embeds=emb, negative_prompt_embeds=np.zeros((2, 77, 96), np.float32), image=img, image_pair=pair)


def test_image_pair_must_have_two_entries(pipe):
    img, pair = _imgs()
    emb = np.zeros((1, 77, 96), np.float32)
    with pytest.raises(ValueError, match="list size equals to two"):
        pipe(prompt_embeds=emb, negative_prompt_embeds=emb, image=img, image_pair=pair + [img])
    with pytest.raises(TypeError, match="image must be passed"):
        pipe(prompt_embeds=emb, negative_prompt_embeds=emb, image="not an image", image_pair=pair)
    with pytest.raises(ValueError, match="image batch size must be same as prompt batch size"):
        big, _ = _imgs(b=3)
        pipe(prompt_embeds=np.zeros((2, 77, 96), np.float32), negative_prompt_embeds=np.zeros((2, 77, 96), np.float32),
             image=big, image_pair=pair)


def test_scale_and_guidance_window(pipe):
    img, pair = _imgs()
    emb = np.zeros((1, 77, 96), np.float32)
    kw = dict(prompt_embeds=emb, negative_prompt_embeds=emb, image=img, image_pair=pair)
    with pytest.raises(TypeError, match="must be type `float`"):
        pipe(controlnet_conditioning_scale=1, **kw)
    with pytest.raises(ValueError, match="cannot be larger or equal"):
        pipe(control_guidance_start=0.5, control_guidance_end=0.5, **kw)
    with pytest.raises(ValueError, match="can't be smaller than 0"):
        pipe(control_guidance_start=-0.1, **kw)
    with pytest.raises(ValueError, match="can't be larger than 1.0"):
        pipe(control_guidance_end=1.5, **kw)
    with pytest.raises(ValueError, match="callback_steps"):
        pipe(callback_steps=0, **kw)
    with pytest.raises(ValueError, match="callback_on_step_end_tensor_inputs"):
        pipe(callback_on_step_end_tensor_inputs=["nope"], **kw)


def test_unsupported_options_raise_not_implemented(pipe):
    img, pair = _imgs()
    emb = np.zeros((1, 77, 96), np.float32)
    kw = dict(prompt_embeds=emb, negative_prompt_embeds=emb, image=img, image_pair=pair)
    with pytest.raises(NotImplementedError):
        pipe(ip_adapter_image=img, **kw)
    with pytest.raises(NotImplementedError):
        pipe(cross_attention_kwargs={"scale": 0.5}, **kw)


def test_prepare_image_matches_reference_convention(pipe):
    """[0,1] range, NCHW, repeat to the batch (pipeline :236-238, :760-788)."""
    from PIL import Image
    im = Image.fromarray((np.arange(64 * 64 * 3) % 256).astype("uint8").reshape(64, 64, 3))
    x = pipe.prepare_image(im, 64, 64, batch_size=3, num_images_per_prompt=1)
    assert x.shape == (3, 3, 64, 64) and x.dtype == np.float32
    assert 0.0 <= x.min() and x.max() <= 1.0
    np.testing.assert_allclose(x[0, :, 0, 1], np.array([3, 4, 5]) / 255.0)
    y = pipe.prepare_image(np.full((2, 32, 32, 3), 0.25, np.float32), None, None, batch_size=2, num_images_per_prompt=2)
    assert y.shape == (4, 3, 32, 32)
