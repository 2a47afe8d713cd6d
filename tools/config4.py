"""BASELINE config #4: 768x768, UniPC 20 steps, bs 8, through PromptDiffusionPipeline with the host UniPC plug-in
(random-init SD1.5 weights, synthetic inputs).  Prints images/s; the loop is engine eps evaluations + host scheduler."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from prompt_diffusion_amd import engine as E, weights as W
from prompt_diffusion_amd.pipeline import PromptDiffusionPipeline
from prompt_diffusion_amd.schedulers import UniPCMultistepScheduler
size = int(sys.argv[1]) if len(sys.argv) > 1 else 768
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
B = 8
e = E.Engine(W.SD15, precision="bf16"); e.init_random_weights(1234)
g = np.random.default_rng(0)
pe = g.standard_normal((B, 77, 768), dtype=np.float32); ne = g.standard_normal((B, 77, 768), dtype=np.float32)
img = lambda: g.random((B, size, size, 3), dtype=np.float32)
q, a, b = img(), img(), img()
lat = g.standard_normal((B, 4, size // 8, size // 8), dtype=np.float32)
pipe = PromptDiffusionPipeline(e, scheduler=UniPCMultistepScheduler())
kw = dict(prompt_embeds=pe, negative_prompt_embeds=ne, image=q, image_pair=[a, b], num_inference_steps=steps, guidance_scale=7.5,
          latents=lat, output_type="latent", height=size, width=size)
out = pipe(**kw).images
t0 = time.perf_counter(); out = pipe(**kw).images; dt = time.perf_counter() - t0
assert np.isfinite(np.asarray(out)).all()
print(f"config4 {size}x{size} UniPC {steps} steps bs {B}: {dt*1e3:.1f} ms -> {B/dt:.3f} img/s; |lat| mean {np.abs(np.asarray(out)).mean():.4f}")
