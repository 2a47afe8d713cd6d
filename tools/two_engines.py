"""Experiment: two engines (own streams, weights) each sampling half of the batch concurrently vs one engine on the full batch."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from prompt_diffusion_amd import engine as E, weights as W
dev = torch.device("cuda", 0)
def mk(B, seed):
    g = torch.Generator(device=dev).manual_seed(seed)
    return dict(x_T=torch.randn((B, 4, 64, 64), generator=g, device=dev), ctx_cond=torch.randn((B, 77, 768), generator=g, device=dev),
                ctx_uncond=torch.randn((B, 77, 768), generator=g, device=dev), pair=torch.rand((B, 6, 512, 512), generator=g, device=dev),
                query=torch.rand((B, 3, 512, 512), generator=g, device=dev), steps=50, cfg_scale=7.5)
nE = int(sys.argv[1]) if len(sys.argv) > 1 else 2
B = 8 // nE
engs = [E.Engine(W.SD15, precision="bf16") for _ in range(nE)]
for i, e in enumerate(engs): e.init_random_weights(1)
kws = [mk(B, i) for i in range(nE)]
def run(i): engs[i].ddim_sample(**kws[i])
def all_():
    th = [threading.Thread(target=run, args=(i,)) for i in range(nE)]
    for t in th: t.start()
    for t in th: t.join()
all_(); torch.cuda.synchronize()
t0 = time.perf_counter(); all_(); all_(); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 2
print(f"{nE} engine(s) x bs {B}: {dt*1e3:.1f} ms per 8 images -> {8/dt:.3f} img/s")
