import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from prompt_diffusion_amd import engine as E, weights as W
g = np.load(os.path.join(ROOT, "tests", "golden", "net_sd15_b1_32x32_s50.npz"))
cfg = W.SD15
B, h, w, S = int(g["B"]), int(g["h"]), int(g["w"]), int(g["S"])
keep = [int(k) for k in g["keep"]]; ref = {k: g["x_inter"][j] for j, k in enumerate(keep)}
inp = W.synth_inputs(cfg, B, h, w)
kw = dict(x_T=inp["x_T"], ctx_cond=inp["ctx_cond"], ctx_uncond=inp["ctx_uncond"], pair=inp["pair"], query=inp["query"], steps=S, cfg_scale=float(g["cfg_scale"]))
relerr = lambda a, b: float(np.abs(a - b).max() / np.abs(b).max())
for name, opts in (("new", {}), ("old-split", {"patch_split_min": 4, "patch_split_tiles": 64}), ("no-slabgn", {"slab_gn": 0}), ("no-ring", {"ring": 0}),
                   ("all-old", {"patch_split_min": 4, "patch_split_tiles": 64, "slab_gn": 0, "ring": 0}), ("nosplit", {"patch_split": 0})):
    e = E.Engine(cfg, precision="f16")
    for n, a in W.iter_synth(cfg): e.load_tensor(n, a)
    for k, v in opts.items(): e.set_option(k, v)
    e.sample_begin(**kw)
    errs = []
    for i in range(0, 6):
        e.sample_set_latents(ref[i]); e.sample_step(i); errs.append(relerr(e.sample_get(), ref[i + 1]))
    e.sample_end(); e.close()
    print(name, ["%.3e" % v for v in errs], flush=True)
