"""Per-shape device-time table of one sampling pass (HIP-event brackets around every contraction launch)."""
import argparse, os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from prompt_diffusion_amd import engine as E, weights as W
ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=8); ap.add_argument("--size", type=int, default=512)
ap.add_argument("--steps", type=int, default=4); ap.add_argument("--out", default="gpurun_out/shapes.csv"); ap.add_argument("--precision", default="bf16"); ap.add_argument("--opt", action="append", default=[])
a = ap.parse_args()
cfg = W.SD15; B = a.batch; h = w = a.size // 8; dev = torch.device("cuda", 0)
e = E.Engine(cfg, precision=a.precision); e.init_random_weights(1)
for o in a.opt:
    k, v = o.split('='); e.set_option(k, int(v))
g = torch.Generator(device=dev).manual_seed(0)
kw = dict(x_T=torch.randn((B, 4, h, w), generator=g, device=dev), ctx_cond=torch.randn((B, 77, 768), generator=g, device=dev),
          ctx_uncond=torch.randn((B, 77, 768), generator=g, device=dev), pair=torch.rand((B, 6, 8 * h, 8 * w), generator=g, device=dev),
          query=torch.rand((B, 3, 8 * h, 8 * w), generator=g, device=dev), steps=a.steps, cfg_scale=7.5)
e.ddim_sample(**kw)
n = e.sample_begin(**kw)          # profile the steps only (setup excluded)
e.set_option("profile", 1)
for i in range(n): e.sample_step(i)
e.synchronize(); e.profile_dump(a.out); e.set_option("profile", 0); e.sample_end()
rows = collections.OrderedDict()
for line in open(a.out).read().split("\n")[1:]:
    if not line: continue
    k, M, N, K, t, ms, fl = line.split(",")
    key = (int(k), int(M), int(N), int(K), int(t))
    r = rows.setdefault(key, [0, 0.0, 0.0]); r[0] += 1; r[1] += float(ms); r[2] += float(fl)
tot = sum(r[1] for r in rows.values())
print(f"{'klass':>5} {'M':>7} {'N':>6} {'K':>6} {'tap':>4} {'n/step':>6} {'ms/step':>8} {'%':>5} {'avg_us':>8} {'TF/s':>7}")
for key, r in sorted(rows.items(), key=lambda kv: -kv[1][1]):
    print(f"{key[0]:>5} {key[1]:>7} {key[2]:>6} {key[3]:>6} {key[4]:>4} {r[0]/n:>6.1f} {r[1]/n:>8.3f} {100*r[1]/tot:>5.1f} {1e3*r[1]/r[0]:>8.1f} {r[2]/r[1]/1e9:>7.1f}")
print("total contraction ms/step", tot / n)
