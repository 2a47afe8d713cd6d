"""Micro-benchmark of the conv3x3 kernels on the hot shapes (HIP events on the engine stream).
usage: python tools/bench_conv.py [precision] [--opt k=v ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from prompt_diffusion_amd import engine as E, weights as W
args = [a for a in sys.argv[1:] if not a.startswith("--")]
e = E.Engine(W.TINY, precision=args[0] if args else "f16")
for a in sys.argv[1:]:
    if a.startswith("--opt="):
        k, v = a[6:].split("=")
        e.set_option(k, int(v))
shapes = [(16, 64, 64, 320, 320), (16, 32, 32, 640, 640), (16, 16, 16, 1280, 1280), (16, 8, 8, 1280, 1280), (16, 64, 64, 640, 320),
          (16, 64, 64, 960, 320), (16, 32, 32, 1280, 640)]
for (B, H, Wd, ci, co) in shapes:
    ms = e.bench_conv3x3(B, H, Wd, ci, co, iters=20)
    fl = 2.0 * B * H * Wd * co * ci * 9
    print(f"conv3x3 B{B} {H}x{Wd} {ci}->{co}: {ms*1e3:8.1f} us  {fl/ms/1e9:7.1f} TF/s")
