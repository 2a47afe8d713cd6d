"""Micro-benchmark of the conv3x3 implicit-GEMM kernel on a few hot shapes (HIP events on the engine stream)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from prompt_diffusion_amd import engine as E, weights as W
e = E.Engine(W.TINY, precision=sys.argv[1] if len(sys.argv) > 1 else "bf16")
shapes = [(16, 64, 64, 320, 320), (16, 32, 32, 640, 640), (16, 16, 16, 1280, 1280), (16, 8, 8, 1280, 1280), (16, 64, 64, 640, 320)]
if len(sys.argv) > 2: shapes = shapes[:int(sys.argv[2])]
for (B, H, Wd, ci, co) in shapes:
    ms = e.bench_conv3x3(B, H, Wd, ci, co, iters=20)
    fl = 2.0 * B * H * Wd * co * ci * 9
    print(f"conv3x3 B{B} {H}x{Wd} {ci}->{co}: {ms*1e3:8.1f} us  {fl/ms/1e9:7.1f} TF/s")
