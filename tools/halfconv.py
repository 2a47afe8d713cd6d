"""The 64x64 320 -> 320 patch conv at batch 16 / 8 / 4 / 2 under every kernel generation (the half-batch launches of the shared CFG front)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from prompt_diffusion_amd import engine as E, weights as W
for opts in ({}, {"patch4": 0}, {"patch4": 0, "conv_patch2": 0}, {"patch4":0, "conv_patch2_tiles": 1}):
    e = E.Engine(W.TINY, precision="f16")
    for k, v in opts.items(): e.set_option(k, v)
    for (B, H, Wd, ci, co) in [(8, 64, 64, 320, 320), (16, 64, 64, 320, 320), (8,64,64,640,320), (4,64,64,320,320), (2,64,64,320,320)]:
        ms = e.bench_conv3x3(B, H, Wd, ci, co, iters=30)
        fl = 2.0 * B * H * Wd * co * ci * 9
        print(f"{opts} conv3x3 B{B} {H}x{Wd} {ci}->{co}: {ms*1e3:8.1f} us  {fl/ms/1e9:7.1f} TF/s", flush=True)
    e.close()
