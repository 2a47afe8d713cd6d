"""Isolated timing of Linear / conv1x1 layers of the headline workload through pd_bench_linear (back-to-back launches).
usage: python tools/bench_linear.py [--opt k=v ...]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from prompt_diffusion_amd import engine as E, weights as W
ap = argparse.ArgumentParser(); ap.add_argument("--opt", action="append", default=[])
a = ap.parse_args()
e = E.Engine(W.SD15, precision="f16")
for o in a.opt:
    k, v = o.split("="); e.set_option(k, int(v))
shapes = [(65536, 320, 320, 0), (65536, 320, 320, 1), (65536, 320, 960, 0), (65536, 1280, 320, 1), (16384, 640, 640, 0), (16384, 640, 640, 1),
          (16384, 640, 1920, 0), (16384, 2560, 640, 1), (4096, 1280, 1280, 1), (4096, 1280, 3840, 0), (4096, 5120, 1280, 1), (1024, 1280, 1280, 1),
          (65536, 320, 2560, 2), (16384, 640, 5120, 2), (4096, 1280, 10240, 2)]
for M, K, N, r in shapes:
    ms = e.lib.pd_bench_linear and None
    import ctypes as C
    t = C.c_float(); e._check(e.lib.pd_bench_linear(e._h, M, K, N, r, 20, C.byref(t))); ms = float(t.value)
    gb = (M * K + M * N * (2 if r else 1)) * 2 / 1e9
    print(f"M={M:6d} K={K:5d} N={N:5d} res={r}  {ms*1e3:8.1f} us  {2*M*K*N/ms/1e9:7.1f} TF/s  {gb/ms*1e3/1e3:6.2f} TB/s(min traffic)")
