"""Asymptotic K-loop rate of igemm_kernel tile variants: large M,N with growing K (pd_bench_linear)."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from prompt_diffusion_amd import engine as E, weights as W
ap = argparse.ArgumentParser(); ap.add_argument("--opt", action="append", default=[])
a = ap.parse_args()
e = E.Engine(W.SD15, precision="f16")
for o in a.opt:
    k, v = o.split("="); e.set_option(k, int(v))
for M, K, N in [(16384, 320, 2560), (16384, 1280, 2560), (16384, 5120, 2560), (16384, 20480, 2560), (65536, 5120, 1280), (8192, 8192, 8192)]:
    ms = e.bench_linear(M, K, N, False, 10)
    print(f"M={M:6d} K={K:5d} N={N:5d}  {ms*1e3:8.1f} us  {2*M*K*N/ms/1e9:7.1f} TF/s")
