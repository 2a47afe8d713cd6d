"""gemm_ring.hip against NumPy (op_linear through the engine's gemm with option ring) + isolated timings with / without it."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from prompt_diffusion_amd import engine as E, weights as W
import ctypes as C
prec = sys.argv[1] if len(sys.argv) > 1 else "f16"
e = E.Engine(W.SD15, precision=prec)
rng = np.random.default_rng(0)
worst = 0.0
for tile in (0, 1):
    e.set_option("ring", 1000); e.set_option("ring_tile", tile)
    for M, K, N in [(128, 128, 160), (130, 192, 164), (1000, 320, 320), (4096, 640, 640), (777, 1280, 1920), (2048 * 9, 128, 480), (65, 2560, 40),
                    (16384, 640, 640), (40000, 320, 320)]:
        x = rng.standard_normal((M, K)).astype(np.float32); w = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
        b = rng.standard_normal(N).astype(np.float32)
        e.set_option("ring", 1000)
        y = e.op_linear(x, w, b)
        e.set_option("ring", 0)
        y0 = e.op_linear(x, w, b)
        h = np.float16 if prec == "f16" else None
        ref = (x.astype(np.float64) @ w.astype(np.float64).T + b)
        err = np.abs(y - ref).max() / np.abs(ref).max(); err0 = np.abs(y0 - ref).max() / np.abs(ref).max()
        same = np.array_equal(y, y0)
        worst = max(worst, err)
        print(f"tile {tile} M={M:6d} K={K:5d} N={N:5d} ring err {err:.2e} igemm err {err0:.2e} identical {same}", flush=True)
assert worst < (2e-3 if prec == "f16" else 2e-2), worst
for M, Cc in [(300, 64), (4096, 320), (1000, 640)]:
    x = rng.standard_normal((M, Cc)).astype(np.float32); w = (rng.standard_normal((8 * Cc, Cc)) / np.sqrt(Cc)).astype(np.float32); b = rng.standard_normal(8 * Cc).astype(np.float32) * 0.1
    e.set_option("ring", 1000); e.set_option("ring_geglu", 1); n0 = e.stat("ring_launches"); y = e.op_linear(x, w, b, geglu=True); took = e.stat("ring_launches") - n0
    e.set_option("ring_geglu", 0); y0 = e.op_linear(x, w, b, geglu=True)
    print(f"GEGLU M={M} C={Cc}: ring launches {took}, identical to igemm {np.array_equal(y, y0)}, max diff {np.abs(y - y0).max():.2e}", flush=True)
shapes = [(65536, 320, 2560, 2), (16384, 640, 5120, 2), (4096, 1280, 10240, 2), (16384, 640, 640, 0), (16384, 640, 640, 1), (16384, 640, 1920, 0), (16384, 2560, 640, 1), (4096, 1280, 1280, 1), (4096, 1280, 3840, 0),
          (4096, 5120, 1280, 1), (65536, 320, 320, 1), (65536, 320, 960, 0), (65536, 1280, 320, 1), (1024, 1280, 1280, 1)]
for M, K, N, r in shapes:
    row = []
    for ring, tile in ((0, 0), (1000, 0), (1000, 1)):
        e.set_option("ring", ring); e.set_option("ring_tile", tile); e.set_option("ring_geglu", 1)
        t = C.c_float(); e._check(e.lib.pd_bench_linear(e._h, M, K, N, r, 20, C.byref(t))); row.append(float(t.value) * 1e3)
    print(f"M={M:6d} K={K:5d} N={N:5d} res={r}  igemm {row[0]:7.1f} us   ring128 {row[1]:7.1f} us ({2*M*K*N/row[1]/1e6:6.0f} TF/s)   ring256 {row[2]:7.1f} us ({2*M*K*N/row[2]/1e6:6.0f} TF/s)", flush=True)
