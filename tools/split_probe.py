"""conv3x3 through the patch kernel with the channel chunks split into slices vs unsplit (igemm / plain patch): f32 engine -> differences
must be summation-order only."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pd_oracle as O
from prompt_diffusion_amd import engine as E, weights as W
for prec in ("f32", "f16"):
    e = E.Engine(W.TINY, precision=prec)
    g = np.random.default_rng(3)
    for B, Cin, H, Cout in [(2, 320, 32, 320), (2, 320, 64, 320), (2, 640, 32, 640), (2, 960, 32, 640), (2, 320, 32, 4), (2, 1280, 16, 1280), (2, 192, 32, 160), (2, 64, 32, 160)]:
        x = g.standard_normal((B, Cin, H, H), dtype=np.float32)
        w = (g.standard_normal((Cout, Cin, 3, 3), dtype=np.float32) / np.sqrt(Cin * 9)).astype(np.float32)
        b = g.standard_normal(Cout, dtype=np.float32) * 0.1
        if prec == "f16":
            x = x.astype(np.float16).astype(np.float32); w = w.astype(np.float16).astype(np.float32)
        ref = O.conv2d(x, w, b)
        out = {}
        for name, opts in (("split", {"patch_split": 1, "patch_split_min": 1, "patch_split_tiles": 32}), ("nosplit", {"patch_split": 0})):
            for k, v in opts.items(): e.set_option(k, v)
            out[name] = e.op_conv2d(x, w, b)
        rel = lambda a, r: float(np.abs(a - r).max() / np.abs(r).max())
        print(f"{prec} B={B} Cin={Cin} {H}x{H} Cout={Cout}: split vs oracle {rel(out['split'], ref):.2e}  nosplit vs oracle {rel(out['nosplit'], ref):.2e}  split vs nosplit {rel(out['split'], out['nosplit']):.2e}", flush=True)
    e.close()
