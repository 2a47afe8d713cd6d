"""conv_patch4.hip against the first generation through the engine (option patch4): differing outputs and their size, per mode."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from prompt_diffusion_amd import engine as E, weights as W
for prec in ("f16", "bf16"):
    e = E.Engine(W.TINY, precision=prec)
    g = np.random.default_rng(21)
    for (B, Cin, H, Cout, ups) in [(48, 64, 32, 168, False), (48, 128, 32, 164, False), (50, 64, 16, 96, True)]:
        x = g.standard_normal((B, Cin, H, H), dtype=np.float32)
        w = (g.standard_normal((Cout, Cin, 3, 3), dtype=np.float32) / np.sqrt(Cin * 9)).astype(np.float32)
        b = g.standard_normal(Cout, dtype=np.float32) * 0.1
        Ho = 2 * H if ups else H
        r = g.standard_normal((B, Cout, Ho, Ho), dtype=np.float32)
        e.set_option("patch4", 1); y1 = e.op_conv2d(x, w, b, upsample=ups); y1r = e.op_conv2d(x, w, b, upsample=ups, scale=0.75, residual=r, stream_out=True)
        e.set_option("patch4", 0); y0 = e.op_conv2d(x, w, b, upsample=ups); y0r = e.op_conv2d(x, w, b, upsample=ups, scale=0.75, residual=r, stream_out=True)
        for name, a, c in (("plain", y1, y0), ("scale+res", y1r, y0r)):
            d = a != c
            idx = np.argwhere(d)
            print(prec, (B, Cin, H, Cout, ups), name, "differing", int(d.sum()), "of", d.size, "max |diff|", float(np.abs(a - c).max()), "first", idx[:3].tolist(),
                  "channels", sorted(set(idx[:, 1].tolist()))[:12] if len(idx) else [])
    e.close()
