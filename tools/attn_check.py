"""Quick numerical check of the attention kernel over a few shapes (debug aid; the assertions live in tests/)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from prompt_diffusion_amd import engine as E, weights as W
prec = sys.argv[1] if len(sys.argv) > 1 else "f16"
e = E.Engine(W.TINY, precision=prec)
heads = W.TINY.num_heads
def ref(q, k, v):
    B, Nq, C = q.shape; dh = C // heads
    sp = lambda t: t.reshape(t.shape[0], t.shape[1], heads, dh).transpose(0, 2, 1, 3).astype(np.float64)
    s = sp(q) @ sp(k).transpose(0, 1, 3, 2) * dh ** -0.5
    s -= s.max(-1, keepdims=True); p = np.exp(s); p /= p.sum(-1, keepdims=True)
    return (p @ sp(v)).transpose(0, 2, 1, 3).reshape(B, Nq, C).astype(np.float32)
g = np.random.default_rng(7)
shapes = [tuple(int(x) for x in a.split(",")) for a in sys.argv[2:]] or [(1, 128, 64, 320), (1, 128, 128, 320), (1, 1024, 1024, 320), (2, 100, 77, 320), (1, 128, 70, 320), (1, 64, 77, 64), (1, 192, 192, 640), (1, 100, 77, 640), (1, 64, 200, 256)]
for B, Nq, Nk, C in shapes:
    q, k, v = (g.standard_normal((B, n, C), dtype=np.float32) for n in (Nq, Nk, Nk))
    r = ref(q, k, v); o = e.op_attention(q, k, v)
    err = np.abs(o - r).max(axis=(0, 2)) / np.abs(r).max()
    print(f"B{B} Nq{Nq} Nk{Nk} C{C} dh{C//heads}: max relerr {err.max():.3e}; worst query {err.argmax()}; per-head {[float('%.2e' % (np.abs(o - r).reshape(B, Nq, heads, -1)[:, :, hh].max() / np.abs(r).max())) for hh in range(heads)]}")
