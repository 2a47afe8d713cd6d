"""Runs the 64x64-level self-attention shape (B*2=16, N=4096, 8 heads x dh 40) a few times through pd_op_attention:
a target for `rocprofv3 --pmc ... -- python tools/attn_bench.py` (counters of attn2_kernel alone)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from prompt_diffusion_amd import engine as E, weights as W
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
e = E.Engine(W.SD15, precision=sys.argv[2] if len(sys.argv) > 2 else "f16")
g = np.random.default_rng(0)
q = g.standard_normal((16, 4096, 320), dtype=np.float32)
k = g.standard_normal((16, 4096, 320), dtype=np.float32)
v = g.standard_normal((16, 4096, 320), dtype=np.float32)
for _ in range(reps):
    o = e.op_attention(q, k, v)
print("ok", float(np.abs(o).mean()))
