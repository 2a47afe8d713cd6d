"""Runs one attention shape a few times through pd_op_attention (default: the 64x64-level self-attention, B*2 = 16,
N = 4096, 8 heads x dh 40): a target for `rocprofv3 --kernel-trace / --pmc ... -- python tools/attn_bench.py`.
usage: attn_bench.py [reps] [precision] [Nq] [Nk] [C]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from prompt_diffusion_amd import engine as E, weights as W
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
e = E.Engine(W.SD15, precision=sys.argv[2] if len(sys.argv) > 2 else "f16")
Nq = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
Nk = int(sys.argv[4]) if len(sys.argv) > 4 else Nq
C = int(sys.argv[5]) if len(sys.argv) > 5 else 320
g = np.random.default_rng(0)
q = g.standard_normal((16, Nq, C), dtype=np.float32)
k = g.standard_normal((16, Nk, C), dtype=np.float32)
v = g.standard_normal((16, Nk, C), dtype=np.float32)
for _ in range(reps):
    o = e.op_attention(q, k, v)
print("ok", float(np.abs(o).mean()))
