import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from prompt_diffusion_amd import engine as E, weights as W
e = E.Engine(W.SD15, precision="bf16")
print(e.bench_linear(8192, 8192, 8192, False, 3))
