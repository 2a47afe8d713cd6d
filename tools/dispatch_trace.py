"""Which kernel family / tile / split every contraction launch of one denoising step takes (engine option verbose = 2, stderr):
python tools/dispatch_trace.py [--batch 8] [--size 512];  prints the distinct lines with their counts."""
import argparse, collections, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    from prompt_diffusion_amd import engine as E, weights as W
    B = int(sys.argv[2]); L = int(sys.argv[3]) // 8
    e = E.Engine(W.SD15, precision="f16")
    e.init_random_weights(3)
    inp = W.synth_inputs(W.SD15, B, L, L)
    kw = dict(x_T=inp["x_T"], ctx_cond=inp["ctx_cond"], ctx_uncond=inp["ctx_uncond"], pair=inp["pair"], query=inp["query"], steps=50, cfg_scale=7.5)
    e.sample_begin(**kw)
    e.sample_step(0)
    e.set_option("verbose", 2)
    e.sample_step(1)
    e.set_option("verbose", 0)
    e.sample_end()
    sys.exit(0)
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--size", type=int, default=512)
a = ap.parse_args()
r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(a.batch), str(a.size)], capture_output=True, text=True)
cnt = collections.Counter(l for l in r.stderr.splitlines() if l.startswith("[pdengine] gemm"))
for l, n in sorted(cnt.items(), key=lambda kv: (kv[0].split(":")[1].split()[0], -kv[1])):
    print(f"{n:3d} x {l[len('[pdengine] gemm '):]}")
