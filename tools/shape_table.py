#!/usr/bin/env python3
"""Per-shape table of the contraction launches of one headline sampling run (HIP-event brackets of the engine's profiler):
python tools/shape_table.py [--batch 8] [--ddim-steps 10] [--klass 1] [--opt key=value]"""
import argparse, collections, csv, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from prompt_diffusion_amd import engine as E, weights as W

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--ddim-steps", type=int, default=10)
ap.add_argument("--klass", type=int, default=-1)
ap.add_argument("--precision", default="f16")
ap.add_argument("--opt", action="append", default=[])
a = ap.parse_args()
cfg = W.SD15
e = E.Engine(cfg, precision=a.precision)
e.init_random_weights(3)
for kv in a.opt:
    k, v = kv.split("=")
    e.set_option(k, int(v))
inp = W.synth_inputs(cfg, a.batch, 64, 64)
kw = dict(x_T=inp["x_T"], ctx_cond=inp["ctx_cond"], ctx_uncond=inp["ctx_uncond"], pair=inp["pair"], query=inp["query"],
          steps=a.ddim_steps, cfg_scale=7.5)
e.ddim_sample(**kw)
e.set_option("two_streams", 0)
e.set_option("profile", 1)
e.ddim_sample(**kw)
path = "/tmp/shape_table.csv"
e.profile_dump(path)
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for r in csv.DictReader(open(path)):
    if a.klass >= 0 and int(r["klass"]) != a.klass:
        continue
    k = (int(r["klass"]), int(r["M"]), int(r["N"]), int(r["K"]), int(r["taps"]))
    agg[k][0] += 1; agg[k][1] += float(r["ms"]); agg[k][2] += float(r["flops"])
S = a.ddim_steps
tot = sum(v[1] for v in agg.values()) / S
print("total bracketed contraction time per step: %.2f ms" % tot)
for k, (n, ms, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print("klass %d M %6d N %5d K %5d taps %3d: %5.1f /step %7.1f us %6.2f ms/step (%4.1f%%) %6.0f TF/s" %
          (*k, n / S, 1e3 * ms / n, ms / S, 100 * ms / S / tot, fl / ms / 1e9))
