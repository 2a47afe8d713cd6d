"""Teacher-forced single-step latent error of the bf16 engine at the 50-step schedule (first and last step)."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from prompt_diffusion_amd import engine as E, weights as W
from oracle import pd_oracle as O
cfg = W.SD15; B, h, w, S = 1, 32, 32, 50
inp = W.synth_inputs(cfg, B, h, w); sd = W.synth_state_dict(cfg); lay = O.make_layouts(cfg, W)
cond = dict(c_crossattn=inp["ctx_cond"], example_pair=inp["pair"], query=inp["query"]); unc = dict(c_crossattn=inp["ctx_uncond"], example_pair=inp["pair"], query=inp["query"])
sched = O.make_schedule(S); tr = np.flip(sched["ddim_timesteps"])
def rel(a, b): return float(np.abs(a - b).max() / np.abs(b).max())
for prec, sf in (("bf16", False), ("bf16", True), ("f32", False)):
    e = E.Engine(cfg, precision=prec, stream_f32=sf); e.load_state_dict(sd)
    n = e.sample_begin(x_T=inp["x_T"], ctx_cond=inp["ctx_cond"], ctx_uncond=inp["ctx_uncond"], pair=inp["pair"], query=inp["query"], steps=S, cfg_scale=7.5)
    e.sample_step(0); x1 = e.sample_get(); eps = e.sample_get(E.PD_GET_EPS)
    if prec == "bf16" and not sf:
        t0 = time.time(); ref, _, ref_e = O.p_sample_ddim(sd, cfg, lay, sched, inp["x_T"], cond, unc, S - 1, int(tr[0]), 7.5); print("oracle step s", time.time() - t0)
    print(prec, "stream_f32", sf, "step0 (t=%d): latent max-rel %.3e  guided-eps max-rel %.3e" % (tr[0], rel(x1, ref), rel(eps, ref_e)))
    e.sample_end(); e.close()
