import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from prompt_diffusion_amd import engine as E, weights as W
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
def relerr(a, b): return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))
def rms(a, b): return float(np.sqrt(((a - b) ** 2).mean()) / np.sqrt((b ** 2).mean()))
for cfgname, tag in (("TINY", "tiny_b2_16x16_s5"), ("SD15", "sd15_b1_32x32_s5")):
    cfg = getattr(W, cfgname)
    g = np.load(os.path.join(G, f"net_{tag}.npz"))
    B, h, w, S = int(g["B"]), int(g["h"]), int(g["w"]), int(g["S"])
    inp = W.synth_inputs(cfg, B, h, w)
    x_in = np.concatenate([inp["x_T"]] * 2); t_in = np.full((2 * B,), int(g["first_step"]), dtype=np.int64)
    ctx = np.concatenate([inp["ctx_uncond"], inp["ctx_cond"]]); pair = np.concatenate([inp["pair"]] * 2); qry = np.concatenate([inp["query"]] * 2)
    for sf in (False, True):
        e = E.Engine(cfg, precision="bf16", stream_f32=sf)
        for n, a in W.iter_synth(cfg): e.load_tensor(n, a)
        eps, ctl = e.eps(x_in, t_in, ctx, pair, qry, return_control=True)
        ge = g["eps"]
        d = (eps[B:] - eps[:B]) - (ge[B:] - ge[:B])
        print(cfgname, "stream_f32", sf, "eps max-rel %.3e rms-rel %.3e | (e_c-e_u) err rms / eps rms %.3e" % (relerr(eps, ge), rms(eps, ge), np.sqrt((d**2).mean()) / np.sqrt((ge**2).mean())))
        out, inter = e.ddim_sample(x_T=inp["x_T"], ctx_cond=inp["ctx_cond"], ctx_uncond=inp["ctx_uncond"], pair=inp["pair"], query=inp["query"], steps=S, cfg_scale=float(g["cfg_scale"]), return_intermediates=True)
        print("   per-step max-rel", ["%.2e" % relerr(inter[i], g["x_inter"][i]) for i in range(S + 1)], " rms-rel", ["%.2e" % rms(inter[i], g["x_inter"][i]) for i in range(1, S + 1)])
        e.close()
e = E.Engine(W.TINY, precision="f32")
gs = np.load(os.path.join(G, "schedule.npz"))
for S, eta in ((5, 0.0), (50, 0.0), (20, 0.0), (50, 0.5), (10, 1.0)):
    s = e.make_schedule(S, eta); tag = f"S{S}_eta{eta}"
    for k in ("ddim_alphas", "ddim_alphas_prev", "ddim_sigmas", "ddim_sqrt_one_minus_alphas"):
        d = np.abs(s[k] - gs[f"{tag}_{k}"]) / (np.abs(gs[f"{tag}_{k}"]) + 1e-30)
        if d.max() > 0: print(tag, k, "max rel diff %.3e at %d" % (d.max(), d.argmax()), s[k][d.argmax()], gs[f"{tag}_{k}"][d.argmax()])
    print(tag, "timesteps equal", np.array_equal(s["ddim_timesteps"], gs[tag + "_timesteps"]))
