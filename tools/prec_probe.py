"""Error budget of the 2-byte engine modes against the reference fixtures (run on the GPU box).

For each fixture network (reduced net, SD1.5 config #1) prints, per engine mode, the eps error of one apply_model and the
per-step latent error of the whole DDIM trajectory (max-abs / max-abs, the metric of tests/test_network_gpu.py).  Also
isolates the weight-rounding share: the fp32 engine fed weights pre-rounded to fp16 / bf16."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from prompt_diffusion_amd import engine as E, weights as W  # noqa: E402

G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def relerr(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def rms(a, b):
    return float(np.sqrt(((a - b) ** 2).mean()) / np.sqrt((b ** 2).mean()))


def round_bf16(a):
    u = a.astype(np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


def run(cfg, g, prec, stream_f32=False, wround=None, opts=()):
    B, h, w, S = int(g["B"]), int(g["h"]), int(g["w"]), int(g["S"])
    inp = W.synth_inputs(cfg, B, h, w)
    x_in = np.concatenate([inp["x_T"]] * 2)
    t_in = np.full((2 * B,), int(g["first_step"]), dtype=np.int64)
    ctx = np.concatenate([inp["ctx_uncond"], inp["ctx_cond"]])
    pair = np.concatenate([inp["pair"]] * 2)
    qry = np.concatenate([inp["query"]] * 2)
    e = E.Engine(cfg, precision=prec, stream_f32=stream_f32)
    for k, v in opts:
        e.set_option(k, v)
    for n, a in W.iter_synth(cfg):
        if wround == "f16" and a.ndim >= 2:
            a = a.astype(np.float16).astype(np.float32)
        elif wround == "bf16" and a.ndim >= 2:
            a = round_bf16(a)
        e.load_tensor(n, a)
    eps = e.eps(x_in, t_in, ctx, pair, qry)
    ge = g["eps"]
    d = (eps[B:] - eps[:B]) - (ge[B:] - ge[:B])
    out, inter = e.ddim_sample(x_T=inp["x_T"], ctx_cond=inp["ctx_cond"], ctx_uncond=inp["ctx_uncond"], pair=inp["pair"],
                               query=inp["query"], steps=S, cfg_scale=float(g["cfg_scale"]), return_intermediates=True)
    e.close()
    steps = [relerr(inter[i], g["x_inter"][i]) for i in range(1, S + 1)]
    return dict(eps_max=relerr(eps, ge), eps_rms=rms(eps, ge), diff_rms=float(np.sqrt((d ** 2).mean()) / np.sqrt((ge ** 2).mean())),
                steps=steps, finite=bool(np.isfinite(out).all()))


if __name__ == "__main__":
    which = sys.argv[1:] or ["TINY", "SD15"]
    for cfgname, tag in (("TINY", "tiny_b2_16x16_s5"), ("SD15", "sd15_b1_32x32_s5")):
        if cfgname not in which:
            continue
        cfg = getattr(W, cfgname)
        g = np.load(os.path.join(G, f"net_{tag}.npz"))
        for label, kw in (("f32", dict(prec="f32")),
                          ("f32 + f16-rounded weights", dict(prec="f32", wround="f16")),
                          ("f32 + bf16-rounded weights", dict(prec="f32", wround="bf16")),
                          ("f16", dict(prec="f16")),
                          ("f16 stream_f32", dict(prec="f16", stream_f32=True)),
                          ("bf16", dict(prec="bf16")),
                          ("bf16 stream_f32", dict(prec="bf16", stream_f32=True))):
            r = run(cfg, g, **kw)
            print(f"{cfgname:5s} {label:28s} eps max {r['eps_max']:.2e} rms {r['eps_rms']:.2e} (e_c-e_u) {r['diff_rms']:.2e} | "
                  f"per-step max-rel {' '.join('%.2e' % v for v in r['steps'])} finite={r['finite']}", flush=True)
