#!/usr/bin/env python3
"""Time the SD3 / MMDiT path (pd_sd3_sample) at SD3-medium size with random-init weights.

    python tools/sd3_bench.py [--precision f16] [--batch 1] [--latent 128] [--ctx 333] [--steps 28] [--cn-layers 6]

Prints one JSON line: seconds per image, ms per denoising step, algorithmic TFLOP/s, launches per step and (with --profile)
the per-class contraction times from the engine's HIP-event brackets.  Not the headline metric (bench.py is): a measurement
of the §8f N4 row."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from prompt_diffusion_amd import sd3  # noqa: E402


def step_flops(cfg, B, N, S):
    """Algorithmic flops of one evaluation of transformer + ControlNet over B samples (2 * M * N * K per linear layer,
    4 * Nq * Nk * dh per attention head)."""
    D = cfg.hidden
    def net(layers, cn):
        f = 0.0
        for i in range(layers):
            pre = (not cn) and i == layers - 1
            f += 2.0 * B * N * D * (3 * D + D + 8 * D)                       # image stream: qkv, out, ff
            f += 2.0 * B * S * D * (3 * D) + (0 if pre else 2.0 * B * S * D * (D + 8 * D))
            nq = N if pre else N + S
            f += 4.0 * B * nq * (N + S) * D                                 # attention over the joint sequence
            f += 2.0 * B * D * (6 * D + (2 if pre else 6) * D)              # modulation vectors
            if cn:
                f += 2.0 * B * N * D * D                                    # controlnet_blocks[i]
        f += 2.0 * B * S * cfg.joint_dim * D                                # context_embedder
        return f
    return net(cfg.layers, False) + (net(cfg.cn_layers, True) if cfg.cn_layers else 0.0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--precision", default="f16", choices=["f16", "bf16", "f16x2", "f32"])
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--latent", type=int, default=128, help="latent side (128 = 1024 x 1024 pixels)")
    ap.add_argument("--ctx", type=int, default=333, help="context tokens (77 CLIP + 256 T5)")
    ap.add_argument("--steps", type=int, default=28)
    ap.add_argument("--cn-layers", type=int, default=6)
    ap.add_argument("--layers", type=int, default=24)
    ap.add_argument("--guidance", type=float, default=7.0)
    ap.add_argument("--repeat", type=int, default=2)
    ap.add_argument("--profile", action="store_true")
    ap.add_argument("--stream-f32", action="store_true")
    ap.add_argument("--fp8", type=int, nargs="?", const=2, default=0,
                    help="e4m3 operands: 1 = the AdaLN-fed projections, 2 (default when given) = also the feed-forward-out projections")
    ap.add_argument("--dump", default=None, help="with --profile: CSV of every profiled launch, and a per-shape table on stderr")
    ap.add_argument("--opt", action="append", default=[], help="engine option key=value (repeatable)")
    a = ap.parse_args()

    import torch
    cfg = sd3.SD3Config(layers=a.layers, cn_layers=a.cn_layers, pos_embed_max_size=max(96, a.latent // 2))
    eng = sd3.SD3Engine(cfg, precision=a.precision, stream_f32=a.stream_f32, fp8=a.fp8)
    eng.init_random_weights(7)
    for kv in a.opt:
        k, v = kv.split("=")
        eng.base.set_option(k, int(v))
    g = torch.Generator(device="cuda").manual_seed(0)
    f = lambda *s: torch.randn(*s, device="cuda", generator=g)
    B, H, S = a.batch, a.latent, a.ctx
    x, cond, pair = f(B, 16, H, H), f(B, 16, H, H), f(B, 16, H, H)
    ctx, nctx, pool, npool = f(B, S, cfg.joint_dim), f(B, S, cfg.joint_dim), f(B, cfg.pooled_dim), f(B, cfg.pooled_dim)
    kw = dict(control_latents=cond if a.cn_layers else None, pair_latents=pair if a.cn_layers else None,
              num_inference_steps=a.steps, guidance_scale=a.guidance)
    run = lambda: eng.sample(x, ctx, pool, nctx, npool, **kw)
    out = run()                                   # warm-up: workspace allocation, kernel attributes
    torch.cuda.synchronize()
    assert torch.isfinite(out).all()
    n0 = eng.base.stat("launches")
    t0 = time.perf_counter()
    for _ in range(a.repeat):
        out = run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.repeat
    launches = (eng.base.stat("launches") - n0) / a.repeat / a.steps
    Bf = 2 * B if a.guidance > 1 else B
    fl = step_flops(cfg, Bf, (H // 2) ** 2, S)
    rec = dict(metric="SD3-medium MMDiT + Prompt-Diffusion ControlNet, seconds per image", precision=a.precision + ("+fp8" + ("" if a.fp8 == 2 else "(level 1)") if a.fp8 else ""), batch=B,
               latent=[H, H], context_tokens=S, steps=a.steps, cn_layers=a.cn_layers, layers=a.layers, guidance=a.guidance,
               s_per_image=dt / B, images_per_s=B / dt, ms_per_step=1e3 * dt / a.steps, tflops_per_step=fl / 1e12,
               path_tflops_per_s=fl * a.steps / dt / 1e12, launches_per_step=launches,
               weight_gib=eng.base.stat("weight_bytes") / 2 ** 30, workspace_gib=eng.base.stat("workspace_bytes") / 2 ** 30,
               data="synthetic, random-init weights")
    if a.profile:
        eng.base.set_option("profile", 1)
        run()
        names = {1: "linear", 2: "attention"}
        rec["by_class"] = {}
        for k, nm in names.items():
            ms, n, flops = eng.base.profile_read(k)
            if n:
                rec["by_class"][nm] = dict(ms_per_step=ms / a.steps, launches_per_step=n / a.steps, tflops_per_s=flops / ms / 1e9)
        if a.dump:
            eng.base.profile_dump(a.dump)
            import collections, csv
            agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
            for r in csv.DictReader(open(a.dump)):
                k = (int(r["klass"]), int(r["M"]), int(r["N"]), int(r["K"]))
                agg[k][0] += 1; agg[k][1] += float(r["ms"]); agg[k][2] += float(r["flops"])
            for k, (n, ms, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
                print("klass %d M %6d N %6d K %5d: %5.1f launches/step %7.1f us  %6.2f ms/step  %6.0f TF/s" %
                      (*k, n / a.steps, 1e3 * ms / n, ms / a.steps, fl / ms / 1e9), file=sys.stderr)
        eng.base.set_option("profile", 0)
    print(json.dumps(rec))
    eng.close()


if __name__ == "__main__":
    main()
