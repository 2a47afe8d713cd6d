"""Trace-derived roofline figures of the contraction kernel families: call-weighted average launch duration from a rocprofv3
kernel_stats.csv and the algorithmic FLOPs per launch that bench.py recorded in the SAME run (its profiled pass).
usage: python tools/roofline_from_trace.py <kernel_stats.csv> <bench_under_rocprof.json> <out.json>
bench.py quotes the newest profiles/*_roofline.json next to its live HIP-event figure (roofline.trace)."""
import csv, json, sys

rows = list(csv.DictReader(open(sys.argv[1])))
bench = json.load(open(sys.argv[2]))
rf = bench["roofline"]
PEAK = 2500.0
fam = {
    "igemm_kernel + rgemm_kernel": (("igemm_kernel<", "rgemm_kernel<"), ("igemm_kernel[conv3x3]", "igemm+rgemm_kernel[linear/conv1x1]")),
    "conv3x3 patch kernels": (("conv3x3_patch_kernel<", "conv3x3_patch2_kernel<", "conv3x3_pp_kernel<", "conv3x3_w4_kernel<"), ("conv3x3_patch_kernel",)),
    "attention": (("attn2_kernel<", "attn_kernel<"), ("attn_kernel",)),
    "st_tail + st_front": (("st_tail_kernel<", "st_front_kernel<"), ("st_tail_kernel",)),
}
out = {"source": "rocprofv3 --kernel-trace --stats -- python bench.py --single-stream --steps 2 --warmup 1 (same run as the bench line in " + sys.argv[2] + ")",
       "definition": "avg_us = sum(TotalDurationNs) / sum(Calls) over the family's rows; tflops = bench.py's algorithmic FLOPs per launch / avg_us",
       "families": {}}
for name, (pats, classes) in fam.items():
    sel = [r for r in rows if any(p in r["Name"] for p in pats)]
    calls = sum(int(r["Calls"]) for r in sel)
    ns = sum(float(r["TotalDurationNs"]) for r in sel)
    if not calls:
        continue
    by = rf["by_kernel"]
    flops = sum(by[c]["tflops"] * by[c]["ms"] * 1e9 for c in classes if c in by)    # per profiled pass
    launches = sum(by[c]["launches"] for c in classes if c in by)
    avg_us = ns / calls / 1e3
    ev_us = sum(by[c]["ms"] for c in classes if c in by) * 1e3 / max(launches, 1)
    tf = flops / max(launches, 1) / avg_us / 1e6
    out["families"][name] = {"trace_calls": calls, "trace_avg_us": avg_us, "flop_per_launch": flops / max(launches, 1), "tflops": tf, "frac_of_2500": tf / PEAK,
                             "hip_event_avg_us_same_run": ev_us, "event_over_trace": ev_us / avg_us, "launches_per_pass": launches}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out["families"], indent=1))
