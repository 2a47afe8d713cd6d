"""MFMA-pipe utilisation per kernel family from one rocprofv3 counter pass over a bench.py run:
    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-include-regex ... --output-format csv
mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x shader cycles of the dispatch), shader cycles = GRBM_GUI_ACTIVE / 8 (rocprofv3
sums the counter over the 8 XCDs; MI355X_MICROARCH.md, DVFS give-back) -- a ratio of cycle counts, so the clock the chip held
cancels; it counts every MFMA issued, padded head dims included.  The clock itself (GRBM cycles / kernel-trace duration) reads
high on dispatches shorter than ~0.3 ms, so it is reported only for context.
usage: python tools/pmc_util.py <pmc_dir> <out.json> "<command that was profiled>" """
import collections, csv, glob, json, sys

FAMILIES = ("igemm_kernel<", "rgemm_kernel<", "conv3x3_patch_kernel<", "conv3x3_patch2_kernel<", "conv3x3_pp_kernel<", "conv3x3_w4_kernel<", "attn2_kernel", "attn_kernel<", "st_tail_kernel<", "st_front_kernel<")
tot = collections.defaultdict(float)
n = collections.Counter()
dur = collections.defaultdict(float)
for path in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        for f in FAMILIES:
            if f in r["Kernel_Name"]:
                tot[(f, r["Counter_Name"])] += float(r["Counter_Value"])
                n[(f, r["Counter_Name"])] += 1
for path in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        for f in FAMILIES:
            if f in r["Kernel_Name"]:
                dur[f] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
                n[(f, "trace")] += 1
out = {"source": "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -- " + sys.argv[3],
       "definition": "mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * GRBM_GUI_ACTIVE / 8); padded MFMA work included", "by_kernel": {}}
for f in FAMILIES:
    k = (f, "SQ_VALU_MFMA_BUSY_CYCLES")
    if not n[k]:
        continue
    busy = tot[k] / n[k]
    gui = tot[(f, "GRBM_GUI_ACTIVE")] / max(n[(f, "GRBM_GUI_ACTIVE")], 1) / 8.0
    d_us = dur[f] / max(n[(f, "trace")], 1) / 1e3
    out["by_kernel"][f.split("<")[0]] = {"launches": n[k], "mfma_busy_cycles_per_launch": busy, "shader_cycles_per_launch": gui,
                                         "mfma_util": busy / (1024.0 * gui) if gui else None, "avg_us_under_counters": d_us,
                                         "clock_mhz_grbm_over_duration": gui / d_us if d_us else None}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out["by_kernel"], indent=1))
