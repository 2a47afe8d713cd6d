"""Summarises rocprofv3 runs of tools/attn_bench.py into profiles/r02_attn_pmc.json.
usage: python tools/attn_pmc.py <stats_dir> <pmc_dir_1> [<pmc_dir_2> ...] <out.json>
Per attention kernel: average duration (kernel trace), algorithmic TFLOP/s at the 64x64-level self-attention shape
(B 16, 8 heads, N 4096, dh 40), and the SQ counters averaged per launch: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE (share of
LDS cycles lost to bank conflicts), SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES, SQ_INSTS_VALU, SQ_INSTS_MFMA, SQ_WAVE_CYCLES."""
import collections, csv, glob, json, sys

import os
stats_dir, pmc_dirs, out = sys.argv[1], sys.argv[2:-1], sys.argv[-1]
NTOK = int(os.environ.get("ATTN_N", "4096"))
FLOPS = 4.0 * 16 * 8 * NTOK * NTOK * 40
res = {"shape": f"B 16 x 8 heads, Nq = Nk = {NTOK}, dh 40 (self-attention of the 64x64 level at 512x512: N 4096; at 768x768, BASELINE config #4: N 9216)",
       "algorithmic_flop_per_launch": FLOPS, "kernels": {}}
for path in glob.glob(stats_dir + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        if "attn" in r["Name"]:
            name = r["Name"].split("(anonymous namespace)::")[-1].split("(")[0]
            res["kernels"].setdefault(name, {})
            res["kernels"][name].update(calls=int(r["Calls"]), avg_us=float(r["AverageNs"]) / 1e3,
                                        tflops=FLOPS / float(r["AverageNs"]) / 1e3)
for d in pmc_dirs:
    tot, n = collections.defaultdict(float), collections.Counter()
    for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            if "attn" not in r["Kernel_Name"]:
                continue
            name = r["Kernel_Name"].split("(anonymous namespace)::")[-1].split("(")[0]
            tot[(name, r["Counter_Name"])] += float(r["Counter_Value"])
            n[(name, r["Counter_Name"])] += 1
    for (name, c), v in tot.items():
        res["kernels"].setdefault(name, {}).setdefault("counters_per_launch", {})[c] = v / n[(name, c)]
for name, k in res["kernels"].items():
    c = k.get("counters_per_launch", {})
    if c.get("SQ_LDS_IDX_ACTIVE"):
        k["lds_bank_conflict_share"] = c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"]
    if c.get("SQ_BUSY_CYCLES") and "SQ_VALU_MFMA_BUSY_CYCLES" in c:
        k["mfma_busy_over_sq_busy"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / c["SQ_BUSY_CYCLES"]
    if c.get("GRBM_GUI_ACTIVE") and "SQ_VALU_MFMA_BUSY_CYCLES" in c:
        # MFMA-pipe busy cycles over (1024 SIMDs x shader cycles of the dispatch); GRBM_GUI_ACTIVE is summed over the 8 XCDs
        k["mfma_util"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * c["GRBM_GUI_ACTIVE"] / 8.0)
        k["clock_mhz"] = c["GRBM_GUI_ACTIVE"] / 8.0 / k["avg_us"] if k.get("avg_us") else None
        if c.get("SQ_INSTS_MFMA"):
            k["valu_per_mfma"] = c.get("SQ_INSTS_VALU", 0.0) / c["SQ_INSTS_MFMA"]
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
