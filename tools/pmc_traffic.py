"""Turns two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; each `--kernel-trace --pmc <counter> --output-format csv`) of
a bench.py run into profiles/rNN_pmc_traffic.json: HBM bytes per launch, per kernel family.
gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE counts 64 B per 128-byte request -> doubled; both
counters are in KB.  usage: python tools/pmc_traffic.py <fetch_dir> <write_dir> <out.json> "<command that was profiled>" """
import collections, csv, glob, json, sys

FAMILIES = ("igemm_kernel<", "rgemm_kernel<", "conv3x3_patch_kernel<", "conv3x3_patch2_kernel<", "conv3x3_pp_kernel<", "conv3x3_w4_kernel<", "attn2_kernel", "gn_apply_kernel<", "gn_stats_kernel<",
            "gn_fused_kernel<", "layernorm_kernel<", "splitk_finalize_kernel", "concat_add_kernel", "st_tail_kernel<", "st_front_kernel<")


def collect(d, counter):
    tot, n = collections.Counter(), collections.Counter()
    for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] != counter:
                continue
            for f in FAMILIES:
                if f in r["Kernel_Name"]:
                    tot[f] += float(r["Counter_Value"])
                    n[f] += 1
    return tot, n


fetch, nf = collect(sys.argv[1], "FETCH_SIZE")
write, nw = collect(sys.argv[2], "WRITE_SIZE")
out = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) -- " + sys.argv[4],
       "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request -> doubled (MI355X_MICROARCH.md, HBM section); WRITE_SIZE exact; both in KB",
       "by_kernel": {}}
for f in FAMILIES:
    if not nf[f]:
        continue
    name = f.split("<")[0]
    fk, wk = fetch[f] / nf[f], write[f] / max(nw[f], 1)
    out["by_kernel"][name] = {"launches": nf[f], "fetch_size_kb_raw": fk, "write_size_kb": wk,
                              "hbm_bytes_per_launch": (2 * fk + wk) * 1024}
# the family bench.py's roofline object is about: every launch of the generic contraction kernels -- igemm_kernel (gemm.hip) and
# rgemm_kernel (gemm_ring.hip, round 3: the short-K linear layers) -- launch-weighted
gk = [out["by_kernel"][k] for k in ("igemm_kernel", "rgemm_kernel") if k in out["by_kernel"]]
out["igemm_kernel_hbm_bytes_per_launch"] = sum(k["hbm_bytes_per_launch"] * k["launches"] for k in gk) / sum(k["launches"] for k in gk)
out["igemm_kernel_hbm_bytes_per_launch_covers"] = "igemm_kernel + rgemm_kernel, launch-weighted"
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out["by_kernel"], indent=1))
