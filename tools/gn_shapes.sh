#!/bin/bash
# per-shape durations of the GroupNorm / finalize / concat / LayerNorm helper kernels: tools/gn_shapes.sh  -> gpurun_out/gn_shapes.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/gnsh; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O -- python bench.py --single-stream --steps 1 --warmup 0 --ddim-steps 10 --no-cpu-baseline --no-parity --no-sd3 --no-f32 --no-profile > $O/bench.log 2>&1
python - "$(find $O -name '*kernel_trace.csv' | head -1)" <<'PY' > gpurun_out/gn_shapes.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: [0, 0.0])
hist = collections.Counter()
for r in rows:
    n = r["Kernel_Name"]
    if not any(k in n for k in ("gn_", "splitk_finalize", "concat_add", "layernorm", "row_stats")):
        continue
    name = n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    key = (name, int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1), int(r["Workgroup_Size_X"]))
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    agg[key][0] += 1
    agg[key][1] += d
steps = 10.0
tot = 0.0
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{k[0][:40]:40s} blocks {k[1]:6d} x {k[2]:4d} thr  calls/step {c / steps:6.1f}  avg_us {t / c:7.1f}  ms/step {t / steps / 1e3:6.3f}")
    tot += t
print(f"total {tot / steps / 1e3:.3f} ms/step")
# duration histogram of the single-kernel GroupNorm launches (the 16x16 and 8x8 levels share a grid size)
for r in rows:
    n = r["Kernel_Name"]
    if "gn_fused" in n:
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        hist[("slab" if "true>" in n else "plain", int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(d))] += 1
for k in sorted(hist): print("gn_fused", k, hist[k])
PY
rm -rf $O
tail -60 gpurun_out/gn_shapes.txt
