#!/bin/bash
# hipGraph replay of the step loop vs eager launches at several batch sizes (same box)
for b in 1 2 8; do for g in 0 1; do
  timeout -k 10 400 python bench.py --no-cpu-baseline --no-f32 --no-profile --no-sd3 --steps 2 --warmup 1 --batch $b --opt graph=$g > gpurun_out/g.log 2>&1 || { echo "FAILED b=$b g=$g"; tail -3 gpurun_out/g.log; exit 1; }
  python - "$b" "$g" <<'PY'
import json,sys
d=json.loads(open("gpurun_out/g.log").read().strip().split("\n")[-1])
print("batch",sys.argv[1],"graph",sys.argv[2], round(d["value"],3), "img/s", round(d["ms_per_step"],1), "ms")
PY
done; done
