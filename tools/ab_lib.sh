#!/bin/bash
# same-box A/B of two library builds: tools/ab_lib.sh <other.so> [rounds]
for r in $(seq 1 ${2:-2}); do
  for v in new old; do
    if [ $v = old ]; then export PDENGINE_LIB=$1; else unset PDENGINE_LIB; fi
    timeout -k 10 300 python bench.py --no-cpu-baseline --no-f32 --no-profile --no-sd3 --steps 2 > gpurun_out/abl_$v.log 2>&1 || { echo "$v FAILED"; tail -3 gpurun_out/abl_$v.log; exit 1; }
    python - "$v" <<'PY'
import json,sys
d=json.loads(open(f"gpurun_out/abl_{sys.argv[1]}.log").read().strip().split("\n")[-1])
print(sys.argv[1], round(d["value"],3), "img/s", round(d["ms_per_step"],1), "ms")
PY
  done
done
