#!/bin/bash
# same-box A/B of several library builds: tools/ab_libs.sh <rounds> name=path.so ...   ("cur" = the in-tree build)
R=$1; shift
for r in $(seq 1 $R); do
  for spec in cur=cur "$@"; do
    name=${spec%%=*}; path=${spec#*=}
    if [ "$path" = cur ]; then unset PDENGINE_LIB; else export PDENGINE_LIB=$path; fi
    timeout -k 10 300 python bench.py --no-cpu-baseline --no-f32 --no-profile --no-sd3 --steps 2 > gpurun_out/abl_$name.log 2>&1 || { echo "$name FAILED"; tail -3 gpurun_out/abl_$name.log; exit 1; }
    python - "$name" <<'PY'
import json,sys
d=json.loads(open(f"gpurun_out/abl_{sys.argv[1]}.log").read().strip().split("\n")[-1])
print(sys.argv[1], round(d["value"],3), "img/s", round(d["ms_per_step"],1), "ms", flush=True)
PY
  done
done
