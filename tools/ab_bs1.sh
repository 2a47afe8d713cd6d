for spec in "$@"; do
  name=${spec%%:*}; opts=${spec#*:}; args=""
  if [ "$opts" != "$spec" ] && [ -n "$opts" ]; then for o in ${opts//,/ }; do args="$args --opt $o"; done; fi
  timeout -k 10 300 python bench.py --batch 1 --no-cpu-baseline --no-f32 --no-profile --no-sd3 --no-parity --steps 3 $args > gpurun_out/ab1_$name.log 2>&1 || { echo "$name FAILED"; tail -3 gpurun_out/ab1_$name.log; exit 1; }
  python - "$name" <<'PY'
import json,sys
d=json.loads(open(f"gpurun_out/ab1_{sys.argv[1]}.log").read().strip().split("\n")[-1])
print(sys.argv[1], round(d["value"],3), "img/s", round(d["ms_per_step"],1), "ms", flush=True)
PY
done
