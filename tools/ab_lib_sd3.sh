#!/bin/bash
# same-box A/B of two library builds on the SD3 bench: tools/ab_lib_sd3.sh <other.so> [rounds] [extra sd3_bench args]
O=$1; R=${2:-2}; shift; shift
for r in $(seq 1 $R); do
  for v in new old; do
    if [ $v = old ]; then export PDENGINE_LIB=$O; else unset PDENGINE_LIB; fi
    timeout -k 10 300 python tools/sd3_bench.py --steps 10 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],2), 'ms/step')"
  done
done
