#!/bin/bash
# per-step kernel table of one bench configuration: tools/kstats.sh <name> [bench args...]   -> gpurun_out/kstats_<name>.{csv,txt}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
N=$1; shift
O=gpurun_out/ks_$N; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python bench.py --single-stream --steps 1 --warmup 1 --no-cpu-baseline --no-parity --no-sd3 --no-f32 --no-profile "$@" > $O/bench.log 2>&1
cp $(find $O -name "*kernel_stats.csv" | head -1) gpurun_out/kstats_$N.csv
grep -a '^{' $O/bench.log | tail -1 > gpurun_out/kstats_$N.json
rm -rf $O
python tools/kstats_table.py gpurun_out/kstats_$N.csv 100 > gpurun_out/kstats_$N.txt
head -34 gpurun_out/kstats_$N.txt
python -c "
import json;d=json.load(open('gpurun_out/kstats_$N.json'));print('bench under rocprof:',round(d['value'],3),'img/s',round(d['ms_per_step']/50,2),'ms per DDIM step (wall)')"
