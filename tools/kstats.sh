#!/bin/bash
# rocprofv3 kernel-trace stats of one bench.py configuration -> gpurun_out/kstats/<tag>_kernel_stats.csv + per-step table
set -e
TAG=${1:-cur}; shift || true
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/kstats; mkdir -p $O; rm -rf $O/raw_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d $O/raw_$TAG -- python bench.py --single-stream --steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-f32 --no-parity --no-sd3 "$@" > $O/$TAG.log 2>&1
cp $(find $O/raw_$TAG -name "*kernel_stats.csv" | head -1) $O/${TAG}_kernel_stats.csv
rm -rf $O/raw_$TAG
python tools/kstats_table.py $O/${TAG}_kernel_stats.csv 100
