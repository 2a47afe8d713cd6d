#!/bin/bash
# rocprofv3 kernel-trace stats of tools/sd3_bench.py (SD3-medium, 10 steps, repeat 1 + warm-up = 20 profiled steps)
# -> gpurun_out/kstats/sd3_<tag>_kernel_stats.csv + per-step table
set -e
TAG=${1:-cur}; shift || true
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/kstats; mkdir -p $O; rm -rf $O/raw_sd3_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d $O/raw_sd3_$TAG -- python tools/sd3_bench.py --steps 10 --repeat 1 "$@" > $O/sd3_$TAG.log 2>&1
cp $(find $O/raw_sd3_$TAG -name "*kernel_stats.csv" | head -1) $O/sd3_${TAG}_kernel_stats.csv
rm -rf $O/raw_sd3_$TAG
python tools/kstats_table.py $O/sd3_${TAG}_kernel_stats.csv 20
