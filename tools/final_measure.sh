#!/bin/bash
# Round-end measurement on the GPU box: kernel-trace stats of the bench command + the two PMC passes.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/final; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python bench.py --single-stream --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_under_rocprof.log 2>&1
PMC_CMD="python bench.py --single-stream --steps 1 --warmup 0 --ddim-steps 5 --no-cpu-baseline --no-profile --no-f32"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- $PMC_CMD > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- $PMC_CMD > $O/write.log 2>&1
python tools/pmc_traffic.py $O/fetch $O/write $O/pmc_traffic.json "$PMC_CMD" > $O/pmc_traffic.log
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
grep -a '^{' $O/bench_under_rocprof.log | tail -1 > $O/bench_under_rocprof.json
rm -rf $O/fetch $O/write $O/stats      # raw traces are large; the summaries above are what gets committed
python bench.py > $O/bench_default.log 2>&1
tail -1 $O/bench_default.log | head -c 600
