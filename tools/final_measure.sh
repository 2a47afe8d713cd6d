#!/bin/bash
# Round-end measurement on the GPU box: kernel-trace stats of the bench command, the two HBM-traffic PMC passes on the SAME
# 50-step workload, the attention kernel's SQ counters, the N = 2 control-flow rehearsal, and the default bench line.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=${1:-r02}
O=gpurun_out/final; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python bench.py --single-stream --steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-sd3 --no-f32 > $O/bench_under_rocprof.log 2>&1
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/${R}_bench_kernel_stats.csv
grep -a '^{' $O/bench_under_rocprof.log | tail -1 > $O/${R}_bench_under_rocprof.json
rm -rf $O/stats
echo "[final] kernel stats done"
# counters only for the contraction and attention kernels, on a 20-step sampling of the SAME workload (identical per-step
# launches; 20 divides 1000 like 50 does): rocprofv3 segfaults beyond roughly 10 000 profiled dispatches (50 steps: 13 000)
KRE="igemm_kernel|conv3x3_patch|attn2_kernel"
PMC_CMD="python bench.py --single-stream --steps 1 --warmup 0 --ddim-steps 20 --no-cpu-baseline --no-profile --no-f32 --no-parity --no-sd3"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --kernel-include-regex "$KRE" --output-format csv -d $O/fetch -- $PMC_CMD > $O/fetch.log 2>&1
echo "[final] FETCH_SIZE pass done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --kernel-include-regex "$KRE" --output-format csv -d $O/write -- $PMC_CMD > $O/write.log 2>&1
echo "[final] WRITE_SIZE pass done"
python tools/pmc_traffic.py $O/fetch $O/write $O/${R}_pmc_traffic.json "$PMC_CMD" > $O/pmc_traffic.log
rm -rf $O/fetch $O/write      # raw traces are large; the summaries above are what gets committed
bash tools/attn_measure.sh f16 > $O/attn.log 2>&1 && cp gpurun_out/attn/r02_attn_pmc_f16.json $O/${R}_attn_pmc.json
echo "[final] attention counters done"
PD_BENCH_REHEARSE=1 python bench.py --gpus 2 --steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-f32 --no-parity --no-sd3 > $O/${R}_bench_rehearse_gpus2.log 2>&1
echo "[final] --gpus 2 rehearsal done"
python bench.py > $O/bench_default.log 2>&1
tail -1 $O/bench_default.log > $O/${R}_bench_default.json
head -c 700 $O/${R}_bench_default.json
