#!/bin/bash
# Round-end measurement on the GPU box: kernel-trace stats of the bench command, the HBM-traffic and MFMA-utilisation counter passes
# on the SAME workload (20 DDIM steps, contraction / attention / fused-transformer kernels only: see profiles/README.md for why),
# the attention kernel's counters at N = 4096 and N = 9216, the N = 2 control-flow rehearsal, and the default bench line.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=${1:-r04}
O=gpurun_out/final; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python bench.py --single-stream --steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-sd3 --no-f32 > $O/bench_under_rocprof.log 2>&1
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/${R}_bench_kernel_stats.csv
grep -a '^{' $O/bench_under_rocprof.log | tail -1 > $O/${R}_bench_under_rocprof.json
rm -rf $O/stats
python tools/kstats_table.py $O/${R}_bench_kernel_stats.csv 200 > $O/${R}_bench_kernel_table.txt
python tools/roofline_from_trace.py $O/${R}_bench_kernel_stats.csv $O/${R}_bench_under_rocprof.json $O/${R}_roofline.json > $O/roofline.log
echo "[final] kernel stats done"
KRE="igemm_kernel|rgemm_kernel|conv3x3_|attn2_kernel|st_tail_kernel|st_front_kernel"
PMC_CMD="python bench.py --single-stream --steps 1 --warmup 0 --ddim-steps 20 --no-cpu-baseline --no-profile --no-f32 --no-parity --no-sd3"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --kernel-include-regex "$KRE" --output-format csv -d $O/fetch -- $PMC_CMD > $O/fetch.log 2>&1
echo "[final] FETCH_SIZE pass done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --kernel-include-regex "$KRE" --output-format csv -d $O/write -- $PMC_CMD > $O/write.log 2>&1
echo "[final] WRITE_SIZE pass done"
python tools/pmc_traffic.py $O/fetch $O/write $O/${R}_pmc_traffic.json "$PMC_CMD" > $O/pmc_traffic.log
rm -rf $O/fetch $O/write      # raw traces are large; the summaries above are what gets committed
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-include-regex "$KRE" --output-format csv -d $O/util -- $PMC_CMD > $O/util.log 2>&1
python tools/pmc_util.py $O/util $O/${R}_pmc_mfma_util.json "$PMC_CMD" > $O/pmc_util.log
rm -rf $O/util
echo "[final] MFMA utilisation pass done"
bash tools/attn_measure.sh f16 4096 $R > $O/attn4096.log 2>&1 && cp gpurun_out/attn/${R}_attn_pmc_N4096.json $O/
bash tools/attn_measure.sh f16 9216 $R > $O/attn9216.log 2>&1 && cp gpurun_out/attn/${R}_attn_pmc_N9216.json $O/
echo "[final] attention counters done"
PD_BENCH_REHEARSE=1 python bench.py --gpus 2 --steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-f32 --no-parity --no-sd3 > $O/${R}_bench_rehearse_gpus2.log 2>&1
echo "[final] --gpus 2 rehearsal done"
python bench.py --batch 1 --steps 3 --no-cpu-baseline --no-profile --no-f32 --no-parity --no-sd3 > $O/bench_bs1.log 2>&1
tail -1 $O/bench_bs1.log > $O/${R}_bench_bs1.json
cp $O/${R}_pmc_traffic.json $O/${R}_pmc_mfma_util.json $O/${R}_roofline.json profiles/   # the default line below quotes the counters measured in THIS call
python bench.py > $O/bench_default.log 2>&1
tail -1 $O/bench_default.log > $O/${R}_bench_default.json
head -c 700 $O/${R}_bench_default.json
