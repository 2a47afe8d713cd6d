#!/bin/bash
# Measurements of the SD3 path (§8f N4) for profiles/: bench lines (f16, f16 + fp8 option), per-shape launch table, rocprofv3
# kernel-trace stats (single stream, so that a kernel duration is its own).  Usage (on the GPU box): bash tools/sd3_measure.sh r02   -> gpurun_out/sd3/<round>_sd3_*
set -e
R=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/sd3; mkdir -p $O
python tools/sd3_bench.py --steps 28 --profile --dump $O/${R}_sd3_launches_f16.csv > $O/${R}_sd3_bench_f16.json 2> $O/${R}_sd3_shapes_f16.txt
python tools/sd3_bench.py --steps 28 --fp8 --profile --dump $O/${R}_sd3_launches_fp8.csv > $O/${R}_sd3_bench_fp8.json 2> $O/${R}_sd3_shapes_fp8.txt
python tools/sd3_bench.py --steps 28 --batch 2 > $O/${R}_sd3_bench_f16_bs2.json 2>/dev/null
python tools/sd3_bench.py --steps 28 --batch 2 --fp8 > $O/${R}_sd3_bench_fp8_bs2.json 2>/dev/null
rm -f $O/${R}_sd3_launches_f16.csv $O/${R}_sd3_launches_fp8.csv
for m in f16 fp8; do grep klass $O/${R}_sd3_shapes_$m.txt > $O/${R}_sd3_shapes_$m.tmp; mv $O/${R}_sd3_shapes_$m.tmp $O/${R}_sd3_shapes_$m.txt; done
rm -rf $O/raw
rocprofv3 --kernel-trace --stats --output-format csv -d $O/raw -- python tools/sd3_bench.py --steps 10 --repeat 1 --fp8 --opt two_streams=0 > $O/prof.log 2>&1
cp $(find $O/raw -name "*kernel_stats.csv" | head -1) $O/${R}_sd3_fp8_kernel_stats.csv
rm -rf $O/raw
python tools/kstats_table.py $O/${R}_sd3_fp8_kernel_stats.csv 20 > $O/${R}_sd3_fp8_kernel_table.txt
cat $O/${R}_sd3_bench_f16.json $O/${R}_sd3_bench_fp8.json $O/${R}_sd3_bench_f16_bs2.json $O/${R}_sd3_bench_fp8_bs2.json | cut -c1-420
head -12 $O/${R}_sd3_fp8_kernel_table.txt
