#!/bin/bash
# attention kernel alone: kernel-trace timing + SQ / GRBM counters (separate rocprofv3 passes) -> gpurun_out/attn/<round>_attn_pmc_N<tokens>.json
#   usage: tools/attn_measure.sh [precision] [tokens] [round]
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
PREC=${1:-f16}; N=${2:-4096}; R=${3:-r03}
O=gpurun_out/attn; mkdir -p $O; rm -rf $O/stats $O/pmc1 $O/pmc2
export ATTN_N=$N
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python tools/attn_bench.py 5 $PREC $N > $O/stats.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc1 -- python tools/attn_bench.py 2 $PREC $N > $O/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/pmc2 -- python tools/attn_bench.py 2 $PREC $N > $O/pmc2.log 2>&1
python tools/attn_pmc.py $O/stats $O/pmc1 $O/pmc2 $O/${R}_attn_pmc_N$N.json > $O/summary_N$N.log
rm -rf $O/stats $O/pmc1 $O/pmc2
grep -E "avg_us|tflops|share|mfma_util|clock_mhz|valu_per_mfma" $O/summary_N$N.log
