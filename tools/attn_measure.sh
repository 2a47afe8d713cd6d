#!/bin/bash
# attention kernel alone: kernel-trace timing + SQ counters (separate rocprofv3 passes) -> gpurun_out/attn/r02_attn_pmc.json
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/attn; rm -rf $O; mkdir -p $O
PREC=${1:-f16}
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python tools/attn_bench.py 5 $PREC > $O/stats.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/pmc1 -- python tools/attn_bench.py 2 $PREC > $O/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/pmc2 -- python tools/attn_bench.py 2 $PREC > $O/pmc2.log 2>&1
python tools/attn_pmc.py $O/stats $O/pmc1 $O/pmc2 $O/r02_attn_pmc_$PREC.json > $O/summary.log
rm -rf $O/stats $O/pmc1 $O/pmc2
grep -E "avg_us|tflops|share|mfma_busy" $O/summary.log
