#!/bin/bash
# SQ counter passes of one command for one kernel family: tools/pmc.sh <kernel-filter> <command...>
set -e
F=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $O/p1 -- "$@" > $O/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD --output-format csv -d $O/p2 -- "$@" > $O/p2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $O/p3 -- "$@" > $O/p3.log 2>&1 || true
python tools/pmc_kernel.py "$F" $O/p1 $O/p2 $O/p3 | tee $O/summary.txt
rm -rf $O/p1 $O/p2 $O/p3
