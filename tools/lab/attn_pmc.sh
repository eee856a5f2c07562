#!/bin/bash
# rocprofv3 PMC passes over tools/lab/attn_lab (counters in their own runs, kernel-trace only — see the gpurun rules)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
TAG=${1:-attn}
shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ap1 /tmp/ap2 /tmp/ap3
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/ap1 -- $R/tools/lab/attn_lab 1 "$@" > $O/${TAG}_pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM --output-format csv -d /tmp/ap2 -- $R/tools/lab/attn_lab 1 "$@" > $O/${TAG}_pmc2.log 2>&1
cd $R
python3 tools/lab/pmc_table.py /tmp/ap1 /tmp/ap2 > $O/${TAG}_pmc_table.txt 2>&1
cat $O/${TAG}_pmc_table.txt
python3 tools/lab/attn_pmc_derive.py /tmp/ap1 /tmp/ap2 > $O/${TAG}_pmc_derived.txt 2>&1
cat $O/${TAG}_pmc_derived.txt
