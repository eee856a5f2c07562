#!/bin/bash
# MFMA-busy PMC of one eager bench step at HEAD (own pass, --kernel-trace only)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pm
timeout 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pm -- python $R/bench.py --steps 1 --warmup 0 --no-graph --chains 1 --also-dtype none --no-cpu-baseline --no-roofline > $O/r4_pmc_mfma.log 2>&1; echo "mfma rc=$?"
cd $R
python tools/bench_pmc_mfma.py /tmp/pm > $O/r4_bench_pmc_mfma_head.json 2> $O/r4_pmc_mfma.err; head -c 1200 $O/r4_bench_pmc_mfma_head.json; echo
