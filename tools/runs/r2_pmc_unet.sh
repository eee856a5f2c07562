#!/bin/bash
# PMC passes over the UNet part of the bench step only (2 eager CFG steps at the bench batch): rocprofv3 --pmc segfaults in
# its dispatch interception on the composite bench command at this commit (see profiles/r2_README.md), these sub-commands run.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
B="python $R/tools/unet_eager_steps.py --steps 2"
rm -rf /tmp/pf /tmp/pw /tmp/pm
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pf -- $B > /dev/null 2>&1; echo "fetch rc=$?"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pw -- $B > /dev/null 2>&1; echo "write rc=$?"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pm -- $B > /dev/null 2>&1; echo "mfma rc=$?"
cd $R
python tools/bench_pmc_traffic.py /tmp/pf /tmp/pw 16 r2 > $O/r2_unet_pmc_traffic.json 2> $O/r2_pmc.err
python tools/bench_pmc_mfma.py /tmp/pm > $O/r2_unet_pmc_mfma.json 2>> $O/r2_pmc.err
head -30 $O/r2_unet_pmc_mfma.json; cat $O/r2_unet_pmc_traffic.json; tail -3 $O/r2_pmc.err
