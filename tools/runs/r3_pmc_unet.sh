#!/bin/bash
# round 3: FETCH_SIZE / WRITE_SIZE passes over the UNet part of the bench step (2 eager CFG steps at the bench batch, one
# kernel chain). rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE segfaults inside the tool on the composite bench command (rc 139 right
# after HSA init, gpurun_out/r3_pmc_fetch.log) while the SQ_VALU_MFMA_BUSY_CYCLES pass of the same command runs; this
# sub-command (96 % of the step's GEMM time) runs.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
B="python $R/tools/unet_eager_steps.py --steps 2"
rm -rf /tmp/pf /tmp/pw
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pf -- $B > $O/r3_unet_pmc_fetch.log 2>&1; echo "fetch rc=$?"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pw -- $B > $O/r3_unet_pmc_write.log 2>&1; echo "write rc=$?"
cd $R
python tools/bench_pmc_traffic.py /tmp/pf /tmp/pw 16 r3 unet_only > $O/r3_unet_pmc_traffic.json 2> $O/r3_pmc.err
cat $O/r3_unet_pmc_traffic.json; tail -3 $O/r3_pmc.err
