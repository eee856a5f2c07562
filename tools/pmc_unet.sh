#!/bin/bash
# FETCH_SIZE / WRITE_SIZE passes (separate, --kernel-trace only) over 2 eager UNet CFG steps at the bench batch and dtype, plus
# the algorithmic bytes of the SAME launches (GEMM family and attention kernel). rocprofv3 crashes with these counters on the
# composite bench command (round 3); the UNet holds 96 % of the step's GEMM time.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
DT=${1:-fp16}
RN=${2:-r6}      # round tag of the output files
cd /tmp && export TMPDIR=/tmp
B="python $R/tools/unet_eager_steps.py --steps 2 --dtype $DT"
rm -rf /tmp/pf /tmp/pw
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pf -- $B --alg-json /tmp/alg.json > $O/${RN}_unet_pmc_fetch.log 2>&1; echo "fetch rc=$?"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pw -- $B > $O/${RN}_unet_pmc_write.log 2>&1; echo "write rc=$?"
cd $R
python tools/bench_pmc_traffic.py /tmp/pf /tmp/pw /tmp/alg.json $RN $O/${RN}_gemm_traffic_by_shape.txt > $O/${RN}_unet_pmc_traffic.json 2> $O/${RN}_pmc.err
head -3 /tmp/pf/*/*counter_collection.csv | cut -c1-600 > $O/${RN}_pmc_csv_head.txt
cut -c1-1500 $O/${RN}_unet_pmc_traffic.json; tail -3 $O/${RN}_pmc.err; cat $O/${RN}_gemm_traffic_by_shape.txt
