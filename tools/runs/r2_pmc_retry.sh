#!/bin/bash
# PMC passes of the bench command (retry; falls back to --vae-precision fast to see whether the VAE's new kernels matter)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 1 --warmup 0 --no-graph --chains 1 --no-cpu-baseline --no-roofline"
rm -rf /tmp/pf /tmp/pw /tmp/pm
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pf -- $B > $O/r2_pmc_fetch.log 2>&1; echo "fetch rc=$?"
if ! find /tmp/pf -name "*counter_collection.csv" | grep -q .; then
  echo "FETCH pass produced nothing; retry with --vae-precision fast"
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pf -- $B --vae-precision fast > $O/r2_pmc_fetch2.log 2>&1; echo "fetch(fast vae) rc=$?"
  exit 0
fi
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pw -- $B > $O/r2_pmc_write.log 2>&1; echo "write rc=$?"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pm -- $B > $O/r2_pmc_mfma.log 2>&1; echo "mfma rc=$?"
cd $R
python tools/bench_pmc_traffic.py /tmp/pf /tmp/pw 16 r2 > $O/r2_bench_pmc_traffic.json 2> $O/r2_pmc.err
python tools/bench_pmc_mfma.py /tmp/pm > $O/r2_bench_pmc_mfma.json 2>> $O/r2_pmc.err
cat $O/r2_bench_pmc_mfma.json | head -20; cat $O/r2_bench_pmc_traffic.json; tail -3 $O/r2_pmc.err
