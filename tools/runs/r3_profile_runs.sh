#!/bin/bash
# round 3: bench line + rocprofv3 kernel stats of one eager step (single kernel chain for the profiled pass)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
TAG=${1:-r3}
cd $R
python bench.py --steps 2 --warmup 1 > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err; tail -c 1500 $O/${TAG}_bench.json
cd /tmp && export TMPDIR=/tmp
rm -rf $O/${TAG}_ks
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_ks -- python $R/bench.py --steps 1 --warmup 0 --no-graph --chains 1 --no-cpu-baseline --no-roofline > $O/${TAG}_bench_under_rocprof.log 2>&1
cd $R
python tools/kstats_top.py $O/${TAG}_ks 40 > $O/${TAG}_bench_kernel_top.txt
cp $(find $O/${TAG}_ks -name "*kernel_stats.csv" | head -1) $O/${TAG}_bench_kernel_stats.csv
find $O/${TAG}_ks -name "*kernel_trace.csv" -delete
head -30 $O/${TAG}_bench_kernel_top.txt
if [ "$2" = "pmc" ]; then
  # PMC passes of the same command (counters in their own runs, kernel-trace only)
  cd /tmp && export TMPDIR=/tmp
  B="python $R/bench.py --steps 1 --warmup 0 --no-graph --chains 1 --no-cpu-baseline --no-roofline"
  rm -rf /tmp/pf /tmp/pw /tmp/pm
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pf -- $B > $O/${TAG}_pmc_fetch.log 2>&1; echo "fetch rc=$?"
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pw -- $B > $O/${TAG}_pmc_write.log 2>&1; echo "write rc=$?"
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pm -- $B > $O/${TAG}_pmc_mfma.log 2>&1; echo "mfma rc=$?"
  cd $R
  python tools/bench_pmc_traffic.py /tmp/pf /tmp/pw 16 r3 > $O/${TAG}_bench_pmc_traffic.json 2> $O/${TAG}_pmc.err
  python tools/bench_pmc_mfma.py /tmp/pm > $O/${TAG}_bench_pmc_mfma.json 2>> $O/${TAG}_pmc.err
  cat $O/${TAG}_bench_pmc_traffic.json; head -30 $O/${TAG}_bench_pmc_mfma.json; tail -3 $O/${TAG}_pmc.err
fi
