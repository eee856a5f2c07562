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
