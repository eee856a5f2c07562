#!/bin/bash
# round 5 final pass at HEAD: headline line, rocprofv3 kernel table of the step, the whole GPU suite, smoke
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
timeout 700 python bench.py --steps 5 --warmup 2 > $O/r5_bench_line.json 2> $O/r5_bench_line.err; echo "bench rc=$?"; head -c 300 $O/r5_bench_line.json; echo
cd /tmp && export TMPDIR=/tmp
rm -rf $O/r5_ks
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r5_ks -- python $R/bench.py --steps 1 --warmup 0 --no-graph --chains 1 --also-dtype none --no-cpu-baseline --no-roofline --profile-markers > $O/r5_bench_under_rocprof.log 2>&1; echo "kstats rc=$?"
cd $R
python tools/kstats_top.py $O/r5_ks 45 > $O/r5_bench_kernel_top_whole_process.txt
python tools/kstats_step.py $O/r5_ks 60 > $O/r5_bench_kernel_top.txt
cp $(find $O/r5_ks -name "*kernel_stats.csv" | head -1) $O/r5_bench_kernel_stats.csv
rm -rf $O/r5_ks
head -8 $O/r5_bench_kernel_top.txt
bash tools/runs/r5_fullsuite.sh
