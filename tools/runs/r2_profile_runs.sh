#!/bin/bash
# bench line + rocprofv3 kernel stats + PMC passes only (single kernel chain for the profiled passes)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
python bench.py --steps 3 --warmup 1 > $O/r2_bench_final.json 2> $O/r2_bench_final.err; tail -c 300 $O/r2_bench_final.json
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 1 --warmup 0 --no-graph --chains 1 --no-cpu-baseline --no-roofline"
rm -rf $O/r2_ks /tmp/pf /tmp/pw /tmp/pm
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r2_ks -- python $R/bench.py --steps 1 --warmup 0 --no-graph --chains 1 --no-cpu-baseline > $O/r2_bench_under_rocprof.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pf -- $B > $O/r2_pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pw -- $B > $O/r2_pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pm -- $B > $O/r2_pmc_mfma.log 2>&1
cd $R
python tools/bench_pmc_traffic.py /tmp/pf /tmp/pw 16 r2 > $O/r2_bench_pmc_traffic.json 2> $O/r2_pmc.err
python tools/bench_pmc_mfma.py /tmp/pm > $O/r2_bench_pmc_mfma.json 2>> $O/r2_pmc.err
python tools/kstats_top.py $O/r2_ks 30 > $O/r2_bench_kernel_top.txt
cp $(find $O/r2_ks -name "*kernel_stats.csv" | head -1) $O/r2_bench_kernel_stats.csv
find $O/r2_ks -name "*kernel_trace.csv" -delete
head -16 $O/r2_bench_kernel_top.txt; cat $O/r2_bench_pmc_mfma.json | head -20; cat $O/r2_bench_pmc_traffic.json; tail -3 $O/r2_pmc.err; tail -2 $O/r2_pmc_fetch.log
