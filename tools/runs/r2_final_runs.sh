#!/bin/bash
# Round-2 measurement pass on the GPU box (one gpurun call): GPU test suite with the parity numbers printed, the default
# bench line, the five BASELINE configs, rocprofv3 kernel stats and the three PMC passes of the bench command.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
python -m pytest tests -m gpu -q -s 2>&1 | grep -v amdgpu.ids > $O/r2_gputest_final.log; tail -3 $O/r2_gputest_final.log
python bench.py --steps 3 --warmup 1 > $O/r2_bench_final.json 2> $O/r2_bench_final.err; tail -c 400 $O/r2_bench_final.json
for c in 1 2 3 4 5; do python bench.py --config $c --steps 2 --warmup 1 --no-cpu-baseline > $O/r2_bench_config$c.json 2>> $O/r2_bench_configs.err; done
python bench.py --config 5 --kv-reuse 0 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $O/r2_bench_config5_noreuse.json 2>> $O/r2_bench_configs.err
python bench.py --batch 1 --steps 3 --warmup 1 --no-cpu-baseline > $O/r2_bench_b1.json 2>> $O/r2_bench_configs.err
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 1 --warmup 0 --no-graph --chains 1 --no-cpu-baseline --no-roofline"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r2_ks -- python $R/bench.py --steps 1 --warmup 0 --no-graph --chains 1 --no-cpu-baseline > $O/r2_bench_under_rocprof.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pf -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pw -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pm -- $B > /dev/null 2>&1
cd $R
python tools/bench_pmc_traffic.py /tmp/pf /tmp/pw 16 r2 > $O/r2_bench_pmc_traffic.json 2> $O/r2_pmc.err
python tools/bench_pmc_mfma.py /tmp/pm > $O/r2_bench_pmc_mfma.json 2>> $O/r2_pmc.err
python tools/kstats_top.py $O/r2_ks 30 > $O/r2_bench_kernel_top.txt
cp $(find $O/r2_ks -name "*kernel_stats.csv" | head -1) $O/r2_bench_kernel_stats.csv
find $O/r2_ks -name "*kernel_trace.csv" -delete
cat $O/r2_bench_kernel_top.txt | head -14; cat $O/r2_bench_pmc_mfma.json | head -20; cat $O/r2_bench_pmc_traffic.json
