#!/bin/bash
# round 4 final pass at HEAD, part 2: PMC traffic of the UNet's launches (the bench line quotes it), the bench line, rocprofv3 kernel stats
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
timeout 420 bash tools/pmc_unet.sh fp16 > $O/r4_pmc_unet.out 2>&1; head -c 400 $O/r4_unet_pmc_traffic.json; echo
[ -s $O/r4_unet_pmc_traffic.json ] && cp $O/r4_unet_pmc_traffic.json profiles/r4_unet_pmc_traffic.json
timeout 400 python bench.py --steps 5 --warmup 2 > $O/r4_bench_line_head.json 2> $O/r4_bench_line_head.err; echo "bench rc=$?"; head -c 600 $O/r4_bench_line_head.json; echo
cd /tmp && export TMPDIR=/tmp
rm -rf $O/r4_ks
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r4_ks -- python $R/bench.py --steps 1 --warmup 0 --no-graph --chains 1 --also-dtype none --no-cpu-baseline --no-roofline > $O/r4_bench_under_rocprof.log 2>&1
cd $R
python tools/kstats_top.py $O/r4_ks 45 > $O/r4_bench_kernel_top.txt
cp $(find $O/r4_ks -name "*kernel_stats.csv" | head -1) $O/r4_bench_kernel_stats.csv
rm -rf $O/r4_ks
head -16 $O/r4_bench_kernel_top.txt
