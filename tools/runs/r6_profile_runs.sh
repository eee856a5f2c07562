#!/bin/bash
# round 6 measurement pass (one box): PMC traffic of the UNet's launches (must precede the bench line that quotes it), headline line
# (fp16 timed + bf16 short pass, roofline, cpu baseline), rocprofv3 kernel stats of one eager step cut at the profile markers,
# MFMA-busy PMC, per-shape GEMM table
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
timeout 500 bash tools/pmc_unet.sh fp16 r6 > $O/r6_pmc_unet.out 2>&1; head -c 400 $O/r6_unet_pmc_traffic.json; echo
[ -s $O/r6_unet_pmc_traffic.json ] && cp $O/r6_unet_pmc_traffic.json profiles/r6_unet_pmc_traffic.json       # bench.py reads profiles/ (same box, same sources)
timeout 400 python bench.py --config 1 --cpu-baseline full --steps 3 --warmup 1 --also-dtype none --no-roofline > $O/r6_bench_config1_cpu_full.json 2> $O/r6_bench_config1_cpu_full.err; echo "config1 cpu rc=$?"
[ -s $O/r6_bench_config1_cpu_full.json ] && cp $O/r6_bench_config1_cpu_full.json profiles/r6_bench_config1_cpu_full.json
timeout 900 python bench.py --steps 5 --warmup 2 > $O/r6_bench_line.json 2> $O/r6_bench_line.err; echo "bench rc=$?"; head -c 300 $O/r6_bench_line.json; echo
cd /tmp && export TMPDIR=/tmp
rm -rf $O/r6_ks
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r6_ks -- python $R/bench.py --steps 1 --warmup 0 --no-graph --chains 1 --also-dtype none --no-cpu-baseline --no-roofline --profile-markers > $O/r6_bench_under_rocprof.log 2>&1; echo "kstats rc=$?"
cd $R
python tools/kstats_top.py $O/r6_ks 45 > $O/r6_bench_kernel_top_whole_process.txt
python tools/kstats_step.py $O/r6_ks 60 > $O/r6_bench_kernel_top.txt
cp $(find $O/r6_ks -name "*kernel_stats.csv" | head -1) $O/r6_bench_kernel_stats.csv
rm -rf $O/r6_ks
head -24 $O/r6_bench_kernel_top.txt
cd /tmp
rm -rf /tmp/pm
timeout 500 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pm -- python $R/bench.py --steps 1 --warmup 0 --no-graph --chains 1 --also-dtype none --no-cpu-baseline --no-roofline > $O/r6_pmc_mfma.log 2>&1; echo "mfma rc=$?"
cd $R
python tools/bench_pmc_mfma.py /tmp/pm > $O/r6_bench_pmc_mfma.json 2> $O/r6_pmc_mfma.err; cat $O/r6_bench_pmc_mfma.json | head -c 800; echo
timeout 400 python tools/gemm_shape_profile.py --unet-steps 4 > $O/r6_gemm_shapes.txt 2> $O/r6_gemm_shapes.err; head -12 $O/r6_gemm_shapes.txt
