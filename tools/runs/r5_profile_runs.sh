#!/bin/bash
# round 5 measurement pass (one box): PMC traffic of the UNet's launches (must precede the bench line that quotes it), headline line
# (fp16 timed + bf16 short pass, roofline, cpu baseline), rocprofv3 kernel stats of one eager step (whole process AND the step only,
# cut at the profile markers), MFMA-busy PMC, per-shape GEMM table, attention PMC per shape
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
timeout 500 bash tools/pmc_unet.sh fp16 > $O/r5_pmc_unet.out 2>&1; head -c 500 $O/r5_unet_pmc_traffic.json; echo
[ -s $O/r5_unet_pmc_traffic.json ] && cp $O/r5_unet_pmc_traffic.json profiles/r5_unet_pmc_traffic.json       # bench.py reads profiles/ (same box, same sources)
timeout 600 python bench.py --steps 5 --warmup 2 > $O/r5_bench_line.json 2> $O/r5_bench_line.err; echo "bench rc=$?"; head -c 500 $O/r5_bench_line.json; echo
cd /tmp && export TMPDIR=/tmp
rm -rf $O/r5_ks
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r5_ks -- python $R/bench.py --steps 1 --warmup 0 --no-graph --chains 1 --also-dtype none --no-cpu-baseline --no-roofline --profile-markers > $O/r5_bench_under_rocprof.log 2>&1; echo "kstats rc=$?"
cd $R
python tools/kstats_top.py $O/r5_ks 45 > $O/r5_bench_kernel_top_whole_process.txt
python tools/kstats_step.py $O/r5_ks 60 > $O/r5_bench_kernel_top.txt
cp $(find $O/r5_ks -name "*kernel_stats.csv" | head -1) $O/r5_bench_kernel_stats.csv
rm -rf $O/r5_ks
head -30 $O/r5_bench_kernel_top.txt
cd /tmp
rm -rf /tmp/pm
timeout 500 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pm -- python $R/bench.py --steps 1 --warmup 0 --no-graph --chains 1 --also-dtype none --no-cpu-baseline --no-roofline > $O/r5_pmc_mfma.log 2>&1; echo "mfma rc=$?"
cd $R
python tools/bench_pmc_mfma.py /tmp/pm > $O/r5_bench_pmc_mfma.json 2> $O/r5_pmc_mfma.err; cat $O/r5_bench_pmc_mfma.json | head -c 800; echo
timeout 400 python tools/gemm_shape_profile.py --unet-steps 4 > $O/r5_gemm_shapes.txt 2> $O/r5_gemm_shapes.err; head -12 $O/r5_gemm_shapes.txt
ATTN_LAB_SHAPES=0,1,3,4 timeout 400 bash tools/lab/attn_pmc.sh r5_attn 0 > $O/r5_attn_pmc.out 2>&1; echo "attn pmc rc=$?"; cat $O/r5_attn_pmc_derived.txt
