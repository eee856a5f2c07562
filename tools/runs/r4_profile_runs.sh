#!/bin/bash
# round 4 measurement pass (one box): PMC traffic of the UNet's launches (must precede the bench line that quotes it), headline line
# (fp16 timed + bf16 short pass, roofline, cpu baseline), rocprofv3 kernel stats + MFMA-busy PMC of one eager step, per-shape GEMM table
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
bash tools/pmc_unet.sh fp16 > $O/r4_pmc_unet.out 2>&1; head -c 600 $O/r4_unet_pmc_traffic.json; echo
cp $O/r4_unet_pmc_traffic.json profiles/r4_unet_pmc_traffic.json       # bench.py reads profiles/ (same box, same sources)
python bench.py --steps 5 --warmup 2 > $O/r4_bench_line.json 2> $O/r4_bench_line.err; head -c 400 $O/r4_bench_line.json; echo
cd /tmp && export TMPDIR=/tmp
rm -rf $O/r4_ks
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r4_ks -- python $R/bench.py --steps 1 --warmup 0 --no-graph --chains 1 --also-dtype none --no-cpu-baseline --no-roofline > $O/r4_bench_under_rocprof.log 2>&1
cd $R
python tools/kstats_top.py $O/r4_ks 45 > $O/r4_bench_kernel_top.txt
cp $(find $O/r4_ks -name "*kernel_stats.csv" | head -1) $O/r4_bench_kernel_stats.csv
rm -rf $O/r4_ks
head -24 $O/r4_bench_kernel_top.txt
cd /tmp
rm -rf /tmp/pm
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pm -- python $R/bench.py --steps 1 --warmup 0 --no-graph --chains 1 --also-dtype none --no-cpu-baseline --no-roofline > $O/r4_pmc_mfma.log 2>&1; echo "mfma rc=$?"
cd $R
python tools/bench_pmc_mfma.py /tmp/pm > $O/r4_bench_pmc_mfma.json 2> $O/r4_pmc_mfma.err; cat $O/r4_bench_pmc_mfma.json | head -c 800; echo
python tools/gemm_shape_profile.py --unet-steps 4 > $O/r4_gemm_shapes.txt 2> $O/r4_gemm_shapes.err; head -12 $O/r4_gemm_shapes.txt
