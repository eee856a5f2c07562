#!/bin/bash
# round 3: GPU test suite, then the bench line with per-phase roofline, request pipelining A/B and config 2
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -40 > $O/r3_pytest_gpu.log; tail -5 $O/r3_pytest_gpu.log
python bench.py --steps 2 --warmup 1 > $O/r3_bench_line.json 2> $O/r3_bench_line.err; tail -c 600 $O/r3_bench_line.json; echo
python bench.py --steps 3 --warmup 1 --overlap 1 --no-cpu-baseline --no-roofline > $O/r3_bench_overlap1.json 2>> $O/r3_bench_line.err; head -c 400 $O/r3_bench_overlap1.json; echo
python bench.py --config 2 --steps 3 --warmup 1 --no-cpu-baseline > $O/r3_bench_config2.json 2>> $O/r3_bench_line.err; head -c 300 $O/r3_bench_config2.json; echo
