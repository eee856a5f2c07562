#!/bin/bash
# round 3 final pass at HEAD: smoke(), GPU test suite, headline line (with PMC traffic quoted), config 2 line
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
python -c "import __graft_entry__ as g; g.smoke()" > $O/r3_smoke.log 2>&1; tail -4 $O/r3_smoke.log
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -40 > $O/r3_pytest_gpu.log; tail -3 $O/r3_pytest_gpu.log
python bench.py --steps 2 --warmup 1 > $O/r3_bench_line.json 2> $O/r3_bench_line.err; head -c 300 $O/r3_bench_line.json; echo
python bench.py --config 2 --steps 3 --warmup 1 --no-cpu-baseline > $O/r3_bench_config2.json 2>> $O/r3_bench_line.err; head -c 250 $O/r3_bench_config2.json; echo
