#!/bin/bash
# round 4 final pass at HEAD: smoke(), the whole GPU suite, the default bench line
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
python -c "import __graft_entry__ as g; g.smoke()" > $O/r4_smoke.log 2>&1; echo "smoke rc=$?"; tail -9 $O/r4_smoke.log
timeout 2400 python -m pytest tests -m gpu -q > $O/r4_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -6 $O/r4_pytest_gpu.log
python bench.py > $O/r4_bench_line_head.json 2> $O/r4_bench_line_head.err; echo "bench rc=$?"; head -c 500 $O/r4_bench_line_head.json; echo
