#!/bin/bash
# round 4 final pass at HEAD, part 1: smoke() and the whole GPU suite
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
python -c "import __graft_entry__ as g; g.smoke()" > $O/r4_smoke.log 2>&1; echo "smoke rc=$?"; tail -9 $O/r4_smoke.log
timeout 1500 python -m pytest tests -m gpu -q > $O/r4_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -8 $O/r4_pytest_gpu.log
