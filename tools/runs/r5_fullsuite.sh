#!/bin/bash
# round 5: the whole GPU suite + smoke at the current tree
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
timeout 1700 python -m pytest tests -m gpu -q -s --timeout 900 > $O/r5_pytest_gpu_full.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|FAILED|ERROR" $O/r5_pytest_gpu_full.log | tail -30
grep "full depth" $O/r5_pytest_gpu_full.log > $O/r5_fulldepth.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/r5_smoke.log 2>&1; echo "smoke rc=$?"; tail -9 $O/r5_smoke.log
