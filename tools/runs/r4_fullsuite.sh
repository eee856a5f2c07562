set -x
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -q -m gpu -x > gpurun_out/r4_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -8 gpurun_out/r4_pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4_smoke.log 2>&1; echo "smoke rc=$?"; tail -8 gpurun_out/r4_smoke.log
