#!/bin/bash
# Final round-2 measurement pass after the fp32-grade VAE became the bench default: bench line + rocprofv3 stats + PMC passes
# (tools/runs/r2_profile_runs.sh), configs 3 and 4 (the ones that decode images), phase times, VAE parity numbers.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
bash $R/tools/runs/r2_profile_runs.sh
cd $R
for c in 3 4; do python bench.py --config $c --steps 2 --warmup 1 --no-cpu-baseline > $O/r2_bench_config$c.json 2>> $O/r2_bench_configs.err; tail -c 200 $O/r2_bench_config$c.json; done
python tools/phase_times.py 2>&1 | grep -v amdgpu.ids > $O/r2_phase_times.txt; cat $O/r2_phase_times.txt
python -m pytest tests/test_vae_gpu.py tests/test_fullsize2_gpu.py -k "vae or split or planes" -q -s 2>&1 | grep -E "rel-L2|passed|failed" > $O/r2_vae_parity.txt; tail -12 $O/r2_vae_parity.txt
