set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for f in 1 0; do
  rm -rf /tmp/ks$f
  SX_RMS_FOLD=$f rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks$f -- python $R/bench.py --config 2 --steps 2 --warmup 1 --also-dtype none --no-cpu-baseline --no-roofline > $O/r4_cfg2_rocprof_fold$f.log 2>&1
  echo "fold=$f rc=$?"
  python $R/tools/kstats_top.py /tmp/ks$f 14 > $O/r4_cfg2_kstats_fold$f.txt
  cat $O/r4_cfg2_kstats_fold$f.txt | cut -c1-170
done
