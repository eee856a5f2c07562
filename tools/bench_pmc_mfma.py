"""MFMA-pipe utilisation of the GEMM/conv and attention kernels over one eager bench step, from a rocprofv3 --pmc pass:

  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pm -- \
      python bench.py --steps 1 --warmup 0 --no-graph --no-cpu-baseline --no-roofline
  python tools/bench_pmc_mfma.py /tmp/pm > profiles/r1_bench_pmc_mfma.json

SQ_VALU_MFMA_BUSY_CYCLES is summed over the 1024 SIMDs (= 16 cycles per 16x16x32 and 32 per 32x32x16 bf16 MFMA);
GRBM_GUI_ACTIVE comes back summed over the 8 XCDs (the raw ratio to wall time is 18.3 "GHz" = 8 x 2.29 GHz), so
active cycles = GRBM_GUI_ACTIVE / 8, effective clock = that / kernel wall time (MI355X_MICROARCH.md, DVFS note) and
util = busy / (1024 SIMDs * active cycles)."""
import collections
import csv
import glob
import json
import sys


def main():
    d = sys.argv[1]
    cnt = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.defaultdict(int)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            fam = "gemm" if ("gemm_kernel" in k or "gemm_pp_kernel" in k) else "attention" if "attn_kernel" in k else None
            if fam is None:
                continue
            cnt[fam][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                n[fam] += 1
    dur = collections.defaultdict(float)
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            fam = "gemm" if ("gemm_kernel" in k or "gemm_pp_kernel" in k) else "attention" if "attn_kernel" in k else None
            if fam:
                dur[fam] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    out = {}
    for fam in cnt:
        busy, act = cnt[fam]["SQ_VALU_MFMA_BUSY_CYCLES"], cnt[fam]["GRBM_GUI_ACTIVE"] / 8.0
        out[fam] = {"launches": n[fam], "mfma_busy_cycles": busy, "gui_active_cycles": act, "kernel_time_s": dur[fam] * 1e-9,
                    "effective_clock_ghz": act / dur[fam] if dur[fam] else None,
                    "mfma_pipe_util": busy / (1024.0 * act) if act else None}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
