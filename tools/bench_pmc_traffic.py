"""Aggregate the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the bench command into per-launch traffic of the
GEMM/conv kernel family (the `roofline.traffic` figure of bench.py).

On the GPU box:
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pf -- python bench.py --steps 1 --warmup 0 \
      --no-graph --no-cpu-baseline --no-roofline
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pw -- python bench.py --steps 1 --warmup 0 ...
  python tools/bench_pmc_traffic.py /tmp/pf /tmp/pw <batch> r2 > profiles/r2_bench_pmc_traffic.json

Corrections per MI355X_MICROARCH.md §HBM: both counters are in KiB; FETCH_SIZE reports half of the bytes of wide coalesced
reads (x2 here); WRITE_SIZE is uncalibrated (as is). Bytes are L2-miss traffic towards Infinity Cache / HBM."""
import csv
import glob
import json
import sys


def family_sum(dirname, counter):
    n, tot = 0, 0.0
    for f in glob.glob(dirname + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == counter and ("gemm_kernel" in r["Kernel_Name"] or "gemm_pp_kernel" in r["Kernel_Name"]):
                n += 1
                tot += float(r["Counter_Value"])
    return n, tot


def kernel_source_sha():
    """Same hash bench.py computes: the profile is only quoted while the GEMM sources are unchanged."""
    import hashlib
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    for f in ("gemm.hip", "gemm_pp.hip", "gemm_common.h", "sx_common.h"):
        with open(os.path.join(root, "seed-x_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def main():
    nf, fetch = family_sum(sys.argv[1], "FETCH_SIZE")
    nw, write = family_sum(sys.argv[2], "WRITE_SIZE")
    assert nf > 0 and nf == nw, (nf, nw)
    fb, wb = 2 * 1024 * fetch, 1024 * write
    print(json.dumps({"kernel": "sxk_gemm::gemm_pp_kernel<*> + gemm_kernel<*>", "kernels": sys.argv[4] if len(sys.argv) > 4 else "r1",
                      "kernel_source_sha": kernel_source_sha(),
                      "batch_per_gpu": int(sys.argv[3]), "launches": nf,
                      "scope": sys.argv[5] if len(sys.argv) > 5 else "bench_step",
                      "fetch_bytes_per_launch": fb / nf, "write_bytes_per_launch": wb / nf,
                      "traffic_bytes_per_launch": (fb + wb) / nf,
                      "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over one eager bench step; "
                                "FETCH_SIZE x2 (gfx950 wide-read correction), KiB units"}))


if __name__ == "__main__":
    main()
