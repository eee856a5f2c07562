"""Aggregate rocprofv3 --pmc counter_collection CSVs: per-kernel mean of a counter (HBM traffic of the GEMM family).

gfx950 / ROCm 7.2 corrections (MI355X_MICROARCH.md §HBM): FETCH_SIZE and WRITE_SIZE are in KiB; FETCH_SIZE reports half
of the bytes of wide coalesced reads → doubled here. WRITE_SIZE is uncalibrated (reported as is)."""
import collections
import csv
import glob
import json
import sys


def load(dirname, counter):
    agg = collections.defaultdict(lambda: [0, 0.0])
    files = glob.glob(dirname + "/**/*counter_collection.csv", recursive=True)
    for f in files:
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter:
                continue
            name = r["Kernel_Name"].split("(")[0]
            agg[name][0] += 1
            agg[name][1] += float(r["Counter_Value"])
    return agg


def main():
    fetch = load(sys.argv[1], "FETCH_SIZE")
    write = load(sys.argv[2], "WRITE_SIZE")
    out = {}
    tot = {"launches": 0, "fetch_bytes": 0.0, "write_bytes": 0.0}
    for name in sorted(fetch, key=lambda k: -fetch[k][1]):
        n, v = fetch[name]
        wn, wv = write.get(name, [0, 0.0])
        rec = {"launches": n, "fetch_bytes_per_launch": 2 * 1024 * v / max(n, 1),
               "write_bytes_per_launch": 1024 * wv / max(wn, 1)}
        out[name] = rec
        if "gemm_kernel" in name:
            tot["launches"] += n
            tot["fetch_bytes"] += 2 * 1024 * v
            tot["write_bytes"] += 1024 * wv
    res = {"gemm_family": {"launches": tot["launches"],
                           "hbm_bytes_per_launch": (tot["fetch_bytes"] + tot["write_bytes"]) / max(tot["launches"], 1),
                           "fetch_bytes_per_launch": tot["fetch_bytes"] / max(tot["launches"], 1),
                           "write_bytes_per_launch": tot["write_bytes"] / max(tot["launches"], 1)},
           "per_kernel": out}
    json.dump(res, open(sys.argv[3], "w"), indent=1)
    print(json.dumps(res["gemm_family"]))


if __name__ == "__main__":
    main()
