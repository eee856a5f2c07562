"""Join rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/gemm_pmc_probe.py with the algorithmic bytes it prints.

gfx950 / ROCm 7.2 corrections (MI355X_MICROARCH.md §HBM): FETCH_SIZE and WRITE_SIZE are in KiB; FETCH_SIZE reports half
of the bytes of wide coalesced reads → doubled here. WRITE_SIZE is uncalibrated (reported as is).
usage: pmc_summary.py <fetch_dir> <write_dir> <probe_log>"""
import csv
import glob
import json
import sys


def load(dirname, counter):
    rows = []
    for f in glob.glob(dirname + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == counter and "gemm_kernel" in r["Kernel_Name"]:
                rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"].split("(")[0], float(r["Counter_Value"])))
    rows.sort()
    return rows


def main():
    fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    alg = None
    for line in open(sys.argv[3]):
        if line.startswith("ALG "):
            alg = json.loads(line[4:])
    assert alg and len(fetch) == 5 * len(alg) == len(write), (len(fetch), len(write), alg and len(alg))
    out = []
    for i, a in enumerate(alg):
        f = [2 * 1024 * v for _, _, v in fetch[5 * i + 1:5 * i + 5]]          # skip the first (cold) launch
        w = [1024 * v for _, _, v in write[5 * i + 1:5 * i + 5]]
        rec = dict(a, kernel=fetch[5 * i][1], fetch_bytes=sum(f) / len(f), write_bytes=sum(w) / len(w))
        rec["traffic_over_algorithmic"] = (rec["fetch_bytes"] + rec["write_bytes"]) / a["alg_bytes"]
        out.append(rec)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
