"""Aggregate the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate passes, --kernel-trace only) of
tools/unet_eager_steps.py into per-launch HBM-side traffic of the GEMM/conv family and of the flash-attention kernel, next to
the ALGORITHMIC bytes of exactly the same launches (written by that command's --alg-json).

  python tools/bench_pmc_traffic.py <fetch_dir> <write_dir> <alg.json> <tag> > profiles/<tag>_unet_pmc_traffic.json

Corrections per MI355X_MICROARCH.md §HBM / rocprofv3: both counters are in KiB; FETCH_SIZE reports half of the bytes of wide
coalesced reads (x2 here); WRITE_SIZE is taken as is. FETCH_SIZE counts requests that leave the XCD's L2 — Infinity-Cache (MALL)
hits included — so `traffic / algorithmic` > 1 means re-reads that missed L2, and < 1 is only possible through L2 hits (an operand
tile shared by the workgroups of one XCD that run at the same time), never through the Infinity Cache."""
import csv
import glob
import json
import sys

FAMILIES = {"gemm": ("gemm_kernel", "gemm_pp_kernel"), "attention": ("attn_kernel",)}


def family_sum(dirname, counter, names):
    n, tot = 0, 0.0
    for f in glob.glob(dirname + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == counter and any(k in r["Kernel_Name"] for k in names):
                n += 1
                tot += float(r["Counter_Value"])
    return n, tot


def family_rows(dirname, counter, names):
    """[(dispatch id, counter value KiB, duration ns)] of the family's launches in dispatch order."""
    rows = []
    for f in glob.glob(dirname + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == counter and any(k in r["Kernel_Name"] for k in names):
                dur = float(r.get("End_Timestamp", 0) or 0) - float(r.get("Start_Timestamp", 0) or 0)
                rows.append((int(r.get("Dispatch_Id", len(rows))), float(r["Counter_Value"]), dur))
    rows.sort()
    return rows


def by_shape(fdir, wdir, alg):
    """Per GEMM shape: algorithmic read / written bytes against FETCH_SIZE x2 / WRITE_SIZE of the same launches (matched by
    dispatch order: launch i of the family in each pass is the i-th hooked sx_gemm call)."""
    fr, wr = family_rows(fdir, "FETCH_SIZE", FAMILIES["gemm"]), family_rows(wdir, "WRITE_SIZE", FAMILIES["gemm"])
    recs = alg.get("gemm_launches") or []
    if not (len(fr) == len(wr) == len(recs)):
        return None
    agg = {}
    for (_, fv, fd), (_, wv, wd), rec in zip(fr, wr, recs):
        key = tuple(rec[:7])
        d = agg.setdefault(key, {"calls": 0, "alg_read": 0.0, "alg_write": 0.0, "fetch": 0.0, "write": 0.0, "ns": 0.0})
        d["calls"] += 1; d["alg_read"] += rec[7]; d["alg_write"] += rec[8]
        d["fetch"] += 2 * 1024 * fv; d["write"] += 1024 * wv; d["ns"] += 0.5 * (fd + wd)
    rows = []
    for key, d in agg.items():
        n = d["calls"]
        rows.append({"mode": key[0], "M": key[1], "N": key[2], "K": key[3], "glu": key[4], "res": key[5], "out32": key[6], "calls": n,
                     "alg_read_mb": d["alg_read"] / n / 1e6, "alg_write_mb": d["alg_write"] / n / 1e6,
                     "fetch_mb": d["fetch"] / n / 1e6, "write_mb": d["write"] / n / 1e6,
                     "fetch_over_alg_read": d["fetch"] / d["alg_read"], "write_over_alg_write": d["write"] / d["alg_write"],
                     "us_profiled": d["ns"] / n / 1e3, "excess_mb_total": (d["fetch"] + d["write"] - d["alg_read"] - d["alg_write"]) / 1e6})
    rows.sort(key=lambda r: -r["excess_mb_total"])
    return rows


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
    fdir, wdir, alg_path, tag = sys.argv[1:5]
    alg = json.load(open(alg_path))
    out = {"kernels": tag, "kernel_source_sha": kernel_source_sha(), "batch_per_gpu": alg["batch"], "dtype": alg["dtype"],
           "scope": "unet_only (%d eager CFG steps, one kernel chain: tools/unet_eager_steps.py)" % alg["steps"],
           "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes of the same command); FETCH_SIZE x2 "
                     "(gfx950 wide-read correction), KiB units; algorithmic bytes = operands once + output once of the SAME launches",
           "families": {}}
    for fam, names in FAMILIES.items():
        nf, fetch = family_sum(fdir, "FETCH_SIZE", names)
        nw, write = family_sum(wdir, "WRITE_SIZE", names)
        a = alg["families"][fam]
        assert nf > 0 and nf == nw == a["launches"], (fam, nf, nw, a["launches"])
        fb, wb = 2 * 1024 * fetch, 1024 * write
        out["families"][fam] = {"launches": nf, "fetch_bytes_per_launch": fb / nf, "write_bytes_per_launch": wb / nf,
                                "traffic_bytes_per_launch": (fb + wb) / nf, "algorithmic_bytes_per_launch": a["bytes"] / nf,
                                "traffic_over_algorithmic": (fb + wb) / a["bytes"], "algorithmic_gflop_per_launch": a["flop"] / nf / 1e9}
    g = out["families"]["gemm"]          # top-level copies: what bench.py quotes for its dominant family
    out.update({"kernel": "sxk_gemm::gemm_pp_kernel<*> + gemm_kernel<*>", "launches": g["launches"],
                "fetch_bytes_per_launch": g["fetch_bytes_per_launch"], "write_bytes_per_launch": g["write_bytes_per_launch"],
                "traffic_bytes_per_launch": g["traffic_bytes_per_launch"], "algorithmic_bytes_per_launch": g["algorithmic_bytes_per_launch"],
                "traffic_over_algorithmic": g["traffic_over_algorithmic"]})
    shapes = by_shape(fdir, wdir, alg)
    if shapes is not None:
        out["gemm_by_shape"] = shapes
        if len(sys.argv) > 5:
            with open(sys.argv[5], "w") as fh:
                fh.write("UNet GEMM launches, PMC traffic (FETCH_SIZE x2 | WRITE_SIZE, KiB) vs algorithmic bytes of the same launches, per shape; "
                         "sorted by total excess traffic. us = kernel duration under the profiler.\n")
                fh.write("mode        M      N      K glu res o32 calls | alg read MB  fetched MB  ratio | alg write MB written MB ratio |     us  TB/s moved\n")
                for r in shapes:
                    tb = (r["fetch_mb"] + r["write_mb"]) / r["us_profiled"] if r["us_profiled"] else 0.0
                    fh.write("%-4s %8d %6d %6d  %d   %d   %d  %5d | %11.1f %11.1f %6.2f | %12.1f %10.1f %5.2f | %7.1f %6.2f\n" % (
                        r["mode"], r["M"], r["N"], r["K"], r["glu"], r["res"], r["out32"], r["calls"], r["alg_read_mb"], r["fetch_mb"],
                        r["fetch_over_alg_read"], r["alg_write_mb"], r["write_mb"], r["write_over_alg_write"], r["us_profiled"], tb))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
