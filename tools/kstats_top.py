"""Print the top kernels of a rocprofv3 --stats kernel_stats.csv (name shortened), with totals."""
import csv
import glob
import sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time %.1f ms (%s)" % (tot / 1e6, f))
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 22]:
    n = r["Name"].replace("void ", "").replace("sxk_", "")[:70]
    print("%-70s %7d calls %9.1f ms %5.1f%%  avg %8.1f us" % (n, int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6,
                                                            100 * float(r["TotalDurationNs"]) / tot, float(r["AverageNs"]) / 1e3))
