"""Per-kernel table of the dispatches BETWEEN the two sx_profile_marker launches of `bench.py --profile-markers` in a rocprofv3
--kernel-trace CSV: the timed step only, without the model build in front of it.   python tools/kstats_step.py <dir> [top N]"""
import csv
import glob
import sys
from collections import defaultdict

f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rows = list(csv.DictReader(open(f)))
key_name = "Kernel_Name" if "Kernel_Name" in rows[0] else "Name"
ks, ke = ("Start_Timestamp", "End_Timestamp") if "Start_Timestamp" in rows[0] else ("Start", "End")
rows.sort(key=lambda r: int(r[ks]))
marks = [i for i, r in enumerate(rows) if "profile_marker_kernel" in r[key_name]]
if len(marks) >= 2:
    rows = rows[marks[0] + 1:marks[-1]]
    scope = "dispatches between the two profile markers (the timed step)"
else:
    scope = "ALL dispatches (no profile markers found)"
agg = defaultdict(lambda: [0, 0])
for r in rows:
    a = agg[r[key_name]]
    a[0] += 1
    a[1] += int(r[ke]) - int(r[ks])
tot = sum(v[1] for v in agg.values())
span = (int(rows[-1][ke]) - int(rows[0][ks])) if rows else 0
print("kernel time %.1f ms over %d dispatches, first start to last end %.1f ms; %s (%s)" % (tot / 1e6, len(rows), span / 1e6, scope, f))
native = sum(v[1] for k, v in agg.items() if "at::native" in k or "rocclr" in k)
nn = sum(v[0] for k, v in agg.items() if "at::native" in k or "rocclr" in k)
print("at::native / runtime copy kernels inside the window: %d dispatches, %.2f ms (%.2f %%)" % (nn, native / 1e6, 100.0 * native / max(tot, 1)))
for name, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    short = name.replace("void ", "").replace("sxk_", "")[:86]
    print("%-86s %7d calls %9.1f ms %5.1f%%  avg %8.1f us" % (short, n, t / 1e6, 100.0 * t / tot, t / n / 1e3))
