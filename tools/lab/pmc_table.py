"""Sum rocprofv3 --pmc counter CSVs per kernel name (first 60 chars) and per grid size; prints counters per dispatch."""
import csv
import glob
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            key = (r["Kernel_Name"][:70], r.get("Grid_Size", ""))
            acc[key][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[key][r["Counter_Name"]] += 1
for key in sorted(acc):
    print(key[0], "grid", key[1])
    for c in sorted(acc[key]):
        n = cnt[key][c]
        print("    %-28s %16.0f per dispatch (%d dispatches)" % (c, acc[key][c] / n, n))
