"""Summarise rocprofv3 counter_collection CSVs: average counter value per dispatch, per kernel."""
import csv
import glob
import json
import sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: [0.0, 0])
for path in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    with open(path) as f:
        for row in csv.DictReader(f):
            k = (row["Kernel_Name"].split("(")[0], row["Counter_Name"])
            acc[k][0] += float(row["Counter_Value"])
            acc[k][1] += 1
out = {}
for (kern, ctr), (tot, cnt) in sorted(acc.items()):
    out.setdefault(kern, {})[ctr] = {"avg_per_dispatch": tot / cnt, "dispatches": cnt}
print(json.dumps(out, indent=1))
