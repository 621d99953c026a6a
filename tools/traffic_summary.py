#!/usr/bin/env python3
"""Per-kernel average FETCH_SIZE / WRITE_SIZE (KB, raw) from tools/profile_traffic.sh."""
import collections
import csv
import glob
import json
import sys

d = sys.argv[1]
out = {}
for tag in ("FETCH_SIZE", "WRITE_SIZE", "odd_FETCH_SIZE", "odd_WRITE_SIZE"):
    agg = collections.defaultdict(list)
    for f in glob.glob("%s/%s/**/*counter_collection.csv" % (d, tag), recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0].split("::")[-1].split("<")[0]
            agg[(name, r["Grid_Size"])].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        out.setdefault("%s grid=%s" % k, {})[tag] = sum(v) / len(v)
for k, v in sorted(out.items(), key=lambda kv: -sum(kv[1].values())):
    if max(v.values()) > 1e4:
        print("%-45s %s" % (k, {a: "%.4g KB" % b for a, b in v.items()}))
json.dump(out, open(d + "/summary.json", "w"), indent=1)
