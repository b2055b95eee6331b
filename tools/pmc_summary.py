#!/usr/bin/env python3
"""rocprofv3 --pmc counter_collection.csv -> one line per (kernel, grid): mean of every counter over the dispatches.
usage: pmc_summary.py <counter_collection.csv> [name-filter]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(dict)
for r in rows:
    if flt and flt not in r["Kernel_Name"]:
        continue
    key = (r["Kernel_Name"][:70], r["Grid_Size"])
    acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    if "Start_Timestamp" in r and r.get("End_Timestamp"):
        dur[key][r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for key, cs in acc.items():
    d = list(dur[key].values())
    print("%s grid %s: %d dispatches, avg %.2f us" % (key[0], key[1], len(d), sum(d) / max(len(d), 1)))
    for name, v in sorted(cs.items()):
        print("    %-28s %.4g" % (name, sum(v) / len(v)))
