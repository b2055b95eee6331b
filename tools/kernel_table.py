#!/usr/bin/env python3
"""Print the per-class kernel table of bench.py JSON lines: tools/kernel_table.py a.json b.json"""
import json
import sys
for f in sys.argv[1:]:
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, d["value"], d["timing"]["value_samples"])
    for k in d["kernels"]:
        print("   %-50s %5d %8.2f %8.3f" % (k["name"][:50], k["launches"], k["avg_us"], k["total_ms"]))
