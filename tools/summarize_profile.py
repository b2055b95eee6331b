#!/usr/bin/env python3
"""rocprofv3 kernel_trace.csv -> per (kernel, grid) table (markdown).  usage: summarize_profile.py trace.csv > out.md"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(list)
for r in rows:
    acc[(r["Kernel_Name"], r["Grid_Size_X"], r["LDS_Block_Size"], r["VGPR_Count"], r["Accum_VGPR_Count"])].append(
        (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in acc.values())
print("| kernel | grid (threads) | LDS B | VGPR+AGPR | launches | avg us | total ms | % |")
print("|---|---|---|---|---|---|---|---|")
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    print("| `%s` | %s | %s | %s+%s | %d | %.2f | %.3f | %.1f |" % (k[0][:90], k[1], k[2], k[3], k[4], len(v), sum(v) / len(v),
                                                               sum(v) / 1e3, 100 * sum(v) / tot))
print("\ntotal kernel time %.3f ms over %d dispatches" % (tot / 1e3, len(rows)))
