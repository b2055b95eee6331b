#!/usr/bin/env python3
"""GPU timeline of a few consecutive training steps from a rocprofv3 kernel trace: start offset, duration and queue of every
kernel between the n-th and the (n + count)-th step front, plus the idle gaps of the whole device.  Shows what the side lane
of the data-parallel step (reduce-scatter / Adam slice / all-gather) overlaps with and what it leaves exposed.
A step of a staged discriminator pass has no front launch of its own (stage_pass): give the substring of a kernel every such step
launches once as the anchor (e.g. de_dcoef_kernel) and the window runs from one occurrence to the next -- one whole step, rotated.
usage: timeline.py <kernel_trace.csv> [first_front=40] [fronts=4] [anchor substring]"""
import csv
import sys

path = sys.argv[1]
first = int(sys.argv[2]) if len(sys.argv) > 2 else 40
count = int(sys.argv[3]) if len(sys.argv) > 3 else 4
rows = [r for r in csv.DictReader(open(path)) if r.get("Start_Timestamp")]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
anchor = sys.argv[4] if len(sys.argv) > 4 else None
fronts = [i for i, r in enumerate(rows) if (anchor in r["Kernel_Name"] if anchor else
                                            "front_kernel" in r["Kernel_Name"] or "densify_rows_kernel" in r["Kernel_Name"])]
if len(fronts) < first + count + 1:
    first = max(0, len(fronts) - count - 1)
a, b = fronts[first], fronts[first + count]
t0 = int(rows[a]["Start_Timestamp"])
busy_end = t0
idle = 0.0
print("%9s %8s %6s  %s" % ("start us", "dur us", "queue", "kernel"))
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s > busy_end:
        idle += (s - busy_end) / 1e3
    busy_end = max(busy_end, e)
    name = r["Kernel_Name"]
    for cut in ("ganmf::", "void "):
        name = name.replace(cut, "")
    print("%9.1f %8.1f %6s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), name[:90]))
span = (int(rows[b]["Start_Timestamp"]) - t0) / 1e3
print("window: %d step fronts, %.1f us, device idle %.1f us (%.1f %%)" % (count, span, idle, 100 * idle / span))
