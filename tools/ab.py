#!/usr/bin/env python3
"""A/B runs of bench.py under different environments: steps/s and the per-class launch averages that changed.
usage: tools/ab.py "NAME=VALUE ..." "NAME=VALUE ..." [--reps N] [--class SUBSTRING]    ("" = default environment)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
reps, cls = 2, None
if "--reps" in args:
    i = args.index("--reps"); reps = int(args[i + 1]); del args[i:i + 2]
if "--class" in args:
    i = args.index("--class"); cls = args[i + 1]; del args[i:i + 2]
for rep in range(reps):
    for spec in args:
        env = dict(os.environ)
        for kv in spec.split():
            k, v = kv.split("=", 1)
            env[k] = v
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline"], env=env, capture_output=True, text=True)
        try:
            d = json.loads(out.stdout.strip().splitlines()[-1])
        except Exception:
            print("%-50s FAILED: %s" % (spec or "(default)", out.stderr[-300:]))
            continue
        extra = ""
        if cls:
            extra = "  ".join("%s:%s %.2f us" % (k.get("step", "?"), k["name"][:28], k["avg_us"]) for k in d["kernels"] if cls in k["name"])
        print("%-50s %8.1f steps/s  D %8.1f  G %8.1f  fn %.3f  %s" % (spec or "(default)", d["value"], d.get("d_steps_per_s", 0), d.get("g_steps_per_s", 0), d["roofline"].get("frac", 0), extra), flush=True)
