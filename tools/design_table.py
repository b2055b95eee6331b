#!/usr/bin/env python3
"""Regenerates the per-class table of DESIGN.md section 4 from profiles/r02_step_classes.md, r02_step_classes_sq.md and
r02_traffic.json (so that the document quotes exactly the committed evidence).  usage: python tools/design_table.py"""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = lambda *a: os.path.join(ROOT, "profiles", *a)
cls, sq = {}, {}
for line in open(P("r02_step_classes.md")):
    m = re.match(r"\| (\S+:[^|]+?) \|.*\| (\d+) \| ([\d.]+) \|$", line.strip())
    if m:
        cls[m.group(1)] = float(m.group(3))
for line in open(P("r02_step_classes_sq.md")):
    parts = [x.strip() for x in line.strip().strip("|").split("|")]
    if len(parts) > 6 and ":" in parts[0] and parts[0][0] in "DGS":
        try:
            sq[parts[0]] = float(parts[-1])
        except ValueError:
            pass
tr = {v["class"]: (v["hbm_bytes_per_launch"] / 1e6, v["algorithmic_bytes"] / 1e6) for v in json.load(open(P("r02_traffic.json"))).values()}
u = lambda k: "%.0f %%" % sq[k] if sq.get(k, 0) > 0 else "—"
h = lambda k: "%.1f / %.1f" % tr[k]
D = sum(v for k, v in cls.items() if k.startswith("D:"))
G = sum(v for k, v in cls.items() if k.startswith("G:"))
rows = [("generator GEMM + CSR rows (one launch)", "D:gen+rows", "generator GEMM + CSR rows", "G:gen+rows", None, None),
        ("encode `[2B,N+1]×[N+1,e]`, split 4", "D:encode", "encode", "G:encode", "D:reduce(encode)", "G:reduce(encode)"),
        ("decode + Δ + Σ² (two paths)", "D:decode", "decode (generated half), split 2", "G:decode", None, "G:reduce(decode)"),
        ("dE, split 4 (slabs only) + d_coef (one launch); slab sum", "D:dE+d_coef", "dE, split 8", "G:dE", "D:reduce(dE)", "G:reduce(dE)"),
        ("gWd_ext + gWe_ext, both + Adam (one launch)", "D:gWd+gWe+adam", "dF, split 2", "G:dF", None, "G:reduce(dF)"),
        (None, None, "gUb (split 29) + gV + Adam (one launch)", "G:gUb+gV+adam", None, None),
        (None, None, "all-rows Adam on U (sums gUb's slabs)", "G:adam_rows_U", None, None)]
out = ["| class (D-step) | µs | MFMA busy | HBM bytes / algorithmic | class (G-step) | µs | MFMA busy | HBM / alg. |", "|---|---|---|---|---|---|---|---|"]
for dn, dk, gn, gk, dr, gr in rows:
    if dk:
        dus = "%.1f" % cls[dk] + (" + %.1f reduce" % cls[dr] if dr else "")
        dcell = "| %s | %s | %s | %s MB " % (dn, dus, u(dk), h(dk))
    else:
        dcell = "| | | | "
    gus = "%.1f" % cls[gk] + (" + %.1f" % cls[gr] if gr else "")
    out.append(dcell + "| %s | %s | %s | %s |" % (gn, gus, u(gk), h(gk)))
out.append("| **D-step** | **%.1f** | | | **G-step** | **%.1f** | | |" % (D, G))
tbl = "\n".join(out) + "\n\n"
path = os.path.join(ROOT, "DESIGN.md")
s = open(path).read()
a = s.index("| class (D-step) | µs | MFMA busy | HBM bytes / algorithmic | class (G-step) |")
b = s.index("(MFMA busy = `SQ_VALU_MFMA_BUSY_CYCLES` per SIMD")
s = s[:a] + tbl + s[b:]
val = json.load(open(P("r02_bench.json")))["value"]
s = re.sub(r"A D\+G pair is \d+ µs of kernels \(bench line: [\d ]+ steps/s", "A D+G pair is %.0f µs of kernels (bench line: %s steps/s" % (D + G, format(int(round(val, -1)), ",").replace(",", " ")), s)
open(path, "w").write(s)
print(tbl)
