#!/usr/bin/env python3
"""rocprofv3 output of a bench.py run -> one row per GEMM / kernel CLASS of the training step, labelled by DISPATCH ORDER.

Several classes share a (kernel, grid) pair (decode of the D-step, decode of the G-step and gUb are all
`gemm_f32_mfma<64,64,64,3,false,true>` on 232 workgroups), so grouping by name cannot separate them.  The library
launches a step in a fixed order (ganmf_hip.hip d_step / g_step):

  D-step: densify, gen, encode[, reduce], decode[, reduce], d_coef, dE[, reduce], gWd+Adam, gWe+Adam
  G-step: densify, gen, encode[, reduce], decode[, reduce], dE[, reduce], dF[, reduce], gUb[, reduce], gV+Adam, adam_rows_U
with the combined launches of gemm_multi.hpp standing in for their parts: `front_kernel` = densify + gen, `pair_kernel` =
gUb + gV+Adam, `gemm_bf16s_red` = gWd+Adam + reduce(dE), `de_dcoef_kernel` = dE + d_coef, `wgrad_pair_kernel` = gWd+Adam + gWe+Adam.

A step starts at `densify_rows_kernel` / `sparse_front_kernel` / `front_kernel`; it is a D-step when it contains `d_coef_kernel` or `de_dcoef_kernel`.

usage: step_classes.py <kernel_trace.csv | counter_collection.csv> [> out.md]
With a counter file every counter is averaged per class next to the duration (PMC runs serialise kernels, so durations
there are for orientation only)."""
import collections
import csv
import sys

D_GEMMS = ["gen", "encode", "decode", "dE", "gWd+adam", "gWe+adam"]
G_GEMMS = ["gen", "encode", "decode", "dE", "dF", "gUb", "gV+adam"]


def load(path):
    rows = list(csv.DictReader(open(path)))
    by = collections.OrderedDict()
    for r in rows:
        d = int(r["Dispatch_Id"])
        e = by.setdefault(d, {"name": r["Kernel_Name"], "counters": {}, "start": None, "end": None,
                              "grid": r.get("Grid_Size_X") or r.get("Grid_Size"), "lds": r.get("LDS_Block_Size"),
                              "wgsize": r.get("Workgroup_Size_X") or r.get("Workgroup_Size") or "256"})
        if r.get("Start_Timestamp") and r.get("End_Timestamp"):
            e["start"], e["end"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if "Counter_Name" in r:
            e["counters"][r["Counter_Name"]] = e["counters"].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return [by[k] for k in sorted(by)]


def _gemm_like(n):
    return "gemm_" in n or "front_kernel" in n or "pair_kernel" in n or "de_dcoef_kernel" in n      # ("wgrad_pair_kernel" contains "pair_kernel")


COUNTS = {"D": 0, "G": 0}      # complete steps labelled (label())


def _is_front(n):
    return "densify_rows_kernel" in n or "sparse_front_kernel" in n or "front_kernel" in n


def _starts_staged_pass(disp, i):
    """A staged discriminator pass (stage_pass): its steps have no front launch of their own, so no front-like kernel lies between the
    first two d_coef launches behind the pass's row expansion."""
    seen = 0
    for x in disp[i + 1:i + 40]:
        n = x["name"]
        if "d_coef_kernel" in n or "de_dcoef_kernel" in n:
            seen += 1
            if seen == 2:
                return True
        elif _is_front(n) or "adam_rows" in n or "pair_kernel<" in n:
            return False
    return False


def label(disp):
    """[(class label, dispatch)] for every dispatch that belongs to a complete training step."""
    steps, cur = [], None
    out_pass = []
    staged = False          # inside a staged discriminator pass (stage_pass: no front launch per step; open_steps_kernel starts it)
    after_dcoef = None      # staged pass: GEMM-like launches seen since the step's d_coef launch
    need = 2
    expect_pass_gemm = False
    for i, d in enumerate(disp):
        n = d["name"]
        if "open_steps_kernel" in n:      # a later pass of a call (the first pass's lr_t table rides in the pass's row-expansion launch)
            gpass = any("adam_rows_advance" in x["name"] for x in disp[i + 1:i + 2])
            if not gpass:
                staged, cur, after_dcoef = True, None, None
            out_pass.append((("Q" if gpass else "P") + ":lr_t of the pass's steps", d))
            continue
        if "adam_rows_advance" in n or "adam_rows_flush" in n:      # the all-rows Adam over U of a whole generator pass (lazy_pass_begin / _end)
            out_pass.append(("Q:all-rows Adam over U, %s" % ("rows advanced to their step (front of the pass)" if "advance" in n else "every row through the pass (end of the pass)"), d))
            if "flush" in n:
                cur = None
            continue
        if "densify_rows_kernel" in n and _starts_staged_pass(disp, i):
            out_pass.append(("P:CSR rows of the whole pass (+ lr_t of its steps)", d))      # (stage_pass: row expansion, then the generated rows)
            cur, expect_pass_gemm = None, True
            staged, after_dcoef = True, None
            continue
        if expect_pass_gemm and "gemm_" in n:
            out_pass.append(("P:generated rows of the whole pass (batched)", d))
            expect_pass_gemm = False
            continue
        if "densify_rows_kernel" in n or "sparse_front_kernel" in n or "front_kernel" in n:
            staged = False
            cur = [d]
            steps.append(cur)
        elif staged:
            if cur is None:
                if "persist" in n or "finish_parts" in n or "rocclr" in n or "mask_topk" in n:
                    staged = False
                    continue
                cur = [d]
                steps.append(cur)
                after_dcoef = None
            else:
                cur.append(d)
            if "d_coef_kernel" in n or "de_dcoef_kernel" in n:
                after_dcoef = 0
                need = 2 if "de_dcoef_kernel" in n else 3
            elif after_dcoef is not None and _gemm_like(n):
                after_dcoef += 2 if "wgrad_pair_kernel" in n else 1
                if after_dcoef >= need:
                    cur, after_dcoef = None, None      # the step's last launch
        elif cur is not None:
            if "persist" in n or "mask_topk" in n or "gather_rows" in n or "finish_parts" in n or "rocclr" in n:
                cur = None      # scoring / epoch end: not part of a step
            else:
                cur.append(d)
    out = list(out_pass)
    for d in disp:      # the scoring GEMM of the bench (ganmf_bench_scores): its own classes, outside the steps
        n = d["name"]
        if "gemm_f32_persist" in n:
            out.append(("S:scoring 6040x3706x250 (fp32 MFMA, persistent)", d))
        elif "gemm_bf16s_mfma<128, 128, 32, false, false, 3" in n:
            out.append(("S:scoring 6040x3706x250 (split-bf16)", d))
        elif "gemm_bf16p_persist" in n:
            out.append(("S:scoring 6040x3706x250 (split-bf16, pre-split planes, persistent)", d))
        elif "presplit_rows_kernel" in n:
            out.append(("S:scoring split pass (both factors -> bf16 x 3 planes)", d))
    for st in steps:
        is_d = any("d_coef_kernel" in d["name"] or "de_dcoef_kernel" in d["name"] for d in st)
        paired = any("pair_kernel" in d["name"] and "wgrad" not in d["name"] for d in st)
        wpaired = any("wgrad_pair_kernel" in d["name"] for d in st)
        names = list(D_GEMMS if is_d else G_GEMMS)
        has_front = any("front_kernel" in d["name"] or "densify_rows" in d["name"] or "sparse_front" in d["name"] for d in st)
        if paired:
            names = names[:5] + ["gUb+gV+adam"]
        if wpaired:
            names = names[:4] + ["gWd+gWe+adam"]
        if not has_front:
            names = names[1:]      # a step of a staged pass: rows and generated rows were formed in front of the pass
        gi, last = 0, None
        kind = "D" if is_d else "G"
        if sum(1 for d in st if _gemm_like(d["name"])) != len(names):
            continue      # a truncated step at the edge of the trace
        for d in st:
            n = d["name"]
            if "front_kernel" in n and "sparse" not in n:
                lab = "gen+rows"; gi += 1; last = "gen"          # generator GEMM + CSR row expansion in one launch
            elif "densify" in n or "sparse_front" in n:
                lab = "densify+gather"
            elif "de_dcoef_kernel" in n:
                lab = names[gi] + "+d_coef"; last = names[gi]; gi += 1    # the hinge scalars / Es ride in the dE launch
            elif "gemm_bf16s_red" in n:
                lab = names[gi] + "+reduce(%s)" % last; gi += 1  # the slab sum of the previous product rides in this launch
            elif _gemm_like(n):
                lab = names[gi]; gi += 1; last = lab
            elif "splitk_reduce" in n:
                lab = "reduce(%s)" % last
            elif "d_coef" in n:
                lab = "d_coef"
            elif "adam_rows" in n:
                lab = "adam_rows_U"
            elif "adam_dense" in n:
                lab = "adam_dense"
            else:
                lab = n.split("(")[0][-30:]
            out.append((kind + ":" + lab, d))
        COUNTS[kind] += 1
    return out


def main():
    disp = load(sys.argv[1])
    lab = label(disp)
    acc = collections.OrderedDict()
    for k, d in lab:
        a = acc.setdefault(k, {"n": 0, "us": 0.0, "c": collections.defaultdict(float), "kern": d["name"], "grid": d["grid"],
                               "lds": d["lds"], "wgsize": d["wgsize"]})
        a["n"] += 1
        if d["start"] is not None:
            a["us"] += (d["end"] - d["start"]) / 1e3
        for cn, cv in d["counters"].items():
            a["c"][cn] += cv
    cnames = sorted({cn for a in acc.values() for cn in a["c"]})
    # derived columns when the SQ pass is present.  SQ_BUSY_CYCLES is summed over the 32 shader engines (8 XCDs x 4) and
    # SQ_VALU_MFMA_BUSY_CYCLES over the 1024 SIMDs (256 CUs x 4): clock = busy cycles per SE / duration, MFMA utilisation
    # = MFMA-busy cycles per SIMD / busy cycles per SE
    derived = "SQ_BUSY_CYCLES" in cnames and "SQ_VALU_MFMA_BUSY_CYCLES" in cnames
    per = {"D": 0.0, "G": 0.0, "P": 0.0, "Q": 0.0}      # total us per kind (P / Q: launches in front of / around a discriminator / generator pass)
    print("| step:class | kernel | workgroups | launches | avg us |" + "".join(" %s |" % c for c in cnames) +
          (" clock GHz | MFMA busy % |" if derived else ""))
    print("|---|---|---|---|---|" + "---|" * (len(cnames) + (2 if derived else 0)))
    for k, a in acc.items():
        kern = a["kern"].replace("void ganmf::", "").replace("ganmf::", "").split("(")[0][:48]
        wg = ""
        try:
            wg = "%d x %s" % (int(a["grid"]) // int(a["wgsize"]), a["wgsize"])
        except (TypeError, ValueError):
            pass
        avg = a["us"] / a["n"]
        if k[0] in per:
            per[k[0]] += a["us"]
        extra = ""
        if derived:
            busy_se = a["c"]["SQ_BUSY_CYCLES"] / a["n"] / 32.0
            extra = " %.2f | %.1f |" % (busy_se / max(avg, 1e-9) / 1e3, 100.0 * a["c"]["SQ_VALU_MFMA_BUSY_CYCLES"] / a["n"] / 1024.0 / max(busy_se, 1.0))
        print("| %s | `%s` | %s | %d | %.2f |" % (k, kern, wg, a["n"], avg) + "".join(" %.4g |" % (a["c"][c] / a["n"]) for c in cnames) + extra)
    nd, ng = max(COUNTS["D"], 1), max(COUNTS["G"], 1)
    dstep = (per["D"] + per["P"]) / nd      # the per-pass launches belong to the steps of their pass
    gstep = (per["G"] + per["Q"]) / ng
    print("\nD-step %.1f us%s, G-step %.1f us%s, D+G pair %.1f us (kernel time of the labelled dispatches / steps: %d D, %d G; %d labelled dispatches)" % (
        dstep, " (of which %.1f us per step in front of the pass)" % (per["P"] / nd) if per["P"] else "",
        gstep, " (of which %.1f us per step around the pass)" % (per["Q"] / ng) if per["Q"] else "", dstep + gstep,
        COUNTS["D"], COUNTS["G"], len(lab)))

if __name__ == "__main__":
    main()
