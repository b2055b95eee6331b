#!/usr/bin/env python3
"""HBM traffic per launch of every kernel class of the bench step from rocprofv3 PMC passes.

    tools/collect_traffic.py FETCH_DIR WRITE_DIR [B N k e U] > profiles/rNN_traffic.json

FETCH_DIR / WRITE_DIR are the outputs of two separate `rocprofv3 --kernel-trace --pmc FETCH_SIZE` / `--pmc WRITE_SIZE`
passes over `python3 bench.py --no-cpu-baseline` (the two counters do not fit one pass, MI355X_MICROARCH §rocprofv3).
Dispatches are labelled by their ORDER inside a step (tools/step_classes.py), so classes that share a kernel
instantiation and grid (decode of the D-step, decode of the G-step, gUb) are kept apart.
Corrections per MI355X_MICROARCH §HBM: counters are in KiB; on gfx950 FETCH_SIZE reports exactly half the bytes of wide
coalesced reads (16 B/lane global_load and global_load_lds alike) -> doubled.
`algorithmic_bytes` = every operand read once, every result written once, fp32: 4(MK + KN + MN) per GEMM (+ the
epilogue's second operand where it has one), and for the fused-Adam weight-gradient GEMMs the SIX streams of the update
(theta, m, v read and written: 24 MN; the gradient itself never reaches HBM)."""
import collections
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from step_classes import label, load  # noqa: E402

TAG = {"gen": "gemm_generator[B,k]x[N,k]^T", "encode": "gemm_encode[2B,N]x[N,e]", "decode": "gemm_decode[2B,e]x[e,N]",
       "dE": "gemm_dE[2B,N]x[e,N]^T", "gWd+adam": "gemm_gWd[2B,e]^Tx[2B,N]", "gWe+adam": "gemm_gWe[2B,N]^Tx[2B,e]",
       "dF": "gemm_dF[B,e]x[N,e]^T", "gUb": "gemm_gUb[B,N]x[N,k]", "gV+adam": "gemm_gV[B,N]^Tx[B,k]",
       "densify+gather": "densify_rows+gather", "d_coef": "d_coef+scale", "adam_rows_U": "adam_rows_U",
       "gen+rows": "gemm_generator[B,k]x[N,k]^T + CSR rows", "gUb+gV+adam": "gemm_gUb[B,N]x[N,k] + gemm_gV[B,N]^Tx[B,k]",
       "gWd+adam+reduce(dE)": "gemm_gWd[2B,e]^Tx[2B,N] + reduce_dE", "dE+d_coef": "gemm_dE[2B,N]x[e,N]^T + d_coef",
       "gWd+gWe+adam": "gemm_gWd + gemm_gWe, fused Adam"}


def algorithmic(cls, B, N, k, e, U, wgs=0):
    step, name = cls.split(":")
    g = lambda M, Nn, K: 4 * (M * K + K * Nn + M * Nn)
    ldk = (k + 1 + 63) // 64 * 64
    if step == "P" and name.startswith("CSR rows"): return 4 * wgs * N                 # one workgroup per scheduled row: the X row written once
    if step == "P" and name.startswith("generated rows"):                              # batches of ceil(B/64) x ceil(N/64) tiles
        return (wgs // max(((B + 63) // 64) * ((N + 63) // 64), 1)) * g(B, N, k)
    if step == "Q" and "advanced" in name: return 16 * (wgs * 256 // max(ldk // 4, 1)) * k    # theta, m, v read, theta written, per scheduled row
    if step == "Q" and "every row" in name: return 24 * U * k                                   # one sweep of theta, m, v per PASS
    if name == "gen": return g(B, N, k)
    if name == "gen+rows": return g(B, N, k) + 4 * B * (N + 2 * k)
    if name == "dE+d_coef": return g(2 * B, e, N) + 8 * 2 * B * e
    if name == "gUb+gV+adam": return g(B, k, N) + 4 * (B * N + B * k) + 24 * N * k
    if name.startswith("gWd+adam+reduce"): return 4 * (2 * B * (e + 1) + 2 * B * N) + 24 * (e + 1) * N + 4 * 2 * B * e
    if name == "gWd+gWe+adam":
        return 4 * (2 * B * (e + 1) + 2 * B * N) + 24 * (e + 1) * N + 4 * (2 * B * (N + 1) + 2 * B * e) + 24 * (N + 1) * e
    if name == "encode": return g(2 * B, e, N + 1)
    if name == "decode": return (2 if step == "D" else 1) * (4 * (B * (e + 1) + 2 * B * N)) + 4 * (e + 1) * N   # + the subtracted input
    if name == "dE": return g(2 * B, e, N) if step == "D" else g(B, e, N) + 8 * B * e
    if name == "gWd+adam": return 4 * (2 * B * (e + 1) + 2 * B * N) + 24 * (e + 1) * N
    if name == "gWe+adam": return 4 * (2 * B * (N + 1) + 2 * B * e) + 24 * (N + 1) * e
    if name == "dF": return g(B, N, e) + 4 * B * N
    if name == "gUb": return g(B, k, N)
    if name == "gV+adam": return 4 * (B * N + B * k) + 24 * N * k
    if name == "adam_rows_U": return 24 * U * k + 4 * B * k
    if name == "densify+gather": return 4 * B * (N + 2 * k)
    if name == "d_coef": return 8 * 2 * B * e
    if name.startswith("scoring split pass"):      # fp32 in once, three bf16 planes out (K padded to 32, rows to 128)
        kp = (k + 31) // 32 * 32
        return 4 * (U * k + N * k) + 6 * kp * ((U + 127) // 128 * 128 + (N + 127) // 128 * 128)
    if name.startswith("scoring"):      # S-step rows of the bench's side measurement: U[6040, k] . V[N, k]^T -> [U, N], every operand once
        return 4 * (U * k + N * k + U * N)
    if name.startswith("reduce("):
        inner = name[7:-1]
        M, Nn = {"encode": (2 * B, e), "decode": (B, N), "dE": ((2 * B if step == "D" else B), e), "dF": (B, N), "gUb": (B, k)}.get(inner, (0, 0))
        return 4 * M * Nn     # + nsplit slabs read, which are not algorithmic
    return 0


def per_class(d, counter):
    f = max(glob.glob(d + "/*/*counter_collection.csv"), key=os.path.getmtime)   # newest pass if the directory was reused
    acc = collections.OrderedDict()
    for cls, disp in label(load(f)):
        a = acc.setdefault(cls, {"n": 0, "v": 0.0, "kernel": disp["name"], "wgs": int(disp["grid"] or 0) // int(disp["wgsize"] or 256)})
        a["n"] += 1
        a["v"] += disp["counters"].get(counter, 0.0)
    return acc


def main():
    fdir, wdir = sys.argv[1:3]
    B, N, k, e, U = [int(x) for x in sys.argv[3:8]] if len(sys.argv) >= 8 else (128, 3706, 250, 992, 6040)
    fetch, write = per_class(fdir, "FETCH_SIZE"), per_class(wdir, "WRITE_SIZE")
    out = collections.OrderedDict()
    for cls, f in fetch.items():
        w = write.get(cls, {"n": 1, "v": 0.0})
        f_kib, w_kib = f["v"] / f["n"], w["v"] / max(w["n"], 1)
        step, name = cls.split(":")
        tag = TAG.get(name, name)
        out["%s (%s-step)" % (tag, step)] = {
            "class": cls, "kernel": f["kernel"].replace("void ganmf::", "").split("(")[0], "launches_sampled": f["n"],
            "FETCH_SIZE_KiB_raw": round(f_kib, 1), "WRITE_SIZE_KiB_raw": round(w_kib, 1),
            "hbm_bytes_per_launch": int((2 * f_kib + w_kib) * 1024),
            "algorithmic_bytes": int(algorithmic(cls, B, N, k, e, U, f.get("wgs", 0)))}
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
