#!/usr/bin/env python3
"""Build-time lint for the inline-asm operand prefetch of gemm_bf16k.hpp (advisor finding, round 3).

bf16k_mainloop (and bf16w_mainloop, gemm_bf16w.hpp) issues its operand loads from `asm volatile("global_load_dwordx4 ...")` and waits for them with hand-counted
`s_waitcnt vmcnt(n)`: hipcc does not know that those registers have a load in flight, so nothing but the source's token
dependencies keeps it from copying or reading one before its wait -- a silent wrong-result bug that depends on the compiler
version.  This tool compiles the kernel instantiations to gfx950 assembly (no GPU needed) and walks every kernel with a model of
the vector-memory counter: every VMEM instruction enters a FIFO, `s_waitcnt vmcnt(n)` retires all but the n youngest, and an
instruction that READS a register whose asm-issued load is still in the FIFO is reported.

    python tools/check_asm_prefetch.py            # exit code 0: clean; 1: a read of an in-flight prefetch register
    python tools/check_asm_prefetch.py --stamp F  # what the Makefile's default target runs: F holds the sha-256 of the kernel headers, this tool
                                                  # and `hipcc --version` of the last CLEAN run -- same hash: nothing to do (exit 0); else the lint
                                                  # runs and F is (re)written only when it is clean

The walk is linear over the listing (one pass over each loop body; the FIFO is carried into a block that is entered by falling
through and forgotten behind an unconditional branch): it sees the pattern the finding is about -- a move or use placed between a load and the wait that covers it -- in the prologue, the loop body and the
epilogue paths; it is a lint, not a proof.
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

SRC = """#include <hip/hip_runtime.h>
#include "%s/ganmf_amd/csrc/gemm_multi.hpp"
namespace ganmf {
template __global__ void gemm_bf16k_mfma<false, false, 3, false>(const GemmP);
template __global__ void gemm_bf16k_mfma<false, true, 3, false>(const GemmP);
template __global__ void gemm_bf16k_mfma<true, true, 3, false>(const GemmP);
template __global__ void gemm_bf16k_mfma<false, false, 1, false>(const GemmP);
template __global__ void gemm_bf16k_mfma<false, true, 1, true>(const GemmP);
template __global__ void front_kernel<4, true>(const GemmP, const DensP);
template __global__ void gemm_bf16w_mfma<false>(const GemmP);
template __global__ void gemm_bf16w_mfma<true>(const GemmP);
template __global__ void gemm_bf16w_mfma<false, 1, true>(const GemmP);
template __global__ void gemm_bf16w_mfma<true, 1, true>(const GemmP);
template __global__ void gemm_bf16w_mfma<false, 1, false>(const GemmP);
template __global__ void front_lp_kernel<true>(const GemmP, const DensP);
template __global__ void front_lp_kernel<false>(const GemmP, const DensP);
template __global__ void pair_lp_kernel<true>(const GemmP, const GemmP);
template __global__ void pair_lp_kernel<false>(const GemmP, const GemmP);
template __global__ void de_dcoef_kernel<4, true>(const GemmP, const DCoefP, const int);
}
""" % ROOT

REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
VMEM = re.compile(r"^(global_load|global_store|global_atomic|buffer_load|buffer_store|buffer_atomic|flat_load|flat_store)")
WAIT = re.compile(r"s_waitcnt\b.*vmcnt\((\d+)\)")


def regs(text):
    out = []
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.append((int(m.group(1)), int(m.group(1))))
        else:
            out.append((int(m.group(2)), int(m.group(3))))
    return out


def overlaps(a, b):
    return a[0] <= b[1] and b[0] <= a[1]


def check_kernel(name, lines):
    fifo = []          # entries: (dest range or None, asm_issued, line number)
    in_asm = False
    problems = []
    for ln, raw in lines:
        text = raw.split(";")[0].strip() if not raw.strip().startswith(";;#") else raw.strip()
        if text.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if text.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not text or text.endswith(":") or text.startswith("."):
            continue
        m = WAIT.search(text)
        if m:
            n = int(m.group(1))
            if len(fifo) > n:
                fifo = fifo[len(fifo) - n:] if n else []
            continue
        if text.startswith("s_waitcnt"):
            continue
        if text.startswith("s_branch") or text.startswith("s_endpgm") or text.startswith("s_setpc"):
            fifo = []      # the next block is not entered by falling through: nothing is known about its queue (no report there)
            continue
        op = text.split()[0]
        operands = text[len(op):]
        rs = regs(operands)
        is_store = "store" in op or op.startswith("ds_write") or op.startswith("global_atomic")
        writes_first = bool(rs) and not is_store and not op.startswith("s_") and not op.startswith("v_cmp") and not op.startswith("v_cmpx")
        reads = rs[1:] if writes_first else rs
        pending = [f for f in fifo if f[0] is not None and f[1]]
        for r in reads:
            for f in pending:
                if overlaps(r, f[0]):
                    problems.append("%s: line %d `%s` reads v[%d:%d], whose asm-issued load (line %d) may still be in flight"
                                    % (name, ln, text, f[0][0], f[0][1], f[2]))
        if VMEM.match(op):
            dest = rs[0] if (rs and "load" in op and "lds" not in op) else None
            fifo.append((dest, in_asm, ln))
    return problems


def input_hash():
    import glob
    import hashlib
    hs = hashlib.sha256()
    ver = subprocess.run([HIPCC, "--version"], capture_output=True, text=True)
    hs.update((ver.stdout if ver.returncode == 0 else "no hipcc").encode())
    for f in sorted(glob.glob(os.path.join(ROOT, "ganmf_amd", "csrc", "*.hpp"))) + [os.path.abspath(__file__)]:
        hs.update(os.path.basename(f).encode() + b"\0")
        hs.update(open(f, "rb").read())
    return hs.hexdigest()


def main():
    stamp = None
    if "--stamp" in sys.argv:
        stamp = sys.argv[sys.argv.index("--stamp") + 1]
        digest = input_hash()
        if os.path.exists(stamp) and open(stamp).read().strip() == digest:
            return 0
    rc = run_lint()
    if stamp:
        if rc == 0:
            open(stamp, "w").write(digest + "\n")
        elif os.path.exists(stamp):
            os.remove(stamp)
    return rc


def run_lint():
    with tempfile.TemporaryDirectory() as d:
        src, asm = os.path.join(d, "k.hip"), os.path.join(d, "k.s")
        open(src, "w").write(SRC)
        res = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-w", src, "-o", asm],
                             capture_output=True, text=True)
        if res.returncode != 0:
            sys.stderr.write(res.stderr[-3000:])
            return 2
        text = open(asm).read().split("\n")
    kernels, cur, name = {}, None, None
    for i, line in enumerate(text, 1):
        m = re.match(r"^(_ZN5ganmf\w+):", line)
        if m:
            name, cur = m.group(1), []
            kernels[name] = cur
            continue
        if cur is not None:
            cur.append((i, line))
            if "s_endpgm" in line:
                cur = None
    problems, checked, asm_loads = [], 0, 0
    for name, lines in kernels.items():
        if not any(k in name for k in ("bf16k", "bf16w", "front_kernel", "de_dcoef", "front_lp", "pair_lp")):
            continue
        n_asm = sum(1 for _, l in lines if "global_load_dwordx4" in l)
        if n_asm == 0:
            continue
        checked += 1
        asm_loads += n_asm
        problems += check_kernel(name, lines)
    print("checked %d kernels (%d prefetch load sites): %d reads of an in-flight prefetch register" % (checked, asm_loads, len(problems)))
    for p in problems[:20]:
        print("  " + p)
    return 1 if problems or checked == 0 else 0


if __name__ == "__main__":
    sys.exit(main())
