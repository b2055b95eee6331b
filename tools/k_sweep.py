"""K sweep of one 16-wave kernel at a one-workgroup-per-CU grid: slope (us per 64-wide K-tile) and intercept (fixed part of a launch).
    python tools/k_sweep.py [M N [layouts [diags]]]      e.g.  python tools/k_sweep.py 256 3706 NN,NT 0,2,4,16,22
Runs the 16-wave split-bf16 kernel (gemm_bf16k.hpp) -- and with diags "f32" the 16-wave fp32 ring kernel -- back to back on one stream.
`diags` (a library built with `make DIAG=1` only; timing-only variants, wrong results): GANMF_BF16K_DIAG bits 2 no splits / plane stores
after the prologue, 4 no MFMAs, 16 no operand fetches after the prologue."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if __name__ == "__main__":
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 3706
    layouts = sys.argv[3].split(",") if len(sys.argv) > 3 else ["NN", "NT", "TN"]
    diags = sys.argv[4].split(",") if len(sys.argv) > 4 else ["0"]
    base = os.environ.get("GANMF_TUNE", "")
    os.environ["GANMF_MFMA"] = "f32"
    os.environ["GANMF_TUNE"] = (base + "," if base else "") + "kg=4,ring=3"
    from ganmf_amd.engine import gemm_f32
    rng = np.random.RandomState(0)
    Ks = [256, 512, 1024, 2048, 3712]
    LAY = {"NN": (False, True), "NT": (False, False), "TN": (True, True)}
    for dg in diags:
        os.environ["GANMF_X3KG"] = "0" if dg == "f32" else "1"
        os.environ["GANMF_BF16K_DIAG"] = "0" if dg == "f32" else dg
        for layout in layouts:
            akm, bkm = LAY[layout]
            ts = []
            for K in Ks:
                A = rng.standard_normal((K, M) if akm else (M, K)).astype(np.float32)
                B = rng.standard_normal((K, N) if bkm else (N, K)).astype(np.float32)
                gemm_f32(A, B, akm, bkm, tile=64, nsplit=1, iters=50)
                _, ms = gemm_f32(A, B, akm, bkm, tile=64, nsplit=1, iters=300)
                ts.append(ms * 1e3)
            tiles = np.array(Ks) / 64.0
            slope, icpt = np.polyfit(tiles, np.array(ts), 1)
            print("%s diag %s %s %dx%d: " % ("f32kg4" if dg == "f32" else "bf16k", dg, layout, M, N) +
                  "  ".join("K=%d %.2f" % (k, t) for k, t in zip(Ks, ts)) + "   -> %.3f us per K-tile, %.2f us fixed" % (slope, icpt), flush=True)
