"""Timing-only decomposition of the persistent scoring GEMM (GANMF_PERSIST_DIAG bits; results are wrong by design).
Needs a diagnostic build of the library: `make -C ganmf_amd/csrc clean && make -C ganmf_amd/csrc DIAG=1`; the shipped
library ignores the variable."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.score_bench import run  # noqa: E402

if __name__ == "__main__":
    M, N, K = 6040, 3706, 250
    for waves in ("1",):
        for diag, what in ((0, "full"), (1, "no C stores"), (3, "no stores, no staging dump"), (4, "no K-tile refills"),
                           (7, "MFMA + fragment reads + barriers only")):
            ms = run(M, N, K, {"GANMF_MFMA": "f32", "GANMF_PERSIST": waves, "GANMF_PERSIST_DIAG": str(diag)})
            print("persist %s waves, diag %d (%-38s): %7.1f us" % ("8" if waves == "1" else "4", diag, what, ms * 1e3), flush=True)
