"""Randomised shapes through ganmf_gemm_f32 in every layout, tile, split and K-loop arithmetic: row / column / K
tails that are not multiples of any tile dimension, single-row and single-column outputs, K shorter than one K-tile,
splits deeper than the K range.  Each arithmetic mode runs in its own process (GANMF_MFMA is read at plan time)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_SNIPPET = r"""
import json, sys
import numpy as np
sys.path.insert(0, %r)
from ganmf_amd.engine import gemm_f32
rng = np.random.RandomState(%d)
worst, n = 0.0, 0
special = [(1, 1, 1), (1, 257, 3), (257, 1, 65), (64, 64, 64), (65, 65, 65), (128, 128, 32), (129, 127, 33), (63, 200, 1), (300, 5, 700)]
cases = special + [(int(rng.randint(1, 301)), int(rng.randint(1, 301)), int(rng.randint(1, 701))) for _ in range(%d)]
for (M, N, K) in cases:
    akm, bkm = [(0, 0), (0, 1), (1, 1)][rng.randint(3)]
    tile = [0, 64, 128][rng.randint(3)]
    nsplit = [0, 1, 2, 3, 7, 16][rng.randint(6)]
    A = rng.standard_normal((K, M) if akm else (M, K)).astype(np.float32)
    B = rng.standard_normal((K, N) if bkm else (N, K)).astype(np.float32)
    C, _ = gemm_f32(A, B, bool(akm), bool(bkm), tile=tile, nsplit=nsplit)
    a = (A.T if akm else A).astype(np.float64); b = (B if bkm else B.T).astype(np.float64)
    ref, bound = a @ b, np.abs(a) @ np.abs(b)
    assert C.shape == ref.shape, (M, N, K, C.shape)
    err = float(np.max(np.abs(C - ref) / (bound + 1e-300)))
    if not np.isfinite(err) or err > %g:
        print(json.dumps({"fail": [M, N, K, akm, bkm, tile, nsplit, err]})); sys.exit(0)
    worst = max(worst, err); n += 1
print(json.dumps({"cases": n, "worst": worst}))
"""


@pytest.mark.parametrize("mode,bound", [("f32", 1e-6), ("bf16x3", 1e-6), ("bf16", 8e-3)])
def test_random_shapes(mode, bound):
    env = dict(os.environ, GANMF_MFMA=mode)
    r = subprocess.run([sys.executable, "-c", _SNIPPET % (ROOT, 1234, 250, bound)], env=env, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    res = json.loads(r.stdout.strip().splitlines()[-1])
    assert "fail" not in res, (mode, res)
    assert res["cases"] == 259, res
