"""The `make ieee` build (correctly rounded sqrtf / divide in every Adam kernel, TF ApplyAdam's own last line; the product library uses
v_sqrt_f32 / v_rcp_f32, gemm_f32.hpp adam_step) stays a passing library: loaded through GANMF_LIB_PATH in a child process (a process maps
one library), it runs the per-step parity cases against the fp64 oracle at their unchanged tolerances and must agree with the product
build to a few ulp of the step.  INTEGRATION.md section D: checkpoints of the two builds are not bit-comparable."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
IEEE_LIB = os.path.join(ROOT, "ganmf_amd", "libganmf_hip_ieee.so")

CHILD = r"""
import json, sys
import numpy as np
sys.path.insert(0, %(root)r)
import tests.test_gpu_parity as P
from oracle.ganmf_oracle import GANMFOracle      # checker only
P.test_single_steps_match_oracle((37, 53, 5, 7, 8), 10.0)
P.test_single_steps_match_oracle((300, 517, 33, 65, 64), 0.001)
P.test_adam_moments_after_steps()
# three epochs from fixed weights: the tensors this build ends with (the parent compares the two builds)
rng = np.random.RandomState(11)
U, N, k, e, B = 300, 517, 33, 65, 64
urm = P._rand_urm(rng, U, N, 0.05)
o = GANMFOracle(U, N, k, e, seed=5, **P.HP)
eng = P._engine_from_oracle(o, urm, B, P.HP)
for ep in range(3):
    eng.train_epoch(np.random.RandomState(ep).permutation(U), 1, 1)
out = {n: P._get(eng, n).astype(np.float64).ravel()[:4096].tolist() for n in P.NAME2ID}
eng.close()
import ganmf_amd._lib as L
print("RESULT " + json.dumps({"lib": L.library_path(), "tensors": out}))
"""


def _run(lib_path):
    env = dict(os.environ)
    if lib_path:
        env["GANMF_LIB_PATH"] = lib_path
    else:
        env.pop("GANMF_LIB_PATH", None)
    res = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}], env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    line = [l for l in res.stdout.splitlines() if l.startswith("RESULT ")][-1]
    return json.loads(line[len("RESULT "):])


@pytest.mark.gpu
def test_adam_ieee_build_passes_parity_and_agrees_with_the_product_build():
    if not os.path.exists(IEEE_LIB):
        pytest.skip("ganmf_amd/libganmf_hip_ieee.so not built (make -C ganmf_amd/csrc ieee; __graft_entry__.build() does)")
    ieee, fast = _run(IEEE_LIB), _run(None)
    assert ieee["lib"] == IEEE_LIB and fast["lib"] != IEEE_LIB
    differ = 0
    for n, a in ieee["tensors"].items():
        a, b = np.asarray(a), np.asarray(fast["tensors"][n])
        scale = np.abs(b).max() + 1e-30
        # 3 epochs x 5 updates, lr <= 2e-3: each update's last line differs by <= 2.5 ulp of a step of at most lr (3e-7 relative)
        assert np.abs(a - b).max() <= 2e-6 * scale, (n, float(np.abs(a - b).max() / scale))
        differ += int(np.any(a != b))
    assert differ > 0, "the two builds are bit-identical: the IEEE switch is not compiled in"
