"""`python bench.py` on the GPU: ONE JSON line on stdout carrying the driver's contract keys, SURVEY 8(d)'s full report (D / G pass rates, the one-thread CPU
figure) and a `roofline` object whose `frac` is the dominant device function over all of its launch classes -- checked for internal consistency, not for speed."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_line_contract_and_consistency():
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "12", "--warmup", "4", "--cpu-seconds", "1.5"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, "stdout must carry exactly one line"
    d = json.loads(lines[0])
    for key, val in (("metric", "GANMF training steps/sec"), ("unit", "steps/s"), ("n_gpus", 1), ("steps", 12), ("warmup", 4), ("higher_is_better", True),
                     ("scaling", "weak"), ("vs_baseline", None), ("dtype", "f32"), ("data", "synthetic")):
        assert d[key] == val, (key, d[key])
    assert "workload" in d["config"] and "6040x3706" in d["config"]["workload"] and "model" not in d["config"]
    assert d["value"] > 0 and abs(d["ms_per_step"] - 1e3 / d["value"]) < 1e-3 * d["ms_per_step"] + 1e-4
    # the two passes apart: each rate is a rate of the same step, so the whole-epoch value lies between them (within the spread of short runs)
    lo, hi = sorted((d["d_steps_per_s"], d["g_steps_per_s"]))
    assert lo > 0 and 0.85 * lo <= d["value"] <= 1.15 * hi, (d["value"], lo, hi)
    assert d["d_steps_per_s"] < d["g_steps_per_s"]            # 9.65 GFLOP + 186 MB of Adam streams against 5.42 GFLOP
    assert len(d["pass_rates"]["d_samples"]) == 5 and len(d["pass_rates"]["g_samples"]) == 5
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 157.3 and r["kernel"].startswith("gemm_bf16k_mfma<false, true, 3, false>")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0.3 < r["frac"] < 1.0
    assert {(c["step"], c["name"][:11]) for c in r["classes"]} == {("D", "gemm_encode"), ("G", "gemm_encode"), ("D", "gemm_decode")}
    assert min(c["frac"] for c in r["classes"]) <= r["frac"] <= r["frac_best_class"] == max(c["frac"] for c in r["classes"])
    fl = sum(c["achieved"] * c["avg_launch_us"] * c["launches"] for c in r["classes"])      # TFLOP/s x us x launches = MFLOP
    t = sum(c["avg_launch_us"] * c["launches"] for c in r["classes"])
    assert abs(fl / t - r["achieved"]) < 0.02 * r["achieved"]
    assert abs(r["flops_per_launch"] - 2.0 * 256 * 992 * 3707) < 0.01 * r["flops_per_launch"]      # encode and the D-step decode: 1.883 / 1.885 GFLOP
    assert r["traffic"] is None or r["traffic"] >= 0.9 * r["algorithmic_bytes_per_launch"]
    fa = d["roofline_fused_adam"]
    assert fa["bound"] == "hbm" and fa["peak"] == 8000.0 and 0.2 < fa["frac"] < 1.0 and abs(fa["algorithmic_bytes_per_step"] - 186.2e6) < 1e6
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "steps/s" and c["value"] > 0 and c["cores"] >= 1 and c["value_1thread"] > 0
    assert {k["step"] for k in d["kernels"]} == {"D", "G"}
