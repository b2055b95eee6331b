"""`python bench.py --gpus N` without a launcher starts its own N rank processes (bench.launch_ranks) before touching
torch or the GPU: environment of the ranks, relay of rank 0's single JSON line, failure handling.  Stand-in rank scripts;
no GPU."""
import importlib.util
import json
import os
import sys
import textwrap
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_importing_bench_touches_neither_torch_nor_the_library():
    code = ("import sys, importlib.util as u; s = u.spec_from_file_location('b', %r); m = u.module_from_spec(s); "
            "s.loader.exec_module(m); assert 'torch' not in sys.modules and 'ganmf_amd' not in sys.modules" % os.path.join(ROOT, "bench.py"))
    import subprocess
    assert subprocess.run([sys.executable, "-c", code]).returncode == 0


def test_launcher_sets_rank_environment_and_relays_rank0(tmp_path, capfd):
    script = tmp_path / "rank.py"
    script.write_text(textwrap.dedent("""
        import json, os, sys
        env = {k: os.environ[k] for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "HSA_ENABLE_IPC_MODE_LEGACY")}
        open(os.path.join(%r, "rank%%s.json" %% env["RANK"]), "w").write(json.dumps({"env": env, "argv": sys.argv[1:]}))
        if env["RANK"] == "0":
            print(json.dumps({"metric": "stub", "n_gpus": int(env["WORLD_SIZE"])}))
        else:
            print("noise from rank", env["RANK"])          # must not reach the launcher's stdout
    """ % str(tmp_path)))
    b = _bench()
    b.launch_ranks(3, ["--gpus", "3", "--steps", "4"], script=str(script))
    out = capfd.readouterr().out.strip().splitlines()
    assert len(out) == 1 and json.loads(out[0]) == {"metric": "stub", "n_gpus": 3}
    seen = [json.load(open(tmp_path / ("rank%d.json" % r))) for r in range(3)]
    ports = {s["env"]["MASTER_PORT"] for s in seen}
    assert len(ports) == 1
    for r, s in enumerate(seen):
        assert s["env"]["RANK"] == s["env"]["LOCAL_RANK"] == str(r) and s["env"]["WORLD_SIZE"] == "3"
        assert s["env"]["MASTER_ADDR"] == "127.0.0.1" and s["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
        assert s["argv"] == ["--gpus", "3", "--steps", "4"]


def test_launcher_fails_and_ends_the_peers_when_a_rank_dies(tmp_path):
    script = tmp_path / "rank.py"
    script.write_text(textwrap.dedent("""
        import os, sys, time
        if os.environ["RANK"] == "1":
            sys.exit(3)
        time.sleep(600)          # a peer waiting in a collective for a rank that is gone
    """))
    b = _bench()
    t0 = time.time()
    with pytest.raises(SystemExit) as ex:
        b.launch_ranks(2, [], script=str(script))
    assert ex.value.code == 1
    assert time.time() - t0 < 60


def test_parallelism_object_shape():
    """bench.py --gpus N > 1 adds a `parallelism` object built from the library's profile classes: per replicated tensor the
    collective time and the exposed (join-wait) time per step, what RCCL reports as its world size, rows/s."""
    b = _bench()

    class Eng(object):
        def comm_info(self):
            return 8, 0
    prof = [{"name": "collective_We (reduce-scatter + all-gather)", "ms": 9.6, "launches": 96, "flops": 0, "bytes": 1},
            {"name": "collective_Wd (reduce-scatter + all-gather)", "ms": 4.8, "launches": 96, "flops": 0, "bytes": 1},
            {"name": "collective_V (reduce-scatter + all-gather)", "ms": 0.96, "launches": 96, "flops": 0, "bytes": 1},
            {"name": "join_wait_We (main lane)", "ms": 0.48, "launches": 48, "flops": 0, "bytes": 0},
            {"name": "adam_dense_D", "ms": 1.92, "launches": 96, "flops": 0, "bytes": 1}]
    p = b.parallelism_object(Eng(), prof, 8, 96, 1.0e6)
    assert p["rccl_world_size"] == 8 and p["launcher_world_size"] == 8 and p["rows_per_s"] == 1.0e6
    assert set(p["per_tensor"]) == {"We", "Wd", "V"}
    assert p["per_tensor"]["We"] == {"collective_us_per_step": 100.0, "exposed_us_per_step": 5.0}
    assert p["per_tensor"]["Wd"]["exposed_us_per_step"] == 0.0
    assert p["collective_us_per_step"] == 160.0 and p["exposed_collective_us_per_step"] == 5.0 and p["adam_slice_us_per_step"] == 20.0
    json.dumps(p)


def test_roofline_is_the_dominant_device_function_over_all_its_classes():
    """SURVEY 8(d) on the driver line: `roofline.frac` = sum of algorithmic FLOPs / sum of launch durations over EVERY launch class of the
    dominant device function (gemm_bf16k_mfma<false, true, 3, false>: encode of both steps + the discriminator step's decode), the best class
    beside it as `frac_best_class`; traffic launch-weighted from the PMC file; every `kernels` row says which step it belongs to."""
    b = _bench()
    enc, dec = "gemm_encode[2B,N]x[N,e]", "gemm_decode[2B,e]x[e,N]"
    prof_d = [{"name": enc, "launches": 48, "ms": 48 * 0.018, "flops": 48 * 1.8828e9, "bytes": 1.0},
              {"name": dec, "launches": 48, "ms": 48 * 0.0204, "flops": 48 * 1.8848e9, "bytes": 1.0},
              {"name": "reduce_encode", "launches": 48, "ms": 0.24, "flops": 0, "bytes": 1.0},
              {"name": "gemm_gWd + gemm_gWe, fused Adam (one launch)", "launches": 48, "ms": 48 * 0.045, "flops": 48 * 3.77e9, "bytes": 48 * 186.2e6}]
    prof_g = [{"name": enc, "launches": 48, "ms": 48 * 0.0178, "flops": 48 * 1.8828e9, "bytes": 1.0},
              {"name": dec, "launches": 48, "ms": 48 * 0.0148, "flops": 48 * 0.9424e9, "bytes": 1.0},      # (gemm_bf16w_mfma: not the dominant function)
              {"name": "gemm_dF[B,e]x[N,e]^T", "launches": 48, "ms": 48 * 0.0146, "flops": 48 * 0.9414e9, "bytes": 1.0}]
    traffic = {enc + " (D-step)": {"hbm_bytes_per_launch": 27.0e6, "algorithmic_bytes": 19.5e6},
               enc + " (G-step)": {"hbm_bytes_per_launch": 27.0e6, "algorithmic_bytes": 19.5e6},
               dec + " (D-step)": {"hbm_bytes_per_launch": 33.0e6, "algorithmic_bytes": 23.3e6},
               "gemm_gWd + gemm_gWe, fused Adam (D-step)": {"hbm_bytes_per_launch": 210.7e6, "algorithmic_bytes": 186.2e6}}
    r, rf, kernels = b.roofline_objects(prof_d, prof_g, traffic, "rXX_traffic.json")
    fl = 2 * 1.8828e9 + 1.8848e9
    us = 18.0 + 17.8 + 20.4
    assert r["kernel"].startswith("gemm_bf16k_mfma<false, true, 3, false>") and r["bound"] == "mfma" and r["launches"] == 144
    assert abs(r["achieved"] - fl / us / 1e6) < 0.05 and abs(r["frac"] - fl / us / 1e6 / b.PEAK_F32_MFMA_TFLOPS) < 1e-3
    assert [(c["step"], c["name"]) for c in r["classes"]] == [("D", enc), ("D", dec), ("G", enc)]
    assert r["best_class"] == "G:" + enc and r["frac_best_class"] > r["frac"] > min(c["frac"] for c in r["classes"])
    assert r["frac_time_weighted"] < r["frac"]            # the generator step's M = 128 products pull the family down
    assert r["traffic"] == round((27.0e6 * 2 + 33.0e6) / 3) and r["traffic_source"] == "rXX_traffic.json"
    assert abs(r["avg_launch_us"] - us / 3) < 0.01
    assert rf["bound"] == "hbm" and abs(rf["achieved"] - 186.2e6 / 45e-6 / 1e9) < 1.0 and rf["traffic"] == 210.7e6
    assert {k["step"] for k in kernels} == {"D", "G"} and len(kernels) == 7
    json.dumps([r, rf, kernels])
    # another plan (none of the dominant function's classes present): the object falls back to the whole 16-wave family instead of vanishing
    r2, _, _ = b.roofline_objects([], prof_g[1:], None, None)
    assert r2["traffic"] is None and r2["launches"] == 96


def test_cpu_baseline_reports_all_threads_and_one_thread(monkeypatch):
    """`cpu_baseline` carries SURVEY 8(d)'s two CPU figures: `value` on the BLAS pool's threads (`cores`) and `value_1thread`."""
    import numpy as np
    import scipy.sparse as sps
    b = _bench()
    w = dict(U=64, N=96, k=8, e=16, B=16, hp=dict(d_lr=1e-4, g_lr=1e-4, d_reg=1e-4, g_reg=0.0, m=10.0, recon_coefficient=0.01))
    rng = np.random.RandomState(0)
    urm = sps.csr_matrix((rng.rand(w["U"], w["N"]) < 0.1).astype(np.float32))
    from ganmf_amd.synthetic import glorot_params
    out = b.cpu_baseline(urm, glorot_params(w["U"], w["N"], w["k"], w["e"], seed=1), w, 0.6)
    assert out["kind"] == "port" and out["unit"] == "steps/s" and out["value"] > 0 and out["cores"] >= 1
    assert out["value_1thread"] is not None and out["value_1thread"] > 0 and "one thread" in out["sample_1thread"]
    json.dumps(out)


class _ReplicaEng(object):
    """stand-in engine: five replicated tensors, what RCCL would report as its world size"""

    def __init__(self, seed, comm_world):
        import numpy as np
        self.t = {tid: np.random.RandomState(seed + tid).rand(7, 5).astype(np.float32) for tid in (0, 1, 2, 3, 101)}
        self.comm_world = comm_world

    def get_tensor(self, tid):
        return self.t[tid]

    def comm_info(self):
        return self.comm_world, 0


def test_replica_check_fields_single_rank():
    """GANMF_BENCH_FORCE_COMM=1 on one GPU: one rank, trivially equal, RCCL world 1 = launcher world 1; a communicator that reports
    another size than the launcher started fails the check."""
    import zlib
    b = _bench()
    crc = lambda a: zlib.crc32(a.tobytes())
    c = b.replica_check(_ReplicaEng(1, 1), 1, None, crc=crc)
    assert c["replicas_bitwise_equal"] is True and c["rccl_world_size_equals_launcher_world_size"] is True and c["ranks_checked"] == 1
    assert set(c["crc32c"]) == {"We", "be", "Wd", "bd", "V"} and all(isinstance(v, str) and len(v) == 8 for v in c["crc32c"].values())
    assert b.replicas_ok(c)
    bad = b.replica_check(_ReplicaEng(1, 2), 1, None, crc=crc)
    assert bad["rccl_world_size_equals_launcher_world_size"] is False and not b.replicas_ok(bad)
    json.dumps(c)


@pytest.mark.parametrize("diverge", [False, True])
def test_replica_check_over_gloo_world_2(tmp_path, capfd, diverge):
    """Two rank processes started by bench.launch_ranks, gloo control plane, stand-in engines: equal replicas -> true on every rank;
    one element of one tensor different on rank 1 -> false, with the per-rank CRCs of that tensor in the object, on EVERY rank (each
    must reach the verdict that makes it exit non-zero)."""
    script = tmp_path / "rank.py"
    script.write_text(textwrap.dedent("""
        import importlib.util, json, os, sys, zlib
        import numpy as np
        import torch.distributed as dist
        rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        spec = importlib.util.spec_from_file_location("bench_under_test", %r)
        b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
        class Eng(object):
            def __init__(self):
                self.t = {tid: np.random.RandomState(tid).rand(33, 9).astype(np.float32) for tid in (0, 1, 2, 3, 101)}
                if %r and rank == 1:
                    self.t[2][5, 3] = np.nextafter(self.t[2][5, 3], np.float32(2.0))      # one ulp in one element of Wd
            def get_tensor(self, tid): return self.t[tid]
            def comm_info(self): return world, rank
        c = b.replica_check(Eng(), world, dist, crc=lambda a: zlib.crc32(a.tobytes()))
        open(os.path.join(%r, "check%%d.json" %% rank), "w").write(json.dumps(c))
        dist.barrier(); dist.destroy_process_group()
        if rank == 0:
            print(json.dumps({"parallelism": c}))
        sys.exit(0 if b.replicas_ok(c) else 3)
    """ % (os.path.join(ROOT, "bench.py"), diverge, str(tmp_path))))
    b = _bench()
    if diverge:
        with pytest.raises(SystemExit) as ex:
            b.launch_ranks(2, [], script=str(script))
        assert ex.value.code == 1
    else:
        b.launch_ranks(2, [], script=str(script))
        line = json.loads(capfd.readouterr().out.strip().splitlines()[-1])
        assert line["parallelism"]["replicas_bitwise_equal"] is True
    checks = [json.load(open(tmp_path / ("check%d.json" % r))) for r in range(2)]
    assert checks[0] == checks[1] and checks[0]["ranks_checked"] == 2
    assert checks[0]["replicas_bitwise_equal"] is (not diverge)
    assert checks[0]["rccl_world_size_equals_launcher_world_size"] is True
    if diverge:
        assert isinstance(checks[0]["crc32c"]["Wd"], list) and len(set(checks[0]["crc32c"]["Wd"])) == 2
        assert all(isinstance(checks[0]["crc32c"][n], str) for n in ("We", "be", "bd", "V"))
