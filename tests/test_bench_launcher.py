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
