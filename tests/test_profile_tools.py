"""tools/step_classes.py and tools/collect_traffic.py on a synthetic rocprofv3 output: dispatches are labelled by their ORDER
inside a step, so classes that share a kernel instantiation and grid stay apart, and the combined launches of
csrc/gemm_multi.hpp stand in for their parts (CPU only: pure CSV processing)."""
import csv
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

F32 = "void ganmf::gemm_f32_mfma<64, 64, 64, 3, false, true, 4>(ganmf::GemmP)"
D_STEP = ["void ganmf::front_kernel<4>(ganmf::GemmP, ganmf::DensP)", F32, "ganmf::splitk_reduce_kernel(ganmf::RedP)", F32,
          "ganmf::d_coef_kernel(float*)", F32, "ganmf::gemm_bf16s_red(ganmf::GemmP, ganmf::RedP, int)",
          "void ganmf::gemm_bf16s_mfma<64, 64, 32, true, true, 3, false>(ganmf::GemmP)"]
G_STEP = ["void ganmf::front_kernel<4>(ganmf::GemmP, ganmf::DensP)", F32, "ganmf::splitk_reduce_kernel(ganmf::RedP)", F32,
          "ganmf::splitk_reduce_kernel(ganmf::RedP)", F32, "ganmf::splitk_reduce_kernel(ganmf::RedP)", F32,
          "ganmf::splitk_reduce_kernel(ganmf::RedP)", "void ganmf::pair_kernel<4>(ganmf::GemmP, ganmf::GemmP)",
          "ganmf::adam_rows_kernel(float*)"]
NEW_D_STEP = ["void ganmf::front_kernel<4>(ganmf::GemmP, ganmf::DensP)", F32, "ganmf::splitk_reduce_kernel(ganmf::RedP)", F32,
              "void ganmf::de_dcoef_kernel<4>(ganmf::GemmP, ganmf::DCoefP, int)", "ganmf::gemm_bf16s_red(ganmf::GemmP, ganmf::RedP, int)",
              "void ganmf::gemm_bf16s_mfma<64, 64, 32, true, true, 3, false>(ganmf::GemmP)"]
WPAIR_D_STEP = ["void ganmf::front_kernel<4>(ganmf::GemmP, ganmf::DensP)", F32, "ganmf::splitk_reduce_kernel(ganmf::RedP)", F32,
                "void ganmf::de_dcoef_kernel<4>(ganmf::GemmP, ganmf::DCoefP, int)", "ganmf::splitk_reduce_kernel(ganmf::RedP)",
                "ganmf::wgrad_pair_kernel(ganmf::GemmP, ganmf::GemmP)"]
OLD_G_STEP = ["ganmf::densify_rows_kernel(ganmf::DensP)", F32, F32, "ganmf::splitk_reduce_kernel(ganmf::RedP)", F32, F32, F32, F32,
              "ganmf::splitk_reduce_kernel(ganmf::RedP)", F32, "ganmf::adam_rows_kernel(float*)"]


def _write(path, steps, counter=None):
    cols = ["Dispatch_Id", "Kernel_Name", "Start_Timestamp", "End_Timestamp", "Grid_Size_X", "Grid_Size", "Workgroup_Size_X",
            "LDS_Block_Size"] + (["Counter_Name", "Counter_Value"] if counter else [])
    t, d = 1000, 0
    with open(path, "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=cols)
        w.writeheader()
        for names in steps:
            for i, n in enumerate(names):
                d += 1
                row = {"Dispatch_Id": d, "Kernel_Name": n, "Start_Timestamp": t, "End_Timestamp": t + 1000 * (i + 1),
                       "Grid_Size_X": 262144, "Grid_Size": 262144, "Workgroup_Size_X": 1024, "LDS_Block_Size": 98304}
                if counter:
                    row.update({"Counter_Name": counter, "Counter_Value": 100.0 * (i + 1)})
                w.writerow(row)
                t += 1000 * (i + 1)


def test_labels_follow_dispatch_order(tmp_path):
    from step_classes import label, load
    p = tmp_path / "trace.csv"
    _write(p, [D_STEP, G_STEP, D_STEP, OLD_G_STEP, NEW_D_STEP, WPAIR_D_STEP, D_STEP[:4]])      # the last step is truncated and must be dropped
    lab = [k for k, _ in label(load(str(p)))]
    assert lab[:8] == ["D:gen+rows", "D:encode", "D:reduce(encode)", "D:decode", "D:d_coef", "D:dE",
                       "D:gWd+adam+reduce(dE)", "D:gWe+adam"]
    assert lab[8:19] == ["G:gen+rows", "G:encode", "G:reduce(encode)", "G:decode", "G:reduce(decode)", "G:dE", "G:reduce(dE)",
                         "G:dF", "G:reduce(dF)", "G:gUb+gV+adam", "G:adam_rows_U"]
    old = lab[27:38]      # one kernel per piece: three consecutive launches of ONE instantiation are three classes
    assert old == ["G:densify+gather", "G:gen", "G:encode", "G:reduce(encode)", "G:decode", "G:dE", "G:dF", "G:gUb",
                   "G:reduce(gUb)", "G:gV+adam", "G:adam_rows_U"]
    assert lab[38:45] == ["D:gen+rows", "D:encode", "D:reduce(encode)", "D:decode", "D:dE+d_coef", "D:gWd+adam+reduce(dE)", "D:gWe+adam"]
    assert lab[45:52] == ["D:gen+rows", "D:encode", "D:reduce(encode)", "D:decode", "D:dE+d_coef", "D:reduce(dE)", "D:gWd+gWe+adam"]
    assert len(lab) == 8 + 11 + 8 + 11 + 7 + 7
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "step_classes.py"), str(p)], capture_output=True, text=True)
    assert out.returncode == 0 and "| D:encode |" in out.stdout and "D+G pair" in out.stdout


def test_staged_pass_labels(tmp_path):
    """A staged discriminator pass (stage_pass): three launches in front of it, then steps WITHOUT a front launch, delimited by their last
    weight-gradient launch; a ragged last minibatch keeps its own front launch."""
    from step_classes import COUNTS, label, load
    p = tmp_path / "trace.csv"
    head = ["ganmf::densify_rows_kernel(ganmf::DensP)", "void ganmf::gemm_bf16k_mfma<false, false, 3, false>(ganmf::GemmP)"]
    ghead = ["ganmf::adam_rows_advance_kernel(float const*)"]
    lazy_g = G_STEP[:-1]      # a generator step of a lazy pass has no all-rows Adam launch
    _write(p, [head, WPAIR_D_STEP[1:], NEW_D_STEP[1:], WPAIR_D_STEP, G_STEP, ghead, lazy_g, lazy_g, ["ganmf::adam_rows_flush_kernel(float*)"]])
    COUNTS["D"] = COUNTS["G"] = 0
    lab = [k for k, _ in label(load(str(p)))]
    assert lab[:2] == ["P:CSR rows of the whole pass (+ lr_t of its steps)", "P:generated rows of the whole pass (batched)"]
    assert lab[2].startswith("Q:all-rows Adam over U, rows advanced") and lab[3].startswith("Q:all-rows Adam over U, every row")
    lab = lab[:2] + ["-"] + lab[4:]      # (the per-pass launches are listed first)
    assert lab[3:9] == ["D:encode", "D:reduce(encode)", "D:decode", "D:dE+d_coef", "D:reduce(dE)", "D:gWd+gWe+adam"]
    assert lab[9:15] == ["D:encode", "D:reduce(encode)", "D:decode", "D:dE+d_coef", "D:gWd+adam+reduce(dE)", "D:gWe+adam"]
    assert lab[15:22] == ["D:gen+rows", "D:encode", "D:reduce(encode)", "D:decode", "D:dE+d_coef", "D:reduce(dE)", "D:gWd+gWe+adam"]
    assert lab[22] == "G:gen+rows" and lab[32] == "G:adam_rows_U" and lab[33] == "G:gen+rows" and lab[42] == "G:gUb+gV+adam"
    assert len(lab) == 3 + 6 + 6 + 7 + 11 + 10 + 10
    assert COUNTS == {"D": 3, "G": 3}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "step_classes.py"), str(p)], capture_output=True, text=True)
    assert out.returncode == 0 and "in front of the pass" in out.stdout, out.stdout + out.stderr


def test_traffic_per_class(tmp_path):
    for name, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        os.makedirs(tmp_path / name / "run")
        _write(tmp_path / name / "run" / "1_counter_collection.csv", [D_STEP, G_STEP, WPAIR_D_STEP], counter)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "collect_traffic.py"), str(tmp_path / "fetch"),
                          str(tmp_path / "write")], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    d = json.loads(out.stdout)
    enc = d["gemm_encode[2B,N]x[N,e] (D-step)"]
    assert enc["FETCH_SIZE_KiB_raw"] == 200.0 and enc["hbm_bytes_per_launch"] == (2 * 200 + 200) * 1024     # FETCH doubled (gfx950)
    assert enc["algorithmic_bytes"] == 4 * (256 * 3707 + 3707 * 992 + 256 * 992)
    gw = d["gemm_gWd[2B,e]^Tx[2B,N] + reduce_dE (D-step)"]
    assert gw["algorithmic_bytes"] >= 24 * 993 * 3706       # the six Adam streams are counted
    assert d["gemm_gWd + gemm_gWe, fused Adam (D-step)"]["algorithmic_bytes"] >= 24 * (993 * 3706 + 3707 * 992)
    assert "gemm_gUb[B,N]x[N,k] + gemm_gV[B,N]^Tx[B,k] (G-step)" in d
    assert len({v["class"] for v in d.values()}) == len(d)      # no entry shared between classes


def _lint():
    import importlib.util
    spec = importlib.util.spec_from_file_location("check_asm_prefetch", os.path.join(ROOT, "tools", "check_asm_prefetch.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_asm_prefetch_lint_model():
    """tools/check_asm_prefetch.py on hand-written listings: a use of an asm-loaded register before the wait that covers it is
    reported; behind a sufficient vmcnt wait, behind an unconditional branch, or for a compiler-issued load it is not."""
    lint = _lint()
    def run(body):
        return lint.check_kernel("k", list(enumerate(body.strip().split("\n"), 1)))
    bad = """
        ;;#ASMSTART
        global_load_dwordx4 v[10:13], v[2:3], off
        ;;#ASMEND
        ;;#ASMSTART
        global_load_dwordx4 v[14:17], v[4:5], off
        ;;#ASMEND
        s_waitcnt vmcnt(1)
        v_mov_b32_e32 v20, v15
    """
    assert len(run(bad)) == 1 and "v[14:17]" in run(bad)[0]
    good = bad.replace("v_mov_b32_e32 v20, v15", "v_mov_b32_e32 v20, v11")      # the older load is covered by vmcnt(1)
    assert run(good) == []
    assert run(bad.replace("s_waitcnt vmcnt(1)", "s_waitcnt vmcnt(0)")) == []
    assert run(bad.replace("s_waitcnt vmcnt(1)", "s_branch .LBB0_3\n.LBB0_2:")) == []      # another block: nothing known, nothing reported
    compiler_load = """
        global_load_dwordx4 v[10:13], v[2:3], off
        v_mov_b32_e32 v20, v11
    """
    assert run(compiler_load) == []      # hipcc's own loads carry hipcc's own waits; only asm-issued loads are tracked
    store_reads = """
        ;;#ASMSTART
        global_load_dwordx4 v[10:13], v[2:3], off
        ;;#ASMEND
        ds_write_b128 v30, v[10:13]
    """
    assert len(run(store_reads)) == 1


def test_asm_prefetch_lint_on_the_shipped_kernels():
    """The 16-wave split-bf16 kernels (gemm_bf16k.hpp, also inside front_kernel / de_dcoef_kernel) as THIS hipcc compiles them:
    no instruction reads a prefetch register between its inline-asm load and the hand-counted wait that covers it (advisor
    finding of round 3; cross-compiles gfx950 assembly, no GPU)."""
    import subprocess
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_asm_prefetch.py")], capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    assert "checked" in res.stdout and " 0 reads" in res.stdout


def _make(*args, timeout=900):
    import subprocess
    return subprocess.run(["make", "-C", os.path.join(ROOT, "ganmf_amd", "csrc")] + list(args), capture_output=True, text=True, timeout=timeout)


def test_build_runs_the_asm_prefetch_lint_and_fails_on_a_finding(tmp_path):
    """csrc/Makefile: the default target runs tools/check_asm_prefetch.py in front of the compile (a hash comparison on an unchanged tree) and a
    finding fails the build before anything is compiled or overwritten.  The lint command is the Makefile's ASM_LINT variable: a stand-in
    that reports a finding (exit 1) must stop `make`; the real one on the shipped tree must pass and leave its stamp, after which a second
    run is the hash comparison only."""
    import time
    bad = _make("ASM_LINT=echo seeded finding; exit 1", "OUT=" + str(tmp_path / "never.so"), "-B")
    assert bad.returncode != 0 and "seeded finding" in bad.stdout and not (tmp_path / "never.so").exists(), bad.stdout[-500:] + bad.stderr[-500:]
    good = _make("asm_lint")
    assert good.returncode == 0, good.stdout[-2000:] + good.stderr[-2000:]
    stamp = os.path.join(ROOT, "ganmf_amd", "csrc", ".asm_lint.stamp")
    assert os.path.exists(stamp) and len(open(stamp).read().strip()) == 64
    t0 = time.time()
    assert _make("asm_lint").returncode == 0
    assert time.time() - t0 < 10.0, "an unchanged tree must not pay for the lint again"


def test_build_refuses_an_untested_hipcc():
    """csrc/Makefile: another hipcc than the tested one fails the build; ALLOW_UNTESTED_HIPCC=1 builds with the inline-asm kernels off as the
    compiled default (-DGANMF_ASM_PREFETCH_UNTESTED: GANMF_X3KG=0, bf16w=0 in abi_core.inc).  `make -n`: nothing is compiled here."""
    res = _make("-n", "-B", "HIPCC_TESTED=0.0.0")
    assert res.returncode != 0 and "is not the tested 0.0.0" in res.stderr, res.stderr[-500:]
    res = _make("-n", "-B", "HIPCC_TESTED=0.0.0", "ALLOW_UNTESTED_HIPCC=1")
    assert res.returncode == 0 and "-DGANMF_ASM_PREFETCH_UNTESTED" in res.stdout, res.stdout[-500:] + res.stderr[-500:]
    res = _make("-n", "-B")
    assert res.returncode == 0 and "GANMF_ASM_PREFETCH_UNTESTED" not in res.stdout
    src = open(os.path.join(ROOT, "ganmf_amd", "csrc", "lib", "abi_core.inc")).read()
    assert "GANMF_ASM_PREFETCH_UNTESTED" in src and 'env_int("GANMF_X3KG", kAsmPrefetchDefault ? 7 : 0)' in src
