"""ctypes binding of libganmf_hip.so — exactly the symbols include/ganmf_hip.h declares."""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("GANMF_LIB_PATH") or os.path.join(_HERE, "libganmf_hip.so")      # (GANMF_LIB_PATH: A/B runs of two builds)
_lib = None

ABI_VERSION = 1
MODEL_GANMF, MODEL_DISGANMF = 0, 1
FLAG_MFMA_F32, FLAG_MFMA_BF16, FLAG_MFMA_F16 = 1, 2, 4     # include/ganmf_hip.h GANMF_FLAG_*
MFMA_FLAGS = {None: 0, "auto": 0, "f32": FLAG_MFMA_F32, "bf16": FLAG_MFMA_BF16, "f16": FLAG_MFMA_F16}
ACT = {"linear": 0, "tanh": 1, "relu": 2, "sigmoid": 3}
T_USER_EMB, T_ITEM_EMB = 100, 101
SLOT_PARAM, SLOT_ADAM_M, SLOT_ADAM_V, SLOT_BEST = 0, 1, 2, 3
EVAL_METRICS = ("ROC_AUC", "PRECISION", "PRECISION_RECALL_MIN_DEN", "RECALL", "MAP", "MRR", "NDCG", "HIT_RATE", "ARHR")
EVAL_MAX_CUTOFFS = 8
PROF_MAX = 48

# every exported symbol of include/ganmf_hip.h (checked by tests/test_abi.py)
SYMBOLS = [
    "ganmf_create", "ganmf_destroy", "ganmf_comm_unique_id", "ganmf_comm_init", "ganmf_comm_init_local", "ganmf_comm_abort", "ganmf_comm_info", "ganmf_set_urm_csr",
    "ganmf_set_tensor", "ganmf_get_tensor", "ganmf_tensor_shape", "ganmf_get_adam_powers",
    "ganmf_set_adam_powers", "ganmf_train_epoch", "ganmf_train_epoch_ragged", "ganmf_train_step", "ganmf_scores",
    "ganmf_set_seen_csr", "ganmf_set_score_filter", "ganmf_recommend", "ganmf_set_test_csr", "ganmf_evaluate", "ganmf_snapshot_best", "ganmf_restore_best", "ganmf_profile_enable", "ganmf_profile_read", "ganmf_stream_timer",
    "ganmf_bench_scores", "ganmf_gemm_f32", "ganmf_crc32c", "ganmf_device_count", "ganmf_abi_version", "ganmf_last_error",
]


class Cfg(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32), ("model", C.c_int32),
        ("num_users", C.c_int64), ("num_items", C.c_int64),
        ("num_factors", C.c_int32), ("emb_dim", C.c_int32),
        ("d_layers", C.c_int32), ("d_act", C.c_int32), ("batch_size", C.c_int32),
        ("d_lr", C.c_float), ("g_lr", C.c_float), ("d_reg", C.c_float), ("g_reg", C.c_float),
        ("m", C.c_float), ("recon_coefficient", C.c_float),
        ("device", C.c_int32), ("world_size", C.c_int32), ("rank", C.c_int32),
        ("row_offset", C.c_int64), ("flags", C.c_uint32),
    ]


class ProfEntry(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("launches", C.c_int64), ("ms", C.c_double),
                ("flops", C.c_double), ("bytes", C.c_double)]


def library_path():
    return _LIB_PATH


def build_library(verbose=False):
    """hipcc --offload-arch=gfx950 build of the in-tree shared library (no GPU needed)."""
    res = subprocess.run(["make", "-C", os.path.join(_HERE, "csrc")], capture_output=True, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout[-4000:])
        print(res.stderr[-4000:])
    if res.returncode != 0:
        raise RuntimeError("building libganmf_hip.so failed")
    return _LIB_PATH


def load_library():
    """Loads libganmf_hip.so or raises: the HIP path is the only path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise RuntimeError(
            "libganmf_hip.so is missing (%s). Build it with `make -C ganmf_amd/csrc` or "
            "`python -c 'import __graft_entry__ as g; g.build()'`. There is no CPU fallback." % _LIB_PATH)
    lib = C.CDLL(_LIB_PATH)
    P = C.POINTER
    vp, i32, i64, f32p = C.c_void_p, C.c_int32, C.c_int64, P(C.c_float)
    sig = {
        "ganmf_create": (C.c_int, [P(Cfg), P(vp)]),
        "ganmf_destroy": (C.c_int, [vp]),
        "ganmf_comm_unique_id": (C.c_int, [P(C.c_uint8)]),
        "ganmf_comm_init": (C.c_int, [vp, P(C.c_uint8)]),
        "ganmf_comm_init_local": (C.c_int, [vp, i32]),
        "ganmf_comm_abort": (C.c_int, [vp]),
        "ganmf_comm_info": (C.c_int, [vp, P(i32), P(i32)]),
        "ganmf_set_urm_csr": (C.c_int, [vp, P(C.c_int64), P(C.c_int32), f32p, i64, i64]),
        "ganmf_set_tensor": (C.c_int, [vp, C.c_int, C.c_int, f32p, i64]),
        "ganmf_get_tensor": (C.c_int, [vp, C.c_int, C.c_int, f32p, i64]),
        "ganmf_tensor_shape": (C.c_int, [vp, C.c_int, P(C.c_int64), P(C.c_int64)]),
        "ganmf_get_adam_powers": (C.c_int, [vp, f32p]),
        "ganmf_set_adam_powers": (C.c_int, [vp, f32p]),
        "ganmf_train_epoch": (C.c_int, [vp, P(C.c_int32), i64, i32, i32, i64, P(C.c_int32), f32p, f32p]),
        "ganmf_train_epoch_ragged": (C.c_int, [vp, P(C.c_int32), i64, i32, i32, i64, P(C.c_int32), P(C.c_int32), f32p, f32p]),
        "ganmf_train_step": (C.c_int, [vp, C.c_int, P(C.c_int32), i32, f32p]),
        "ganmf_scores": (C.c_int, [vp, P(C.c_int32), i64, C.c_int, f32p]),
        "ganmf_set_seen_csr": (C.c_int, [vp, P(C.c_int64), P(C.c_int32), i64, i64]),
        "ganmf_set_score_filter": (C.c_int, [vp, P(C.c_int32), i64, C.c_int]),
        "ganmf_recommend": (C.c_int, [vp, P(C.c_int32), i64, C.c_int, i32, C.c_int, P(C.c_int32), f32p]),
        "ganmf_set_test_csr": (C.c_int, [vp, P(C.c_int64), P(C.c_int32), P(C.c_double), i64, i64]),
        "ganmf_evaluate": (C.c_int, [vp, P(C.c_int32), i64, C.c_int, C.c_int, P(C.c_int32), i32, P(C.c_double), P(C.c_double),
                                     P(C.c_double)]),
        "ganmf_crc32c": (C.c_uint32, [C.c_uint32, vp, C.c_uint64]),
        "ganmf_snapshot_best": (C.c_int, [vp]),
        "ganmf_restore_best": (C.c_int, [vp]),
        "ganmf_profile_enable": (C.c_int, [vp, C.c_int]),
        "ganmf_profile_read": (C.c_int, [vp, P(ProfEntry), i32, P(i32)]),
        "ganmf_stream_timer": (C.c_int, [vp, C.c_int, P(C.c_double)]),
        "ganmf_bench_scores": (C.c_int, [vp, i64, C.c_int, i32, f32p]),
        "ganmf_gemm_f32": (C.c_int, [C.c_int, f32p, f32p, f32p, i64, i64, i64, C.c_int, C.c_int, C.c_int,
                                     C.c_int, C.c_int, f32p]),
        "ganmf_device_count": (C.c_int, []),
        "ganmf_abi_version": (C.c_int, []),
        "ganmf_last_error": (C.c_char_p, []),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.ganmf_abi_version() != ABI_VERSION:
        raise RuntimeError("libganmf_hip.so ABI %d != binding %d" % (lib.ganmf_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


class GanmfError(RuntimeError):
    pass


def check(rc, what=""):
    if rc != 0:
        msg = load_library().ganmf_last_error()
        text = "%s failed (%d): %s" % (what, rc, msg.decode() if msg else "?")
        if b"out of memory" in (msg or b"").lower():
            raise MemoryError(text)
        raise GanmfError(text)
