"""CPU checks of the drop-in boundary: the shared library loads without a GPU, exports every
symbol include/ganmf_hip.h declares, and fails loudly (no CPU fallback) when asked to compute."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "ganmf_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ganmf_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported_and_bound():
    from ganmf_amd import _lib as L
    lib = L.load_library()
    declared = _declared_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), "libganmf_hip.so does not export " + name
    assert sorted(L.SYMBOLS) == declared, (sorted(set(declared) - set(L.SYMBOLS)), sorted(set(L.SYMBOLS) - set(declared)))
    assert lib.ganmf_abi_version() == L.ABI_VERSION


def test_cfg_struct_matches_header_layout():
    """ganmf_cfg field order / sizes as declared in the header (x86-64 natural alignment)."""
    from ganmf_amd import _lib as L
    names = [f[0] for f in L.Cfg._fields_]
    assert names == ["abi_version", "model", "num_users", "num_items", "num_factors", "emb_dim", "d_layers", "d_act",
                     "batch_size", "d_lr", "g_lr", "d_reg", "g_reg", "m", "recon_coefficient", "device", "world_size",
                     "rank", "row_offset", "flags"]
    assert ctypes.sizeof(L.Cfg) == 96
    assert ctypes.sizeof(L.ProfEntry) == 48 + 8 + 3 * 8


def test_no_cpu_fallback_without_gpu():
    from ganmf_amd import _lib as L
    lib = L.load_library()
    if lib.ganmf_device_count() > 0:
        pytest.skip("GPU present")
    from ganmf_amd.engine import Engine
    with pytest.raises(L.GanmfError):
        Engine(10, 10, 2, 2, 4)
    import scipy.sparse as sps
    from ganmf_amd.GANMF import GANMF
    m = GANMF(sps.identity(8, format="csr", dtype=np.float32), is_experiment=True)
    with pytest.raises(L.GanmfError):
        m.fit(epochs=1)
    with pytest.raises(RuntimeError):
        m._compute_item_score(np.arange(2))


def test_constructor_contract_and_import_path():
    """GANMF.py:26-36: ValueError on unknown modes; item mode trains on URM^T; the class is
    importable as GANRec.GANMF.GANMF (RecSysExp.py:202-204 checks the module's first component)."""
    import scipy.sparse as sps
    from GANRec.GANMF import GANMF
    assert GANMF.__module__.split(".")[0] == "GANRec" and GANMF.RECOMMENDER_NAME == "GANMF"
    urm = sps.random(12, 7, density=0.3, format="csr", dtype=np.float32, random_state=1)
    with pytest.raises(ValueError):
        GANMF(urm, mode="both", is_experiment=True)
    u = GANMF(urm, mode="user", is_experiment=True)
    i = GANMF(urm, mode="item", is_experiment=True)
    assert (u.num_users, u.num_items) == (12, 7) and (i.num_users, i.num_items) == (7, 12)
    assert i.URM_train.shape == (7, 12) and i.get_URM_train().shape == (7, 12)
