"""ganmf_amd — MI355X-native GANMF / DisGANMF training hot path.

Python host mirror of the reference's recommender interface (GANRec/GANMF.py,
GANRec/DisGANMF.py) over the C ABI of libganmf_hip.so (include/ganmf_hip.h).  There is no CPU
fallback: without the HIP library and a GPU the classes raise.
"""
from ._lib import build_library, library_path, load_library  # noqa: F401

__all__ = ["build_library", "library_path", "load_library", "GANMF", "DisGANMF"]


def __getattr__(name):
    if name == "GANMF":
        from .GANMF import GANMF
        return GANMF
    if name == "DisGANMF":
        from .DisGANMF import DisGANMF
        return DisGANMF
    raise AttributeError(name)
