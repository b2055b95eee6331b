from ganmf_amd.GANMF import GANMF as _GANMF


class GANMF(_GANMF):
    """`GANRec.GANMF.GANMF` — the MI355X implementation under the reference's import path."""
