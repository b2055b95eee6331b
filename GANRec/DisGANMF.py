from ganmf_amd.DisGANMF import DisGANMF as _DisGANMF


class DisGANMF(_DisGANMF):
    """`GANRec.DisGANMF.DisGANMF` — the MI355X implementation under the reference's import path."""
