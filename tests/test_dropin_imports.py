"""The drop-in claim of INTEGRATION.md option A: this repository's root placed AHEAD of the reference tree on
sys.path.  RecSysExp.py:41-45 imports GANRec.GANMF, GANRec.DisGANMF, GANRec.CFGAN and GANRec.CAAE in a row and
later tells GAN models apart by `cls.__module__.split('.')[0] == 'GANRec'` (RecSysExp.py:202-204).  The test builds a
stand-in reference tree (stub GANRec/CAAE.py, GANRec/CFGAN.py and a GANRec/GANMF.py that must NOT win) behind the
repository and performs that import sequence in a fresh interpreter."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_driver_import_sequence_with_reference_tree_behind(tmp_path):
    ref = tmp_path / "reference_tree" / "GANRec"
    ref.mkdir(parents=True)
    (ref / "__init__.py").write_text("")
    (ref / "CAAE.py").write_text("class CAAE(object):\n    RECOMMENDER_NAME = 'CAAE'\n")
    (ref / "CFGAN.py").write_text("class CFGAN(object):\n    RECOMMENDER_NAME = 'CFGAN'\n")
    (ref / "GANMF.py").write_text("raise ImportError('the reference GANMF (TensorFlow) must be shadowed')\n")
    script = textwrap.dedent("""
        import sys
        sys.path[:0] = [%r, %r]
        from GANRec.GANMF import GANMF
        from GANRec.DisGANMF import DisGANMF
        from GANRec.CFGAN import CFGAN
        from GANRec.CAAE import CAAE
        import ganmf_amd.GANMF, ganmf_amd.DisGANMF
        assert issubclass(GANMF, ganmf_amd.GANMF.GANMF) and issubclass(DisGANMF, ganmf_amd.DisGANMF.DisGANMF)
        for cls in (GANMF, DisGANMF, CFGAN, CAAE):
            assert cls.__module__.split('.')[0] == 'GANRec', cls.__module__      # the driver's isGAN test
        assert CAAE.RECOMMENDER_NAME == 'CAAE' and CFGAN.RECOMMENDER_NAME == 'CFGAN'
        assert GANMF.RECOMMENDER_NAME == 'GANMF' and DisGANMF.RECOMMENDER_NAME == 'DisGANMF'
        import inspect
        fit = inspect.signature(GANMF.fit).parameters
        assert list(fit)[:6] == ['self', 'num_factors', 'emb_dim', 'epochs', 'batch_size', 'd_lr']
        print('OK')
    """) % (ROOT, str(tmp_path / "reference_tree"))
    res = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0 and res.stdout.strip().endswith("OK"), res.stdout + res.stderr
