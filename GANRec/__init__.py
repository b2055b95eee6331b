"""Drop-in package name for the two recommenders this repository implements.

RecSysExp.py decides "isGAN" from `cls.__module__.split('.')[0] == 'GANRec'` (RecSysExp.py:202-204) and imports
`GANRec.GANMF.GANMF` / `GANRec.DisGANMF.DisGANMF`, but the same driver also imports `GANRec.CAAE` and `GANRec.CFGAN`
(RecSysExp.py:41-45), which live only in the reference tree.  With this repository AHEAD of the reference on
`sys.path` (option A) this package must therefore not hide the reference's: `extend_path` appends every
other `GANRec/` directory found on `sys.path` to the package search path, so `GANRec.GANMF` / `GANRec.DisGANMF` resolve
here (first entry) and every other submodule resolves in the reference's directory."""
import pkgutil

__path__ = pkgutil.extend_path(__path__, __name__)
