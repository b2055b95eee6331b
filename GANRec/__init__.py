"""Drop-in package name: RecSysExp.py decides "isGAN" from `cls.__module__.split('.')[0] == 'GANRec'`
(RecSysExp.py:202-204) and imports `GANRec.GANMF.GANMF` / `GANRec.DisGANMF.DisGANMF`."""
