for mode in auto f32 bf16x3; do for tile in 0 64 128; do GANMF_MFMA=$mode GANMF_DEBUG_PLAN=1 python tools/gemm_one.py NT 128 50000 250 $tile 0 20 2>&1 | tr '\n' ' '; echo " [mode=$mode tile=$tile]"; done; done
GANMF_X3KG=1 GANMF_DEBUG_PLAN=1 python tools/gemm_one.py NT 128 50000 250 64 0 20 2>&1 | tr '\n' ' '; echo " [x3kg 64]"
echo gUb; for mode in auto bf16x3; do GANMF_MFMA=$mode GANMF_DEBUG_PLAN=1 python tools/gemm_one.py NN 128 250 50000 0 0 20 2>&1 | tr '\n' ' '; echo " [mode=$mode]"; done
GANMF_X3KG=3 GANMF_DEBUG_PLAN=1 python tools/gemm_one.py NN 128 250 50000 64 0 20 2>&1 | tr '\n' ' '; echo " [x3kg]"
