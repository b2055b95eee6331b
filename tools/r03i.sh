for v in 1 3; do
  echo "== GANMF_ADAM_NFAST=$v"
  GANMF_ADAM_NFAST=$v python tools/config_profiles.py c4_e32 c5 c3 2>/dev/null | grep "steps/s\|gemm_gV\|dis_gW\|gUb\] + gemm_gV\|reduce_dE"
  GANMF_ADAM_NFAST=$v python tools/ab.py "" --reps 2 --class "gUb" 2>&1 | tail -2
done
