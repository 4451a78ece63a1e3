for env in "PP_FUSED_ATTENTION=0 PP_ATTN_PERTURB=3e-7" "PP_FUSED_ATTENTION=0 PP_ATTN_PERTURB=1e-6" "PP_FUSED_ATTENTION=0 PP_ATTN_PERTURB=1e-5"; do
  echo "== $env"
  env $env timeout -k 10 300 python -m pytest tests/test_train_gpu.py -m gpu -q -s -k "full_backward_matches and f16x3" 2>&1 | grep "full backward \[" | cut -c1-420
done
