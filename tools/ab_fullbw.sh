#!/bin/bash
# usage (GPU box): bash tools/ab_fullbw.sh — the full-backward fixture test's per-tensor report with each training-path switch off in turn
# (fused attention under autograd, K slices of the weight gradients, the single ViT pass): which one moves which gradient tensors.
for env in "" "PP_KSPLIT=0" "PP_FUSED_ATTENTION=0" "PP_BATCH_VIT_TRAIN=0" "PP_KSPLIT=0 PP_FUSED_ATTENTION=0 PP_BATCH_VIT_TRAIN=0"; do
  echo "== $env"
  env $env timeout -k 10 300 python -m pytest tests/test_train_gpu.py -m gpu -q -s -k "full_backward_matches and f16x3" 2>&1 | grep "full backward \[" | cut -c1-420
done
