#!/bin/bash
# GPU box: A/B the in-tree library against another build of it (QIDDM_HIP_LIB) on the recorded adjoint training step.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for b in 102 205 256 307 512; do
  echo "default: $(QIDDM_TRAIN_BATCH=$b python3 $ROOT/tools/profile_train.py adjoint | tail -1)"
  for lib in "$@"; do
    echo "$lib: $(QIDDM_HIP_LIB=$ROOT/$lib QIDDM_TRAIN_BATCH=$b python3 $ROOT/tools/profile_train.py adjoint | tail -1)"
  done
done
