#!/bin/bash
# A/B of the folded-table forward against the general gate-by-gate forward (QIDDM_NO_FOLD=1), run on the GPU box
for v in fold nofold; do
  if [ $v = nofold ]; then export QIDDM_NO_FOLD=1; else unset QIDDM_NO_FOLD; fi
  echo "== $v"
  timeout -k 10 400 python tools/microbench.py 2>&1 | grep -E "B= 65536 f32|B=  4096 f32|n=10.*B=  1024 f32|n=10.*B=   256 f32"
done
