#!/bin/bash
# every small-tile configuration (KALLE_SKINNY="wm,wn,slices") against the planner's own choice: bash tools/skinny_brute.sh M
M=${1:-252}
echo "== auto"; python tools/skinny_gemm_bench.py $M 2>&1 | grep -E "^(qkv|out|q |kv|ff1|ff2)"
for cfg in 1,1,1 2,1,1 2,2,1 1,1,2 2,1,2 2,2,2 1,1,3 2,1,3 2,2,3 1,1,4 2,1,4 2,2,4 1,1,6 2,1,6 2,2,6 1,1,8; do
  echo "== $cfg"; KALLE_SKINNY=$cfg python tools/skinny_gemm_bench.py $M 2>&1 | grep -E "^(qkv|out|q |kv|ff1|ff2)"
done
