#!/bin/bash
# same-box A/B: attention outputs in 16-byte stores (product build) against 8-byte stores (variant -DKALLE_ATTN_ST16=0),
# and the LayerNorm-backward workgroup count at B = 16.  python -m kalle_audio_amd.build --variant st8 -DKALLE_ATTN_ST16=0 first.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/attn_ab
rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests -q -m gpu -x -k "attention or attn or block or headline" > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/tests.log
tail -n 3 $O/tests.log
grep -q "tests rc=0" $O/tests.log || exit 1
for i in 1 2; do
  echo "== st16 (product)"; timeout -k 10 120 python tools/attn_bench.py || exit 1
  echo "== st8 (variant)"; KALLE_LIB_PATH=$R/kalle_audio_amd/libkalle_hip_st8.so timeout -k 10 120 python tools/attn_bench.py || exit 1
done 2>&1 | tee $O/attn_bench.log
for i in 1 2; do
  for rpb in 4 8 16; do
    echo "== B=16 KALLE_LN_BWD_RPB=$rpb"
    KALLE_LN_BWD_RPB=$rpb timeout -k 10 200 python bench.py --no-cpu-baseline --batch 16 --steps 20 --warmup 5 2>/dev/null | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])" || exit 1
  done
done 2>&1 | tee $O/ln_rpb.log
for i in 1 2; do
  echo "== B=256 st16"; timeout -k 10 300 python bench.py --no-cpu-baseline 2>/dev/null | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])" || exit 1
  echo "== B=256 st8"; KALLE_LIB_PATH=$R/kalle_audio_amd/libkalle_hip_st8.so timeout -k 10 300 python bench.py --no-cpu-baseline 2>/dev/null | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])" || exit 1
done 2>&1 | tee $O/b256.log
echo done
