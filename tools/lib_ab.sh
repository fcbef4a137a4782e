#!/bin/bash
# same-box A/B of variant builds on the default bench: bash tools/lib_ab.sh <pattern-of-[shape]-lines> <variant> [<variant> ...]
# ("product" = libkalle_hip.so; other names = libkalle_hip_<name>.so from python -m kalle_audio_amd.build --variant <name> -D...)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/lib_ab; mkdir -p $O
PAT=$1; shift
cd $R
for i in 1 2; do
  for v in "$@"; do
    L=$R/kalle_audio_amd/libkalle_hip.so; [ "$v" != product ] && L=$R/kalle_audio_amd/libkalle_hip_$v.so
    echo "== $v"
    KALLE_LIB_PATH=$L KALLE_BENCH_SHAPES=1 timeout -k 10 300 python bench.py --no-cpu-baseline ${BENCH_ARGS} 2> $O/err.txt | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms_per_step', round(d['ms_per_step'], 2))" || exit 1
    grep -E "$PAT" $O/err.txt | head -n 6
  done
done 2>&1 | tee $O/log.txt
