#!/bin/bash
# same-box A/B of the software-pipelined 256 x 256 main loop against the round-1 loop (variant build `old`):
#   python -m kalle_audio_amd.build --variant old -DKALLE_GEMM_PIPE=0 ; gpurun -- bash tools/pipe_ab.sh
cd ${GRAFT_REPO_ROOT:-.}
OUT=gpurun_out/r3p; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py tests/test_round2_gpu.py tests/test_round3_gpu.py -q -m gpu -x -k "gemm or wgrad or headline" > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -3 $OUT/tests.log
SH="32256 4608 1536 32256 1536 1536 32256 12288 1536 32256 1536 6144"
for rep in 1 2; do
for V in new old; do
  if [ $V = old ]; then export KALLE_LIB_PATH=$PWD/kalle_audio_amd/libkalle_hip_old.so; else unset KALLE_LIB_PATH; fi
  echo "== $V (rep $rep)"
  timeout -k 10 200 python tools/gemm_shapes.py nt $SH 2>&1 | tail -4
  timeout -k 10 200 python tools/gemm_shapes.py nn 32256 1536 1536 32256 1536 12288 32256 6144 1536 2>&1 | tail -3
  timeout -k 10 200 python tools/gemm_shapes.py tn 1536 1536 32256 12288 1536 32256 4608 1536 32256 2>&1 | tail -3
done; done 2>&1 | tee $OUT/shapes.log
for V in new old new old; do
  if [ $V = old ]; then export KALLE_LIB_PATH=$PWD/kalle_audio_amd/libkalle_hip_old.so; else unset KALLE_LIB_PATH; fi
  echo "== bench $V"; timeout -k 10 300 python bench.py --no-cpu-baseline --steps 10 --warmup 3 2>&1 | grep -o '"ms_per_step": [0-9.]*'
done 2>&1 | tee $OUT/bench.log
