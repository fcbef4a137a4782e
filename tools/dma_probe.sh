#!/bin/bash
# what holds the 256 x 256 main loop back: chip clock under load (zero operands), row-stride effects (padded leading dimensions),
# and the LDS-DMA stream alone (variant ko4: -DKALLE_GEMM_PIPE=0 -DKALLE_GEMM_KNOCKOUT=4) under the same paddings
cd ${GRAFT_REPO_ROOT:-.}
OUT=gpurun_out/r3p; mkdir -p $OUT
export KALLE_LIB_PATH=$PWD/kalle_audio_amd/libkalle_hip_old.so
run() { timeout -k 10 200 python tools/gemm_shapes.py nt 32256 4608 1536 32256 1536 1536 32256 1536 6144 2>&1 | tail -3
        timeout -k 10 200 python tools/gemm_shapes.py nn 32256 1536 1536 32256 1536 12288 2>&1 | tail -2
        timeout -k 10 200 python tools/gemm_shapes.py tn 4608 1536 32256 12288 1536 32256 2>&1 | tail -2; }
{
echo "== old loop, random"; run
echo "== old loop, zeros"; KALLE_SHAPE_ZEROS=1 run
for P in 32 64 128 256; do echo "== old loop, random, pad $P"; KALLE_SHAPE_PAD=$P run; done
export KALLE_LIB_PATH=$PWD/kalle_audio_amd/libkalle_hip_ko4.so
for P in 0 32 64 128 256; do echo "== DMA stream alone (ko4), pad $P"; KALLE_SHAPE_PAD=$P run; done
} 2>&1 | tee $OUT/dma_probe.log
