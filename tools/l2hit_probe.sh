# does the L2 hit rate of the operand stream bound the 256 x 256 main loop?  KALLE_GEMM_DBG=4 makes every workgroup fetch the same
# four operand panels (~100 % L2 hits; results wrong, timing valid); variant ko4 = the LDS-DMA stream alone
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r3v
run() { timeout -k 10 200 python tools/gemm_shapes.py nt 32256 4608 1536 32256 1536 1536 32256 1536 6144 2>&1 | tail -3
        timeout -k 10 200 python tools/gemm_shapes.py nn 32256 1536 1536 32256 1536 12288 2>&1 | tail -2; }
{
for rep in 1 2; do
echo "== full kernel"; run
echo "== full kernel, shared panels (DBG=4)"; KALLE_GEMM_DBG=4 run
done
export KALLE_LIB_PATH=$PWD/kalle_audio_amd/libkalle_hip_ko4.so
echo "== DMA alone"; run
echo "== DMA alone, shared panels"; KALLE_GEMM_DBG=4 run
} 2>&1 | tee gpurun_out/r3v/l2hit.log
