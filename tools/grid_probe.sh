cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r3s
for G in 256 240 224 208 256 240; do
  echo "== grid $G"
  KALLE_GEMM_GRID=$G timeout -k 10 200 python tools/gemm_shapes.py nt 32256 4608 1536 32256 1536 1536 32256 12288 1536 32256 1536 6144 2>&1 | tail -4
  KALLE_GEMM_GRID=$G timeout -k 10 200 python tools/gemm_shapes.py nn 32256 1536 1536 32256 1536 12288 32256 6144 1536 2>&1 | tail -3
  KALLE_GEMM_GRID=$G timeout -k 10 300 python bench.py --no-cpu-baseline --steps 10 --warmup 3 2>&1 | grep -o '"ms_per_step": [0-9.]*'
done 2>&1 | tee gpurun_out/r3s/grid_probe.log
