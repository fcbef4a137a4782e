cd ${GRAFT_REPO_ROOT:-.}
timeout -k 10 600 python -m pytest tests/test_round2_gpu.py tests/test_kernels_gpu.py -q -m gpu -k "gemm" 2>&1 | tail -5
for MM in 512 2048; do echo "== KALLE_SKINNY_MAX_M=$MM"; KALLE_SKINNY_MAX_M=$MM KALLE_BENCH_SHAPES=1 timeout -k 10 300 python bench.py --no-cpu-baseline --batch 16 --steps 10 --warmup 3 2>&1 | grep -E "shape\]|ms_per_step" | head -14 | cut -c1-160; done
