cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r3r
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py tests/test_round2_gpu.py tests/test_round3_gpu.py -q -m gpu -x -k "gemm or wgrad or headline or block" > gpurun_out/r3r/tests.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/r3r/tests.log
for D in 0 1; do KALLE_GEMM_DYNAMIC=$D timeout -k 10 300 python tools/cu_hold_probe.py; done 2>&1 | grep -v Warning | tee gpurun_out/r3r/cu_hold_probe.log
for D in 0 1 0 1; do echo "== bench dynamic=$D"; KALLE_GEMM_DYNAMIC=$D timeout -k 10 300 python bench.py --no-cpu-baseline --steps 10 --warmup 3 2>&1 | grep -o '"ms_per_step": [0-9.]*'; done 2>&1 | tee gpurun_out/r3r/bench.log
