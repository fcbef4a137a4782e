cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r3z
timeout -k 10 800 python -m pytest tests/test_round3_gpu.py tests/test_round2_gpu.py -q -m gpu -x -k "conv_weight_gradient or vae_backward or enable_grad" > gpurun_out/r3z/tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/r3z/tests.log
echo "== new"; timeout -k 10 300 python tools/vae_train_bench.py 2 2.0 2>&1 | grep -v amdgpu
echo "== old (KALLE_CONV_WGRAD_V1=1)"; KALLE_CONV_WGRAD_V1=1 timeout -k 10 300 python tools/vae_train_bench.py 2 2.0 2>&1 | grep -v amdgpu
echo "== new, B=4 x 5 s"; timeout -k 10 300 python tools/vae_train_bench.py 4 5.0 2>&1 | grep -v amdgpu
