cd ${GRAFT_REPO_ROOT:-.}
for C in auto 1,1,1 2,1,1 2,2,1; do echo "== KALLE_SKINNY=$C"; if [ $C = auto ]; then timeout -k 10 100 python tools/skinny_gemm_bench.py 252 "" check 2>&1 | grep -v amdgpu; else KALLE_SKINNY=$C timeout -k 10 100 python tools/skinny_gemm_bench.py 252 "" check 2>&1 | grep -v amdgpu; fi; done
for C in 1,1,2 1,1,4 2,1,4 2,2,4 2,2,8; do echo "== ff2 KALLE_SKINNY=$C"; KALLE_SKINNY=$C timeout -k 10 100 python tools/skinny_gemm_bench.py 252 ff2+res check 2>&1 | grep -v amdgpu; done
