cd ${GRAFT_REPO_ROOT:-.}
for T in 128 256; do echo "== unsplit v2, tile $T"; KALLE_GEMM_FEW_ROWS=0 KALLE_V2_MIN_M=128 KALLE_GEMM_TILE=$T timeout -k 10 100 python tools/skinny_gemm_bench.py 252 "" check 2>&1 | grep -v amdgpu; done
