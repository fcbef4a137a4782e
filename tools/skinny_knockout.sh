# knock-out timing of the few-rows slab kernel (rocprofv3 kernel durations): full / no stores / no pipeline / neither
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02ko; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for S in q ff2+res; do for D in 0 1 2 3; do
  KALLE_FEW_ROWS_DBG=$D timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p -- python3 $R/tools/skinny_gemm_bench.py 252 $S > $O/log 2>&1 || exit 1
  echo "== $S dbg=$D: $(grep -h gemm2_kernel $(find $O/p -name '*kernel_stats.csv') | cut -d, -f2-4)"
  rm -rf $O/p
done; done
cd $R
for T in 128 256; do echo "== unsplit v2, tile $T"; KALLE_GEMM_FEW_ROWS=0 KALLE_V2_MIN_M=128 KALLE_GEMM_TILE=$T timeout -k 10 100 python tools/skinny_gemm_bench.py 252 "" check 2>&1 | grep -v amdgpu; done
