R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02sk; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for S in q qkv ff1+glu ff2+res kv; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_$S -- python3 $R/tools/skinny_gemm_bench.py 252 $S > $O/$S.log 2>&1 || exit 1
  cp $(find $O/p_$S -name "*kernel_stats.csv" | head -n 1) $O/stats_$S.csv; rm -rf $O/p_$S
  echo "== $S"; grep "TFLOP" $O/$S.log; head -4 $O/stats_$S.csv | cut -c1-150
done
