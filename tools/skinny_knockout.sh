# knock-out timing of the few-row GEMM kernels (rocprofv3 kernel durations). dbg bits: 1 no stores, 2 no pipeline, 4 no MFMAs, 8 no in-loop DMA.
# The `GemmParams.dbg` switches this script drives (KALLE_FEW_ROWS_DBG) were taken out of the shipped kernels again once the numbers in
# DESIGN.md section 5.0 were recorded (they cost a branch per MFMA block); `git log -S'KALLE_FEW_ROWS_DBG' -- kalle_audio_amd/csrc/gemm2.hip`
# finds the commits that carry them.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02ko; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for CFG in 1,1,1 2,2,1; do for S in q ff2+res; do for D in 0 1 4 8 12; do
  KALLE_SKINNY=$CFG KALLE_FEW_ROWS_DBG=$D timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p -- python3 $R/tools/skinny_gemm_bench.py 252 $S > $O/log 2>&1 || exit 1
  echo "== $CFG $S dbg=$D: $(python3 $R/tools/kstats.py $(find $O/p -name '*kernel_stats.csv') 2 | grep gemm2)"
  rm -rf $O/p
done; done; done
