R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02llasa; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p -- python3 $R/tools/llasa_bench.py 16 1024 4 > $O/log 2> $O/err || exit 1
cp $(find $O/p -name "*kernel_stats.csv" | head -n 1) $O/kernel_stats.csv; rm -rf $O/p
grep -v amdgpu $O/log; python3 $R/tools/kstats.py $O/kernel_stats.csv 22
