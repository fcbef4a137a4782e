R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02ada; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p -- python3 $R/bench.py --no-cpu-baseline --global-cond-type adaLN --steps 4 --warmup 2 > $O/b.json 2> $O/err || exit 1
cp $(find $O/p -name "*kernel_stats.csv" | head -n 1) $O/kernel_stats.csv; rm -rf $O/p
python3 $R/tools/kstats.py $O/kernel_stats.csv 26
