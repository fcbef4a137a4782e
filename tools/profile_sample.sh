# rocprofv3 kernel stats of the sampling path (B = 1, CFG, 20 DDIM steps); run on the GPU box
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02smp
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/tools/sample_bench.py 1 20 > $O/under_rocprof.log 2> $O/prof.err || exit 1
cp $(find $O/prof -name "*kernel_stats.csv" | head -n 1) $O/kernel_stats.csv
rm -rf $O/prof
cd $R
timeout -k 10 300 python tools/sample_bench.py 1 50 > $O/plain.log 2>&1
cat $O/plain.log | grep -v amdgpu
