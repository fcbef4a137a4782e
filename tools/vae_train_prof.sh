set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/vaetrain; rm -rf $O; mkdir -p $O
cd $R; timeout -k 10 300 python tools/vae_train_bench.py 2 2.0 2>&1 | grep -v amdgpu | tee $O/bench.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/tools/vae_train_bench.py 2 2.0 > $O/prof.log 2>&1 || exit 1
cp $(find $O/prof -name "*kernel_stats.csv" | head -n 1) $O/kernel_stats.csv; rm -rf $O/prof
