# kernel trace (launch order, grid sizes) of two train steps of the bench workload: bash tools/trace_one_step.sh [B]
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; B=${1:-256}
O=$R/gpurun_out/trace1; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/t -- python3 $R/tools/bench_one_step.py $B 1 > $O/log.txt 2>&1 || exit 1
cp $(find $O/t -name "*kernel_trace.csv" | head -n 1) $O/kernel_trace.csv
cp $(find $O/t -name "*memory_copy_trace.csv" | head -n 1) $O/memory_copy_trace.csv 2>/dev/null
rm -rf $O/t
echo trace done
