set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3q
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for B in 16; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof$B -- python3 $R/bench.py --no-cpu-baseline --batch $B --steps 10 --warmup 3 > $O/b$B.json 2> $O/b$B.err || exit 1
  cp $(find $O/prof$B -name "*kernel_stats.csv" | head -n 1) $O/kernel_stats_b$B.csv
  rm -rf $O/prof$B
done
cd $R
KALLE_BENCH_SHAPES=1 timeout -k 10 300 python bench.py --no-cpu-baseline --batch 16 --steps 10 --warmup 3 > $O/shapes16.json 2> $O/shapes16.err
echo done
