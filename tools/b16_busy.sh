# is the small-batch step bound by the GPU or by the host's launch rate?  kernel-time sum (rocprofv3) against the step time
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; B=${1:-16}
O=$R/gpurun_out/busy; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p -- python3 $R/bench.py --no-cpu-baseline --no-kernel-timer --batch $B --steps 20 --warmup 5 > $O/b.json 2> $O/b.err || exit 1
cp $(find $O/p -name "*kernel_stats.csv" | head -n 1) $O/kernel_stats.csv; rm -rf $O/p
cd $R
python - <<'PY'
import csv, json
rows = list(csv.DictReader(open("gpurun_out/busy/kernel_stats.csv")))
d = json.loads(open("gpurun_out/busy/b.json").read().strip().splitlines()[-1])
tot = sum(float(r["TotalDurationNs"]) for r in rows) / 1e6
adam = sum(float(r["TotalDurationNs"]) for r in rows if "adam_kernel" in r["Name"]) / 1e6
print(f"B={d['config']['per_gpu_batch']}: {d['ms_per_step']:.2f} ms per step under rocprofv3; kernel-time sum over 25 steps + init {tot:.1f} ms -> <= {tot / 25:.2f} ms per step")
PY
