"""profiles/r01_pmc_hbm_traffic_b256.json + profiles/traffic_r01.json (what bench.py puts into roofline.traffic) from two
rocprofv3 passes over `tools/bench_one_step.py 256 1`: --kernel-trace --pmc FETCH_SIZE and --kernel-trace --pmc WRITE_SIZE
(separate passes).  python tools/pmc_traffic_summary.py FETCH_DIR WRITE_DIR"""
import collections, csv, glob, json, os, re, subprocess, sys


def load(d, counter):
    agg = collections.defaultdict(lambda: [0, 0.0])
    files = sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)
    if not files:
        sys.exit(f"no counter_collection.csv under {d}")
    if len(files) > 1:      # gpurun MERGES gpurun_out/ back: older passes pile up locally - the newest one is this run's
        print(f"note: {len(files)} counter files under {d}; using the newest ({files[-1]})", file=sys.stderr)
    for r in csv.DictReader(open(files[-1])):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
        agg[name][0] += 1
        agg[name][1] += float(r["Counter_Value"])
    return agg


f, w = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
TAG = sys.argv[3] if len(sys.argv) > 3 else "r01"
full, short = {}, {"_note": "HBM-side bytes per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on "
                            "tools/bench_one_step.py 256 1 (2 steps of the default bench workload, B=256/GPU); bytes = "
                            "2*FETCH_SIZE*1024 (gfx950 half-count correction for 16 B/lane streams, MI355X_MICROARCH.md HBM "
                            "section) + WRITE_SIZE*1024, averaged over all launches of the kernel instantiation (the fused-SwiGLU "
                            "GEMMs are their own instantiations)"}
for k, (n, fs) in sorted(f.items(), key=lambda kv: -kv[1][1]):
    ws = w.get(k, [0, 0.0])[1]
    full[k] = {"launches": n, "fetch_raw_MB_per_launch": fs * 1024 / n / 1e6, "fetch_corrected_MB_per_launch": 2 * fs * 1024 / n / 1e6,
               "write_MB_per_launch": ws * 1024 / n / 1e6}
    if k.startswith("gemm3_wgrad_group_kernel"):
        short["gemm3_wgrad_group_kernel"] = int((2 * fs + ws) * 1024 / n)
    m = re.match(r"(gemm\d?_?\w*kernel)<(true|false), (true|false), (true|false)(?:, (\d))?", k)
    if m and "gemm" in k:
        glu = m.group(5) if (m.group(5) and k.startswith("gemm3")) else None
        key = "%s<%d,%d,%d%s>" % (m.group(1), m.group(2) == "true", m.group(3) == "true", m.group(4) == "true",
                                  ",glu%s" % glu if glu and glu != "0" else "")
        short[key] = int((2 * fs + ws) * 1024 / n)
# which code the counters were collected from: the kernel-source hash written on the box next to the passes, and the commit
# checked out when the summary was made (with a flag if the tree's kernel sources no longer hash to the run's stamp)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kalle_audio_amd.build import source_stamp  # noqa: E402
stamp_file = os.path.join(os.path.dirname(os.path.normpath(sys.argv[1])), "source_stamp.txt")
run_stamp = open(stamp_file).read().strip() if os.path.exists(stamp_file) else None
try:
    commit = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True).stdout.strip()
except OSError:
    commit = None
short["_source"] = {"csrc_sha16": run_stamp, "commit_at_publish": commit, "tree_matches_run": run_stamp == source_stamp(),
                    "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE -- python3 tools/bench_one_step.py 256 1"}
full["_source"] = short["_source"]
json.dump(full, open(f"profiles/{TAG}_pmc_hbm_traffic_b256.json", "w"), indent=1)
json.dump(short, open(f"profiles/traffic_{TAG}.json", "w"), indent=1)
print(json.dumps(short, indent=1))
