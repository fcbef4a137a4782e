"""profiles/r01_pmc_hbm_traffic_b256.json + profiles/traffic_r01.json (what bench.py puts into roofline.traffic) from two
rocprofv3 passes over `tools/bench_one_step.py 256 1`: --kernel-trace --pmc FETCH_SIZE and --kernel-trace --pmc WRITE_SIZE
(separate passes).  python tools/pmc_traffic_summary.py FETCH_DIR WRITE_DIR"""
import collections, csv, glob, json, re, sys


def load(d, counter):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0])):
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
json.dump(full, open(f"profiles/{TAG}_pmc_hbm_traffic_b256.json", "w"), indent=1)
json.dump(short, open(f"profiles/traffic_{TAG}.json", "w"), indent=1)
print(json.dumps(short, indent=1))
