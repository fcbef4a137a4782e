"""profiles/r01_pmc_mfma_util.md from `rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -- python3 bench.py
--steps 2 --warmup 1 --no-cpu-baseline`.   python tools/pmc_mfma_summary.py DIR"""
import collections, csv, glob, os, sys
d = sys.argv[1]
TAG = sys.argv[2] if len(sys.argv) > 2 else "r01"
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt, dur = collections.Counter(), collections.defaultdict(float)
files = sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)
if not files:
    sys.exit(f"no counter_collection.csv under {d}")
if len(files) > 1:      # gpurun MERGES gpurun_out/ back: older passes pile up locally - the newest one is this run's
    print(f"note: {len(files)} counter files under {d}; using the newest ({files[-1]})", file=sys.stderr)
for r in csv.DictReader(open(files[-1])):
    name = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
    agg[name][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        cnt[name] += 1
        dur[name] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
tot = sum(dur.values())
L = [f"# MFMA-pipe utilisation on the DiT path (rocprofv3 PMC, {TAG})\n",
     "`rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -- python3 tools/bench_one_step.py 256 1` (2 steps of the bench workload)",
     "(its own pass, no other counters or traces).  SQ_VALU_MFMA_BUSY_CYCLES counts matrix-pipe busy cycles summed over the 1024",
     "SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs, so clock = GUI_ACTIVE / 8 / kernel time and",
     "MFMA busy = BUSY_CYCLES / (GUI_ACTIVE / 8 x 1024).  The chip holds ~2.1 GHz under this load (2.4 GHz is what the 2.5 PFLOP/s",
     "peak assumes), so 'busy' x 2.5 PFLOP/s x clock / 2.4 is the rate the kernel reaches.  (GUI_ACTIVE includes the dispatch's",
     "ramp, so the clock column over-reads for kernels of a few tens of microseconds.)\n",
     "| kernel | launches | total ms | % of kernel time | held clock GHz | MFMA pipe busy |", "|---|---|---|---|---|---|"]
rows = sorted(((dur[k], k) for k in agg if ("gemm" in k or "attn" in k) and agg[k]["GRBM_GUI_ACTIVE"] > 0), reverse=True)
for du, k in rows:
    act = agg[k]["GRBM_GUI_ACTIVE"] / 8.0
    busy = agg[k]["SQ_VALU_MFMA_BUSY_CYCLES"] / (act * 1024.0)
    L.append(f"| `{k[:64]}` | {cnt[k]} | {du/1e6:.1f} | {100*du/tot:.1f} | {act/du:.2f} | {100*busy:.1f} % |")
gb = sum(agg[k]["SQ_VALU_MFMA_BUSY_CYCLES"] for _, k in rows if "gemm" in k)
ga = sum(agg[k]["GRBM_GUI_ACTIVE"] / 8.0 for _, k in rows if "gemm" in k)
L.append(f"\nAll GEMM kernels together: MFMA pipe {100*gb/(ga*1024):.1f} % busy; all kernels of the step: "
         f"{100*sum(v['SQ_VALU_MFMA_BUSY_CYCLES'] for v in agg.values())/(sum(v['GRBM_GUI_ACTIVE'] for v in agg.values())/8*1024):.1f} %.")
open(f"profiles/{TAG}_pmc_mfma_util.md", "w").write("\n".join(L) + "\n")
print("\n".join(L))
