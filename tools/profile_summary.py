"""Write profiles/<tag>_summary.md from a rocprofv3 kernel_stats.csv and the two bench JSON lines.
python tools/profile_summary.py TAG kernel_stats.csv under_rocprof.json default.json 'command'"""
import csv, json, sys
tag, stats, jprof, jdef, cmd = sys.argv[1:6]
rows = list(csv.DictReader(open(stats)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
bp, bd = json.load(open(jprof)), json.load(open(jdef))
L = [f"# rocprofv3 --kernel-trace --stats of the default bench command ({tag})\n",
     f"Command (GPU box, 1 x MI355X): `{cmd}`\n",
     f"Bench line under rocprof: {bp['ms_per_step']:.1f} ms/step, {bp['value']:.0f} audio-s/s; un-profiled run "
     f"({tag}_default.json): {bd['ms_per_step']:.1f} ms/step, {bd['value']:.0f} audio-s/s, "
     f"{bd['algorithmic_tflops_per_gpu']:.0f} algorithmic TFLOP/s.\n",
     f"Total kernel time {tot/1e6:.0f} ms over {bp['steps'] + bp['warmup']} steps ({bp['warmup']} warm-up + {bp['steps']} timed) + model init.\n",
     "| kernel | calls | total ms | avg us | % |", "|---|---|---|---|---|"]
for r in rows[:28]:
    name = r["Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:78]
    L.append(f"| `{name}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.2f} | {float(r['AverageNs'])/1e3:.1f} | {float(r['Percentage']):.2f} |")
rf = bd["roofline"]
L += ["", "HIP-event timing inside bench.py per GEMM instantiation (un-profiled run; one extra step after the timed region with events "
      "around every launch - inside the timed region only the roofline kernel is bracketed, two event records per launch cost ~4.5 us each):\n",
      "| variant | launches | avg us | TFLOP/s | share of step |", "|---|---|---|---|---|"]
for k, v in sorted(rf["all_gemm_variants"].items(), key=lambda kv: -kv[1]["time_share_of_step"]):
    L.append(f"| `{k}` | {v['launches']} | {v['avg_us']:.1f} | {v['tflops']:.0f} | {100*v['time_share_of_step']:.1f} % |")
L += ["", f"roofline kernel `{rf['kernel']}`: HIP events {rf['avg_launch_us']:.1f} us avg over {rf['launches']} launches ({rf.get('hip_events_in', 'the timed region')}) = "
      f"{rf['achieved']:.0f} TFLOP/s = {100*rf['frac']:.1f} % of the 2.5 PFLOP/s dense-bf16 peak; the rocprof row of the same instantiation "
      "averages all profiled steps.", "",
      f"HBM-side traffic (separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes, {tag[:3]}_pmc_hbm_traffic_b256.json; FETCH doubled per the gfx950 "
      f"correction): {rf['traffic']/1e6:.0f} MB per launch vs {rf['algorithmic_bytes_per_launch']/1e6:.0f} MB algorithmic (operands once + output "
      "once). FETCH_SIZE counts L2 misses incl. Infinity-Cache hits, so the ratio is L2 re-fetch of the streamed panels, not DRAM traffic.",
      "", f"cpu_baseline: {json.dumps(bd.get('cpu_baseline'))}"]
open(f"profiles/{tag}_summary.md", "w").write("\n".join(L) + "\n")
print("wrote", f"profiles/{tag}_summary.md")
