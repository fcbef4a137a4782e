"""SURVEY.md 8(d) sweep on one GPU: per-GPU batch {16, 64, 256} x io_channels {64, 512, 1024}, the adaLN variant and the
rectified-flow objective.  Runs bench.py once per configuration (fresh process), prints / writes a markdown table.
python tools/bench_sweep.py [out.md]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
runs = [dict(batch=b, io=1024) for b in (16, 64, 256)] + [dict(batch=256, io=c) for c in (64, 512)] + \
       [dict(batch=256, io=1024, gct="adaLN"), dict(batch=64, io=1024, gct="adaLN"), dict(batch=256, io=1024, obj="rectified_flow")]
rows = []
for r in runs:
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "2", "--batch", str(r["batch"]),
           "--io-channels", str(r["io"]), "--global-cond-type", r.get("gct", "prepend"), "--objective", r.get("obj", "v"),
           "--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    line = [l for l in out.stdout.splitlines() if l.startswith("{")]
    if not line:
        print("FAILED", cmd, out.stderr[-500:]); continue
    d = json.loads(line[-1])
    rows.append((r, d))
    print(f"B={r['batch']:4d} io={r['io']:5d} {r.get('gct','prepend'):8s} {r.get('obj','v'):15s} {d['ms_per_step']:8.1f} ms "
          f"{d['value']:8.0f} audio-s/s {d['algorithmic_tflops_per_gpu']:6.0f} TF", flush=True)
md = ["| per-GPU batch | io_channels | global cond | objective | ms/step | audio-s/s | algorithmic TFLOP/s | dominant GEMM TFLOP/s |",
      "|---|---|---|---|---|---|---|---|"]
for r, d in rows:
    md.append(f"| {r['batch']} | {r['io']} | {r.get('gct','prepend')} | {r.get('obj','v')} | {d['ms_per_step']:.1f} | {d['value']:.0f} | "
              f"{d['algorithmic_tflops_per_gpu']:.0f} | {d['roofline']['achieved']:.0f} ({d['roofline']['kernel']}) |")
text = "\n".join(md) + "\n"
print(text)
if len(sys.argv) > 1:
    open(sys.argv[1], "w").write("# DiT train-step sweep, 1 x MI355X (tools/bench_sweep.py; bench.py --steps 5 --warmup 2)\n\n" + text)
