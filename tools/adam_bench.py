"""time kalle_adam_step over a flat buffer of n parameters (default: the DiT's 1.05 B)"""
import sys, torch
sys.path.insert(0, ".")
from kalle_audio_amd import ops

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_050_000_000
d = "cuda"
p = torch.randn(n, device=d); g = torch.randn(n, device=d); m = torch.zeros(n, device=d); v = torch.zeros(n, device=d)
pb = torch.empty(n, device=d, dtype=torch.bfloat16)
for _ in range(3):
    ops.adam_step(p, g, m, v, pb, lr=1e-4, step=1)
e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
e0.record()
for it in range(10):
    ops.adam_step(p, g, m, v, pb, lr=1e-4, step=it + 2)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print(f"n={n} {ms:.3f} ms  {30 * n / ms / 1e9:.2f} TB/s")
