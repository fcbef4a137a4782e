"""the seven weight gradients of one bench-width TransformerBlock through kalle_gemm_wgrad_group at a given token count:
python tools/wgrad_group_bench.py TOKENS [overwrite]   (KALLE_WGRAD_GROUP_PLAN="whole,slices" forces a plan, KALLE_GEMM_DEBUG=1 prints it)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kalle_audio_amd import ops
dev = torch.device("cuda")
T = int(sys.argv[1]) if len(sys.argv) > 1 else 32256
over = len(sys.argv) > 2 and sys.argv[2] == "1"
Tc = T // 126 * 130
D, DC = 1536, 768
shapes = [(T, 3 * D, D), (T, D, D), (T, D, D), (Tc, 2 * DC, DC), (T, D, D), (T, 8 * D, D), (T, D, 4 * D)]
mk = lambda r, c: (torch.randn(r, c, device=dev) * 0.5).bfloat16()
probs = [(mk(t, n), mk(t, k), torch.zeros(n, k, device=dev)) for t, n, k in shapes]
flops = sum(2.0 * t * n * k for t, n, k in shapes)
for _ in range(3):
    ops.gemm_wgrad_group(probs, overwrite=over)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    ops.gemm_wgrad_group(probs, overwrite=over)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print(f"tokens {T} overwrite {int(over)} plan {os.environ.get('KALLE_WGRAD_GROUP_PLAN', 'auto')}: {ms*1e3:.0f} us {flops/ms/1e9:.0f} TFLOP/s", flush=True)
