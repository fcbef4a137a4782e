"""column sums of a [rows][cols] matrix (bias gradients): python tools/colsum_bench.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kalle_audio_amd import ops
dev = torch.device("cuda")
for rows, cols, dt in ((32256, 1536, torch.bfloat16), (32256, 1536, torch.float32), (32256, 12288, torch.bfloat16), (2016, 1536, torch.bfloat16),
                       (32256, 1540, torch.bfloat16)):
    x = torch.randn(rows, cols, device=dev).to(dt)
    out = torch.zeros(cols, device=dev)
    for _ in range(3):
        ops.colsum(x, out=out, accumulate=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.colsum(x, out=out, accumulate=True)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"colsum {rows} x {cols} {str(dt).replace('torch.', '')}: {us:.1f} us, {x.numel() * x.element_size() / us / 1e6:.2f} TB/s")
