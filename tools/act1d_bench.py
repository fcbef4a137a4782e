"""anti-aliased activation kernel (mel-VAE AMP blocks): python tools/act1d_bench.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kalle_audio_amd import conv_ops
dev = torch.device("cuda")
filt = conv_ops.kaiser_sinc_filter12(dev)
for B, C, L, dt in ((2, 512, 220160, torch.float32), (2, 128, 880640, torch.float32), (2, 512, 220160, torch.bfloat16)):
    x = torch.randn(B, C, L, device=dev).to(dt)
    a = torch.zeros(C, device=dev)
    conv_ops.act1d(x, filt, a, a, True); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); conv_ops.act1d(x, filt, a, a, True); e1.record(); ts.append((e0, e1))
    torch.cuda.synchronize()
    ms = sorted(p.elapsed_time(q) for p, q in ts)[2]
    byts = 2 * x.numel() * x.element_size()
    print(f"act1d {B}x{C}x{L} {str(dt)[6:]}: {ms*1e3:.0f} us, {byts/ms/1e6:.0f} GB/s algorithmic (read + write once), "
          f"{x.numel()*(2*6*2+2*12+20)/ms/1e9:.1f} TFLOP/s-ish")
