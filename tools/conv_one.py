"""One conv shape, repeated: for rocprofv3 kernel-trace / PMC runs of the VAE conv kernels.
python tools/conv_one.py KIND Cin Cout K stride dil L B [reps] [res]   (KIND conv|convT)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kalle_audio_amd import conv_ops
kind, Cin, Cout, K, stride, dil, L, B = sys.argv[1], *map(int, sys.argv[2:9])
reps = int(sys.argv[9]) if len(sys.argv) > 9 else 5
res = len(sys.argv) > 10
dev = torch.device("cuda")
torch.manual_seed(0)
x = torch.randn(B, Cin, L, device=dev)
a = torch.zeros(Cin, device=dev)
if kind == "conv":
    w = conv_ops.weight_norm_fold(torch.randn(Cout, Cin, K, device=dev) * 0.05, None)
    r = torch.randn(B, Cout, L, device=dev) if res else None
    fn = lambda: conv_ops.conv1d(x, w, None, Cout=Cout, K=K, stride=stride, padding=dil * (K - 1) // 2, dilation=dil,
                                 act=0 if res else 1, alpha=a, beta=a, residual=r)
else:
    w = conv_ops.weight_norm_fold(torch.randn(Cin, Cout, K, device=dev) * 0.05, None, transposed=True)
    fn = lambda: conv_ops.conv_transpose1d(x, w, None, Cout=Cout, K=K, stride=stride, padding=(K - stride) // 2)
y = fn(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    y = fn()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
fl = 2.0 * Cin * Cout * K * B * (y.shape[2] if kind == "conv" else L)
print(f"{kind} {Cin}->{Cout} k{K} s{stride} d{dil} L{L} B{B}: {ms*1e3:.0f} us, {fl/ms/1e9:.1f} TFLOP/s")
