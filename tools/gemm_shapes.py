"""time arbitrary GEMM shapes: python tools/gemm_shapes.py LAYOUT M N K [M N K ...]   (LAYOUT nt|nn|tn)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kalle_audio_amd import ops
dev = torch.device("cuda")
lay = sys.argv[1]
dims = list(map(int, sys.argv[2:]))
# KALLE_SHAPE_ZEROS=1: all-zero operands (how far the chip's clock under load holds the kernel back);
# KALLE_SHAPE_PAD=P: leading dimensions padded by P elements (address-to-channel effects of power-of-two-ish row strides)
ZEROS = os.environ.get("KALLE_SHAPE_ZEROS") == "1"
PAD = int(os.environ.get("KALLE_SHAPE_PAD", "0"))
def mk(r, c):
    t = torch.zeros(r, c + PAD, device=dev, dtype=torch.bfloat16) if ZEROS else (torch.randn(r, c + PAD, device=dev) * 0.5).bfloat16()
    return t[:, :c] if PAD else t
for i in range(0, len(dims), 3):
    M, N, K = dims[i:i + 3]
    if lay == "nt":
        a, b = mk(M, K), mk(N, K); fn = lambda: ops.gemm(a, b)
    elif lay == "nn":
        a, b = mk(M, K), mk(K, N); fn = lambda: ops.gemm(a, b, b_kmajor=True)
    else:
        a, b = mk(K, M), mk(K, N); fn = lambda: ops.gemm(a, b, a_kmajor=True, b_kmajor=True, out_dtype=torch.float32)
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(10):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); ts.append((e0, e1))
    torch.cuda.synchronize()
    ms = sorted(x.elapsed_time(y) for x, y in ts)[5]
    print(f"{lay} {M}x{N}x{K}: {ms*1e3:.1f} us {2.0*M*N*K/ms/1e9:.0f} TF")
