"""weight-gradient GEMM shapes of the bench step: time vs forced split-K count (KALLE_GEMM_SPLITS) and tile (KALLE_GEMM_TILE);
run once per setting: python tools/wgrad_sweep.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kalle_audio_amd import ops, _lib
dev = torch.device("cuda")
T = 32256
shapes = [("qkv", 4608, 1536, T), ("out", 1536, 1536, T), ("to_kv", 1536, 768, 33280), ("ff1", 12288, 1536, T), ("ff2", 1536, 6144, T)]
mk = lambda r, c: (torch.randn(r, c, device=dev) * 0.5).bfloat16()
res = []
for name, M, N, K in shapes:
    dy, x = mk(K, M), mk(K, N)
    out = torch.empty(M, N, device=dev)
    fn = lambda: ops.gemm(dy, x, a_kmajor=True, b_kmajor=True, out=out)
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    plan = _lib.load().kalle_gemm_last_plan()
    res.append(f"{name} {ms*1e3:.0f}us {2.0*M*N*K/ms/1e9:.0f}TF plan={plan & 255}/s{plan >> 8}")
print(os.environ.get("KALLE_GEMM_SPLITS", "-"), os.environ.get("KALLE_GEMM_TILE", "-"), " | ".join(res))
