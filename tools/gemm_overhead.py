"""fixed cost per output tile of the 256 x 256 kernel: time of (32256 x 1536 x K) for growing K, bf16 output and fp32 + residual
output; intercept = launch + prologue + epilogue rounds, slope = K-tile time.  python tools/gemm_overhead.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kalle_audio_amd import ops, _lib
dev = torch.device("cuda")
M, N = 32256, 1536
mk = lambda r, c: (torch.randn(r, c, device=dev) * 0.5).bfloat16()
res = torch.randn(M, N, device=dev)
for mode in ("bf16 NT", "bf16 NN(b k-major)", "f32+res NT"):
    for K in (64, 256, 768, 1536, 3072, 6144):
        x = mk(M, K)
        w = mk(K, N) if "NN" in mode else mk(N, K)
        kw = dict(b_kmajor="NN" in mode)
        if "f32" in mode:
            kw.update(out_dtype=torch.float32, residual=res)
        fn = lambda: ops.gemm(x, w, **kw)
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 10 * 1e3
        plan = _lib.load().kalle_gemm_last_plan()
        print(f"{mode:20s} K={K:5d}: {us:7.1f} us  {2.0*M*N*K/us/1e6:6.0f} TFLOP/s  plan {plan & 255}", flush=True)
