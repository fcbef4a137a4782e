"""the forward GEMMs of one DiT block at generation batch sizes (M = 252 rows for B = 1 with CFG), each timed alone with HIP events
over graph-free back-to-back launches: python tools/skinny_gemm_bench.py [M]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kalle_audio_amd import ops, _lib
dev = torch.device("cuda")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 252
Mc = M // 126 * 130
shapes = [("qkv", M, 4608, 1536, {}), ("out+res", M, 1536, 1536, {"res": True}), ("q", M, 1536, 1536, {}), ("kv", Mc, 1536, 768, {}),
          ("ff1+glu", M, 12288, 1536, {"glu": True}), ("ff2+res", M, 1536, 6144, {"res": True})]
if len(sys.argv) > 2 and sys.argv[2]:
    shapes = [s for s in shapes if s[0] == sys.argv[2]]
mk = lambda r, c: (torch.randn(r, c, device=dev) * 0.5).bfloat16()
tot = 0.0
for name, m, n, k, opt in shapes:
    x, w = mk(m, k), mk(n, k)
    kw = {}
    if opt.get("res"):
        kw = dict(out_dtype=torch.float32, residual=torch.randn(m, n, device=dev))
    if opt.get("glu"):       # as dit_ops.ff_fwd: GLU projection with the SwiGLU in the epilogue
        hf, act = torch.empty(m, n, device=dev, dtype=torch.bfloat16), torch.empty(m, n // 2, device=dev, dtype=torch.bfloat16)
        kw = dict(bias=torch.randn(n, device=dev), out=hf, glu_mode=1, glu_inner=n // 2, glu_aux=act)
    fn = lambda: ops.gemm(x, w, **kw)
    assert fn() is not None, "shape not supported on this path"
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    # 20 launches inside one HIP graph: device time per launch without the host's per-call cost (~15 us of Python + ctypes)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        fn()
        side.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=side):
            for _ in range(20):
                fn()
    gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        gr.replay()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 100 * 1e3
    plan = _lib.load().kalle_gemm_last_plan()
    if len(sys.argv) > 3:        # check against fp32 torch
        ref = x.float() @ w.float().t()
        got = fn()
        if opt.get("res"):
            ref = ref + kw["residual"]
        if not opt.get("glu"):
            print("   rel err", ((got.float() - ref).norm() / ref.norm()).item())
    tot += us * (2 if name in ("out+res",) else 1)
    print(f"{name:8s} M={m} N={n} K={k}: {us:6.1f} us  {2.0*m*n*k/us/1e6:6.0f} TFLOP/s  weights {n*k*2/us/1e6:5.2f} TB/s  plan {plan & 255}/s{plan >> 8}")
print(f"block total (out counted twice): {tot:.0f} us -> x24 = {tot*24/1e3:.2f} ms")
