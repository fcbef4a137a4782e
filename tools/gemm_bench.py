"""Per-shape timing of the DiT's GEMMs (fwd NT / dgrad NN / wgrad TN) on one GPU: HIP events, interleaved rounds.
usage: python tools/gemm_bench.py [B] [rounds]"""
import sys
import os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kalle_audio_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
R = int(sys.argv[2]) if len(sys.argv) > 2 else 10
D, Dc, N, S = 1536, 768, 126, 130
M, Mc = B * N, B * S
dev = torch.device("cuda")
lin = [("qkv", M, 3 * D, D), ("out", M, D, D), ("to_kv", Mc, 2 * Dc, Dc), ("ff1", M, 8 * D, D), ("ff2", M, D, 4 * D)]


def mk(r, c):
    return (torch.randn(r, c, device=dev) * 0.5).bfloat16()


cases = []
for name, m, n, k in lin:
    x, w, dy = mk(m, k), mk(n, k), mk(m, n)
    res = torch.randn(m, n, device=dev)
    cases.append((f"{name}.fwd  NT bf16 [{m}x{n}x{k}]", 2.0 * m * n * k, lambda x=x, w=w: ops.gemm(x, w)))
    if name in ("out", "ff2"):
        cases.append((f"{name}.fwd  NT f32+res [{m}x{n}x{k}]", 2.0 * m * n * k,
                      lambda x=x, w=w, res=res: ops.gemm(x, w, out_dtype=torch.float32, residual=res)))
    cases.append((f"{name}.dgrad NN [{m}x{k}x{n}]", 2.0 * m * n * k, lambda dy=dy, w=w: ops.gemm(dy, w, b_kmajor=True)))
    cases.append((f"{name}.wgrad TN [{n}x{k}x{m}]", 2.0 * m * n * k,
                  lambda dy=dy, x=x: ops.gemm(dy, x, a_kmajor=True, b_kmajor=True, out_dtype=torch.float32)))

for _, _, f in cases:
    f()
torch.cuda.synchronize()
times = [[] for _ in cases]
for r in range(R):
    for i, (_, _, f) in enumerate(cases):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        f()
        e1.record()
        times[i].append((e0, e1))
torch.cuda.synchronize()
tot_f = tot_t = 0.0
for (name, fl, _), ts in zip(cases, times):
    ms = sorted(a.elapsed_time(b) for a, b in ts)
    med = ms[len(ms) // 2]
    print(f"{name:44s} {med*1e3:9.1f} us  {fl/med/1e9:7.0f} TF  (min {ms[0]*1e3:.1f})")
    w = 3 if name.startswith("out") else 1   # out-proj shape occurs 3x per block (self out, to_q, cross out)
    tot_f += fl * w
    tot_t += med * w
print(f"weighted block total: {tot_t:.3f} ms -> {tot_f/tot_t/1e9:.0f} TF avg")
