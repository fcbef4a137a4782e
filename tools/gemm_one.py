"""run a few GEMM shapes N times each (for rocprofv3 --pmc): python tools/gemm_one.py [M]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kalle_audio_amd import ops
dev = torch.device("cuda")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 32256
N, K = 4608, 1536
mk = lambda r, c: (torch.randn(r, c, device=dev) * 0.5).bfloat16()
x, w, dy = mk(M, K), mk(N, K), mk(M, N)
for _ in range(3):
    ops.gemm(x, w)                                                           # NT
    ops.gemm(dy, w, b_kmajor=True)                                           # NN
    ops.gemm(dy, x, a_kmajor=True, b_kmajor=True, out_dtype=torch.float32)   # TN
torch.cuda.synchronize()
