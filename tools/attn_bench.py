"""time the fused attention kernels at the bench shapes: python tools/attn_bench.py [B]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kalle_audio_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
H, Hkv, N, D, Dc = 24, 12, 126, 1536, 768
S = int(sys.argv[2]) if len(sys.argv) > 2 else 130
dev = torch.device("cuda")
mk = lambda *s: (torch.randn(*s, device=dev) * 0.7).bfloat16()
qkv, q, kv, dout = mk(B, N, 3 * D), mk(B, N, D), mk(B, S, 2 * Dc), mk(B, N, D)
inv = 1.0 / (10000 ** (torch.arange(0, 32, 2, device=dev).float() / 32))
f = torch.arange(N, device=dev).float()[:, None] * inv[None]
rope = (f.cos().contiguous(), f.sin().contiguous())
sa = dict(ldq=3 * D, q_off=0, ldk=3 * D, k_off=D, ldv=3 * D, v_off=2 * D, B=B, H=H, Hkv=H, Nq=N, Nk=N)
ca = dict(ldq=D, q_off=0, ldk=2 * Dc, k_off=0, ldv=2 * Dc, v_off=Dc, B=B, H=H, Hkv=Hkv, Nq=N, Nk=S)
o1, l1 = ops.attention_fwd(qkv, qkv, qkv, rope=rope, **sa)
o2, l2 = ops.attention_fwd(q, kv, kv, **ca)
dqkv, dq, dkv = torch.empty_like(qkv), torch.empty_like(q), torch.empty_like(kv)
cases = [("self fwd", 4.0 * B * H * N * N * 64, lambda: ops.attention_fwd(qkv, qkv, qkv, rope=rope, **sa)),
         ("cross fwd", 4.0 * B * H * N * S * 64, lambda: ops.attention_fwd(q, kv, kv, **ca)),
         ("self bwd", 10.0 * B * H * N * N * 64, lambda: ops.attention_bwd(qkv, qkv, qkv, o1, dout, l1, dqkv, dqkv, dqkv, rope=rope, **sa)),
         ("cross bwd", 10.0 * B * H * N * S * 64, lambda: ops.attention_bwd(q, kv, kv, o2, dout, l2, dq, dkv, dkv, **ca))]
for _, _, fn in cases:
    fn()
torch.cuda.synchronize()
for name, fl, fn in cases:
    ts = []
    for _ in range(8):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        ts.append((e0, e1))
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in ts)[len(ts) // 2]
    print(f"{name:10s} {ms*1e3:8.1f} us  {fl/ms/1e9:6.0f} TF")
