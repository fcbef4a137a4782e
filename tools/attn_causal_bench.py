"""attention kernels at the Llasa train shape (L = 1024, 32 heads, batch 16), one feature switched on at a time:
python tools/attn_causal_bench.py [B] [L]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kalle_audio_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
L = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
H = 32
dev = torch.device("cuda")
mk = lambda *s: (torch.randn(*s, device=dev) * 0.7).bfloat16()


def tables(rot):
    inv = 1.0 / (500000.0 ** (torch.arange(0, rot, 2, device=dev).float() / rot))
    f = torch.arange(L, device=dev).float()[:, None] * inv[None]
    return f.cos().contiguous(), f.sin().contiguous()


def run(name, Hkv, causal, rot, mask):
    ld = (H + 2 * Hkv) * 64
    qkv, dout = mk(B, L, ld), mk(B, L, H * 64)
    rope = tables(rot) if rot else None
    m = torch.ones(B, L, dtype=torch.bool, device=dev) if mask else None
    kw = dict(ldq=ld, q_off=0, ldk=ld, k_off=H * 64, ldv=ld, v_off=(H + Hkv) * 64, B=B, H=H, Hkv=Hkv, Nq=L, Nk=L, rope=rope,
              key_mask=m, causal=causal)
    o, lse = ops.attention_fwd(qkv, qkv, qkv, **kw)
    dqkv = torch.empty_like(qkv)
    fl = 4.0 * B * H * L * L * 64 * (0.5 if causal else 1.0)
    for what, mult, fn in (("fwd", 1.0, lambda: ops.attention_fwd(qkv, qkv, qkv, **kw)),
                           ("bwd", 2.5, lambda: ops.attention_bwd(qkv, qkv, qkv, o, dout, lse, dqkv, dqkv, dqkv, **kw))):
        fn(); torch.cuda.synchronize()
        ts = []
        for _ in range(6):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); ts.append((e0, e1))
        torch.cuda.synchronize()
        ms = sorted(a.elapsed_time(b) for a, b in ts)[len(ts) // 2]
        print(f"{name:34s} {what} {ms*1e3:8.1f} us  {fl*mult/ms/1e9:6.0f} TFLOP/s (algorithmic)")


run("plain MHA", 32, False, 0, False)
run("+ causal", 32, True, 0, False)
run("+ rotary 32", 32, True, 32, False)
run("+ rotary 64", 32, True, 64, False)
run("+ GQA 8 kv heads", 8, True, 64, False)
run("+ key mask (Llasa layer)", 8, True, 64, True)
run("GQA only", 8, False, 0, False)
