"""Where does a workgroup of the attention kernels spend its life?  python tools/attn_stamps.py
(kalle_attn_debug_stamps: s_memrealtime stamps of thread 0 of every (batch, head) workgroup; B = 256, 24 heads, 126 tokens)"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kalle_audio_amd import ops, _lib
dev = torch.device("cuda")
lib = _lib.load()
B, H, N, S, Hkv = 256, 24, 126, 130, 12
D = H * 64
g = torch.Generator(device=dev).manual_seed(0)
qkv = (torch.randn(B, N, 3 * D, device=dev, generator=g) * 0.5).bfloat16()
dout = torch.randn(B, N, D, device=dev, generator=g).bfloat16()
half = 16
f = torch.arange(N, device=dev, dtype=torch.float32)[:, None] * (1.0 / (10000 ** (torch.arange(half, device=dev) / half)))[None, :]
rope = (f.cos().contiguous(), f.sin().contiguous())
q = (torch.randn(B, N, D, device=dev, generator=g) * 0.5).bfloat16()
kv = (torch.randn(B, S, 2 * Hkv * 64, device=dev, generator=g) * 0.5).bfloat16()


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def stamped(fn, labels):
    st = torch.zeros(B * H * 8, device=dev, dtype=torch.int64)
    lib.kalle_attn_debug_stamps(ctypes.c_void_p(st.data_ptr()))
    fn()
    torch.cuda.synchronize()
    lib.kalle_attn_debug_stamps(None)
    s = st.view(B * H, 8).cpu().double() * 0.01
    s = s[s[:, 0] > 0]
    t0 = s[:, 0].min()
    span = (s[:, len(labels)].max() - t0).item()
    d = [s[:, i + 1] - s[:, i] for i in range(len(labels))]
    life = s[:, len(labels)] - s[:, 0]
    print("   " + "  ".join(f"{l} {x.median():.2f} (p90 {x.quantile(0.9):.2f})" for l, x in zip(labels, d))
          + f"  | workgroup life {life.median():.2f} us, launch span {span:.1f} us, {len(s)} workgroups")


def self_fwd():
    return ops.attention_fwd(qkv, qkv, qkv, ldq=3 * D, q_off=0, ldk=3 * D, k_off=D, ldv=3 * D, v_off=2 * D, B=B, H=H, Hkv=H,
                             Nq=N, Nk=N, rope=rope)


out, lse = self_fwd()
dqkv = torch.empty_like(qkv)


def self_bwd():
    ops.attention_bwd(qkv, qkv, qkv, out, dout, lse, dqkv, dqkv, dqkv, ldq=3 * D, q_off=0, ldk=3 * D, k_off=D, ldv=3 * D,
                      v_off=2 * D, B=B, H=H, Hkv=H, Nq=N, Nk=N, rope=rope)


def cross_fwd():
    return ops.attention_fwd(q, kv, kv, ldq=D, q_off=0, ldk=2 * Hkv * 64, k_off=0, ldv=2 * Hkv * 64, v_off=Hkv * 64, B=B, H=H,
                             Hkv=Hkv, Nq=N, Nk=S)


print(f"== self-attention forward: {timed(self_fwd):.1f} us")
stamped(self_fwd, ["load+stage", "compute", "store issue", "store ack"])
print(f"== cross-attention forward (130 keys, GQA 2): {timed(cross_fwd):.1f} us")
stamped(cross_fwd, ["load+stage", "compute", "store issue", "store ack"])
print(f"== self-attention fused backward: {timed(self_bwd):.1f} us")
stamped(self_bwd, ["load+stage", "row stats", "dK dV compute", "dK dV store", "dQ compute", "dQ store", "store ack"])
