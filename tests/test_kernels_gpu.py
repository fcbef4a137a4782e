"""-m gpu: every HIP kernel called through the C-ABI, checked against plain torch fp32 math of the same op
(torch is only the checker here; the product path never calls these torch ops)."""
import math
import os
import sys

import pytest
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
import kalle_oracle as ko  # noqa: E402

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    a = a.float()
    b = b.float()
    return ((a - b).norm() / (b.norm() + 1e-12)).item()


@pytest.fixture(scope="module")
def ops(dev):
    from kalle_audio_amd import ops as _ops
    return _ops


# ------------------------------------------------------------------------------------------------ GEMM
def _mk(shape, dev, ints=False, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    if ints:
        return torch.randint(-3, 4, shape, generator=g).float().to(dev)
    return torch.randn(shape, generator=g).to(dev)


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (300, 136, 72), (1008, 384, 192), (77, 24, 16), (256, 512, 1536),
                                   (1000, 264, 128), (520, 128, 64), (2048, 1024, 640), (256, 128, 4096),
                                   (768, 384, 2048)])
@pytest.mark.parametrize("akm,bkm", [(0, 0), (0, 1), (1, 1), (1, 0)])
def test_gemm_layouts_exact_int(ops, dev, M, N, K, akm, bkm):
    """small-integer operands: products/sums are exact in bf16 x bf16 -> fp32, so any layout / lane-map error shows
    as an exact mismatch (asymmetric random B: a transposed C write cannot hide)."""
    if akm and M % 8:
        M = (M + 7) // 8 * 8
    A = _mk((M, K), dev, ints=True, seed=1)
    Bm = _mk((N, K), dev, ints=True, seed=2)
    ref = A @ Bm.t()
    a = (A.t().contiguous() if akm else A).bfloat16()
    b = (Bm.t().contiguous() if bkm else Bm).bfloat16()
    out = ops.gemm(a, b, a_kmajor=bool(akm), b_kmajor=bool(bkm), out_dtype=torch.float32)
    torch.cuda.synchronize()
    assert out.shape == (M, N)
    assert torch.equal(out, ref), f"max diff {(out - ref).abs().max().item()}"


@pytest.mark.parametrize("akm,bkm", [(0, 0), (0, 1), (1, 1)])
def test_gemm_random_bf16_out(ops, dev, akm, bkm):
    M, N, K = 520, 264, 328
    A = _mk((M, K), dev, seed=3).bfloat16()
    Bm = _mk((N, K), dev, seed=4).bfloat16()
    ref = A.float() @ Bm.float().t()
    a = A.t().contiguous() if akm else A
    b = Bm.t().contiguous() if bkm else Bm
    out = ops.gemm(a, b, a_kmajor=bool(akm), b_kmajor=bool(bkm))
    assert out.dtype == torch.bfloat16
    assert rel_l2(out, ref) < 4e-3  # bf16 output rounding only


@pytest.mark.parametrize("shape", ["small_v1", "large_v2"])
def test_gemm_epilogue(ops, dev, shape):
    Bt, T, N, K = (3, 40, 136, 72) if shape == "small_v1" else (4, 126, 384, 256)
    M = Bt * T
    A = _mk((M, K), dev, seed=5).bfloat16()
    W = _mk((N, K), dev, seed=6).bfloat16()
    bias = _mk((N,), dev, seed=7)
    gate = _mk((Bt, N), dev, seed=8)
    res = _mk((M, N), dev, seed=9)
    base = A.float() @ W.float().t()
    ref = (base + bias) * torch.sigmoid(1 - gate).repeat_interleave(T, 0) + res
    out = ops.gemm(A, W, out_dtype=torch.float32, bias=bias, gate=gate, rows_per_batch=T, residual=res)
    assert rel_l2(out, ref) < 1e-5
    # accumulate + alpha
    c = res.clone()
    ops.gemm(A, W, out=c, accumulate=True, alpha=0.5)
    assert rel_l2(c, res + 0.5 * base) < 1e-5
    # row remap: write behind one prepended row per batch
    stream = torch.zeros((Bt, T + 1, N), device=dev)
    ops.gemm(A, W, out=stream.view(Bt * (T + 1), N), c_rows_per_batch=T, c_batch_rows=T + 1, c_row_offset=1)
    assert rel_l2(stream[:, 1:], base.view(Bt, T, N)) < 1e-5
    assert stream[:, 0].abs().max().item() == 0.0


def test_gemm_fused_swiglu_matches_unfused(ops, dev):
    """256x256 kernel with the SwiGLU forward / backward (+ GLU bias gradient) in its epilogue vs GEMM + swiglu kernels
    (bit for bit: both sides accumulate K in one pass - the few-rows split-K path, which sums K slices, is switched off here
    and has its own tests in test_round2_gpu.py)"""
    ops.FEW_ROWS = False
    try:
        _fused_swiglu_body(ops, dev)
    finally:
        ops.FEW_ROWS = True


def _fused_swiglu_body(ops, dev):
    M, D, inner = 2304, 256, 512          # (more rows than the small-tile path takes: both sides run the 256-row kernels)
    x = _mk((M, D), dev, seed=80).bfloat16()
    w1 = (_mk((2 * inner, D), dev, seed=81) * 0.1).bfloat16()
    b1 = _mk((2 * inner,), dev, seed=82) * 0.1
    w2 = (_mk((D, inner), dev, seed=83) * 0.1).bfloat16()
    gb = _mk((M, D), dev, seed=84).bfloat16()
    h_ref = ops.gemm(x, w1, bias=b1)
    act_ref = ops.swiglu_fwd(h_ref)
    h = torch.empty_like(h_ref)
    act = torch.empty_like(act_ref)
    assert ops.gemm(x, w1, bias=b1, out=h, glu_mode=1, glu_inner=inner, glu_aux=act) is not None
    assert torch.equal(h, h_ref) and torch.equal(act, act_ref)
    dact = ops.gemm(gb, w2, b_kmajor=True)
    db_ref = torch.zeros(2 * inner, device=dev)
    dh_ref = ops.swiglu_bwd(dact, h_ref, db_ref)
    dh = torch.empty_like(h_ref)
    db = torch.zeros(2 * inner, device=dev)
    assert ops.gemm(gb, w2, b_kmajor=True, out=dh, N=inner, glu_mode=2, glu_inner=inner, glu_aux=h_ref,
                    glu_dbias=db) is not None
    # the activation backward is compiled twice (GEMM epilogue / stand-alone kernel, both -ffast-math): the same formula may round
    # a sigmoid differently by one ulp - a handful of the 2.4 M outputs differ by one bf16 step, everything else bit for bit
    ne = dh != dh_ref
    assert int(ne.sum()) <= 32 and rel_l2(dh.float(), dh_ref.float()) < 1e-4, (int(ne.sum()), rel_l2(dh.float(), dh_ref.float()))
    assert rel_l2(db, db_ref) < 1e-5
    # few rows (<= 512): the small-tile kernels carry the same fused epilogue (64 / 128-column tiles, 32 x + 32 gate columns per wave)
    # (bit for bit against the same kernel family without the fused epilogue; against the 256-row kernels the sum order differs)
    h2, act2 = torch.zeros_like(h_ref[:100]), torch.zeros_like(act_ref[:100])
    assert ops.gemm(x[:100], w1, bias=b1, out=h2, glu_mode=1, glu_inner=inner, glu_aux=act2) is not None
    h2_ref = ops.gemm(x[:100], w1, bias=b1)
    assert torch.equal(h2, h2_ref) and torch.equal(act2, ops.swiglu_fwd(h2_ref))
    assert rel_l2(h2, h_ref[:100]) < 1e-2
    # shapes no fused kernel takes report "unsupported" so the caller un-fuses (here: an fp32 output)
    hf = torch.empty((M, 2 * inner), device=dev)
    assert ops.gemm(x, w1, bias=b1, out=hf, glu_mode=1, glu_inner=inner, glu_aux=act) is None


# ------------------------------------------------------------------------------------------------ norms
@pytest.mark.parametrize("D", [128, 1536, 2048])
@pytest.mark.parametrize("ada", [False, True])
def test_layernorm_fwd_bwd(ops, dev, D, ada):
    Bt, T = 3, 21
    rows = Bt * T
    x = _mk((rows, D), dev, seed=10) * 2 + 0.3
    gamma = 1 + 0.1 * _mk((D,), dev, seed=11)
    scale = 0.2 * _mk((Bt, D), dev, seed=12) if ada else None
    shift = 0.2 * _mk((Bt, D), dev, seed=13) if ada else None
    dy = _mk((rows, D), dev, seed=14).bfloat16()
    dres = _mk((rows, D), dev, seed=15)

    xr = x.clone().requires_grad_(True)
    gr = gamma.clone().requires_grad_(True)
    sr = scale.clone().requires_grad_(True) if ada else None
    hr = shift.clone().requires_grad_(True) if ada else None
    ln = F.layer_norm(xr, (D,), gr, None)
    yr = ln
    if ada:
        yr = ln * (1 + sr.repeat_interleave(T, 0)) + hr.repeat_interleave(T, 0)
    yr.backward(dy.float())

    y, mean, rstd = ops.layernorm_fwd(x, gamma, None, scale, shift, rows_per_batch=T)
    assert rel_l2(y, yr) < 4e-3
    dx, dgamma, _ = ops.layernorm_bwd(dy, x, gamma, mean, rstd, scale=scale, rows_per_batch=T, dres=dres)
    assert rel_l2(dx, xr.grad + dres) < 1e-4
    assert rel_l2(dgamma, gr.grad) < 1e-4
    if ada:
        dsc, dsh = ops.adaln_mod_bwd(dy, x, gamma, None, mean, rstd, Bt, T)
        assert rel_l2(dsc, sr.grad) < 1e-4
        assert rel_l2(dsh, hr.grad) < 1e-4


@pytest.mark.parametrize("D", [64, 2048])
def test_rmsnorm(ops, dev, D):
    rows = 50
    x = (_mk((rows, D), dev, seed=16)).bfloat16()
    scale = 1 + 0.1 * _mk((D,), dev, seed=17)
    dy = _mk((rows, D), dev, seed=18).bfloat16()
    xr = x.float().requires_grad_(True)
    sr = scale.clone().requires_grad_(True)
    yr = xr * (sr * torch.rsqrt((xr ** 2).mean(-1, keepdim=True) + 1e-6))
    yr.backward(dy.float())
    y, rr = ops.rmsnorm_fwd(x, scale, eps=1e-6)
    assert rel_l2(y, yr) < 4e-3
    dx, dscale = ops.rmsnorm_bwd(dy, x, scale, rr)
    assert rel_l2(dx, xr.grad) < 1e-4
    assert rel_l2(dscale, sr.grad) < 1e-4


def test_colsum(ops, dev):
    x = _mk((777, 130), dev, seed=19)
    assert rel_l2(ops.colsum(x), x.sum(0)) < 1e-5
    xb = x.bfloat16()
    assert rel_l2(ops.colsum(xb), xb.float().sum(0)) < 1e-5
    # many rows, column slices of a wider matrix (row stride > columns, bases on and off the 16-byte grid), accumulating
    big = _mk((3000, 2048), dev, seed=20)
    for t in (big, big.bfloat16()):
        for view in (t[:2016, :1536], t[:, 8:1544], t[:, 4:1540], t[:1000, 512:]):
            want = view.float().sum(0)
            assert rel_l2(ops.colsum(view), want) < 1e-5, (t.dtype, tuple(view.shape), view.storage_offset())
            acc = torch.ones(view.shape[1], device=dev)
            assert rel_l2(ops.colsum(view, out=acc, accumulate=True), want + 1.0) < 1e-5


# ------------------------------------------------------------------------------------------------ elementwise
def test_swiglu_silu(ops, dev):
    h = _mk((37, 256), dev, seed=20).bfloat16()
    dout = _mk((37, 128), dev, seed=21).bfloat16()
    hr = h.float().requires_grad_(True)
    xx, gg = hr.chunk(2, -1)
    o = xx * F.silu(gg)
    o.backward(dout.float())
    assert rel_l2(ops.swiglu_fwd(h), o) < 4e-3
    assert rel_l2(ops.swiglu_bwd(dout, h), hr.grad) < 5e-3
    db = torch.zeros(256, device=dev)
    dh = ops.swiglu_bwd(dout, h, db)
    assert rel_l2(db, dh.float().sum(0)) < 1e-5       # fused bias gradient = column sums of what was written
    x = _mk((1000,), dev, seed=22)
    xr = x.clone().requires_grad_(True)
    F.silu(xr).backward(torch.ones_like(x) * 0.5)
    assert rel_l2(ops.silu_fwd(x), F.silu(x)) < 1e-6
    assert rel_l2(ops.silu_bwd(torch.full_like(x, 0.5), x), xr.grad) < 1e-5


@pytest.mark.parametrize("objective", ["v", "rectified_flow"])
def test_diffuse_and_mse(ops, dev, objective):
    B, C, T = 4, 16, 125
    x = _mk((B, C, T), dev, seed=23)
    n = _mk((B, C, T), dev, seed=24)
    t = torch.rand(B, device=dev)
    if objective == "v":
        a, s = torch.cos(t * math.pi / 2), torch.sin(t * math.pi / 2)
    else:
        a, s = 1 - t, t
    a, s = a[:, None, None], s[:, None, None]
    xt, tgt = ops.diffuse_fwd(x, n, t, objective)
    assert rel_l2(xt, x * a + n * s) < 1e-6
    assert rel_l2(tgt, (n * a - x * s) if objective == "v" else (n - x)) < 1e-6
    out = _mk((B, C, T), dev, seed=25)
    mask = torch.rand(B, T, device=dev) > 0.3
    for m in (None, mask):
        o = out.clone().requires_grad_(True)
        l = F.mse_loss(o, tgt, reduction="none")
        if m is not None:
            l = l[m.unsqueeze(1).repeat(1, C, 1)]
        l = l.mean()
        l.backward()
        loss, dout = ops.mse_loss(out, tgt, m)
        assert abs(loss.item() - l.item()) < 1e-5 * max(1, abs(l.item()))
        assert rel_l2(dout, o.grad) < 1e-5


def test_transpose_copy_cast_fourier(ops, dev):
    x = _mk((3, 50, 40), dev, seed=26)
    assert torch.equal(ops.transpose_2d(x), x.transpose(1, 2).contiguous())
    assert rel_l2(ops.transpose_2d(x, out_dtype=torch.bfloat16), x.transpose(1, 2)) < 4e-3
    # drop first row of each batch while transposing: in rows 1.., R=49
    o = ops.transpose_2d(x[:, 1:], R=49, Cn=40, in_batch_stride=x.stride(0), in_ld=x.stride(1))
    assert torch.equal(o, x[:, 1:].transpose(1, 2).contiguous())
    dst = torch.zeros((3, 51, 40), device=dev)
    ops.copy_rows(x, dst[:, 1:], 3, 50, 40, x.stride(0), x.stride(1), dst.stride(0), dst.stride(1))
    assert torch.equal(dst[:, 1:], x) and dst[:, 0].abs().max().item() == 0
    assert torch.equal(ops.cast(x, torch.bfloat16), x.bfloat16())
    t = torch.rand(5, device=dev)
    w = _mk((128,), dev, seed=27)
    f = 2 * math.pi * t[:, None] * w[None, :]
    assert rel_l2(ops.fourier_features(t, w), torch.cat([f.cos(), f.sin()], -1)) < 1e-4


@pytest.mark.parametrize("decoupled", [False, True])
def test_adam(ops, dev, decoupled):
    n = 1003
    p = _mk((n,), dev, seed=28)
    ref = torch.nn.Parameter(p.clone())
    opt = (torch.optim.AdamW if decoupled else torch.optim.Adam)([ref], lr=1e-2, weight_decay=0.1)
    m = torch.zeros(n, device=dev)
    v = torch.zeros(n, device=dev)
    pb = torch.empty(n, device=dev, dtype=torch.bfloat16)
    for step in range(1, 4):
        g = _mk((n,), dev, seed=30 + step)
        ref.grad = g.clone()
        opt.step()
        ops.adam_step(p, g, m, v, pb, lr=1e-2, weight_decay=0.1, decoupled=decoupled, step=step)
    assert rel_l2(p, ref.data) < 1e-5
    assert torch.equal(pb, p.bfloat16())


# ------------------------------------------------------------------------------------------------ attention
def _rope_tables(n, dev):
    inv = 1.0 / (10000 ** (torch.arange(0, 32, 2, device=dev).float() / 32))
    f = torch.arange(n, device=dev).float()[:, None] * inv[None, :]
    return f.cos().contiguous(), f.sin().contiguous()


def _rope_ref(t, cos, sin):
    # t [B,H,N,64]; rotary on first 32 dims (transformer.py:146-170)
    c = torch.cat([cos, cos], -1)[None, None]
    s = torch.cat([sin, sin], -1)[None, None]
    r, u = t[..., :32], t[..., 32:]
    x1, x2 = r[..., :16], r[..., 16:]
    rh = torch.cat([-x2, x1], -1)
    return torch.cat([r * c + rh * s, u], -1)


def _attn_ref(q, k, v, mask, rope, H, Hkv):
    # q [B,Nq,H*64], k/v [B,Nk,Hkv*64] fp32
    B, Nq, _ = q.shape
    Nk = k.shape[1]
    qh = q.view(B, Nq, H, 64).transpose(1, 2)
    kh = k.view(B, Nk, Hkv, 64).transpose(1, 2)
    vh = v.view(B, Nk, Hkv, 64).transpose(1, 2)
    if rope is not None:
        qh = _rope_ref(qh, *rope)
        kh = _rope_ref(kh, *rope)
    if H != Hkv:
        kh = kh.repeat_interleave(H // Hkv, 1)
        vh = vh.repeat_interleave(H // Hkv, 1)
    dots = qh @ kh.transpose(-1, -2) / 8.0
    if mask is not None:
        dots = dots.masked_fill(~mask[:, None, None, :], -torch.finfo(dots.dtype).max)
    o = dots.softmax(-1) @ vh
    return o.transpose(1, 2).reshape(B, Nq, H * 64)


@pytest.mark.parametrize("N,use_rope,use_mask", [(126, True, False), (126, True, True), (40, False, False),
                                                  (300, True, True), (128, True, False)])
def test_self_attention_fwd_bwd(ops, dev, N, use_rope, use_mask):
    B, H = 2, 3
    D = H * 64
    qkv = (_mk((B, N, 3 * D), dev, seed=40) * 0.8).bfloat16()
    dout = _mk((B, N, D), dev, seed=41).bfloat16()
    rope = _rope_tables(N, dev) if use_rope else None
    mask = None
    if use_mask:
        mask = torch.rand(B, N, device=dev) > 0.25
        mask[:, 0] = True
    qr = qkv.float().requires_grad_(True)
    q, k, v = qr.chunk(3, -1)
    ref = _attn_ref(q, k, v, mask, rope, H, H)
    ref.backward(dout.float())
    out, lse = ops.attention_fwd(qkv, qkv, qkv, ldq=3 * D, q_off=0, ldk=3 * D, k_off=D, ldv=3 * D, v_off=2 * D,
                                 B=B, H=H, Hkv=H, Nq=N, Nk=N, rope=rope, key_mask=mask)
    assert rel_l2(out, ref) < 1e-2, rel_l2(out, ref)
    dqkv = torch.zeros_like(qkv)
    ops.attention_bwd(qkv, qkv, qkv, out, dout, lse, dqkv, dqkv, dqkv, ldq=3 * D, q_off=0, ldk=3 * D, k_off=D,
                      ldv=3 * D, v_off=2 * D, B=B, H=H, Hkv=H, Nq=N, Nk=N, rope=rope, key_mask=mask)
    g = qr.grad
    for name, sl in (("dq", slice(0, D)), ("dk", slice(D, 2 * D)), ("dv", slice(2 * D, 3 * D))):
        e = rel_l2(dqkv[..., sl], g[..., sl])
        assert e < 2e-2, (name, e)


@pytest.mark.parametrize("Nq,Nk", [(126, 130), (126, 7), (260, 200)])
def test_cross_attention_gqa_fwd_bwd(ops, dev, Nq, Nk):
    B, H, Hkv = 2, 4, 2
    D, Dc = H * 64, Hkv * 64
    q = (_mk((B, Nq, D), dev, seed=42) * 0.8).bfloat16()
    kv = (_mk((B, Nk, 2 * Dc), dev, seed=43) * 0.8).bfloat16()
    dout = _mk((B, Nq, D), dev, seed=44).bfloat16()
    mask = torch.rand(B, Nk, device=dev) > 0.2
    mask[:, 0] = True
    qr = q.float().requires_grad_(True)
    kvr = kv.float().requires_grad_(True)
    k, v = kvr.chunk(2, -1)
    ref = _attn_ref(qr, k, v, mask, None, H, Hkv)
    ref.backward(dout.float())
    out, lse = ops.attention_fwd(q, kv, kv, ldq=D, q_off=0, ldk=2 * Dc, k_off=0, ldv=2 * Dc, v_off=Dc,
                                 B=B, H=H, Hkv=Hkv, Nq=Nq, Nk=Nk, key_mask=mask)
    assert rel_l2(out, ref) < 1e-2
    dq = torch.zeros_like(q)
    dkv = torch.zeros_like(kv)
    ops.attention_bwd(q, kv, kv, out, dout, lse, dq, dkv, dkv, ldq=D, q_off=0, ldk=2 * Dc, k_off=0, ldv=2 * Dc,
                      v_off=Dc, B=B, H=H, Hkv=Hkv, Nq=Nq, Nk=Nk, key_mask=mask)
    assert rel_l2(dq, qr.grad) < 2e-2
    assert rel_l2(dkv, kvr.grad) < 2e-2


# ------------------------------------------------------------------------------------------------ conv
def _wn(v, g):
    return g.view(-1, 1, 1) * v / v.flatten(1).norm(dim=1).view(-1, 1, 1)


@pytest.mark.parametrize("Cin,Cout,K,stride,dil,L", [(2, 128, 7, 1, 1, 300), (64, 64, 7, 1, 9, 200),
                                                      (64, 128, 4, 2, 1, 257), (128, 64, 16, 8, 1, 512),
                                                      (96, 40, 3, 1, 1, 77), (32, 32, 1, 1, 1, 130),
                                                      (24, 24, 7, 1, 3, 1500), (128, 2, 7, 1, 1, 1300),
                                                      (20, 1, 7, 1, 1, 515), (9, 17, 11, 1, 5, 700),
                                                      (24, 40, 8, 4, 1, 2100), (16, 72, 4, 2, 1, 3001),
                                                      (10, 24, 16, 8, 1, 2500), (8, 8, 6, 2, 1, 130),
                                                      # 8-wave tiles (x staged once per 128 output channels): k = 1, wide k = 7
                                                      (72, 200, 1, 1, 1, 2500), (136, 136, 1, 1, 1, 2600),
                                                      (16, 264, 7, 1, 3, 14000)])
@pytest.mark.parametrize("act", [0, 1, 2])
def test_conv1d(ops, dev, Cin, Cout, K, stride, dil, L, act):
    from kalle_audio_amd import conv_ops
    B = 2
    pad = math.ceil(stride / 2) if stride > 1 else dil * (K - 1) // 2
    x = _mk((B, Cin, L), dev, seed=50)
    v = _mk((Cout, Cin, K), dev, seed=51) * 0.2
    g = 1 + 0.1 * _mk((Cout,), dev, seed=52)
    bias = _mk((Cout,), dev, seed=53)
    alpha = 0.3 * _mk((Cin,), dev, seed=54)
    beta = 0.3 * _mk((Cin,), dev, seed=55)
    w = _wn(v, g)
    xa = x
    if act == 1:
        xa = x + torch.sin(x * alpha.exp()[None, :, None]) ** 2 / (beta.exp()[None, :, None] + 1e-9)
    elif act == 2:
        xa = F.elu(x)
    ref = F.conv1d(xa, w, bias, stride=stride, padding=pad, dilation=dil)
    wp = conv_ops.weight_norm_fold(v, g, transposed=False)
    res = _mk(tuple(ref.shape), dev, seed=56) if (stride == 1 and Cin == Cout) else None
    y = conv_ops.conv1d(x, wp, bias, Cout=Cout, K=K, stride=stride, padding=pad, dilation=dil, act=act,
                        alpha=alpha, beta=beta, residual=res, post=1)
    r = ref + res if res is not None else ref
    assert rel_l2(y, torch.tanh(r)) < 2e-5


@pytest.mark.parametrize("Cin,Cout,stride,L", [(128, 64, 2, 100), (64, 32, 4, 130), (64, 48, 8, 65), (32, 16, 5, 40),
                                               (24, 40, 4, 700), (16, 2, 2, 1100), (12, 9, 3, 333), (40, 24, 8, 300)])
def test_conv_transpose1d(ops, dev, Cin, Cout, stride, L):
    from kalle_audio_amd import conv_ops
    B = 2
    K = 2 * stride + stride % 2
    pad = math.ceil(stride / 2)
    x = _mk((B, Cin, L), dev, seed=60)
    v = _mk((Cin, Cout, K), dev, seed=61) * 0.2
    g = 1 + 0.1 * _mk((Cin,), dev, seed=62)
    bias = _mk((Cout,), dev, seed=63)
    alpha = 0.3 * _mk((Cin,), dev, seed=64)
    beta = 0.3 * _mk((Cin,), dev, seed=65)
    w = _wn(v, g)
    xa = x + torch.sin(x * alpha.exp()[None, :, None]) ** 2 / (beta.exp()[None, :, None] + 1e-9)
    ref = F.conv_transpose1d(xa, w, bias, stride=stride, padding=pad)
    wp = conv_ops.weight_norm_fold(v, g, transposed=True)
    y = conv_ops.conv_transpose1d(x, wp, bias, Cout=Cout, K=K, stride=stride, padding=pad, act=1, alpha=alpha,
                                  beta=beta)
    assert y.shape == ref.shape
    assert rel_l2(y, ref) < 2e-5


def test_snake(ops, dev):
    from kalle_audio_amd import conv_ops
    x = _mk((2, 8, 100), dev, seed=70)
    a = 0.3 * _mk((8,), dev, seed=71)
    b = 0.3 * _mk((8,), dev, seed=72)
    ref = x + torch.sin(x * a.exp()[None, :, None]) ** 2 / (b.exp()[None, :, None] + 1e-9)
    assert rel_l2(conv_ops.snake_beta(x, a, b), ref) < 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("L", [1, 7, 255, 256, 257, 1000])
@pytest.mark.parametrize("beta", [False, True])
def test_act1d_matches_oracle(ops, dev, L, beta):
    """fused 2x-up FIR -> snake -> 2x-down FIR kernel vs the restated alias-free-torch Activation1d (edges included)"""
    from kalle_audio_amd import conv_ops
    torch.manual_seed(L)
    x = torch.randn(2, 5, L)
    a, b = 0.3 * torch.randn(5), 0.3 * torch.randn(5)
    want = ko.activation1d(x, a, b if beta else None, logscale=True)
    filt = conv_ops.kaiser_sinc_filter12(dev)
    assert torch.equal(filt.cpu(), ko.kaiser_sinc_filter1d(0.25, 0.3, 12))
    got = conv_ops.act1d(x.to(dev), filt, a.to(dev), (b if beta else a).to(dev), True)
    assert (got.cpu() - want).abs().max() < 2e-5
    gb = conv_ops.act1d(x.to(dev).bfloat16(), filt, a.to(dev), (b if beta else a).to(dev), True)
    assert (gb.float().cpu() - want).abs().max() < 0.06


@pytest.mark.gpu
def test_conv1d_causal_leaky_gate_scale_accumulate(ops, dev):
    """the conv options the mel-VAE adds: left-only padding, LeakyReLU / WaveNet-gate input activation, output scale,
    accumulate into y, trimmed transposed conv"""
    from kalle_audio_amd import conv_ops
    torch.manual_seed(5)
    Bn, Cin, Cout, K, d, L = 2, 24, 40, 5, 3, 301
    x = torch.randn(Bn, Cin, L)
    w = torch.randn(Cout, Cin, K) / (Cin * K) ** 0.5
    bias = torch.randn(Cout)
    res = torch.randn(Bn, Cout, L)
    y0 = torch.randn(Bn, Cout, L)
    wp = conv_ops.weight_norm_fold(w.to(dev), None)
    ref = F.conv1d(F.pad(F.leaky_relu(x, 0.2), (d * (K - 1), 0)), w, bias, dilation=d)
    got = conv_ops.conv1d(x.to(dev), wp, bias.to(dev), Cout=Cout, K=K, padding=d * (K - 1), pad_right=0, dilation=d,
                          act=3, act_param=0.2)
    assert (got.cpu() - ref).abs().max() < 1e-4
    acc = y0.to(dev).clone()
    got = conv_ops.conv1d(x.to(dev), wp, bias.to(dev), Cout=Cout, K=K, padding=d * (K - 1), pad_right=0, dilation=d,
                          act=3, act_param=0.2, residual=res.to(dev), out_scale=0.5, accumulate_into=acc)
    assert got.data_ptr() == acc.data_ptr()
    assert (got.cpu() - (y0 + 0.5 * (ref + res))).abs().max() < 1e-4
    xg = torch.randn(Bn, 2 * Cin, L)
    refg = F.conv1d(torch.tanh(xg[:, :Cin]) * torch.sigmoid(xg[:, Cin:]), w, bias, padding=2)
    gotg = conv_ops.conv1d(xg.to(dev), wp, bias.to(dev), Cout=Cout, K=K, padding=2, act=4)
    assert (gotg.cpu() - refg).abs().max() < 1e-4
    s = 4
    wt = torch.randn(Cin, Cout, 2 * s) / (Cin * 2) ** 0.5
    wtp = conv_ops.weight_norm_fold(wt.to(dev), None, transposed=True)
    reft = F.conv_transpose1d(x, wt, bias, stride=s)[:, :, :-s]
    gott = conv_ops.conv_transpose1d(x.to(dev), wtp, bias.to(dev), Cout=Cout, K=2 * s, stride=s, padding=0, trim=s)
    assert gott.shape == reft.shape and (gott.cpu() - reft).abs().max() < 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("force_v1", [False, True])
def test_conv_post_activation_and_fallback_kernels(ops, dev, force_v1, monkeypatch):
    """output-side activation (the consumer's input activation applied at the producer's store) for conv and transposed
    conv, on the main kernels and on the fallback kernels (KALLE_CONV_V1=1)"""
    from kalle_audio_amd import conv_ops
    if force_v1:
        monkeypatch.setenv("KALLE_CONV_V1", "1")
    torch.manual_seed(11)
    Bn, Cin, Cout, K, L = 2, 40, 24, 7, 900
    x = torch.randn(Bn, Cin, L, device=dev)
    w = torch.randn(Cout, Cin, K, device=dev) / (Cin * K) ** 0.5
    bias = torch.randn(Cout, device=dev)
    a, b = 0.3 * torch.randn(Cout, device=dev), 0.3 * torch.randn(Cout, device=dev)
    res = torch.randn(Bn, Cout, L, device=dev)
    snake = lambda t: t + torch.sin(t * a.exp()[None, :, None]) ** 2 / (b.exp()[None, :, None] + 1e-9)
    wp = conv_ops.weight_norm_fold(w, None)
    ref = snake(F.conv1d(x, w, bias, padding=9, dilation=3) + res)
    got = conv_ops.conv1d(x, wp, bias, Cout=Cout, K=K, padding=9, dilation=3, residual=res, post_act=(1, a, b, True, 0.0))
    assert rel_l2(got, ref) < 2e-5
    ref = F.elu(F.conv1d(x, w, bias, padding=3))
    got = conv_ops.conv1d(x, wp, bias, Cout=Cout, K=K, padding=3, post_act=(2, None, None, False, 0.0))
    assert rel_l2(got, ref) < 2e-5
    if not force_v1:
        wt = torch.randn(Cin, Cout, 8, device=dev) / (Cin * 2) ** 0.5
        wtp = conv_ops.weight_norm_fold(wt, None, transposed=True)
        ref = snake(F.conv_transpose1d(x, wt, bias, stride=4, padding=2))
        got = conv_ops.conv_transpose1d(x, wtp, bias, Cout=Cout, K=8, stride=4, padding=2, post_act=(1, a, b, True, 0.0))
        assert rel_l2(got, ref) < 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize("N,use_mask", [(77, False), (128, True), (300, True), (513, False)])
def test_causal_gqa_full_rotary_attention(ops, dev, N, use_mask):
    """the Llama decoder's attention: causal, GQA (H=4 over Hkv=2), rotary over all 64 dims, key-padding mask; q/k/v read in
    place from one fused projection output [B, N, (H + 2 Hkv) * 64]"""
    B, H, Hkv = 2, 4, 2
    Dq, Dk = H * 64, Hkv * 64
    ld = Dq + 2 * Dk
    qkv = (_mk((B, N, ld), dev, seed=90) * 0.8).bfloat16()
    dout = _mk((B, N, Dq), dev, seed=91).bfloat16()
    pos = torch.arange(N, device=dev, dtype=torch.float32)
    inv = 1.0 / (500000.0 ** (torch.arange(0, 64, 2, device=dev, dtype=torch.float32) / 64))
    fr = pos[:, None] * inv[None, :]
    cos, sin = fr.cos().contiguous(), fr.sin().contiguous()          # [N, 32]
    mask = None
    if use_mask:
        mask = torch.ones(B, N, dtype=torch.bool, device=dev)
        mask[0, N - N // 5:] = False                                  # right padding on one sample
    qr = qkv.float().requires_grad_(True)
    q, k, v = qr[..., :Dq], qr[..., Dq:Dq + Dk], qr[..., Dq + Dk:]

    def rope(t):   # [B, h, N, 64], HF apply_rotary_pos_emb
        c = torch.cat([cos, cos], -1)[None, None]
        s = torch.cat([sin, sin], -1)[None, None]
        return t * c + torch.cat([-t[..., 32:], t[..., :32]], -1) * s
    qh = rope(q.reshape(B, N, H, 64).transpose(1, 2))
    kh = rope(k.reshape(B, N, Hkv, 64).transpose(1, 2)).repeat_interleave(H // Hkv, 1)
    vh = v.reshape(B, N, Hkv, 64).transpose(1, 2).repeat_interleave(H // Hkv, 1)
    dots = qh @ kh.transpose(-1, -2) / 8.0
    allow = torch.ones(N, N, dtype=torch.bool, device=dev).tril()[None, None]
    if mask is not None:
        allow = allow & mask[:, None, None, :]
    dots = dots.masked_fill(~allow, -torch.finfo(dots.dtype).max)
    ref = (dots.softmax(-1) @ vh).transpose(1, 2).reshape(B, N, Dq)
    valid = mask if mask is not None else torch.ones(B, N, dtype=torch.bool, device=dev)
    ref.backward(dout.float() * valid[..., None])                     # padded query rows carry no gradient
    kw = dict(ldq=ld, q_off=0, ldk=ld, k_off=Dq, ldv=ld, v_off=Dq + Dk, B=B, H=H, Hkv=Hkv, Nq=N, Nk=N,
              rope=(cos, sin), key_mask=mask, causal=True)
    out, lse = ops.attention_fwd(qkv, qkv, qkv, **kw)
    assert rel_l2(out[valid], ref[valid]) < 1e-2, rel_l2(out[valid], ref[valid])
    dqkv = torch.zeros_like(qkv)
    ops.attention_bwd(qkv, qkv, qkv, out, (dout * valid[..., None]).contiguous(), lse, dqkv, dqkv, dqkv, **kw)
    g = qr.grad
    for name, sl in (("dq", slice(0, Dq)), ("dk", slice(Dq, Dq + Dk)), ("dv", slice(Dq + Dk, ld))):
        e = rel_l2(dqkv[..., sl], g[..., sl])
        assert e < 2e-2, (name, e)


@pytest.mark.gpu
def test_peak_normalize_int16_and_axpby_inplace(ops, dev):
    x = torch.randn(2, 3001, device=dev) * 0.3
    got, peak = ops.peak_normalize_int16(x)
    want = x.div(x.abs().max()).clamp(-1, 1).mul(32767).to(torch.int16)
    assert peak.item() == x.abs().max().item()
    assert (got.int() - want.int()).abs().max() <= 1 and (got != want).float().mean() < 1e-3
    assert got.abs().max().item() == 32767
    a, b = torch.randn(1000, device=dev), torch.randn(1000, device=dev)
    want = 0.9 * a + 0.1 * b
    ops.axpby(a, b, 0.9, 0.1, out=a)
    assert torch.allclose(a, want, atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("N,K", [(3072, 2048), (2048, 8192), (130, 72), (17, 8)])
def test_gemv_single_row(ops, dev, N, K):
    """the M = 1 path of ops.gemm (decoding against the KV cache): same result as the MFMA GEMM on a padded problem"""
    x = _mk((1, K), dev, seed=95).bfloat16()
    w = (_mk((N, K), dev, seed=96) * 0.1).bfloat16()
    res = _mk((1, N), dev, seed=97)
    ref = x.float() @ w.float().T
    got = ops.gemm(x, w)
    assert got.dtype == torch.bfloat16 and rel_l2(got, ref) < 4e-3
    got = ops.gemm(x, w, out_dtype=torch.float32, residual=res)
    assert rel_l2(got, ref + res) < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("mix", [None, "1,3", "2,5", "3,12", "2,0"])
def test_wgrad_mixed_split_k(ops, dev, mix, monkeypatch):
    """weight-gradient GEMM (TN, fp32, atomics) with some tiles cut into sa and the others into sa + 1 K slices (1-D grid,
    long slices first): same result as fp32 matmul for forced plans and for the planner's own choice"""
    if mix:
        monkeypatch.setenv("KALLE_GEMM_MIX", mix)
    M, N, K = 768, 1024, 8192                                            # 12 tiles of 256 x 256, 128 K-tiles
    dy = (_mk((K, M), dev, seed=150) * 0.5).bfloat16()
    x = (_mk((K, N), dev, seed=151) * 0.5).bfloat16()
    ref = dy.float().T @ x.float()
    got = ops.gemm(dy, x, a_kmajor=True, b_kmajor=True, out_dtype=torch.float32)
    assert rel_l2(got, ref) < 2e-6, rel_l2(got, ref)
    acc = torch.ones(M, N, device=dev)
    ops.gemm(dy, x, a_kmajor=True, b_kmajor=True, out=acc, accumulate=True)
    assert rel_l2(acc, ref + 1.0) < 2e-6


@pytest.mark.gpu
@pytest.mark.parametrize("M,N", [(12288, 1536), (1536, 6144), (4608, 1536)])
def test_wgrad_planned_split_k_at_bench_shapes(ops, dev, M, N):
    """the weight-gradient shapes of the bench step (K = 32256 tokens) under the planner's own split-K choice (mixed for the
    two feed-forward shapes): against fp64-accumulated fp32 matmul, and accumulation on top of an existing gradient"""
    K = 32256
    dy = (_mk((K, M), dev, seed=160) * 0.25).bfloat16()
    x = (_mk((K, N), dev, seed=161) * 0.25).bfloat16()
    ref = (dy.float().T.double() @ x.float().double()).float()
    out = torch.full((M, N), 3.0, device=dev)
    ops.gemm(dy, x, a_kmajor=True, b_kmajor=True, out=out)                     # overwrites (clears C itself)
    assert rel_l2(out, ref) < 3e-6, rel_l2(out, ref)
    ops.gemm(dy, x, a_kmajor=True, b_kmajor=True, out=out, accumulate=True)    # adds on top
    assert rel_l2(out, 2 * ref) < 3e-6


@pytest.mark.gpu
@pytest.mark.parametrize("N,K", [(16384, 2048), (2048 + 512, 2048), (4098, 264), (1030, 8192)])
def test_gemv_rows_per_wave_and_bias(ops, dev, N, K):
    """wide outputs walk several row pairs per wave; a Linear bias rides in the residual slot of the single-row kernel"""
    x = _mk((1, K), dev, seed=98).bfloat16()
    w = (_mk((N, K), dev, seed=99) * 0.1).bfloat16()
    bias = _mk((N,), dev, seed=100)
    ref = x.float() @ w.float().T
    assert rel_l2(ops.gemm(x, w, out_dtype=torch.float32), ref) < 1e-5
    assert rel_l2(ops.gemm(x, w, out_dtype=torch.float32, bias=bias), ref + bias) < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("Nk", [1, 37, 300, 1500])
@pytest.mark.parametrize("rot,H,Hkv,use_mask", [(64, 4, 2, False), (64, 4, 2, True), (32, 3, 3, False), (0, 2, 1, True)])
def test_single_query_attention_matches_tiled_kernel(ops, dev, Nk, rot, H, Hkv, use_mask):
    """Nq == 1 (decoding against a KV cache) runs the vector-ALU single-query kernel: same output and lse as the last
    query row of the tiled MFMA kernel over the whole causal sequence"""
    B = 2
    Dq, Dk = H * 64, Hkv * 64
    ld = Dq + 2 * Dk
    qkv = (_mk((B, Nk, ld), dev, seed=120) * 0.8).bfloat16()
    rope = None
    if rot:
        inv = 1.0 / (10000.0 ** (torch.arange(0, rot, 2, device=dev, dtype=torch.float32) / rot))
        fr = torch.arange(Nk, device=dev, dtype=torch.float32)[:, None] * inv[None, :]
        rope = (fr.cos().contiguous(), fr.sin().contiguous())
    mask = None
    if use_mask:
        mask = torch.rand(B, Nk, device=dev) > 0.3
        mask[:, -1] = True
    kw = dict(ldk=ld, k_off=Dq, ldv=ld, v_off=Dq + Dk, B=B, H=H, Hkv=Hkv, Nk=Nk, rope=rope, key_mask=mask, causal=True)
    full, lse_full = ops.attention_fwd(qkv, qkv, qkv, ldq=ld, q_off=0, Nq=Nk, **kw)
    if Nk == 1:   # the tiled reference itself would take the single-query path: check against the value row instead
        want = qkv[:, :, Dq + Dk:].reshape(B, 1, Hkv, 64).repeat_interleave(H // Hkv, 2).reshape(B, 1, Dq)
        got, _ = ops.attention_fwd(qkv, qkv, qkv, ldq=ld, q_off=0, Nq=1, **kw)
        assert rel_l2(got, want) < 1e-6
        return
    qlast = qkv[:, -1:, :Dq].contiguous()                               # [B, 1, Dq], ldq = Dq
    got, lse = ops.attention_fwd(qlast, qkv, qkv, ldq=Dq, q_off=0, Nq=1, **kw)
    assert rel_l2(got, full[:, -1:]) < 4e-3, rel_l2(got, full[:, -1:])
    assert (lse[..., 0] - lse_full[..., -1]).abs().max() < 2e-3


@pytest.mark.gpu
@pytest.mark.parametrize("Cin,Cout,K,dil,L,act", [(64, 256, 7, 1, 215, 0), (96, 520, 7, 9, 333, 1), (128, 300, 1, 1, 77, 2),
                                                  (40, 256, 3, 3, 1000, 1), (1024, 1024, 7, 3, 100, 1),
                                                  (2048, 128, 3, 1, 215, 1)])   # last two: input channels split over workgroups
@pytest.mark.parametrize("stride", [1])
def test_conv1d_channels_per_lane_kernel(ops, dev, Cin, Cout, K, dil, L, act, stride, monkeypatch):
    """the few-positions / many-channels kernel (pad+activate pass, then lane = 4 output channels): forced on, against torch"""
    from kalle_audio_amd import conv_ops
    monkeypatch.setenv("KALLE_CONV_CFIRST", "1")
    B = 2
    pad = dil * (K - 1) // 2
    x = _mk((B, Cin, L), dev, seed=110)
    w = _mk((Cout, Cin, K), dev, seed=111) / (Cin * K) ** 0.5
    bias = _mk((Cout,), dev, seed=112)
    alpha, beta = 0.3 * _mk((Cin,), dev, seed=113), 0.3 * _mk((Cin,), dev, seed=114)
    pa, pb = 0.3 * _mk((Cout,), dev, seed=115), 0.3 * _mk((Cout,), dev, seed=116)
    res = _mk((B, Cout, L), dev, seed=117)
    xa = x
    if act == 1:
        xa = x + torch.sin(x * alpha.exp()[None, :, None]) ** 2 / (beta.exp()[None, :, None] + 1e-9)
    elif act == 2:
        xa = F.elu(x)
    r = F.conv1d(xa, w, bias, padding=pad, dilation=dil) + res
    ref = r + torch.sin(r * pa.exp()[None, :, None]) ** 2 / (pb.exp()[None, :, None] + 1e-9)
    wp = conv_ops.weight_norm_fold(w, None)
    got = conv_ops.conv1d(x, wp, bias, Cout=Cout, K=K, padding=pad, dilation=dil, act=act, alpha=alpha, beta=beta,
                          residual=res, post_act=(1, pa, pb, True, 0.0))
    assert rel_l2(got, ref) < 3e-5, rel_l2(got, ref)
    y0 = _mk((B, Cout, L), dev, seed=118)
    acc = y0.clone()
    got = conv_ops.conv1d(x, wp, None, Cout=Cout, K=K, padding=pad, dilation=dil, out_scale=0.5, accumulate_into=acc, post=1)
    assert rel_l2(got, torch.tanh(y0 + 0.5 * F.conv1d(x, w, None, padding=pad, dilation=dil))) < 3e-5


@pytest.mark.gpu
@pytest.mark.parametrize("Cin,Cout,stride,L,trim", [(64, 256, 8, 27, 0), (96, 300, 4, 130, 0), (128, 256, 2, 77, 2),
                                                    (40, 264, 5, 33, 0), (2048, 1024, 8, 20, 0)])
def test_conv_transpose1d_channels_per_lane_kernel(ops, dev, Cin, Cout, stride, L, trim, monkeypatch):
    from kalle_audio_amd import conv_ops
    monkeypatch.setenv("KALLE_CONV_CFIRST", "1")
    B = 2
    K = 2 * stride + stride % 2
    pad = 0 if trim else math.ceil(stride / 2)
    x = _mk((B, Cin, L), dev, seed=120)
    w = _mk((Cin, Cout, K), dev, seed=121) / (Cin * 2) ** 0.5
    bias = _mk((Cout,), dev, seed=122)
    alpha, beta = 0.3 * _mk((Cin,), dev, seed=123), 0.3 * _mk((Cin,), dev, seed=124)
    xa = x + torch.sin(x * alpha.exp()[None, :, None]) ** 2 / (beta.exp()[None, :, None] + 1e-9)
    ref = F.conv_transpose1d(xa, w, bias, stride=stride, padding=pad)
    if trim:
        ref = ref[:, :, :-trim]
    wp = conv_ops.weight_norm_fold(w, None, transposed=True)
    got = conv_ops.conv_transpose1d(x, wp, bias, Cout=Cout, K=K, stride=stride, padding=pad, act=1, alpha=alpha, beta=beta,
                                    trim=trim)
    assert got.shape == ref.shape and rel_l2(got, ref) < 3e-5, rel_l2(got, ref)


@pytest.mark.gpu
@pytest.mark.parametrize("Cin,Cout,K,stride,pad,L", [(64, 256, 16, 8, 4, 1720), (96, 300, 8, 4, 2, 333), (40, 256, 4, 2, 1, 77),
                                                     (24, 264, 6, 3, 2, 100), (16, 256, 8, 4, 3, 129),
                                                     (512, 520, 16, 8, 4, 800)])      # (workgroup-level channel split)
def test_strided_conv_channels_per_lane_kernel(ops, dev, Cin, Cout, K, stride, pad, L, monkeypatch):
    """strided convs on the channels-per-lane kernel: the padded copy of x is de-interleaved into `stride` phase rows"""
    from kalle_audio_amd import conv_ops
    monkeypatch.setenv("KALLE_CONV_CFIRST", "1")
    B = 2
    x = _mk((B, Cin, L), dev, seed=130)
    w = _mk((Cout, Cin, K), dev, seed=131) / (Cin * K) ** 0.5
    bias = _mk((Cout,), dev, seed=132)
    ref = F.conv1d(F.elu(x), w, bias, stride=stride, padding=pad)
    wp = conv_ops.weight_norm_fold(w, None)
    got = conv_ops.conv1d(x, wp, bias, Cout=Cout, K=K, stride=stride, padding=pad, act=2)
    assert got.shape == ref.shape and rel_l2(got, ref) < 3e-5, rel_l2(got, ref)


@pytest.mark.gpu
def test_channels_per_lane_convs_with_many_batch_channel_rows(ops, dev):
    """B x Cin > 65535 (all chunks of a long clip on the batch axis of a 1024-channel layer): the pad + activate pass carries its rows
    on grid x; the strided 256+-channel convs pick the channels-per-lane kernel at every batch size (dispatch rule of round 3)"""
    from kalle_audio_amd import conv_ops
    B, Cin, Cout, K, stride, pad, L = 66, 1024, 256, 4, 2, 1, 24
    assert B * Cin > 65535
    x = _mk((B, Cin, L), dev, seed=140)
    w = _mk((Cout, Cin, K), dev, seed=141) / (Cin * K) ** 0.5
    bias = _mk((Cout,), dev, seed=142)
    wp = conv_ops.weight_norm_fold(w, None)
    got = conv_ops.conv1d(x, wp, bias, Cout=Cout, K=K, stride=stride, padding=pad, act=2)
    ref = F.conv1d(F.elu(x), w, bias, stride=stride, padding=pad)
    assert got.shape == ref.shape and rel_l2(got, ref) < 3e-5, rel_l2(got, ref)
    w1 = _mk((Cout, Cin, 3), dev, seed=143) / (Cin * 3) ** 0.5
    got = conv_ops.conv1d(x, conv_ops.weight_norm_fold(w1, None), bias, Cout=Cout, K=3, padding=1, act=2)
    assert rel_l2(got, F.conv1d(F.elu(x), w1, bias, padding=1)) < 3e-5
    y = conv_ops.activate(x, 2)
    assert torch.equal(y, F.elu(x)) or rel_l2(y, F.elu(x)) < 1e-6
