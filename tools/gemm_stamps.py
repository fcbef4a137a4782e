"""Where does a 256 x 256 tile of the K = 1536 GEMMs spend its time?  In-kernel s_memrealtime stamps (kalle_gemm_debug_stamps)
per workgroup: entry -> first K-tile landed -> main loop done -> epilogue barrier -> stores issued -> stores acknowledged.
python tools/gemm_stamps.py [shape ...]   shapes: qkv | out_res | glu2 | glu1 | dgrad | ff2_res"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kalle_audio_amd import ops, _lib
dev = torch.device("cuda")
M, D = 32256, 1536
mk = lambda r, c: (torch.randn(r, c, device=dev) * 0.5).bfloat16()
lib = _lib.load()


def case(name):
    if name == "qkv":
        x, w = mk(M, D), mk(3 * D, D)
        return 126 * 18, 2.0 * M * 3 * D * D, lambda: ops.gemm(x, w)
    if name == "out_res":
        x, w, res = mk(M, D), mk(D, D), torch.randn(M, D, device=dev)
        return 126 * 6, 2.0 * M * D * D, lambda: ops.gemm(x, w, out_dtype=torch.float32, residual=res)
    if name == "ff2_res":
        x, w, res, b = mk(M, 4 * D), mk(D, 4 * D), torch.randn(M, D, device=dev), torch.randn(D, device=dev)
        return 126 * 6, 2.0 * M * D * 4 * D, lambda: ops.gemm(x, w, bias=b, out_dtype=torch.float32, residual=res)
    if name == "dgrad":
        dy, w = mk(M, D), mk(D, D)
        return 126 * 6, 2.0 * M * D * D, lambda: ops.gemm(dy, w, b_kmajor=True)
    if name == "glu1":
        x, w, b = mk(M, D), mk(8 * D, D), torch.randn(8 * D, device=dev)
        hf = torch.empty(M, 8 * D, device=dev, dtype=torch.bfloat16)
        act = torch.empty(M, 4 * D, device=dev, dtype=torch.bfloat16)
        return 126 * 48, 2.0 * M * 8 * D * D, lambda: ops.gemm(x, w, bias=b, out=hf, glu_mode=1, glu_inner=4 * D, glu_aux=act)
    if name == "glu2":
        gb, w2 = mk(M, D), mk(D, 4 * D)
        hf = mk(M, 8 * D)
        dhf = torch.empty_like(hf)
        db = torch.zeros(8 * D, device=dev)
        return 126 * 24, 2.0 * M * 4 * D * D, lambda: ops.gemm(gb, w2, b_kmajor=True, out=dhf, N=4 * D, glu_mode=2,
                                                               glu_inner=4 * D, glu_aux=hf, glu_dbias=db)
    raise SystemExit(name)


for name in (sys.argv[1:] or ["qkv", "out_res", "glu2", "dgrad"]):
    wgs, flops, fn = case(name)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    st = torch.zeros(wgs * 2 * 8, device=dev, dtype=torch.int64)
    lib.kalle_gemm_debug_stamps(ctypes.c_void_p(st.data_ptr()))
    fn()
    torch.cuda.synchronize()
    lib.kalle_gemm_debug_stamps(None)
    s = st.view(wgs, 2, 8).cpu().double() * 0.01          # us
    t0 = s[:, :, 0].min()
    s = s - t0
    print(f"== {name}: {us:.1f} us/launch = {flops / us / 1e6:.0f} TFLOP/s, {wgs} workgroups = {wgs / 256:.2f} rounds; stamped launch spans "
          f"{s[:, :, 5].max():.1f} us")
    lab = ["entry->landed", "main loop", "drain+barrier", "epilogue issue", "store ack"]
    for w in range(2):
        d = [s[:, w, i + 1] - s[:, w, i] for i in range(5)]
        print(f"   wave {0 if w == 0 else 7}: " + "  ".join(f"{l} {x.median():.2f} (p90 {x.quantile(0.9):.2f})" for l, x in zip(lab, d))
              + f"  | whole tile {(s[:, w, 5] - s[:, w, 0]).median():.2f}")
    # the rounds: start times of the workgroups in dispatch order, in buckets
    start = s[:, 0, 0].sort().values
    print("   entry time quantiles (us): " + " ".join(f"{start[int(q * (wgs - 1))]:.1f}" for q in (0, .1, .25, .4, .5, .6, .75, .9, 1.0)))
    end = s[:, 0, 4]
    print(f"   gap between a tile's stores-issued and the next entry on the chip: first-round tiles end at median {end[:256].median():.1f}, "
          f"the 257th..512th workgroups enter at median {s[256:512, 0, 0].median() if wgs > 300 else float('nan'):.1f}")
