"""What a persistent GEMM does when another kernel holds some CUs (RCCL's all-reduce on the communication stream, the optimizer
slice on its side stream): python tools/cu_hold_probe.py   (run once with KALLE_GEMM_DYNAMIC=0 and once with =1)

A stand-in kernel (kalle_debug_hold_cus: C workgroups x 256 threads x 16 KiB LDS, resident for a fixed time) is launched on a side
stream and, once it is running, a GEMM of the train step on the main stream; the GEMM's own duration is measured with events.
Static walk: the workgroups that found no CU start when the stand-in ends (or when the first ones have finished all their
tiles); dynamic hand-out: the resident workgroups take over their tiles."""
import ctypes, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kalle_audio_amd import _lib, ops

dev = torch.device("cuda")
lib = _lib.load()
mk = lambda r, c: (torch.randn(r, c, device=dev) * 0.5).bfloat16()
side = torch.cuda.Stream()
shapes = [("qkv nt", 32256, 4608, 1536, False), ("dgrad nn", 32256, 1536, 1536, True), ("ff-in dgrad nn", 32256, 1536, 12288, True)]
mode = "dynamic" if os.environ.get("KALLE_GEMM_DYNAMIC", "1") != "0" else "static"
for name, M, N, K, bk in shapes:
    a = mk(M, K)
    b = mk(K, N) if bk else mk(N, K)
    fn = lambda: ops.gemm(a, b, b_kmajor=bk)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()

    def timed(held, hold_us):
        ts = []
        for _ in range(7):
            torch.cuda.synchronize()
            if held:
                with torch.cuda.stream(side):
                    rc = lib.kalle_debug_hold_cus(held, 16384, hold_us, ctypes.c_void_p(side.cuda_stream))
                    assert rc == 0, rc
                time.sleep(0.0003)          # the stand-in is resident before the GEMM's workgroups are dispatched
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        return sorted(ts)[len(ts) // 2]

    alone = timed(0, 0)
    line = f"{mode:8s} {name:16s} alone {alone:7.1f} us |"
    for held in (16, 32, 64):
        t = timed(held, int(alone * 3))      # held for the whole GEMM and beyond
        line += f" {held} CUs held: {t:7.1f} us ({t / alone:4.2f}x, ideal {256 / (256 - held):4.2f}x) |"
    t = timed(32, int(alone * 0.5))          # held for the first half of the GEMM only
    line += f" 32 CUs for half the GEMM: {t:7.1f} us ({t / alone:4.2f}x)"
    print(line, flush=True)
