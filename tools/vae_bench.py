"""Oobleck VAE conv path timing (Stable-Audio-Open layout: channels=128, c_mults=[1,2,4,8,16], strides=[2,4,4,8,8],
latent 64, snake) - decode z [B,64,215] and encode wav [B,2,441000].  python tools/vae_bench.py [B] [dtype]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kalle_audio_amd
kalle_audio_amd.install()
from stable_audio_tools.models.factory import create_model_from_config
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
half = len(sys.argv) > 2 and sys.argv[2] == "bf16"
cfg = {"model_type": "autoencoder", "sample_rate": 44100, "sample_size": 441000, "audio_channels": 2,
       "model": {"encoder": {"type": "oobleck", "config": {"in_channels": 2, "channels": 128, "c_mults": [1, 2, 4, 8, 16],
                                                          "strides": [2, 4, 4, 8, 8], "latent_dim": 128, "use_snake": True}},
                 "decoder": {"type": "oobleck", "config": {"out_channels": 2, "channels": 128, "c_mults": [1, 2, 4, 8, 16],
                                                          "strides": [2, 4, 4, 8, 8], "latent_dim": 64, "use_snake": True,
                                                          "final_tanh": False}},
                 "bottleneck": {"type": "vae"}, "latent_dim": 64, "downsampling_ratio": 2048, "io_channels": 2}}
LAT, RATIO = 64, 2048
if "--ref-defaults" in sys.argv:
    # the in-tree defaults of OobleckEncoder / OobleckDecoder (autoencoders.py:117-124, 151-158): 4 levels, latent 32
    for part, lat in (("encoder", 64), ("decoder", 32)):
        cfg["model"][part]["config"].update(c_mults=[1, 2, 4, 8], strides=[2, 4, 8, 8], latent_dim=lat)
    cfg["model"].update(latent_dim=32, downsampling_ratio=512)
    LAT, RATIO = 32, 512
dev = torch.device("cuda")
torch.manual_seed(0)
with torch.device(dev):
    ae = create_model_from_config(cfg)
ae.eval().requires_grad_(False)
dt = torch.bfloat16 if half else torch.float32
z = torch.randn(B, LAT, 440320 // RATIO, device=dev).to(dt)
wav = (torch.rand(B, 2, 440320, device=dev) * 2 - 1).to(dt)
with torch.no_grad():
    for name, fn in (("decode", lambda: ae.decode(z)), ("encode", lambda: ae.encode(wav))):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 3
        for _ in range(n):
            y = fn()
        torch.cuda.synchronize()
        dtm = (time.perf_counter() - t0) / n
        print(f"{name}: {dtm*1e3:.1f} ms for B={B} ({B*10/dtm:.1f} audio-s/s) out {tuple(y.shape)} {y.dtype}")

if "--layers" in sys.argv:
    # per-layer table of ONE decode pass: HIP events around every conv launch (launch stream), algorithmic bytes/FLOPs
    import json
    from kalle_audio_amd import conv_ops
    rec = []
    def wrap(fn, kind):
        def inner(x, w, bias, **kw):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); ret = fn(x, w, bias, **kw); e1.record()
            y = ret[0] if isinstance(ret, tuple) else ret            # want_raw: (activated, raw) - two stores
            es = x.element_size()
            nout = 2 if isinstance(ret, tuple) else 1
            byts = (x.numel() + nout * y.numel() + (y.numel() if kw.get("residual") is not None else 0)) * es + w.numel() * 4
            fl = 2.0 * w.numel() * x.shape[0] * (y.shape[2] if kind == "conv" else x.shape[2])
            rec.append(dict(kind=kind, Cin=x.shape[1], Cout=y.shape[1], K=kw["K"], stride=kw.get("stride", 1),
                            dil=kw.get("dilation", 1), Lout=y.shape[2], bytes=byts, flops=fl, ev=(e0, e1)))
            return ret
        return inner
    conv_ops.conv1d, conv_ops.conv_transpose1d = wrap(conv_ops.conv1d, "conv"), wrap(conv_ops.conv_transpose1d, "convT")
    which = "encode" if "--encode" in sys.argv else "decode"
    with torch.no_grad():
        ae.encode(wav) if which == "encode" else ae.decode(z)
    torch.cuda.synchronize()
    out = []
    for r in rec:
        ms = r["ev"][0].elapsed_time(r["ev"][1])
        out.append({k: v for k, v in r.items() if k != "ev"} | {"us": ms * 1e3, "GBps": r["bytes"] / ms / 1e6,
                                                                "TFLOPs": r["flops"] / ms / 1e9})
    json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gpurun_out",
                                     f"vae_layers_{which}.json" if which == "encode" else "vae_layers.json"), "w"))
    tot_t = sum(o["us"] for o in out); tot_f = sum(o["flops"] for o in out); tot_b = sum(o["bytes"] for o in out)
    if "--md" in sys.argv:
        md = sys.argv[sys.argv.index("--md") + 1]
        with open(md, "a") as f:
            f.write(f"\n## {which}, B={B} ({B * 10} s of stereo 44.1 kHz audio), {'bf16' if half else 'fp32'} activations\n\n"
                    f"{len(out)} conv launches, {tot_t/1e3:.1f} ms ({B*10/(tot_t/1e6):.0f} audio-s/s), {tot_f/tot_t/1e6:.1f} TFLOP/s fp32 "
                    f"vector (peak 157.3), {tot_b/tot_t/1e3:.0f} GB/s algorithmic HBM (peak ~8000)\n\n"
                    "| kernel | Cin -> Cout | k | stride | dil | Lout | us | fp32 TFLOP/s | algorithmic GB/s |\n|---|---|---|---|---|---|---|---|---|\n")
            for o in out:
                f.write(f"| {o['kind']} | {o['Cin']} -> {o['Cout']} | {o['K']} | {o['stride']} | {o['dil']} | {o['Lout']} | {o['us']:.0f} | "
                        f"{o['TFLOPs']:.1f} | {o['GBps']:.0f} |\n")
    print(f"{which} B={B}: {len(out)} conv launches, {tot_t/1e3:.1f} ms, {tot_f/tot_t/1e6:.1f} TFLOP/s, {tot_b/tot_t/1e3:.0f} GB/s algorithmic")
