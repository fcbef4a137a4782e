#!/usr/bin/env python3
"""bench.py - DiT train-step throughput on MI355X (BASELINE.json metric: audio-seconds/sec).

A "step" = one full data-parallel train step of the stable_audio_tools DiT objective on one batch of synthetic
10 s @ 12.5 Hz x 1024 latents: noise/target -> DiT forward (24 blocks, D=1536) -> MSE -> backward -> gradient
all-reduce (RCCL, overlapped) -> fused Adam.  Inputs are resident in HBM before the timed region.

  python bench.py --gpus 1 --steps 5 --warmup 2
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
      bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel (by summed time) among the GEMM instantiations,
timed with HIP events on the launch stream during the timed steps; `cpu_baseline` times the CPU oracle (a port of
the reference algorithm, oracle/kalle_oracle.py) on the host cores for a bounded sample (rank 0, N=1 only).
"""
import argparse
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# Benchmark DiT shape - chosen by this build (the reference defines no DiT hyper-parameters):
CFG = dict(io_channels=1024, embed_dim=1536, depth=24, num_heads=24, cond_token_dim=768, project_cond_tokens=False,
           global_cond_dim=1536, transformer_type="continuous_transformer", global_cond_type="prepend")
T_FRAMES, S_CTX, CLIP_SECONDS = 125, 130, 10.0
PEAK_BF16_TFLOPS = 2500.0   # dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md


def flops_per_clip_fwd(cfg, T=T_FRAMES, S=S_CTX):
    """algorithmic forward FLOPs per clip: 24 x (36 N D^2 + 4 S Dc^2 + 4 N^2 D + 4 N S D) + 4 N C D"""
    D, Dc, C, L = cfg["embed_dim"], cfg["cond_token_dim"], cfg["io_channels"], cfg["depth"]
    ada = cfg.get("global_cond_type") == "adaLN"
    N = T if ada else T + 1                     # prepend: one conditioning token joins the sequence
    per_block = 36 * N * D * D + 4 * S * Dc * Dc + 4 * N * N * D + 4 * N * S * D + (12 * D * D if ada else 0)
    return L * per_block + 4 * N * C * D


def build_model(device, seed=1234, cfg=CFG):
    from kalle_audio_amd.stable_audio_tools.models.diffusion import ConditionedDiffusionModelWrapper, DiTWrapper
    torch.manual_seed(seed)
    with torch.device(device):
        dit = DiTWrapper(**cfg)
    # reference init leaves several output projections at zero (transformer.py:255-258,300-301; dit.py:131-133);
    # re-draw them N(0, 0.02^2) then x0.5 like DiTWrapper does (models/diffusion.py:505-507) - BASELINE.md section 2
    g = torch.Generator(device=device).manual_seed(seed + 1)
    with torch.no_grad():
        for p in dit.parameters():
            if p.abs().max() == 0:
                p.copy_(torch.randn(p.shape, generator=g, device=device) * 0.02 * 0.5)
    return ConditionedDiffusionModelWrapper(dit, None, io_channels=cfg["io_channels"], sample_rate=44100,
                                            min_input_length=1, diffusion_objective="v",
                                            cross_attn_cond_ids=["prompt"], global_cond_ids=["global"])


def make_batch(B, device, seed, cfg=CFG, T=None):
    T = T_FRAMES if T is None else T
    g = torch.Generator(device=device).manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g, device=device)
    lat = r(B, cfg["io_channels"], T)
    noise = r(B, cfg["io_channels"], T)
    t = torch.rand(B, generator=g, device=device)
    cond = {"prompt": (r(B, S_CTX, cfg["cond_token_dim"]), torch.ones(B, S_CTX, dtype=torch.bool, device=device)),
            "global": (r(B, cfg["global_cond_dim"]), None)}
    return lat, noise, t, cond


def cpu_baseline(sample_clips=2, depth=None):
    """time the CPU oracle (port of the reference algorithm) for one full train step on `sample_clips` clips"""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import kalle_oracle as ko
    # the 1-GPU box's CPU share is 16 cores (of the host's os.cpu_count()); more threads only oversubscribe
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, int(os.environ.get("KALLE_CPU_THREADS", "16"))))
    torch.set_num_threads(cores)
    cfg = dict(CFG)
    if depth is not None:
        cfg["depth"] = depth
    shapes = ko.dit_shapes(cfg["io_channels"], cfg["embed_dim"], cfg["depth"], cond_token_dim=cfg["cond_token_dim"],
                           global_cond_dim=cfg["global_cond_dim"], project_cond_tokens=False)
    g = torch.Generator().manual_seed(7)
    sd = {}
    for n, s in shapes:
        fan = s[1] if len(s) > 1 else 1
        sd[n] = (torch.randn(s, generator=g) * (0.5 / math.sqrt(fan))).requires_grad_(True)
    lat, noise, t, cond = make_batch(sample_clips, "cpu", 99, cfg)
    opt = torch.optim.Adam(list(sd.values()), lr=1e-5)
    ocfg = dict(embed_dim=cfg["embed_dim"], depth=cfg["depth"], num_heads=cfg["num_heads"], global_cond_type="prepend")

    def step():
        opt.zero_grad(set_to_none=True)
        loss, *_ = ko.train_step_loss(sd, ocfg, lat, noise, t, "v", cross_attn_cond=cond["prompt"][0],
                                      global_embed=cond["global"][0])
        loss.backward()
        opt.step()
        return loss.item()

    step()                                  # warm-up (allocations, thread pool)
    t0 = time.perf_counter()
    step()
    dt = time.perf_counter() - t0
    scale = CFG["depth"] / cfg["depth"]
    return {"value": sample_clips * CLIP_SECONDS / (dt * scale), "unit": "audio-seconds/sec", "cores": cores,
            "kind": "port",
            "sample": f"{sample_clips} clips, 1 full train step (fwd+bwd+Adam, fp32) of the {cfg['depth']}-block DiT"
                      + ("" if depth is None else f" scaled x{scale:g} to 24 blocks") + f", {dt:.1f} s on {cores} threads"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=256, help="clips per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timer", action="store_true")
    ap.add_argument("--time-all-launches", action="store_true", help="events around EVERY GEMM launch of the timed region (the bench "
                    "of rounds 1-3; kept to measure what those events cost)")
    ap.add_argument("--cpu-depth", type=int, default=None, help="time a shallower oracle and scale (debug)")
    # sweep axes of SURVEY.md 8(d) - the defaults are the headline configuration
    ap.add_argument("--io-channels", type=int, default=CFG["io_channels"], help="latent channels: 64 | 512 | 1024")
    ap.add_argument("--global-cond-type", default=CFG["global_cond_type"], choices=["prepend", "adaLN"])
    ap.add_argument("--objective", default="v", choices=["v", "rectified_flow"])
    args = ap.parse_args()
    CFG["io_channels"], CFG["global_cond_type"] = args.io_channels, args.global_cond_type

    import torch.distributed as dist
    from kalle_audio_amd import engine, ops
    rank, world, local = engine.init_distributed()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X GPU (the product path has no CPU fallback)")
    local = local % torch.cuda.device_count()       # (one rank per GPU; a gloo rehearsal may put several ranks on one device)
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)

    model = build_model(device)
    trainer = engine.DataParallelTrainer(model, lr=1e-5, optimizer="Adam")
    lat, noise, t, cond = make_batch(args.batch, device, 1234 + rank)

    def step():
        return trainer.train_step(model, lat, t, noise, cond, objective=args.objective)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    # HIP events around kernel launches cost time themselves (two records per launch, ~300 GEMMs per step: same box 222.3 -> 224.9 ms
    # at B = 256, 29.8 -> 31.1 at B = 16), so the timed region brackets only the DOMINANT kernel's launches - the figure `roofline.achieved` is built from -
    # and one extra step after it (not timed, not counted) collects the table of every GEMM variant and confirms which one dominates
    DOMINANT = "gemm3_wgrad_group_kernel"
    timer = None if (args.no_kernel_timer or rank != 0) else []
    ops.KERNEL_TIMER, ops.KERNEL_TIMER_ONLY = timer, (None if args.time_all_launches else DOMINANT)
    trainer.comm_timing = []          # events around the all-reduce waits of the timed steps (exposed communication)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    fence()
    dt = time.perf_counter() - t0
    ops.KERNEL_TIMER = ops.KERNEL_TIMER_ONLY = None
    if world > 1:
        tt = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = tt.item()
    loss_v = loss.item()
    comm = trainer.comm_summary()     # RCCL rank count, buckets per step, exposed (non-overlapped) all-reduce time
    trainer.comm_timing = None
    full = None
    if not args.no_kernel_timer:      # the extra step: every rank runs it (collectives), rank 0 brackets every GEMM launch
        full = [] if rank == 0 else None
        ops.KERNEL_TIMER = full
        step()
        fence()
        ops.KERNEL_TIMER = None

    if rank == 0:
        clips = args.batch * world * args.steps
        value = clips * CLIP_SECONDS / dt
        fwd = flops_per_clip_fwd(CFG)
        out = {
            "metric": "audio-seconds/sec (node) DiT train step, 10s@12.5Hz x1024 latents",
            "value": value, "unit": "audio-seconds/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "DiT train step (fwd+bwd+allreduce+Adam), 24 blocks D=1536 h=24, "
                                   "10 s @ 12.5 Hz x 1024 latents (T=125+1 prepend), S=130 x 768 cross-attn cond; "
                                   "DiT shape chosen by this build (reference defines none)",
                       "global_batch": args.batch * world, "per_gpu_batch": args.batch, "parallelism": f"dp{world}",
                       "objective": args.objective, "io_channels": CFG["io_channels"],
                       "global_cond_type": CFG["global_cond_type"], "loss": loss_v},
            "algorithmic_tflops_per_gpu": 3 * fwd * args.batch * args.steps / dt / 1e12,
            "mfma_roofline_frac_step": 3 * fwd * args.batch * args.steps / dt / 1e12 / PEAK_BF16_TFLOPS,
            "comm": comm,
        }
        if full:
            agg = {}
            shapes = {}
            for variant, fl, e0, e1, ab, shp in full:
                sa = shapes.setdefault((variant, shp), [0.0, 0.0, 0])
                sa[0] += fl
                sa[1] += e0.elapsed_time(e1) * 1e-3
                sa[2] += 1
                a = agg.setdefault(variant, [0.0, 0.0, 0, 0.0])
                a[0] += fl
                a[1] += e0.elapsed_time(e1) * 1e-3
                a[2] += 1
                a[3] += ab
            if os.environ.get("KALLE_BENCH_SHAPES"):        # per-shape table on stderr (not part of the JSON line)
                for (variant, shp), (fl, sec, n) in sorted(shapes.items(), key=lambda kv: -kv[1][1]):
                    print(f"[shape] {variant:34s} M,N,K={shp}  {n:3d}/step  {sec / n * 1e6:8.1f} us  "
                          f"{fl / sec / 1e12:7.0f} TFLOP/s  {100 * sec / (dt / args.steps):5.2f} % of step", file=sys.stderr)
            dom = max(agg.items(), key=lambda kv: kv[1][1])
            name, (fl, sec, n, ab) = dom
            timed_in = "one extra step after the timed region"
            if name == DOMINANT and timer:      # the usual case: its launches were bracketed inside the timed region
                timer = [r for r in timer if r[0] == DOMINANT]
                fl = sum(r[1] for r in timer)
                sec = sum(r[2].elapsed_time(r[3]) for r in timer) * 1e-3
                n, ab = len(timer), sum(r[4] for r in timer)
                timed_in = "the timed region"
            ach = fl / sec / 1e12
            # HBM-side bytes per launch of this kernel from the committed PMC passes (profiles/traffic_rNN.json, newest round;
            # tools/pmc_traffic_summary.py) - counters cannot be read inside this process, so the line says where the number
            # comes from: the kernel-source hash of the profiled build, the commit, and whether the running tree still matches
            traffic = traffic_source = None
            import glob
            tjs = sorted(glob.glob(os.path.join(ROOT, "profiles", "traffic_r[0-9][0-9].json")))
            if tjs:
                tjd = json.load(open(tjs[-1]))
                traffic = tjd.get(name)
                from kalle_audio_amd.build import source_stamp
                src = tjd.get("_source") or {}
                traffic_source = {"file": os.path.relpath(tjs[-1], ROOT), "csrc_sha16": src.get("csrc_sha16"),
                                  "commit": src.get("commit_at_publish"),
                                  "matches_running_code": src.get("csrc_sha16") == source_stamp()}
            out["roofline"] = {"bound": "mfma", "kernel": name, "achieved": ach, "peak": PEAK_BF16_TFLOPS,
                               "unit": "TFLOP/s", "frac": ach / PEAK_BF16_TFLOPS, "traffic": traffic,
                               "traffic_source": traffic_source, "hip_events_in": timed_in,
                               "launches": n, "avg_launch_us": sec / n * 1e6, "avg_flop_per_launch": fl / n,
                               "algorithmic_bytes_per_launch": ab / n,
                               "all_gemm_variants": {k: {"tflops": v[0] / v[1] / 1e12, "avg_us": v[1] / v[2] * 1e6,
                                                         "launches": v[2], "time_share_of_step":
                                                             v[1] / (dt / args.steps)} for k, v in agg.items()},
                               "all_gemm_variants_from": "one extra step after the timed region (events around every launch)"}
        if world == 1 and not args.no_cpu_baseline:
            del trainer, model
            torch.cuda.empty_cache()
            out["cpu_baseline"] = cpu_baseline(depth=args.cpu_depth)
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        if world > 1:
            dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
