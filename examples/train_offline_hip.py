#!/usr/bin/env python3
"""The reference's `train_offline.py` loop (lines 47-300) on the MI355X path: same experiment YAML, same checkpoint naming
(`output/epoch_{e}_step_{s}.pt` = model.state_dict()), same loss weighting, AdamW + cosine-with-warmup - with
`accelerate` / DDP / torch AdamW replaced by `kalle_audio_amd.engine.DataParallelTrainer` (flat buckets, per-layer RCCL
all-reduce overlapped with backward, fused AdamW) and `model_sigmaVAE.Llasa` by its HIP drop-in.

  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 examples/train_offline_hip.py \
      --config configs/twj_0828.yaml [--steps K] [--synthetic B L]

The reference's dataset (`twj_dataset_offline.TTSDataset_online_parquet`, parquet shards under /mnt/...) is outside this
build's scope: pass `--dataset-module twj_dataset_offline` with the reference checkout on PYTHONPATH to use it unchanged, or
`--synthetic B L` for collate()-shaped random batches (text prefix, audio frames, right padding)."""
import argparse
import datetime
import importlib
import os
import shutil
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import kalle_audio_amd  # noqa: E402
from kalle_audio_amd import config as kcfg, engine  # noqa: E402


class _LenTokenizer:
    """stand-in when no tokenizer directory is available: Llasa only needs len(tokenizer)"""

    def __init__(self, n):
        self.n = n

    def __len__(self):
        return self.n


def synthetic_batches(B, L, latent_dim, vocab, device, seed):
    g = torch.Generator(device=device).manual_seed(seed)
    while True:
        nt = torch.randint(8, L // 4, (B,), generator=g, device=device)
        na = torch.randint(L // 2, L - L // 4, (B,), generator=g, device=device)
        pos = torch.arange(L, device=device)[None]
        ids_mask = (pos < nt[:, None]).float()
        audio_mask = ((pos >= nt[:, None]) & (pos < (nt + na)[:, None])).float()
        yield {"input_ids": torch.randint(0, vocab, (B, L), generator=g, device=device),
               "audio_latents": torch.randn(B, L, latent_dim, generator=g, device=device),
               "distribute_lables": torch.randn(B, L, latent_dim, generator=g, device=device),
               "text_ids_mask": ids_mask, "audio_latents_mask": audio_mask,
               "distribute_lables_mask": ((pos >= (nt - 1)[:, None]) & (pos < (nt + na - 1)[:, None])).float(),
               "enddist_mask": (pos == (nt + na - 1)[:, None]).float()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", required=True)
    ap.add_argument("--steps", type=int, default=None, help="stop after this many optimizer steps (default: total_steps)")
    ap.add_argument("--synthetic", type=int, nargs=2, metavar=("B", "L"), default=None)
    ap.add_argument("--dataset-module", default=None)
    args = ap.parse_args()

    rank, world, local = engine.init_distributed()
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)
    config = kcfg.load_experiment_config(args.config)
    if rank == 0:
        for k in ("exp_dir", "log_dir", "output_dir", "resume_dir"):
            os.makedirs(config[k], exist_ok=True)
        shutil.copyfile(args.config, os.path.join(config["exp_dir"], "config.yaml"))
    kalle_audio_amd.install()                              # `from model_sigmaVAE import Llasa` -> the HIP drop-in
    from model_sigmaVAE import Llasa

    tok_path = config.get("tokenizer_path")
    if tok_path and os.path.isdir(tok_path):
        from transformers import AutoTokenizer
        tokenizer = AutoTokenizer.from_pretrained(tok_path)
    else:
        tokenizer = _LenTokenizer(int(config.get("tokenizer_len", 128264)))
    with torch.device(device):
        model = Llasa(config["model"], tokenizer, use_flash_attention=config.get("use_flash_attation", True))
    model.to(device)

    epoch, step = 0, 0
    last = kcfg.latest_checkpoint(config["output_dir"])
    ckpt = last[0] if last else config.get("start_checkpoint")
    if last:
        _, epoch, step = last
    if ckpt:
        model.load_state_dict(torch.load(ckpt, map_location="cpu"))
        print(f"resumed from {ckpt} (epoch {epoch}, step {step})")

    total = int(config.get("total_steps", 10 ** 9))
    warm = int(config.get("warmup_steps", 0))
    step0 = step       # resume offset, bound once: the trainer passes the number of COMPLETED optimizer steps (0, 1, ...) as `s` (LambdaLR's convention)
    trainer = engine.DataParallelTrainer(
        model, lr=config["lr"], optimizer="AdamW", weight_decay=config["weight_decay"],
        grad_accum_steps=config["gradient_accumulation_steps"],
        lr_schedule=lambda s: engine.cosine_with_warmup(s + step0, warm, total))

    if args.synthetic:
        B, L = args.synthetic
        batches = synthetic_batches(B, L, config["model"]["latent_dim"], len(tokenizer), device, 1234 + rank)
    elif args.dataset_module:
        ds_mod = importlib.import_module(args.dataset_module)      # the reference's dataset + collate, unchanged
        ds = ds_mod.TTSDataset_online_parquet(config["dataset"], tokenizer, [config["dataset"]["meta_path"]], device,
                                              output_bf16=False)
        batches = iter(torch.utils.data.DataLoader(ds, batch_size=None, collate_fn=getattr(ds_mod, "collate", None)))
    else:
        raise SystemExit("pass --synthetic B L or --dataset-module NAME")

    stop = step + args.steps if args.steps else total
    wa, we = config["audio_loss_weight"], config["end_loss_weight"]
    micro = 0
    while step < stop:
        batch = {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in next(batches).items()}
        out = model(input_ids=batch["input_ids"], audio_latents=batch["audio_latents"],
                    audio_distribution_l=batch["distribute_lables"], ids_mask=batch["text_ids_mask"],
                    audio_mask=batch["audio_latents_mask"], target_mask=batch["distribute_lables_mask"],
                    end_mask=batch["enddist_mask"])
        trainer.backward(out["audio_loss"] * wa + out["end_loss"] * we)    # bwd + all-reduce + (on the boundary) AdamW
        micro += 1
        if micro % config["gradient_accumulation_steps"]:
            continue
        step += 1
        if step % config["log_interval"] == 0 and rank == 0:               # the only host syncs: the logged scalars
            print(f"[{datetime.datetime.now():%H:%M:%S}] epoch {epoch} step {step} lr "
                  f"{trainer.last_lr:.3e} "
                  f"audio_loss {out['audio_loss'].item():.4f} end_loss {out['end_loss'].item():.4f}", flush=True)
        if step % int(config.get("save_interval", 10 ** 9)) == 0 and rank == 0:
            torch.save(model.state_dict(), os.path.join(config["output_dir"], f"epoch_{epoch}_step_{step}.pt"))
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
