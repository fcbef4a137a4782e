"""Config readers that accept the reference's files unchanged:
  * the experiment YAML read by train_offline.py:47-57 (`yaml.safe_load`, then exp/log/output/resume dirs derived),
  * the `accelerate launch` YAML (default_config.yaml / default_config_cpu.yaml), of which only
    distributed_type / num_processes / mixed_precision matter to this build's launcher.
Values the reference casts on use (lr, weight_decay: train_offline.py:96-97) are returned as floats."""
import os

import yaml

REQUIRED = ("project_name", "exp_dir", "model", "lr", "weight_decay", "gradient_accumulation_steps")


def load_experiment_config(path):
    with open(path) as f:
        cfg = yaml.safe_load(f)
    missing = [k for k in REQUIRED if k not in cfg]
    if missing:
        raise KeyError(f"experiment config {path} lacks {missing}")
    cfg["lr"] = float(cfg["lr"])
    cfg["weight_decay"] = float(cfg["weight_decay"])
    for k in ("warmup_steps", "total_steps", "save_interval", "gradient_accumulation_steps"):
        if k in cfg:
            cfg[k] = int(cfg[k])
    exp = os.path.join(cfg["exp_dir"], cfg["project_name"])          # train_offline.py:49-52
    cfg["exp_dir"] = exp
    cfg["log_dir"] = os.path.join(exp, "logs")
    cfg["output_dir"] = os.path.join(exp, "output")
    cfg["resume_dir"] = os.path.join(exp, "resume")
    cfg.setdefault("log_interval", 1)                                # read unconditionally at train_offline.py:266
    cfg.setdefault("audio_loss_weight", 1.0)
    cfg.setdefault("end_loss_weight", 1.0)
    return cfg


def load_accelerate_config(path):
    with open(path) as f:
        acc = yaml.safe_load(f)
    acc.setdefault("distributed_type", "NO")
    acc["num_processes"] = int(acc.get("num_processes", 1))
    acc.setdefault("mixed_precision", "no")
    return acc


def latest_checkpoint(output_dir):
    """train_offline.py:117-125: newest epoch_{e}_step_{s}.pt by mtime -> (path, epoch, step) or None."""
    if not os.path.isdir(output_dir):
        return None
    cks = [os.path.join(output_dir, d) for d in os.listdir(output_dir) if d.startswith("epoch_")]
    if not cks:
        return None
    last = max(cks, key=os.path.getmtime)
    parts = os.path.basename(last).split("_")
    return last, int(parts[1]), int(parts[3].split(".")[0])
