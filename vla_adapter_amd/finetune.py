"""Native counterpart of vla-scripts/finetune.py for the accelerated path (adapter-only fine-tune).

Keeps the reference's flat ``FinetuneConfig`` (finetune.py:66-128) and its ``--flag value`` command line (draccus is
absent here: parsed with argparse from the dataclass fields), the per-step metric names (finetune.py:421-444), the LR
warm-up / MultiStepLR schedule (:903-921, 1061-1065) and the checkpoint file names (:527-572).  What is NOT here, on
purpose: HF-hub / network loaders (:752-754), the RLDS/TensorFlow input pipeline (out of scope) -
batches come from ``synthetic.make_batch`` (same collator contract) unless a ``--batch_file`` (.pt dict) is given,
weights are random-init unless ``--vlm_path`` / ``--resum_vla_path`` point at local state-dict files - LoRA and
full-unfreeze (config 4/5) raise NotImplementedError.
"""
from __future__ import annotations

import argparse
import dataclasses
import json
import os
import time
from dataclasses import dataclass
from pathlib import Path
from typing import Optional

import torch


@dataclass
class FinetuneConfig:
    # fmt: off
    config_file_path: str = "openvla/openvla-7b"
    vlm_path: str = "openvla/openvla-7b"
    use_minivlm: bool = False
    resum_vla_path: str = "openvla/openvla-7b"
    # Dataset
    data_root_dir: Path = Path("datasets/rlds")
    dataset_name: str = "aloha_scoop_x_into_bowl"
    run_root_dir: Path = Path("runs")
    shuffle_buffer_size: int = 100_000
    # Algorithm and architecture
    use_l1_regression: bool = True
    use_diffusion: bool = False
    num_diffusion_steps: int = 50
    use_film: bool = False
    num_images_in_input: int = 1
    use_proprio: bool = False
    phase1_path: str = "None"
    # Training configuration
    batch_size: int = 8
    learning_rate: float = 5e-4
    lr_warmup_steps: float = 0.1
    num_steps_before_decay: int = 100000
    grad_accumulation_steps: int = 1
    max_steps: int = 200000
    use_val_set: bool = False
    val_freq: int = 10_000
    val_time_limit: int = 180
    save_freq: int = 10_000
    save_latest_checkpoint_only: bool = False
    resume: bool = False
    resume_step: Optional[int] = None
    image_aug: bool = True
    diffusion_sample_freq: int = 50
    # LoRA
    use_lora: bool = False
    lora_rank: int = 32
    lora_dropout: float = 0.0
    merge_lora_during_training: bool = False
    # Full Finetune
    use_fz: bool = False
    # Logging
    wandb_entity: str = "your-wandb-entity"
    wandb_project: str = "your-wandb-project"
    run_id_note: Optional[str] = None
    run_id_override: Optional[str] = None
    wandb_log_freq: int = 10
    # revision version
    use_pro_version: bool = True
    phase: str = "Training"
    # native additions (not in the reference)
    tiny: bool = False                    # prismatic-tiny plumbing config (BASELINE configs[0])
    seed: int = 0
    batch_file: Optional[str] = None      # torch-saved dict with the collator's keys
    use_graph: bool = True                # replay the captured hipGraphs
    # fmt: on


def parse_args(argv=None) -> FinetuneConfig:
    ap = argparse.ArgumentParser(description="native VLA-Adapter fine-tune (same flags as the reference's FinetuneConfig)")
    for f in dataclasses.fields(FinetuneConfig):
        default = f.default
        t = type(default) if default is not None else str
        if t is bool:
            ap.add_argument(f"--{f.name}", type=lambda s: str(s).lower() in ("1", "true", "yes"), default=default)
        elif t is type(Path(".")):
            ap.add_argument(f"--{f.name}", type=Path, default=default)
        else:
            ap.add_argument(f"--{f.name}", type=(int if f.name in ("resume_step",) else t), default=default)
    return FinetuneConfig(**vars(ap.parse_args(argv)))


def lr_at(step: int, cfg: FinetuneConfig) -> float:
    """finetune.py:1061-1065 warm-up (10 % -> 100 % over lr_warmup_steps) on top of MultiStepLR(gamma 0.1) (:917)."""
    lr = cfg.learning_rate * (0.1 if step >= cfg.num_steps_before_decay else 1.0)
    return lr * (0.1 + 0.9 * min((step + 1) / cfg.lr_warmup_steps, 1.0))


def save_training_checkpoint(cfg: FinetuneConfig, run_dir: Path, step: int, eng) -> None:
    """File names / key layout of finetune.py:527-572 (rank 0)."""
    suffix = "latest_checkpoint.pt" if cfg.save_latest_checkpoint_only else f"{step}_checkpoint.pt"
    d = run_dir if cfg.save_latest_checkpoint_only else Path(str(run_dir) + f"--{step}_chkpt")
    os.makedirs(d, exist_ok=True)
    torch.save({k: v.cpu() for k, v in eng.head.head_state_dict().items()}, d / f"action_head--{suffix}")
    torch.save({k: v.clone().cpu() for k, v in eng.head.proprio_views().items()}, d / f"proprio_projector--{suffix}")
    torch.save({"weight": eng.head.P.view("action_queries").clone().cpu()}, d / f"action_queries--{suffix}")
    json.dump({}, open(d / "dataset_statistics.json", "w"))


def finetune(cfg: FinetuneConfig) -> dict:
    from . import ddp, engine as E, synthetic as S
    if cfg.use_lora or cfg.use_film or cfg.use_diffusion or not cfg.use_l1_regression:
        raise NotImplementedError("native path: adapter-only L1-regression fine-tune (LoRA / FiLM / diffusion not accelerated yet)")
    rank, local, world = ddp.init_process_group_from_env()
    torch.cuda.set_device(local)
    dev = f"cuda:{local}"
    mcfg = E.tiny_config() if cfg.tiny else E.config2()
    mcfg.n_img = cfg.num_images_in_input
    mcfg.pro = bool(cfg.use_pro_version)
    W = S.make_weights(mcfg, dev, seed=cfg.seed)                  # identical on all ranks == DDP's initial broadcast
    if cfg.vlm_path and os.path.isfile(cfg.vlm_path):             # local VLM state dict (HF-style or native Prismatic keys)
        from . import checkpoints as CK
        W.update(CK.split_reference_state_dict(CK.load_file(cfg.vlm_path), mcfg))
    if cfg.resume and cfg.resum_vla_path and os.path.isdir(cfg.resum_vla_path):   # head / proprio only (finetune.py:275-278)
        from . import checkpoints as CK
        W["head"], W["proprio"] = CK.load_run_dir(cfg.resum_vla_path, cfg.resume_step)
    eng = E.VLAEngine(mcfg, W, dev)
    if world > 1:
        eng.reducer = ddp.FlatGradReducer()
    if cfg.batch_file:
        batch = {k: v.to(dev) for k, v in torch.load(cfg.batch_file, weights_only=True).items()}
    else:
        batch = S.make_batch(mcfg, cfg.batch_size, dev, seed=1000 * cfg.seed + rank, P=32, ragged=True)
    D = mcfg.llm.d
    gen = torch.Generator(device=dev).manual_seed(cfg.seed * 7919 + rank)
    noise = torch.zeros(mcfg.chunk, mcfg.action_dim * D, device=dev, dtype=torch.bfloat16)
    run_dir = Path(cfg.run_root_dir) / (cfg.run_id_override or f"native+{cfg.dataset_name}+b{cfg.batch_size * world}+lr-{cfg.learning_rate}")
    if cfg.use_graph:
        eng.capture(batch, noise if cfg.phase == "Training" else None)
    log, t0 = [], time.time()
    for step in range(cfg.max_steps):
        if cfg.phase == "Training":   # fresh N(0, 0.02^2) perturbation every call (action_heads.py:14-17, 69-72)
            noise.copy_((torch.randn(noise.shape, device=dev, generator=gen) * 0.02).to(torch.bfloat16))
        lr = lr_at(step, cfg)
        if cfg.use_graph:
            loss3 = eng.train_step_graphed(lr)
        else:
            loss3 = eng.train_step(batch, lr, noise if cfg.phase == "Training" else None)
        if step % cfg.wandb_log_freq == 0 or step == cfg.max_steps - 1:       # the only host sync, every log_freq steps
            l = loss3.tolist()
            log.append(dict(step=step, loss_value=l[0], curr_action_l1_loss=l[1], next_actions_l1_loss=l[2], lr=lr))
            if rank == 0:
                print(json.dumps(log[-1]), flush=True)
        if step > 0 and step % cfg.save_freq == 0:
            eng.flush()                      # the graphed step leaves its parameter update pending (engine.capture)
            if rank == 0:
                save_training_checkpoint(cfg, run_dir, step, eng)
    eng.flush()
    torch.cuda.synchronize()
    return dict(log=log, seconds=time.time() - t0, steps=cfg.max_steps, world=world)
