"""Native counterpart of vla-scripts/finetune.py for the accelerated path (adapter-only fine-tune).

Keeps the reference's flat ``FinetuneConfig`` (finetune.py:66-128) and its ``--flag value`` command line (draccus is
absent here: parsed with argparse from the dataclass fields), the per-step metric names (finetune.py:421-444), the LR
warm-up / MultiStepLR schedule (:903-921, 1061-1065) and the checkpoint file names (:527-572).  What is NOT here, on
purpose: HF-hub / network loaders (:752-754), the RLDS/TensorFlow input pipeline (out of scope) -
batches come from an iterable / ``--batch_file`` (a .pt dict or a directory of them: the collator's contract) or
``synthetic.make_batch``; weights are random-init unless ``--vlm_path`` / ``--resum_vla_path`` point at local state-dict
files.  Every reference flag is either honoured or refused with an error (``check_supported``); none is silently dropped.
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
    max_seq_len: int = 0                  # static token length every batch is right-padded to (0: length of the first batch)
    conservative_rows: bool = False       # captured live-row window starts at the first text row instead of the first batch's action block
    dataset_statistics_file: Optional[str] = None   # JSON written next to every checkpoint (finetune.py:531)
    objective: str = "l1"                 # "l1": action head + L1 regression (the reference's finetune.py); "token_ce": the native VLM / VLA trainer's
                                          # token cross-entropy (base_strategy.py:257-417) - LoRA / full modes only, no action head in the loss
    fp8_base_weights: bool = False        # --use_lora: the frozen base weights' products (forward and dX) on OCP e4m3 operands, the rank-r branch in
                                          # bf16 inside the same accumulator (BASELINE configs[4] "fp8 MFMA weight path"; parity unpinned: the reference is bf16)
    ddp_algo: str = "allreduce"           # data-parallel exchange per gradient bucket: "allreduce", or "rs_ag" = reduce-scatter + all-gather
                                          # (every rank talks to every peer directly: all 7 xGMI links instead of a ring)
    sync_check_freq: int = 0              # > 0: every that many optimizer steps the ranks compare a checksum of their parameters (one 8-byte
                                          # all-reduce) and the run stops on a mismatch; 0 = off
    backbone: Optional[str] = None        # model geometry: a name of engine.NAMED_CONFIGS ("config2", "dinosiglip-0_5b", "config5",
                                          # "tiny", "tiny_fused") - default: inferred from the --vlm_path state dict, else "config2"
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
    ns = ap.parse_args(argv)
    cfg = FinetuneConfig(**vars(ns))
    import sys
    given = [a[2:].split("=")[0] for a in (sys.argv[1:] if argv is None else argv) if str(a).startswith("--")]
    cfg._explicit = tuple(given)          # flags the user passed: out-of-path ones are refused, not ignored (check_supported)
    return cfg


# Flags of the reference that only configure subsystems outside the accelerated path (RLDS/TensorFlow input pipeline,
# W&B, HF hub): the native entry point consumes pre-collated batches, so their DEFAULT values are inert - but a value the
# user passes explicitly cannot be honoured and is refused instead of ignored.
OUT_OF_PATH_FLAGS = ("data_root_dir", "shuffle_buffer_size", "image_aug", "wandb_entity", "wandb_project", "run_id_note",
                     "config_file_path", "phase1_path", "num_diffusion_steps", "diffusion_sample_freq", "val_freq", "val_time_limit",
                     "use_minivlm")


def train_mode(cfg: FinetuneConfig) -> str:
    """Which parameters train.  Reference: use_lora -> peft adapters on every Linear of the VLM + action_queries + head
    (finetune.py:832-844); otherwise EVERY VLM parameter keeps requires_grad (:846-849) - ``use_fz`` only renames the run there
    (:183-184), freezing the VLM is its evident intent and what BASELINE configs[1] ("adapter-only") means: implemented."""
    return "lora" if cfg.use_lora else ("adapter" if cfg.use_fz else "full")


def check_supported(cfg: FinetuneConfig, explicit=()) -> None:
    """Raise for every reference option this path does not implement (nothing is parsed and silently dropped)."""
    if cfg.grad_accumulation_steps < 1:
        raise ValueError("grad_accumulation_steps must be >= 1")
    if cfg.use_lora and not 0.0 <= cfg.lora_dropout < 1.0:
        raise ValueError("--lora_dropout must lie in [0, 1)")
    if cfg.use_lora and cfg.lora_dropout > 0.0 and cfg.fp8_base_weights:
        raise NotImplementedError("--lora_dropout > 0 together with --fp8_base_weights: the dropped inputs are bf16 (use one or the other)")
    if cfg.fp8_base_weights and not cfg.use_lora:
        raise NotImplementedError("--fp8_base_weights needs --use_lora True (frozen base weights); the adapter-only forward has engine.enable_fp8_frozen()")
    if cfg.ddp_algo not in ("allreduce", "rs_ag"):
        raise ValueError("--ddp_algo is allreduce or rs_ag")
    if cfg.objective not in ("l1", "token_ce"):
        raise ValueError("--objective is l1 or token_ce")
    if cfg.objective == "token_ce" and train_mode(cfg) == "adapter":
        raise NotImplementedError("--objective token_ce trains the VLM (LoRA or full fine-tune); --use_fz True freezes it")
    if cfg.backbone is not None:
        from .engine import NAMED_CONFIGS
        if cfg.backbone not in NAMED_CONFIGS:
            raise ValueError(f"--backbone {cfg.backbone!r}: known geometries are {sorted(NAMED_CONFIGS)}")
    if cfg.use_film or cfg.use_diffusion or not cfg.use_l1_regression:
        raise NotImplementedError("native path = L1-regression action head; --use_film / --use_diffusion are not built")
    if cfg.use_val_set:
        raise NotImplementedError("--use_val_set: validation needs the RLDS val split (out of scope, SURVEY section 2 #16)")
    if not cfg.use_proprio:
        # the reference passes proprio_projector=None into predict_action, which calls it (action_heads.py:54): TypeError
        raise TypeError("use_proprio=False: 'NoneType' object is not callable (the reference's predict_action requires the "
                        "proprio projector, action_heads.py:53-55); pass --use_proprio True as every reference launch script does")
    if cfg.grad_accumulation_steps < 1:
        raise ValueError("grad_accumulation_steps must be >= 1")
    if cfg.resume and cfg.resume_step is None:
        raise ValueError("--resume needs --resume_step (finetune.py:1056 computes log_step = resume_step + gradient_step_idx)")
    bad = [n for n in explicit if n in OUT_OF_PATH_FLAGS]
    if bad:
        raise NotImplementedError(f"flags {bad} configure parts of the reference outside the accelerated path (input pipeline / "
                                  "logging services / hub loaders): the native entry point takes pre-collated batches "
                                  "(--batch_file) and cannot honour them")


def lr_at(gradient_step_idx: int, cfg: FinetuneConfig) -> float:
    """Learning rate the optimizer step of gradient step g runs with (finetune.py:917, 1061-1065, 1078-1082).
    lr_warmup_steps > 0 (the default 0.1): the warm-up block overwrites param_group['lr'] with
    original_lr * (0.1 + 0.9 * min((g + 1) / warmup, 1)) on EVERY iteration before optimizer.step(), which also undoes the
    MultiStepLR decay applied after the previous step - the decay never takes effect.  lr_warmup_steps <= 0: plain
    MultiStepLR, factor 0.1 from optimizer step num_steps_before_decay on."""
    if cfg.lr_warmup_steps > 0:
        return cfg.learning_rate * (0.1 + 0.9 * min((gradient_step_idx + 1) / cfg.lr_warmup_steps, 1.0))
    return cfg.learning_rate * (0.1 if gradient_step_idx >= cfg.num_steps_before_decay else 1.0)


def loop_plan(cfg: FinetuneConfig):
    """The reference loop's bookkeeping (finetune.py:1018-1122) as a generator of
    (batch_idx, gradient_step_idx, log_step, optimizer_step?, save?, last?) - one item per micro-batch.  The reference breaks
    behind the FIRST micro-batch whose log_step == max_steps (:1119-1121): with grad_accumulation_steps == 1 that batch's
    optimizer step has run (max_steps + 1 gradient steps from a fresh start); with ga > 1 it is the first micro-batch of gradient
    step max_steps - its forward / backward run, its gradients are never applied (boundary False).  A checkpoint is due when
    gradient_step_idx > 0 and log_step % save_freq == 0 (the reference re-saves on every micro-batch of such a step; here once,
    behind the optimizer step that completes it)."""
    ga = cfg.grad_accumulation_steps
    base = cfg.resume_step if cfg.resume else 0
    batch_idx = 0
    while True:
        g = batch_idx // ga
        log_step = base + g
        boundary = (batch_idx + 1) % ga == 0
        save = boundary and g > 0 and log_step % cfg.save_freq == 0
        last = log_step >= cfg.max_steps
        yield batch_idx, g, log_step, boundary, save, last
        if last:
            return
        batch_idx += 1


def save_training_checkpoint(cfg: FinetuneConfig, run_dir: Path, step: int, eng, dataset_statistics: Optional[dict] = None,
                             trainer=None) -> Path:
    """File names / key layout of finetune.py:527-572 (rank 0).  Where the VLM goes follows the reference: ``use_fz`` ->
    ``vla.module.save_pretrained(checkpoint_dir)`` (the whole VLM incl. the trained action queries at the top level, :551-552);
    otherwise ``save_pretrained(adapter_dir)`` - peft's adapter under ``lora_adapter/`` with LoRA, and, a quirk kept, the whole
    VLM under ``lora_adapter/`` for the full fine-tune (:553-554); the LoRA-merged VLM at the top level (:579-601).
    ``action_queries--{suffix}`` is a native addition (peft's adapter file does not hold them; the reference patches them into the
    merged model, :586-587), read back by checkpoints.load_run_dir."""
    suffix = "latest_checkpoint.pt" if cfg.save_latest_checkpoint_only else f"{step}_checkpoint.pt"
    d = run_dir if cfg.save_latest_checkpoint_only else Path(str(run_dir) + f"--{step}_chkpt")
    os.makedirs(d, exist_ok=True)
    torch.save({k: v.cpu() for k, v in eng.head.head_state_dict().items()}, d / f"action_head--{suffix}")
    torch.save({k: v.clone().cpu() for k, v in eng.head.proprio_views().items()}, d / f"proprio_projector--{suffix}")
    torch.save({"weight": eng.head.P.view("action_queries").clone().cpu()}, d / f"action_queries--{suffix}")
    from . import checkpoints as CK
    from safetensors.torch import save_file
    if cfg.use_lora and trainer is not None:    # peft's adapter directory (finetune.py:537-541: vla.module.save_pretrained(adapter_dir))
        ad = d / "lora_adapter"
        os.makedirs(ad, exist_ok=True)
        save_file({k: v.contiguous().cpu() for k, v in trainer.lora_state_dict().items()}, str(ad / "adapter_model.safetensors"))
        json.dump(dict(peft_type="LORA", r=cfg.lora_rank, lora_alpha=2 * cfg.lora_rank, lora_dropout=cfg.lora_dropout, target_modules="all-linear",
                       init_lora_weights="gaussian"), open(ad / "adapter_config.json", "w"), indent=2)
        if cfg.merge_lora_during_training:      # finetune.py:579-601: merge the adapter into a bf16 base and save the whole VLM
            merged = trainer.merged_weights()
            saved = {}
            for key, wm in merged.items():
                holder, wk = trainer._base(key)
                saved[key] = holder[wk].clone()
                holder[wk].copy_(wm)
            try:
                save_file({k: v.contiguous().cpu() for k, v in CK.engine_vlm_state_dict(eng).items()}, str(d / "model.safetensors"))
            finally:
                for key, w0 in saved.items():
                    holder, wk = trainer._base(key)
                    holder[wk].copy_(w0)
    elif trainer is not None:                   # full fine-tune: vla.module.save_pretrained(adapter_dir) (finetune.py:553-554)
        ad = d / "lora_adapter"
        os.makedirs(ad, exist_ok=True)
        save_file({k: v.contiguous().cpu() for k, v in CK.engine_vlm_state_dict(eng).items()}, str(ad / "model.safetensors"))
    else:                                       # adapter-only (use_fz): vla.module.save_pretrained(checkpoint_dir) (:551-552)
        save_file({k: v.contiguous().cpu() for k, v in CK.engine_vlm_state_dict(eng).items()}, str(d / "model.safetensors"))
    if dataset_statistics is not None:          # save_dataset_statistics (finetune.py:531): q01/q99 etc. used to un-normalise actions
        json.dump(dataset_statistics, open(d / "dataset_statistics.json", "w"), indent=2)
    return d


def _pad_to(batch: dict, L: int, pad_id: int) -> dict:
    """Right-pad (collator semantics, data_utils.py:114-134) a batch to the static sequence length of the captured step."""
    cur = batch["input_ids"].shape[1]
    if cur == L:
        return batch
    if cur > L:
        raise ValueError(f"batch with {cur} tokens exceeds the captured sequence length {L}: raise --max_seq_len")
    out = dict(batch)
    pad = lambda t, v: torch.nn.functional.pad(t, (0, L - cur), value=v)
    out["input_ids"], out["labels"] = pad(batch["input_ids"], pad_id), pad(batch["labels"], -100)
    out["attention_mask"] = pad(batch["attention_mask"].to(torch.bool), False)
    return out


def batch_stream(cfg: FinetuneConfig, mcfg, dev: str, rank: int, batches=None):
    """Endless iterator over collated batches: an explicit iterable, ``--batch_file`` (one .pt dict, or a directory of them,
    cycled in sorted order; every rank starts at its own offset - the reference's ranks draw independent shuffles,
    finetune.py:988-994), or seeded synthetic batches (a new one every micro-step)."""
    from . import synthetic as S
    if batches is not None:
        while True:
            n = 0
            for b in batches:
                n += 1
                yield {k: v.to(dev) for k, v in b.items()}
            if n == 0:
                raise ValueError("empty batch iterable")
    elif cfg.batch_file:
        files = sorted(str(p) for p in Path(cfg.batch_file).glob("*.pt")) if os.path.isdir(cfg.batch_file) else [cfg.batch_file]
        if not files:
            raise FileNotFoundError(f"no .pt batch files under {cfg.batch_file}")
        i = rank % len(files)
        while True:
            yield {k: v.to(dev) for k, v in torch.load(files[i], weights_only=True).items()}
            i = (i + 1) % len(files)
    else:
        i = 0
        while True:
            yield S.make_batch(mcfg, cfg.batch_size, dev, seed=1_000_003 * cfg.seed + 7919 * rank + i, P=32, ragged=True)
            i += 1


def finetune(cfg: FinetuneConfig, batches=None, explicit=()) -> dict:
    """``batches``: optional iterable of collated batch dicts (util/data_utils.py:165-172 contract); ``explicit``: names of
    the flags given on the command line (parse_args records them)."""
    from . import ddp, engine as E, synthetic as S
    check_supported(cfg, explicit or getattr(cfg, "_explicit", ()))
    rank, local, world = ddp.init_process_group_from_env()
    torch.cuda.set_device(local)
    dev = f"cuda:{local}"
    from . import checkpoints as CK
    # model geometry: explicit --backbone, else read off the --vlm_path state dict (which backbones, widths, depths), else BASELINE
    # configs[1].  The reference builds the model from the checkpoint's config.json (finetune.py:777-816); --tiny is the plumbing size.
    vlm_sd = CK.load_file(cfg.vlm_path) if (cfg.vlm_path and os.path.isfile(cfg.vlm_path)) else None
    if cfg.tiny:
        mcfg = E.NAMED_CONFIGS[cfg.backbone or "tiny"]()
    elif cfg.backbone is not None:
        mcfg = E.NAMED_CONFIGS[cfg.backbone]()
    elif vlm_sd is not None:
        mcfg = CK.infer_config(vlm_sd)
    else:
        mcfg = E.config2()
    mcfg.n_img = cfg.num_images_in_input
    mcfg.pro = bool(cfg.use_pro_version)
    mode = train_mode(cfg)
    W = S.make_weights(mcfg, dev, seed=cfg.seed)                  # identical on all ranks == DDP's initial broadcast
    if vlm_sd is not None:                                        # local VLM state dict (HF-style or native Prismatic keys)
        W.update(CK.split_reference_state_dict(vlm_sd, mcfg))
    lora_sd = None
    if cfg.resume:                    # head / proprio (finetune.py:275-278) + the action queries + whatever else this mode trained
        if not (cfg.resum_vla_path and os.path.isdir(cfg.resum_vla_path)):
            raise FileNotFoundError(f"--resume: --resum_vla_path {cfg.resum_vla_path!r} is not a checkpoint directory")
        W["head"], W["proprio"], aq = CK.load_run_dir(cfg.resum_vla_path, cfg.resume_step, with_action_queries=True)
        if aq is not None:
            W["action_queries"] = aq
        rd = Path(cfg.resum_vla_path)
        if mode == "full":            # the trained VLM lives under lora_adapter/ (save_training_checkpoint, reference quirk kept)
            f = rd / "lora_adapter" / "model.safetensors"
            if not f.exists():
                raise FileNotFoundError(f"--resume of a full fine-tune needs {f}")
            W.update(CK.split_reference_state_dict(CK.load_file(str(f)), mcfg))
        elif mode == "lora":
            f = rd / "lora_adapter" / "adapter_model.safetensors"
            if not f.exists():
                raise FileNotFoundError(f"--resume of a LoRA fine-tune needs {f}")
            lora_sd = CK.load_file(str(f))
    eng = E.VLAEngine(mcfg, W, dev)
    if world > 1:
        eng.reducer = ddp.FlatGradReducer(algo=cfg.ddp_algo)
    trainer = None
    if mode == "lora":
        from .trainers import LoRAFinetune
        trainer = LoRAFinetune(eng, rank=cfg.lora_rank, seed=cfg.seed, fp8=cfg.fp8_base_weights, dropout=cfg.lora_dropout)
        if lora_sd is not None:
            trainer.load_lora_state_dict(lora_sd)
    elif mode == "full":
        from .trainers import FullFinetune
        trainer = FullFinetune(eng)
    use_graph = cfg.use_graph
    (trainer or eng).set_grad_accumulation(cfg.grad_accumulation_steps)
    if cfg.objective != "l1":
        trainer.set_objective(cfg.objective)
    stream = batch_stream(cfg, mcfg, dev, rank, batches)
    pad_id = min(S.PAD_ID, mcfg.llm.vocab - 1)
    cur = next(stream)
    L = cfg.max_seq_len or cur["input_ids"].shape[1]
    cur = _pad_to(cur, L, pad_id)
    training = cfg.phase == "Training"
    gen = torch.Generator(device=dev).manual_seed(cfg.seed * 7919 + rank)
    noise = torch.zeros(mcfg.chunk, mcfg.action_dim * mcfg.llm.d, device=dev, dtype=torch.bfloat16)
    run_dir = Path(cfg.run_root_dir) / (cfg.run_id_override or f"native+{cfg.dataset_name}+b{cfg.batch_size * world}+lr-{cfg.learning_rate}")
    stats = json.load(open(cfg.dataset_statistics_file)) if cfg.dataset_statistics_file else None
    static = None
    if use_graph:
        static = {k: v.clone() for k, v in cur.items()}
        if trainer is not None:       # LoRA / full fine-tune: forward + backward as one captured graph (trainers.BackboneTrainer.capture)
            trainer.capture(static, noise if training else None)
        else:
            eng.capture(static, noise if training else None, conservative_rows=cfg.conservative_rows)
    log, t0, saved_at, steps_done = [], time.time(), None, 0
    for batch_idx, g, log_step, boundary, save, last in loop_plan(cfg):
        nxt = _pad_to(next(stream), L, pad_id)                    # one batch of look-ahead: its vision stage runs inside this step
        if training:   # fresh N(0, 0.02^2) perturbation every call (action_heads.py:14-17, 69-72)
            noise.copy_((torch.randn(noise.shape, device=dev, generator=gen) * 0.02).to(torch.bfloat16))
        lr = lr_at(g, cfg)
        if trainer is not None and use_graph:
            for k in static:
                static[k].copy_(cur[k])
            loss3 = trainer.train_step_graphed(lr)
        elif trainer is not None:
            loss3 = trainer.train_step(cur, lr, noise if training else None)
        elif use_graph:
            for k in static:
                if k != "pixel_values" or batch_idx == 0:
                    static[k].copy_(cur[k])
            eng.stage_next_pixels(nxt["pixel_values"])
            loss3 = eng.train_step_graphed(lr)
        else:
            loss3 = eng.train_step(cur, lr, noise if training else None)
        steps_done += int(boundary)
        if boundary and world > 1 and cfg.sync_check_freq > 0 and steps_done % cfg.sync_check_freq == 0:
            eng.flush()                      # (the captured adapter step leaves its update pending: compare what the ranks really hold)
            flats = [eng.head.P.data] + ([trainer.P.data] if trainer is not None else [])
            ddp.assert_ranks_in_sync(flats, what=f"parameters after optimizer step {steps_done}")
        if boundary and (log_step % cfg.wandb_log_freq == 0 or last):          # the only host sync, every log_freq gradient steps
            l = loss3.tolist()
            if not all(x == x for x in l):
                raise FloatingPointError(f"non-finite loss at step {log_step}: {l} (a captured step replayed on a batch whose action "
                                         "block starts before the frozen live-row window poisons the loss: --conservative_rows true)")
            log.append(dict(step=log_step, loss_value=l[0], curr_action_l1_loss=l[1], next_actions_l1_loss=l[2], lr=lr))
            if rank == 0:
                print(json.dumps(log[-1]), flush=True)
        if save:
            eng.flush()                      # the graphed step leaves its parameter update pending (engine.capture)
            if rank == 0:
                save_training_checkpoint(cfg, run_dir, log_step, eng, stats, trainer)
            saved_at = log_step
            if world > 1:
                torch.distributed.barrier()  # finetune.py:544, 575
        cur = nxt
        final_step = log_step
    eng.flush()
    torch.cuda.synchronize()
    if saved_at != final_step and rank == 0:     # never discard a run: the reference only saves on save_freq multiples
        save_training_checkpoint(cfg, run_dir, final_step, eng, stats, trainer)
    model = dict(vit=[dict(v.as_oracle(), img=v.img) for v in mcfg.vit], llm=dict(mcfg.llm.as_oracle(), d=mcfg.llm.d, inter=mcfg.llm.inter, vocab=mcfg.llm.vocab),
                 n_img=mcfg.n_img, num_blocks=mcfg.num_blocks, pro=mcfg.pro)
    return dict(log=log, seconds=time.time() - t0, steps=steps_done, world=world, final_step=final_step, run_dir=str(run_dir), mode=mode, model=model)
