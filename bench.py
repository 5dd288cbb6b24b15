#!/usr/bin/env python3
"""Headline benchmark: adapter-only fine-tune throughput (samples/s) of SigLIP-224 + Qwen2.5-0.5B + Pro action head,
bf16, per-GPU batch 32 (BASELINE.json configs[1]; configs[2] with --gpus 8), forward + backward + AdamW, synthetic
224x224 image + 32-token prompt batches, random-init weights (no network).

    python bench.py --gpus N --steps K --warmup W
    N>1:  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Rank 0 prints ONE JSON line.  `value` = whole-job samples/s (all ranks), inputs resident in HBM when the timed
region starts.  `roofline` prices the dominant kernel (the bf16 MFMA GEMM) with HIP events on the launch stream;
`cpu_baseline` times the CPU oracle (fp32, torch CPU threads) on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# (GPU_MAX_HW_QUEUES is left at the HIP default of 4 on purpose: with the exchange stream running, the rehearsed step
#  measured 28.0 ms at 4 queues, 29.9 at 6 or 16 and 37.7 at 8 - same box, --rehearse-exchange)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

MFMA_BF16_PEAK_TFLOPS = 2500.0     # dense, /opt/skills/guides/MI355X_MICROARCH.md "Peak BF16/FP16 MFMA"
MFMA_FP8_PEAK_TFLOPS = 5000.0      # dense, same guide ("Peak FP8 MFMA"): the e4m3 base products of --mode lora --fp8


def gemm_flops_of_call(a, b, batched, ext=None):
    M, K = a.shape[-2:]
    N = b.shape[-2]
    nb = a.shape[0] if batched else 1
    return 2.0 * nb * M * N * (K + (ext[0].shape[1] if ext is not None else 0))      # (K extension: the low-rank branch of a LoRA Linear)


def gemm_bytes_of_call(a, b, batched, kw):
    """Algorithmic HBM bytes of one launch: A and B read once, C written once (bf16), + residual read, + the second SwiGLU
    output / the SwiGLU-backward pre-activations; rows skipped by c_live are not counted."""
    M, K = a.shape[-2:]
    N = b.shape[-2]
    nb = a.shape[0] if batched else 1
    c_rows = M
    if kw.get("c_live") is not None:
        period, first = kw["c_live"]
        c_rows = M * (period - first) // period
    eb = 1.0 if kw.get("fp8") is not None else 2.0                      # e4m3 operands: one byte per element (+ the fp32 row scales)
    by = eb * nb * (M * K + N * K) + 2.0 * nb * c_rows * N + (4.0 * (M + N) if kw.get("fp8") is not None else 0.0)
    if kw.get("ext") is not None:
        by += 2.0 * (M + N) * kw["ext"][0].shape[1]
    if kw.get("residual") is not None:
        by += 2.0 * nb * M * N
    if kw.get("act", 0) == 4:            # SwiGLU forward: h = [M, N/2]
        by += 2.0 * nb * M * (N // 2)
    if "_swiglu_bwd" in kw:              # dGU [M, 2N] written, GU [M, 2N] read (C above counted [M, N] once)
        by += 2.0 * nb * (3 * M * N)
    return by


def measure_gemm_roofline(eng, batch, noise, lr, reps=10, record_only=False, step_fn=None):
    """Live HIP-event timing of the dominant kernel (gemm_nt_kernel) on the stream it is launched on (torch's current
    stream == the stream handed to the C ABI).  One eager step records every GEMM launch of a training step (operands,
    epilogue, algorithmic FLOPs); each DISTINCT launch signature is then replayed `reps` times back-to-back between two
    events (so the GPU, not the Python launcher, paces the interval) and weighted by its count per step.
    step_fn: the eager step to record (default: the adapter-only engine step; --mode lora / full pass their trainer's)."""
    from vla_adapter_amd import ops
    calls, fam_of = {}, {}          # fam_of: launch signature -> kernel family (for the per-family traffic split of the PMC passes)
    orig = ops.gemm_nt

    def rec(a, b, **kw):
        r = orig(a, b, **kw)
        ext = kw.get("ext")
        sig = (tuple(a.shape), tuple(b.shape), a.stride(-2), b.stride(-2), kw.get("act", 0), kw.get("bias") is not None,
               kw.get("residual") is not None, ext[0].shape[1] if ext is not None else 0, kw.get("fp8") is not None)
        if sig not in calls:
            kw2 = dict(kw)
            if kw2.get("out") is None and kw2.get("act", 0) != 4:
                kw2["out"] = r
            calls[sig] = [0, gemm_flops_of_call(a, b, a.dim() == 3, ext), a, b, kw2, gemm_bytes_of_call(a, b, a.dim() == 3, kw2)]
            fam_of[sig] = "gemm256_kernel" if (kw.get("fp8") is None and ext is None and orig(a, b, **dict(kw2, query_256=True))) else "gemm_nt_kernel"
        calls[sig][0] += 1
        return r

    orig_sw = ops.gemm_swiglu_bwd

    def rec_sw(d_, w_, gu_, out=None, gu_group=None, ext=None, fp8=None):        # the dH GEMM with the SwiGLU backward in its epilogue
        r = orig_sw(d_, w_, gu_, out=out, gu_group=gu_group, ext=ext, fp8=fp8)
        sig = (tuple(d_.shape), tuple(w_.shape), d_.stride(-2), w_.stride(-2), 5, False, True, ext[0].shape[1] if ext is not None else 0, fp8 is not None)
        if sig not in calls:
            kw_sw = dict(_swiglu_bwd=(gu_, r, gu_group), ext=ext, fp8=fp8)
            calls[sig] = [0, gemm_flops_of_call(d_, w_, False, ext), d_, w_, kw_sw, gemm_bytes_of_call(d_, w_, False, kw_sw)]
        calls[sig][0] += 1
        return r

    orig_tng = ops.gemm_tn_grouped

    def rec_tng(problems):                                    # one launch over the tile lists of several dW products (<= 48 each)
        r = orig_tng(problems)
        for i in range(0, len(problems), ops.TN_GROUP_MAX):
            chunk = problems[i:i + ops.TN_GROUP_MAX]
            sig = ("tng",) + tuple((d.M, d.N1, d.N2) for d in chunk)
            if sig not in calls:
                fl = sum(2.0 * d.M * d.N1 * d.N2 for d in chunk)
                by = sum(2.0 * (d.M * d.N1 + d.M * d.N2 + d.N1 * d.N2) for d in chunk)
                calls[sig] = [0, fl, chunk, None, dict(_tng=True), by]
            calls[sig][0] += 1
        return r

    orig_tn = ops.gemm_tn

    def rec_tn(a, b, **kw):                                   # dW = dY^T X (vla_gemm_bf16_tn): contraction over the rows
        r = orig_tn(a, b, **kw)
        M = kw.get("rows") or a.shape[-2]
        N1 = kw["a_cols"][0] if kw.get("a_cols") else a.shape[-1]
        N2, nb = b.shape[-1], (a.shape[0] if a.dim() == 3 else 1)
        sig = ("tn", tuple(a.shape), tuple(b.shape), M, a.stride(-2), b.stride(-2), bool(kw.get("accumulate")))
        if sig not in calls:
            kw2 = dict(kw, out=r, _tn=True)
            calls[sig] = [0, 2.0 * nb * M * N1 * N2, a, b, kw2, 2.0 * nb * (M * N1 + M * N2 + N1 * N2 * (2 if kw.get("accumulate") else 1))]
        calls[sig][0] += 1
        return r

    def replay(a, b, kw):
        if "_swiglu_bwd" in kw:
            gu_, out_, grp = kw["_swiglu_bwd"]
            return orig_sw(a, b, gu_, out=out_, gu_group=grp, ext=kw.get("ext"), fp8=kw.get("fp8"))
        if kw.get("_tng"):
            return orig_tng(a)
        if kw.get("_tn"):
            return orig_tn(a, b, **{k: v for k, v in kw.items() if k != "_tn"})
        return orig(a, b, **kw)

    ops.gemm_nt, ops.gemm_swiglu_bwd, ops.gemm_tn, ops.gemm_tn_grouped = rec, rec_sw, rec_tn, rec_tng
    reducer, eng.reducer = eng.reducer, None      # rank-0-only probe step: it must not issue collectives
    try:
        (step_fn or (lambda: eng.train_step(batch, lr, noise)))()
        torch.cuda.synchronize()
    finally:
        ops.gemm_nt, ops.gemm_swiglu_bwd, ops.gemm_tn, ops.gemm_tn_grouped = orig, orig_sw, orig_tn, orig_tng
        eng.reducer = reducer
    if record_only:
        fams = {}
        for sig, c in calls.items():
            fam = fam_of.get(sig, "gemm_tn_kernel" if sig[0] in ("tn", "tng") else "gemm_nt_kernel")
            d_ = fams.setdefault(fam, dict(launches=0, algorithmic_bytes=0.0, flops=0.0))
            d_["launches"] += c[0]
            d_["algorithmic_bytes"] += c[0] * c[5]
            d_["flops"] += c[0] * c[1]
        return dict(launches=sum(c[0] for c in calls.values()), flops=sum(c[0] * c[1] for c in calls.values()),
                    bytes=sum(c[0] * c[5] for c in calls.values()), families=fams)
    total_t = total_f = total_b = total_pk = 0.0
    n = 0
    per = []
    for sig, (cnt, fl, a, b, kw, nbytes) in calls.items():
        for _ in range(2):
            replay(a, b, kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            replay(a, b, kw)
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) * 1e-3 / reps
        total_t += cnt * t
        total_f += cnt * fl
        total_b += cnt * nbytes
        # seconds this launch would take at ITS peak: e4m3 base products against the fp8 MFMA peak (their bf16 extension is < 10 % of
        # the contraction and is priced with them - optimistic for the launch, i.e. pessimistic for the fraction)
        total_pk += cnt * fl / ((MFMA_FP8_PEAK_TFLOPS if kw.get("fp8") is not None else MFMA_BF16_PEAK_TFLOPS) * 1e12)
        n += cnt
        if sig[0] == "tng":
            per.append((cnt * t, cnt, ("TN-grouped", len(sig) - 1), sig[1], fl / t / 1e12))
        else:
            per.append((cnt * t, cnt, ("TN",) + tuple(sig[1]) if sig[0] == "tn" else sig[0], sig[2] if sig[0] == "tn" else sig[1], fl / t / 1e12))
    per.sort(key=lambda x: x[0], reverse=True)      # (by time only: equal times must not fall through to comparing shape tuples with the "TN" tag)
    if os.environ.get("VLA_DUMP_GEMMS"):          # full per-signature table (tuning aid)
        with open(os.environ["VLA_DUMP_GEMMS"], "w") as fdump:
            for x in per:
                fdump.write(f"{x[0] * 1e3:8.3f} ms/step  count {x[1]:4d}  A {list(x[2])}  B {list(x[3])}  {x[4]:7.1f} TF/s\n")
    top = [dict(ms_per_step=round(x[0] * 1e3, 3), count=x[1], A=list(x[2]), B=list(x[3]), tflops=round(x[4], 1)) for x in per[:8]]
    return dict(launches=n, seconds=total_t, flops=total_f, tflops=total_f / total_t / 1e12, top=top, bytes=total_b,
                frac_of_own_peaks=total_pk / total_t)


def roofline_block(roof, convention):
    """The `roofline` object of a --mode lora / full line: every GEMM launch of one eager step replayed in isolation (see
    measure_gemm_roofline).  `frac` prices bf16 launches against the bf16 MFMA peak and e4m3-operand launches against the fp8 peak."""
    return {"bound": "mfma", "kernel": "gemm256_kernel + gemm_nt_kernel (+ K extension) + gemm_tn*_kernel: all GEMM launches of one step",
            "achieved": round(roof["tflops"], 1), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(roof["frac_of_own_peaks"], 4), "frac_vs_bf16_peak": round(roof["tflops"] / MFMA_BF16_PEAK_TFLOPS, 4),
            "peak_fp8_launches": MFMA_FP8_PEAK_TFLOPS, "traffic": None, "flop_convention": convention,
            "timing": "isolated: each distinct launch signature of one eager step replayed back-to-back between two HIP events on its launch "
                      "stream, weighted by its count per step",
            "launches_per_step": roof["launches"], "gemm_ms_per_step": round(roof["seconds"] * 1e3, 3), "gemm_flops_per_step": roof["flops"],
            "algorithmic_bytes_per_launch": round(roof["bytes"] / roof["launches"]), "avg_launch_us": round(roof["seconds"] / roof["launches"] * 1e6, 1),
            "top_launches": roof["top"]}


def pmc_traffic():
    """HBM-side bytes per GEMM launch from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate
    runs of this same command; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950).  None if not collected."""
    for name in ("r04_gemm_traffic_pmc.json", "r03_gemm_traffic_pmc.json", "r02_gemm_traffic_pmc.json", "r01_gemm_traffic_pmc.json"):
        f = os.path.join(ROOT, "profiles", name)
        try:
            d = json.load(open(f))
            return {"bytes_per_launch": round(d["bytes_per_launch"]), "source": "profiles/" + name, "by_kernel_family": d.get("by_kernel_family"),
                    "note": "TCC fetch/write sizes = traffic between the 8 private L2s and the fabric (Infinity Cache + HBM), not HBM alone: "
                            "every XCD streams the weight operand into its own L2 (DESIGN.md 4)"}
        except Exception:
            continue
    return None


def in_situ_roofline(gemm_flops_per_step):
    """GEMM FLOPs of one step / sum of the IN-SITU GEMM kernel time of one step (rocprofv3 --kernel-trace of the default step alone,
    tools/profile_step.sh -> profiles/r02_gemm_in_situ.json; reproducible from profiles/r02_kernel_summary.csv with a calculator:
    sum us_per_step over the gemm rows).  Kernels of concurrent streams share the CUs, so a launch's in-situ duration includes the
    time it waits for CUs other streams hold: this is a LOWER bound on the kernel's own rate."""
    from vla_adapter_amd import flops
    for name in ("r04_gemm_in_situ.json", "r03_gemm_in_situ.json", "r02_gemm_in_situ.json"):
        try:
            d = json.load(open(os.path.join(ROOT, "profiles", name)))
        except Exception:
            continue
        us = d["gemm_us_per_step_in_situ"]
        # the profile is evidence for the code it was taken from: with other kernel / schedule sources it is reported as stale and
        # frac_in_situ is withheld (null) instead of dividing today's FLOPs by another build's kernel time
        stale = d.get("source_digest") != flops.source_digest()
        return {"frac_in_situ": None if stale else round(gemm_flops_per_step / us / 1e6 / MFMA_BF16_PEAK_TFLOPS, 4),
                "frac_in_situ_of_profiled_build": d.get("frac_in_situ"), "stale": stale, "profile_source_digest": d.get("source_digest"),
                "gemm_us_per_step_in_situ": us, "gemm_launches_per_step": d["gemm_launches_per_step"],
                "source": f"profiles/{name} (+ {name[:3]}_kernel_summary_steady.csv)"}
    return None


def gemm256_counters():
    """rocprofv3 --pmc counters of the dominant kernel (gemm256_kernel) on the four big step shapes (tools/collect_gemm256_pmc.sh ->
    profiles/r04_gemm256_pmc.json): MFMA-pipe busy share, clock held, wave wait share, fabric bytes over algorithmic bytes.  Stamped
    with the source digest of the build they were taken from: another build's counters are reported as stale."""
    from vla_adapter_amd import flops
    f = os.path.join(ROOT, "profiles", "r04_gemm256_pmc.json")
    try:
        d = json.load(open(f))
    except Exception:
        return None
    keep = ("name", "M", "N", "K", "tiles", "avg_us", "tflops", "mfma_busy_share", "clock_ghz_from_gui_active", "wave_wait_any_share",
            "wave_issue_stall_share", "lds_bank_conflict_share_of_lds_cycles", "traffic_over_algorithmic")
    return {"source": "profiles/r04_gemm256_pmc.json", "stale": d.get("source_digest") != flops.source_digest(),
            "profile_source_digest": d.get("source_digest"), "shapes": [{k: sh.get(k) for k in keep} for sh in d.get("shapes", [])]}


def host_cores() -> int:
    """CPU threads this process may really use (affinity and cgroup quota; the GPU box exposes 128 logical CPUs
    but grants a one-GPU job a share of them)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except Exception:
        pass
    return max(1, min(n, int(os.environ.get("VLA_CPU_THREADS", "16"))))


def cpu_baseline(cfg, W, batch, noise, nsample, seconds_cap=60.0):
    """The CPU oracle (fp32 restatement, torch CPU threads) on `nsample` samples of the same batch: forward + backward
    + AdamW on the trainable set.  kind="port" (the reference's own Python cannot travel to the GPU box)."""
    from oracle import vla_oracle as O
    torch.set_num_threads(host_cores())
    f = lambda sd: {k: v.float().cpu() for k, v in sd.items()}
    llm = f(W["llm"])
    leaf = lambda d: {k: v.requires_grad_(True) for k, v in d.items()}
    OW = dict(vit=[f(s) for s in W["vit"]], proj=f(W["proj"]), llm=llm, embed=llm["embed_tokens.weight"],
              action_queries=W["action_queries"].float().cpu().requires_grad_(True), head=leaf(f(W["head"])), proprio=leaf(f(W["proprio"])))
    cb = {k: v[:nsample].cpu() for k, v in batch.items()}
    cb["pixel_values"], cb["proprio"] = cb["pixel_values"].float(), cb["proprio"].bfloat16().float()
    ocfg = dict(vit=[v.as_oracle() for v in cfg.vit], fused=cfg.fused, llm=cfg.llm.as_oracle(), n_img=cfg.n_img, pro=True,
                num_blocks=cfg.num_blocks)
    nz = noise.float().cpu() if noise is not None else None
    params = [OW["action_queries"]] + [p for k, p in OW["head"].items() if "film_gen" not in k] + list(OW["proprio"].values())
    steps, t0 = 0, time.time()
    while True:
        out = O.vla_forward(cb, OW, ocfg, emu=False, noise=nz)
        out["loss"].backward()
        with torch.no_grad():
            for p in params:
                if p.grad is not None:
                    pn, _, _ = O.adamw_step(p, p.grad, torch.zeros_like(p), torch.zeros_like(p), 1, 5e-4)
                    p.copy_(pn)
                    p.grad = None
        steps += 1
        el = time.time() - t0
        if el > 10.0 or steps >= 3 or el * (steps + 1) / steps > seconds_cap:
            break
    return dict(value=steps * nsample / el, unit="samples/s", cores=torch.get_num_threads(), kind="port",
                sample=f"{steps} step(s) x {nsample} sample(s) of the same synthetic batch, fp32 torch-CPU oracle fwd+bwd+AdamW, {el:.1f} s")


def bench_full(args, cfg, W, eng, batch, noise, lr, rank, local, world, B, P):
    """BASELINE configs[3]: full-backbone unfreeze (ViT + 0.5B LLM + adapter), bf16, batch 16 / GPU: forward + full backward
    (dX and dW of every Linear, norms, embeddings) + AdamW over ~1.1 G parameters.  Not the headline metric (that is the
    adapter-only configs[1]); same JSON contract."""
    from vla_adapter_amd import flops
    from vla_adapter_amd.full_finetune import FullFinetune
    ft = FullFinetune(eng)
    ft.set_objective(args.objective)

    def sync():
        if world > 1:
            dist.barrier(device_ids=[local]) if dist.get_backend() == "nccl" else dist.barrier()
        torch.cuda.synchronize()

    if args.eager:
        step = lambda: ft.train_step(batch, lr, noise)
    else:
        ft.capture(batch, noise)
        step = lambda: ft.train_step_graphed(lr)
    for _ in range(args.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss3 = step()
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=f"cuda:{local}", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    ms = dt / args.steps * 1e3
    fl = flops.step_flops_per_sample(cfg, L=P + 64, row0=0)
    full = 3.0 * fl["forward"]                               # config 4 ~ 3 x forward (dX + dW for every op)
    roof = None
    if rank == 0 and not args.no_probe:
        roof = roofline_block(measure_gemm_roofline(eng, batch, noise, lr, step_fn=lambda: ft.train_step(batch, lr, noise)),
                              "executed GEMM FLOPs of one step: forward, dX and dW = dY^T X of every Linear (attention products are not GEMM launches)")
    if rank == 0:
        nparam = ft.P.numel + ft.head.P.numel
        print(json.dumps({
            "metric": "fine-tune samples/sec (224px img + 32-tok prompt), FULL unfreeze (ViT + LLM + adapter), fwd+bwd+AdamW" +
                      ("" if args.objective == "l1" else ", token cross-entropy objective (lm_head on the text rows)"),
            "value": round(world * B * args.steps / dt, 2), "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": ("BASELINE configs[3]: Prismatic SigLIP-224 + Qwen2.5-0.5B" if args.backbone == "config2" else args.backbone) +
                                   f" + Pro action head, full-backbone unfreeze, {cfg.n_img} image(s) ({cfg.n_patches} patches) + 32-token prompt + "
                                   f"64 action queries (S={cfg.n_patches + P + 64})",
                       "global_batch": world * B, "per_gpu_batch": B, "seq_len": cfg.n_patches + P + 64, "parallelism": f"dp{world}",
                       "weights": "random-init", "trainable_parameters": nparam, "launch": "eager" if args.eager else "hipGraph replay",
                       "captured_segment_graphs": None if args.eager else len(ft._segs),
                       "final_loss": round(float(loss3[0]), 5)},
            "step_tflops_per_gpu": round(full * B / (ms * 1e-3) / 1e12, 1),
            "step_frac_of_bf16_mfma_peak": round(full * B / (ms * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4),
            "gflop_per_sample": {"autograd_convention": round(full / 1e9, 1)},
            "roofline": roof,
        }), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def bench_lora(args, cfg, W, eng, batch, noise, lr, rank, local, world, B, P):
    """LoRA fine-tune step (reference --use_lora, rank 64, all Linears of the ViT / projector / LLM + the adapter head): forward with the
    low-rank terms, full-sequence dX chain, dA / dB, AdamW over adapters + head.  --fp8: the same step a second time with the frozen
    base weights' products on e4m3 operands (BASELINE configs[4]: "LoRA-adapter fine-tune, fp8 MFMA weight path"), reported BESIDE
    the bf16 line (reduced precision: never `value`).  Not the headline metric; same JSON contract."""
    import gc
    from vla_adapter_amd import flops
    from vla_adapter_amd.lora_finetune import LoRAFinetune
    fl = flops.step_flops_per_sample(cfg, L=P + 64, row0=0)
    work = 2.0 * fl["forward"] + fl["head_fwd"]                # forward + dX of every op + the head's dW (base dW is not computed; rank terms ~2-10 %)
    conv = ("executed GEMM FLOPs of one step: base products of forward and dX incl. their rank-r K extension, the skinny t = 2 x A^T / dt = 2 dy B "
            "products, dA / dB as TN products, the action head's forward / dX / dW")

    def run(fp8):
        ft = LoRAFinetune(eng, rank=args.lora_rank, fp8=fp8)
        ft.set_objective(args.objective)
        if args.eager:
            step = lambda: ft.train_step(batch, lr, noise)
        else:
            ft.capture(batch, noise)
            step = lambda: ft.train_step_graphed(lr)
        for _ in range(args.warmup):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loss3 = step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        roof = None
        if not args.no_probe:
            roof = roofline_block(measure_gemm_roofline(eng, batch, noise, lr, step_fn=lambda: ft.train_step(batch, lr, noise)), conv)
        out = dict(dt=dt, ms=dt / args.steps * 1e3, loss=round(float(loss3[0]), 5), roof=roof, nparam=ft.P.numel + ft.head.P.numel,
                   nseg=None if args.eager else len(ft._segs), fp8_keys=(len(ft.Q), len(ft.QT), len(ft.L)))
        del ft, step
        gc.collect()
        torch.cuda.empty_cache()
        return out

    if args.fp8_only:                    # profiling aid: the trace then holds the fp8 step alone
        r8 = run(True)
        if rank == 0:
            print(json.dumps({"metric": "LoRA step with fp8 base-weight products only (profiling run)", "value": round(B * args.steps / r8["dt"], 2), "unit": "samples/s",
                              "ms_per_step": round(r8["ms"], 3), "steps": args.steps, "warmup": args.warmup, "dtype": "fp8 base products + bf16", "roofline": r8["roof"]}), flush=True)
        return
    r16 = run(False)
    r8 = run(True) if args.fp8 else None
    desc = {"config2": "Prismatic SigLIP-224 + Qwen2.5-0.5B", "dinosiglip-0_5b": "DINOv2-L + SigLIP-so400m fused + Qwen2.5-0.5B (the reference's documented recipe)",
            "config5": "BASELINE configs[4]: DINOv2-L + SigLIP-so400m fused + Qwen2.5-1.5B"}[args.backbone]
    if rank == 0:
        fp8_line = None
        if r8 is not None:
            fp8_line = {"value": round(B * args.steps / r8["dt"], 2), "unit": "samples/s", "ms_per_step": round(r8["ms"], 3), "steps": args.steps,
                        "dtype": "fp8 (OCP e4m3) base-weight products with fp32 accumulation + bf16 rank-r branch, activations / gradients / adapters bf16",
                        "final_loss": r8["loss"], "linears_on_fp8_forward_backward_of": list(r8["fp8_keys"]),
                        "step_tflops_per_gpu": round(work * B / (r8["ms"] * 1e-3) / 1e12, 1), "roofline": r8["roof"],
                        "what": "every base product x W^T (forward) and dy W (dX) of the LoRA-wrapped Linears on e4m3 operands: weights quantised once "
                                "(one scale per output / input channel), activations per row inside the norm that produces them or by one pass over "
                                "the producer's output, dy per row; the low-rank branch adds in bf16 to the dequantised product in the same "
                                "accumulator (GEMM K extension).  The reference has no fp8 code: parity unpinned, never the headline"}
        print(json.dumps({
            "metric": "fine-tune samples/sec (224px img + 32-tok prompt), LoRA rank %d on every Linear + adapter head, fwd+bwd+AdamW" % args.lora_rank +
                      ("" if args.objective == "l1" else ", token cross-entropy objective (lm_head on the text rows)"),
            "value": round(B * args.steps / r16["dt"], 2), "unit": "samples/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(r16["ms"], 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"{desc} + Pro action head, LoRA fine-tune (vla-scripts/finetune.py:832-844), {cfg.n_img} image(s) "
                                   f"({cfg.n_patches} patches) + 32-token prompt + 64 action queries (S={cfg.n_patches + P + 64})",
                       "global_batch": B, "per_gpu_batch": B, "seq_len": cfg.n_patches + P + 64, "parallelism": "dp1", "weights": "random-init",
                       "trainable_parameters": r16["nparam"], "captured_segment_graphs": r16["nseg"],
                       "launch": "eager" if args.eager else "hipGraph replay", "final_loss": r16["loss"]},
            "step_tflops_per_gpu": round(work * B / (r16["ms"] * 1e-3) / 1e12, 1),
            "step_frac_of_bf16_mfma_peak": round(work * B / (r16["ms"] * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4),
            "gflop_per_sample": {"forward": round(fl["forward"] / 1e9, 1), "counted": round(work / 1e9, 1)},
            "roofline": r16["roof"],
            "fp8_base_weights_variant": fp8_line,
        }), flush=True)


MARKER_BYTES = 16 * 256 * 4099      # 4099 workgroups of 256 threads x 16 B (tools/summarise_profiles.py looks for this grid)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch (BASELINE configs[1]: 32; --mode full = configs[3]: 16)")
    ap.add_argument("--mode", default="adapter", choices=["adapter", "full", "lora"],
                    help="adapter: BASELINE configs[1]/[2] (the headline metric); full: configs[3], every VLM parameter trains; "
                         "lora: rank-r adapters on every Linear (the reference's --use_lora) on the 0.5B backbone")
    ap.add_argument("--lora-rank", type=int, default=64)
    ap.add_argument("--fp8", action="store_true", help="--mode lora: also time the step with the frozen base weights' products on OCP e4m3 operands "
                                                       "(BASELINE configs[4]'s fp8 MFMA weight path), printed beside the bf16 line")
    ap.add_argument("--fp8-only", action="store_true", help="--mode lora: time ONLY the fp8 variant (profiling runs)")
    ap.add_argument("--ddp-algo", default=os.environ.get("VLA_DDP_ALGO", "allreduce"), choices=["allreduce", "rs_ag"],
                    help="gradient exchange per bucket: one all-reduce, or reduce-scatter + all-gather (all xGMI links of the node at once)")
    ap.add_argument("--objective", default="l1", choices=["l1", "token_ce"],
                    help="--mode full / lora: L1 regression through the action head (the reference's finetune.py) or the token cross-entropy of its "
                         "native trainer (SURVEY 8f-4: lm_head on the text rows, no action head)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-samples", type=int, default=4, help="samples of the batch the CPU oracle is timed on (3 steps each: about 12 s of CPU work)")
    ap.add_argument("--no-full-backward", action="store_true", help="skip the extra timing of the reference-shaped full LLM backward")
    ap.add_argument("--no-fp8-variant", action="store_true", help="skip the extra timing of the opt-in fp8 frozen-weight forward (NOT the headline: reduced precision)")
    ap.add_argument("--eager", action="store_true", help="launch every kernel from Python instead of replaying hipGraphs")
    ap.add_argument("--rehearse-exchange", action="store_true", help="one GPU: run the step with a one-rank RCCL group and forced gradient collectives")
    ap.add_argument("--no-probe", action="store_true", help="skip the isolated GEMM replays (profiling runs: the trace then holds training steps only)")
    ap.add_argument("--backbone", default="config2", choices=["config2", "dinosiglip-0_5b", "config5"],
                    help="config2: SigLIP-224 + Qwen2.5-0.5B (the headline); dinosiglip-0_5b: DINOv2+SigLIP fused + Qwen2.5-0.5B (the reference's "
                         "documented recipe, README.md:254-274); config5: DINOv2+SigLIP fused + Qwen2.5-1.5B (BASELINE configs[4]'s backbone)")
    ap.add_argument("--n-img", type=int, default=None, help="images per sample (default 1; the documented recipe and BASELINE configs[4] use 2)")
    ap.add_argument("--ragged", action="store_true", help="prompt lengths in [24, 32], right-padded (exercises the mask path; SURVEY 8d)")
    args = ap.parse_args()

    from vla_adapter_amd import ddp, engine as E, flops, synthetic as S
    rank, local, world = ddp.init_process_group_from_env()
    assert world == args.gpus or world == 1 and args.gpus == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    local = local % torch.cuda.device_count()      # ranks > GPUs only in the gloo rehearsal on a one-GPU box
    torch.cuda.set_device(local)
    dev = f"cuda:{local}"
    cfg = E.NAMED_CONFIGS[args.backbone]()
    cfg.n_img = args.n_img or 1
    B, P = (args.batch or (32 if args.mode == "adapter" else 16)), 32
    W = S.make_weights(cfg, dev, seed=0)                    # identical on every rank (== DDP's initial broadcast)
    eng = E.VLAEngine(cfg, W, dev)
    batch = S.make_batch(cfg, B, dev, seed=1000 + rank, P=P, ragged=args.ragged)  # every rank draws its own samples (finetune.py:988-994)
    batch["pixel_values"] = batch["pixel_values"].to(torch.bfloat16)   # finetune.py:339
    noise = (torch.randn(cfg.chunk, cfg.action_dim * cfg.llm.d, device=dev) * 0.02).to(torch.bfloat16)  # phase="Training"
    if world > 1:
        eng.reducer = ddp.FlatGradReducer(algo=args.ddp_algo)
    elif args.rehearse_exchange:
        # one-GPU rehearsal of the RCCL exchange: a ONE-rank "nccl" group and a reducer that issues its collectives as an
        # N-rank job would (a one-rank all-reduce is the identity; grad scale stays 1).  Costs what the exchange machinery
        # costs (streams, events, RCCL launches of the 437 MB buffer), not what the links cost.
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        dist.init_process_group(backend="nccl", rank=0, world_size=1)
        eng.reducer = ddp.FlatGradReducer(algo=args.ddp_algo)
        eng.reducer.world = 2
        type(eng.reducer).grad_scale = property(lambda self: 1.0)
    lr = 5e-4
    if args.mode == "full":
        return bench_full(args, cfg, W, eng, batch, noise, lr, rank, local, world, B, P)
    if args.mode == "lora":
        assert world == 1, "--mode lora: single-GPU measurement"
        return bench_lora(args, cfg, W, eng, batch, noise, lr, rank, local, world, B, P)

    def barrier():
        if dist.get_backend() == "nccl":
            dist.barrier(device_ids=[local])        # pin the device: RCCL otherwise guesses it from the rank
        else:
            dist.barrier()

    def sync():
        if world > 1:
            barrier()
        torch.cuda.synchronize()

    if args.eager:
        step = lambda: eng.train_step(batch, lr, noise)
    else:
        eng.capture(batch, noise)
        step = lambda: eng.train_step_graphed(lr)
    for _ in range(args.warmup):
        step()
    eng.flush()
    sync()
    # profiling runs bracket the timed region with a marker launch (a zero-fill of MARKER_BYTES: a grid no step kernel has), so
    # that tools/summarise_profiles.py can cut the steady-state steps out of the kernel trace (weight init, capture warm-ups
    # and the recording step stay outside)
    marker = torch.empty(MARKER_BYTES, dtype=torch.uint8, device=dev) if args.no_probe else None
    if marker is not None:
        from vla_adapter_amd import ops as _ops
        _ops.zero_(marker)
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss3 = step()
    eng.flush()          # the graphed step defers its RCCL exchange + AdamW into the next step: K steps = K updates
    sync()
    dt = time.perf_counter() - t0
    if marker is not None:
        _ops.zero_(marker)
        torch.cuda.synchronize()
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    ms = dt / args.steps * 1e3
    value = world * B * args.steps / dt
    row0 = eng.live_row0()
    fl = flops.step_flops_per_sample(cfg, L=P + 64, row0=row0)
    full_bwd = None
    if world == 1 and not args.eager and row0 > 0 and not args.no_full_backward:
        # the same step with the reference-shaped backward (every row of the frozen LLM gets dX, task-token dX included):
        # what torch.autograd executes; identical parameter gradients (tests/test_engine_gpu.py), more FLOPs
        eng.full_llm_backward = True
        eng.capture(batch, noise)
        for _ in range(2):
            step()
        eng.flush()
        sync()
        t1 = time.perf_counter()
        nfull = max(3, args.steps // 2)
        for _ in range(nfull):
            step()
        eng.flush()
        sync()
        dtf = (time.perf_counter() - t1) / nfull
        full_bwd = {"value": round(B / dtf, 2), "unit": "samples/s", "ms_per_step": round(dtf * 1e3, 3), "steps": nfull,
                    "step_tflops": round(fl["step"] * B / dtf / 1e12, 1),
                    "frac_of_bf16_mfma_peak": round(fl["step"] * B / dtf / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4)}
        eng.full_llm_backward = False
    if rank == 0 and args.no_probe:
        rec = measure_gemm_roofline(eng, batch, noise, lr, record_only=True)      # ONE more (eager) training step, no replays
        print(json.dumps({"metric": "fine-tune samples/sec (profiling run, no probe)", "value": round(value, 2), "unit": "samples/s",
                          "ms_per_step": round(ms, 3), "steps": args.steps, "warmup": args.warmup,
                          "executed_steps": eng.executed_steps, "marker_grid_x": MARKER_BYTES // 16, "gemm_flops_per_step": rec["flops"],
                          "source_digest": flops.source_digest(),
                          "gemm_launches_per_step": rec["launches"], "gemm_algorithmic_bytes_per_step": rec["bytes"],
                          "gemm_families": rec["families"]}), flush=True)
    elif rank == 0:
        roof = measure_gemm_roofline(eng, batch, noise, lr)
        fp8_var = None
        if world == 1 and not args.eager and not args.no_fp8_variant and not args.no_full_backward:
            # the same step with the frozen backbones' norm-fed projections on e4m3 operands (engine.enable_fp8_frozen: BASELINE
            # configs[4]'s "fp8 MFMA weight path", first part).  Reduced precision, therefore NEVER `value`: reported beside it.
            eng.enable_fp8_frozen()
            eng.capture(batch, noise)
            for _ in range(2):
                step()
            eng.flush()
            sync()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                l8 = step()
            eng.flush()
            sync()
            dt8 = (time.perf_counter() - t1) / args.steps
            fp8_var = {"value": round(B / dt8, 2), "unit": "samples/s", "ms_per_step": round(dt8 * 1e3, 3), "steps": args.steps,
                       "final_loss": round(float(l8[0]), 5),
                       "what": "ViT qkv / fc1 and LLM q|k|v / gate|up forward products on OCP e4m3 (per-row input scales from the fused "
                               "norm+quantise kernels, per-channel weight scales); everything else, incl. the whole backward, bf16"}
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(cfg, W, batch, noise, args.cpu_samples)
        line = {
            "metric": "fine-tune samples/sec (224px img + 32-tok prompt), adapter-only, fwd+bwd+AdamW",
            "value": round(value, 2), "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": ("BASELINE configs[1]: Prismatic SigLIP-224 + Qwen2.5-0.5B + Pro action head, adapter-only fine-tune, "
                                    "1 image (256 patches) + 32-token prompt + 64 action queries (S=352)") if args.backbone == "config2" else
                                   ("BASELINE configs[4] BACKBONE (DINOv2-L + SigLIP-so400m fused, Qwen2.5-1.5B) + Pro action head, adapter-only "
                                    "fine-tune (not the config's LoRA + fp8 mode), 1 image + 32-token prompt + 64 action queries"),
                       "global_batch": world * B, "per_gpu_batch": B, "seq_len": cfg.n_patches + P + 64,
                       "parallelism": f"dp{world}", "weights": "random-init", "launch": "eager" if args.eager else "hipGraph replay",
                       "gradient_exchange": None if world == 1 else f"{dist.get_backend()} {args.ddp_algo}, bucketed, under the backward / the next step's vision stage",
                       "prompts": "ragged 24..32 tokens, right-padded" if args.ragged else "32 tokens",
                       "llm_backward": (f"live rows >= {row0} of {cfg.n_patches + P + 64} (gradient rows that only reach frozen inputs are "
                                        "not computed; parameter gradients identical)") if row0 else "all rows",
                       "final_loss": round(float(loss3[0]), 5)},
            "samples_per_s_per_gpu": round(value / world, 2),
            # FLOPs really executed per step (vla_adapter_amd/flops.py `step_live`) over the measured step time
            "step_executed_tflops_per_gpu": round(fl["step_live"] * B / (ms * 1e-3) / 1e12, 1),
            "step_frac_of_bf16_mfma_peak": round(fl["step_live"] * B / (ms * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4),
            "gflop_per_sample": {"executed": round(fl["step_live"] / 1e9, 1), "autograd_convention": round(fl["step"] / 1e9, 1)},
            "full_backward_variant": full_bwd,
            "fp8_frozen_forward_variant": fp8_var,
            "roofline": {"bound": "mfma", "kernel": "gemm256_kernel + gemm_nt_kernel + gemm_tn_kernel / gemm_tn256_kernel (bf16 MFMA GEMMs, all launches of one step)",
                         "achieved": round(roof["tflops"], 1), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(roof["tflops"] / MFMA_BF16_PEAK_TFLOPS, 4),
                         "frac_in_situ": (in_situ_roofline(roof["flops"]) or {}).get("frac_in_situ"), "in_situ": in_situ_roofline(roof["flops"]),
                         "traffic": (pmc_traffic() or {}).get("bytes_per_launch"), "traffic_detail": pmc_traffic(),
                         "gemm256_counters": gemm256_counters(),
                         "algorithmic_bytes_per_launch": round(roof["bytes"] / roof["launches"]),
                         "avg_launch_us": round(roof["seconds"] / roof["launches"] * 1e6, 1),
                         "timing": "each distinct launch signature of one step replayed back-to-back between two HIP events on its "
                                   "launch stream, weighted by its count (`frac`); `frac_in_situ` divides the same FLOPs by the "
                                   "summed GEMM kernel time of the profiled step (concurrent streams share the CUs)",
                         "launches_per_step": roof["launches"], "gemm_ms_per_step": round(roof["seconds"] * 1e3, 3),
                         "gemm_flops_per_step": roof["flops"], "top_launches": roof["top"]},
            "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
