#!/usr/bin/env python3
"""Batch-1 inference latency of OpenVLAForActionPrediction.predict_action (reference call site
experiments/robot/openvla_utils.py:737-825 -> modeling_prismatic.py:892-972): one call = one 8-action chunk.
The reference README quotes 0.036 s per chunk on an H100 for its default LIBERO setup (DINOv2+SigLIP fused backbone,
third-person + wrist image, Qwen2.5-0.5B, Pro head).  Random-init weights, synthetic inputs."""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from vla_adapter_amd import engine as E, synthetic as S  # noqa: E402
from vla_adapter_amd.modeling_prismatic import OpenVLAForActionPrediction  # noqa: E402

dev = "cuda"
stats = {"libero": {"action": {"q01": [-1.0] * 7, "q99": [1.0] * 7, "min": [-1.0] * 7, "max": [1.0] * 7, "mask": [True] * 6 + [False]}}}
configs = {
    "siglip224+qwen2.5-0.5b, 1 image (BASELINE configs[1] backbone)": E.config2(),
    "dinov2+siglip fused, 2 images, qwen2.5-0.5b (reference LIBERO default)": E.VLACfg(vit=[E.DINOV2_L_REG4, E.SIGLIP_SO400M], n_img=2),
}
if len(sys.argv) > 1:            # optional: index of the one configuration to run (profiling)
    configs = dict([list(configs.items())[int(sys.argv[1])]])
P = 48          # LIBERO prompt with the Qwen chat template (~48 ids)
for name, cfg in configs.items():
    W = S.make_weights(cfg, dev, seed=0)
    vla = OpenVLAForActionPrediction(cfg, W, dev, norm_stats=stats)
    g = torch.Generator().manual_seed(1)
    nch = 3 * len(cfg.vit) * cfg.n_img
    ids = torch.randint(0, 151000, (1, P), generator=g)
    px = torch.randn(1, nch, 224, 224, generator=g).clamp_(-3, 3).to(torch.bfloat16)
    am = torch.ones_like(ids, dtype=torch.bool)
    proprio = np.zeros(8, np.float32)
    call = lambda: vla.predict_action(input_ids=ids, proprio=proprio, proprio_projector=True, action_head=True, pixel_values=px,
                                      attention_mask=am)
    for _ in range(3):
        a, _ = call()
    torch.cuda.synchronize()
    n = 30
    t0 = time.perf_counter()
    for _ in range(n):
        a, _ = call()                      # ends with the D2H copy of the actions: fully synchronous per call
    t = (time.perf_counter() - t0) / n
    # device-side time of one replay of the captured segments (events on the caller's stream around the call)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    batch_static = next(iter(vla.engine._predict_graphs.values()))[1]
    e0.record()
    for _ in range(n):
        vla.engine.predict(batch_static)
    e1.record()
    torch.cuda.synchronize()
    tg = e0.elapsed_time(e1) / n
    S_len = cfg.n_patches + P + 65
    print(json.dumps({"config": name, "seq_len": S_len, "latency_ms_per_chunk_host_to_host": round(t * 1e3, 3),
                      "graph_replay_ms": round(tg, 3), "chunks_per_s": round(1 / t, 1), "actions_per_s": round(8 / t, 1),
                      "reference_published_s_per_chunk_H100": 0.036}), flush=True)
    del vla, W
    torch.cuda.empty_cache()
