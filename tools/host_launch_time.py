"""Host-side cost of enqueueing one captured step (graph launches, event records / waits, AdamW launches): must stay well
below the device time of a step, or the launcher - not the GPU - paces the job."""
import sys
import time

import torch

sys.path.insert(0, ".")
from vla_adapter_amd import engine as E, synthetic as S  # noqa: E402

dev = "cuda"
cfg = E.config2()
eng = E.VLAEngine(cfg, S.make_weights(cfg, dev, seed=0), dev)
batch = S.make_batch(cfg, 32, dev, seed=1000, P=32)
batch["pixel_values"] = batch["pixel_values"].to(torch.bfloat16)
noise = (torch.randn(cfg.chunk, cfg.action_dim * cfg.llm.d, device=dev) * 0.02).to(torch.bfloat16)
eng.capture(batch, noise)
for _ in range(3):
    eng.train_step_graphed(5e-4)
torch.cuda.synchronize()
host = []
for _ in range(10):
    torch.cuda.synchronize()                      # empty queues: the enqueue below never blocks on a full queue
    t0 = time.perf_counter()
    eng.train_step_graphed(5e-4)
    host.append((time.perf_counter() - t0) * 1e3)
torch.cuda.synchronize()
print(f"host enqueue time per step: median {sorted(host)[len(host) // 2]:.2f} ms, max {max(host):.2f} ms "
      f"({len(eng._segs)} segments + vision graph + 2 AdamW launches)")
