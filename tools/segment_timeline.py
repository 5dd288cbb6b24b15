"""Un-profiled timeline of the captured step: HIP timing events around every schedule segment (engine._run_segments)."""
import sys

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
eng.flush()
torch.cuda.synchronize()
ref = torch.cuda.Event(enable_timing=True)
ref.record()
eng._timeline = []
eng.train_step_graphed(5e-4)
end = torch.cuda.Event(enable_timing=True)
end.record()
eng.flush()
torch.cuda.synchronize()
print(f"step (adamw + segments, vision of the next step inside): {ref.elapsed_time(end):.3f} ms")
for st, k, t0, t1 in eng._timeline:
    seg = eng._segs[k] if k >= 0 else ("V", None, "after seg %d" % eng._vis_after, "vision of the next step")
    print(f"{st} seg {k:2d}  start {ref.elapsed_time(t0):7.3f}  end {ref.elapsed_time(t1):7.3f}  dur {t0.elapsed_time(t1):6.3f}  wait={seg[2]} signal={seg[3]}")
