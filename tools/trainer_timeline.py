"""Un-profiled timeline of the captured full / LoRA fine-tune step (trainers.BackboneTrainer._run): HIP timing events around every
schedule segment on its stream, plus the step's end.  usage: trainer_timeline.py [lora|full] [batch] [config2|config5]"""
import sys

import torch

sys.path.insert(0, ".")
from vla_adapter_amd import engine as E, synthetic as S  # noqa: E402
from vla_adapter_amd.trainers import FullFinetune, LoRAFinetune  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "lora"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
dev = "cuda"
cfg = E.config2()
if len(sys.argv) > 3 and sys.argv[3] == "config5":      # BASELINE configs[4]: DINOv2 + SigLIP, two images, Qwen2.5-1.5B
    cfg = E.NAMED_CONFIGS["config5"]()
    cfg.n_img = 2
eng = E.VLAEngine(cfg, S.make_weights(cfg, dev, seed=0), dev)
batch = S.make_batch(cfg, B, dev, seed=1000, P=32)
batch["pixel_values"] = batch["pixel_values"].to(torch.bfloat16)
noise = (torch.randn(cfg.chunk, cfg.action_dim * cfg.llm.d, device=dev) * 0.02).to(torch.bfloat16)
ft = LoRAFinetune(eng, rank=64) if mode == "lora" else FullFinetune(eng)
ft.capture(batch, noise)
for _ in range(3):
    ft.train_step_graphed(5e-4)
torch.cuda.synchronize()
ref = torch.cuda.Event(enable_timing=True)
ref.record()
ft._timeline = []
ft.train_step_graphed(5e-4)
end = torch.cuda.Event(enable_timing=True)
end.record()
torch.cuda.synchronize()
print(f"{mode} step, batch {B}: {ref.elapsed_time(end):.3f} ms")
for st, k, t0, t1 in ft._timeline:
    seg = ft._segs[k]
    print(f"{st} seg {k:2d}  start {ref.elapsed_time(t0):7.3f}  end {ref.elapsed_time(t1):7.3f}  dur {t0.elapsed_time(t1):6.3f}  wait={seg[2]} signal={seg[3]} ranges={len(seg[4]) if seg[4] else 0}")
