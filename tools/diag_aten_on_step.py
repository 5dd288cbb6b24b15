#!/usr/bin/env python3
"""Which torch (ATen) operators still launch a GPU kernel inside one eager training step - with the Python call site."""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vla_adapter_amd import engine as E, synthetic as S  # noqa: E402


def main():
    dev = "cuda"
    cfg = E.config2()
    W = S.make_weights(cfg, dev, seed=0)
    eng = E.VLAEngine(cfg, W, dev)
    batch = S.make_batch(cfg, 8, dev, seed=1, P=32, ragged=False)
    batch["pixel_values"] = batch["pixel_values"].to(torch.bfloat16)
    noise = (torch.randn(cfg.chunk, cfg.action_dim * cfg.llm.d, device=dev) * 0.02).to(torch.bfloat16)
    for _ in range(2):
        eng.train_step(batch, 5e-4, noise)
    torch.cuda.synchronize()
    import traceback
    orig_copy, orig_contig = torch.Tensor.copy_, torch.Tensor.contiguous

    def where():
        return " | ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in traceback.extract_stack()[:-2] if "vla_adapter_amd" in f.filename)[-160:]

    def copy_(self, src, *a, **k):
        print("copy_", tuple(self.shape), self.dtype, "<-", tuple(src.shape), src.dtype, where())
        return orig_copy(self, src, *a, **k)

    def contiguous(self, *a, **k):
        if not self.is_contiguous():
            print("contiguous() copies", tuple(self.shape), self.stride(), where())
        return orig_contig(self, *a, **k)

    torch.Tensor.copy_, torch.Tensor.contiguous = copy_, contiguous
    eng.train_step(batch, 5e-4, noise)
    torch.cuda.synchronize()
    torch.Tensor.copy_, torch.Tensor.contiguous = orig_copy, orig_contig
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
        eng.train_step(batch, 5e-4, noise)
        torch.cuda.synchronize()
    for avg in sorted(prof.key_averages(group_by_input_shape=True, group_by_stack_n=6), key=lambda a: -a.count):
        dt = getattr(avg, "self_device_time_total", getattr(avg, "self_cuda_time_total", 0))
        if avg.key.startswith("aten::") and dt > 0:
            st = [x for x in (avg.stack or []) if "vla_adapter_amd" in x or "tools/" in x][:2]
            print(avg.count, avg.key, str(avg.input_shapes)[:90], f"{dt:.0f}us", " <- ", " | ".join(st))


if __name__ == "__main__":
    main()
