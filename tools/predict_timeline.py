"""Un-profiled segment timeline of the captured batch-1 forward (engine.predict) for the reference's LIBERO layout."""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from vla_adapter_amd import engine as E, synthetic as S  # noqa: E402
from vla_adapter_amd.modeling_prismatic import OpenVLAForActionPrediction  # noqa: E402

dev = "cuda"
cfg = E.VLACfg(vit=[E.DINOV2_L_REG4, E.SIGLIP_SO400M], n_img=2) if len(sys.argv) < 2 else E.config2()
vla = OpenVLAForActionPrediction(cfg, S.make_weights(cfg, dev, seed=0), dev)
ids = torch.randint(0, 151000, (1, 48))
px = torch.randn(1, 3 * len(cfg.vit) * cfg.n_img, 224, 224).clamp_(-3, 3).to(torch.bfloat16)
i2, am, lab = vla.prepare_inference_inputs(ids.to(dev), torch.ones_like(ids, dtype=torch.bool).to(dev))
batch = dict(input_ids=i2, labels=lab, attention_mask=am.bool(), pixel_values=px.to(dev), proprio=torch.zeros(1, 8, device=dev))
eng = vla.engine
for _ in range(3):
    eng.predict(batch)
torch.cuda.synchronize()
ref = torch.cuda.Event(enable_timing=True)
ref.record()
eng._timeline = []
eng.predict(batch)
end = torch.cuda.Event(enable_timing=True)
end.record()
torch.cuda.synchronize()
print(f"predict: {ref.elapsed_time(end):.3f} ms")
segs = next(iter(eng._predict_graphs.values()))[2]
for st, k, t0, t1 in eng._timeline:
    print(f"{st:3s} seg {k:2d} start {ref.elapsed_time(t0):7.3f} end {ref.elapsed_time(t1):7.3f} dur {t0.elapsed_time(t1):6.3f} wait={segs[k][2]} signal={segs[k][3]}")
