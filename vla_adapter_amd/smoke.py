"""One tiny invocation of the hot path on cuda:0, checked against the CPU oracle (used by __graft_entry__.smoke)."""
import os
import sys

import torch


def run():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    from oracle import vla_oracle as O          # checker only
    from . import engine as E, synthetic as S
    cfg = E.tiny_config()
    W = S.make_weights(cfg, "cuda", seed=1, std=0.05)
    batch = S.make_batch(cfg, 2, "cuda", seed=2, P=24, ragged=True)
    eng = E.VLAEngine(cfg, W, "cuda")
    pred = eng.forward(batch, None)
    loss3 = eng.loss_and_backward(pred, batch["actions"])
    eng.optimizer_step(5e-4)
    torch.cuda.synchronize()
    f = lambda sd: {k: v.float().cpu() for k, v in sd.items()}
    llm = f(W["llm"])
    OW = dict(vit=[f(s) for s in W["vit"]], proj=f(W["proj"]), llm=llm, embed=llm["embed_tokens.weight"],
              action_queries=W["action_queries"].float().cpu(), head=f(W["head"]), proprio=f(W["proprio"]))
    cb = {k: v.cpu() for k, v in batch.items()}
    cb["pixel_values"], cb["proprio"] = cb["pixel_values"].float(), cb["proprio"].bfloat16().float()
    ocfg = dict(vit=[v.as_oracle() for v in cfg.vit], fused=cfg.fused, llm=cfg.llm.as_oracle(), n_img=cfg.n_img, pro=True,
                num_blocks=cfg.num_blocks)
    out = O.vla_forward(cb, OW, ocfg, emu=True)
    rel = ((pred.float().cpu() - out["pred"]).norm() / out["pred"].norm()).item()
    dl = abs(loss3[0].item() - out["loss"].item()) / abs(out["loss"].item())
    print(f"smoke: pred rel-L2 vs oracle {rel:.3e}, loss {loss3[0].item():.5f} vs {out['loss'].item():.5f} (rel {dl:.2e})")
    assert rel < 2e-2 and dl < 1e-2, "native hot path disagrees with the oracle"
    print("smoke OK")
