"""Token cross-entropy path (prismatic/models/vlms/prismatic.py:312-481 -> HF shifted causal-LM loss; trainer:
prismatic/training/strategies/base_strategy.py:257-417): the forward at the REAL vocabulary (151 936) against oracle.token_ce, and
the backward (softmax - onehot formed in place on the label rows' logits, d hidden / d lm_head products, tied embedding table) of the
full fine-tune and LoRA trainers against autograd through the oracle.  PARITY UNPINNED by reference execution (the package needs the
absent draccus / timm): the oracle restates the text of the reference and of transformers' ForCausalLMLoss."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(__file__))

from oracle import vla_oracle as O  # noqa: E402
from test_engine_gpu import budget, budget_family, cpu_f32  # noqa: E402

DEV, BF = "cuda", torch.bfloat16


def _ce_batch(cfg, B, seed):
    from vla_adapter_amd import synthetic as S
    batch = S.make_batch(cfg, B, DEV, seed=seed, P=20, ragged=True)
    batch["labels"] = torch.where(batch["labels"] != -100, batch["input_ids"], batch["labels"])       # targets = the ids themselves
    return batch


def _oracle_ce(cfg, W, batch, emu, leaves=False):
    mk = (lambda d: {k: v.detach().float().cpu().clone().requires_grad_(True) for k, v in d.items()}) if leaves else cpu_f32
    vit, proj, llm = [mk(s) for s in W["vit"]], mk(W["proj"]), mk(W["llm"])
    px = batch["pixel_values"].float().cpu()
    nbk = len(cfg.vit)
    feats = []
    for im in range(cfg.n_img):
        ch = px[:, im * 3 * nbk:(im + 1) * 3 * nbk]
        feats.append(torch.cat([O.vit_forward(ch[:, 3 * j:3 * j + 3], vit[j], cfg.vit[j].as_oracle(), emu) for j in range(nbk)], dim=2))
    patches = O.projector(torch.cat(feats, dim=1), proj, cfg.fused, emu)
    ids, am = batch["input_ids"].cpu(), batch["attention_mask"].cpu()
    emb = llm["embed_tokens.weight"][ids]
    mm = torch.cat([emb[:, :1], patches, emb[:, 1:]], dim=1)
    mask = torch.cat([am[:, :1].bool(), torch.ones(ids.shape[0], patches.shape[1], dtype=torch.bool), am[:, 1:].bool()], dim=1)
    hs = O.qwen2_forward(mm, mask, llm, cfg.llm.as_oracle(), emu)
    loss, logits = O.token_ce(hs[-1], llm["embed_tokens.weight"], batch["labels"].cpu(), patches.shape[1], emu)
    return loss, logits, dict(vit=vit, proj=proj, llm=llm)


def test_token_ce_forward_at_the_real_vocabulary():
    """PrismaticVLM.forward's loss and logits with the Qwen2.5 vocabulary (151 936 ids incl. the 256 extra tokens and the x64
    padding, qwen25.py:40-85) on a 2-layer LLM: logits rows and the loss against oracle.token_ce under the fp32-truth budget."""
    from vla_adapter_amd import engine as E, synthetic as S
    cfg = E.tiny_config()
    cfg.llm = E.LLMCfg(256, 2, 4, 2, 64, 512, 1e-6, 1e6, 151936)
    W = S.make_weights(cfg, DEV, seed=11, std=0.05)
    batch = S.make_batch(cfg, 3, DEV, seed=12, P=20, ragged=True)
    assert int(batch["input_ids"].max()) > 151386, "the batch must reach into the action-token range of the real vocabulary"
    batch["labels"] = torch.where(batch["labels"] != -100, batch["input_ids"], batch["labels"])
    eng = E.VLAEngine(cfg, W, DEV)
    eng.forward_vlm(batch, action_queries=False)
    loss, logits = eng.token_ce(batch["labels"])
    torch.cuda.synchronize()
    res = {emu: _oracle_ce(cfg, W, batch, emu) for emu in (True, False)}
    Np = cfg.n_patches
    valid = torch.zeros(logits.shape[:2], dtype=torch.bool)
    valid[:, Np:-1] = batch["labels"][:, 1:].cpu() != -100
    budget(logits.float().cpu()[valid], res[True][1][valid], res[False][1][valid], "token-CE logits at V = 151936 (label rows)")
    le, lt = res[True][0].item(), res[False][0].item()
    print(f"token-CE loss: native {loss.item():.5f}  oracle(emu) {le:.5f}  fp32 {lt:.5f}")
    assert abs(loss.item() - lt) <= 1.25 * abs(le - lt) + 2e-3 * abs(lt)
    assert logits.shape == (3, batch["input_ids"].shape[1] + Np, 151936)


@pytest.mark.parametrize("which", ["tiny", "tiny_fused"])
def test_token_ce_full_finetune_gradients_match_oracle_autograd(which):
    """FullFinetune with the token-CE objective: gradients of the tied embedding table (lookup + lm_head contributions), the final
    norm, every LLM / ViT / projector matrix, against loss.backward() through the oracle."""
    from vla_adapter_amd import engine as E, synthetic as S
    from vla_adapter_amd.trainers import FullFinetune
    cfg = E.tiny_config() if which == "tiny" else E.tiny_fused_config()
    W = S.make_weights(cfg, DEV, seed=13, std=0.05)
    batch = _ce_batch(cfg, 3, 14)
    ft = FullFinetune(E.VLAEngine(cfg, W, DEV))
    ft.set_objective("token_ce")
    ft._run_inline(ft._segments(batch, None))
    torch.cuda.synchronize()
    res = {}
    for emu in (True, False):
        loss, _, OW = _oracle_ce(cfg, W, batch, emu, leaves=True)
        loss.backward()
        res[emu] = (loss.item(), OW)
    ln, le, lt = ft._loss3[0].item(), res[True][0], res[False][0]
    print(f"token-CE training loss: native {ln:.5f}  oracle(emu) {le:.5f}  fp32 {lt:.5f}")
    assert abs(ln - lt) <= 1.25 * abs(le - lt) + 2e-3 * abs(lt)
    got = ft.reference_named_gradients()
    names = ["vision_backbone.featurizer.", "vision_backbone.fused_featurizer."]
    ref = {}
    for emu in (True, False):
        d = {"language_model.model." + k: v.grad for k, v in res[emu][1]["llm"].items() if v.grad is not None}
        for pre, sd in zip(names, res[emu][1]["vit"]):
            d.update({pre + k: v.grad for k, v in sd.items() if v.grad is not None})
        d.update({"projector." + k: v.grad for k, v in res[emu][1]["proj"].items()})
        ref[emu] = d
    fam = [(k, g, ref[True][k].reshape(g.shape), ref[False][k].reshape(g.shape)) for k, g in got.items() if k in ref[False]]
    gmax = max(t[3].norm().item() for t in fam)
    emb = [t for t in fam if "embed_tokens" in t[0]][0]
    budget(emb[1], emb[2], emb[3], "token-CE: tied embedding table gradient (lookup + lm_head)", factor=1.5)
    budget_family([t for t in fam if t[1].dim() >= 2 and "embed_tokens" not in t[0]], "token-CE: weight-matrix gradients", absfloor=1e-3 * gmax)
    budget_family([t for t in fam if t[1].dim() < 2], "token-CE: vector gradients (biases, norms incl. the FINAL norm)", absfloor=1e-3 * gmax)
    assert any("model.norm.weight" in t[0] and t[3].abs().max().item() > 0 for t in fam), "the final norm is live under this objective"


@pytest.mark.parametrize("mode", ["full", "lora"])
def test_token_ce_trainers_train_eager_and_captured(mode):
    from vla_adapter_amd import engine as E, synthetic as S
    from vla_adapter_amd.trainers import FullFinetune, LoRAFinetune
    cfg = E.tiny_config()
    W = S.make_weights(cfg, DEV, seed=15, std=0.05)
    batch = _ce_batch(cfg, 4, 16)
    out = []
    for captured in (False, True):
        eng = E.VLAEngine(cfg, W, DEV)
        tr = FullFinetune(eng) if mode == "full" else LoRAFinetune(eng, rank=8, seed=2)
        tr.set_objective("token_ce")
        if captured:
            tr.capture({k: v.clone() for k, v in batch.items()}, None)
            ls = [tr.train_step_graphed(2e-3)[0].item() for _ in range(8)]
        else:
            ls = [tr.train_step(batch, 2e-3)[0].item() for _ in range(8)]
        torch.cuda.synchronize()
        assert all(v == v for v in ls) and ls[-1] < ls[0], ls
        out.append(ls)
    assert all(abs(a - b) <= 2e-2 * abs(a) for a, b in zip(*out)), out
