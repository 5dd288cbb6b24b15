"""Full fine-tune (BASELINE configs[3]: every VLM parameter trains, vla-scripts/finetune.py:846-849, 903-910): gradients of
one tensor of every kind against autograd through the CPU oracle, under the fp32-truth error budget of
tests/test_engine_gpu.py (all three backward passes driven by the engine's own upstream gradient)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import vla_oracle as O  # noqa: E402
import os, sys
sys.path.insert(0, os.path.dirname(__file__))
from test_engine_gpu import budget, budget_family, cpu_f32, oracle_cfg, rel  # noqa: E402

DEV, BF = "cuda", torch.bfloat16


def _oracle_full(cfg, W, batch, emu):
    leaf = lambda d: {k: v.detach().float().cpu().clone().requires_grad_(True) for k, v in d.items()}
    llm = leaf(W["llm"])
    OW = dict(vit=[leaf(s) for s in W["vit"]], proj=leaf(W["proj"]), llm=llm, embed=llm["embed_tokens.weight"],
              action_queries=W["action_queries"].float().cpu().clone().requires_grad_(True), head=leaf(W["head"]), proprio=leaf(W["proprio"]))
    cb = {k: v.cpu() for k, v in batch.items()}
    cb["pixel_values"], cb["proprio"] = cb["pixel_values"].float(), cb["proprio"].to(BF).float()
    return O.vla_forward(cb, OW, oracle_cfg(cfg), emu=emu, noise=None), OW


def _ref_grads(OW, cfg):
    """Oracle gradients under the reference's state-dict names (what FullFinetune.reference_named_gradients returns)."""
    out = {}
    for k, v in OW["llm"].items():
        if v.grad is not None:
            out["language_model.model." + k] = v.grad
    for pre, sd in zip(("vision_backbone.featurizer.", "vision_backbone.fused_featurizer."), OW["vit"]):
        for k, v in sd.items():
            if v.grad is not None:
                out[pre + k] = v.grad
    for k, v in OW["proj"].items():
        out["projector." + k] = v.grad
    return out


@pytest.mark.parametrize("which", ["tiny", "tiny_fused"])
def test_full_finetune_gradients_match_oracle_autograd(which):
    """tiny: BASELINE configs[0].  tiny_fused: the reference's default setup at plumbing size - a DINOv2-like backbone (cls + 4
    register tokens, LayerScale as its own parameter) fused with a SigLIP-like one, two images per sample, 3-layer projector
    (modeling_prismatic.py:58-66, 196-237): gradients of ls1 / ls2.scale_factor, cls_token, reg_token and both pos_embeds included."""
    from vla_adapter_amd import engine as E, synthetic as S, ops
    from vla_adapter_amd.full_finetune import FullFinetune
    cfg = E.tiny_config() if which == "tiny" else E.tiny_fused_config()
    W = S.make_weights(cfg, DEV, seed=3, std=0.05)
    batch = S.make_batch(cfg, 3, DEV, seed=4, P=20, ragged=True)
    eng = E.VLAEngine(cfg, W, DEV)
    ft = FullFinetune(eng)
    pred = ft.forward(batch, None)
    _, dpred = ops.l1_loss(pred, batch["actions"].to(BF), True)
    ft.backward(pred, batch["actions"])
    torch.cuda.synchronize()
    G = {}
    for emu in (True, False):
        out, OW = _oracle_full(cfg, W, batch, emu)
        if emu:
            budget_pred = (pred, out["pred"])
        out["pred"].backward(dpred.float().cpu())
        G[emu] = (_ref_grads(OW, cfg), OW)
        if not emu:
            budget(budget_pred[0], budget_pred[1], out["pred"], "full fine-tune forward (unfused GELU, saved activations): actions")
    got = ft.reference_named_gradients()
    n, nbv = cfg.llm.n_layers, sum(len(v.blocks) for v in eng.vits)
    if which == "tiny_fused":
        for need in ("ls1.scale_factor", "ls2.scale_factor", "cls_token", "reg_token", "fused_featurizer.pos_embed", "projector.fc3.weight"):
            assert any(need in k for k in got), need
    fam = []
    skipped = []
    for k, g in got.items():
        if k not in G[False][0]:
            skipped.append(k)
            continue
        r_e, r_t = G[True][0][k], G[False][0][k]
        fam.append((k, g, r_e.reshape(g.shape), r_t.reshape(g.shape)))
    # the oracle computes the (dead) last ViT block too; the engine's parameter list stops at the last useful block
    assert all("featurizer.blocks.%d." % (cfg.vit[0].depth - 1) not in k for k in got), "last ViT block is never part of the path"
    assert len(fam) >= 8 * nbv + 10 * n + 6, (len(fam), skipped[:5])
    gmax = max(t[3].norm().item() for t in fam)
    mats = [t for t in fam if t[1].dim() >= 2 and "embed_tokens" not in t[0]]
    vecs = [t for t in fam if t[1].dim() < 2]
    budget_family(mats, "full fine-tune: weight-matrix gradients (ViT, projector, LLM)", absfloor=1e-3 * gmax)
    budget_family(vecs, "full fine-tune: bias / LayerNorm / RMSNorm gradients", absfloor=1e-3 * gmax)
    emb = [t for t in fam if "embed_tokens" in t[0]][0]
    budget(emb[1], emb[2], emb[3], "full fine-tune: embedding-table gradient", factor=1.5)
    touched = (emb[3].abs().sum(1) > 0)
    assert torch.equal((emb[1].float().cpu().abs().sum(1) > 0), touched), "exactly the rows of the tokens in the batch receive a gradient"
    budget(eng.head.P.g("action_queries"), G[True][1]["action_queries"].grad, G[False][1]["action_queries"].grad, "full fine-tune: action_queries", factor=1.5)
    hk = "model.mlp_resnet_blocks.0.k_task.weight"
    budget(eng.head.named_views(eng.head.P.grad)[hk], G[True][1]["head"][hk].grad, G[False][1]["head"][hk].grad, "full fine-tune: head k_task", factor=1.5)


def test_full_finetune_trains_and_updates_every_tensor():
    from vla_adapter_amd import engine as E, synthetic as S
    from vla_adapter_amd.full_finetune import FullFinetune
    cfg = E.tiny_config()
    W = S.make_weights(cfg, DEV, seed=5, std=0.05)
    batch = S.make_batch(cfg, 4, DEV, seed=6, P=24, ragged=True)
    ft = FullFinetune(E.VLAEngine(cfg, W, DEV))
    p0 = ft.P.data.clone()
    losses = [ft.train_step(batch, 3e-4)[0].item() for _ in range(10)]
    torch.cuda.synchronize()
    assert all(l == l for l in losses) and losses[-1] < 0.8 * losses[0], losses
    # every tensor received a gradient (first AdamW moment non-zero) and every tensor that CAN move in bf16 did: LayerNorm / RMSNorm
    # weights sit at ~1.0, where one bf16 ulp (7.8e-3) is far above lr = 3e-4 - they stay put in the reference's bf16 AdamW too
    fed, moved = {}, {}
    for s in ft.slots:
        off, shape = ft.P.offsets[s.name]
        n = 1
        for d in shape:
            n *= d
        fed[s.name] = bool((ft.P.m[off:off + n] != 0).any())
        moved[s.name] = bool((ft.P.data[off:off + n] != p0[off:off + n]).any())
    assert all(fed.values()), [k for k, v in fed.items() if not v][:10]
    still = [k for k, v in moved.items() if not v]
    assert all(k.split(".")[-1] in ("n1w", "n2w", "n1", "n2", "norm") for k in still), still[:10]
    # the fused operands the kernels read ARE the parameters: the engine's weight handles alias the flat buffer
    assert ft.llm.layers[0]["wqkv"].data_ptr() == ft.P.view("llm.0.wqkv").data_ptr()
    # zero-padded regions stay zero (zero weight, zero gradient under AdamW)
    v = ft.vit
    if v.mlp_pad != v.cfg.mlp:
        assert bool((v.blocks[0]["w1"][v.cfg.mlp:] == 0).all()) and bool((v.blocks[0]["w2"][:, v.cfg.mlp:] == 0).all())
    assert bool((v.wpe[:, 3 * v.cfg.patch ** 2:] == 0).all())


@pytest.mark.parametrize("mode", ["full", "lora"])
def test_llm_layers_above_the_heads_last_block_are_left_alone(mode):
    """Qwen2.5-1.5B has 28 layers, the head 24 blocks (action_heads.py:117-118 reads hidden_states[1..24]): layers 25-28 and the
    final norm never reach the loss.  The reference computes them anyway; autograd hands their parameters ALL-ZERO gradients (zeros
    travel back through torch.cat / index_select of finetune.py:396-409), so AdamW moves them by the weight-decay factor 1 - lr wd
    alone - which in bf16 rounds to 1 for lr wd < 2^-9: they never change.  Here (3 layers, 2 blocks) they are neither computed
    nor touched: the gradients of the live layers still match the oracle's autograd through ALL layers, and after training steps
    every parameter of the dead layer is bit-identical to its initial value."""
    from vla_adapter_amd import engine as E, synthetic as S, ops
    from vla_adapter_amd.trainers import FullFinetune, LoRAFinetune
    cfg = E.tiny_config()
    cfg.llm = E.LLMCfg(256, 3, 4, 2, 64, 512, 1e-6, 1e6, 1024)
    cfg.num_blocks = 2
    W = S.make_weights(cfg, DEV, seed=3, std=0.05)
    batch = S.make_batch(cfg, 3, DEV, seed=4, P=20, ragged=True)
    eng = E.VLAEngine(cfg, W, DEV)
    tr = FullFinetune(eng) if mode == "full" else LoRAFinetune(eng, rank=8, seed=1)
    assert tr.n_active == 2
    if mode == "full":
        pred = tr.forward(batch, None)
        _, dpred = ops.l1_loss(pred, batch["actions"].to(BF), True)
        tr.backward(pred, batch["actions"])
        torch.cuda.synchronize()
        res = {}
        for emu in (True, False):
            out, OW = _oracle_full(cfg, W, batch, emu)
            out["pred"].backward(dpred.float().cpu())
            res[emu] = (_ref_grads(OW, cfg), OW)
        gdead = res[False][1]["llm"]["layers.2.mlp.down_proj.weight"].grad
        assert gdead is None or gdead.abs().max().item() == 0.0, "autograd gives the dead layer an all-zero gradient"
        got = tr.reference_named_gradients()
        fam = [(k, got[k], res[True][0][k].reshape(got[k].shape), res[False][0][k].reshape(got[k].shape)) for k in got
               if k in res[False][0] and "language_model.model.layers." in k and "layers.2." not in k and got[k].dim() >= 2]
        assert len(fam) == 2 * 7
        assert all(got[k].float().abs().max().item() == 0.0 for k in got if "layers.2." in k), "the dead layer's gradient is exactly zero"
        budget_family(fam, "live LLM layers under a dead one: weight-matrix gradients", absfloor=1e-3 * max(t[3].norm().item() for t in fam))
    p0 = tr.P.data.clone()
    losses = [tr.train_step(batch, 1e-3)[0].item() for _ in range(4)]
    torch.cuda.synchronize()
    assert all(l == l for l in losses)
    dead = [k for k in tr.P.offsets if ("llm.2." in k if mode == "full" else "layers.2." in k)] + (["llm.norm"] if mode == "full" else [])
    live = [k for k in tr.P.offsets if ("llm.1." in k if mode == "full" else "layers.1." in k)]
    assert dead and live
    for k in dead:
        off, shape = tr.P.offsets[k]
        n = 1
        for d in shape:
            n *= d
        assert torch.equal(tr.P.data[off:off + n], p0[off:off + n]), f"{k} must be left alone (no gradient, no weight decay)"
    moved = 0
    for k in live:
        off, shape = tr.P.offsets[k]
        n = 1
        for d in shape:
            n *= d
        moved += int(not torch.equal(tr.P.data[off:off + n], p0[off:off + n]))
    assert moved >= len(live) // 2, "the live layer below must train"


def test_overlapped_and_plain_updates_cover_the_same_parameters(monkeypatch):
    """the range-by-range update under the backward (_update_ranges over the 'tail' range) and the single update at the
    end (_adam_ranges) must step the SAME parameters when LLM layers lie above the head's last block.  lr is chosen so large
    that the weight-decay factor 1 - lr wd is visible in bf16 (0.995): a dead norm / bias / final norm that receives AdamW
    moves, one that is skipped keeps its bits."""
    from vla_adapter_amd import engine as E, synthetic as S
    from vla_adapter_amd.trainers import FullFinetune
    cfg = E.tiny_config()
    cfg.llm = E.LLMCfg(256, 3, 4, 2, 64, 512, 1e-6, 1e6, 1024)
    cfg.num_blocks = 2
    batch = S.make_batch(cfg, 3, DEV, seed=4, P=20, ragged=True)
    out = {}
    for overlap in (True, False):
        W = S.make_weights(cfg, DEV, seed=3, std=0.05)
        tr = FullFinetune(E.VLAEngine(cfg, W, DEV))
        tr.overlap_update = overlap
        p0 = tr.P.data.clone()
        tr.train_step(batch, 0.5)
        torch.cuda.synchronize()
        out[overlap] = tr.P.data.clone()
        for k in [k for k in tr.P.offsets if "llm.2." in k] + ["llm.norm"]:
            off, shape = tr.P.offsets[k]
            n = 1
            for d in shape:
                n *= d
            assert torch.equal(tr.P.data[off:off + n], p0[off:off + n]), f"{k} (dead) moved with overlap_update={overlap}"
    assert torch.equal(out[True], out[False]), "the two update paths are the same arithmetic on the same parameters"


@pytest.mark.parametrize("mode", ["full", "lora"])
def test_second_backbone_on_its_own_stream_equals_the_serial_schedule(mode, monkeypatch):
    """Round 4: with two vision backbones (DINOv2 + SigLIP) the second one's forward and backward run on a stream of their own ("V"
    segments).  Same kernels on the same data, only the streams differ: after three steps - eager three-stream schedule and captured -
    every parameter and the losses equal the one-backbone-after-the-other schedule (VLA_SERIAL_BACKBONES=1) bit for bit.  A race between
    the two chains (a shared scratch buffer, a missing event) would show up here as a difference."""
    from vla_adapter_amd import engine as E, synthetic as S
    from vla_adapter_amd.trainers import FullFinetune, LoRAFinetune
    cfg = E.tiny_fused_config()
    cfg.n_img = 2
    batch = S.make_batch(cfg, 3, DEV, seed=14, P=20, ragged=True)
    res = {}
    for serial in (False, True):
        if serial:
            monkeypatch.setenv("VLA_SERIAL_BACKBONES", "1")
        else:
            monkeypatch.delenv("VLA_SERIAL_BACKBONES", raising=False)
        W = S.make_weights(cfg, DEV, seed=5, std=0.05)
        eng = E.VLAEngine(cfg, W, DEV)
        tr = FullFinetune(eng) if mode == "full" else LoRAFinetune(eng, rank=16, seed=2)
        assert (tr.vstream is None) == serial
        losses = [tr.train_step(batch, 1e-3)[0].item() for _ in range(3)]
        tr.capture(batch, None)
        for _ in range(3):
            tr.train_step_graphed(1e-3)
            losses.append(tr._loss3[0].item())
        torch.cuda.synchronize()
        res[serial] = (tr.P.data.clone(), tr.head.P.data.clone(), losses)
    assert res[False][2] == res[True][2], (res[False][2], res[True][2])
    assert torch.equal(res[False][0], res[True][0]) and torch.equal(res[False][1], res[True][1])
