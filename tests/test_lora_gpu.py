"""LoRA fine-tune (vla-scripts/finetune.py:832-844) against autograd through the oracle with peft's Linear semantics
(oracle.LORA registry): forward and the gradients of the A / B pairs of every kind of target, under the fp32-truth budget.
PARITY UNPINNED (peft is not importable here)."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(__file__))

from oracle import vla_oracle as O  # noqa: E402
from test_engine_gpu import budget, budget_family, oracle_cfg  # noqa: E402

DEV, BF = "cuda", torch.bfloat16


def _oracle_lora(cfg, W, batch, lo, emu):
    """lo: LoRAFinetune.  Builds oracle weights (frozen) + LoRA leaves registered on the matching base tensors."""
    f = lambda d: {k: v.detach().float().cpu().clone() for k, v in d.items()}
    llm = f(W["llm"])
    leaf = lambda d: {k: v.requires_grad_(True) for k, v in d.items()}
    OW = dict(vit=[f(s) for s in W["vit"]], proj=f(W["proj"]), llm=llm, embed=llm["embed_tokens.weight"],
              action_queries=W["action_queries"].float().cpu().clone().requires_grad_(True), head=leaf(f(W["head"])), proprio=leaf(f(W["proprio"])))
    sd = {k: v.detach().float().cpu().clone().requires_grad_(True) for k, v in lo.lora_state_dict().items()}
    O.LORA.clear()
    pre = "base_model.model."

    def reg(base, key):
        O.LORA[id(base)] = (sd[pre + key + ".lora_A.weight"], sd[pre + key + ".lora_B.weight"], 2.0)
    for j, (vn, v) in enumerate(zip(("featurizer", "fused_featurizer"), lo.vits)):
        for i in range(len(v.blocks)):
            for n in ("attn.qkv", "attn.proj", "mlp.fc1", "mlp.fc2"):
                reg(OW["vit"][j][f"blocks.{i}.{n}.weight"], f"vision_backbone.{vn}.blocks.{i}.{n}")
    for k in OW["proj"]:
        if k.endswith("weight"):
            reg(OW["proj"][k], "projector." + k[:-7])
    for i in range(cfg.llm.n_layers):
        for n in ("self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj", "self_attn.o_proj", "mlp.gate_proj", "mlp.up_proj", "mlp.down_proj"):
            reg(llm[f"layers.{i}.{n}.weight"], f"language_model.model.layers.{i}.{n}")
    cb = {k: v.cpu() for k, v in batch.items()}
    cb["pixel_values"], cb["proprio"] = cb["pixel_values"].float(), cb["proprio"].to(BF).float()
    O.LORA_FUSED = True          # the native evaluation: low-rank branch inside the base product's accumulator (trainers.py)
    try:
        out = O.vla_forward(cb, OW, oracle_cfg(cfg), emu=emu, noise=None)
    finally:
        O.LORA.clear()
        O.LORA_FUSED = False
    return out, OW, sd


def _cfg(which):
    from vla_adapter_amd import engine as E
    if which == "tiny":
        return E.tiny_config()
    if which == "tiny_fused":          # DINOv2-like (prefix tokens, LayerScale) + SigLIP-like, two images: the reference's default recipe
        return E.tiny_fused_config()
    if which == "padded_mlp":          # ViT MLP width that is not a multiple of 128 (SigLIP so400m: 4304 -> 4352)
        c = E.tiny_config()
        c.vit = [E.ViTCfg(192, 3, 3, 760, 14, 56, 0, False)]
        return c
    return E.qwen15b_geometry_config(2)   # "qwen15b": head dim 128 (unfused RoPE), d 1536, MLP 8960 - BASELINE configs[4]'s layer


@pytest.mark.parametrize("which", ["tiny", "tiny_fused", "padded_mlp", "qwen15b"])
def test_lora_forward_and_gradients_match_oracle_autograd(which):
    from vla_adapter_amd import engine as E, synthetic as S, ops
    from vla_adapter_amd.lora_finetune import LoRAFinetune
    cfg = _cfg(which)
    W = S.make_weights(cfg, DEV, seed=3, std=0.03 if which == "qwen15b" else 0.05)
    batch = S.make_batch(cfg, 3, DEV, seed=4, P=20, ragged=True)
    eng = E.VLAEngine(cfg, W, DEV)
    lo = LoRAFinetune(eng, rank=8, seed=1)
    # peft starts with B = 0 (the wrapped model equals the base model): give B a value so that both branches carry signal
    g = torch.Generator(device=DEV).manual_seed(2)
    for l in lo.L.values():
        for p, _ in l.projs:
            Bv = lo.P.view(f"{l.name}.{p}.lora_B")
            Bv[:, :l.r] = (torch.randn(Bv.shape[0], l.r, generator=g, device=DEV) * 0.05).to(BF)
    # the ViT MLP's width padding: B rows of fc1 beyond the true width hold no parameter (init_ leaves A's padding columns zero)
    for j, v in enumerate(lo.vits):
        for i in range(len(v.blocks)):
            lo.P.view(f"{lo.L[f'vit{j}.{i}.fc1'].name}.fc1.lora_B")[v.cfg.mlp:] = 0
            assert bool((lo.P.view(f"{lo.L[f'vit{j}.{i}.fc2'].name}.fc2.lora_A")[:, v.cfg.mlp:] == 0).all())
    lo.refresh()
    pred = lo.forward(batch, None)
    n = cfg.llm.n_layers
    hs_native = {i: eng.llm.HS[i].clone() for i in (0, n)}
    _, dpred = ops.l1_loss(pred, batch["actions"].to(BF), True)
    lo.backward(pred, batch["actions"])
    torch.cuda.synchronize()
    pred = pred.clone()
    base_pred = eng.forward(batch, None)
    assert (pred.float() - base_pred.float()).abs().max().item() > 1e-3, "the LoRA branches must change the output"
    res = {}
    for emu in (True, False):
        out, OW, sd = _oracle_lora(cfg, W, batch, lo, emu)
        out["pred"].backward(dpred.float().cpu())
        res[emu] = (out, OW, sd)
    budget(pred, res[True][0]["pred"], res[False][0]["pred"], "LoRA forward: actions")
    for i in (0, n):
        budget(hs_native[i], res[True][0]["hidden_states"][i], res[False][0]["hidden_states"][i], f"LoRA forward: hidden_states[{i}]")
    gsd = {}
    for l in lo.L.values():
        for p, _ in l.projs:
            gA, gB = lo.P.g(f"{l.name}.{p}.lora_A"), lo.P.g(f"{l.name}.{p}.lora_B")
            # every padded entry carries NO gradient: rank padding, the ViT MLP's width padding (rows of fc1's B, columns of fc2's A)
            assert bool((gA[l.r:] == 0).all()) and bool((gA[:, l.k_real:] == 0).all()), f"{l.name}.{p}: gradient on A's padding"
            assert bool((gB[:, l.r:] == 0).all()) and bool((gB[l.n_real:] == 0).all()), f"{l.name}.{p}: gradient on B's padding"
            gsd[f"{l.name}.{p}.lora_A.weight"], gsd[f"{l.name}.{p}.lora_B.weight"] = gA[:l.r, :l.k_real], gB[:l.n_real, :l.r]
    famA = [(k, t, res[True][2][k].grad, res[False][2][k].grad) for k, t in gsd.items() if "lora_A" in k]
    famB = [(k, t, res[True][2][k].grad, res[False][2][k].grad) for k, t in gsd.items() if "lora_B" in k]
    assert len(famA) == 4 * sum(len(v.blocks) for v in lo.vits) + (3 if cfg.fused else 2) + 7 * n
    gmax = max(t[3].norm().item() for t in famA + famB)
    budget_family(famA, "LoRA A gradients (ViT / projector / LLM)", absfloor=1e-3 * gmax)
    budget_family(famB, "LoRA B gradients (ViT / projector / LLM)", absfloor=1e-3 * gmax)
    if which in ("tiny", "tiny_fused"):      # end-to-end smoke through the whole adapted stack (the single-layer checks of
        # tests/test_layer_gradients_gpu.py are the backward's gate; the padded / 1.5B variants are here for their layouts)
        budget(eng.head.P.g("action_queries"), res[True][1]["action_queries"].grad, res[False][1]["action_queries"].grad, "LoRA: action_queries", factor=1.5)



@pytest.mark.parametrize("which", ["tiny", "padded_mlp", "tiny_fused"])
def test_lora_training_moves_only_the_adapters_and_merges(which):
    from vla_adapter_amd import engine as E, synthetic as S
    from vla_adapter_amd.lora_finetune import LoRAFinetune
    cfg = _cfg(which)
    W = S.make_weights(cfg, DEV, seed=5, std=0.05)
    batch = S.make_batch(cfg, 4, DEV, seed=6, P=24, ragged=True)
    eng = E.VLAEngine(cfg, W, DEV)
    lo = LoRAFinetune(eng, rank=8, seed=1)
    w0 = eng.llm.layers[0]["wqkv"].clone()
    pred0 = lo.forward(batch, None).clone()
    base = eng.forward(batch, None)
    assert torch.equal(pred0, base), "B = 0 at initialisation: the wrapped model IS the base model (peft init_lora_weights)"
    losses = [lo.train_step(batch, 1e-3)[0].item() for _ in range(12)]
    torch.cuda.synchronize()
    assert losses[-1] < 0.8 * losses[0], losses
    assert torch.equal(w0, eng.llm.layers[0]["wqkv"]), "base weights are frozen"
    sd = lo.lora_state_dict()
    assert any(v.abs().max().item() > 0 for k, v in sd.items() if "lora_B" in k), "the B matrices must have left zero"
    # width / rank paddings never receive a parameter value (the saved adapter must be the function that was trained)
    for l in lo.L.values():
        for p, _ in l.projs:
            A, Bm = lo.P.view(f"{l.name}.{p}.lora_A"), lo.P.view(f"{l.name}.{p}.lora_B")
            assert bool((A[l.r:] == 0).all()) and bool((A[:, l.k_real:] == 0).all()) and bool((Bm[:, l.r:] == 0).all()) and bool((Bm[l.n_real:] == 0).all()), l.name
    # save -> load round trip of the adapter (lora_adapter/adapter_model.safetensors layout)
    lo2 = LoRAFinetune(E.VLAEngine(cfg, W, DEV), rank=8, seed=99)
    lo2.load_lora_state_dict({k: v.clone().cpu() for k, v in sd.items()})
    assert torch.equal(lo2.P.data, lo.P.data)
    k = "base_model.model.language_model.model.layers.0.self_attn.q_proj.lora_A.weight"
    assert k in sd and tuple(sd[k].shape) == (8, cfg.llm.d)
    # merging the adapter into the base (finetune.py:579-601) reproduces the adapted forward
    merged = lo.merged_weights()
    pred_l = lo.forward(batch, None).clone()
    for key, wm in merged.items():
        holder, wk = lo._base(key)
        holder[wk].copy_(wm)
    pred_m = eng.forward(batch, None)
    from test_engine_gpu import rel
    assert rel(pred_m, pred_l) < 2e-2, rel(pred_m, pred_l)


def test_lora_captured_step_matches_eager_steps():
    """hipGraph replay of the LoRA step (LoRAFinetune.capture / train_step_graphed) == the same steps launched eagerly: losses
    and every adapter / head parameter bit-identical after three updates."""
    from vla_adapter_amd import engine as E, synthetic as S
    from vla_adapter_amd.lora_finetune import LoRAFinetune
    cfg = E.tiny_config()
    batch = S.make_batch(cfg, 4, DEV, seed=16, P=24, ragged=True)
    res = []
    for graphed in (False, True):
        W = S.make_weights(cfg, DEV, seed=15, std=0.05)
        eng = E.VLAEngine(cfg, W, DEV)
        lo = LoRAFinetune(eng, rank=8, seed=2)
        if graphed:
            state = [t.clone() for t in (lo.P.data, eng.head.P.data)]
            lo.capture(batch, None)                 # (its warm-up passes update nothing: no optimiser step inside)
            assert all(torch.equal(a, b) for a, b in zip(state, (lo.P.data, eng.head.P.data)))
            losses = [lo.train_step_graphed(1e-3)[0].item() for _ in range(3)]
        else:
            losses = [lo.train_step(batch, 1e-3)[0].item() for _ in range(3)]
        torch.cuda.synchronize()
        res.append((losses, lo.P.data.clone(), eng.head.P.data.clone()))
    assert res[0][0] == res[1][0], (res[0][0], res[1][0])
    assert torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][2], res[1][2])


def test_finetune_entry_point_lora_dropout(tmp_path):
    """`vla-scripts/finetune.py --use_lora True --lora_dropout 0.05` (finetune.py:110) on the plumbing-size dual config, captured step:
    completes, the loss falls, the mask counter advanced once per step (fresh masks on every graph replay), adapter_config.json carries
    the value; together with the fp8 base products it is refused."""
    import glob
    import json
    from vla_adapter_amd import engine as E, finetune as F, synthetic as S
    batches = [S.make_batch(E.tiny_fused_config(), 3, "cuda", seed=710 + i, P=24, ragged=True) for i in range(2)]
    cfg = F.parse_args(["--tiny", "true", "--backbone", "tiny_fused", "--num_images_in_input", "2", "--use_lora", "True", "--lora_rank", "64",
                        "--lora_dropout", "0.05", "--batch_size", "3", "--max_steps", "10", "--learning_rate", "1e-3", "--wandb_log_freq", "5",
                        "--save_freq", "10", "--run_root_dir", str(tmp_path), "--phase", "Training", "--use_proprio", "True"])
    out = F.finetune(cfg, batches=batches)
    assert out["mode"] == "lora" and out["log"][-1]["loss_value"] < out["log"][0]["loss_value"], out["log"]
    d = glob.glob(os.path.join(str(tmp_path), "*--10_chkpt"))[0]
    assert json.load(open(os.path.join(d, "lora_adapter", "adapter_config.json")))["lora_dropout"] == 0.05
    with pytest.raises(NotImplementedError):
        F.finetune(F.parse_args(["--tiny", "true", "--use_lora", "True", "--lora_dropout", "0.05", "--fp8_base_weights", "True", "--max_steps", "1",
                                 "--run_root_dir", str(tmp_path)]))
