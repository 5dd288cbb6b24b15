"""GPU checks of the reference-API mirrors (OpenVLAForActionPrediction.forward, L1RegressionActionHead.predict_action,
the finetune entry point) on the prismatic-tiny configuration."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import vla_oracle as O  # noqa: E402

DEV, BF = "cuda", torch.bfloat16


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).norm() / (b.norm() + 1e-12)).item()


def test_model_forward_and_head_predict_action_like_run_forward_pass():
    """The reference's call sequence (finetune.py:336-411): vla(...) -> regroup hidden states -> head.predict_action."""
    from vla_adapter_amd import engine as E, synthetic as S
    from vla_adapter_amd.action_heads import L1RegressionActionHead
    from vla_adapter_amd.modeling_prismatic import OpenVLAForActionPrediction
    from vla_adapter_amd.projectors import ProprioProjector
    cfg = E.tiny_config()
    W = S.make_weights(cfg, DEV, seed=5, std=0.05)
    batch = S.make_batch(cfg, 2, DEV, seed=6, P=24, ragged=True)
    vla = OpenVLAForActionPrediction(cfg, W, DEV)
    out = vla(input_ids=batch["input_ids"], attention_mask=batch["attention_mask"], pixel_values=batch["pixel_values"].to(BF),
              labels=batch["labels"], output_hidden_states=True, output_projector_features=True)
    n, Np = cfg.llm.n_layers, cfg.n_patches
    assert len(out.hidden_states) == n + 1 and out.loss is None and out.logits is None
    assert vla.vision_backbone.get_num_patches() == Np and vla.llm_dim == cfg.llm.d
    # regroup exactly as finetune.py:396-409 does (host glue of the reference, index ops only)
    labels = batch["labels"].cpu()
    hs_cpu = [h.float().cpu() for h in out.hidden_states]
    mlhs = O.regroup_hidden_states(hs_cpu, labels, Np)
    head = L1RegressionActionHead(input_dim=cfg.llm.d, hidden_dim=cfg.llm.d, action_dim=7, num_task_tokens=Np,
                                  use_pro_version=True, device=DEV, num_blocks=cfg.num_blocks)
    head.load_state_dict(W["head"], W["proprio"])
    pp = ProprioProjector(cfg.llm.d, 8, DEV)
    pp.load_state_dict(W["proprio"])
    pred = head.predict_action(mlhs.to(BF).to(DEV), proprio=batch["proprio"], proprio_projector=pp, phase="Inference")
    f = lambda sd: {k: v.float().cpu() for k, v in sd.items()}
    ref = O.head_predict_action(mlhs.to(BF).float(), batch["proprio"].cpu().to(BF).float(), f(W["head"]), f(W["proprio"]), Np, True, None,
                                True, cfg.num_blocks)
    assert rel(pred, ref) < 8e-3, rel(pred, ref)
    # the proprio projector alone
    assert rel(pp(batch["proprio"]), O.proprio_projector(batch["proprio"].cpu().to(BF).float(), f(W["proprio"]), True)) < 5e-3
    # and the engine's in-place path gives the same actions as the regrouped-tensor API
    pred2 = vla.engine.forward(batch, None)
    assert rel(pred2, pred) < 8e-3


def test_finetune_entry_point_tiny(tmp_path):
    from vla_adapter_amd import finetune as F
    cfg = F.parse_args(["--tiny", "true", "--batch_size", "4", "--max_steps", "16", "--learning_rate", "2e-3", "--wandb_log_freq", "5",
                        "--save_freq", "10", "--run_root_dir", str(tmp_path), "--phase", "Inference"])
    out = F.finetune(cfg)
    log = out["log"]
    assert set(log[0]) >= {"loss_value", "curr_action_l1_loss", "next_actions_l1_loss"}
    assert log[-1]["loss_value"] < log[0]["loss_value"]
    import glob, os
    files = glob.glob(os.path.join(str(tmp_path), "*", "*"))
    names = {os.path.basename(f) for f in files}
    assert "action_head--10_checkpoint.pt" in names and "proprio_projector--10_checkpoint.pt" in names
    sd = torch.load([f for f in files if f.endswith("action_head--10_checkpoint.pt")][0], weights_only=True)
    assert "model.mlp_resnet_blocks.0.q_proj.weight" in sd and "model.fc2.bias" in sd


def test_predict_action_batch1_inference_matches_oracle():
    """OpenVLAForActionPrediction.predict_action (modeling_prismatic.py:892-972): prompt ids + 64 placeholder ids + stop id,
    fake labels marking the 64 action positions, forward in phase Inference, q01/q99 un-normalisation.  The oracle runs
    the same prepared batch; replayed (captured) calls must reproduce the first (eager-warmed) one, and new inputs must
    flow through the static buffers."""
    import numpy as np
    from vla_adapter_amd import constants as K, engine as E, synthetic as S
    from vla_adapter_amd.action_heads import L1RegressionActionHead
    from vla_adapter_amd.modeling_prismatic import OpenVLAForActionPrediction
    from vla_adapter_amd.projectors import ProprioProjector
    cfg = E.tiny_config()
    W = S.make_weights(cfg, DEV, seed=5, std=0.05)
    stats = {"libero_object": {"action": {"q01": [-0.5, -0.4, -0.3, -0.2, -0.1, -0.6, 0.0], "q99": [0.5, 0.6, 0.7, 0.8, 0.9, 0.4, 1.0],
                                          "min": [-1] * 7, "max": [1] * 7, "mask": [True] * 6 + [False]}}}
    vla = OpenVLAForActionPrediction(cfg, W, DEV, norm_stats=stats)
    g = torch.Generator().manual_seed(31)
    ids = torch.randint(3, 700, (1, 19), generator=g)
    px = torch.randn(1, 3, cfg.vit[0].img, cfg.vit[0].img, generator=g).clamp_(-3, 3)
    proprio = np.linspace(-0.5, 0.5, 8).astype(np.float32)
    head = L1RegressionActionHead(input_dim=cfg.llm.d, hidden_dim=cfg.llm.d, action_dim=7, num_task_tokens=cfg.n_patches,
                                  use_pro_version=True, device=DEV, num_blocks=cfg.num_blocks)
    head.load_state_dict(W["head"], W["proprio"])
    pp = ProprioProjector(cfg.llm.d, 8, DEV)
    pp.load_state_dict(W["proprio"])
    call = lambda i, p: vla.predict_action(input_ids=i, unnorm_key="libero_object", proprio=proprio, proprio_projector=pp,
                                           action_head=head, pixel_values=p.to(BF), attention_mask=torch.ones_like(i, dtype=torch.bool))
    act, hid = call(ids, px)
    assert act.shape == (cfg.chunk, 7) and tuple(hid.shape) == (1, 1, K.NUM_TOKENS, cfg.llm.d)
    act2, _ = call(ids, px)                                   # graph replay
    assert np.array_equal(act, act2)
    # oracle on the same prepared inputs
    pids, pam, plab = OpenVLAForActionPrediction.prepare_inference_inputs(ids, torch.ones_like(ids, dtype=torch.bool))
    assert pids.shape[1] == 19 + 65 and int((plab > K.ACTION_TOKEN_BEGIN_IDX).sum()) == 64 and int(plab[0, -1]) == K.STOP_INDEX
    f = lambda sd: {k: v.float().cpu() for k, v in sd.items()}
    llm = f(W["llm"])
    OW = dict(vit=[f(s) for s in W["vit"]], proj=f(W["proj"]), llm=llm, embed=llm["embed_tokens.weight"],
              action_queries=W["action_queries"].float().cpu(), head=f(W["head"]), proprio=f(W["proprio"]))
    ocfg = dict(vit=[v.as_oracle() for v in cfg.vit], fused=cfg.fused, llm=cfg.llm.as_oracle(), n_img=cfg.n_img, pro=cfg.pro,
                num_blocks=cfg.num_blocks)
    cb = dict(input_ids=pids, labels=plab, attention_mask=pam.bool(), pixel_values=px.to(BF).float(),
              proprio=torch.tensor(proprio).to(BF).float()[None], actions=torch.zeros(1, cfg.chunk, 7))
    ref = O.vla_forward(cb, OW, ocfg, emu=True, noise=None)["pred"].reshape(cfg.chunk, 7).to(BF).float().numpy()
    st = stats["libero_object"]["action"]
    lo, hi, mk = np.array(st["q01"]), np.array(st["q99"]), np.array(st["mask"])
    ref_un = np.where(mk, 0.5 * (ref + 1) * (hi - lo + 1e-8) + lo, ref)
    err = np.linalg.norm(act - ref_un) / np.linalg.norm(ref_un)
    assert err < 2e-2, err
    assert np.allclose(act[:, 6], vla.engine._pred_out.float().cpu().numpy()[0, :, 6])      # masked dim stays normalised
    # different inputs through the same captured graph
    ids2, px2 = torch.randint(3, 700, (1, 19), generator=g), torch.randn(1, 3, cfg.vit[0].img, cfg.vit[0].img, generator=g).clamp_(-3, 3)
    act3, _ = call(ids2, px2)
    assert not np.allclose(act3, act) and len(vla.engine._predict_graphs) == 1
    with pytest.raises(NotImplementedError):
        vla.predict_action(input_ids=ids, action_head=None, pixel_values=px, attention_mask=torch.ones_like(ids))


def test_engine_from_reference_layout_state_dict(tmp_path):
    """A VLM state dict in the reference's HF key layout (saved as .safetensors) loads into the engine and gives the
    forward of the original weights (checkpoints.split_reference_state_dict / merge_reference_state_dict)."""
    from safetensors.torch import save_file
    from vla_adapter_amd import checkpoints as CK, engine as E, synthetic as S
    cfg = E.tiny_config()
    W = S.make_weights(cfg, DEV, seed=21, std=0.05)
    batch = S.make_batch(cfg, 2, DEV, seed=22, P=24)
    ref = E.VLAEngine(cfg, W, DEV).forward(batch, None).clone()
    hf = {k: v.cpu().contiguous() for k, v in CK.merge_reference_state_dict(W, cfg).items()}
    assert "language_model.model.layers.0.self_attn.q_proj.weight" in hf and "vision_backbone.featurizer.pos_embed" in hf
    save_file(hf, str(tmp_path / "vla.safetensors"))
    W2 = CK.split_reference_state_dict(CK.load_file(str(tmp_path / "vla.safetensors")), cfg, head=W["head"], proprio=W["proprio"])
    got = E.VLAEngine(cfg, W2, DEV).forward(batch, None)
    assert torch.equal(got, ref)
