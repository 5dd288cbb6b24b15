"""GPU checks of the reference-API mirrors (OpenVLAForActionPrediction.forward, L1RegressionActionHead.predict_action,
the finetune entry point) on the prismatic-tiny configuration."""
import os
import sys

import pytest
sys.path.insert(0, os.path.dirname(__file__))
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


def _tiny_batches(n, B=4, seed0=300):
    from vla_adapter_amd import engine as E, synthetic as S
    return [S.make_batch(E.tiny_config(), B, "cuda", seed=seed0 + i, P=32, ragged=True) for i in range(n)]


def test_finetune_entry_point_tiny(tmp_path):
    """Reference loop bookkeeping on the native engine: batches cycle through an iterable (their pixels staged one step ahead
    of the captured graphs), gradient steps 0..max_steps inclusive, checkpoints on save_freq multiples plus a final one."""
    from vla_adapter_amd import finetune as F
    cfg = F.parse_args(["--tiny", "true", "--batch_size", "4", "--max_steps", "16", "--learning_rate", "2e-3", "--wandb_log_freq", "5",
                        "--save_freq", "10", "--run_root_dir", str(tmp_path), "--phase", "Inference", "--use_proprio", "True", "--use_fz", "True"])
    out = F.finetune(cfg, batches=_tiny_batches(2))
    log = out["log"]
    assert out["steps"] == 17 and out["final_step"] == 16 and [l["step"] for l in log] == [0, 5, 10, 15, 16]
    assert set(log[0]) >= {"loss_value", "curr_action_l1_loss", "next_actions_l1_loss"}
    assert log[-1]["loss_value"] < log[0]["loss_value"]
    import glob, os
    files = glob.glob(os.path.join(str(tmp_path), "*", "*"))
    names = {os.path.basename(f) for f in files}
    assert "action_head--10_checkpoint.pt" in names and "proprio_projector--10_checkpoint.pt" in names
    assert "action_head--16_checkpoint.pt" in names, "the run must not end without a checkpoint of its last step"
    sd = torch.load([f for f in files if f.endswith("action_head--10_checkpoint.pt")][0], weights_only=True)
    assert "model.mlp_resnet_blocks.0.q_proj.weight" in sd and "model.fc2.bias" in sd


def test_finetune_graphed_loop_equals_eager_loop_on_a_batch_sequence(tmp_path):
    """Three different batches through the captured path (static buffers refreshed every step, next pixels staged a step ahead)
    give the same losses as the eager path on the same sequence."""
    from vla_adapter_amd import finetune as F
    bs = _tiny_batches(3, seed0=320)
    base = ["--tiny", "true", "--batch_size", "4", "--max_steps", "5", "--learning_rate", "1e-3", "--wandb_log_freq", "1", "--save_freq", "1000",
            "--phase", "Inference", "--use_proprio", "True", "--use_fz", "True"]
    a = F.finetune(F.parse_args(base + ["--run_root_dir", str(tmp_path / "a"), "--use_graph", "true"]), batches=bs)
    b = F.finetune(F.parse_args(base + ["--run_root_dir", str(tmp_path / "b"), "--use_graph", "false"]), batches=bs)
    la, lb = [l["loss_value"] for l in a["log"]], [l["loss_value"] for l in b["log"]]
    assert len(la) == 6 and abs(la[0] - lb[0]) < 1e-6
    assert all(abs(x - y) <= 2e-2 * abs(y) for x, y in zip(la, lb)), (la, lb)
    assert len({round(x, 5) for x in lb[:3]}) == 3, "the three batches must differ for the check to mean anything"


def test_finetune_resume_restores_every_trainable_tensor_and_the_step_counter(tmp_path):
    """resume used to restart the action queries from random init and the step counter from 0."""
    from vla_adapter_amd import finetune as F, engine as E, synthetic as S, checkpoints as CK
    bs = _tiny_batches(2, seed0=340)
    base = ["--tiny", "true", "--batch_size", "4", "--learning_rate", "2e-3", "--wandb_log_freq", "1", "--phase", "Inference", "--use_proprio", "True", "--use_fz", "True",
            "--run_root_dir", str(tmp_path), "--run_id_override", "r"]
    F.finetune(F.parse_args(base + ["--max_steps", "4", "--save_freq", "4"]), batches=bs)
    ck = str(tmp_path / "r--4_chkpt")
    head, pp, aq = CK.load_run_dir(ck, 4, with_action_queries=True)
    out = F.finetune(F.parse_args(base + ["--max_steps", "6", "--save_freq", "100", "--resume", "True", "--resume_step", "4", "--resum_vla_path", ck,
                                          "--learning_rate", "0.0"]), batches=bs)
    assert [l["step"] for l in out["log"]] == [4, 5, 6]                    # log_step = resume_step + gradient_step_idx (:1056)
    h2, p2, aq2 = CK.load_run_dir(str(tmp_path / "r--6_chkpt"), 6, with_action_queries=True)
    # lr 0 and weight decay x lr = 0: the resumed run must hand back exactly what it loaded
    assert torch.equal(aq, aq2) and all(torch.equal(head[k], h2[k]) for k in head) and all(torch.equal(pp[k], p2[k]) for k in pp)
    fresh = S.make_weights(E.tiny_config(), "cuda", seed=0)["action_queries"].cpu()
    assert not torch.equal(aq, fresh), "the checkpoint must hold TRAINED action queries"


def test_gradient_accumulation_matches_one_big_batch():
    """finetune.py:1039-1042: two micro-batches of 2 with loss / 2 == one batch of 4 (mean L1 over twice the samples), eager
    and captured; the optimizer steps once per two micro-steps."""
    from vla_adapter_amd import engine as E, synthetic as S
    cfg = E.tiny_config()
    W = S.make_weights(cfg, "cuda", seed=3, std=0.05)
    big = S.make_batch(cfg, 4, "cuda", seed=360, P=32)
    halves = [{k: v[i:i + 2].contiguous() for k, v in big.items()} for i in (0, 2)]
    e_big, e_acc, e_gr = E.VLAEngine(cfg, W, "cuda"), E.VLAEngine(cfg, W, "cuda"), E.VLAEngine(cfg, W, "cuda")
    e_big.train_step(big, 1e-3)
    e_acc.set_grad_accumulation(2)
    p0 = e_acc.head.P.data.clone()
    e_acc.train_step(halves[0], 1e-3)
    assert torch.equal(p0, e_acc.head.P.data) and e_acc.step_count == 0, "no optimizer step on the first micro-batch"
    e_acc.train_step(halves[1], 1e-3)
    torch.cuda.synchronize()
    assert e_acc.step_count == 1
    g_big, g_acc = e_big.head.P.grad.float(), e_acc.head.P.grad.float()
    assert (g_big - g_acc).norm() <= 1.5e-2 * g_big.norm(), ((g_big - g_acc).norm() / g_big.norm()).item()
    # captured: same two micro-steps through the graphs
    e_gr.set_grad_accumulation(2)
    static = {k: v.clone() for k, v in halves[0].items()}
    e_gr.capture(static, None)
    e_gr.stage_next_pixels(halves[1]["pixel_values"])
    e_gr.train_step_graphed(1e-3)
    assert e_gr._pending_lr is None, "no update pending after the first micro-step"
    for k in static:
        static[k].copy_(halves[1][k])
    e_gr.train_step_graphed(1e-3)
    e_gr.flush()
    torch.cuda.synchronize()
    assert e_gr.step_count == 1
    assert (e_gr.head.P.grad.float() - g_acc).norm() <= 2e-3 * g_acc.norm()
    assert (e_gr.head.P.data.float() - e_acc.head.P.data.float()).norm() <= 1e-3 * e_acc.head.P.data.float().norm()


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


def test_prismatic_vlm_forward_token_ce_loss_and_logits():
    """prismatic/models/vlms/prismatic.py:312-481 (the native, non-HF API): plain VLM forward - no action queries - with the HF
    shifted cross-entropy over the vocabulary.  Hidden states, logits and loss against the oracle restatement."""
    from oracle import vla_oracle as O
    from vla_adapter_amd import engine as E, synthetic as S, modeling_prismatic as M
    cfg = E.tiny_config()
    W = S.make_weights(cfg, "cuda", seed=21, std=0.05)
    batch = S.make_batch(cfg, 3, "cuda", seed=22, P=24, ragged=True)
    vla = M.OpenVLAForActionPrediction(cfg, W, "cuda")
    vlm = M.PrismaticVLM(vla)
    labels = batch["labels"].clone()
    labels[labels > cfg.llm.vocab - 1] = cfg.llm.vocab - 7            # tiny vocab: keep the targets inside the lm_head's range
    out = vlm(input_ids=batch["input_ids"], attention_mask=batch["attention_mask"], pixel_values=batch["pixel_values"], labels=labels,
              output_hidden_states=True)
    torch.cuda.synchronize()
    f = lambda sd: {k: v.float().cpu() for k, v in sd.items()}
    res = {}
    Np, n = cfg.n_patches, cfg.llm.n_layers
    for emu in (True, False):
        px = batch["pixel_values"].float().cpu()
        patches = O.projector(O.vit_forward(px[:, :3], f(W["vit"][0]), cfg.vit[0].as_oracle(), emu), f(W["proj"]), cfg.fused, emu)
        emb = f(W["llm"])["embed_tokens.weight"]
        e = emb[batch["input_ids"].cpu()]
        mm = torch.cat([e[:, :1], patches, e[:, 1:]], dim=1)
        am = batch["attention_mask"].cpu().bool()
        mask = torch.cat([am[:, :1], torch.ones(3, Np, dtype=torch.bool), am[:, 1:]], dim=1)
        hs = O.qwen2_forward(mm, mask, f(W["llm"]), cfg.llm.as_oracle(), emu)
        loss, logits = O.token_ce(hs[-1], emb, labels.cpu(), Np, emu)
        res[emu] = (hs, loss, logits)
    valid = torch.cat([am[:, :1], torch.ones(3, Np, dtype=torch.bool), am[:, 1:]], dim=1)
    from test_engine_gpu import budget
    budget(out.hidden_states[n].float().cpu()[valid], res[True][0][n][valid], res[False][0][n][valid], "PrismaticVLM hidden_states[-1]")
    budget(out.logits.float().cpu()[valid], res[True][2][valid], res[False][2][valid], "PrismaticVLM logits")
    le, lt = res[True][1].item(), res[False][1].item()
    assert abs(out.loss.item() - lt) <= 1.25 * abs(le - lt) + 2e-3 * abs(lt), (out.loss.item(), le, lt)
    assert out.logits.shape == (3, batch["input_ids"].shape[1] + Np, cfg.llm.vocab)


@pytest.mark.parametrize("mode", ["lora", "full"])
def test_finetune_entry_point_lora_and_full_modes(tmp_path, mode):
    """--use_lora True (peft-style adapters on every Linear of the VLM, finetune.py:832-844) and the reference's default with
    use_lora False (every VLM parameter trains, :846-849): loss falls, the checkpoints the reference writes exist with its key
    names (lora_adapter/ for LoRA, optionally the merged VLM; the whole VLM for full fine-tune)."""
    from vla_adapter_amd import finetune as F
    from safetensors.torch import load_file
    import glob
    extra = ["--use_lora", "True", "--lora_rank", "8", "--merge_lora_during_training", "True"] if mode == "lora" else []
    cfg = F.parse_args(["--tiny", "true", "--batch_size", "4", "--max_steps", "10", "--learning_rate", "1e-3" if mode == "lora" else "3e-4",
                        "--wandb_log_freq", "5", "--save_freq", "10", "--run_root_dir", str(tmp_path), "--phase", "Inference", "--use_proprio", "True"] + extra)
    out = F.finetune(cfg, batches=_tiny_batches(2, seed0=360))
    assert out["mode"] == mode and out["log"][-1]["loss_value"] < out["log"][0]["loss_value"], out["log"]
    d = glob.glob(os.path.join(str(tmp_path), "*--10_chkpt"))[0]
    names = set(os.listdir(d))
    assert {"action_head--10_checkpoint.pt", "proprio_projector--10_checkpoint.pt"} <= names
    # where the VLM goes follows the reference: merged LoRA model at the top level (:579-601), the full fine-tune's
    # save_pretrained(adapter_dir) under lora_adapter/ (:553-554)
    vlm = load_file(os.path.join(d, "model.safetensors") if mode == "lora" else os.path.join(d, "lora_adapter", "model.safetensors"))
    assert "language_model.model.layers.0.self_attn.q_proj.weight" in vlm and "vision_backbone.featurizer.blocks.0.mlp.fc1.weight" in vlm
    assert "projector.fc1.weight" in vlm and "action_queries.weight" in vlm
    if mode == "lora":
        ad = load_file(os.path.join(d, "lora_adapter", "adapter_model.safetensors"))
        k = "base_model.model.language_model.model.layers.1.mlp.gate_proj.lora_B.weight"
        assert k in ad and ad[k].shape[1] == 8 and ad[k].abs().max() > 0



def _fused_batches(n, B=3, seed0=500):
    from vla_adapter_amd import engine as E, synthetic as S
    return [S.make_batch(E.tiny_fused_config(), B, "cuda", seed=seed0 + i, P=24, ragged=True) for i in range(n)]


def test_documented_recipe_runs_on_the_tiny_dual_config(tmp_path):
    """README.md:254-274 of the reference: ``--vlm_path ...dinosiglip-224px-0_5b --num_images_in_input 2 --use_lora True
    --lora_rank 64 --merge_lora_during_training True`` - here on the plumbing-size DINOv2 + SigLIP geometry: the
    run completes, the loss falls, the adapter holds pairs for BOTH backbones, the merged VLM carries LayerScale and the cls /
    register tokens under the reference's key names."""
    from vla_adapter_amd import finetune as F
    from safetensors.torch import load_file
    import glob
    cfg = F.parse_args(["--tiny", "true", "--backbone", "tiny_fused", "--num_images_in_input", "2", "--use_lora", "True", "--lora_rank", "64",
                        "--merge_lora_during_training", "True", "--batch_size", "3", "--max_steps", "10", "--learning_rate", "1e-3", "--wandb_log_freq", "5",
                        "--save_freq", "10", "--run_root_dir", str(tmp_path), "--phase", "Training", "--use_proprio", "True"])
    out = F.finetune(cfg, batches=_fused_batches(2))
    assert out["mode"] == "lora" and out["model"]["n_img"] == 2 and len(out["model"]["vit"]) == 2
    assert out["log"][-1]["loss_value"] < out["log"][0]["loss_value"], out["log"]
    d = glob.glob(os.path.join(str(tmp_path), "*--10_chkpt"))[0]
    ad = load_file(os.path.join(d, "lora_adapter", "adapter_model.safetensors"))
    for k in ("base_model.model.vision_backbone.featurizer.blocks.0.attn.qkv.lora_A.weight",
              "base_model.model.vision_backbone.fused_featurizer.blocks.1.mlp.fc2.lora_B.weight",
              "base_model.model.projector.fc3.lora_A.weight", "base_model.model.language_model.model.layers.1.mlp.down_proj.lora_B.weight"):
        assert k in ad and 64 in ad[k].shape, k
    vlm = load_file(os.path.join(d, "model.safetensors"))
    for k in ("vision_backbone.featurizer.blocks.0.ls1.scale_factor", "vision_backbone.featurizer.cls_token", "vision_backbone.featurizer.reg_token",
              "vision_backbone.fused_featurizer.pos_embed", "projector.fc3.weight", "action_queries.weight"):
        assert k in vlm, k


@pytest.mark.parametrize("mode", ["lora", "full"])
def test_lora_and_full_resume_restore_what_they_trained(tmp_path, mode):
    """--resume in LoRA / full mode used to restore head + proprio + queries only: now the adapter
    (lora_adapter/adapter_model.safetensors) / the trained VLM come back too - a resumed run at lr 0 hands back what it loaded."""
    from vla_adapter_amd import finetune as F
    from safetensors.torch import load_file
    extra = ["--use_lora", "True", "--lora_rank", "8"] if mode == "lora" else []
    base = ["--tiny", "true", "--batch_size", "4", "--wandb_log_freq", "1", "--phase", "Inference", "--use_proprio", "True", "--run_root_dir", str(tmp_path),
            "--run_id_override", "r"] + extra
    bs = _tiny_batches(2, seed0=380)
    F.finetune(F.parse_args(base + ["--max_steps", "4", "--save_freq", "4", "--learning_rate", "1e-3"]), batches=bs)
    ck = str(tmp_path / "r--4_chkpt")
    F.finetune(F.parse_args(base + ["--max_steps", "6", "--save_freq", "100", "--resume", "True", "--resume_step", "4", "--resum_vla_path", ck,
                                    "--learning_rate", "0.0"]), batches=bs)
    f = "adapter_model.safetensors" if mode == "lora" else "model.safetensors"
    a, b = load_file(os.path.join(ck, "lora_adapter", f)), load_file(os.path.join(str(tmp_path / "r--6_chkpt"), "lora_adapter", f))
    assert set(a) == set(b) and all(torch.equal(a[k], b[k]) for k in a)
    if mode == "lora":
        assert any(v.abs().max() > 0 for k, v in a.items() if "lora_B" in k), "the checkpoint must hold TRAINED adapters"


def test_finetune_picks_the_model_from_the_checkpoint(tmp_path):
    """Without --backbone / --tiny the geometry is read off the --vlm_path state dict (the reference builds the model from the
    checkpoint's config, finetune.py:777-816): a fused DINOv2 + SigLIP checkpoint selects the fused geometry."""
    from vla_adapter_amd import finetune as F, engine as E, synthetic as S, checkpoints as CK
    from safetensors.torch import save_file
    cfg = E.tiny_fused_config()
    W = S.make_weights(cfg, "cuda", seed=7)
    f = str(tmp_path / "vlm.safetensors")
    save_file({k: v.contiguous().cpu() for k, v in CK.merge_reference_state_dict(W, cfg).items()}, f)
    out = F.finetune(F.parse_args(["--vlm_path", f, "--num_images_in_input", "2", "--use_fz", "True", "--batch_size", "3", "--max_steps", "2",
                                   "--wandb_log_freq", "1", "--save_freq", "100", "--run_root_dir", str(tmp_path), "--phase", "Inference", "--use_proprio", "True"]),
                     batches=_fused_batches(2, seed0=520))
    m = out["model"]
    assert [v["d"] for v in m["vit"]] == [192, 128] and m["vit"][0]["n_prefix"] == 5 and m["vit"][0]["layerscale"] and m["llm"]["d"] == 256
    assert all(l["loss_value"] == l["loss_value"] for l in out["log"])


@pytest.mark.parametrize("mode", ["full", "lora"])
def test_trainer_gradient_accumulation_matches_one_big_batch(mode):
    """finetune.py:1039-1042 for the backbone trainers: two micro-batches of 2 with loss / 2 == one batch of 4, eager and captured;
    one optimizer step per two micro-steps."""
    from vla_adapter_amd import engine as E, synthetic as S
    from vla_adapter_amd.trainers import FullFinetune, LoRAFinetune
    cfg = E.tiny_config()
    W = S.make_weights(cfg, "cuda", seed=3, std=0.05)
    big = S.make_batch(cfg, 4, "cuda", seed=361, P=32)
    halves = [{k: v[i:i + 2].contiguous() for k, v in big.items()} for i in (0, 2)]
    mk = lambda: (FullFinetune(E.VLAEngine(cfg, W, "cuda")) if mode == "full" else LoRAFinetune(E.VLAEngine(cfg, W, "cuda"), rank=8, seed=3))
    t_big, t_acc, t_gr = mk(), mk(), mk()
    if mode == "lora":                               # B = 0 at initialisation gives the A matrices no gradient: start from a trained-looking B
        g = torch.Generator(device="cuda").manual_seed(5)
        for t in (t_big, t_acc, t_gr):
            g.manual_seed(5)
            for l in t.L.values():
                for p_, _ in l.projs:
                    Bv = t.P.view(f"{l.name}.{p_}.lora_B")
                    Bv[:l.n_real, :l.r] = (torch.randn(min(l.n_real, Bv.shape[0]), l.r, generator=g, device="cuda") * 0.05).to(torch.bfloat16)
            t.refresh()
    t_big.train_step(big, 1e-3)
    t_acc.set_grad_accumulation(2)
    p0 = t_acc.P.data.clone()
    t_acc.train_step(halves[0], 1e-3)
    assert torch.equal(p0, t_acc.P.data) and t_acc.step_count == 0, "no optimizer step on the first micro-batch"
    t_acc.train_step(halves[1], 1e-3)
    torch.cuda.synchronize()
    assert t_acc.step_count == 1
    for a, b in ((t_big.P.grad, t_acc.P.grad), (t_big.head.P.grad, t_acc.head.P.grad)):
        a, b = a.float(), b.float()
        assert (a - b).norm() <= 2e-2 * a.norm(), ((a - b).norm() / a.norm()).item()
    t_gr.set_grad_accumulation(2)
    static = {k: v.clone() for k, v in halves[0].items()}
    t_gr.capture(static, None)
    t_gr.train_step_graphed(1e-3)
    assert t_gr.step_count == 0
    for k in static:
        static[k].copy_(halves[1][k])
    t_gr.train_step_graphed(1e-3)
    torch.cuda.synchronize()
    assert t_gr.step_count == 1
    assert (t_gr.P.grad.float() - t_acc.P.grad.float()).norm() <= 2e-3 * t_acc.P.grad.float().norm()
    assert (t_gr.P.data.float() - t_acc.P.data.float()).norm() <= 1e-3 * t_acc.P.data.float().norm()
