#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REFERENCE's own importable modules (run in the build container only;
/root/reference does not exist on the GPU box).  The fixtures are data (inputs, seeded weights, expected
outputs / gradients); no reference source is copied.

Imported from the reference by path (skipping prismatic/__init__.py, which needs the absent `draccus`):
  prismatic/vla/constants.py, prismatic/training/train_utils.py,
  prismatic/models/action_heads.py, prismatic/models/projectors.py
Third-party stand-in for the reference's pinned transformers fork: installed transformers Qwen2ForCausalLM
(built from a local config, eager attention).  torch.optim.AdamW is what finetune.py:910 instantiates.

Usage: python tools/make_golden.py [--ref /root/reference] [--out tests/golden]
"""
import argparse
import importlib
import os
import sys
import types

import numpy as np
import torch


def import_reference(ref_root: str):
    sys.path.insert(0, ref_root)
    for name in ("prismatic", "prismatic.vla", "prismatic.models", "prismatic.training"):
        m = types.ModuleType(name)
        m.__path__ = [os.path.join(ref_root, *name.split("."))]
        sys.modules[name] = m
    mods = {}
    for name in ("prismatic.vla.constants", "prismatic.training.train_utils",
                 "prismatic.models.action_heads", "prismatic.models.projectors"):
        mods[name.split(".")[-1]] = importlib.import_module(name)
    return mods


def npd(d):
    return {k: (v.detach().to(torch.float32).numpy() if isinstance(v, torch.Tensor) else np.asarray(v))
            for k, v in d.items()}


def seeded_init_(module: torch.nn.Module, seed: int, std: float = 0.05):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in module.named_parameters():
            if name.endswith("gating_factor"):
                p.copy_(torch.tensor([0.3]))
            elif p.dim() >= 2:
                p.copy_(torch.randn(p.shape, generator=g) * std)
            elif "norm" in name or "ffn.0" in name:
                p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g) if name.endswith("weight")
                        else 0.1 * torch.randn(p.shape, generator=g))
            else:
                p.copy_(torch.randn(p.shape, generator=g) * std)


def bf16_head_fixtures(args, ah, pj, C):
    # ---------------- (ii-b) action head run the way finetune.py runs it: bf16 module, CPU ----------------
    # finetune.py:280-281 casts the module with .to(torch.bfloat16); :411 calls action_head.module.predict_action(mlhs, proprio,
    # proprio_projector, phase) OUTSIDE autocast on bf16 hidden states; :418 L1Loss against actions.to(bf16).  Same call here,
    # at MFMA-capable widths (D = 128: head dim 16; D = 896: the real head, head dim 112, 256 / 512 task tokens).  Inputs and
    # the 218 M parameters are regenerated from seeds on both sides (tests/golden_gen.py); the fixture keeps the outputs of the
    # bf16 run, of an fp32 run of the same module (the "truth" the bf16 error budget is measured against), a few gradients,
    # and a digest of every generated input.
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
    import golden_gen as GG
    for case in GG.CASES:
        pro, Dh, Kt, B, phase, nb = GG.case_cfg(case)
        inp = GG.case_inputs(case)
        res = {}
        for dt, tag in ((torch.bfloat16, "bf16"), (torch.float32, "fp32")):
            head = ah.L1RegressionActionHead(input_dim=Dh, hidden_dim=Dh, action_dim=C.ACTION_DIM, num_task_tokens=Kt, use_pro_version=pro)
            if nb != 24:                           # the reference hard-codes 24 blocks (:35); MLPResNet itself takes num_blocks (:87)
                head.model = ah.MLPResNet(num_blocks=nb, input_dim=Dh * C.ACTION_DIM, hidden_dim=Dh, output_dim=C.ACTION_DIM, use_pro_version=pro)
            ppm = pj.ProprioProjector(llm_dim=Dh, proprio_dim=C.PROPRIO_DIM)
            missing = head.load_state_dict(inp["head"], strict=False)
            assert all("film_gen" in k for k in missing.missing_keys) and not missing.unexpected_keys, missing
            ppm.load_state_dict(inp["proprio"])
            head, ppm = head.to(dt), ppm.to(dt)
            if dt == torch.float32:            # predict_action casts proprio to bf16 (:53) whatever the module dtype: feed fp32 back in
                class Up(torch.nn.Module):
                    def __init__(s, m): super().__init__(); s.m = m
                    def forward(s, x): return s.m(x.float())
                ppc = Up(ppm)
            else:
                ppc = ppm
            mlhs = inp["mlhs"].to(dt).requires_grad_(True)
            blk_out = {}
            hooks = [head.model.mlp_resnet_blocks[i].register_forward_hook(lambda m, a, o, i=i: blk_out.__setitem__(i, o.detach().float().numpy()))
                     for i in GG.block_taps(case)]
            orig = ah.learnable_random_perturbations
            if phase == "Training":            # the reference draws fresh N(0, 0.02^2) noise (:14-17, 69-72): inject the seeded one
                ah.learnable_random_perturbations = lambda seq_len, dim, device, dtype: inp["noise"].to(dtype)
            try:
                out = head.predict_action(mlhs, proprio=inp["prop"], proprio_projector=ppc, phase=phase)
            finally:
                ah.learnable_random_perturbations = orig
            loss = torch.nn.L1Loss()(out, inp["target"].to(dt))          # finetune.py:418 (value only; kept for reference)
            out.backward(inp["dpred"].to(dt))                            # fixed upstream gradient (golden_gen.case_inputs)
            res[f"out_{tag}"], res[f"loss_{tag}"] = out.detach().float().numpy(), loss.detach().float().numpy()
            res[f"xblk_{tag}"] = np.stack([blk_out[i] for i in GG.block_taps(case)])      # block outputs [taps, B, 8, D]
            for h in hooks:
                h.remove()
            named = dict(head.named_parameters())
            for k in GG.grad_keys(case):
                res[f"g_{tag}.{k}"] = named[k].grad.float().numpy()
            for k in GG.weight_grad_rows(case):
                res[f"g_{tag}.{k}"] = named[k].grad[:16].float().numpy()
            res[f"g_{tag}.proprio.fc2.bias"] = ppm.fc2.bias.grad.float().numpy()
            if Dh <= 128 or nb == 1:
                res[f"dx_{tag}"] = mlhs.grad[:, GG.dx_layers(case)][:, :, GG.dx_rows(case)].float().numpy()
        np.savez_compressed(os.path.join(args.out, f"head_bf16_{case}.npz"), digest=np.array(GG.digest(inp)),
                            meta=np.array([int(pro), Dh, Kt, B, int(phase == "Training"), nb]), **res)
        print(f"  {case}: bf16-vs-fp32 of the REFERENCE itself: pred rel-L2 "
              f"{np.linalg.norm(res['out_bf16'] - res['out_fp32']) / np.linalg.norm(res['out_fp32']):.3e}")



def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(os.path.dirname(__file__), "..", "tests", "golden"))
    ap.add_argument("--only-bf16-head", action="store_true", help="regenerate just the head_bf16_* fixtures")
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    R = import_reference(args.ref)
    tu, ah, pj, C = R["train_utils"], R["action_heads"], R["projectors"], R["constants"]
    torch.manual_seed(0)
    if args.only_bf16_head:
        bf16_head_fixtures(args, ah, pj, C)
        return
    g = torch.Generator().manual_seed(1234)

    # ---------------- (i) mask KATs: train_utils.py:8-41 ----------------
    rows = []
    L = 100
    for pfx, pad in ((35, 0), (20, 15), (30, 5), (0, 35)):
        ids = torch.randint(C.ACTION_TOKEN_BEGIN_IDX + 1, 151643, (65,), generator=g)
        ids[0] = 13  # last prompt id kept in the labels (datasets.py:124): fails the > ACTION_TOKEN_BEGIN_IDX test
        row = torch.cat([torch.full((pfx,), C.IGNORE_INDEX), ids, torch.full((pad,), C.IGNORE_INDEX)])
        rows.append(row[:L] if row.numel() >= L else torch.cat([row, torch.full((L - row.numel(),), C.IGNORE_INDEX)]))
    labels = torch.stack(rows).long()
    # an adversarial row: non-action ids inside the tail, fewer than 64 hits
    adv = labels[0].clone()
    adv[40:44] = 17
    labels_adv = torch.stack([adv, labels[1]])
    np.savez_compressed(os.path.join(args.out, "masks.npz"),
                        labels=labels.numpy(), cur=tu.get_current_action_mask(labels).numpy(),
                        nxt=tu.get_next_actions_mask(labels).numpy(),
                        labels_shift=labels[:, 1:].numpy(),
                        cur_shift=tu.get_current_action_mask(labels[:, 1:]).numpy(),
                        nxt_shift=tu.get_next_actions_mask(labels[:, 1:]).numpy(),
                        labels_adv=labels_adv.numpy(), cur_adv=tu.get_current_action_mask(labels_adv).numpy(),
                        nxt_adv=tu.get_next_actions_mask(labels_adv).numpy(),
                        constants=np.array([C.IGNORE_INDEX, C.ACTION_TOKEN_BEGIN_IDX, C.NUM_TOKENS, C.ACTION_DIM,
                                            C.NUM_ACTIONS_CHUNK, C.PROPRIO_DIM]))

    # ---------------- (iii) proprio projector: projectors.py:6-24 ----------------
    D = 32
    pp = pj.ProprioProjector(llm_dim=D, proprio_dim=C.PROPRIO_DIM)
    seeded_init_(pp, 7, std=0.2)
    proprio = torch.rand(3, C.PROPRIO_DIM, generator=g) * 2 - 1
    np.savez_compressed(os.path.join(args.out, "proprio_projector.npz"), proprio=proprio.numpy(),
                        out=pp(proprio).detach().numpy(), **{"w." + k: v for k, v in npd(pp.state_dict()).items()})

    # ---------------- (ii) action head, Pro + original: action_heads.py ----------------
    for pro in (True, False):
        for Kt in (8, 24):
            B = 2
            head = ah.L1RegressionActionHead(input_dim=D, hidden_dim=D, action_dim=C.ACTION_DIM,
                                             num_task_tokens=Kt, use_pro_version=pro)
            seeded_init_(head, 11 + Kt + int(pro), std=0.15)
            mlhs = torch.randn(B, 25, Kt + C.NUM_TOKENS, D, generator=g, requires_grad=True)
            prop = (torch.rand(B, C.PROPRIO_DIM, generator=g) * 2 - 1)
            target = torch.rand(B, C.NUM_ACTIONS_CHUNK, C.ACTION_DIM, generator=g) * 2 - 1

            class F32Proprio(torch.nn.Module):  # reference casts proprio to bf16 before the projector (:53); keep fp32 here
                def __init__(s, m): super().__init__(); s.m = m
                def forward(s, x): return s.m(x.float())
            out = head.predict_action(mlhs, proprio=prop, proprio_projector=F32Proprio(pp), phase="Inference")
            # NB reference's `.to(torch.bfloat16)` of proprio (:53) rounds the 8 proprio values: apply the same rounding
            loss = torch.nn.L1Loss()(out, target)
            loss.backward()
            fx = dict(mlhs=mlhs, proprio=prop, target=target, out=out, loss=loss, d_mlhs=mlhs.grad,
                      noise=torch.zeros(1))
            sd = head.state_dict()
            for k in ("model.fc1.weight", "model.fc2.weight", "model.layer_norm1.bias",
                      "model.mlp_resnet_blocks.0.gating_factor", "model.mlp_resnet_blocks.23.o_proj.weight",
                      "model.mlp_resnet_blocks.5.ffn.0.weight",
                      "model.mlp_resnet_blocks.3.k_task.weight" if pro else "model.mlp_resnet_blocks.3.k_proj.weight",
                      "model.mlp_resnet_blocks.7.q_proj.bias"):
                fx["g." + k] = dict(head.named_parameters())[k].grad
            fx["g.proprio.fc1.weight"] = pp.fc1.weight.grad.clone()
            pp.zero_grad()
            # Training phase with the noise tensor captured (action_heads.py:14-17, 69-72)
            cap = {}
            orig = ah.learnable_random_perturbations
            def capture(seq_len, dim, device, dtype):
                t = orig(seq_len, dim, device, dtype); cap["noise"] = t.detach().clone(); return t
            ah.learnable_random_perturbations = capture
            out_tr = head.predict_action(mlhs.detach(), proprio=prop, proprio_projector=F32Proprio(pp), phase="Training")
            ah.learnable_random_perturbations = orig
            fx["noise"], fx["out_train"] = cap["noise"], out_tr
            np.savez_compressed(os.path.join(args.out, f"head_{'pro' if pro else 'orig'}_kt{Kt}.npz"), **npd(fx),
                                **{"w." + k: v for k, v in npd(sd).items()},
                                **{"pw." + k: v for k, v in npd(pp.state_dict()).items()})

    bf16_head_fixtures(args, ah, pj, C)

    # ---------------- (iv) Qwen2 tiny via installed transformers ----------------
    from transformers import Qwen2Config, Qwen2ForCausalLM
    qc = Qwen2Config(hidden_size=128, intermediate_size=320, num_hidden_layers=3, num_attention_heads=4,
                     num_key_value_heads=2, vocab_size=512, rms_norm_eps=1e-6, rope_theta=1000000.0,
                     max_position_embeddings=512, tie_word_embeddings=True, attention_dropout=0.0)
    lm = Qwen2ForCausalLM._from_config(qc, attn_implementation="eager").eval().float()
    seeded_init_(lm, 99, std=0.08)
    B, S = 3, 40
    x = torch.randn(B, S, 128, generator=g, requires_grad=True)
    mask = torch.ones(B, S, dtype=torch.bool)
    mask[1, 33:] = False
    mask[2, 20:] = False
    o = lm(inputs_embeds=x, attention_mask=mask, output_hidden_states=True, use_cache=False)
    hs = torch.stack(o.hidden_states)  # [n+1, B, S, D]
    wsum = torch.randn(hs.shape, generator=g)
    valid = mask[None, :, :, None].float()
    (hs * wsum * valid).sum().backward()
    sd = {k.replace("model.", "", 1): v for k, v in lm.state_dict().items() if k.startswith("model.")}
    np.savez_compressed(os.path.join(args.out, "qwen2_tiny.npz"), x=x.detach().numpy(), mask=mask.numpy(),
                        hs=hs.detach().numpy(), wsum=wsum.numpy(), dx=x.grad.numpy(),
                        cfg=np.array([3, 4, 2, 32]), **{"w." + k: v for k, v in npd(sd).items()})

    # ---------------- (v) AdamW (torch.optim.AdamW, finetune.py:910), fp32 and bf16 states ----------------
    for dt, tag in ((torch.float32, "f32"), (torch.bfloat16, "bf16")):
        p0 = (torch.randn(1024, generator=g) * 0.1).to(dt)
        p = torch.nn.Parameter(p0.clone())
        opt = torch.optim.AdamW([p], lr=5e-4)
        grads, ps = [], []
        for step in range(3):
            gr = (torch.randn(1024, generator=g) * 0.01).to(dt)
            p.grad = gr.clone()
            opt.step()
            grads.append(gr.float()); ps.append(p.detach().float().clone())
        st = opt.state[p]
        np.savez_compressed(os.path.join(args.out, f"adamw_{tag}.npz"), p0=p0.float().numpy(),
                            grads=torch.stack(grads).numpy(), ps=torch.stack(ps).numpy(),
                            m=st["exp_avg"].float().numpy(), v=st["exp_avg_sq"].float().numpy(),
                            hyper=np.array([5e-4, 0.9, 0.999, 1e-8, 0.01]))
    print("golden fixtures written to", os.path.abspath(args.out))
    for f in sorted(os.listdir(args.out)):
        print(f"  {f}: {os.path.getsize(os.path.join(args.out, f)) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
