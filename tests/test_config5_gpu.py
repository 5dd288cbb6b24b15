"""BASELINE.json configs[4]: DINOv2 + SigLIP dual vision + Qwen2.5-1.5B + LoRA rank 64 + the fp8 MFMA weight path - exercised
together ("config5: no full-size GPU test; trainers refuse the fp8 path").

* single-layer gradient checks at the Qwen2.5-1.5B GEOMETRY (d 1536, 12 heads of 128, 2 KV heads, MLP 8960: head dim 128 takes the
  unfused RoPE and the 128-wide attention kernels) for the full fine-tune, LoRA, and LoRA with the base products on e4m3 operands
  (forward and dX) - the same Checker and wrong-scale self-test as tests/test_layer_gradients_gpu.py;
* the whole config at FULL size, batch 2, two images: the step is finite, every adapter of a live layer receives a gradient, the
  four LLM layers above the head's last block stay untouched, a few steps lower the loss - bf16 and fp8.
Reference: vla-scripts/finetune.py:832-844 (peft LoRA on every Linear), prismatic/models/backbones/llm/qwen25.py:26-28 (the 1.5B
backbone); the reference has no fp8 code (PARITY UNPINNED: the oracle's FP8 registry restates the native arithmetic).
"""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(__file__))

from oracle import vla_oracle as O  # noqa: E402
from test_layer_gradients_gpu import Checker, TOL_BIAS, llm_layer_oracle  # noqa: E402

DEV, BF = "cuda", torch.bfloat16
LLM_NAMES = ["self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj", "self_attn.o_proj", "mlp.gate_proj", "mlp.up_proj", "mlp.down_proj"]
# e4m3 operands: native and oracle quantise bf16 tensors that agree to ~5e-3 (the layer's own intermediates: the bf16 test above),
# and a value that moves by 0.5 % crosses an e4m3 rounding boundary (6-12 % apart) in ~5 % of the elements, each flip a FULL e4m3 step
# (3.5 x the rms quantisation error): two valid evaluations of the same fp8 arithmetic sit about as far from each other as either sits
# from the un-rounded truth.  Measured on the MI355X (round 4, gpurun_out/t_r4_b.log): native-vs-emu 1.3e-2 ... 4.2e-2 per tensor with
# native-vs-fp32 = emu-vs-fp32 to within 5 % on every one of the 15 tensors.  The criterion is therefore the error budget of the
# end-to-end tests - native no further from the truth, and from the emulation, than 1.25 x the emulation is from the truth - and the
# self-test scales a gradient by 1.10 (a 1.05 error drowns in e4m3 noise on the tensors that sit 5e-2 from the truth).
F8_BUDGET = 1.25
# bf16 at this geometry (d 1536, head dim 128, the plumbing-size batch of 3 x 136 rows, weights N(0, 0.03)): dX sits 5.0e-3 from the
# emulating oracle with BOTH 5.0e-3 from fp32 (gpurun_out/t_r4_all.log) - two valid bf16 evaluations apart; bound = measured + 50 %
TOL_DX_15B = 7.5e-3


def _geometry_setup(seed=5):
    from vla_adapter_amd import engine as E, synthetic as S
    cfg = E.qwen15b_geometry_config(2)
    W = S.make_weights(cfg, DEV, seed=seed, std=0.03)
    batch = S.make_batch(cfg, 3, DEV, seed=seed + 1, P=24, ragged=True)
    return cfg, W, batch


def test_full_finetune_one_llm_layer_at_the_15b_geometry():
    from vla_adapter_amd import engine as E
    from vla_adapter_amd.trainers import FullFinetune
    cfg, W, batch = _geometry_setup()
    eng = E.VLAEngine(cfg, W, DEV)
    ft = FullFinetune(eng)
    kl = 1
    ft.taps = {("llm", kl): {}}
    pred = ft.forward(batch, None)
    loss3 = ft.backward(pred, batch["actions"])
    torch.cuda.synchronize()
    assert torch.isfinite(loss3).all() and torch.isfinite(ft.P.grad.float()).all()
    G = ft.reference_named_gradients()
    B, S_, D = eng.B, eng.S, cfg.llm.d
    assert D == 1536 and cfg.llm.dh == 128
    t = ft.taps[("llm", kl)]
    res = {emu: llm_layer_oracle(W["llm"], kl, eng.llm.HS[kl], eng.llm.kmask.bool().cpu(), t["d_out"].view(B, S_, D), cfg, emu) for emu in (True, False)}
    ck = Checker(f"full fine-tune, Qwen2.5-1.5B geometry: LLM layer {kl}")
    ck.add("dX", t["d_in"].view(B, S_, D), res[True][0], res[False][0], TOL_DX_15B)
    for k in sorted(res[True][1]):
        ck.add(k, G["language_model.model." + k], res[True][1][k].grad, res[False][1][k].grad, TOL_BIAS if k.endswith(".bias") else TOL_DX_15B)
    ck.run()
    ck.must_catch_a_wrong_scale("dX")
    ck.must_catch_a_wrong_scale(f"layers.{kl}.mlp.down_proj.weight")
    ck.must_catch_a_wrong_scale(f"layers.{kl}.self_attn.k_proj.weight")


@pytest.mark.parametrize("fp8", [False, True])
def test_lora_one_llm_layer_at_the_15b_geometry(fp8):
    """LoRA rank 64 on the 1.5B layer; fp8=True: the seven base products of the layer on e4m3 operands in forward AND dX (the oracle
    registers the same seven weights in its FP8 registry, FP8_BWD on)."""
    from vla_adapter_amd import engine as E
    from vla_adapter_amd.trainers import LoRAFinetune
    cfg, W, batch = _geometry_setup()
    eng = E.VLAEngine(cfg, W, DEV)
    lo = LoRAFinetune(eng, rank=64, seed=1, fp8=fp8)
    if fp8:
        fk, bk = lo.fp8_keys()
        assert all(f"llm.{i}.{k}" in fk and f"llm.{i}.{k}" in bk for i in range(2) for k in ("qkv", "o", "gu", "down")), (fk, bk)
    g = torch.Generator(device=DEV).manual_seed(2)
    for l in lo.L.values():               # peft starts at B = 0: give B a value so that both branches carry signal
        for p_, _ in l.projs:
            Bv_ = lo.P.view(f"{l.name}.{p_}.lora_B")
            Bv_[:l.n_real, :l.r] = (torch.randn(min(l.n_real, Bv_.shape[0]), l.r, generator=g, device=DEV) * 0.01).to(BF)
    lo.refresh()
    kl = 1
    lo.taps = {("llm", kl): {}}
    pred = lo.forward(batch, None)
    loss3 = lo.backward(pred, batch["actions"])
    torch.cuda.synchronize()
    assert torch.isfinite(loss3).all() and torch.isfinite(lo.P.grad.float()).all()
    sd = {k: v.detach().float().cpu().clone() for k, v in lo.lora_state_dict().items()}
    pre = f"base_model.model.language_model.model.layers.{kl}."
    gsd = {}
    for l in lo.L.values():
        for p_, _ in l.projs:
            gsd[f"{l.name}.{p_}.lora_A.weight"], gsd[f"{l.name}.{p_}.lora_B.weight"] = lo.P.g(f"{l.name}.{p_}.lora_A")[:l.r, :l.k_real], lo.P.g(f"{l.name}.{p_}.lora_B")[:l.n_real, :l.r]
    B, S_, D = eng.B, eng.S, cfg.llm.d
    t = lo.taps[("llm", kl)]

    def registrar(store, with_fp8=fp8):
        def reg(p):
            for n in LLM_NAMES:
                A = sd[f"{pre}{n}.lora_A.weight"].clone().requires_grad_(True)
                Bm = sd[f"{pre}{n}.lora_B.weight"].clone().requires_grad_(True)
                store[n] = (A, Bm)
                key = [k for k in p if k.endswith(n + ".weight")][0]
                O.LORA[id(p[key])] = (A, Bm, 2.0)
                if with_fp8:
                    O.FP8.add(id(p[key]))
        return reg

    O.LORA_FUSED, O.FP8_BWD = True, fp8
    try:
        res, stores = {}, {}
        for emu in (True, False):
            stores[emu] = {}
            O.FP8.clear()
            res[emu] = llm_layer_oracle(W["llm"], kl, eng.llm.HS[kl], eng.llm.kmask.bool().cpu(), t["d_out"].view(B, S_, D), cfg, emu, lora=registrar(stores[emu]))
        rows = [("dX", t["d_in"].view(B, S_, D), res[True][0], res[False][0])]
        for n in LLM_NAMES:
            for w, idx in (("lora_A", 0), ("lora_B", 1)):
                rows.append((f"{n}.{w}", gsd[f"{pre}{n}.{w}.weight"], stores[True][n][idx].grad, stores[False][n][idx].grad))
        if not fp8:
            ck = Checker(f"LoRA, Qwen2.5-1.5B geometry: LLM layer {kl}")
            for r_ in rows:
                ck.add(*r_, TOL_DX_15B)
            ck.run()
            ck.must_catch_a_wrong_scale("mlp.up_proj.lora_B")
            ck.must_catch_a_wrong_scale("self_attn.k_proj.lora_A")
        else:
            from test_layer_gradients_gpu import rel

            def f8_check(rows_, verbose):
                for name, nat, emu_, tru in rows_:
                    emu_, tru = emu_.reshape(nat.shape), tru.reshape(nat.shape)
                    r_ne, r_nt, r_et = rel(nat, emu_), rel(nat, tru), rel(emu_, tru)
                    if verbose:
                        print(f"  LoRA + fp8 base products, 1.5B geometry / {name}: native-vs-emu {r_ne:.2e}  native-vs-fp32 {r_nt:.2e}  emu-vs-fp32 {r_et:.2e}")
                    assert r_nt <= F8_BUDGET * r_et + 2e-3 and r_ne <= F8_BUDGET * r_et + 2e-3, (name, r_ne, r_nt, r_et)

            f8_check(rows, True)
            for bad in ("dX", "mlp.up_proj.lora_B", "self_attn.k_proj.lora_A", "self_attn.o_proj.lora_B"):      # discriminating power: one gradient x 1.10
                with pytest.raises(AssertionError):
                    f8_check([(n_, a_ * 1.10 if n_ == bad else a_, e_, t_) for n_, a_, e_, t_ in rows], False)
        if fp8:          # the e4m3 products really ran: against the bf16 oracle (same LoRA pairs, no FP8 registry) the distance is the quantisation's
            from test_layer_gradients_gpu import rel
            O.FP8.clear()
            r16 = llm_layer_oracle(W["llm"], kl, eng.llm.HS[kl], eng.llm.kmask.bool().cpu(), t["d_out"].view(B, S_, D), cfg, True, lora=registrar({}, False))
            d = rel(t["d_in"].view(B, S_, D), r16[0])
            print(f"fp8 dX vs the bf16 oracle: rel-L2 {d:.3e} (the e4m3 quantisation error; a bf16 run sits at ~2e-3)")
            assert 8e-3 < d < 2e-1, d
    finally:
        O.LORA_FUSED, O.FP8_BWD = False, False
        O.LORA.clear()
        O.FP8.clear()


@pytest.mark.parametrize("fp8", [False, True])
def test_config5_full_size_lora_step(fp8):
    """The whole of BASELINE configs[4] at full size - DINOv2-L + SigLIP-so400m, two images per sample (S = 608), Qwen2.5-1.5B (28
    layers, 24 of them under the head), LoRA rank 64 on all 287 fused Linears, bf16 or the fp8 base-weight path - at batch 2."""
    from vla_adapter_amd import engine as E, synthetic as S
    from vla_adapter_amd.trainers import LoRAFinetune
    cfg = E.config5_backbone()
    cfg.n_img = 2
    W = S.make_weights(cfg, DEV, seed=0)
    batch = S.make_batch(cfg, 2, DEV, seed=11, P=32)
    batch["pixel_values"] = batch["pixel_values"].to(BF)
    eng = E.VLAEngine(cfg, W, DEV)
    del W
    lo = LoRAFinetune(eng, rank=64, seed=1, fp8=fp8)
    assert lo.n_active == 24 and cfg.llm.n_layers == 28 and eng.cfg.n_patches == 512
    if fp8:
        fk, bk = lo.fp8_keys()
        assert len(fk) == len(lo.L) and len(bk) == len(lo.L), "every contraction length of config 5 is a multiple of 128: all base products on e4m3"
    g = torch.Generator(device=DEV).manual_seed(2)
    for l in lo.L.values():
        for p_, _ in l.projs:
            Bv_ = lo.P.view(f"{l.name}.{p_}.lora_B")
            Bv_[:l.n_real, :l.r] = (torch.randn(min(l.n_real, Bv_.shape[0]), l.r, generator=g, device=DEV) * 0.01).to(BF)
    lo.refresh()
    p0 = lo.P.data.clone()
    pred = lo.forward(batch, None)
    loss3 = lo.backward(pred, batch["actions"])
    torch.cuda.synchronize()
    assert eng.S == 608 and torch.isfinite(loss3).all() and torch.isfinite(lo.P.grad.float()).all()
    dead, fed = 0, 0
    for key, l in lo.L.items():
        is_dead = key.startswith("llm.") and int(key.split(".")[1]) >= 24
        for p_, _ in l.projs:
            for w in ("lora_A", "lora_B"):
                gmax = lo.P.g(f"{l.name}.{p_}.{w}").float().abs().max().item()
                if is_dead:
                    assert gmax == 0.0, f"{key}.{p_}.{w}: layers above the head's last block receive no gradient"
                    dead += 1
                else:
                    assert gmax > 0.0, f"{key}.{p_}.{w}: every adapter of a live layer is fed"
                    fed += 1
    assert dead == 4 * 7 * 2 and fed > 500, (dead, fed)
    l0 = loss3[0].item()
    lo.optimizer_step(1e-4)
    ls = [lo.train_step(batch, 1e-4)[0].item() for _ in range(5)]
    torch.cuda.synchronize()
    # AdamW's first update moves every one of the head's ~640 M random-init parameters by lr at once (the loss jumps, as in the
    # reference's optimizer); from there the steps must bring it down steadily (measured 2.68 -> 1.69 -> 1.29 -> 1.00 -> ...)
    # (batch 2 at lr 1e-4 is noisy once the loss is below 1: the fifth step may give back a few per cent - 0.880 -> 0.890 on the fp8 path,
    #  0.837 -> 0.892 in bf16 were seen; the first three steps must each come down, the run must more than halve the loss)
    assert all(v == v for v in ls) and all(b_ < a_ for a_, b_ in zip(ls[:4], ls[1:4])) and ls[-1] < 1.15 * ls[-2] and ls[-1] < 0.5 * ls[0], (l0, ls)
    first_dead = lo.P.offsets[f"{lo.L['llm.24.qkv'].name}.q_proj.lora_A"][0]
    assert torch.equal(lo.P.data[first_dead:], p0[first_dead:]), "the adapters of the four dead layers are left alone"
    assert not torch.equal(lo.P.data[:first_dead], p0[:first_dead])
    print(f"config 5 full size, LoRA{' + fp8' if fp8 else ''}: loss {l0:.4f} -> {ls}")


def test_finetune_entry_point_lora_with_fp8_base_weights(tmp_path):
    """`vla-scripts/finetune.py --use_lora True --fp8_base_weights True` on the plumbing-size DINOv2 + SigLIP dual config: the run
    completes, the loss falls, the adapter is saved.  (At this size only the Linears whose contraction length is a multiple of 128
    take the e4m3 path - the 256-wide LLM and the 128 / 512-wide ViT products; the 192-wide ones keep bf16: the mixed case.)"""
    import glob
    from safetensors.torch import load_file
    from vla_adapter_amd import engine as E, finetune as F, synthetic as S
    batches = [S.make_batch(E.tiny_fused_config(), 3, "cuda", seed=700 + i, P=24, ragged=True) for i in range(2)]
    cfg = F.parse_args(["--tiny", "true", "--backbone", "tiny_fused", "--num_images_in_input", "2", "--use_lora", "True", "--lora_rank", "64",
                        "--fp8_base_weights", "True", "--batch_size", "3", "--max_steps", "10", "--learning_rate", "1e-3", "--wandb_log_freq", "5",
                        "--save_freq", "10", "--run_root_dir", str(tmp_path), "--phase", "Training", "--use_proprio", "True"])
    out = F.finetune(cfg, batches=batches)
    assert out["mode"] == "lora" and out["log"][-1]["loss_value"] < out["log"][0]["loss_value"], out["log"]
    d = glob.glob(os.path.join(str(tmp_path), "*--10_chkpt"))[0]
    ad = load_file(os.path.join(d, "lora_adapter", "adapter_model.safetensors"))
    assert "base_model.model.language_model.model.layers.1.mlp.down_proj.lora_B.weight" in ad
    with pytest.raises(NotImplementedError):
        F.finetune(F.parse_args(["--tiny", "true", "--fp8_base_weights", "True", "--max_steps", "1", "--run_root_dir", str(tmp_path)]))
