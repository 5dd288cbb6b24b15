"""Pin oracle/vla_oracle.py against fixtures generated from the reference's own modules
(tools/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import vla_oracle as O

G = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    z = np.load(os.path.join(G, name))
    return {k: torch.from_numpy(z[k]) for k in z.files if z[k].dtype.kind != 'U'}


def close(a, b, rel=1e-4):
    """max-norm relative check: fp32 reassociation noise scales with the tensor's magnitude."""
    err = (a - b).abs().max().item()
    ref = b.abs().max().item()
    assert err <= rel * ref + 1e-9, f"max|a-b|={err:.3e} vs rel*max|b|={rel * ref:.3e}"


def sub(d, prefix):
    return {k[len(prefix):]: v for k, v in d.items() if k.startswith(prefix)}


def test_constants_and_masks():
    z = load("masks.npz")
    assert z["constants"].tolist() == [O.IGNORE_INDEX, O.ACTION_TOKEN_BEGIN_IDX, O.NUM_TOKENS, O.ACTION_DIM,
                                       O.NUM_ACTIONS_CHUNK, O.PROPRIO_DIM]
    for lab, cur, nxt in (("labels", "cur", "nxt"), ("labels_shift", "cur_shift", "nxt_shift"),
                          ("labels_adv", "cur_adv", "nxt_adv")):
        assert torch.equal(O.current_action_mask(z[lab]), z[cur].bool())
        assert torch.equal(O.next_actions_mask(z[lab]), z[nxt].bool())
    m = O.all_actions_mask(z["labels"])
    assert (m.sum(1) == O.NUM_TOKENS).all()           # exactly 64 per row
    assert (O.current_action_mask(z["labels"]).sum(1) == 6).all()   # 6 + 58 (first kept id is a prompt id)


def test_proprio_projector():
    z = load("proprio_projector.npz")
    out = O.proprio_projector(z["proprio"], sub(z, "w."))
    torch.testing.assert_close(out, z["out"], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("name,pro,kt", [("head_pro_kt8", True, 8), ("head_pro_kt24", True, 24),
                                         ("head_orig_kt8", False, 8), ("head_orig_kt24", False, 24)])
def test_action_head_forward_backward(name, pro, kt):
    z = load(name + ".npz")
    hp, pp = sub(z, "w."), sub(z, "pw.")
    hp = {k: v.clone().requires_grad_(True) for k, v in hp.items()}
    pp = {k: v.clone().requires_grad_(True) for k, v in pp.items()}
    mlhs = z["mlhs"].clone().requires_grad_(True)
    prop = z["proprio"].to(torch.bfloat16).float()    # action_heads.py:53 rounds proprio to bf16
    out = O.head_predict_action(mlhs, prop, hp, pp, kt, pro)
    torch.testing.assert_close(out, z["out"], rtol=2e-4, atol=2e-5)
    loss = O.l1_loss(out, z["target"])
    torch.testing.assert_close(loss, z["loss"], rtol=1e-5, atol=1e-6)
    loss.backward()
    close(mlhs.grad, z["d_mlhs"], 1e-3)
    for k, v in z.items():
        if k.startswith("g.proprio."):
            close(pp[k[len("g.proprio."):]].grad, v, 5e-3)
        elif k.startswith("g."):
            close(hp[k[2:]].grad, v, 5e-3)
    # Training phase with the captured noise tensor injected
    with torch.no_grad():
        out_tr = O.head_predict_action(z["mlhs"], prop, sub(z, "w."), sub(z, "pw."), kt, pro, noise=z["noise"])
    close(out_tr, z["out_train"], 2e-4)


def test_qwen2_tiny_forward_backward():
    z = load("qwen2_tiny.npz")
    n, H, KV, dh = z["cfg"].tolist()
    cfg = dict(n_layers=n, heads=H, kv_heads=KV, dh=dh, eps=1e-6, theta=1e6)
    x = z["x"].clone().requires_grad_(True)
    hs = torch.stack(O.qwen2_forward(x, z["mask"].bool(), sub(z, "w."), cfg))
    valid = z["mask"].bool()[None, :, :, None]
    # padded query rows are compared too (same causal&key-mask semantics), but they carry no information
    torch.testing.assert_close(hs, z["hs"], rtol=2e-4, atol=2e-5)
    (hs * z["wsum"] * valid.float()).sum().backward()
    close(x.grad, z["dx"])


@pytest.mark.parametrize("name", ["vit_siglip_tiny", "vit_dinov2reg_tiny"])
def test_vit_restatement_matches_independent_implementations(name):
    """SURVEY a3, pinned in round 4 by third-party stand-ins: oracle.vit_forward - restated from the reference text because timm is
    absent - against installed transformers' SiglipVisionModel / Dinov2WithRegistersModel built from local configs
    (tools/make_golden_vit.py): patch convolution, position embedding on the patch tokens, cls + 4 register tokens in front (DINOv2),
    pre-norm blocks with biased q/k/v, erf GELU, LayerScale, and the output BEHIND block depth-2 without the prefix tokens and without a
    final norm (modeling_prismatic.py:141-142, 196-237)."""
    z = load(name + ".npz")
    d, depth, heads, mlp, P, npre, ls = z["cfg"].tolist()
    cfg = dict(d=d, depth=depth, heads=heads, mlp=mlp, patch=P, n_prefix=npre, layerscale=bool(ls), eps=1e-6, gelu_tanh=False)
    out = O.vit_forward(z["pixels"], sub(z, "w."), cfg, emu=False)
    torch.testing.assert_close(out, z["out"], rtol=2e-4, atol=2e-5)
    # the rounding-point emulation stays a small perturbation of the same function
    emu = O.vit_forward(z["pixels"], sub(z, "w."), cfg, emu=True)
    assert ((emu - z["out"]).norm() / z["out"].norm()).item() < 2e-2


def test_token_ce_matches_the_hf_causal_lm_loss():
    """SURVEY 8f-4, pinned in round 4: oracle.token_ce on the last hidden state = the loss and the logits installed transformers'
    Qwen2ForCausalLM returns for the same multimodal labels (tools/make_golden_ce.py; stand-in for the reference's pinned fork, whose
    LLM computes this loss behind PrismaticVLM.forward, vlms/prismatic.py:469-481): label placement around the patch rows, the shift
    by one, the mean over the labelled positions, IGNORE_INDEX rows and a padded tail."""
    z = load("qwen2_tiny_ce.npz")
    loss, logits = O.token_ce(z["hidden_last"], z["lm_head"], z["labels"].long(), int(z["num_patches"]), emu=False)
    torch.testing.assert_close(logits, z["logits"], rtol=1e-4, atol=1e-5)
    assert abs(loss.item() - float(z["loss"])) <= 2e-6 * abs(float(z["loss"])), (loss.item(), float(z["loss"]))
    # and the arithmetic itself, independent of any fixture: torch's own cross entropy on the shifted logits
    import torch.nn.functional as F
    B, Np = z["labels"].shape[0], int(z["num_patches"])
    mm = torch.cat([z["labels"][:, :1].long(), torch.full((B, Np), -100), z["labels"][:, 1:].long()], 1)
    ref = F.cross_entropy(logits[:, :-1].reshape(-1, logits.shape[-1]), mm[:, 1:].reshape(-1), ignore_index=-100)
    assert abs(loss.item() - ref.item()) <= 2e-6 * abs(ref.item())


@pytest.mark.parametrize("tag,emu", [("f32", False), ("bf16", True)])
def test_adamw(tag, emu):
    z = load(f"adamw_{tag}.npz")
    lr, b1, b2, eps, wd = z["hyper"].tolist()
    p, m, v = z["p0"].clone(), torch.zeros(1024), torch.zeros(1024)
    for step in range(3):
        p, m, v = O.adamw_step(p, z["grads"][step], m, v, step + 1, lr, b1, b2, eps, wd, emu=emu)
        if emu:
            assert torch.equal(p, z["ps"][step]), f"bf16 AdamW not bit-exact at step {step}"
        else:
            torch.testing.assert_close(p, z["ps"][step], rtol=1e-6, atol=1e-8)
    torch.testing.assert_close(m, z["m"], rtol=1e-5 if not emu else 0, atol=1e-9 if not emu else 0)
    torch.testing.assert_close(v, z["v"], rtol=1e-5 if not emu else 0, atol=1e-12 if not emu else 0)


def test_lr_schedule():
    # finetune.py:1061-1065 with default lr_warmup_steps=0.1 -> 100 % at step 0, and the warm-up block keeps overwriting the
    # MultiStepLR decay (x0.1 only exists with lr_warmup_steps <= 0; pinned against torch in test_host_api_cpu.py)
    assert O.lr_at(0, 5e-4) == pytest.approx(5e-4)
    assert O.lr_at(100000, 5e-4) == pytest.approx(5e-4)
    assert O.lr_at(100000, 5e-4, warmup_steps=0) == pytest.approx(5e-5)
    assert O.lr_at(0, 1.0, warmup_steps=10) == pytest.approx(0.1 + 0.9 * 0.1)


# ------------------------------------------------------------------------------------------------------------------
# Reference head run the way finetune.py runs it (module.to(bfloat16), CPU; finetune.py:280-281, 411) at MFMA-capable
# widths: tests/golden/head_bf16_*.npz hold that run's outputs, an fp32 run of the same module, and a digest of the
# seeded inputs (tests/golden_gen.py).  Pins BOTH oracle modes: emu=False against the fp32 run, emu=True - the mode
# every GPU parity test compares with - against the bf16 run, with the reference's OWN bf16-vs-fp32 gap as the bound.
# ------------------------------------------------------------------------------------------------------------------
def rel_l2(a, b):
    return ((a.float() - b.float()).norm() / (b.float().norm() + 1e-30)).item()


def _oracle_head_run(inp, case, emu, with_grad):
    import golden_gen as GG
    pro, D, Kt, B, phase, nb = GG.case_cfg(case)
    leaf = lambda d: {k: v.clone().requires_grad_(with_grad) for k, v in d.items()}
    hp, pp = leaf(inp["head"]), leaf(inp["proprio"])
    mlhs = inp["mlhs"].clone().requires_grad_(with_grad)
    taps = {}
    out = O.head_predict_action(mlhs, inp["prop"], hp, pp, Kt, pro, inp["noise"], emu, num_blocks=nb, taps=taps)
    if with_grad:
        out.backward(inp["dpred"])
    return out.detach(), hp, pp, mlhs, taps


def _fixture(case):
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    import golden_gen as GG
    z = load(f"head_bf16_{case}.npz")
    inp = GG.case_inputs(case)
    assert GG.digest(inp) == str(np.load(os.path.join(G, f"head_bf16_{case}.npz"))["digest"]), "seeded inputs differ from the fixture's"
    return GG, z, inp


@pytest.mark.parametrize("case", ["pro1_d128_kt64", "orig1_d128_kt64", "pro1_d896_kt256"])
def test_oracle_one_block_head_tracks_the_reference_bf16_run(case):
    """ONE block between the input stage and the output stage: no accumulated drift, so the bf16-emulating oracle must
    reproduce the reference's bf16 run almost bit for bit - forward (block output: isolated one-ulp flips only) AND backward
    (every bias / LayerNorm / gate gradient, slices of every weight gradient, the hidden-state gradient) to <= 2.5e-2 per
    tensor (ATen's fused bf16 backward kernels round once per op, autograd through the restated math at each primitive), well
    below the reference's own bf16-vs-fp32 gap on the same tensors (up to 1e-1).  This is the pin of the
    rounding points of emu=True, the mode every GPU parity test compares with."""
    GG, z, inp = _fixture(case)
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    out32, hp32, pp32, mlhs32, taps32 = _oracle_head_run(inp, case, False, True)
    oute, hpe, ppe, mlhse, tapse = _oracle_head_run(inp, case, True, True)
    assert rel_l2(out32, z["out_fp32"]) < 2e-5 and rel_l2(taps32[0], z["xblk_fp32"][0]) < 2e-5
    flips = (tapse[0] != z["xblk_bf16"][0]).float().mean().item()
    print(f"{case}: block output {flips * 100:.3f} % elements differ from the reference bf16 run; pred rel {rel_l2(oute, z['out_bf16']):.2e} "
          f"(reference bf16-vs-fp32 {rel_l2(z['out_bf16'], z['out_fp32']):.2e})")
    # D = 128: bit-exact.  D = 896: the 896-long fp32 dot products of mkldnn and of a plain matmul differ in the last bits, which
    # flips a bf16 rounding in ~1e-3 of the outputs of every Linear; the block output collects them (one ulp each)
    big = GG.case_cfg(case)[1] > 128
    assert flips <= (0.1 if big else 5e-3) and rel_l2(oute, z["out_bf16"]) <= (3e-3 if big else 1e-6)
    worst, worst_gap = 0.0, 0.0
    for k in GG.grad_keys(case) + GG.weight_grad_rows(case) + ["proprio.fc2.bias"]:
        ref16, ref32 = z[f"g_bf16.{k}"], z[f"g_fp32.{k}"]
        g = lambda d, dp: (dp["fc2.bias"] if k == "proprio.fc2.bias" else d[k]).grad
        ge, g32 = g(hpe, ppe), g(hp32, pp32)
        if ge.dim() == 2 and ge.shape[0] > ref16.shape[0]:
            ge, g32 = ge[:16], g32[:16]
        assert rel_l2(g32, ref32) < 2e-4, (k, rel_l2(g32, ref32))
        r, gap = rel_l2(ge, ref16), rel_l2(ref16, ref32)
        worst, worst_gap = max(worst, r), max(worst_gap, gap)
        assert r <= 2.5e-2, f"{k}: emu-vs-reference-bf16 {r:.3e} (reference bf16-vs-fp32 {gap:.3e})"
    dxe, dx32 = mlhse.grad[:, GG.dx_layers(case)][:, :, GG.dx_rows(case)], mlhs32.grad[:, GG.dx_layers(case)][:, :, GG.dx_rows(case)]
    assert rel_l2(dx32, z["dx_fp32"]) < 2e-4
    r = rel_l2(dxe, z["dx_bf16"])
    print(f"{case}: gradients emu-vs-reference-bf16: worst parameter {worst:.2e}, hidden states {r:.2e}; reference bf16-vs-fp32 up to {worst_gap:.2e}")
    assert r <= 2.5e-2


@pytest.mark.parametrize("case", ["pro_d128_kt64", "pro_d128_kt64_train", "orig_d128_kt64", "pro_d896_kt256", "pro_d896_kt256_train",
                                  "pro_d896_kt512"])
def test_oracle_modes_against_reference_bf16_and_fp32_runs(case):
    """What is asserted, and why these bounds:
    * emu=False reproduces the reference's fp32 run (2e-5: fp32 reassociation only);
    * emu=True reproduces the reference's bf16 run OP BY OP: the outputs of the first blocks agree bit for bit except for
      isolated one-ulp flips where fp32 summation orders differ (mkldnn vs a plain matmul) - that pins every rounding point
      (incl. the bf16 inv_freq buffer and the double rounding of Linear on strided h_t);
    * further down, two VALID bf16 evaluations drift apart: every flip is carried forward and the 24-block map settles at a
      noise floor of ~5e-3 on the block outputs, ~1e-2 on the actions - the same size as the reference's own bf16-vs-fp32
      gap.  Independent realisations at distance g from the truth sit sqrt(2) g apart, so the actions are bounded by their
      distance to the fp32 TRUTH (<= 1.25 x the reference's own) and by 1.25 sqrt(2) g between each other."""
    GG, z, inp = _fixture(case)
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    big = GG.CASES[case][1] > 128
    out32, hp32, _, mlhs32, taps32 = _oracle_head_run(inp, case, False, not big)
    assert rel_l2(out32, z["out_fp32"]) < 2e-5, rel_l2(out32, z["out_fp32"])
    for j, i in enumerate(GG.BLOCK_TAPS):
        assert rel_l2(taps32[i], z["xblk_fp32"][j]) < 2e-5
    oute, hpe, ppe, mlhse, tapse = _oracle_head_run(inp, case, True, True)
    # --- rounding points, op by op: early block outputs of the bf16 run
    for j, i in enumerate(GG.BLOCK_TAPS[:3]):
        ref = z["xblk_bf16"][j]
        flips = (tapse[i] != ref).float().mean().item()
        r = rel_l2(tapse[i], ref)
        print(f"{case}: block {i} output vs reference bf16 run: {flips * 100:.3f} % elements differ, rel-L2 {r:.2e}")
        # D = 128 Inference: bit-exact.  Otherwise isolated one-ulp flips (last-bit differences of the fp32 dot products of
        # mkldnn and of a plain matmul: ~6 % of a D = 896 block output), which the following blocks spread
        assert flips <= 0.1 * 2.2 ** i and r <= 1.5e-3 * (i + 1), (i, flips, r)
    # --- actions
    gap = rel_l2(z["out_bf16"], z["out_fp32"])
    r_ref, r_truth = rel_l2(oute, z["out_bf16"]), rel_l2(oute, z["out_fp32"])
    print(f"{case}: pred  emu-vs-ref_bf16 {r_ref:.3e}  emu-vs-ref_fp32 {r_truth:.3e}  ref_bf16-vs-ref_fp32 {gap:.3e}")
    assert r_truth <= 1.25 * gap, f"oracle(emu) is further from the fp32 truth ({r_truth:.3e}) than 1.25 x the reference's bf16 run ({gap:.3e})"
    assert r_ref <= 1.25 * 2 ** 0.5 * gap
    # --- gradients (fixed upstream gradient): distance to the fp32 truth against the reference bf16 run's own distance.  The kept
    # tensors are small (7 .. 896 elements): one ReLU pre-activation within rounding of zero that falls on the other side moves a
    # 128-element bias / LayerNorm gradient by ~10 %, so a single tensor only has to stay below max(4 x its budget, 0.15); the
    # aggregate over all of them is what has to stay within 1.5 x the reference's own.
    se = sr = 0.0
    for k in GG.GRAD_KEYS + ["proprio.fc2.bias"]:
        ref16, ref32 = z[f"g_bf16.{k}"], z[f"g_fp32.{k}"]
        got = (ppe["fc2.bias"] if k == "proprio.fc2.bias" else hpe[k]).grad
        if not big and k != "proprio.fc2.bias":
            assert rel_l2(hp32[k].grad, ref32) < 1e-4, (k, rel_l2(hp32[k].grad, ref32))
        n32 = ref32.float().norm().item() + 1e-30
        e, r = (got.float() - ref32.float()).norm().item() / n32, (ref16.float() - ref32.float()).norm().item() / n32
        se, sr = se + e * e, sr + r * r
        assert e <= max(4.0 * r + 5e-3, 0.15), f"{k}: emu-vs-fp32 {e:.3e} against the reference's bf16-vs-fp32 {r:.3e}"
    print(f"{case}: gradients: rms distance to fp32 truth  oracle(emu) {se ** 0.5:.3e}  reference bf16 run {sr ** 0.5:.3e}")
    assert se ** 0.5 <= 1.5 * sr ** 0.5 + 2e-3
    if not big:
        dx = mlhse.grad[:, GG.DX_LAYERS]
        e = (dx - z["dx_fp32"]).norm().item()
        budget = max(1.5 * (z["dx_bf16"] - z["dx_fp32"]).norm().item(), 0.15 * z["dx_fp32"].norm().item())   # 0.15: ReLU-flip outliers, see above
        assert e <= budget, (e, budget)
        assert rel_l2(mlhs32.grad[:, GG.DX_LAYERS], z["dx_fp32"]) < 1e-4


@pytest.mark.parametrize("H,W,oh,ow", [(256, 256, 224, 224), (300, 200, 224, 224), (128, 128, 224, 224), (224, 224, 224, 224), (480, 640, 224, 224),
                                        (17, 23, 8, 9)])
def test_resize_restatement_is_bit_exact_against_pillow(H, W, oh, ow):
    """processing_prismatic.py:128-145 resizes with TVF.resize(PIL image, BICUBIC, antialias=True) = PIL.Image.resize: the
    arithmetic is Pillow's.  The oracle restates its fixed-point two-pass resampler; pinned bit for bit against the installed
    Pillow (down- and up-scaling, non-square, identity)."""
    from PIL import Image
    rng = np.random.default_rng(H * 1000 + W)
    img = rng.integers(0, 256, size=(H, W, 3), dtype=np.uint8)
    img[: H // 3] = np.linspace(0, 255, W, dtype=np.uint8)[None, :, None]          # smooth ramps + noise + saturated corners
    img[-2:, -2:] = 255
    ref = np.asarray(Image.fromarray(img).resize((ow, oh), resample=Image.BICUBIC))
    got = O.resize_bicubic_u8(img, oh, ow)
    assert got.shape == ref.shape and np.array_equal(got, ref), f"max |diff| {np.abs(got.astype(int) - ref.astype(int)).max()}"
