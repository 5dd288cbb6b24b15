"""The HIP action head fed DIRECTLY with the reference-run fixtures (tests/golden/head_bf16_*.npz: the reference's own
L1RegressionActionHead executed in bf16 on CPU, the way finetune.py:280-281, 411 runs it, plus an fp32 run of the same module
as the truth).  Inputs / the 218 M parameters are regenerated from seeds (tests/golden_gen.py, digest-checked).

Bars (printed with every number):
  * one-block heads: block output and actions within a few 1e-3 of the reference's bf16 run (the MFMA attention keeps its
    scores in fp32 where ATen rounds them to bf16, so not bit-exact), every gradient within 5e-2 of the reference's bf16
    autograd - the reference's own bf16-vs-fp32 gap on those gradients is up to 1e-1;
  * 24-block heads: two valid bf16 evaluations drift to a noise floor (tests/test_oracle_golden.py), so the bar is the distance
    to the fp32 TRUTH: native <= 1.5 x the reference's own bf16 run (actions: 112 numbers), aggregate <= 1.5 x (gradients).
"""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(__file__))
import golden_gen as GG  # noqa: E402

G = os.path.join(os.path.dirname(__file__), "golden")
DEV, BF = "cuda", torch.bfloat16


def rel(a, b):
    a, b = a.detach().float().cpu(), torch.as_tensor(b).float()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def _run_native(case, ref_softmax=False):
    from vla_adapter_amd import engine as E
    z = np.load(os.path.join(G, f"head_bf16_{case}.npz"))
    inp = GG.case_inputs(case)
    assert GG.digest(inp) == str(z["digest"]), "seeded inputs differ from the fixture's"
    pro, D, Kt, B, phase, nb = GG.case_cfg(case)
    cfg = E.VLACfg(llm=E.LLMCfg(d=D), num_blocks=nb, pro=pro)
    head = E.Head(cfg, DEV)
    head.ref_softmax = ref_softmax
    head.load_state_dicts({k: v.to(DEV) for k, v in inp["head"].items()}, {k: v.to(DEV) for k, v in inp["proprio"].items()})
    hs = inp["mlhs"].to(BF).permute(1, 0, 2, 3).contiguous().to(DEV)                 # [nb+1, B, Kt+64, D]: per-layer "sequences"
    pos1 = torch.arange(GG.NUM_TOKENS, device=DEV, dtype=torch.int32)[None].expand(B, -1).contiguous()
    noise = inp["noise"].to(DEV) if inp["noise"] is not None else None
    pred = head.forward(hs, pos1, inp["prop"].to(DEV), Kt, noise)
    dHS = torch.zeros_like(hs)
    head.backward(inp["dpred"].to(BF).to(DEV), dHS)
    torch.cuda.synchronize()
    grads = dict(head.named_views(head.P.grad))
    grads.update({"proprio." + k: v for k, v in head.proprio_views(head.P.grad).items()})
    taps = {i: head.X[i + 1].view(B, GG.CHUNK, D) for i in GG.block_taps(case)}
    dx = dHS.permute(1, 0, 2, 3)[:, GG.dx_layers(case)][:, :, GG.dx_rows(case)]     # [B, layers, rows, D]
    return z, pred, grads, taps, dx


# one-block bounds against the reference's own bf16 run: (block output, actions).  Default forward = flash-style softmax (weights
# rounded before normalisation); ref_softmax = weights rounded where ATen rounds them (vla_head_attn_desc.ref_softmax).
# Measured on the MI355X (round 4, gpurun_out/t_r4_fixture.log) - bounds = measured + 25 %, see the table printed by the test.
#   case              flash-style softmax (default)              ref_softmax
#   pro1_d128_kt64    block 1.31e-3 (9.5 % of elements)  actions 2.22e-3     block 0 (BIT-IDENTICAL: 0 % of elements differ)  actions 0
#   orig1_d128_kt64   block 1.55e-3 (9.4 %)               actions 2.06e-3     block 0 (BIT-IDENTICAL)                          actions 0
#   pro1_d896_kt256   block 1.87e-3 (15 %)                actions 3.04e-3     block 1.24e-3 (8.3 %: one-ulp flips of 896-long   actions 2.32e-3
#                                                                             fp32 dot products summed in another order)
ONE_BLOCK_BOUNDS = {False: (2.4e-3, 3.8e-3), True: (1.6e-3, 2.9e-3)}


@pytest.mark.parametrize("ref_softmax", [False, True])
@pytest.mark.parametrize("case", ["pro1_d128_kt64", "orig1_d128_kt64", "pro1_d896_kt256"])
def test_one_block_head_tracks_the_reference_bf16_run(case, ref_softmax):
    z, pred, grads, taps, dx = _run_native(case, ref_softmax)
    rb, rp = rel(taps[0], z["xblk_bf16"][0]), rel(pred, z["out_bf16"])
    gap = rel(torch.as_tensor(z["out_bf16"]), z["out_fp32"])
    nflip = (taps[0].float().cpu() != torch.as_tensor(z["xblk_bf16"][0]).float()).float().mean().item()
    print(f"{case} ref_softmax={ref_softmax}: native-vs-reference-bf16  block output {rb:.2e} ({100 * nflip:.2f} % of its elements differ)  "
          f"actions {rp:.2e}   (reference bf16-vs-fp32 actions {gap:.2e})")
    bb, ba = ONE_BLOCK_BOUNDS[ref_softmax]
    assert rb <= bb and rp <= ba, (rb, rp)
    if ref_softmax and "d128" in case:
        # north_star: "action-head logits within 1e-3 of reference".  With the attention weights rounded where ATen rounds them, the HIP
        # head reproduces the reference's own bf16 run of this block BIT FOR BIT (fc1, q/k/v projections with RoPE, attention over the
        # three segments with the tanh gate, o-proj, LayerNorm, ffn, the output layers): every hidden unit and every action
        assert nflip == 0.0 and rb == 0.0 and rp == 0.0, (nflip, rb, rp)
    if ref_softmax:
        assert rb <= 1.6e-3, "north_star's 1e-3-class bound on the block output (1.24e-3 measured at D = 896: summation-order flips)"
    worst = (0.0, "")
    gmax = max(np.linalg.norm(z[f"g_bf16.{k}"]) for k in GG.grad_keys(case))
    for k in GG.grad_keys(case) + GG.weight_grad_rows(case) + ["proprio.fc2.bias"]:
        ref16, ref32 = torch.as_tensor(z[f"g_bf16.{k}"]), torch.as_tensor(z[f"g_fp32.{k}"])
        g = grads[k].float().cpu().reshape(-1, ref16.shape[-1])[:ref16.reshape(-1, ref16.shape[-1]).shape[0]].reshape(ref16.shape)
        r, gp = rel(g, ref16), rel(ref16, ref32)
        err = (g - ref16).norm().item()
        if r > worst[0]:
            worst = (r, k)
        # tensors far below the head's dominant gradients sit on the absolute bf16 noise floor of the chain that feeds them
        assert r <= 5e-2 or err <= 1e-3 * gmax, f"{k}: native-vs-reference-bf16 {r:.3e} (reference bf16-vs-fp32 {gp:.3e})"
    rdx = rel(dx, z["dx_bf16"])
    print(f"{case}: gradients native-vs-reference-bf16: worst parameter {worst[0]:.2e} ({worst[1]}), hidden states {rdx:.2e}")
    assert rdx <= 3e-2


@pytest.mark.parametrize("ref_softmax", [False, True])
@pytest.mark.parametrize("case", ["pro_d128_kt64", "pro_d128_kt64_train", "orig_d128_kt64", "pro_d896_kt256", "pro_d896_kt256_train",
                                  "pro_d896_kt512"])
def test_full_head_within_the_reference_bf16_error_budget(case, ref_softmax):
    z, pred, grads, taps, dx = _run_native(case, ref_softmax)
    for j, i in enumerate(GG.block_taps(case)[:3]):
        r = rel(taps[i], z["xblk_bf16"][j])
        nfl = (taps[i].float().cpu() != torch.as_tensor(z["xblk_bf16"][j]).float()).float().mean().item()
        print(f"{case} ref_softmax={ref_softmax}: block {i} output native-vs-reference-bf16 {r:.2e} ({100 * nfl:.2f} % of its elements differ)")
        assert r <= 4e-3 * (i + 1)
        if ref_softmax and case in ("pro_d128_kt64", "orig_d128_kt64"):
            assert nfl == 0.0, "phase Inference at D = 128: the tapped blocks reproduce the reference's bf16 run bit for bit"
    gap = rel(torch.as_tensor(z["out_bf16"]), z["out_fp32"])
    r_ref, r_truth = rel(pred, z["out_bf16"]), rel(pred, z["out_fp32"])
    print(f"{case} ref_softmax={ref_softmax}: actions  native-vs-ref_bf16 {r_ref:.3e}  native-vs-ref_fp32 {r_truth:.3e}  ref_bf16-vs-ref_fp32 {gap:.3e}")
    # the actions are a 112-element tensor at the end of the chain: one realisation's distance fluctuates by ~ +-20 % around the
    # expected one (measured spread over the six cases: 0.67 .. 1.25 x the reference's own), hence 1.5 here; the large tensors
    # (hidden states, tests/test_engine_gpu.py) get the 1.25
    assert r_truth <= 1.5 * gap, "further from the fp32 truth than 1.5 x the reference's own bf16 run"
    assert r_ref <= 1.5 * 2 ** 0.5 * gap
    if ref_softmax:
        return              # (the backward is the same kernel in either mode: its budget is asserted once, on the default forward)
    se = sr = 0.0
    for k in GG.grad_keys(case) + ["proprio.fc2.bias"]:
        ref16, ref32 = torch.as_tensor(z[f"g_bf16.{k}"]), torch.as_tensor(z[f"g_fp32.{k}"])
        g = grads[k].float().cpu().reshape(ref16.shape)
        n32 = ref32.norm().item() + 1e-30
        e, r = (g - ref32).norm().item() / n32, (ref16 - ref32).norm().item() / n32
        se, sr = se + e * e, sr + r * r
        assert e <= max(4.0 * r + 5e-3, 0.15), f"{k}: native-vs-fp32 {e:.3e} against the reference's bf16-vs-fp32 {r:.3e}"
    print(f"{case}: gradients: rms distance to fp32 truth  native {se ** 0.5:.3e}  reference bf16 run {sr ** 0.5:.3e}")
    assert se ** 0.5 <= 1.5 * sr ** 0.5 + 2e-3
    if "dx_bf16" in z.files:
        e = (dx.float().cpu() - torch.as_tensor(z["dx_fp32"])).norm().item()
        budget = max(1.5 * np.linalg.norm(z["dx_bf16"] - z["dx_fp32"]), 0.15 * np.linalg.norm(z["dx_fp32"]))
        assert e <= budget, (e, budget)


def test_mirror_predict_action_on_the_fixture():
    """The reference-named entry point (L1RegressionActionHead.predict_action, action_heads.py:43-49) on the same fixture."""
    from vla_adapter_amd.action_heads import L1RegressionActionHead
    from vla_adapter_amd.projectors import ProprioProjector
    case = "pro_d896_kt256"
    z = np.load(os.path.join(G, f"head_bf16_{case}.npz"))
    inp = GG.case_inputs(case)
    pro, D, Kt, B, phase, nb = GG.case_cfg(case)
    head = L1RegressionActionHead(input_dim=D, hidden_dim=D, action_dim=GG.ACTION_DIM, num_task_tokens=Kt, use_pro_version=pro)
    pp = ProprioProjector(llm_dim=D, proprio_dim=GG.PROPRIO_DIM)
    pp.load_state_dict({k: v.to(DEV) for k, v in inp["proprio"].items()})
    head.load_state_dict({k: v.to(DEV) for k, v in inp["head"].items()}, pp.state_dict())
    pred = head.predict_action(inp["mlhs"].to(DEV), proprio=inp["prop"].to(DEV), proprio_projector=pp, phase="Inference")
    gap = rel(torch.as_tensor(z["out_bf16"]), z["out_fp32"])
    assert rel(pred, z["out_fp32"]) <= 1.25 * gap, (rel(pred, z["out_fp32"]), gap)
