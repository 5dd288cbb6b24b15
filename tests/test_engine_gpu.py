"""End-to-end parity of the native engine (forward, backward, optimiser step) against the CPU oracle on the
prismatic-tiny configuration (BASELINE.json configs[0]) with the same seeded weights and batch.

Bar: the whole pipeline is bf16 with fp32 accumulation, and two valid bf16 evaluations of the same
network drift apart (tests/test_oracle_golden.py measures it on the reference itself), so every end-to-end check is an
ERROR BUDGET against the fp32 truth:   |native - oracle_fp32|  <=  1.25 x |oracle_emu - oracle_fp32|
where oracle_emu is the bf16-emulating oracle (rounding points pinned op by op against the reference's bf16 run).  Both
numbers are printed.  Gradients get 1.5: they pass through ReLU masks, and ONE pre-activation within rounding of zero that
falls on the other side moves a 128-wide gradient by ~10 % (measured between the reference's own bf16 run and the oracle,
tests/test_oracle_golden.py).  For families of small gradient tensors a single tensor's ratio fluctuates (one realisation of
rounding noise each): each tensor gets 4 x its own budget plus a floor, the aggregate over the family 1.5 x.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import vla_oracle as O  # noqa: E402

DEV = "cuda"
BF = torch.bfloat16


def cpu_f32(sd):
    return {k: v.detach().float().cpu() for k, v in sd.items()}


def oracle_weights(W):
    llm = cpu_f32(W["llm"])
    return dict(vit=[cpu_f32(s) for s in W["vit"]], proj=cpu_f32(W["proj"]), llm=llm, embed=llm["embed_tokens.weight"],
                action_queries=W["action_queries"].float().cpu(), head=cpu_f32(W["head"]), proprio=cpu_f32(W["proprio"]))


def oracle_cfg(cfg):
    return dict(vit=[v.as_oracle() for v in cfg.vit], fused=cfg.fused, llm=cfg.llm.as_oracle(), n_img=cfg.n_img, pro=cfg.pro,
                num_blocks=cfg.num_blocks)


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).norm() / (b.norm() + 1e-12)).item()


def budget(native, emu, truth, what, factor=1.25, floor=0.0):
    """|native - fp32| <= factor x |emu - fp32| + floor x |fp32|; returns (native distance, emu distance) relative to |fp32|."""
    n, e, t = (x.detach().float().cpu().reshape(-1) for x in (native, emu, truth))
    nt = t.norm().item() + 1e-30
    dn, de = (n - t).norm().item() / nt, (e - t).norm().item() / nt
    print(f"budget {what}: native-vs-fp32 {dn:.3e}   oracle(emu)-vs-fp32 {de:.3e}   ratio {dn / (de + 1e-30):.2f}")
    assert dn <= factor * de + floor, f"{what}: native is {dn:.3e} from the fp32 truth, the bf16-emulating oracle {de:.3e} (x{factor} + {floor})"
    return dn, de


def budget_family(items, what, each=2.5, total=1.5, floor=2e-3, absfloor=0.0):
    """items: [(name, native, emu, truth)].  Per tensor: native <= each x emu + floor (or |err| <= absfloor: tensors orders of
    magnitude below the family's dominant ones sit on the absolute noise floor of the chain feeding them); aggregate
    (root of summed squared relative distances) <= total x emu's + floor.  `each`: 4.0 until round 3 (too loose); the
    worst single-tensor ratios measured over every call site are 0.8 - 2.07 (LoRA A of one gate_proj: one ReLU / rounding flip
    upstream of a rank-8 tensor) and 2.36 for the original block's shared k/v gradient (that call passes each=3), printed per run."""
    sn = se = 0.0
    worst = (0.0, None, 0.0, 0.0)
    for name, native, emu, truth in items:
        n, e, t = (x.detach().float().cpu().reshape(-1) for x in (native, emu, truth))
        nt = t.norm().item() + 1e-30
        dn, de = (n - t).norm().item() / nt, (e - t).norm().item() / nt
        if dn * nt <= absfloor:
            continue
        sn, se = sn + dn * dn, se + de * de
        if dn > floor and dn / (de + 1e-30) > worst[0]:
            worst = (dn / (de + 1e-30), name, dn, de)
        assert dn <= each * de + floor, f"{what} / {name}: native-vs-fp32 {dn:.3e}, oracle(emu)-vs-fp32 {de:.3e}"
    print(f"budget {what}: rms over {len(items)} tensors  native-vs-fp32 {sn ** 0.5:.3e}   oracle(emu)-vs-fp32 {se ** 0.5:.3e}"
          f"   worst single tensor above the floor: x{worst[0]:.2f} ({worst[1]}: {worst[2]:.2e} vs {worst[3]:.2e})")
    assert sn ** 0.5 <= total * se ** 0.5 + floor, f"{what}: aggregate {sn ** 0.5:.3e} vs {se ** 0.5:.3e}"


@pytest.fixture(scope="module")
def setup():
    assert torch.cuda.is_available()
    from vla_adapter_amd import engine as E, synthetic as S
    cfg = E.tiny_config()
    W = S.make_weights(cfg, DEV, seed=3, std=0.05)
    batch = S.make_batch(cfg, 3, DEV, seed=4, P=20, ragged=True)
    eng = E.VLAEngine(cfg, W, DEV)
    return cfg, W, batch, eng


def _oracle_run(cfg, W, batch, noise, emu, head_block_count):
    OW = oracle_weights(W)
    leaf = lambda d: {k: v.clone().requires_grad_(True) for k, v in d.items()}
    OW["head"], OW["proprio"] = leaf(OW["head"]), leaf(OW["proprio"])
    OW["action_queries"] = OW["action_queries"].clone().requires_grad_(True)
    cb = {k: v.cpu() for k, v in batch.items()}
    cb["pixel_values"] = cb["pixel_values"].float()
    cb["proprio"] = cb["proprio"].to(BF).float()           # action_heads.py:53
    out = O.vla_forward(cb, OW, oracle_cfg(cfg), emu=emu, noise=noise)
    return out, OW


def test_counts_are_64(setup):
    cfg, W, batch, eng = setup
    from vla_adapter_amd import ops
    for shift in (0, 1):
        _, _, cnt = ops.action_mask(batch["labels"], shift)
        assert cnt.cpu().tolist() == [64] * batch["labels"].shape[0]


@pytest.mark.parametrize("use_noise", [False, True])
def test_forward_parity(setup, use_noise):
    cfg, W, batch, eng = setup
    noise = None
    if use_noise:
        noise = (torch.randn(cfg.chunk, cfg.action_dim * cfg.llm.d, generator=torch.Generator().manual_seed(9)) * 0.02).to(BF).float()
    pred = eng.forward(batch, noise.to(DEV) if use_noise else None)
    torch.cuda.synchronize()
    out, _ = _oracle_run(cfg, W, batch, noise, True, cfg.num_blocks)
    tru, _ = _oracle_run(cfg, W, batch, noise, False, cfg.num_blocks)
    n = cfg.llm.n_layers
    Np = cfg.n_patches
    budget(eng.llm.HS[0][:, 1:Np + 1], out["patches"], tru["patches"], "projected patches")
    for i in range(n + 1):
        budget(eng.llm.HS[i], out["hidden_states"][i], tru["hidden_states"][i], f"hidden_states[{i}]")
    budget(pred, out["pred"], tru["pred"], "predicted actions")
    loss3, _ = __import__("vla_adapter_amd.ops", fromlist=["ops"]).l1_loss(pred, batch["actions"].to(BF), False)
    assert abs(loss3[0].item() - tru["loss"].item()) <= 1.25 * abs(out["loss"].item() - tru["loss"].item()) + 1e-3 * abs(tru["loss"].item())


def test_backward_and_step_parity(setup):
    cfg, W, batch, eng = setup
    pred = eng.forward(batch, None)
    loss3 = eng.loss_and_backward(pred, batch["actions"])
    torch.cuda.synchronize()
    out, OW = _oracle_run(cfg, W, batch, None, True, cfg.num_blocks)
    tru, TW = _oracle_run(cfg, W, batch, None, False, cfg.num_blocks)
    # L1's gradient is sign(pred - target)/n: a bf16-level difference in pred flips signs and changes the whole backward signal
    # discretely, so all three backward passes are driven by the SAME upstream gradient (the engine's); the L1 kernel itself
    # is pinned in test_kernels_gpu.py.
    from vla_adapter_amd import ops
    _, dpred = ops.l1_loss(pred, batch["actions"].to(BF), True)
    out["pred"].backward(dpred.float().cpu())
    tru["pred"].backward(dpred.float().cpu())
    assert abs(loss3[0].item() - tru["loss"].item()) <= 1.25 * abs(out["loss"].item() - tru["loss"].item()) + 1e-3 * abs(tru["loss"].item())
    g_head = eng.head.named_views(eng.head.P.grad)
    gmax = max(v.grad.norm().item() for v in TW["head"].values() if v.grad is not None)
    fam = [(k, v, OW["head"][k].grad.reshape(v.shape), TW["head"][k].grad.reshape(v.shape)) for k, v in g_head.items() if TW["head"][k].grad is not None]
    fam += [("proprio." + k, v, OW["proprio"][k].grad, TW["proprio"][k].grad) for k, v in eng.head.proprio_views(eng.head.P.grad).items()]
    budget_family(fam, "head + proprio gradients (end to end)", absfloor=1e-3 * gmax)
    budget(eng.head.P.g("action_queries"), OW["action_queries"].grad, TW["action_queries"].grad, "action_queries gradient (through the frozen LLM)", factor=1.5)
    # optimiser step: bit-exact AdamW on the engine's own gradients
    P = eng.head.P
    p0, g0 = P.data.float().cpu().clone(), P.grad.float().cpu().clone()
    eng.optimizer_step(5e-4)
    torch.cuda.synchronize()
    pr, mr, vr = O.adamw_step(p0, g0, torch.zeros_like(p0), torch.zeros_like(p0), 1, 5e-4, emu=True)
    assert torch.equal(P.data.float().cpu(), pr) and torch.equal(P.m.float().cpu(), mr) and torch.equal(P.v.float().cpu(), vr)


def test_training_reduces_loss(setup):
    """A few native steps on a fixed batch must drive the L1 loss down (end-to-end sanity of fwd+bwd+AdamW)."""
    cfg, W, batch, _ = setup
    from vla_adapter_amd import engine as E
    eng = E.VLAEngine(cfg, W, DEV)
    losses = [eng.train_step(batch, 2e-3)[0].item() for _ in range(12)]
    assert losses[-1] < 0.8 * losses[0], losses


def test_graph_replay_matches_eager(setup):
    """The hipGraph-captured step (what bench.py times) must reproduce the eager step: same loss, same gradients
    (up to the fp32 atomic-add order of the bias / LayerNorm / gate reductions) and the same AdamW update."""
    cfg, W, batch, _ = setup
    from vla_adapter_amd import engine as E
    e1, e2 = E.VLAEngine(cfg, W, DEV), E.VLAEngine(cfg, W, DEV)
    l1 = e1.train_step(batch, 1e-3)[0].item()
    g1, p1 = e1.head.P.grad.float().cpu().clone(), e1.head.P.data.float().cpu().clone()
    e2.capture(batch, None)
    l2 = e2.train_step_graphed(1e-3)[0].item()
    p_before = e2.head.P.data.clone()
    e2.flush()                                   # the graphed step leaves its AdamW update pending until the next step / flush
    torch.cuda.synchronize()
    assert not torch.equal(p_before, e2.head.P.data)
    g2, p2 = e2.head.P.grad.float().cpu(), e2.head.P.data.float().cpu()
    assert abs(l1 - l2) < 1e-6
    assert (g1 - g2).norm() <= 2e-3 * g1.norm()
    assert (p1 - p2).norm() <= 1e-3 * p1.norm()
    # replay again: the graph must keep working on updated weights (transposes are part of the graph)
    l3 = e2.train_step_graphed(1e-3)[0].item()
    l1b = e1.train_step(batch, 1e-3)[0].item()
    assert abs(l3 - l1b) <= 2e-2 * abs(l1b)


@pytest.mark.parametrize("pro", [True, False])
def test_head_only_parity_identical_inputs(setup, pro):
    """Head forward/backward on IDENTICAL hidden states (N(0,1) inputs: a deliberately harsh, peaky-softmax regime), for
    MLPResNetBlock_Pro and for the original MLPResNetBlock (shared k/v projections, no RoPE: action_heads.py:168-283)."""
    cfg, W, batch, _ = setup
    from vla_adapter_amd import ops, engine as E, synthetic as S
    if not pro:
        cfg = E.tiny_config()
        cfg.pro = False
        W = S.make_weights(cfg, DEV, seed=3, std=0.05)
        assert "model.mlp_resnet_blocks.0.k_proj.weight" in W["head"] and "model.mlp_resnet_blocks.0.k_task.weight" not in W["head"]
    eng = E.VLAEngine(cfg, W, DEV)        # fresh parameters (the shared engine has taken an optimiser step)
    B, L = batch["input_ids"].shape
    Np, D, nb = cfg.n_patches, cfg.llm.d, cfg.num_blocks
    S = L + Np
    g = torch.Generator().manual_seed(21)
    HS = torch.randn(nb + 1, B, S, D, generator=g).to(BF)
    _, pos1, _ = ops.action_mask(batch["labels"], 1)
    head = eng.head
    pred = head.forward(HS.to(DEV), pos1, batch["proprio"], Np, None)
    dpred = (torch.randn(B, cfg.chunk, cfg.action_dim, generator=g) * 0.01).to(BF)
    dHS = torch.zeros(nb + 1, B, S, D, dtype=BF, device=DEV)
    head.backward(dpred.to(DEV), dHS)
    torch.cuda.synchronize()
    f = lambda sd: {k: v.float().cpu().clone().requires_grad_(True) for k, v in sd.items()}
    res = {}
    for emu in (True, False):
        hp, pp = f(W["head"]), f(W["proprio"])
        hs = HS.float().requires_grad_(True)
        mlhs = O.regroup_hidden_states([hs[i] for i in range(nb + 1)], batch["labels"].cpu(), Np)
        ref = O.head_predict_action(mlhs, batch["proprio"].cpu().to(BF).float(), hp, pp, Np, pro, None, emu, nb)
        ref.backward(dpred.float())
        res[emu] = (ref.detach(), hp, pp, hs.grad)
    (re_, hpe, ppe, dhe), (rt, hpt, ppt, dht) = res[True], res[False]
    budget(pred, re_, rt, f"head-only actions (pro={pro})")
    gmax = max(v.grad.norm().item() for v in hpt.values() if v.grad is not None)
    fam = [(k, v, hpe[k].grad.reshape(v.shape), hpt[k].grad.reshape(v.shape)) for k, v in head.named_views(head.P.grad).items() if hpt[k].grad is not None]
    fam += [("proprio." + k, v, ppe[k].grad, ppt[k].grad) for k, v in head.proprio_views(head.P.grad).items()]
    budget_family(fam, f"head-only gradients, identical inputs (pro={pro})", absfloor=1e-3 * gmax)
    budget(dHS[1:], dhe[1:], dht[1:], "head-only hidden-state gradients")


@pytest.mark.parametrize("pd", [7, 14])
def test_head_backward_other_proprio_widths(setup, pd):
    """PROPRIO_DIM is 7 on BRIDGE and 14 on ALOHA (prismatic/vla/constants.py:38-52): neither is a multiple of the TN product's
    8-element column chunk, so the proprio fc1 weight gradient contracts against the 64-column padded input."""
    cfg0, _, batch, _ = setup
    from vla_adapter_amd import ops, engine as E, synthetic as S
    cfg = E.tiny_config()
    cfg.proprio_dim = pd
    W = S.make_weights(cfg, DEV, seed=3, std=0.05)
    eng = E.VLAEngine(cfg, W, DEV)
    B, L = batch["input_ids"].shape
    Np, D, nb = cfg.n_patches, cfg.llm.d, cfg.num_blocks
    Sq = L + Np
    g = torch.Generator().manual_seed(22)
    HS = (torch.randn(nb + 1, B, Sq, D, generator=g) * 0.5).to(BF)
    proprio = (torch.rand(B, pd, generator=g) * 2 - 1).to(DEV)
    _, pos1, _ = ops.action_mask(batch["labels"], 1)
    head = eng.head
    pred = head.forward(HS.to(DEV), pos1, proprio, Np, None)
    dpred = (torch.randn(B, cfg.chunk, cfg.action_dim, generator=g) * 0.01).to(BF)
    dHS = torch.zeros(nb + 1, B, Sq, D, dtype=BF, device=DEV)
    head.backward(dpred.to(DEV), dHS)
    torch.cuda.synchronize()
    f = lambda sd: {k: v.float().cpu().clone().requires_grad_(True) for k, v in sd.items()}
    res = {}
    for emu in (True, False):
        hp, pp = f(W["head"]), f(W["proprio"])
        mlhs = O.regroup_hidden_states([HS[i].float() for i in range(nb + 1)], batch["labels"].cpu(), Np)
        ref = O.head_predict_action(mlhs, proprio.cpu().to(BF).float(), hp, pp, Np, True, None, emu, nb)
        ref.backward(dpred.float())
        res[emu] = (ref.detach(), pp)
    (re_, ppe), (rt, ppt) = res[True], res[False]
    budget(pred, re_, rt, f"actions, proprio_dim {pd}")
    gv = head.proprio_views(head.P.grad)
    assert tuple(gv["fc1.weight"].shape) == (D, pd)
    budget_family([("proprio." + k, v, ppe[k].grad, ppt[k].grad) for k, v in gv.items()], f"proprio projector gradients, proprio_dim {pd}")


def test_llm_only_backward_identical_inputs(setup):
    """Frozen-LLM dX on identical inputs_embeds and an arbitrary gradient on every hidden state."""
    cfg, W, batch, eng = setup
    c = cfg.llm
    B, S, D, n = 2, 72, c.d, c.n_layers
    g = torch.Generator().manual_seed(22)
    x = torch.randn(B, S, D, generator=g).to(BF)
    km = torch.ones(B, S, dtype=torch.bool)
    km[1, 60:] = False
    dH = (torch.randn(n + 1, B, S, D, generator=g) * 0.01).to(BF)
    llm = eng.llm
    llm._alloc(B, S)
    llm.HS[0].copy_(x.to(DEV))
    llm.forward(B, S, km.to(torch.uint8).to(DEV))
    dx = llm.backward(dH.to(DEV), B, S)
    torch.cuda.synchronize()
    res = {}
    for emu in (True, False):
        xr = x.float().requires_grad_(True)
        hs = O.qwen2_forward(xr, km, {k: v.float().cpu() for k, v in W["llm"].items()}, c.as_oracle(), emu)
        sum((hs[i] * dH[i].float()).sum() for i in range(1, n + 1)).backward()
        res[emu] = ([h.detach() for h in hs], xr.grad)
    valid = km[:, :, None].expand(B, S, D)          # padded positions hold don't-care values in every implementation
    for i in range(1, n + 1):
        budget(llm.HS[i].float().cpu()[valid], res[True][0][i][valid], res[False][0][i][valid], f"LLM-only hidden_states[{i}]")
    budget(dx.float().cpu()[valid], res[True][1][valid], res[False][1][valid], "LLM-only dX")
    llm._buf_key = None      # other tests use a different (B, S)


def test_pipelined_eager_matches_sequential(setup):
    """Two-stream schedule (head trailing / leading the LLM by one layer) == sequential single-stream step."""
    cfg, W, batch, _ = setup
    from vla_adapter_amd import engine as E
    e1, e2 = E.VLAEngine(cfg, W, DEV), E.VLAEngine(cfg, W, DEV)
    for it in range(2):
        l1 = e1.train_step(batch, 1e-3)[0].item()
        l2 = e2.train_step_pipelined(batch, 1e-3)[0].item()
        torch.cuda.synchronize()
        assert abs(l1 - l2) <= 1e-6 + 2e-2 * it * abs(l1)
        g1, g2 = e1.head.P.grad.float().cpu(), e2.head.P.grad.float().cpu()
        assert (g1 - g2).norm() <= (2e-3 + 3e-2 * it) * g1.norm()


def test_live_row_backward_equals_full_backward():
    """Adapter-only fine-tune: the LLM backward restricted to the rows >= row0 (first action-query position rounded
    down to 32) must give the SAME gradients for every trainable tensor as the full-sequence backward the reference's
    autograd performs - and both must match the oracle's autograd gradient of action_queries."""
    from vla_adapter_amd import engine as E, synthetic as S, ops
    cfg = E.tiny_config()
    W = S.make_weights(cfg, DEV, seed=3, std=0.05)
    batch = S.make_batch(cfg, 3, DEV, seed=5, P=56, ragged=True)       # first action query at row >= 16 + 48 = 64
    e_live, e_full = E.VLAEngine(cfg, W, DEV), E.VLAEngine(cfg, W, DEV)
    e_full.full_llm_backward = True
    grads = []
    for e in (e_live, e_full):
        pred = e.forward(batch, None)
        e.loss_and_backward(pred, batch["actions"])
        torch.cuda.synchronize()
        grads.append(e.head.P.grad.float().cpu().clone())
    assert e_live.live_row0() >= 64 and e_full.live_row0() == 0
    assert e_live._dHS.shape[2] == e_live.S - e_live.live_row0() and e_full._dHS.shape[2] == e_full.S
    aq = e_live.head.P.offsets["action_queries"][0]
    assert torch.equal(grads[0][aq:], grads[1][aq:]), "action_queries gradient must not depend on the dead rows"
    # head / proprio grads: identical computation except fp32 atomic orders of the bias / LayerNorm reductions
    assert (grads[0][:aq] - grads[1][:aq]).norm() <= 2e-3 * grads[1][:aq].norm()
    # and against the oracle's full autograd
    _, dpred = ops.l1_loss(e_live.head.pred.view(3, cfg.chunk, cfg.action_dim), batch["actions"].to(BF), True)
    g = {}
    for emu in (True, False):
        out, OW = _oracle_run(cfg, W, batch, None, emu, cfg.num_blocks)
        out["pred"].backward(dpred.float().cpu())
        g[emu] = OW["action_queries"].grad
    budget(e_live.head.P.g("action_queries"), g[True], g[False], "action_queries gradient, live-row backward vs full autograd", factor=1.5)


def test_graph_replay_guards_frozen_row_window():
    """A captured step freezes row0; replaying it on a batch whose action queries start earlier must poison the loss."""
    from vla_adapter_amd import engine as E, synthetic as S
    cfg = E.tiny_config()
    W = S.make_weights(cfg, DEV, seed=3, std=0.05)
    batch = S.make_batch(cfg, 2, DEV, seed=6, P=56)
    eng = E.VLAEngine(cfg, W, DEV)
    eng.capture(batch, None)
    assert eng._row0 == 64
    assert torch.isfinite(eng.train_step_graphed(1e-3)).all()
    short = S.make_batch(cfg, 2, DEV, seed=7, P=56)
    lab, ids = short["labels"], short["input_ids"]          # same shapes, action block moved 30 tokens earlier in row 1
    lab[1], ids[1] = torch.roll(lab[1], -30), torch.roll(ids[1], -30)
    for k in batch:
        batch[k].copy_(short[k])
    assert torch.isnan(eng.train_step_graphed(1e-3)).all()


def test_graphed_vision_lead_uses_staged_pixels():
    """Captured mode runs the vision stage of step k+1 during step k on the pixels handed to stage_next_pixels():
    two graphed steps on batches A then B must reproduce two eager steps on A then B."""
    from vla_adapter_amd import engine as E, synthetic as S
    cfg = E.tiny_config()
    W = S.make_weights(cfg, DEV, seed=3, std=0.05)
    A, Bb = S.make_batch(cfg, 2, DEV, seed=11, P=40), S.make_batch(cfg, 2, DEV, seed=12, P=40)
    e1, e2 = E.VLAEngine(cfg, W, DEV), E.VLAEngine(cfg, W, DEV)
    la = e1.train_step(A, 1e-3)[0].item()
    lb = e1.train_step(Bb, 1e-3)[0].item()
    static = {k: v.clone() for k, v in A.items()}
    e2.capture(static, None)
    e2.stage_next_pixels(Bb["pixel_values"])          # vision of step 2 runs inside step 1
    ga = e2.train_step_graphed(1e-3)[0].item()
    for k in static:                                   # step 2 trains on batch B (its pixels were staged one step ahead)
        static[k].copy_(Bb[k])
    gb = e2.train_step_graphed(1e-3)[0].item()
    e2.flush()
    torch.cuda.synchronize()
    assert abs(ga - la) < 1e-6, (ga, la)
    assert abs(gb - lb) <= 2e-2 * abs(lb), (gb, lb)
    assert abs(la - lb) > 1e-3, "the two batches must differ for the check to mean anything"
    assert (e1.head.P.data.float() - e2.head.P.data.float()).norm() <= 2e-3 * e1.head.P.data.float().norm()


def test_original_head_block_end_to_end():
    """use_pro_version=False (MLPResNetBlock, action_heads.py:168-283) through the whole engine: forward parity with the
    oracle, shared k/v gradient = sum over the three segments (checked against autograd), captured step trains."""
    from vla_adapter_amd import engine as E, synthetic as S, ops
    cfg = E.tiny_config()
    cfg.pro = False
    W = S.make_weights(cfg, DEV, seed=13, std=0.05)
    batch = S.make_batch(cfg, 2, DEV, seed=14, P=40)
    eng = E.VLAEngine(cfg, W, DEV)
    pred = eng.forward(batch, None)
    eng.loss_and_backward(pred, batch["actions"])
    torch.cuda.synchronize()
    out, OW = _oracle_run(cfg, W, batch, None, True, cfg.num_blocks)
    tru, TW = _oracle_run(cfg, W, batch, None, False, cfg.num_blocks)
    budget(pred, out["pred"], tru["pred"], "original-block head: actions")
    _, dpred = ops.l1_loss(pred, batch["actions"].to(BF), True)
    out["pred"].backward(dpred.float().cpu())
    tru["pred"].backward(dpred.float().cpu())
    g = eng.head.named_views(eng.head.P.grad)
    gmax = max(v.grad.norm().item() for v in TW["head"].values() if v.grad is not None)
    keys = [f"model.mlp_resnet_blocks.{b}.{n}.weight" for b in (0, 1) for n in ("k_proj", "v_proj", "q_proj", "o_proj")]
    budget_family([(k, g[k], OW["head"][k].grad, TW["head"][k].grad) for k in keys], "original-block head: shared k/v gradients",
                  each=3.0, total=2.0, absfloor=1e-3 * gmax)  # 8 small tensors of a 2-block head: the aggregate itself fluctuates
                                                              # (worst single tensor measured x2.36: block 0's shared v_proj)
    e2 = E.VLAEngine(cfg, W, DEV)
    e2.capture(batch, None)
    losses = [e2.train_step_graphed(2e-3)[0].item() for _ in range(12)]
    e2.flush()
    assert losses[-1] < 0.8 * losses[0], losses


def test_fused_two_backbone_two_image_config():
    """The reference's default layout (modeling_prismatic.py:196-237): DINOv2-style (prefix tokens, LayerScale) + SigLIP-style
    backbones on channel-stacked pixels, two images per sample (all images of a backbone go through it in ONE pass here),
    fused 3-layer projector - forward parity with the oracle, backward parity of action_queries, captured step trains."""
    from vla_adapter_amd import engine as E, synthetic as S, ops
    cfg = E.tiny_fused_config()
    W = S.make_weights(cfg, DEV, seed=17, std=0.05)
    batch = S.make_batch(cfg, 2, DEV, seed=18, P=24, ragged=True)
    assert batch["pixel_values"].shape[1] == 12 and cfg.n_patches == 32 and cfg.vis_dim == 320
    eng = E.VLAEngine(cfg, W, DEV)
    pred = eng.forward(batch, None)
    eng.loss_and_backward(pred, batch["actions"])
    torch.cuda.synchronize()
    out, OW = _oracle_run(cfg, W, batch, None, True, cfg.num_blocks)
    tru, TW = _oracle_run(cfg, W, batch, None, False, cfg.num_blocks)
    budget(eng.llm.HS[0][:, 1:cfg.n_patches + 1], out["patches"], tru["patches"], "fused two-backbone patches")
    budget(pred, out["pred"], tru["pred"], "fused config: actions")
    _, dpred = ops.l1_loss(pred, batch["actions"].to(BF), True)
    out["pred"].backward(dpred.float().cpu())
    tru["pred"].backward(dpred.float().cpu())
    budget(eng.head.P.g("action_queries"), OW["action_queries"].grad, TW["action_queries"].grad, "fused config: action_queries gradient", factor=1.5)
    ef = E.VLAEngine(cfg, W, DEV)
    ef.full_llm_backward = True
    ef.loss_and_backward(ef.forward(batch, None), batch["actions"])
    assert torch.equal(ef.head.P.g("action_queries"), eng.head.P.g("action_queries"))
    e2 = E.VLAEngine(cfg, W, DEV)
    e2.capture(batch, None)
    losses = [e2.train_step_graphed(2e-3)[0].item() for _ in range(10)]
    e2.flush()
    assert losses[-1] < 0.85 * losses[0], losses


def test_engine_forward_backward_random_batches():
    """Seeded sweep over batch size, prompt length and ragged right-padding (key-padding masks of 0-8 trailing keys that cut
    attention tiles at arbitrary rows) on both plumbing configs: forward parity with the oracle, live-row backward
    bit-identical to the full backward."""
    from vla_adapter_amd import engine as E, synthetic as S
    rng = torch.Generator().manual_seed(8642)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=rng))
    for case in range(6):
        cfg = E.tiny_fused_config() if case % 2 else E.tiny_config()
        W = S.make_weights(cfg, DEV, seed=40 + case, std=0.05)
        B, P = ri(1, 4), ri(10, 60)
        batch = S.make_batch(cfg, B, DEV, seed=50 + case, P=P, ragged=True)
        eng = E.VLAEngine(cfg, W, DEV)
        pred = eng.forward(batch, None)
        out, _ = _oracle_run(cfg, W, batch, None, True, cfg.num_blocks)
        tru, _ = _oracle_run(cfg, W, batch, None, False, cfg.num_blocks)
        n = cfg.llm.n_layers
        valid = batch["attention_mask"].cpu()
        Np = cfg.n_patches
        full_valid = torch.cat([torch.ones(B, 1, dtype=torch.bool), torch.ones(B, Np, dtype=torch.bool), valid[:, 1:]], 1)
        for i in range(n + 1):      # rows of padded positions hold don't-care values in both implementations
            budget(eng.llm.HS[i].float().cpu()[full_valid], out["hidden_states"][i].detach()[full_valid], tru["hidden_states"][i].detach()[full_valid],
                   f"case {case} (B{B} P{P} fused={cfg.fused}) hidden_states[{i}]")
        budget(pred, out["pred"], tru["pred"], f"case {case}: actions")
        eng.loss_and_backward(pred, batch["actions"])
        ef = E.VLAEngine(cfg, W, DEV)
        ef.full_llm_backward = True
        ef.loss_and_backward(ef.forward(batch, None), batch["actions"])
        assert torch.equal(ef.head.P.g("action_queries"), eng.head.P.g("action_queries")), f"case {case}: live vs full backward"


def test_full_size_forward_parity_config2():
    """BASELINE configs[1] at FULL size (SigLIP so400m 27 blocks, Qwen2.5-0.5B 24 layers, Pro head 24 blocks), batch 2 with a
    ragged prompt: ViT features, every hidden state and the predicted actions against the oracle (bf16-emulating, ~10 s of
    host time).  Depth accumulates independent bf16 rounding realisations - measured: 1.5e-2 on the projected patches
    (26 ViT blocks), 1.5e-2 -> 2.1e-2 over the 24 LLM layers, 1.1e-2 on the actions, 6e-4 on the loss; bounds are ~1.6x that."""
    from vla_adapter_amd import engine as E, synthetic as S, ops
    cfg = E.config2()
    W = S.make_weights(cfg, DEV, seed=0)
    batch = S.make_batch(cfg, 2, DEV, seed=77, P=32, ragged=True)
    eng = E.VLAEngine(cfg, W, DEV)
    pred = eng.forward(batch, None)
    torch.cuda.synchronize()
    torch.set_num_threads(min(16, torch.get_num_threads() or 16))
    out, _ = _oracle_run(cfg, W, batch, None, True, cfg.num_blocks)
    tru, _ = _oracle_run(cfg, W, batch, None, False, cfg.num_blocks)
    Np, n, B = cfg.n_patches, cfg.llm.n_layers, 2
    budget(eng.llm.HS[0][:, 1:Np + 1], out["patches"], tru["patches"], "full size: projected patches (26 ViT blocks)")
    valid = batch["attention_mask"].cpu()
    fv = torch.cat([torch.ones(B, 1 + Np, dtype=torch.bool), valid[:, 1:]], 1)
    for i in range(n + 1):
        budget(eng.llm.HS[i].float().cpu()[fv], out["hidden_states"][i].detach()[fv], tru["hidden_states"][i].detach()[fv], f"full size: hidden_states[{i}]")
    budget(pred, out["pred"], tru["pred"], "full size: actions")
    l_native, _ = ops.l1_loss(pred, batch["actions"].to(BF), False)
    assert abs(l_native[0].item() - tru["loss"].item()) <= 1.25 * abs(out["loss"].item() - tru["loss"].item()) + 1e-3 * abs(tru["loss"].item())


def test_full_size_backward_config2():
    """Full-size backward: action_queries gradient of the live-row backward bit-identical to the full-sequence backward,
    and within the end-to-end gradient tolerance of the oracle's autograd (driven by the engine's own dpred)."""
    from vla_adapter_amd import engine as E, synthetic as S, ops
    cfg = E.config2()
    W = S.make_weights(cfg, DEV, seed=0)
    batch = S.make_batch(cfg, 2, DEV, seed=78, P=32, ragged=True)
    eng = E.VLAEngine(cfg, W, DEV)
    pred = eng.forward(batch, None)
    eng.loss_and_backward(pred, batch["actions"])
    g_live = eng.head.P.grad.clone()
    assert eng.live_row0() == 256                     # ragged prompts 24..32: first action query >= row 280 -> window from 256
    eng.full_llm_backward = True
    eng.loss_and_backward(eng.forward(batch, None), batch["actions"])
    aq = eng.head.P.offsets["action_queries"][0]
    assert torch.equal(g_live[aq:], eng.head.P.grad[aq:]), "full-size: action_queries gradient must not depend on the dead rows"
    assert (g_live[:aq].float() - eng.head.P.grad[:aq].float()).norm() <= 2e-3 * eng.head.P.grad[:aq].float().norm()
    _, dpred = ops.l1_loss(eng.head.pred.view(2, cfg.chunk, cfg.action_dim), batch["actions"].to(BF), True)
    G = {}
    keys = ("model.fc2.weight", "model.mlp_resnet_blocks.23.o_proj.weight", "model.mlp_resnet_blocks.12.v_task.weight",
            "model.mlp_resnet_blocks.0.k_task.weight", "model.mlp_resnet_blocks.0.ffn.1.weight", "model.mlp_resnet_blocks.12.ffn.0.weight",
            "model.mlp_resnet_blocks.23.q_proj.bias")
    for emu in (True, False):
        out, OW = _oracle_run(cfg, W, batch, None, emu, cfg.num_blocks)
        out["pred"].backward(dpred.float().cpu())
        G[emu] = (OW["action_queries"].grad, {k: OW["head"][k].grad for k in keys})
    budget(g_live[aq:].view(64, -1), G[True][0], G[False][0], "full size: action_queries gradient through 24 frozen layers", factor=1.5)
    nv = eng.head.named_views(g_live)
    budget_family([(k, nv[k], G[True][1][k].reshape(nv[k].shape), G[False][1][k].reshape(nv[k].shape)) for k in keys], "full size: head gradients")


@pytest.mark.parametrize("B,P,ragged", [(3, 37, True), (1, 12, False), (5, 64, True), (9, 33, True), (8, 20, False)])
def test_captured_step_odd_shapes_match_eager(B, P, ragged):
    """Captured (segment graphs, vision lead, deferred update; from batch 8 on the LLM forward as two half-batch pipelines
    on two streams) vs eager sequential step over three steps at odd batch sizes / prompt lengths: same losses (first step
    bit-equal, later steps within the bf16 drift of the updates)."""
    from vla_adapter_amd import engine as E, synthetic as S
    cfg = E.tiny_config()
    W = S.make_weights(cfg, DEV, seed=61, std=0.05)
    batch = S.make_batch(cfg, B, DEV, seed=62, P=P, ragged=ragged)
    e1, e2 = E.VLAEngine(cfg, W, DEV), E.VLAEngine(cfg, W, DEV)
    eager = [e1.train_step(batch, 1e-3)[0].item() for _ in range(3)]
    e2.capture({k: v.clone() for k, v in batch.items()}, None)
    graphed = [e2.train_step_graphed(1e-3)[0].item() for _ in range(3)]
    e2.flush()
    torch.cuda.synchronize()
    assert abs(eager[0] - graphed[0]) < 1e-6, (eager, graphed)
    for a, b in zip(eager[1:], graphed[1:]):
        assert abs(a - b) <= 2e-2 * abs(a), (eager, graphed)
    assert (e1.head.P.data.float() - e2.head.P.data.float()).norm() <= 3e-3 * e1.head.P.data.float().norm()


def test_batch32_config2_matches_batch2_and_trains(monkeypatch):
    """BASELINE configs[1] at the METRIC's shape, batch 32 (two 16-sample LLM pipelines, the 5.3 GB pre-activation buffer, the
    256x256 GEMM tiles): every op of the forward is sample-wise and every single-pass GEMM accumulates each output element over
    K in one fixed order whatever the tile (tests/test_kernels_gpu.py::test_gemm256_bit_identical_to_128_tiles), so samples
    0-1 of the B = 32 forward must equal the B = 2 forward BIT FOR BIT - with split-K off for the small run (at M = 512 the
    long-K GEMMs would otherwise meet their K slices in fp32 planes: another summation order); and three captured steps must
    keep a finite loss."""
    monkeypatch.setenv("VLA_NO_SPLITK", "1")
    from vla_adapter_amd import engine as E, synthetic as S
    cfg = E.config2()
    W = S.make_weights(cfg, DEV, seed=0)
    big = S.make_batch(cfg, 32, DEV, seed=91, P=32, ragged=True)
    big["pixel_values"] = big["pixel_values"].to(BF)
    two = {k: v[:2].contiguous() for k, v in big.items()}
    e32, e2 = E.VLAEngine(cfg, W, DEV), E.VLAEngine(cfg, W, DEV)
    p32 = e32.forward(big, None)
    p2 = e2.forward(two, None)
    torch.cuda.synchronize()
    n = cfg.llm.n_layers
    for i in range(n + 1):
        assert torch.equal(e32.llm.HS[i][:2], e2.llm.HS[i]), f"hidden_states[{i}]: batch-32 rows differ from the batch-2 run"
    assert torch.equal(p32[:2], p2), "actions of samples 0-1"
    del e2
    noise = (torch.randn(cfg.chunk, cfg.action_dim * cfg.llm.d, device=DEV) * 0.02).to(BF)
    e32.capture({k: v.clone() for k, v in big.items()}, noise)
    losses = [e32.train_step_graphed(5e-4)[0].item() for _ in range(3)]
    e32.flush()
    torch.cuda.synchronize()
    assert all(l == l and abs(l) < 1e2 for l in losses), losses      # finite and bounded (AdamW's first unit-size steps overshoot on random data)


def test_full_size_dinov2_backbone_forward_budget():
    """The DINOv2-L/reg4 branch at FULL size (d 1024, 23 useful blocks, cls + 4 register tokens, LayerScale folded into the
    projections) next to SigLIP so400m in the reference's default fused two-image layout, batch 1: projected patches within
    the fp32-truth budget (this branch was parity-tested at plumbing size only)."""
    from vla_adapter_amd import engine as E, synthetic as S
    cfg = E.VLACfg(vit=[E.DINOV2_L_REG4, E.SIGLIP_SO400M], n_img=2)
    W = S.make_weights(cfg, DEV, seed=2)
    batch = S.make_batch(cfg, 1, DEV, seed=79, P=32)
    eng = E.VLAEngine(cfg, W, DEV)
    eng._vision(batch)
    torch.cuda.synchronize()
    torch.set_num_threads(min(16, torch.get_num_threads() or 16))
    px = batch["pixel_values"].float().cpu()
    res = {}
    for emu in (True, False):
        feats = []
        for im in range(2):
            ch = px[:, im * 6:(im + 1) * 6]
            f = [O.vit_forward(ch[:, 3 * j:3 * j + 3], cpu_f32(W["vit"][j]), cfg.vit[j].as_oracle(), emu) for j in range(2)]
            feats.append(torch.cat(f, dim=2))
        res[emu] = (torch.cat(feats, dim=1), O.projector(torch.cat(feats, dim=1), cpu_f32(W["proj"]), True, emu))
    d0 = cfg.vit[0].d
    budget(eng.feats[:, :, :d0], res[True][0][:, :, :d0], res[False][0][:, :, :d0], "full size: DINOv2-L features (23 blocks, LayerScale folded)")
    budget(eng.feats[:, :, d0:], res[True][0][:, :, d0:], res[False][0][:, :, d0:], "full size: SigLIP features in the fused layout")
    budget(eng.patches, res[True][1], res[False][1], "full size: fused 3-layer projector output")


def test_full_size_full_finetune_step_config4():
    """BASELINE configs[3] at full size (SigLIP so400m + Qwen2.5-0.5B, every parameter trainable), batch 2: one eager step and two
    captured steps - finite loss, every gradient region finite and non-zero, parameters move."""
    from vla_adapter_amd import engine as E, synthetic as S
    from vla_adapter_amd.full_finetune import FullFinetune
    cfg = E.config2()
    W = S.make_weights(cfg, DEV, seed=0)
    batch = S.make_batch(cfg, 2, DEV, seed=80, P=32, ragged=True)
    batch["pixel_values"] = batch["pixel_values"].to(BF)
    ft = FullFinetune(E.VLAEngine(cfg, W, DEV))
    p0 = ft.P.data[:4096].clone()
    l0 = ft.train_step(batch, 1e-4)[0].item()
    torch.cuda.synchronize()
    g = ft.P.grad.float()
    assert l0 == l0 and torch.isfinite(g).all()
    for name in ("llm.0.wqkv", "llm.23.wd", "vit0.0.wqkv", "vit0.25.w2", "proj.fc1.weight", "llm.norm", "vit0.pos", "llm.11.n2", "vit0.12.b1"):
        off, shape = ft.P.offsets[name]
        n = 1
        for d in shape:
            n *= d
        assert g[off:off + n].abs().max().item() > 0, name
    assert not torch.equal(p0, ft.P.data[:4096])
    ft.capture(batch, None)
    ls = [ft.train_step_graphed(1e-4)[0].item() for _ in range(2)]
    torch.cuda.synchronize()
    assert all(l == l and abs(l) < 1e2 for l in ls), ls


def test_adapter_step_skips_llm_layers_above_the_heads_last_block():
    """Qwen2.5-1.5B has 28 layers, the head 24 blocks: hidden_states[25..28] reach neither the loss nor the actions.  The adapter-only
    step and predict() run the first n_act layers only (round 4; the LoRA / full trainers did since round 3) - every gradient must be
    BIT-IDENTICAL to the run that computes all layers, the captured step must work, and forward_vlm() - the API twin that returns
    every hidden state (modeling_prismatic.py:680-686) - still fills them all.  3 layers, 2 blocks."""
    from vla_adapter_amd import engine as E, synthetic as S
    cfg = E.tiny_config()
    cfg.llm = E.LLMCfg(256, 3, 4, 2, 64, 512, 1e-6, 1e6, 1024)
    cfg.num_blocks = 2
    W = S.make_weights(cfg, DEV, seed=3, std=0.05)
    batch = S.make_batch(cfg, 3, DEV, seed=4, P=20, ragged=True)
    grads, preds = {}, {}
    for n_act in (2, 3):
        eng = E.VLAEngine(cfg, W, DEV)
        assert eng.n_act == 2
        eng.n_act = n_act
        pred = eng.forward(batch, None, for_training=True)
        eng.loss_and_backward(pred, batch["actions"])
        torch.cuda.synchronize()
        grads[n_act], preds[n_act] = eng.head.P.grad.clone(), pred.clone()
    assert torch.equal(preds[2], preds[3]) and torch.equal(grads[2], grads[3]), "dead layers must contribute exactly nothing"
    assert grads[2].float().abs().max().item() > 0
    eng = E.VLAEngine(cfg, W, DEV)
    eng.forward_vlm(batch)                                  # API twin: all n + 1 hidden states
    torch.cuda.synchronize()
    out, _ = _oracle_run(cfg, W, batch, None, True, cfg.num_blocks)
    tru, _ = _oracle_run(cfg, W, batch, None, False, cfg.num_blocks)
    for i in range(cfg.llm.n_layers + 1):
        budget(eng.llm.HS[i], out["hidden_states"][i], tru["hidden_states"][i], f"3-layer LLM under a 2-block head: hidden_states[{i}]")
    a = eng.predict({k: v for k, v in batch.items()})
    torch.cuda.synchronize()
    budget(a, out["pred"].detach(), tru["pred"].detach(), "predict() on n_act layers")
    eng.capture({k: v.clone() for k, v in batch.items()}, None)
    ls = [eng.train_step_graphed(5e-4)[0].item() for _ in range(4)]
    eng.flush()
    torch.cuda.synchronize()
    assert all(l == l for l in ls) and ls[-1] < ls[0], ls


def test_config4_per_gpu_shape_batch16_captured_steps(monkeypatch):
    """BASELINE configs[3] at the per-GPU shape it names - full-backbone unfreeze, bf16, batch 16 per GPU - as the captured three-stream
    step with AdamW range by range under the backward (the form bench.py --mode full times; the eight-GPU exchange itself is rehearsed
    on gloo in tests/test_ddp_gpu.py and unmeasured on hardware): four steps, finite and falling loss, every parameter group fed, and
    the batch-16 rows 0-1 of the first step's forward equal to a batch-2 forward of the same two samples bit for bit (split-K off: at
    M = 704 the long-K products would otherwise meet their K slices in fp32 planes - another summation order)."""
    monkeypatch.setenv("VLA_NO_SPLITK", "1")
    from vla_adapter_amd import engine as E, synthetic as S
    from vla_adapter_amd.full_finetune import FullFinetune
    cfg = E.config2()
    W = S.make_weights(cfg, DEV, seed=0)
    batch = S.make_batch(cfg, 16, DEV, seed=81, P=32)
    batch["pixel_values"] = batch["pixel_values"].to(BF)
    two = {k: v[:2].contiguous() for k, v in batch.items()}
    e2 = E.VLAEngine(cfg, W, DEV)
    f2 = FullFinetune(e2)
    p2 = f2.forward(two, None).clone()
    h2 = e2.llm.HS[cfg.llm.n_layers].clone()
    del f2, e2
    eng = E.VLAEngine(cfg, W, DEV)
    ft = FullFinetune(eng)
    p16 = ft.forward(batch, None)
    torch.cuda.synchronize()
    assert torch.equal(p16[:2], p2) and torch.equal(eng.llm.HS[cfg.llm.n_layers][:2], h2), "batch-16 rows differ from the batch-2 run"
    p0 = ft.P.data.clone()
    ft.capture(batch, None)
    ls = [ft.train_step_graphed(2e-5)[0].item() for _ in range(4)]
    torch.cuda.synchronize()
    assert all(l == l and abs(l) < 1e2 for l in ls) and ls[-1] < ls[0], ls
    g = ft.P.grad.float()
    assert torch.isfinite(g).all()
    for name in ("llm.0.wqkv", "llm.23.wd", "vit0.0.wqkv", "vit0.25.w2", "proj.fc1.weight", "llm.embed", "vit0.pos", "llm.11.n2", "vit0.12.b1"):
        off, shape = ft.P.offsets[name]
        n = 1
        for d in shape:
            n *= d
        assert g[off:off + n].abs().max().item() > 0, name
        # matrices must have moved; vectors near 1.0 (norm weights: one bf16 ulp = 7.8e-3) do not see lr = 2e-5 - in the reference's
        # bf16 optimizer either; the token table moves on the rows of the batch's ids only
        if len(shape) >= 2 and name != "llm.embed":
            assert not torch.equal(ft.P.data[off:off + n], p0[off:off + n]), name


def test_qwen25_15b_layer_geometry_forward_backward():
    """Qwen2.5-1.5B's layer geometry (BASELINE configs[4]: d 1536, 12 x 128 heads, 2 KV heads, MLP 8960) at two layers: head
    dim 128 takes the unfused-RoPE projection, the 128-wide attention forward / backward kernels, and the head's first LayerNorm
    spans 7 x 1536 columns.  Same error-budget criteria as the 0.5B tests."""
    from vla_adapter_amd import engine as E, synthetic as S, ops
    cfg = E.qwen15b_geometry_config(2)
    W = S.make_weights(cfg, DEV, seed=31, std=0.03)
    batch = S.make_batch(cfg, 2, DEV, seed=32, P=20, ragged=True)
    eng = E.VLAEngine(cfg, W, DEV)
    pred = eng.forward(batch, None)
    loss3 = eng.loss_and_backward(pred, batch["actions"])
    torch.cuda.synchronize()
    out, OW = _oracle_run(cfg, W, batch, None, True, cfg.num_blocks)
    tru, TW = _oracle_run(cfg, W, batch, None, False, cfg.num_blocks)
    for i in range(cfg.llm.n_layers + 1):
        budget(eng.llm.HS[i], out["hidden_states"][i], tru["hidden_states"][i], f"1.5B geometry hidden_states[{i}]")
    budget(pred, out["pred"], tru["pred"], "1.5B geometry predicted actions")
    assert torch.isfinite(loss3).all()
    _, dpred = ops.l1_loss(pred, batch["actions"].to(BF), True)
    out["pred"].backward(dpred.float().cpu())
    tru["pred"].backward(dpred.float().cpu())
    g_head = eng.head.named_views(eng.head.P.grad)
    gmax = max(v.grad.norm().item() for v in TW["head"].values() if v.grad is not None)
    fam = [(k, v, OW["head"][k].grad.reshape(v.shape), TW["head"][k].grad.reshape(v.shape)) for k, v in g_head.items() if TW["head"][k].grad is not None]
    budget_family(fam, "1.5B geometry head gradients", absfloor=1e-3 * gmax)
    budget(eng.head.P.g("action_queries"), OW["action_queries"].grad, TW["action_queries"].grad,
           "1.5B geometry action_queries gradient (through the frozen LLM)", factor=1.5)


def test_fp8_frozen_forward_matches_its_emulation_and_trains():
    """Opt-in fp8 weight path (BASELINE configs[4]; no reference code: PARITY UNPINNED).  The engine with
    enable_fp8_frozen() against the oracle whose ViT qkv / fc1 and LLM q|k|v / gate|up Linears fake-quantise both operands the
    same way (oracle.FP8 registry): hidden states and actions within the bf16 path's own error budget of THAT function; the
    deviation from the bf16 engine is the quantisation error (a few per cent) and is printed; the adapter still trains."""
    from vla_adapter_amd import engine as E, synthetic as S
    cfg = E.tiny_config()
    # K of the quantised products must be a multiple of 128: ViT-T (192) does not qualify, use a 256-wide ViT
    cfg.vit = [E.ViTCfg(256, 3, 4, 1024, 14, 56, 0, False)]
    W = S.make_weights(cfg, DEV, seed=41, std=0.05)
    batch = S.make_batch(cfg, 3, DEV, seed=42, P=20, ragged=True)
    ref_eng = E.VLAEngine(cfg, W, DEV)
    pred_bf16 = ref_eng.forward(batch, None).clone()
    eng = E.VLAEngine(cfg, W, DEV)
    eng.enable_fp8_frozen()
    pred = eng.forward(batch, None)
    torch.cuda.synchronize()
    OWs = []
    for emu in (True, False):
        OW = oracle_weights(W)
        O.FP8.clear()
        for i in range(len(eng.vits[0].blocks)):
            O.FP8.add(id(OW["vit"][0][f"blocks.{i}.attn.qkv.weight"]))
            O.FP8.add(id(OW["vit"][0][f"blocks.{i}.mlp.fc1.weight"]))
        for i in range(cfg.llm.n_layers):
            for n in ("self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj", "mlp.gate_proj", "mlp.up_proj"):
                O.FP8.add(id(OW["llm"][f"layers.{i}.{n}.weight"]))
        cb = {k: v.cpu() for k, v in batch.items()}
        cb["pixel_values"] = cb["pixel_values"].float()
        cb["proprio"] = cb["proprio"].to(BF).float()
        OWs.append(O.vla_forward(cb, OW, oracle_cfg(cfg), emu=emu, noise=None))
    O.FP8.clear()
    out, tru = OWs
    # per-channel weight scales: the native build quantises the FUSED q|k|v and gate|up weights row by row, which is the same
    # as quantising the separate projections row by row; the input rows are shared
    for i in range(cfg.llm.n_layers + 1):
        budget(eng.llm.HS[i], out["hidden_states"][i], tru["hidden_states"][i], f"fp8 path hidden_states[{i}]", factor=1.5, floor=2e-3)
    budget(pred, out["pred"], tru["pred"], "fp8 path predicted actions", factor=1.5, floor=2e-3)
    dev = rel(pred, pred_bf16)
    print(f"fp8 frozen forward vs bf16 forward: predicted actions differ by {dev:.3e} (relative L2)")
    assert dev < 0.15
    losses = [eng.train_step(batch, 1e-3)[0].item() for _ in range(10)]
    torch.cuda.synchronize()
    assert all(map(lambda v: v == v, losses)) and losses[-1] < 0.9 * losses[0], losses
