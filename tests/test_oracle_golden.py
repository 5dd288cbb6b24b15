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
    return {k: torch.from_numpy(z[k]) for k in z.files}


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
