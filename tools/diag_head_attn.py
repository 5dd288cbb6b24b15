#!/usr/bin/env python3
"""Diagnostic: error of the VALU vs MFMA action-head attention kernels against fp32 autograd of the oracle."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vla_adapter_amd import ops
from oracle import vla_oracle as O
DEV, BF = "cuda", torch.bfloat16
def gen(*s, seed=0, scale=1.0):
    return (torch.randn(*s, generator=torch.Generator().manual_seed(seed)) * scale).to(BF)
rel = lambda a, b: ((a.float().cpu() - b).norm() / (b.norm() + 1e-12)).item()
for (B, Ka, Kt, D, sc) in [(2, 65, 256, 896, 0.3), (3, 65, 16, 256, 1.0), (3, 65, 16, 256, 0.3)]:
    H, T = 8, 8
    dh = D // H
    x3, a2, t2 = gen(B, T, 3 * D, seed=50, scale=sc), gen(B, Ka, 2 * D, seed=51, scale=sc), gen(B, Kt, 2 * D, seed=52, scale=sc)
    gate = torch.tensor([0.7]).to(BF)
    dout = gen(B, T, D, seed=53)
    hd = lambda t, L: t.float().reshape(B, L, H, dh).transpose(1, 2)
    leaf = lambda t, L: hd(t, L).clone().requires_grad_(True)
    q, ks, vs = leaf(x3[:, :, :D], T), leaf(x3[:, :, D:2 * D], T), leaf(x3[:, :, 2 * D:], T)
    ka, va, kt, vt = leaf(a2[:, :, :D], Ka), leaf(a2[:, :, D:], Ka), leaf(t2[:, :, :D], Kt), leaf(t2[:, :, D:], Kt)
    g = gate.float().clone().requires_grad_(True)
    ref = O.head_attention_core(q, [(ks, vs), (ka, va), (kt, vt)], torch.tanh(g), False)
    (ref * hd(dout, T)).sum().backward()
    refe = O.head_attention_core(q.detach(), [(ks.detach(), vs.detach()), (ka.detach(), va.detach()), (kt.detach(), vt.detach())], torch.tanh(g.detach()), True)
    un = lambda t, L: t.transpose(1, 2).reshape(B, L, D)
    for mode in ("VALU", "MFMA"):
        if mode == "VALU": os.environ["VLA_HEAD_ATTN_VALU"] = "1"
        else: os.environ.pop("VLA_HEAD_ATTN_VALU", None)
        dx3, da2, dt2 = x3.to(DEV), a2.to(DEV), t2.to(DEV)
        args = (dx3[:, :, :D], dx3[:, :, D:2 * D], dx3[:, :, 2 * D:], da2[:, :, :D], da2[:, :, D:], dt2[:, :, :D], dt2[:, :, D:])
        out, probs = ops.head_attn_fwd(*args, gate.to(DEV), H)
        g3, ga, gt = torch.zeros_like(dx3), torch.zeros_like(da2), torch.zeros_like(dt2)
        dgate = torch.zeros(1, device=DEV)
        ops.head_attn_bwd(dout.to(DEV), out, *args, gate.to(DEV), probs, dgate, g3[:, :, :D], g3[:, :, D:2 * D], g3[:, :, 2 * D:],
                          ga[:, :, :D], ga[:, :, D:], gt[:, :, :D], gt[:, :, D:], H)
        print(f"B{B} Kt{Kt} D{D} sc{sc} {mode}: out vs emu {rel(out, un(refe, T)):.2e} vs f32 {rel(out, un(ref.detach(), T)):.2e} | dq {rel(g3[:, :, :D], un(q.grad, T)):.2e} dks {rel(g3[:, :, D:2*D], un(ks.grad, T)):.2e} dvs {rel(g3[:, :, 2*D:], un(vs.grad, T)):.2e} dka {rel(ga[:, :, :D], un(ka.grad, Ka)):.2e} dva {rel(ga[:, :, D:], un(va.grad, Ka)):.2e} dkt {rel(gt[:, :, :D], un(kt.grad, Kt)):.2e} dvt {rel(gt[:, :, D:], un(vt.grad, Kt)):.2e} dgate {dgate.item():.4f}/{g.grad.item():.4f}")
