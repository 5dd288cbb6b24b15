import os, sys, torch
sys.path.insert(0, os.getcwd())
from vla_adapter_amd import ops
DEV, BF = "cuda", torch.bfloat16
B, T, Ka, Kt, D, H = 3, 8, 65, 256, 896, 8
g = torch.Generator(device=DEV).manual_seed(0)
x3 = (torch.randn(B, T, 3 * D, device=DEV, generator=g) * .3).to(BF); a2 = (torch.randn(B, Ka, 2 * D, device=DEV, generator=g) * .3).to(BF); t2 = (torch.randn(B, Kt, 2 * D, device=DEV, generator=g) * .3).to(BF)
gate = torch.tensor([0.7]).to(BF).to(DEV); dout = torch.randn(B, T, D, device=DEV, generator=g).to(BF)
args = (x3[:, :, :D], x3[:, :, D:2 * D], x3[:, :, 2 * D:], a2[:, :, :D], a2[:, :, D:], t2[:, :, :D], t2[:, :, D:])
out, probs = ops.head_attn_fwd(*args, gate, H)
tabs = ops.rope_inter_tables(max(T, Ka, Kt), D // H, DEV)
res = []
for comb in (True, False):
    if comb: os.environ["VLA_HEAD_BWD_COMBINED"] = "1"
    else: os.environ.pop("VLA_HEAD_BWD_COMBINED", None)
    g3, ga, gt = torch.zeros_like(x3), torch.zeros_like(a2), torch.zeros_like(t2); dg = torch.zeros(1, device=DEV)
    ops.head_attn_bwd(dout, out, *args, gate, probs, dg, g3[:, :, :D], g3[:, :, D:2 * D], g3[:, :, 2 * D:], ga[:, :, :D], ga[:, :, D:], gt[:, :, :D], gt[:, :, D:], H, rope=tabs)
    res.append((g3.clone(), ga.clone(), gt.clone()))
for name, a, b in (("self k|v", res[0][0][:, :, D:], res[1][0][:, :, D:]), ("adp", res[0][1], res[1][1]), ("task", res[0][2], res[1][2])):
    d = (a.float() - b.float()).abs()
    nz = (d > 0)
    print(name, "differing elements", int(nz.sum()), "of", d.numel(), "max abs", d.max().item(), "max rel", (d / (a.float().abs() + 1e-12))[nz].max().item() if nz.any() else 0)
    if nz.any():
        idx = nz.nonzero()[:5]; print(idx.tolist())
