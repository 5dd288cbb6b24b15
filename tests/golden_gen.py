"""Seeded, ORDER-INDEPENDENT generation of the large inputs of the reference-run fixtures (tests/golden/head_bf16_*.npz).

The Pro action head at D = 896 holds 218 M parameters (436 MB in bf16) and its input [B, 25, 320, 896] another 29 MB: too
large to commit.  Every tensor is therefore derived from (seed, tensor name, shape) alone by ``tensor()`` below - the
fixture script (tools/make_golden.py, run where /root/reference exists) fills the REFERENCE module with these values, runs
it, and commits only the outputs plus a digest of every generated input; the tests regenerate the same inputs (same torch
CPU generator, same image), check the digest, and compare.  No reference code is involved here.
"""
import hashlib
import zlib

import torch

BF = torch.bfloat16


def tensor(seed: int, name: str, shape, std: float = 1.0, mean: float = 0.0) -> torch.Tensor:
    """bf16-representable fp32 tensor ~ N(mean, std^2), a function of (seed, name, shape) only."""
    g = torch.Generator().manual_seed((zlib.crc32(name.encode()) * 2654435761 + seed * 97) % (2 ** 63 - 1))
    return (torch.randn(tuple(shape), generator=g) * std + mean).to(BF).float()


def head_param(seed: int, name: str, shape) -> torch.Tensor:
    """Values for one parameter of L1RegressionActionHead / ProprioProjector by its reference state-dict name."""
    if name.endswith("gating_factor"):
        return torch.full(tuple(shape), 0.3).to(BF).float()
    is_norm = "layer_norm" in name or ".ffn.0." in name
    if is_norm:
        return tensor(seed, name, shape, 0.1, 1.0 if name.endswith("weight") else 0.0)
    if name.endswith("ffn.1.bias"):
        # A random head without this is a CHAOTIC map: every block is x <- ReLU(W LN(attn + x) + b) with no outer residual, and with
        # b ~ 0 a perturbation grows ~1.2x per block (measured: the reference's own bf16 run then differs from its fp32 run by
        # 30-50 % after 24 blocks - a fixture that cannot tell a bug from rounding).  A bias of the same power as W z makes the
        # block a contraction (gain^2 ~ 1/2), so rounding noise stays at the bf16 level and an implementation error stands out.
        return tensor(seed, name, shape, min(0.05, 1.0 / shape[-1] ** 0.5) * shape[-1] ** 0.5)
    if name.endswith("bias"):
        return tensor(seed, name, shape, 0.02)
    fan_in = shape[-1]
    return tensor(seed, name, shape, min(0.05, 1.0 / fan_in ** 0.5))      # linear weights: activations stay O(1) through 24 blocks


CASES = {
    # name: (pro, D, Kt, B, phase[, num_blocks])   D = 128 -> head dim 16 (smallest MFMA-capable), D = 896 -> head dim 112 (the real head)
    "pro_d128_kt64": (True, 128, 64, 2, "Inference"),
    "pro_d128_kt64_train": (True, 128, 64, 2, "Training"),
    # (the original block has no RoPE: in phase "Inference" the 8 chunk rows of a sample stay IDENTICAL through all 24 blocks, and a
    #  single ReLU pre-activation within rounding of zero then flips for all rows at once - a degenerate input where one flip moves a
    #  bias gradient by 9 %; the Training-phase perturbation makes the rows distinct)
    "orig_d128_kt64": (False, 128, 64, 2, "Training"),
    "pro_d896_kt256": (True, 896, 256, 2, "Inference"),
    "pro_d896_kt256_train": (True, 896, 256, 2, "Training"),
    "pro_d896_kt512": (True, 896, 512, 1, "Inference"),
    # ONE-block heads (MLPResNet(num_blocks=1)): forward and backward rounding points without 24 blocks of accumulated drift -
    # two bf16 evaluations of these agree to a few 1e-3 on every gradient, so an implementation error cannot hide in noise
    "pro1_d128_kt64": (True, 128, 64, 2, "Training", 1),
    "orig1_d128_kt64": (False, 128, 64, 2, "Training", 1),
    "pro1_d896_kt256": (True, 896, 256, 2, "Training", 1),
}
SEED = 20260
NUM_TOKENS, CHUNK, ACTION_DIM, PROPRIO_DIM, NUM_BLOCKS = 64, 8, 7, 8, 24


def case_cfg(case: str):
    c = CASES[case]
    return c[0], c[1], c[2], c[3], c[4], (c[5] if len(c) > 5 else NUM_BLOCKS)


def head_keys(D: int, pro: bool, nb: int = NUM_BLOCKS):
    """(name, shape) of every parameter the forward uses (film_gen exists in the Pro state dict but is never read)."""
    Da = ACTION_DIM
    out = [("model.layer_norm1.weight", (Da * D,)), ("model.layer_norm1.bias", (Da * D,)), ("model.fc1.weight", (D, Da * D)),
           ("model.fc1.bias", (D,)), ("model.layer_norm2.weight", (D,)), ("model.layer_norm2.bias", (D,)),
           ("model.fc2.weight", (Da, D)), ("model.fc2.bias", (Da,))]
    lin = ("q_proj", "k_self", "v_self", "k_adapter", "v_adapter", "k_task", "v_task", "o_proj", "ffn.1") if pro else (
        "q_proj", "k_proj", "v_proj", "o_proj", "ffn.1")
    for i in range(nb):
        p = f"model.mlp_resnet_blocks.{i}."
        for n in lin:
            out += [(p + n + ".weight", (D, D)), (p + n + ".bias", (D,))]
        out += [(p + "ffn.0.weight", (D,)), (p + "ffn.0.bias", (D,)), (p + "gating_factor", (1,))]
    return out


def proprio_keys(D: int):
    return [("fc1.weight", (D, PROPRIO_DIM)), ("fc1.bias", (D,)), ("fc2.weight", (D, D)), ("fc2.bias", (D,))]


def case_inputs(case: str):
    """-> dict(head=sd, proprio=sd, mlhs [B,25,Kt+64,D], prop [B,8], target [B,8,7], noise [8,7D] or None), all fp32 tensors
    holding bf16-representable values."""
    pro, D, Kt, B, phase, nb = case_cfg(case)
    s = SEED + (zlib.crc32(case.encode()) % 1000)
    head = {n: head_param(s, n, sh) for n, sh in head_keys(D, pro, nb)}
    prop_sd = {n: head_param(s, "proprio." + n, sh) for n, sh in proprio_keys(D)}
    mlhs = tensor(s, "mlhs", (B, nb + 1, Kt + NUM_TOKENS, D))
    prop = tensor(s, "proprio_in", (B, PROPRIO_DIM), 0.5)
    target = tensor(s, "target", (B, CHUNK, ACTION_DIM), 0.5)
    noise = tensor(s, "noise", (CHUNK, ACTION_DIM * D), 0.02) if phase == "Training" else None
    # upstream gradient of the backward: the shape of an L1 gradient (sign / n) but FIXED - sign(pred - target) itself flips on
    # bf16-level differences of pred and would turn the gradient comparison into a coin toss (the L1 kernel is pinned on its own)
    dpred = (tensor(s, "dpred", (B, CHUNK, ACTION_DIM)).sign() / (B * CHUNK * ACTION_DIM)).to(BF).float()
    return dict(head=head, proprio=prop_sd, mlhs=mlhs, prop=prop, target=target, noise=noise, dpred=dpred)


def digest(inp: dict) -> str:
    h = hashlib.sha256()
    for grp in ("head", "proprio"):
        for k in sorted(inp[grp]):
            h.update(k.encode())
            h.update(inp[grp][k].to(BF).view(torch.int16).numpy().tobytes())
    for k in ("mlhs", "prop", "target", "noise", "dpred"):
        if inp[k] is not None:
            h.update(inp[k].to(BF).view(torch.int16).numpy().tobytes())
    return h.hexdigest()


# gradients kept in the fixtures: small tensors that sit at the END of the backward chain (everything upstream feeds them)
GRAD_KEYS = ["model.fc2.weight", "model.fc2.bias", "model.layer_norm2.weight", "model.layer_norm1.bias", "model.fc1.bias",
             "model.mlp_resnet_blocks.23.ffn.0.weight", "model.mlp_resnet_blocks.23.o_proj.bias", "model.mlp_resnet_blocks.12.q_proj.bias",
             "model.mlp_resnet_blocks.12.gating_factor", "model.mlp_resnet_blocks.0.ffn.1.bias", "model.mlp_resnet_blocks.0.gating_factor",
             "model.mlp_resnet_blocks.0.ffn.0.bias"]


def grad_keys(case: str):
    """24-block cases: GRAD_KEYS.  One-block cases: EVERY 1-D parameter (biases, LayerNorms, gate) + fc2.weight."""
    pro, D, Kt, B, phase, nb = case_cfg(case)
    if nb == NUM_BLOCKS:
        return list(GRAD_KEYS)
    return ["model.fc2.weight"] + [n for n, sh in head_keys(D, pro, nb) if len(sh) == 1]


def weight_grad_rows(case: str):
    """One-block cases also keep the first 16 rows of these weight gradients (dW = dY^T X products)."""
    pro, D, Kt, B, phase, nb = case_cfg(case)
    if nb == NUM_BLOCKS:
        return []
    lin = ("q_proj", "k_self", "v_self", "k_adapter", "v_adapter", "k_task", "v_task", "o_proj", "ffn.1") if pro else ("q_proj", "k_proj", "v_proj", "o_proj", "ffn.1")
    return [f"model.mlp_resnet_blocks.0.{n}.weight" for n in lin] + ["model.fc1.weight"]


DX_LAYERS = [1, 12, 24]        # hidden-state indices whose input gradient is kept (D = 128, 24-block cases; one-block cases keep layer 1)
BLOCK_TAPS = [0, 1, 2, 3, 12, 23]   # (24-block cases; one-block cases tap block 0)    # block outputs kept: the first blocks pin the rounding points op by op (no accumulated drift yet)


def dx_layers(case: str):
    return DX_LAYERS if case_cfg(case)[5] == NUM_BLOCKS else [1]


def block_taps(case: str):
    return BLOCK_TAPS if case_cfg(case)[5] == NUM_BLOCKS else [0]


def dx_rows(case: str):
    """Token rows of the kept hidden-state gradient: all of them at D = 128, the last 16 task tokens + the 64 action tokens at D = 896."""
    pro, D, Kt, B, phase, nb = case_cfg(case)
    return slice(0, Kt + NUM_TOKENS) if D <= 128 else slice(Kt - 16, Kt + NUM_TOKENS)
