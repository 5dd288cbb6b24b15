#!/usr/bin/env python3
"""tests/golden/vit_{siglip,dinov2reg}_tiny.npz: the two vision-backbone flavours of the reference against INDEPENDENT third-party
implementations - installed transformers' SiglipVisionModel and Dinov2WithRegistersModel built from local configs (random weights, no
download).  The reference runs timm 0.9.10's VisionTransformer (vit_so400m_patch14_siglip_224 / vit_large_patch14_reg4_dinov2:
modeling_prismatic.py:120-144, 196-237), which is absent here; the HF models implement the same published architectures, so they pin
what oracle.vit_forward restates from the reference text: patch embedding as a P x P convolution, learned position embedding on the
patch tokens, pre-norm blocks x + ls1(attn(norm1 x)), x + ls2(mlp(norm2 x)), biased q/k/v, exact-erf GELU, eps 1e-6, no cls token
(SigLIP) / cls + 4 register tokens in front of the patches and LayerScale (DINOv2), the hidden state BEHIND block depth-2 with the
prefix tokens dropped and no final norm.  One convention differs and is mapped, not restated: HF's DINOv2 adds a position embedding
to the cls token, timm's reg4 checkpoints (no_embed_class=True) carry it folded into the token itself: cls_timm = cls_hf + pos_hf[0].
Stand-ins, not the reference: row a3 moves from "unpinned" to "pinned by third-party stand-in", like a5 (Qwen2).
    python tools/make_golden_vit.py"""
import os

import numpy as np
import torch
from transformers import Dinov2WithRegistersConfig, Dinov2WithRegistersModel, SiglipVisionConfig, SiglipVisionModel

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")
g = torch.Generator().manual_seed(777)


def seeded_(m, std):
    with torch.no_grad():
        for n, p in m.named_parameters():
            if p.dim() > 1 or "token" in n or "position" in n or "lambda" in n:
                p.copy_(torch.randn(p.shape, generator=g) * (0.5 if "lambda" in n else std))
            elif "norm" in n and n.endswith("weight"):
                p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
            else:
                p.copy_(torch.randn(p.shape, generator=g) * 0.05)


d, depth, heads, mlp, img, P = 96, 4, 3, 256, 56, 14
npd = lambda t: t.detach().numpy()

# ---- SigLIP flavour
c = SiglipVisionConfig(hidden_size=d, intermediate_size=mlp, num_hidden_layers=depth, num_attention_heads=heads, image_size=img, patch_size=P,
                       hidden_act="gelu", layer_norm_eps=1e-6, attention_dropout=0.0)
m = SiglipVisionModel._from_config(c, attn_implementation="eager").eval().float()
seeded_(m, 0.08)
x = torch.randn(2, 3, img, img, generator=g)
with torch.no_grad():
    hs = m(pixel_values=x, output_hidden_states=True).hidden_states
sd = m.state_dict()
pre = "" if "embeddings.patch_embedding.weight" in sd else "vision_model."
w = {"patch_embed.proj.weight": sd[pre + "embeddings.patch_embedding.weight"], "patch_embed.proj.bias": sd[pre + "embeddings.patch_embedding.bias"],
     "pos_embed": sd[pre + "embeddings.position_embedding.weight"][None]}
for i in range(depth):
    L, b = f"{pre}encoder.layers.{i}.", f"blocks.{i}."
    w[b + "norm1.weight"], w[b + "norm1.bias"] = sd[L + "layer_norm1.weight"], sd[L + "layer_norm1.bias"]
    w[b + "attn.qkv.weight"] = torch.cat([sd[L + f"self_attn.{n}_proj.weight"] for n in "qkv"], 0)
    w[b + "attn.qkv.bias"] = torch.cat([sd[L + f"self_attn.{n}_proj.bias"] for n in "qkv"], 0)
    w[b + "attn.proj.weight"], w[b + "attn.proj.bias"] = sd[L + "self_attn.out_proj.weight"], sd[L + "self_attn.out_proj.bias"]
    w[b + "norm2.weight"], w[b + "norm2.bias"] = sd[L + "layer_norm2.weight"], sd[L + "layer_norm2.bias"]
    for n in ("fc1", "fc2"):
        w[b + f"mlp.{n}.weight"], w[b + f"mlp.{n}.bias"] = sd[L + f"mlp.{n}.weight"], sd[L + f"mlp.{n}.bias"]
np.savez_compressed(os.path.join(OUT, "vit_siglip_tiny.npz"), pixels=npd(x), out=npd(hs[depth - 1]), cfg=np.array([d, depth, heads, mlp, P, 0, 0]),
                    **{"w." + k: npd(v) for k, v in w.items()})

# ---- DINOv2 with registers
mlp = 3 * d                     # (HF's DINOv2 config takes an integer MLP ratio)
c = Dinov2WithRegistersConfig(hidden_size=d, num_hidden_layers=depth, num_attention_heads=heads, mlp_ratio=3, image_size=img, patch_size=P,
                              num_register_tokens=4, hidden_act="gelu", layer_norm_eps=1e-6, layerscale_value=1.0, qkv_bias=True, use_swiglu_ffn=False,
                              attention_probs_dropout_prob=0.0, hidden_dropout_prob=0.0, drop_path_rate=0.0)
m = Dinov2WithRegistersModel._from_config(c, attn_implementation="eager").eval().float()
seeded_(m, 0.08)
x = torch.randn(2, 3, img, img, generator=g)
with torch.no_grad():
    hs = m(pixel_values=x, output_hidden_states=True).hidden_states
sd = m.state_dict()
pos = sd["embeddings.position_embeddings"]                      # [1, Np + 1, d]: cls first
w = {"patch_embed.proj.weight": sd["embeddings.patch_embeddings.projection.weight"], "patch_embed.proj.bias": sd["embeddings.patch_embeddings.projection.bias"],
     "pos_embed": pos[:, 1:], "cls_token": sd["embeddings.cls_token"] + pos[:, :1], "reg_token": sd["embeddings.register_tokens"]}
for i in range(depth):
    L, b = f"encoder.layer.{i}.", f"blocks.{i}."
    w[b + "norm1.weight"], w[b + "norm1.bias"] = sd[L + "norm1.weight"], sd[L + "norm1.bias"]
    w[b + "attn.qkv.weight"] = torch.cat([sd[L + f"attention.attention.{n}.weight"] for n in ("query", "key", "value")], 0)
    w[b + "attn.qkv.bias"] = torch.cat([sd[L + f"attention.attention.{n}.bias"] for n in ("query", "key", "value")], 0)
    w[b + "attn.proj.weight"], w[b + "attn.proj.bias"] = sd[L + "attention.output.dense.weight"], sd[L + "attention.output.dense.bias"]
    w[b + "ls1.scale_factor"], w[b + "ls2.scale_factor"] = sd[L + "layer_scale1.lambda1"], sd[L + "layer_scale2.lambda1"]
    w[b + "norm2.weight"], w[b + "norm2.bias"] = sd[L + "norm2.weight"], sd[L + "norm2.bias"]
    for n in ("fc1", "fc2"):
        w[b + f"mlp.{n}.weight"], w[b + f"mlp.{n}.bias"] = sd[L + f"mlp.{n}.weight"], sd[L + f"mlp.{n}.bias"]
np.savez_compressed(os.path.join(OUT, "vit_dinov2reg_tiny.npz"), pixels=npd(x), out=npd(hs[depth - 1][:, 5:]), cfg=np.array([d, depth, heads, mlp, P, 5, 1]),
                    **{"w." + k: npd(v) for k, v in w.items()})
print("written", [f for f in os.listdir(OUT) if f.startswith("vit_")])
