#!/usr/bin/env python3
"""tests/golden/qwen2_tiny_ce.npz: the HF shifted causal-LM loss of installed transformers' Qwen2ForCausalLM (third-party stand-in for
the reference's pinned transformers fork, as for tests/golden/qwen2_tiny.npz) on multimodal labels - the loss PrismaticVLM.forward
returns (prismatic/models/vlms/prismatic.py:411-422 builds the labels, :469-481 calls the LLM with them).  Pins oracle.token_ce
(SURVEY 8f-4), which until round 4 was checked against nothing but the native kernel.  Run in the build container:
    python tools/make_golden_ce.py"""
import os

import numpy as np
import torch
from transformers import Qwen2Config, Qwen2ForCausalLM

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
g = torch.Generator().manual_seed(4242)
qc = Qwen2Config(hidden_size=128, intermediate_size=320, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2, vocab_size=512,
                 rms_norm_eps=1e-6, rope_theta=1000000.0, max_position_embeddings=512, tie_word_embeddings=True, attention_dropout=0.0)
lm = Qwen2ForCausalLM._from_config(qc, attn_implementation="eager").eval().float()
with torch.no_grad():
    for p in lm.parameters():
        p.copy_(torch.randn(p.shape, generator=g) * 0.08 if p.dim() > 1 else 1.0 + 0.1 * torch.randn(p.shape, generator=g))
B, Np, L = 3, 9, 14
S = Np + L
x = torch.randn(B, S, 128, generator=g)
labels = torch.randint(0, 512, (B, L), generator=g)
labels[:, :5] = -100                     # prompt positions carry no label (IGNORE_INDEX), as the collator leaves them
labels[1, 11:] = -100                    # a padded tail
mm = torch.cat([labels[:, :1], torch.full((B, Np), -100), labels[:, 1:]], 1)         # vlms/prismatic.py:411-422
with torch.no_grad():
    o = lm(inputs_embeds=x, labels=mm, output_hidden_states=True, use_cache=False)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "qwen2_tiny_ce.npz"), hidden_last=o.hidden_states[-1].numpy(), lm_head=lm.lm_head.weight.detach().numpy(),
                    labels=labels.numpy(), num_patches=np.array(Np), loss=o.loss.numpy(), logits=o.logits.numpy())
print("loss", float(o.loss), "labelled positions", int((mm[:, 1:] != -100).sum()))
