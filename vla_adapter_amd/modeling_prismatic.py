"""Host-side mirror of the reference's model API for the fine-tune hot path, on the native engine.

Keeps:
  * ``OpenVLAForActionPrediction.forward`` - kwargs of prismatic/extern/hf/modeling_prismatic.py:525-544; the
    multimodal branch (:596-655) returns ``PrismaticCausalLMOutputWithPast`` with ``hidden_states`` = tuple of n+1
    ``[B, S, D]`` tensors (HF convention: [0] inputs_embeds ... [n] final-norm output), ``loss=None``, ``logits=None``
    (the reference discards the lm_head output, :680-686; this build never computes it);
  * attributes callers use: ``vision_backbone.get_num_patches()/get_num_images_in_input()/set_num_images_in_input()``
    (:166-193), ``llm_dim`` (:372), ``action_queries``, ``norm_stats``;
  * ``PrismaticVLM.forward`` signature of prismatic/models/vlms/prismatic.py:312-325.
There is no autograd graph: training goes through ``engine.VLAEngine`` (explicit backward kernels).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Dict, Optional, Tuple

import torch

from . import engine as E
from . import ops
from .ops import BF16


@dataclass
class PrismaticCausalLMOutputWithPast:            # modeling_prismatic.py:278-289
    loss: Optional[torch.Tensor] = None
    logits: Optional[torch.Tensor] = None
    past_key_values: Any = None
    hidden_states: Optional[Tuple[torch.Tensor, ...]] = None
    attentions: Any = None
    projector_features: Optional[torch.Tensor] = None


class _VisionBackboneView:
    def __init__(self, cfg: E.VLACfg):
        self._cfg = cfg

    def get_num_patches(self) -> int:
        return self._cfg.vit[0].n_patches

    def get_num_images_in_input(self) -> int:
        return self._cfg.n_img

    def set_num_images_in_input(self, n: int) -> None:
        assert n == 1 or self._cfg.fused, "Multi-image inputs require using fused backbone!"   # modeling_prismatic.py:213
        self._cfg.n_img = n


class OpenVLAForActionPrediction:
    def __init__(self, cfg: E.VLACfg, weights: Dict, device="cuda", norm_stats: Optional[dict] = None):
        self.engine = E.VLAEngine(cfg, weights, device)
        self.cfg, self.device = cfg, device
        self.vision_backbone = _VisionBackboneView(cfg)
        self.llm_dim = cfg.llm.d
        self.norm_stats = norm_stats or {}
        self.training = True

    @property
    def action_queries(self) -> torch.Tensor:       # nn.Embedding(64, D).weight in the reference (:375-376)
        return self.engine.head.P.view("action_queries")

    def forward(self, input_ids=None, attention_mask=None, pixel_values=None, labels=None, inputs_embeds=None,
                past_key_values=None, use_cache=None, output_attentions=None, output_hidden_states=None,
                output_projector_features=None, return_dict=None, proprio=None, proprio_projector=None, noisy_actions=None,
                noisy_action_projector=None, diffusion_timestep_embeddings=None, use_film: bool = False):
        if use_film:
            raise NotImplementedError("FiLM (use_film) is outside the accelerated path (off in every shipped script)")
        if past_key_values is not None or inputs_embeds is not None or pixel_values is None or labels is None:
            raise NotImplementedError("only the multimodal training branch (modeling_prismatic.py:596-655) is accelerated")
        assert input_ids.shape[0] == pixel_values.shape[0], "Non-homogenous batch of (text, image) input"
        eng = self.engine
        batch = dict(input_ids=input_ids.to(self.device), labels=labels.to(self.device),
                     attention_mask=attention_mask.to(self.device), pixel_values=pixel_values.to(self.device).contiguous())
        eng.forward_vlm(batch)
        n = self.cfg.llm.n_layers
        hs = tuple(eng.llm.HS[i] for i in range(n + 1)) if (output_hidden_states is None or output_hidden_states) else None
        Np = self.cfg.n_patches
        pf = eng.llm.HS[0][:, 1:Np + 1] if output_projector_features else None
        return PrismaticCausalLMOutputWithPast(loss=None, logits=None, hidden_states=hs, projector_features=pf)

    __call__ = forward

    def predict_action(self, *a, **k):
        raise NotImplementedError("batch-1 inference predict_action (modeling_prismatic.py:892-972) is a 'next' row (SURVEY 8f-1)")


class PrismaticVLM:
    """Signature keeper for prismatic/models/vlms/prismatic.py:312-325 (native, non-HF API)."""

    def __init__(self, model: OpenVLAForActionPrediction):
        self.model = model

    def forward(self, input_ids=None, attention_mask=None, pixel_values=None, labels=None, inputs_embeds=None,
                past_key_values=None, use_cache=None, output_attentions=None, output_hidden_states=None, return_dict=None,
                multimodal_indices=None):
        if isinstance(pixel_values, dict):          # {"dino": .., "siglip": ..} (dinosiglip_vit.py:158-170)
            pixel_values = torch.cat([pixel_values["dino"], pixel_values["siglip"]], dim=1)
        if multimodal_indices is not None and len(multimodal_indices) != input_ids.shape[0]:
            raise NotImplementedError("mixed unimodal/multimodal batches are outside the accelerated path")
        out = self.model.forward(input_ids=input_ids, attention_mask=attention_mask, pixel_values=pixel_values, labels=labels,
                                 output_hidden_states=True)
        # token-CE loss over the 151 936-way vocab (lm_head) is the 'next' row 8f-4; hidden states are returned.
        return out

    __call__ = forward
