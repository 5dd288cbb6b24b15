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

    # ---- batch-1 inference (modeling_prismatic.py:892-972; driver experiments/robot/openvla_utils.py:737-825) ------
    @staticmethod
    def _check_unnorm_key(norm_stats, unnorm_key):                                # modeling_prismatic.py:975-990
        if unnorm_key is None:
            assert len(norm_stats) == 1, ("Your model was trained on more than one dataset, please pass a `unnorm_key` from the "
                                          f"following options to choose the statistics used for un-normalizing actions: {norm_stats.keys()}")
            unnorm_key = next(iter(norm_stats.keys()))
        assert unnorm_key in norm_stats, ("The `unnorm_key` you chose is not in the set of available dataset statistics, "
                                          f"please choose from: {norm_stats.keys()}")
        return unnorm_key

    def get_action_dim(self, unnorm_key=None) -> int:
        return len(self.norm_stats[self._check_unnorm_key(self.norm_stats, unnorm_key)]["action"]["min"])

    def get_action_stats(self, unnorm_key=None):
        return self.norm_stats[self._check_unnorm_key(self.norm_stats, unnorm_key)]["action"]

    def _unnormalize_actions(self, normalized_actions, unnorm_key=None):          # modeling_prismatic.py:784-805
        import numpy as np
        from . import constants as K
        st = self.get_action_stats(unnorm_key)
        if K.ACTION_PROPRIO_NORMALIZATION_TYPE == "bounds":
            mask = st.get("mask", np.ones_like(st["min"], dtype=bool))
            high, low = np.array(st["max"]), np.array(st["min"])
        elif K.ACTION_PROPRIO_NORMALIZATION_TYPE == "bounds_q99":
            mask = st.get("mask", np.ones_like(st["q01"], dtype=bool))
            high, low = np.array(st["q99"]), np.array(st["q01"])
        else:
            raise ValueError("Unsupported action/proprio normalization type detected!")
        return np.where(mask, 0.5 * (normalized_actions + 1) * (high - low + 1e-8) + low, normalized_actions)

    @staticmethod
    def prepare_inference_inputs(input_ids: torch.Tensor, attention_mask: torch.Tensor):
        """_prepare_input_for_action_prediction + _prepare_labels_for_action_prediction (:747-782): append the 64
        placeholder action ids (value 1) and the stop id, fake labels that mark exactly those 64 positions."""
        from . import constants as K
        B, L0 = input_ids.shape
        ids = torch.cat([input_ids, torch.ones(B, K.NUM_TOKENS, dtype=input_ids.dtype, device=input_ids.device),
                         torch.full((B, 1), K.STOP_INDEX, dtype=input_ids.dtype, device=input_ids.device)], dim=-1)
        am = torch.cat([attention_mask, torch.ones(B, ids.shape[-1] - L0, dtype=attention_mask.dtype, device=attention_mask.device)], dim=-1)
        labels = torch.full_like(ids, K.IGNORE_INDEX)
        labels[:, L0:] = K.ACTION_TOKEN_BEGIN_IDX + 1
        labels[:, -1] = K.STOP_INDEX
        return ids, am, labels

    def predict_action(self, input_ids=None, unnorm_key=None, proprio=None, proprio_projector=None, action_head=None,
                       noisy_action_projector=None, use_film: bool = False, **kwargs):
        """Same call as the reference (``pixel_values`` / ``attention_mask`` in kwargs, batch 1): returns
        (un-normalised actions ndarray [chunk, action_dim], action hidden states [1, 1, 64, D] of the last layer).
        ``action_head`` / ``proprio_projector``: this package's mirrors (their parameters are loaded into the engine) or
        ``True`` for the parameters the engine already holds (trained in place).  The discrete-token branch
        (``action_head=None`` -> lm_head argmax) is not built."""
        import numpy as np
        if use_film or noisy_action_projector is not None:
            raise NotImplementedError("FiLM / diffusion heads are outside the accelerated path")
        if action_head is None:
            raise NotImplementedError("discrete-token action prediction needs the lm_head (SURVEY 8f-4); pass the L1 regression head")
        assert input_ids.shape[0] == 1, "Generation with batch size > 1 is not currently supported!"     # :700-703
        eng, dev = self.engine, self.device
        if action_head is not True:
            sd = action_head.state_dict()
            psd = proprio_projector.state_dict() if (proprio_projector is not None and proprio_projector is not True) else {
                k: v for k, v in eng.head.proprio_views().items()}
            key = (id(action_head), id(proprio_projector), getattr(action_head, "_version", 0), getattr(proprio_projector, "_version", 0))
            if getattr(self, "_loaded_head", None) != key:
                eng.head.load_state_dicts(sd, psd)
                self._loaded_head = key
        ids, am, labels = self.prepare_inference_inputs(input_ids.to(dev), kwargs["attention_mask"].to(dev))
        use_proprio = proprio_projector is not None and proprio is not None
        assert use_proprio, "the regression head dereferences proprio and its projector (action_heads.py:53-54)"
        pr = torch.as_tensor(np.asarray(proprio), dtype=torch.float32, device=dev).reshape(1, -1)
        batch = dict(input_ids=ids, labels=labels, attention_mask=am.bool(), pixel_values=kwargs["pixel_values"].to(dev).contiguous(),
                     proprio=pr)
        pred = eng.predict(batch)                                                   # [1, chunk, action_dim] bf16
        normalized = pred.reshape(self.cfg.chunk, self.cfg.action_dim).float().cpu().numpy()
        n, Np = self.cfg.llm.n_layers, self.cfg.n_patches
        s0 = Np + input_ids.shape[-1] - 1                                           # NUM_PATCHES + NUM_PROMPT_TOKENS (:856)
        from . import constants as K
        hid = eng.llm.HS[n][:, s0:s0 + K.NUM_TOKENS].reshape(1, 1, K.NUM_TOKENS, -1)
        return self._unnormalize_actions(normalized, unnorm_key), hid


@dataclass
class CausalLMOutputWithPast:                     # transformers.modeling_outputs.CausalLMOutputWithPast (what the reference returns)
    loss: Optional[torch.Tensor] = None
    logits: Optional[torch.Tensor] = None
    past_key_values: Any = None
    hidden_states: Optional[Tuple[torch.Tensor, ...]] = None
    attentions: Any = None


class PrismaticVLM:
    """prismatic/models/vlms/prismatic.py:312-481 on the native engine: the fully-multimodal branch of the plain VLM
    forward - ViT(s) -> projector -> [tok0 | patches | tok1..] splice (NO action queries: they belong to
    OpenVLAForActionPrediction) -> Qwen2 stack.  Every argument the native path cannot honour raises; nothing is dropped."""

    def __init__(self, model: OpenVLAForActionPrediction):
        self.model = model

    def forward(self, input_ids=None, attention_mask=None, pixel_values=None, labels=None, inputs_embeds=None,
                past_key_values=None, use_cache=None, output_attentions=None, output_hidden_states=None, return_dict=None,
                multimodal_indices=None):
        if input_ids is None or input_ids.shape[1] == 1 or pixel_values is None:
            if past_key_values is not None:
                raise NotImplementedError("cached single-token decoding (prismatic.py:328-342) is outside the accelerated path")
            raise RuntimeError("Invalid `forward()` call!")                      # prismatic.py:344-345
        if inputs_embeds is not None or past_key_values is not None or use_cache:
            raise NotImplementedError("inputs_embeds / past_key_values / use_cache: only the uncached multimodal forward is built")
        if output_attentions:
            raise NotImplementedError("output_attentions: the flash-style attention kernel never materialises the probabilities")
        if return_dict is False:
            raise NotImplementedError("return_dict=False: a CausalLMOutputWithPast is always returned")
        if isinstance(pixel_values, dict):          # {"dino": .., "siglip": ..} (dinosiglip_vit.py:158-170)
            pixel_values = torch.cat([pixel_values["dino"], pixel_values["siglip"]], dim=1)
        B = input_ids.shape[0]
        if multimodal_indices is not None and sorted(int(i) for i in multimodal_indices) != list(range(B)):
            raise NotImplementedError("mixed unimodal/multimodal batches (prismatic.py:424-467) are outside the accelerated path")
        eng, dev = self.model.engine, self.model.device
        if attention_mask is None:
            attention_mask = torch.ones_like(input_ids, dtype=torch.bool)
        batch = dict(input_ids=input_ids.to(dev), attention_mask=attention_mask.to(dev), pixel_values=pixel_values.to(dev).contiguous())
        eng.forward_vlm(batch, action_queries=False)
        n = self.model.cfg.llm.n_layers
        loss = logits = None
        if labels is not None:                      # HF shifted cross-entropy over the vocabulary
            loss, logits = eng.token_ce(labels.to(dev))
        hs = tuple(eng.llm.HS[i] for i in range(n + 1)) if output_hidden_states else None
        return CausalLMOutputWithPast(loss=loss, logits=logits, hidden_states=hs)

    __call__ = forward
