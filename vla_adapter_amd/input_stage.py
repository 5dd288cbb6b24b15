"""On-GPU input stage for the fine-tune path: raw frames / actions / prompt ids -> the batch dict
the model consumes, following the reference's CPU pipeline step by step:

  * ``RLDSBatchTransform.__call__`` (prismatic/vla/datasets/datasets.py:29-143, ``use_minivlm`` branch): action chunk
    -> 56 token ids with ``ActionTokenizer`` (action_tokenizer.py:60-74), extended to NUM_TOKENS = 64 by
    ``random.choices`` over the 56, appended to the prompt ids minus their last three; labels = ids with everything but the
    last 65 positions set to IGNORE_INDEX;
  * ``PrismaticImageProcessor.apply_transform`` (processing_prismatic.py:128-145): ToTensor + Normalize per backbone,
    channel-stacked; frames must already have the model's input size (the LIBERO RLDS pipeline resizes to 224 x 224 -
    resampling is NOT done here);
  * ``PaddedCollatorForActionPrediction`` (prismatic/util/data_utils.py:95-175): right padding, attention mask,
    primary || wrist images on the channel dimension, stacked actions / proprio.

The heavy parts (pixels, binning) run as HIP kernels on the device; the variable-length id bookkeeping is a few hundred
integers per batch and stays on the host, using Python's ``random`` exactly like the reference so that a seeded run draws
the same 8 filler tokens.
"""
from __future__ import annotations

import random
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import ops
from .constants import IGNORE_INDEX, NUM_TOKENS

# timm data configs of the two backbones (pretrained_models/configs/preprocessor_config.json: means / stds)
IMAGENET_MEAN, IMAGENET_STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)        # DINOv2 (featurizer, channels 0-2)
SIGLIP_MEAN, SIGLIP_STD = (0.5, 0.5, 0.5), (0.5, 0.5, 0.5)                          # SigLIP (fused_featurizer, channels 3-5)


class GPUInputStage:
    def __init__(self, device="cuda", tokenizer_len: int = 151643, n_bins: int = 256, min_action: float = -1.0, max_action: float = 1.0,
                 pad_token_id: int = 151643, model_max_length: int = 2048, backbones: Sequence[str] = ("siglip",),
                 out_dtype=torch.bfloat16):
        self.device, self.tokenizer_len, self.pad, self.max_len = device, tokenizer_len, pad_token_id, model_max_length
        self.lo, self.hi = float(min_action), float(max_action)
        self.bins = torch.from_numpy(np.linspace(min_action, max_action, n_bins)).to(device)       # f64, numpy's own edges
        self.norm = [dict(dino=(IMAGENET_MEAN, IMAGENET_STD), siglip=(SIGLIP_MEAN, SIGLIP_STD))[b] for b in backbones]
        self.out_dtype = out_dtype

    def tokenize_actions(self, actions: torch.Tensor) -> torch.Tensor:
        """[..., action_dim] f32 on the device -> int64 token ids (same shape)."""
        return ops.action_tokenize(actions.to(self.device, torch.float32).contiguous(), self.bins, self.tokenizer_len, self.lo, self.hi)

    def pixels(self, frames_u8: Sequence[torch.Tensor]) -> torch.Tensor:
        """frames_u8: list over images per sample (primary first, then wrist ...) of uint8 [B, H, W, 3] tensors ->
        [B, 3 * n_backbones * n_images, H, W]: per image, one 3-channel block per backbone (apply_transform's vstack)."""
        B, H, W, _ = frames_u8[0].shape
        nb = len(self.norm)
        out = torch.empty(B, 3 * nb * len(frames_u8), H, W, device=self.device, dtype=self.out_dtype)
        for im, fr in enumerate(frames_u8):
            fr = fr.to(self.device).contiguous()
            for j, (mean, std) in enumerate(self.norm):
                ops.image_normalize_u8_(fr, out, 3 * (im * nb + j), mean, std)
        return out

    def build(self, frames_u8: Sequence[torch.Tensor], prompt_ids: List[List[int]], actions: torch.Tensor,
              proprio: Optional[torch.Tensor] = None, rng: Optional[random.Random] = None) -> Dict[str, torch.Tensor]:
        """prompt_ids: tokenizer output of the chat prompt per sample (still carrying the three trailing ids the reference
        deletes); actions [B, chunk, action_dim] normalised continuous actions (window: current + future)."""
        rng = rng or random
        B = actions.shape[0]
        tok = self.tokenize_actions(actions.reshape(B, -1)).cpu().tolist()           # 56 ids per sample (8 x 7)
        rows, labels = [], []
        for b in range(B):
            ids = list(prompt_ids[b])
            if len(ids) >= 3:
                del ids[-3:]                                                         # datasets.py:76-79
            flat = tok[b]
            if NUM_TOKENS < len(flat):
                ids = ids + flat[:NUM_TOKENS]
            else:
                ids = ids + flat + rng.choices(flat, k=NUM_TOKENS - len(flat))       # datasets.py:81-87
            lab = list(ids)
            for k in range(len(lab) - (NUM_TOKENS + 1)):                             # labels[: -(action_chunk_len + 1)] = IGNORE
                lab[k] = IGNORE_INDEX
            rows.append(ids)
            labels.append(lab)
        L = min(max(len(r) for r in rows), self.max_len)
        ids_t = torch.full((B, L), self.pad, dtype=torch.int64)
        lab_t = torch.full((B, L), IGNORE_INDEX, dtype=torch.int64)
        for b in range(B):                                                           # right padding + truncation
            n = min(len(rows[b]), L)
            ids_t[b, :n] = torch.tensor(rows[b][:n])
            lab_t[b, :n] = torch.tensor(labels[b][:n])
        ids_t, lab_t = ids_t.to(self.device), lab_t.to(self.device)
        batch = dict(pixel_values=self.pixels(frames_u8), input_ids=ids_t, labels=lab_t, attention_mask=ids_t.ne(self.pad),
                     actions=actions.to(self.device))
        if proprio is not None:
            batch["proprio"] = proprio.to(self.device, torch.float32).reshape(B, -1)
        return batch
