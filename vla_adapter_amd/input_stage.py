"""On-GPU input stage for the fine-tune path: raw frames / actions / prompt ids -> the batch dict
the model consumes, following the reference's CPU pipeline step by step:

  * ``RLDSBatchTransform.__call__`` (prismatic/vla/datasets/datasets.py:29-143, ``use_minivlm`` branch): action chunk
    -> 56 token ids with ``ActionTokenizer`` (action_tokenizer.py:60-74), extended to NUM_TOKENS = 64 by
    ``random.choices`` over the 56, appended to the prompt ids minus their last three; labels = ids with everything but the
    last 65 positions set to IGNORE_INDEX;
  * ``PrismaticImageProcessor.apply_transform`` (processing_prismatic.py:128-145): ToTensor + Normalize per backbone,
    channel-stacked, after ``TVF.resize(img, (224, 224), BICUBIC, antialias=True)`` ("resize-naive", preprocessor_config.json)
    = ``PIL.Image.resize``: Pillow's two-pass 8-bit fixed-point bicubic resampler, reproduced bit for bit by
    ``vla_resample_u8`` with host-computed taps (``pil_bicubic_coeffs``);
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


def _bicubic(x: float) -> float:
    a = -0.5
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def pil_bicubic_coeffs(in_size: int, out_size: int):
    """Pillow's precompute_coeffs + normalize_coeffs_8bpc for the BICUBIC filter (src/libImaging/Resample.c): per output position
    the first source index, the tap count and the taps in 22-bit fixed point; the support widens with the downscale factor
    (antialiasing).  Same double arithmetic, same rounding, so that the device pass is bit-identical to PIL.Image.resize."""
    import math
    prec = 32 - 8 - 2
    scale = in_size / out_size
    fscale = max(scale, 1.0)
    support = 2.0 * fscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    coefs = np.zeros((out_size, ksize), np.int32)
    inv = 1.0 / fscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        cnt = min(int(center + support + 0.5), in_size) - xmin
        w = [_bicubic((x + xmin - center + 0.5) * inv) for x in range(cnt)]
        ww = 0.0
        for v in w:
            ww += v
        if ww != 0.0:
            w = [v / ww for v in w]
        bounds[xx] = (xmin, cnt)
        for x, v in enumerate(w):
            coefs[xx, x] = int(-0.5 + v * (1 << prec)) if v < 0 else int(0.5 + v * (1 << prec))
    return bounds, coefs


class GPUInputStage:
    def __init__(self, device="cuda", tokenizer_len: int = 151643, n_bins: int = 256, min_action: float = -1.0, max_action: float = 1.0,
                 pad_token_id: int = 151643, model_max_length: int = 2048, backbones: Sequence[str] = ("siglip",),
                 out_dtype=torch.bfloat16, image_size: int = 224):
        self.device, self.tokenizer_len, self.pad, self.max_len = device, tokenizer_len, pad_token_id, model_max_length
        self.lo, self.hi = float(min_action), float(max_action)
        self.bins = torch.from_numpy(np.linspace(min_action, max_action, n_bins)).to(device)       # f64, numpy's own edges
        self.norm = [dict(dino=(IMAGENET_MEAN, IMAGENET_STD), siglip=(SIGLIP_MEAN, SIGLIP_STD))[b] for b in backbones]
        self.out_dtype = out_dtype
        self.image_size = image_size
        self._taps = {}

    def resize(self, frames_u8: torch.Tensor, out_h: int = 224, out_w: int = 224) -> torch.Tensor:
        """uint8 [B, H, W, 3] -> uint8 [B, out_h, out_w, 3], bit-identical to PIL.Image.resize((out_w, out_h), BICUBIC): horizontal
        pass into an 8-bit intermediate, then the vertical pass (Pillow's order)."""
        fr = frames_u8.to(self.device).contiguous()
        B, H, W, Cc = fr.shape
        assert fr.dtype == torch.uint8 and Cc == 3

        def taps(n_in, n_out):
            key = (n_in, n_out)
            if key not in self._taps:
                b, c = pil_bicubic_coeffs(n_in, n_out)
                self._taps[key] = (torch.from_numpy(b).to(self.device), torch.from_numpy(c).to(self.device), c.shape[1])
            return self._taps[key]
        cur = fr
        if W != out_w:
            b, c, ks = taps(W, out_w)
            nxt = torch.empty(B, H, out_w, 3, device=self.device, dtype=torch.uint8)
            ops.N.check(ops._lib().vla_resample_u8(ops._st(), ops._p(cur), ops._p(nxt), B * H, W, out_w, 3, ops._p(b), ops._p(c), ks), "resample_u8")
            cur = nxt
        if H != out_h:
            b, c, ks = taps(H, out_h)
            nxt = torch.empty(B, out_h, out_w, 3, device=self.device, dtype=torch.uint8)
            ops.N.check(ops._lib().vla_resample_u8(ops._st(), ops._p(cur), ops._p(nxt), B, H, out_h, out_w * 3, ops._p(b), ops._p(c), ks), "resample_u8")
            cur = nxt
        return cur

    def tokenize_actions(self, actions: torch.Tensor) -> torch.Tensor:
        """[..., action_dim] f32 on the device -> int64 token ids (same shape)."""
        return ops.action_tokenize(actions.to(self.device, torch.float32).contiguous(), self.bins, self.tokenizer_len, self.lo, self.hi)

    def pixels(self, frames_u8: Sequence[torch.Tensor]) -> torch.Tensor:
        """frames_u8: list over images per sample (primary first, then wrist ...) of uint8 [B, H, W, 3] tensors ->
        [B, 3 * n_backbones * n_images, H, W]: per image, one 3-channel block per backbone (apply_transform's vstack)."""
        if tuple(frames_u8[0].shape[1:3]) != (self.image_size, self.image_size):       # apply_transform: resize first
            frames_u8 = [self.resize(f, self.image_size, self.image_size) for f in frames_u8]
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
