"""Native execution engine of the VLA-Adapter fine-tune hot path on MI355X.

Orchestrates the HIP kernels of ``libvla_native.so`` (through ``ops``) for the whole training step of
``vla-scripts/finetune.py:run_forward_pass`` + backward + AdamW (reference citations per method).  PyTorch is used
for device memory, streams and (in ``ddp.py``) RCCL only; every FLOP and every byte moved on the hot path is a
hand-written gfx950 kernel.  No autograd: backward is explicit, activations live in buffers allocated once and
reused every step (static shapes -> the step is hipGraph-capturable).

Layout decisions (MI355X-first, 288 GB HBM):
  * frozen weights are stored twice: [out,in] for the forward NT GEMM and pre-transposed [in,out] for dX, so every
    product is the K-contiguous NT form the MFMA kernel wants;
  * q/k/v and gate/up projections are fused ([1152,896] and a 16-row interleaved [9728,896]) so the SwiGLU product
    is formed in the GEMM epilogue;
  * ViT MLP width 4304 is zero-padded to 4352 (=34*128) so every GEMM dim is tile-aligned;
  * LLM hidden states for all layers live in one [n+2, B, S, D] buffer that the action head reads in place
    (no [B,25,576,896] regroup copy, finetune.py:396-409); the lm_head (modeling_prismatic.py:680-686, discarded by
    the reference) and the last ViT block (output unused) are never computed;
  * trainable parameters (action head, proprio projector, action queries) are views into ONE flat bf16 buffer
    grouped by kind across the 24 blocks -> batched GEMMs over layers, one AdamW launch, one gradient bucket.
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch

from . import ops
from .ops import ACT_GELU, ACT_GELU_TANH, ACT_NONE, ACT_RELU, ACT_SWIGLU, BF16

NUM_TOKENS = 64          # prismatic/vla/constants.py:15
IGNORE_INDEX = -100


def rup(n: int, m: int) -> int:
    return (n + m - 1) // m * m


# ------------------------------------------------------------------------------------------------ configs
@dataclass
class ViTCfg:
    d: int = 1152
    depth: int = 27
    heads: int = 16
    mlp: int = 4304
    patch: int = 14
    img: int = 224
    n_prefix: int = 0          # 0 SigLIP; 5 DINOv2 (cls + 4 reg)
    layerscale: bool = False
    eps: float = 1e-6
    gelu_tanh: bool = False

    @property
    def n_patches(self):
        return (self.img // self.patch) ** 2

    def as_oracle(self):
        return dict(d=self.d, depth=self.depth, heads=self.heads, mlp=self.mlp, patch=self.patch, n_prefix=self.n_prefix,
                    layerscale=self.layerscale, eps=self.eps, gelu_tanh=self.gelu_tanh)


SIGLIP_SO400M = ViTCfg(1152, 27, 16, 4304, 14, 224, 0, False)
DINOV2_L_REG4 = ViTCfg(1024, 24, 16, 4096, 14, 224, 5, True)


@dataclass
class LLMCfg:
    d: int = 896
    n_layers: int = 24
    heads: int = 14
    kv_heads: int = 2
    dh: int = 64
    inter: int = 4864
    eps: float = 1e-6
    theta: float = 1e6
    vocab: int = 151936

    def as_oracle(self):
        return dict(n_layers=self.n_layers, heads=self.heads, kv_heads=self.kv_heads, dh=self.dh, eps=self.eps, theta=self.theta)


@dataclass
class VLACfg:
    vit: List[ViTCfg] = field(default_factory=lambda: [SIGLIP_SO400M])
    llm: LLMCfg = field(default_factory=LLMCfg)
    n_img: int = 1
    num_blocks: int = 24       # MLPResNet(num_blocks=24), action_heads.py:35
    action_dim: int = 7
    chunk: int = 8
    proprio_dim: int = 8
    pro: bool = True

    @property
    def fused(self):
        return len(self.vit) == 2

    @property
    def n_patches(self):
        return self.vit[0].n_patches * self.n_img

    @property
    def vis_dim(self):
        return sum(v.d for v in self.vit)


def config2() -> VLACfg:
    """BASELINE.json configs[1]: SigLIP-224 + Qwen2.5-0.5B, adapter-only, 1 image."""
    return VLACfg()


def tiny_config() -> VLACfg:
    """BASELINE.json configs[0] 'prismatic-tiny': 2 useful ViT-T blocks + 2-layer 256-d LLM + adapter.
    The head reads hidden_states[i+1] for every block i (action_heads.py:117-118), so it can have at most
    n_layers blocks: 2 here (the reference hard-codes 24 and therefore needs >= 24 LLM layers)."""
    return VLACfg(vit=[ViTCfg(192, 3, 3, 768, 14, 56, 0, False)],
                  llm=LLMCfg(256, 2, 4, 2, 64, 512, 1e-6, 1e6, 1024), num_blocks=2)


def config5_backbone() -> VLACfg:
    """BASELINE.json configs[4]'s backbone at full size: DINOv2-L/14 (reg4) + SigLIP-so400m fused vision, Qwen2.5-1.5B language
    model (28 layers, d 1536, 12 heads of 128, 2 KV heads, MLP 8960), Pro action head on 24 of the 28 hidden states.  (The
    config's LoRA + fp8 training mode is a separate matter: bench.py runs this backbone adapter-only.)"""
    return VLACfg(vit=[DINOV2_L_REG4, SIGLIP_SO400M], llm=LLMCfg(1536, 28, 12, 2, 128, 8960, 1e-6, 1e6, 151936))


def qwen15b_geometry_config(n_layers: int = 2) -> VLACfg:
    """BASELINE.json configs[4]'s language-model GEOMETRY at plumbing depth: Qwen2.5-1.5B's layer (d 1536, 12 heads of 128,
    2 KV heads, MLP 8960) - head dim 128 (unfused RoPE, the 128-wide attention kernels) and a 7 x 1536-wide first head
    LayerNorm - under a tiny ViT, `n_layers` layers and as many head blocks."""
    return VLACfg(vit=[ViTCfg(192, 3, 3, 768, 14, 56, 0, False)],
                  llm=LLMCfg(1536, n_layers, 12, 2, 128, 8960, 1e-6, 1e6, 1024), num_blocks=n_layers)


def tiny_fused_config() -> VLACfg:
    """Plumbing-size version of the reference's default setup: DINOv2-like backbone (cls + 4 register tokens,
    LayerScale) + SigLIP-like backbone, fused 3-layer projector, two images per sample."""
    return VLACfg(vit=[ViTCfg(192, 3, 3, 768, 14, 56, 5, True), ViTCfg(128, 3, 2, 512, 14, 56, 0, False)], n_img=2,
                  llm=LLMCfg(256, 2, 4, 2, 64, 512, 1e-6, 1e6, 1024), num_blocks=2)


def dinosiglip_05b_config(n_img: int = 2) -> VLACfg:
    """The reference's documented recipe (README.md:254-274: ``--vlm_path .../prism-qwen25-extra-dinosiglip-224px-0_5b
    --num_images_in_input 2``): DINOv2-L/14 reg4 + SigLIP-so400m fused vision, 3-layer projector, Qwen2.5-0.5B."""
    return VLACfg(vit=[DINOV2_L_REG4, SIGLIP_SO400M], llm=LLMCfg(), n_img=n_img)


def _config5_two_images() -> VLACfg:
    c = config5_backbone()
    c.n_img = 2
    return c


# names accepted by ``finetune.py --backbone`` / ``bench.py --backbone`` (n_img follows --num_images_in_input where given)
NAMED_CONFIGS = {"config2": config2, "dinosiglip-0_5b": dinosiglip_05b_config, "config5": config5_backbone, "tiny": tiny_config,
                 "tiny_fused": tiny_fused_config, "qwen15b-geometry": qwen15b_geometry_config}


# ------------------------------------------------------------------------------------------------ frozen ViT
class ViT:
    """timm VisionTransformer forward up to block depth-2, no final norm (modeling_prismatic.py:120-144,196-237;
    block structure film_vit_wrapper.py:69-75).  Frozen: forward only."""

    def __init__(self, cfg: ViTCfg, sd: Dict[str, torch.Tensor], device):
        self.cfg = cfg
        d, P = cfg.d, cfg.patch
        g = lambda k: sd[k].to(device=device, dtype=BF16).contiguous()
        self.kpe = rup(3 * P * P, 64)
        wpe = torch.zeros(d, self.kpe, device=device, dtype=BF16)
        wpe[:, :3 * P * P] = g("patch_embed.proj.weight").reshape(d, -1)
        self.wpe, self.bpe = wpe, g("patch_embed.proj.bias")
        self.pos = g("pos_embed").reshape(-1, d)
        assert self.pos.shape[0] == cfg.n_patches, "pos_embed covers patches only (no_embed_class / no cls token)"
        self.prefix = None
        if cfg.n_prefix:
            toks = [g("cls_token").reshape(1, d)] + ([g("reg_token").reshape(-1, d)] if "reg_token" in sd else [])
            self.prefix = torch.cat(toks, 0)
            assert self.prefix.shape[0] == cfg.n_prefix
        self.mlp_pad = rup(cfg.mlp, 128)
        self.blocks = []
        for i in range(cfg.depth - 1):          # the last block's output is never used
            p = f"blocks.{i}."
            w1 = torch.zeros(self.mlp_pad, d, device=device, dtype=BF16)
            w1[:cfg.mlp] = g(p + "mlp.fc1.weight")
            b1 = torch.zeros(self.mlp_pad, device=device, dtype=BF16)
            b1[:cfg.mlp] = g(p + "mlp.fc1.bias")
            w2 = torch.zeros(d, self.mlp_pad, device=device, dtype=BF16)
            w2[:, :cfg.mlp] = g(p + "mlp.fc2.weight")
            wproj, bproj, b2 = g(p + "attn.proj.weight"), g(p + "attn.proj.bias"), g(p + "mlp.fc2.bias")
            blk = dict(n1w=g(p + "norm1.weight"), n1b=g(p + "norm1.bias"), wqkv=g(p + "attn.qkv.weight"),
                       bqkv=g(p + "attn.qkv.bias"), wproj=wproj, bproj=bproj, n2w=g(p + "norm2.weight"),
                       n2b=g(p + "norm2.bias"), w1=w1, b1=b1, w2=w2, b2=b2)
            if cfg.layerscale:                  # LayerScale (modeling_prismatic.py:58-66): the parameters themselves ...
                blk["ls1"], blk["ls2"] = g(p + "ls1.scale_factor"), g(p + "ls2.scale_factor")
            self.blocks.append(blk)
        # ... and, for the FROZEN forward (adapter-only fine-tune, inference), folded into the projections: y = ls * (W x + b) =
        # (ls W) x + ls b - one bf16 rounding point moves (scale applied to the weight instead of to the bf16 output), no extra
        # kernel on the step.  A trainer of the backbone (LoRA / full fine-tune) calls fold_layerscale(False): ViT.forward then
        # applies the scale as the reference does (vla_layerscale_fwd), on the weights that are being trained.
        self.ls_folded = False
        self.fold_layerscale(True)

    def fold_layerscale(self, on: bool):
        if not self.cfg.layerscale or on == self.ls_folded:
            self.ls_folded = on and self.cfg.layerscale
            return
        for b in self.blocks:
            if on:
                ls1, ls2 = b["ls1"].float(), b["ls2"].float()
                b["wproj_f"], b["bproj_f"] = (b["wproj"].float() * ls1[:, None]).to(BF16), (b["bproj"].float() * ls1).to(BF16)
                b["w2_f"], b["b2_f"] = (b["w2"].float() * ls2[:, None]).to(BF16), (b["b2"].float() * ls2).to(BF16)
            else:
                for k in ("wproj_f", "bproj_f", "w2_f", "b2_f"):
                    b.pop(k, None)
        self.ls_folded = on

    def enable_fp8(self):
        """BASELINE configs[4] 'fp8 MFMA weight path', first part: the frozen qkv and fc1 weights as OCP e4m3 with one scale per
        output channel; their inputs come out of LayerNorm already quantised per row (vla_layernorm_fwd_q8), the products run on
        the fp8 MFMA.  proj / fc2 (inputs produced by attention / a GEMM epilogue: no row statistics at hand) stay bf16."""
        for b in self.blocks:
            b["wqkv_q"], b["wqkv_s"] = ops.quant_fp8_rows(b["wqkv"])
            b["w1_q"], b["w1_s"] = ops.quant_fp8_rows(b["w1"])
        self.fp8 = True

    def _ln_q8(self, x, w, b_):
        rows, cols = x.shape
        if getattr(self, "_q8", None) is None or self._q8.shape != (rows, cols):
            self._q8 = torch.empty(rows, cols, device=x.device, dtype=torch.uint8)
            self._qs = torch.empty(rows, device=x.device, dtype=torch.float32)
        return ops.layernorm_fwd_q8(x, w, b_, self.cfg.eps, self._q8, self._qs)

    def forward(self, pixels: torch.Tensor, c0: int, out: torch.Tensor, c_group=None):
        """pixels [B, C, H, W] (channels c0..c0+2 used) -> writes patch features into ``out`` (a [B*Np, d] window,
        possibly a column slice of the fused feature buffer)."""
        cfg = self.cfg
        B, Np, d, T = pixels.shape[0], cfg.n_patches, cfg.d, cfg.n_patches + cfg.n_prefix
        cols = ops.im2col_patch(pixels, c0, cfg.patch, self.kpe)
        if cfg.n_prefix:
            x = torch.empty(B, T, d, device=pixels.device, dtype=BF16)
            tmp = ops.gemm_nt(cols, self.wpe, bias=self.bpe, residual=self.pos, res_mod=Np)
            x[:, cfg.n_prefix:] = tmp.view(B, Np, d)
            x[:, :cfg.n_prefix] = self.prefix
            x = x.view(B * T, d)
        else:
            x = ops.gemm_nt(cols, self.wpe, bias=self.bpe, residual=self.pos, res_mod=Np)
        act = ACT_GELU_TANH if cfg.gelu_tanh else ACT_GELU
        dh = d // cfg.heads
        nb = len(self.blocks)
        fp8 = getattr(self, "fp8", False)
        for i, b in enumerate(self.blocks):
            if fp8:
                q8, qs = self._ln_q8(x, b["n1w"], b["n1b"])
                qkv = ops.gemm_nt(q8, b["wqkv_q"], bias=b["bqkv"], fp8=(qs, b["wqkv_s"])).view(B, T, 3 * d)
            else:
                h = ops.layernorm_fwd(x, b["n1w"], b["n1b"], cfg.eps)
                qkv = ops.gemm_nt(h, b["wqkv"], bias=b["bqkv"]).view(B, T, 3 * d)
            a = ops.attn_fwd(qkv[:, :, :d], qkv[:, :, d:2 * d], qkv[:, :, 2 * d:], cfg.heads, cfg.heads, dh, False)
            unf = cfg.layerscale and not self.ls_folded          # scale applied to the bf16 projection output, as the reference does
            wp, bp, w2, b2 = (b["wproj"], b["bproj"], b["w2"], b["b2"]) if (unf or not cfg.layerscale) else (b["wproj_f"], b["bproj_f"], b["w2_f"], b["b2_f"])
            if unf:
                ops.layerscale_fwd(ops.gemm_nt(a.view(B * T, d), wp, bias=bp), b["ls1"], x, out=x)
            else:
                ops.gemm_nt(a.view(B * T, d), wp, bias=bp, residual=x, out=x)
            if fp8:
                q8, qs = self._ln_q8(x, b["n2w"], b["n2b"])
                m = ops.gemm_nt(q8, b["w1_q"], bias=b["b1"], act=act, fp8=(qs, b["w1_s"]))
            else:
                h = ops.layernorm_fwd(x, b["n2w"], b["n2b"], cfg.eps)
                m = ops.gemm_nt(h, b["w1"], bias=b["b1"], act=act)
            if unf:
                ops.layerscale_fwd(ops.gemm_nt(m, w2, bias=b2), b["ls2"], x, out=x)
                if i == nb - 1 and cfg.n_prefix == 0:
                    assert c_group is None
                    out.copy_(x)
            elif i == nb - 1 and cfg.n_prefix == 0:
                ops.gemm_nt(m, w2, bias=b2, residual=x, out=out, c_group=c_group)
            else:
                ops.gemm_nt(m, w2, bias=b2, residual=x, out=x)
        if cfg.n_prefix:
            assert c_group is None
            out.view(B, Np, -1).copy_(x.view(B, T, d)[:, cfg.n_prefix:])
        return out


# ------------------------------------------------------------------------------------------------ frozen LLM
class LLM:
    """Qwen2 decoder stack (transformers Qwen2ForCausalLM; reference call site modeling_prismatic.py:644-655),
    forward with hidden-state taps and explicit dX backward (weights frozen: adapter-only fine-tune)."""

    def __init__(self, cfg: LLMCfg, sd: Dict[str, torch.Tensor], device):
        self.cfg, self.device = cfg, device
        D, H, KV, dh, I = cfg.d, cfg.heads, cfg.kv_heads, cfg.dh, cfg.inter
        assert D % 64 == 0 and I % 64 == 0 and (H + 2 * KV) * dh % 64 == 0 and I % 16 == 0
        g = lambda k: sd[k].to(device=device, dtype=BF16).contiguous()
        self.layers = []
        for i in range(cfg.n_layers):
            p = f"layers.{i}."
            wqkv = torch.cat([g(p + "self_attn.q_proj.weight"), g(p + "self_attn.k_proj.weight"), g(p + "self_attn.v_proj.weight")], 0)
            bqkv = torch.cat([g(p + "self_attn.q_proj.bias"), g(p + "self_attn.k_proj.bias"), g(p + "self_attn.v_proj.bias")], 0)
            wg, wu = g(p + "mlp.gate_proj.weight"), g(p + "mlp.up_proj.weight")
            wgu = torch.stack([wg.view(I // 16, 16, D), wu.view(I // 16, 16, D)], 1).reshape(2 * I, D).contiguous()
            wo, wd = g(p + "self_attn.o_proj.weight"), g(p + "mlp.down_proj.weight")
            self.layers.append(dict(n1=g(p + "input_layernorm.weight"), n2=g(p + "post_attention_layernorm.weight"),
                                    wqkv=wqkv.contiguous(), bqkv=bqkv.contiguous(), wo=wo, wgu=wgu, wd=wd,
                                    wqkvT=wqkv.t().contiguous(), woT=wo.t().contiguous(), wguT=wgu.t().contiguous(),
                                    wdT=wd.t().contiguous()))
        self.norm = g("norm.weight")
        self.embed = g("embed_tokens.weight")
        self._buf_key = None
        self.gu_row0 = 0

    def _alloc(self, B: int, S: int):
        if self._buf_key == (B, S):
            return
        c, dev = self.cfg, self.device
        n, M, D = c.n_layers, B * S, c.d
        W = (c.heads + 2 * c.kv_heads) * c.dh
        e = lambda *s, dt=BF16: torch.empty(*s, device=dev, dtype=dt)
        # slots 0..n-1: residual stream entering layer i (slot 0 = inputs_embeds); slot n = FINAL-NORM output
        # (= hidden_states[n] of the HF convention); slot n+1 = raw output of the last layer.
        self.HS = e(n + 2, B, S, D)
        self.X1, self.QKV, self.AO = e(n, M, D), e(n, M, W), e(n, M, c.heads * c.dh)
        self.GU = e(n, M, 2 * c.inter)
        self.LSE = e(n, B, c.heads, S, dt=torch.float32)
        self.R1, self.R2, self.RF = e(n, M, dt=torch.float32), e(n, M, dt=torch.float32), e(M, dt=torch.float32)
        self.nbuf, self.hbuf = e(M, D), e(M, c.inter)
        self.q8, self.qs = e(M, D, dt=torch.uint8), e(M, dt=torch.float32)      # fp8 form of the normalised rows (enable_fp8)
        self.d_a, self.d_b, self.d_h, self.d_gu, self.d_qkv, self.d_n = e(M, D), e(M, D), e(M, c.inter), e(M, 2 * c.inter), e(M, W), e(M, D)
        self.cos, self.sin = ops.rope_half_tables(S, c.dh, c.theta, dev)
        self._buf_key = (B, S)

    def out_slot(self, i: int) -> int:
        """HS slot holding the output of layer i (0-based)."""
        n = self.cfg.n_layers
        return i + 1 if i < n - 1 else n + 1

    def forward(self, B: int, S: int, kmask_u8: torch.Tensor, keep_from_row: int = 0, n_run: Optional[int] = None):
        """HS[0] must already hold inputs_embeds [B,S,D].  Fills HS[1..n] (HF hidden_states semantics).  n_run < n_layers: only the
        first n_run layers (hidden_states[1..n_run]: all the action head reads when it has fewer blocks than the LLM has layers)."""
        n = self.cfg.n_layers
        n_run = n if n_run is None else n_run
        self.fwd_begin(B, S, kmask_u8, keep_from_row)
        for i in range(n_run):
            self.fwd_layer(i)
        if n_run == n:
            self.fwd_final()

    def fwd_begin(self, B: int, S: int, kmask_u8: torch.Tensor, keep_from_row: int = 0):
        """keep_from_row: the backward of this forward will only visit the rows >= keep_from_row of every sequence
        (LLM.backward row0 >= keep_from_row); backward-only tensors skip the rows below it."""
        self.kmask, self.B, self.S = kmask_u8, B, S
        self.gu_row0 = keep_from_row

    def fwd_layer(self, i: int, b0: int = 0, b1: Optional[int] = None):
        """Layer i on the samples [b0, b1) of the batch (default: all).  Every op is row- or sample-wise, so disjoint sample
        ranges can run on different streams (the step schedule pipelines the two halves of the batch)."""
        c, S = self.cfg, self.S
        b1 = self.B if b1 is None else b1
        B, r0, r1 = b1 - b0, b0 * S, b1 * S
        M, D, H, KV, dh = B * S, c.d, c.heads, c.kv_heads, c.dh
        L = self.layers[i]
        x = self.HS[i].view(-1, D)[r0:r1]
        nbuf, hbuf = self.nbuf[r0:r1], self.hbuf[r0:r1]
        fp8 = getattr(self, "fp8", False)
        q8, qs = self.q8[r0:r1], self.qs[r0:r1]
        qkv = self.QKV[i][r0:r1]
        # (RMSNorm folded into the neighbouring GEMMs - 47 launches fewer, built and measured SLOWER in round 3 - left the tree in
        #  round 4: DESIGN section 4, tools/diag/gemm256_pruned_paths.patch)
        if fp8:
            ops.rmsnorm_fwd_q8(x, L["n1"], c.eps, q8, qs, rstd=self.R1[i][r0:r1])
        else:
            self._rms(x, L["n1"], nbuf, self.R1[i][r0:r1])
        if fp8 and dh in (64, 128):
            ops.gemm_nt(q8, L["wqkv_q"], bias=L["bqkv"], out=qkv, rope=(1, self.cos, self.sin, S, dh, (H + KV) * dh), fp8=(qs, L["wqkv_s"]))
        elif fp8:
            ops.gemm_nt(q8, L["wqkv_q"], bias=L["bqkv"], out=qkv, fp8=(qs, L["wqkv_s"]))
            ops.rope_half_(qkv[:, :H * dh], self.cos, self.sin, S, H, dh)
            ops.rope_half_(qkv[:, H * dh:(H + KV) * dh], self.cos, self.sin, S, KV, dh)
        elif dh in (64, 128):   # RoPE fused into the projection's epilogue (head dim 128: the 128-row kernel's column map, round 4)
            ops.gemm_nt(nbuf, L["wqkv"], bias=L["bqkv"], out=qkv, rope=(1, self.cos, self.sin, S, dh, (H + KV) * dh))
        else:
            ops.gemm_nt(nbuf, L["wqkv"], bias=L["bqkv"], out=qkv)
            ops.rope_half_(qkv[:, :H * dh], self.cos, self.sin, S, H, dh)
            ops.rope_half_(qkv[:, H * dh:(H + KV) * dh], self.cos, self.sin, S, KV, dh)
        self._attn_fwd(qkv.view(B, S, -1), i, b0, b1, S)
        x1 = self.X1[i][r0:r1]
        ops.gemm_nt(self.AO[i][r0:r1], L["wo"], residual=x, out=x1)
        # the pre-activations are kept for the backward only: rows below the live window are never read again
        if fp8:
            ops.rmsnorm_fwd_q8(x1, L["n2"], c.eps, q8, qs, rstd=self.R2[i][r0:r1])
            ops.gemm_nt(q8, L["wgu_q"], act=ACT_SWIGLU, out=self.GU[i][r0:r1], out2=hbuf,
                        c_live=(S, self.gu_row0) if self.gu_row0 else None, fp8=(qs, L["wgu_s"]))
        else:
            self._rms(x1, L["n2"], nbuf, self.R2[i][r0:r1])
            ops.gemm_nt(nbuf, L["wgu"], act=ACT_SWIGLU, out=self.GU[i][r0:r1], out2=hbuf,
                        c_live=(S, self.gu_row0) if self.gu_row0 else None)
        ops.gemm_nt(hbuf, L["wd"], residual=x1, out=self.HS[self.out_slot(i)].view(-1, D)[r0:r1])

    def fwd_final(self, b0: int = 0, b1: Optional[int] = None):
        """hidden_states[n] = final RMSNorm of the last layer's output (HF convention)."""
        n, D, S = self.cfg.n_layers, self.cfg.d, self.S
        r0, r1 = b0 * S, (self.B if b1 is None else b1) * S
        self._rms(self.HS[n + 1].view(-1, D)[r0:r1], self.norm, self.HS[n].view(-1, D)[r0:r1], self.RF[r0:r1])

    def enable_fp8(self):
        """fp8 weight path (see ViT.enable_fp8): the frozen q|k|v and gate/up weights in e4m3, their inputs quantised inside
        RMSNorm (vla_rmsnorm_fwd_q8); o-proj / down stay bf16.  Forward only: the dX products keep the bf16 W^T."""
        for L in self.layers:
            L["wqkv_q"], L["wqkv_s"] = ops.quant_fp8_rows(L["wqkv"])
            L["wgu_q"], L["wgu_s"] = ops.quant_fp8_rows(L["wgu"])
        self.fp8 = True

    def _rms(self, x, w, out, rstd):
        ops.N.check(ops._lib().vla_rmsnorm_fwd(ops._st(), ops._p(x), ops._p(w), ops._p(out), ops._p(rstd), x.shape[0],
                                               x.shape[1], self.cfg.eps), "rmsnorm_fwd")

    def _attn_views(self, t3):
        c = self.cfg
        a, b = c.heads * c.dh, (c.heads + c.kv_heads) * c.dh
        return t3[:, :, :a], t3[:, :, a:b], t3[:, :, b:]

    def _attn_fwd(self, q3, i, b0, b1, S):
        import ctypes as C
        c = self.cfg
        q, k, v = self._attn_views(q3)
        o = self.AO[i].view(self.B, S, -1)[b0:b1]
        km = self.kmask[b0:b1] if self.kmask is not None else None
        d = ops._attn_desc(q, k, v, o, self.LSE[i][b0:b1], km, True, c.dh ** -0.5, c.heads, c.kv_heads, c.dh)
        ops.N.check(ops._lib().vla_attn_fwd(ops._st(), C.byref(d)), "attn_fwd")

    # ---- backward: dX only (frozen weights), restricted to the LIVE rows ------------------------------------------
    # The only trainable tensor upstream of the LLM is `action_queries`, spliced in at sequence positions >= r_first
    # (after tok0 + patches + prompt).  Causal attention makes row i of every hidden state a function of the input rows
    # <= i only, so d loss / d inputs_embeds[rows >= r0] depends on d loss / d hidden[rows >= r0] alone, for any
    # r0 <= r_first: the gradient rows < r0 flow exclusively into frozen inputs (patch / prompt embeddings) and are dead.
    # torch.autograd (the reference) computes them anyway because it prunes by tensor, not by row.  `row0` selects the
    # window; every op below works on compact [B * (S - row0), .] gradients and reads the forward's tensors through
    # row-window addressing.  row0 = 0 is the plain full-sequence backward (needed as soon as ViT / projector / LoRA
    # weights train).  The surviving gradients are the same numbers either way (tests/test_engine_gpu.py).
    def backward(self, dHS: torch.Tensor, B: int, S: int, row0: int = 0, n_run: Optional[int] = None) -> torch.Tensor:
        """dHS [n+1, B, S - row0, D]: gradient w.r.t. rows >= row0 of hidden_states[i] (i = 0..n, HF convention; [0]
        unused).  Returns the gradient w.r.t. rows >= row0 of inputs_embeds, [B, S - row0, D] (frozen weights: no dW).
        n_run < n_layers: the layers above n_run never reached the loss (forward(n_run=)): the chain starts at layer n_run - 1."""
        n_run = self.cfg.n_layers if n_run is None else n_run
        self.bwd_begin(dHS, row0, n_run)
        for i in range(n_run - 1, -1, -1):
            self.bwd_layer(i, dHS)
        return self.bwd_result()

    def bwd_begin(self, dHS: torch.Tensor, row0: int = 0, n_run: Optional[int] = None):
        n, S, D = self.cfg.n_layers, self.S, self.cfg.d
        self._n_run = n if n_run is None else n_run
        assert 0 <= row0 < S and row0 % 32 == 0, "live-row window must start on a multiple of 32"
        assert row0 >= self.gu_row0, "the forward dropped backward-only rows this window needs"
        self.r0, self.Rl = row0, S - row0
        Mr = self.B * self.Rl
        assert tuple(dHS.shape[1:]) == (self.B, self.Rl, D)
        self._win = (self.Rl, S, row0)                        # (rows per sequence, sequence rows, first row)
        if self._n_run == n:
            self._d = ops.rmsnorm_bwd(dHS[n].view(Mr, D), self.HS[n + 1].view(-1, D), self.norm, self.RF, out=self.d_a[:Mr],
                                      x_rows=self._win)
        else:           # hidden_states[n_run] is a raw layer output (no final norm behind it): its gradient is the head's alone
            self._d = ops.copy2d(dHS[self._n_run].view(Mr, D), self.d_a[:Mr], Mr, D, D, D)
        self._other = self.d_b[:Mr]

    def bwd_layer(self, i: int, dHS: torch.Tensor):
        c, B, S, r0, R = self.cfg, self.B, self.S, self.r0, self.Rl
        n, M, D, H, KV, dh, I = c.n_layers, B * S, c.d, c.heads, c.kv_heads, c.dh, c.inter
        Mr = B * R
        L, d, other = self.layers[i], self._d, self._other
        if i < self._n_run - 1:                         # head contribution to the output of layer i (the top layer's came in bwd_begin)
            ops.add_(d, dHS[i + 1].view(Mr, D))
        gu_live = self.GU[i][r0:]                       # first sequence's window; the others by row-group addressing
        d_gu, d_n = self.d_gu[:Mr], self.d_n[:Mr]
        if I % 64 == 0 and not os.environ.get("VLA_NO_FUSED_SWIGLU_BWD"):   # dH GEMM + SwiGLU backward in its epilogue
            ops.gemm_swiglu_bwd(d, L["wdT"], gu_live, out=d_gu, gu_group=(R, S * 2 * I))
        else:
            ops.gemm_nt(d, L["wdT"], out=self.d_h[:Mr])
            gu_c = self.GU[i].view(B, S, 2 * I)[:, r0:].contiguous().view(Mr, 2 * I) if r0 else self.GU[i]
            ops.swiglu_bwd(self.d_h[:Mr], gu_c, out=d_gu)
        ops.gemm_nt(d_gu, L["wguT"], out=d_n)
        d1 = ops.rmsnorm_bwd(d_n, self.X1[i], L["n2"], self.R2[i], dres=d, out=other, x_rows=self._win)
        dao = ops.gemm_nt(d1, L["woT"], out=d_n)
        q, k, v = self._attn_views(self.QKV[i].view(B, S, -1))
        W = self.QKV.shape[-1]
        d_qkv = self.d_qkv[:Mr]
        dq, dk, dv = self._attn_views(d_qkv.view(B, R, W))
        ops.attn_bwd(dao.view(B, R, -1), q[:, r0:], k, v, self.AO[i].view(B, S, -1)[:, r0:], self.LSE[i], H, KV, dh, True,
                     self.kmask, dq=dq, dk=dk, dv=dv, rope=(self.cos, self.sin) if dh in (64, 128) else None, row0=r0)
        if dh not in (64, 128):
            cs, sn = self.cos[r0:], self.sin[r0:]
            ops.rope_half_(d_qkv[:, :H * dh], cs, sn, R, H, dh, sign=-1)
            ops.rope_half_(d_qkv[:, H * dh:(H + KV) * dh], cs, sn, R, KV, dh, sign=-1)
        ops.gemm_nt(d_qkv, L["wqkvT"], out=d_n)
        d_new = ops.rmsnorm_bwd(d_n, self.HS[i].view(M, D), L["n1"], self.R1[i], dres=d1, out=d, x_rows=self._win)
        self._d, self._other = d_new, d1

    def bwd_result(self) -> torch.Tensor:
        return self._d.view(self.B, self.Rl, self.cfg.d)


# ------------------------------------------------------------------------------------------------ trainable params
class FlatParams:
    """Named bf16 views into one flat buffer (+ matching flat grad / AdamW state buffers)."""

    def __init__(self, spec: List[Tuple[str, Tuple[int, ...]]], device):
        self.offsets, off = {}, 0
        for name, shape in spec:
            n = math.prod(shape)
            self.offsets[name] = (off, shape)
            off += rup(n, 8)                      # every region 16-B aligned
        self.numel = off
        self.data = torch.zeros(off, device=device, dtype=BF16)
        self.grad = torch.zeros(off, device=device, dtype=BF16)
        self.m = torch.zeros(off, device=device, dtype=BF16)
        self.v = torch.zeros(off, device=device, dtype=BF16)

    def view(self, name, buf=None):
        off, shape = self.offsets[name]
        return (self.data if buf is None else buf)[off:off + math.prod(shape)].view(shape)

    def g(self, name):
        return self.view(name, self.grad)


class Head:
    """L1RegressionActionHead + ProprioProjector + action_queries: the trainable set of the adapter-only fine-tune
    (action_heads.py:21-121; projectors.py:6-24; modeling_prismatic.py:375-376).  cfg.pro selects the block:
    MLPResNetBlock_Pro (:287-410, the reference default) - separate k/v projections per segment, RoPE on q/k - or the
    original MLPResNetBlock (:168-283) - ONE k_proj / v_proj shared by the three segments, no RoPE.  Both put the
    tanh(gating_factor) on the task-token segment and share every kernel; the original block simply reads its k|v
    weights (rows D..3D of the fused q|k|v matrix) for all three segments and sums their three gradients."""

    H = 8

    def __init__(self, cfg: VLACfg, device):
        self.cfg, self.device, self.pro = cfg, device, cfg.pro
        D, nb, Da, Pd = cfg.llm.d, cfg.num_blocks, cfg.action_dim, cfg.proprio_dim
        assert D % 64 == 0 and (D // self.H) % 8 == 0
        self.D, self.nb, self.Din = D, nb, Da * D
        assert self.Din % 64 == 0
        seg = [("w_adp", (nb, 2 * D, D)), ("b_adp", (nb, 2 * D)),    # k_adapter | v_adapter
               ("w_task", (nb, 2 * D, D)), ("b_task", (nb, 2 * D))] if cfg.pro else []   # k_task | v_task
        spec = [("w_x", (nb, 3 * D, D)), ("b_x", (nb, 3 * D))] + seg + [   # q_proj | k_self | v_self   (orig: q | k | v)
                ("w_o", (nb, D, D)), ("b_o", (nb, D)), ("w_ffn", (nb, D, D)), ("b_ffn", (nb, D)),
                ("ln_w", (nb, D)), ("ln_b", (nb, D)), ("gate", (nb, 8)),
                ("ln1_w", (self.Din,)), ("ln1_b", (self.Din,)), ("fc1_w", (D, self.Din)), ("fc1_b", (D,)),
                ("ln2_w", (D,)), ("ln2_b", (D,)), ("fc2_w", (Da, D)), ("fc2_b", (Da,)),
                ("p_fc1_w", (D, Pd)), ("p_fc1_b", (D,)), ("p_fc2_w", (D, D)), ("p_fc2_b", (D,)),
                ("action_queries", (NUM_TOKENS, D))]
        self.P = FlatParams(spec, device)
        self.film = {}            # film_gen.0.{weight,bias}: in the state dict, never used, never updated (:327-329)
        # transposed copies for the dX products (rebuilt after every optimiser step)
        z = lambda *s: torch.zeros(*s, device=device, dtype=BF16)
        self.T = dict(w_x=z(nb, D, 3 * D), w_o=z(nb, D, D), w_ffn=z(nb, D, D), p_fc2_w=z(D, D))
        if cfg.pro:
            self.T.update(w_adp=z(nb, D, 2 * D), w_task=z(nb, D, 2 * D))
        self.fc2T = z(D, 64)                       # fc2^T zero-padded to K=64
        self.pfc1_pad = z(D, 64)                   # proprio fc1 weight zero-padded to K=64
        self.dirty = True
        self._key = None
        self.rope_tab = None     # f32 [max(T, Ka, Kt), dh] cos/sin tables (positions restart per segment: one table serves all)
        # attention weights rounded to bf16 AFTER normalisation, as the reference's bf16 softmax emits them (action_heads.py:397): a second
        # pass over the keys (vla_head_attn_desc.ref_softmax).  Off by default: +N us per block on a chain the step's turn-around waits
        # for; the reference-run fixture tests switch it on (tests/test_head_fixture_gpu.py prints both distances)
        self.ref_softmax = bool(int(os.environ.get("VLA_HEAD_REF_SOFTMAX", "0")))

    # ---- reference state-dict interop (file names / keys: finetune.py:527-572) -------------------------
    _BLK_PRO = [("q_proj", "w_x", "b_x", 0), ("k_self", "w_x", "b_x", 1), ("v_self", "w_x", "b_x", 2),
                ("k_adapter", "w_adp", "b_adp", 0), ("v_adapter", "w_adp", "b_adp", 1),
                ("k_task", "w_task", "b_task", 0), ("v_task", "w_task", "b_task", 1),
                ("o_proj", "w_o", "b_o", 0), ("ffn.1", "w_ffn", "b_ffn", 0)]
    _BLK_ORIG = [("q_proj", "w_x", "b_x", 0), ("k_proj", "w_x", "b_x", 1), ("v_proj", "w_x", "b_x", 2),
                 ("o_proj", "w_o", "b_o", 0), ("ffn.1", "w_ffn", "b_ffn", 0)]

    @property
    def _BLK(self):
        return self._BLK_PRO if self.pro else self._BLK_ORIG

    def _kv(self, which: str, i: int):
        """(weight [2D, D], bias [2D], transposed weight [D, 2D]) of the k|v projection of segment `which` in block i."""
        D = self.D
        if self.pro:
            return self.P.view("w_" + which)[i], self.P.view("b_" + which)[i], self.T["w_" + which][i]
        return self.P.view("w_x")[i, D:], self.P.view("b_x")[i, D:], self.T["w_x"][i][:, D:]

    def named_views(self, buf=None) -> Dict[str, torch.Tensor]:
        """Reference-named views ('model.mlp_resnet_blocks.N.q_proj.weight', ...) into the flat buffer."""
        P, D, out = self.P, self.D, {}
        v = lambda n: P.view(n, buf)
        for i in range(self.nb):
            pre = f"model.mlp_resnet_blocks.{i}."
            for name, w, b, k in self._BLK:
                out[pre + name + ".weight"] = v(w)[i, k * D:(k + 1) * D]
                out[pre + name + ".bias"] = v(b)[i, k * D:(k + 1) * D]
            out[pre + "ffn.0.weight"], out[pre + "ffn.0.bias"] = v("ln_w")[i], v("ln_b")[i]
            out[pre + "gating_factor"] = v("gate")[i, :1]
        for a, b in (("model.layer_norm1.weight", "ln1_w"), ("model.layer_norm1.bias", "ln1_b"), ("model.fc1.weight", "fc1_w"),
                     ("model.fc1.bias", "fc1_b"), ("model.layer_norm2.weight", "ln2_w"), ("model.layer_norm2.bias", "ln2_b"),
                     ("model.fc2.weight", "fc2_w"), ("model.fc2.bias", "fc2_b")):
            out[a] = v(b)
        return out

    def proprio_views(self, buf=None):
        v = lambda n: self.P.view(n, buf)
        return {"fc1.weight": v("p_fc1_w"), "fc1.bias": v("p_fc1_b"), "fc2.weight": v("p_fc2_w"), "fc2.bias": v("p_fc2_b")}

    def load_state_dicts(self, head_sd: Dict[str, torch.Tensor], proprio_sd: Dict[str, torch.Tensor],
                         action_queries: Optional[torch.Tensor] = None):
        nv = self.named_views()
        for k, t in nv.items():
            t.copy_(head_sd[k].to(self.device, BF16).reshape(t.shape))
        for k, t in head_sd.items():
            if "film_gen" in k:
                self.film[k] = t.to(self.device, BF16)
        for k, t in self.proprio_views().items():
            t.copy_(proprio_sd[k].to(self.device, BF16))
        if action_queries is not None:
            self.P.view("action_queries").copy_(action_queries.to(self.device, BF16))
        self.dirty = True

    def head_state_dict(self):
        sd = {k: v.clone() for k, v in self.named_views().items()}
        sd.update(self.film)
        return sd

    # `dirty` is raised whenever the flat parameter buffer changed (AdamW, checkpoint load); it expands into two stale
    # marks: the K-padded proprio fc1 weight the FORWARD reads, and the W^T operands only the BACKWARD reads.
    @property
    def dirty(self) -> bool:
        return self._stale_fwd or self._stale_bwd

    @dirty.setter
    def dirty(self, v: bool):
        self._stale_fwd = self._stale_bwd = bool(v)

    def refresh_forward_operands(self):
        if self._stale_fwd:
            pd = self.cfg.proprio_dim                  # (native strided copy: a sliced assignment is an ATen kernel on the step)
            ops.copy2d(self.P.view("p_fc1_w"), self.pfc1_pad, self.D, pd, pd, self.pfc1_pad.stride(0))
            self._stale_fwd = False

    def refresh_transposes(self):
        if not self._stale_bwd:
            return
        P = self.P
        for k in (("w_x", "w_adp", "w_task", "w_o", "w_ffn") if self.pro else ("w_x", "w_o", "w_ffn")):
            ops.transpose(P.view(k), out=self.T[k])
        ops.transpose(P.view("p_fc2_w"), out=self.T["p_fc2_w"])
        ops.transpose(P.view("fc2_w"), out=self.fc2T)             # [7, D] -> columns 0..6 of the zero-padded [D, 64]
        self._stale_bwd = False

    def _alloc(self, B: int, Kt: int):
        if self._key == (B, Kt):
            return
        D, nb, T, dev = self.D, self.nb, self.cfg.chunk, self.device
        Ka = NUM_TOKENS + 1
        e = lambda *s, dt=BF16: torch.empty(*s, device=dev, dtype=dt)
        z = lambda *s, dt=BF16: torch.zeros(*s, device=dev, dtype=dt)
        R = B * T
        self.Ka, self.Kt, self.R = Ka, Kt, R
        self.h_adp = e(nb, B, Ka, D)
        self.KV_adp, self.KV_task = e(nb, B * Ka, 2 * D), e(nb, B * Kt, 2 * D)
        self.dKV_adp, self.dKV_task = e(nb, B * Ka, 2 * D), e(nb, B * Kt, 2 * D)
        self.X = e(nb + 1, R, D)                       # block inputs; X[nb] = output of the last block
        self.QKVx, self.dQKVx = e(nb, R, 3 * D), e(nb, R, 3 * D)
        self.AOx, self.O2, self.LNo = e(nb, R, D), e(nb, R, D), e(nb, R, D)
        self.dO2, self.dFF = e(nb, R, D), e(nb, R, D)
        self.probs = e(nb, B, self.H, T, T + Ka + Kt, dt=torch.float32)
        self.stats = e(nb, R, 2, dt=torch.float32)
        # fp32 gradient accumulators (bias column sums, LayerNorm dw/db, gate): views of ONE buffer, zeroed by one fill
        bkeys = tuple(k for k in ("b_x", "b_adp", "b_task", "b_o", "b_ffn", "fc1_b", "fc2_b", "p_fc1_b", "p_fc2_b") if k in self.P.offsets)
        shapes = [("dgate", (nb,)), ("ln_dw", (nb, D)), ("ln_db", (nb, D)), ("ln1_dw", (self.Din,)), ("ln1_db", (self.Din,)),
                  ("ln2_dw", (D,)), ("ln2_db", (D,)), ("d_pf", (B, D))] + [("b:" + k, tuple(self.P.offsets[k][1])) for k in bkeys]
        self.acc32 = z(sum(rup(math.prod(sh), 4) for _, sh in shapes), dt=torch.float32)
        off, views = 0, {}
        for name, sh in shapes:
            views[name] = self.acc32[off:off + math.prod(sh)].view(sh)
            off += rup(math.prod(sh), 4)
        self.dgate, self.ln_dw, self.ln_db = views["dgate"], views["ln_dw"], views["ln_db"]
        self.ln1_dw, self.ln1_db, self.ln2_dw, self.ln2_db = views["ln1_dw"], views["ln1_db"], views["ln2_dw"], views["ln2_db"]
        self.b_f32 = {k: views["b:" + k] for k in bkeys}
        self.d_pf32 = views["d_pf"]
        # gather / scatter row indices of the action-query hidden states (+ proprio slot) and the live-row guard: static buffers
        # filled by vla_head_index_prep (a captured step replays the kernel on the new batch's mask positions)
        self.row_idx_ka = torch.empty(B, Ka, device=dev, dtype=torch.int32)
        self.row_idx_live_ka = torch.empty(B, Ka, device=dev, dtype=torch.int32)
        self._idx_scratch = torch.empty(B, Ka, device=dev, dtype=torch.int32)
        self.guard = torch.zeros(1, device=dev, dtype=torch.float32)
        self.x_in = z(R, self.Din)
        self.pr_in = z(B, 64)
        # (the dW products read dY and X as they lie: vla_gemm_bf16_tn contracts over their rows - no transposed operand copies)
        self.h_task_c = e(nb, B * Kt, D) if Kt % 64 else None      # plumbing sizes only: task rows compacted (row groups need Kt % 64 == 0)
        self.dfc2 = e(64, D)
        self.dpfc1 = e(D, 64)                          # proprio fc1 weight gradient against the 64-column padded input (any proprio_dim)
        self.dh_adp = e(nb, B * Ka, D)
        self.dpad = z(R, 64)
        self.rope_tab = ops.rope_inter_tables(max(T, Ka, Kt), D // self.H, dev)
        self._key = (B, Kt)

    # ---- forward (action_heads.py:43-81, 111-121, 337-410) -------------------------------------------------
    def forward(self, HS: torch.Tensor, pos1: torch.Tensor, proprio: torch.Tensor, Np: int,
                noise: Optional[torch.Tensor] = None) -> torch.Tensor:
        """HS [>=nb+1, B, S, D] hidden states (HF indexing: block i reads HS[i+1]); pos1 int32 [B,64] = positions of
        the action-query hidden states in text coordinates (mask on labels[:,1:], finetune.py:351-353, 399-405);
        proprio [B, Pd]; noise [chunk, 7*D] or None (phase Inference).  Returns predicted actions [B, chunk, 7].
        Sequential composition of the per-layer pieces the pipelined schedule (VLAEngine.step_pipelined) interleaves
        with the LLM on a second stream."""
        self.fwd_begin(HS, pos1, proprio, Np, noise)
        for i in range(self.nb):
            self.fwd_layer(i)
        return self.fwd_end()

    def fwd_begin(self, HS, pos1, proprio, Np, noise=None):
        """Everything that does not depend on the LLM's output: buffers, proprio projector, input stage
        (zeros/noise -> LN -> fc1 -> ReLU, action_heads.py:60-72, 113-115)."""
        cfg, P = self.cfg, self.P
        B, S = HS.shape[1], HS.shape[2]
        T = cfg.chunk
        self._alloc(B, Np)
        self.refresh_forward_operands()
        self.HSref, self.Np, self.S, self.B, self.pos1 = HS, Np, S, B, pos1
        # proprio projector (projectors.py:19-24); proprio rounded to bf16 first (action_heads.py:53)
        assert proprio.dim() == 2 and proprio.stride(1) == 1 and proprio.dtype in (BF16, torch.float32)
        ops.copy2d(proprio, self.pr_in, B, cfg.proprio_dim, proprio.stride(0), 64)
        self.pp_pre = ops.gemm_nt(self.pr_in, self.pfc1_pad, bias=P.view("p_fc1_b"))
        self.pp_act = ops.gelu_fwd(self.pp_pre)
        self.pf = ops.gemm_nt(self.pp_act, P.view("p_fc2_w"), bias=P.view("p_fc2_b"))       # [B, D]
        # rows of the 64 action-query hidden states inside one [B*S, D] layer slab.  The adapter segment of every block =
        # [64 action-query hidden states | proprio token]: the gather writes straight into h_adp[i] (index -2 = leave the row
        # alone), the proprio token is filled in once for all blocks
        ops.head_index_prep(pos1, pos1, pos1, self.row_idx_ka, self._idx_scratch, None, B, S, Np, 0)
        ops.copy2d(self.pf, self.h_adp[0, 0, NUM_TOKENS], self.nb * B, self.D, self.D, self.Ka * self.D, src_mod=B)
        if noise is not None:
            assert tuple(noise.shape) == (T, self.Din) and noise.is_contiguous()
            ops.copy2d(noise, self.x_in, B * T, self.Din, self.Din, self.Din, src_mod=T)
        else:
            ops.zero_(self.x_in)
        self.x_ln, self.st1 = ops.layernorm_fwd(self.x_in, P.view("ln1_w"), P.view("ln1_b"), 1e-5, want_stats=True)
        ops.gemm_nt(self.x_ln, P.view("fc1_w"), bias=P.view("fc1_b"), act=ACT_RELU, out=self.X[0])

    def fwd_layer(self, i: int):
        """Block i (action_heads.py:337-410): needs hidden_states[i+1] of the LLM and the previous block's output."""
        cfg, P, D, H = self.cfg, self.P, self.D, self.H
        HS, B, S, Kt, Ka, T = self.HSref, self.B, self.S, self.Kt, self.Ka, cfg.chunk
        dh = D // H
        rc, rs_ = self.rope_tab
        hs2 = HS[i + 1].view(B * S, D)
        # adapter tokens: the 64 action-query hidden states + the proprio token (:347); K/V projections, RoPE on K
        ops.gather_rows(hs2, self.row_idx_ka.view(-1), self.h_adp[i].view(B * Ka, D))
        # RoPE (positions restart per segment, action_heads.py:383-388) on q and on every segment's k rides in the projections'
        # epilogues (columns [0, D) of the k|v outputs: the K halves; position = row % segment length); the original block has none
        fuse = self.pro and D % 64 == 0 and not os.environ.get("VLA_NO_KV_ROPE_FUSE")
        wa, ba, _ = self._kv("adp", i)
        ops.gemm_nt(self.h_adp[i].view(B * Ka, D), wa, bias=ba, out=self.KV_adp[i], rope=(2, rc, rs_, Ka, dh, D) if fuse else None)
        # task tokens = HS[i+1][:, :Np] read in place (row-group addressing)
        wt, bt, _ = self._kv("task", i)
        # the reference feeds a STRIDED slice here: torch's CPU Linear then rounds the product before adding the bias, for B > 1
        # only (oracle.linear, vla_native.h bias_post_round) - reproduced so that the head tracks the reference's bf16 run
        # (the task segment's RoPE rides in the 256-row kernel's epilogue since the end of round 3; before, fusing it sent the
        #  8192 x 1792 product to the 128-row kernel: 76 us in situ against 39 + the stand-alone pass)
        ops.gemm_nt(hs2[:B * Kt], wt, bias=bt, out=self.KV_task[i], a_group=(Kt, S * D), bias_post_round=B > 1,
                    rope=(2, rc, rs_, Kt, dh, D) if fuse else None)
        x = self.X[i]
        if self.pro:
            if not fuse:
                ops.rope_inter_(self.KV_adp[i][:, :D], rc, rs_, Ka, H, dh, 0)
                ops.rope_inter_(self.KV_task[i][:, :D], rc, rs_, Kt, H, dh, 0)
            ops.gemm_nt(x, P.view("w_x")[i], bias=P.view("b_x")[i], out=self.QKVx[i], rope=(2, rc, rs_, T, dh, 2 * D))  # q, k_self
        else:
            ops.gemm_nt(x, P.view("w_x")[i], bias=P.view("b_x")[i], out=self.QKVx[i])
        self._attn(i, fwd=True)
        ops.gemm_nt(self.AOx[i], P.view("w_o")[i], bias=P.view("b_o")[i], residual=x, out=self.O2[i])
        self._ln(self.O2[i], P.view("ln_w")[i], P.view("ln_b")[i], self.LNo[i], self.stats[i])
        ops.gemm_nt(self.LNo[i], P.view("w_ffn")[i], bias=P.view("b_ffn")[i], act=ACT_RELU, out=self.X[i + 1])

    def fwd_end(self) -> torch.Tensor:
        P, nb = self.P, self.nb
        self.xf_ln, self.st2 = ops.layernorm_fwd(self.X[nb], P.view("ln2_w"), P.view("ln2_b"), 1e-5, want_stats=True)
        self.pred = ops.gemm_nt(self.xf_ln, P.view("fc2_w"), bias=P.view("fc2_b"))
        return self.pred.view(self.B, self.cfg.chunk, self.cfg.action_dim)

    def _ln(self, x, w, b, y, stats):
        ops.N.check(ops._lib().vla_layernorm_fwd(ops._st(), ops._p(x), ops._p(w), ops._p(b), ops._p(y), ops._p(stats),
                                                 x.shape[0], x.shape[1], x.stride(0), y.stride(0), 1e-5), "layernorm_fwd")

    def _attn(self, i: int, fwd: bool, dout=None):
        import ctypes as C
        B, T, D, Ka, Kt, H = self.B, self.cfg.chunk, self.D, self.Ka, self.Kt, self.H
        qkv = self.QKVx[i].view(B, T, 3 * D)
        ka = self.KV_adp[i].view(B, Ka, 2 * D)
        kt = self.KV_task[i].view(B, Kt, 2 * D)
        args = (qkv[:, :, :D], qkv[:, :, D:2 * D], qkv[:, :, 2 * D:], ka[:, :, :D], ka[:, :, D:], kt[:, :, :D], kt[:, :, D:])
        gate = self.P.view("gate")[i]
        out = self.AOx[i].view(B, T, D)
        if fwd:
            d = ops.head_attn_desc(*args, gate, self.probs[i], out, H)
            d.ref_softmax = int(self.ref_softmax)
            ops.N.check(ops._lib().vla_head_attn_fwd(ops._st(), C.byref(d)), "head_attn_fwd")
        else:
            g = self.dQKVx[i].view(B, T, 3 * D)
            ga = self.dKV_adp[i].view(B, Ka, 2 * D)
            gt = self.dKV_task[i].view(B, Kt, 2 * D)
            ops.head_attn_bwd(dout.view(B, T, D), out, *args, gate, self.probs[i], self.dgate[i:i + 1], g[:, :, :D], g[:, :, D:2 * D],
                              g[:, :, 2 * D:], ga[:, :, :D], ga[:, :, D:], gt[:, :, :D], gt[:, :, D:], H,
                              rope=self.rope_tab if self.pro else None)

    # ---- backward ---------------------------------------------------------------------------------------------
    def backward(self, dpred: torch.Tensor, dHS: torch.Tensor, row0: int = 0):
        """dpred [B, chunk, 7] bf16.  Writes parameter gradients into the flat grad buffer and the hidden-state
        gradients into dHS [nb+1, B, S - row0, D] (HF indexing; rows not touched by the head must be pre-zeroed).
        Sequential composition of bwd_begin / bwd_layer / bwd_end."""
        self.prep_backward(self.pos1, self.Np, self.B, self.S, row0)
        self.bwd_begin(dpred, row0)
        for i in range(self.nb - 1, -1, -1):
            self.bwd_layer(i, dHS)
        self.bwd_end()

    def prep_backward(self, pos1: torch.Tensor, Np: int, B: int, S: int, row0: int, pos0=None, cnt0=None):
        """Scatter indices of the action-row gradients inside the live window [row0, S) of every sequence (-1: dead row; the
        proprio token's row scatters nowhere) and - given the unshifted mask positions - the NaN guard of a frozen window.
        Depends on the batch only, so the step schedule runs it long before the backward."""
        self._alloc(B, Np)
        g = self.guard if pos0 is not None else None
        ops.head_index_prep(pos1, pos0 if pos0 is not None else pos1, cnt0 if cnt0 is not None else pos1, self._idx_scratch,
                            self.row_idx_live_ka, g, B, S, Np, row0)
        self._prep_key = (B, S, Np, row0)

    def bwd_begin(self, dpred: torch.Tensor, row0: int = 0):
        """row0: first live row of the LLM backward (LLM.backward); dHS handed to bwd_layer is [nb+1, B, S - row0, D].
        row0 > 0 also means nothing upstream of the task tokens trains: their dX is dead and skipped."""
        cfg, P, nb = self.cfg, self.P, self.nb
        R, Da = self.R, cfg.action_dim
        self.row0 = row0
        assert getattr(self, "_prep_key", None) == (self.B, self.S, self.Np, row0), "prep_backward() first"
        self.refresh_transposes()                                  # W^T operands of the dX products (stale after AdamW)
        ops.zero_(self.acc32)                                      # every fp32 gradient accumulator in one fill
        dp = dpred.reshape(R, Da)
        ops.zero_(self.dpad)
        ops.copy2d(dp, self.dpad, R, Da, Da, 64)
        ops.colsum_(dp, self.b_f32["fc2_b"])
        self._dw(self.dpad, self.xf_ln, out=self.dfc2)                         # action_dim 7 rides zero-padded to 64 columns
        ops.copy2d(self.dfc2, P.g("fc2_w"), Da, self.D, self.D, self.D)
        d_ln2 = ops.gemm_nt(self.dpad, self.fc2T)                              # [R, D]
        self.dx = ops.layernorm_bwd(d_ln2, self.X[nb], P.view("ln2_w"), self.st2, self.ln2_dw, self.ln2_db)

    def bwd_layer(self, i: int, dHS: torch.Tensor):
        """Backward of block i: the x-chain (critical path), then this layer's hidden-state gradients
        dHS[i+1] (task rows written in place, action rows scattered) - all the LLM backward of layer i waits for."""
        P, D = self.P, self.D
        B, S, Kt, Ka = self.B, self.S, self.Kt, self.Ka
        dff = self.dFF[i]
        ops.N.check(ops._lib().vla_relu_bwd(ops._st(), ops._p(self.dx), ops._p(self.X[i + 1]), ops._p(dff), self.dx.numel()), "relu_bwd")
        d_ln = ops.gemm_nt(dff, self.T["w_ffn"][i])
        do2 = self.dO2[i]
        self._ln_bwd(d_ln, self.O2[i], P.view("ln_w")[i], self.stats[i], do2, self.ln_dw[i], self.ln_db[i])
        d_ao = ops.gemm_nt(do2, self.T["w_o"][i])
        self._attn(i, fwd=False, dout=d_ao)          # returns dq / dk already through the RoPE transpose
        self.dx = ops.gemm_nt(self.dQKVx[i], self.T["w_x"][i], residual=do2)
        # d h_adapter -> action rows of dHS[i+1] (+ the proprio token's gradient); d h_task -> dHS[i+1][:, :Np] in place
        ops.gemm_nt(self.dKV_adp[i], self._kv("adp", i)[2], out=self.dh_adp[i])
        ops.scatter_add_rows(self.dh_adp[i], self.row_idx_live_ka.view(-1), dHS[i + 1].view(-1, D))
        if self.row0 == 0:
            ops.gemm_nt(self.dKV_task[i], self._kv("task", i)[2], out=dHS[i + 1].view(B * S, D)[:B * Kt], c_group=(Kt, S * D))

    def bwd_end(self):
        """Off the critical path: input stage, proprio projector, and every dW as batched NT GEMMs on transposed operands."""
        cfg, P, D, nb = self.cfg, self.P, self.D, self.nb
        B, Ka = self.B, self.Ka
        G = P.g
        # input stage: relu -> fc1 -> layer_norm1 (input is noise/zeros: only parameter gradients)
        dy1 = ops.relu_bwd(self.dx, self.X[0])
        ops.colsum_(dy1, self.b_f32["fc1_b"])
        self._dw(dy1, self.x_ln, out=G("fc1_w"))
        d_xln = ops.gemm_nt(dy1, self._t(P.view("fc1_w")))
        ops.layernorm_bwd(d_xln, self.x_in, P.view("ln1_w"), self.st1, self.ln1_dw, self.ln1_db, want_dx=False)
        # proprio projector backward (its token's gradient = sum over the blocks)
        ops.N.check(ops._lib().vla_colsum_bf16(ops._st(), ops._p(self.dh_adp[0, NUM_TOKENS]), ops._p(self.d_pf32), nb, D, B * Ka * D, B,
                                                Ka * D, D), "colsum(d_pf)")             # [nb, B, D] strided -> sum over the blocks
        d_pf = ops.cast_f32_bf16(self.d_pf32)
        ops.colsum_(d_pf, self.b_f32["p_fc2_b"])
        self._dw(d_pf, self.pp_act, out=G("p_fc2_w"))
        d_act = ops.gemm_nt(d_pf, self.T["p_fc2_w"])
        d_pre = ops.gelu_bwd(d_act, self.pp_pre)
        ops.colsum_(d_pre, self.b_f32["p_fc1_b"])
        # proprio_dim is 8 (LIBERO), 7 (BRIDGE) or 14 (ALOHA) - prismatic/vla/constants.py:38-52: the TN product wants 8-element
        # column chunks, so it contracts against the zero-padded 64-column input (as the forward does) and the first proprio_dim
        # columns are copied out (the fc2_w gradient above takes the same route for action_dim 7)
        self._dw(d_pre, self.pr_in, out=self.dpfc1)
        ops.copy2d(self.dpfc1, G("p_fc1_w"), D, cfg.proprio_dim, 64, cfg.proprio_dim)
        # batched dW products over the nb blocks: dW = dY^T . X as TN GEMMs on dY and X as they lie in memory (the contraction
        # runs over their rows); the task tokens are read in place from the hidden states (row groups: Kt rows of every sequence)
        Kt, S = self.Kt, self.S
        h_adp = self.h_adp.view(nb, B * Ka, D)
        if self.h_task_c is None:
            h_task, tg = self.HSref[1:nb + 1, 0, :Kt], (Kt, S * D)
        else:
            for i in range(nb):
                ops.copy_rows3d(self.HSref[i + 1], self.h_task_c[i], B, Kt, D, S * D, D, Kt * D, D)
            h_task, tg = self.h_task_c, None
        ops.gemm_tn(self.dQKVx, self.X[:nb], out=G("w_x"))
        ops.gemm_tn(self.dO2, self.AOx, out=G("w_o"))
        ops.gemm_tn(self.dFF, self.LNo, out=G("w_ffn"))
        ops.colsum_(self.dQKVx, self.b_f32["b_x"]); ops.colsum_(self.dO2, self.b_f32["b_o"]); ops.colsum_(self.dFF, self.b_f32["b_ffn"])
        if self.pro:
            ops.gemm_tn(self.dKV_adp, h_adp, out=G("w_adp"))
            ops.gemm_tn(self.dKV_task, h_task, out=G("w_task"), rows=B * Kt, b_group=tg)
            ops.colsum_(self.dKV_adp, self.b_f32["b_adp"]); ops.colsum_(self.dKV_task, self.b_f32["b_task"])
        else:             # shared k_proj / v_proj: the three segments' gradients add up (autograd accumulates them in bf16 too)
            gkv = G("w_x")[:, self.D:]
            ops.gemm_tn(self.dKV_adp, h_adp, out=gkv, accumulate=True)
            ops.gemm_tn(self.dKV_task, h_task, out=gkv, accumulate=True, rows=B * Kt, b_group=tg)
            bkv = self.b_f32["b_x"][:, self.D:]
            ops.colsum_(self.dKV_adp, bkv); ops.colsum_(self.dKV_task, bkv)
        for k, t in self.b_f32.items():
            ops.cast_f32_bf16(t, out=G(k))
        ops.cast_f32_bf16(self.ln_dw, out=G("ln_w")); ops.cast_f32_bf16(self.ln_db, out=G("ln_b"))
        ops.cast_f32_bf16(self.ln1_dw, out=G("ln1_w")); ops.cast_f32_bf16(self.ln1_db, out=G("ln1_b"))
        ops.cast_f32_bf16(self.ln2_dw, out=G("ln2_w")); ops.cast_f32_bf16(self.ln2_db, out=G("ln2_b"))
        ops.copy2d(self.dgate, G("gate"), nb, 1, 1, 8)

    def _t(self, w2d):
        return ops.transpose(w2d.contiguous())

    def _dw(self, dy, x, out=None):
        """dW[N, K] = dY[R, N]^T . X[R, K] for the small one-off layers."""
        return ops.gemm_tn(dy, x, out=out, split=0)

    def _ln_bwd(self, dy, x, w, stats, dx, dw, db):
        ops.N.check(ops._lib().vla_layernorm_bwd(ops._st(), ops._p(dy), ops._p(x), ops._p(w), ops._p(stats), ops._p(dx), ops._p(dw),
                                                 ops._p(db), x.shape[0], x.shape[1], x.stride(0), dy.stride(0), dx.stride(0)), "layernorm_bwd")


# ------------------------------------------------------------------------------------------------ whole model
class VLAEngine:
    """Adapter-only fine-tune step of VLA-Adapter (finetune.py:288-447 + 1039-1082) on one GPU."""

    def __init__(self, cfg: VLACfg, weights: Dict, device="cuda"):
        """weights = dict(vit=[sd...], proj=sd, llm=sd (HF names without 'model.' prefix incl. embed_tokens/norm),
        head=sd, proprio=sd, action_queries=tensor)."""
        self.cfg, self.device = cfg, device
        self.vits = [ViT(c, sd, device) for c, sd in zip(cfg.vit, weights["vit"])]
        self.llm = LLM(cfg.llm, weights["llm"], device)
        self.head = Head(cfg, device)
        self.head.load_state_dicts(weights["head"], weights["proprio"], weights.get("action_queries"))
        g = lambda k: weights["proj"][k].to(device=device, dtype=BF16).contiguous()
        self.proj = {k: g(k) for k in weights["proj"]}
        self.step_count = 0
        self._dHS = None
        # adapter-only fine-tune: the LLM backward only has to cover the rows that can reach `action_queries`
        # (LLM.backward).  VLA_FULL_LLM_BWD=1 / full_llm_backward=True runs the reference-shaped full-sequence backward.
        self.full_llm_backward = bool(int(os.environ.get("VLA_FULL_LLM_BWD", "0")))
        self._row0 = None          # frozen by capture(); None = derive from every batch (one host sync)
        self.reducer = None        # ddp.FlatGradReducer when world_size > 1
        self.ga, self._micro, self._gacc = 1, 0, None     # gradient accumulation (set_grad_accumulation)
        self.executed_steps = 0    # forward+backward passes enqueued so far (eager, pipelined or replayed): profile bookkeeping
        # LLM layers above the head's last block (Qwen2.5-1.5B: 28 layers, 24 blocks - action_heads.py:117-118 reads hidden_states[1..24])
        # never reach the loss or the predicted actions: the training step and predict() run the first n_act layers only, as the
        # LoRA / full trainers do; forward_vlm() - the API twin that RETURNS every hidden state - runs them all
        self.n_act = min(cfg.llm.n_layers, cfg.num_blocks)
        self.fp8_frozen = False
        if os.environ.get("VLA_FP8_FROZEN"):
            self.enable_fp8_frozen()

    def enable_fp8_frozen(self):
        """Opt-in, NOT the reference's arithmetic (it is bf16 throughout; BASELINE configs[4] names an fp8 weight path, the
        reference has no code for it): run the frozen backbones' LayerNorm/RMSNorm-fed projections (ViT qkv / fc1, LLM q|k|v /
        gate|up) on e4m3 weights and row-quantised inputs.  Adapter-only training and inference; the backward is unchanged."""
        for v in self.vits:
            v.enable_fp8()
        self.llm.enable_fp8()
        self.fp8_frozen = True

    def set_grad_accumulation(self, n: int):
        """finetune.py:1039-1042, 1078-1082: loss / n on every micro-batch, gradients summed over n micro-batches (in bf16, as
        autograd accumulates ``.grad``), one optimizer step per n.  The data-parallel exchange runs on the boundary
        micro-step only (the reference's DDP all-reduces on every one; same result, n-1 exchanges saved).  Call before
        capture(): the captured loss kernel carries the 1/n."""
        assert n >= 1 and getattr(self, "_graphs", None) is None, "set_grad_accumulation() before capture()"
        self.ga, self._micro = int(n), 0
        self._gacc = torch.zeros_like(self.head.P.grad) if n > 1 else None

    def _accumulate(self) -> bool:
        """Fold the micro-step's gradient into the accumulator; True on the boundary micro-step (P.grad then holds the sum)."""
        if self.ga == 1:
            return True
        G = self.head.P.grad
        if self._micro == 0:
            self._gacc.copy_(G)
        else:
            ops.add_(self._gacc, G)
        self._micro += 1
        if self._micro < self.ga:
            return False
        self._micro = 0
        G.copy_(self._gacc)
        return True

    def forward(self, batch: Dict[str, torch.Tensor], noise: Optional[torch.Tensor] = None, for_training: bool = False):
        """VLM forward + action head (finetune.py:336-411) -> predicted actions [B, chunk, 7]."""
        self.forward_vlm(batch, for_training)
        return self.head.forward(self.llm.HS, self.pos1, batch["proprio"], self.Np, noise)

    # ---- batch-1 inference (modeling_prismatic.py:892-972): forward only, captured per input shape ------------------
    def predict(self, batch: Dict[str, torch.Tensor]) -> torch.Tensor:
        """Forward pass in phase "Inference" (no input perturbation) -> normalised actions [B, chunk, action_dim] bf16.
        At batch 1 every kernel is a handful of workgroups and the ~1000 launches form dependent chains, so the forward is
        cut into single-stream segments like the training step: one stream per vision backbone (they are independent),
        the LLM on the caller's stream with the action head trailing it on the head stream; each segment is a linear
        hipGraph captured on first use of an input shape and replayed on static input buffers afterwards."""
        key = (tuple(batch["input_ids"].shape), tuple(batch["pixel_values"].shape), batch["pixel_values"].dtype)
        cache = self.__dict__.setdefault("_predict_graphs", {})
        if os.environ.get("VLA_PREDICT_EAGER"):
            return self.forward(batch, None)
        self._ensure_streams()
        if key not in cache:
            static = {k: v.clone() for k, v in batch.items()}
            for _ in range(2):                       # allocate buffers / set kernel attributes outside the capture
                self.forward(static, None)
            torch.cuda.synchronize()
            segs = self._predict_segments(static)
            with ops.latency_hint():                 # sub-chip launches on an idle chip: the deep-ring GEMM (bit-identical; baked into the graphs)
                graphs = self._capture_segments(segs, {})
            torch.cuda.synchronize()
            cache[key] = (graphs, static, segs)
        graphs, static, segs = cache[key]
        for k, v in batch.items():
            static[k].copy_(v)
        self.head.refresh_forward_operands()     # parameters may have changed since the capture (no-op when fresh)
        ev = self._run_segments(segs, graphs, getattr(self, "_timeline", None))
        torch.cuda.current_stream().wait_event(ev[("end", 0)])
        return self._pred_out

    def _predict_segments(self, batch):
        cfg, llm, head = self.cfg, self.llm, self.head
        n, nb = cfg.llm.n_layers, cfg.num_blocks
        self._vision_begin(batch)                                 # host-side bookkeeping only
        n_all, n = n, self.n_act                                  # (layers above the head's last block do not reach the actions)
        segs = [(f"V{j}", (lambda j=j: self._vision_backbone(j, batch)), None, ("v", j)) for j in range(len(self.vits))]
        ch = self._chunks(n, [6] * max(0, (n - 6) // 6) + [4, 2]) if n >= 12 else self._chunks(n, [1])   # few launches: the caller blocks on every call

        def m_fwd(c, lo, hi):
            def fn():
                if c == 0:
                    self._vision_project()
                    llm.fwd_begin(self.B, self.S, self._embed(batch), 0)
                for i in range(lo, hi):
                    llm.fwd_layer(i)
                if hi == n_all:
                    llm.fwd_final()
            return fn

        def h_fwd(c, lo, hi, last):
            def fn():
                if c == 0:
                    head.fwd_begin(llm.HS, self.pos1, batch["proprio"], self.Np, None)
                for i in range(lo, min(hi, nb)):
                    head.fwd_layer(i)
                if last:
                    self._pred_out = head.fwd_end()
            return fn

        for c, (lo, hi) in enumerate(ch):
            segs.append(("M", m_fwd(c, lo, hi), [("v", j) for j in range(len(self.vits))] if c == 0 else None, ("f", c)))
            segs.append(("H", h_fwd(c, lo, hi, c == len(ch) - 1), ("f", c), ("end", 0) if c == len(ch) - 1 else None))
        return segs

    # modeling_prismatic.py:596-655 (multimodal forward): fills llm.HS with the n+1 hidden states
    def forward_vlm(self, batch: Dict[str, torch.Tensor], for_training: bool = False, action_queries: bool = True):
        """for_training: a loss_and_backward() follows - backward-only tensors are kept for its live rows only
        (reads the mask positions back: one host sync, eager path).  action_queries=False: the plain VLM forward of
        prismatic/models/vlms/prismatic.py:312-481 (embedding gather + patch splice only; ``labels`` not needed)."""
        self._vision(batch)
        mm = self._embed(batch, action_queries)
        self.llm.forward(self.B, self.S, mm, self.live_row0() if (for_training and action_queries) else 0,
                         n_run=self.n_act if for_training else None)

    def token_ce(self, labels: torch.Tensor):
        """HF shifted token cross-entropy of the last forward_vlm (prismatic/models/vlms/prismatic.py:469-481):
        logits = lm_head(hidden_states[-1]) over ALL positions (bf16, returned like the reference does), multimodal labels =
        [labels[:, :1] | IGNORE for the patches | labels[:, 1:]] (:411-422), loss = mean over the positions whose NEXT label is
        valid of logsumexp(float(logits)) - logits[label].  lm_head = the checkpoint's ``lm_head.weight`` or, when tied
        (Qwen2.5-0.5B: tie_word_embeddings), the embedding table.  Returns (loss f32 scalar tensor, logits [B, S, V])."""
        llm, B, S, Np = self.llm, self.B, self.S, self.Np
        n, D = self.cfg.llm.n_layers, self.cfg.llm.d
        W = getattr(llm, "lm_head", None)
        W = llm.embed if W is None else W
        V = W.shape[0]
        assert V % 8 == 0 and tuple(labels.shape) == (B, S - Np)
        logits = ops.gemm_nt(llm.HS[n].view(B * S, D), W, split_k=0).view(B, S, V)
        # target of sequence row s = multimodal label of row s + 1; the last row has none
        tgt = torch.full((B, S), IGNORE_INDEX, device=self.device, dtype=torch.int64)
        tgt[:, Np:S - 1] = labels[:, 1:]                  # rows Np .. S-2 predict text tokens 1 .. L-1 (row 0 predicts a patch: ignored)
        out2 = torch.zeros(2, device=self.device, dtype=torch.float32)
        ops.N.check(ops._lib().vla_token_ce(ops._st(), ops._p(logits), V, ops._p(tgt), B * S, V, ops._p(out2)), "token_ce")
        return out2[0] / out2[1], logits

    def _vision_and_embed(self, batch: Dict[str, torch.Tensor]) -> torch.Tensor:
        """ViT(s) -> projector -> action masks -> embedding/query splice into llm.HS[0]; returns the key mask [B,S] u8."""
        self._vision(batch)
        return self._embed(batch)

    def _vision(self, batch: Dict[str, torch.Tensor]):
        """Frozen part that reads NO trainable tensor: ViT(s) + projector -> self.patches [B, Np, D] (its own buffer, so
        that the vision stage of the NEXT step can run while the current step is still using llm.HS)."""
        self._vision_begin(batch)
        for j in range(len(self.vits)):
            self._vision_backbone(j, batch)
        self._vision_project()

    def _vision_begin(self, batch: Dict[str, torch.Tensor]):
        cfg = self.cfg
        B, L = batch["input_ids"].shape
        Np = cfg.n_patches
        self.llm._alloc(B, L + Np)
        bufs = self.__dict__.setdefault("_vis_bufs", {})     # one pair per batch size: captured graphs keep their addresses
        if (B, Np) not in bufs:
            bufs[(B, Np)] = (torch.empty(B, Np, cfg.vis_dim, device=self.device, dtype=BF16),
                             torch.empty(B, Np, cfg.llm.d, device=self.device, dtype=BF16))
        self.feats, self.patches = bufs[(B, Np)]
        self.B, self.S, self.Np = B, L + Np, Np

    def _vision_backbone(self, j: int, batch: Dict[str, torch.Tensor]):
        """Backbone j over ALL images of the batch in one pass (modeling_prismatic.py:196-237 runs them one by one): image
        `im` uses channels 3*(im*n_backbones + j) .. +2 and fills feats[:, im*npi:(im+1)*npi, column block of backbone j]."""
        cfg, vit, px = self.cfg, self.vits[j], batch["pixel_values"]
        B, Np, nbk, npi = self.B, self.Np, len(cfg.vit), cfg.vit[0].n_patches
        col = sum(v.cfg.d for v in self.vits[:j])
        if nbk == 1 and cfg.n_img == 1:
            vit.forward(px, 0, self.feats.view(B * Np, -1))
            return
        if cfg.n_img == 1:
            stacked, c0 = px, 3 * j
        else:          # [n_img * B, 3, H, W]: the images of one backbone stacked along the batch
            stacked = torch.cat([px[:, 3 * (im * nbk + j):3 * (im * nbk + j) + 3] for im in range(cfg.n_img)], 0).contiguous()
            c0 = 0
        tmp = torch.empty(cfg.n_img * B * npi, vit.cfg.d, device=self.device, dtype=BF16)
        vit.forward(stacked, c0, tmp)
        t4 = tmp.view(cfg.n_img, B, npi, vit.cfg.d)
        for im in range(cfg.n_img):
            self.feats[:, im * npi:(im + 1) * npi, col:col + vit.cfg.d] = t4[im]

    def _vision_project(self):
        """PrismaticProjector (modeling_prismatic.py:261-273) -> self.patches."""
        cfg, B, Np = self.cfg, self.B, self.Np
        f2 = self.feats.view(B * Np, -1)
        h = ops.gemm_nt(f2, self.proj["fc1.weight"], bias=self.proj["fc1.bias"], act=ACT_GELU)
        dst = self.patches.view(B * Np, cfg.llm.d)
        if cfg.fused:
            h = ops.gemm_nt(h, self.proj["fc2.weight"], bias=self.proj["fc2.bias"], act=ACT_GELU)
            ops.gemm_nt(h, self.proj["fc3.weight"], bias=self.proj["fc3.bias"], out=dst)
        else:
            ops.gemm_nt(h, self.proj["fc2.weight"], bias=self.proj["fc2.bias"], out=dst)

    def _embed(self, batch: Dict[str, torch.Tensor], action_queries: bool = True) -> torch.Tensor:
        """Action masks + embedding gather + action-query splice (train_utils.py:8-41; modeling_prismatic.py:601-636)
        into rows 0 and Np+1.. of llm.HS[0] (reads the trainable `action_queries`); returns the key mask [B,S] u8."""
        cfg, llm = self.cfg, self.llm
        ids, labels, am = batch["input_ids"], batch.get("labels"), batch["attention_mask"]
        B, L = ids.shape
        Np = cfg.n_patches
        S = L + Np
        llm._alloc(B, S)
        X0 = llm.HS[0]
        ops.copy2d(self.patches, X0[0, 1], B * Np, cfg.llm.d, cfg.llm.d, cfg.llm.d, d_group=(Np, S * cfg.llm.d))   # patches: rows 1..Np
        if action_queries:
            self.qidx0, self.pos0, self.cnt0 = ops.action_mask(labels, 0)
            _, self.pos1, self.cnt1 = ops.action_mask(labels, 1)
        else:                                                     # plain VLM forward: no slot is overwritten
            self.qidx0 = torch.full((B, L), -1, device=self.device, dtype=torch.int32)
        mm = torch.empty(B, S, device=self.device, dtype=torch.uint8)
        am8 = am.view(torch.uint8) if am.dtype == torch.bool and am.is_contiguous() else am.to(torch.uint8).contiguous()
        ops.embed_splice(ids, am8, self.qidx0, llm.embed, self.head.P.view("action_queries"), X0, mm, Np)
        self.B, self.S, self.Np = B, S, Np
        return mm

    def live_row0(self) -> int:
        """First sequence row the backward has to cover: the largest multiple of 32 not above the first action-query
        position of any sample of the current batch (0 = whole sequence).  Reads the mask positions back (host sync)."""
        if self.full_llm_backward:
            return 0
        first = self.pos0[:, 0]
        if bool((self.cnt0 < 1).any()):
            return 0
        return (self.Np + int(first.min())) // 32 * 32

    def _to_bf16(self, t: torch.Tensor) -> torch.Tensor:
        """bf16 copy of a small contiguous fp32 / bf16 tensor (targets) by the native cast-copy."""
        if t.dtype == BF16:
            return t
        t = t.contiguous()
        out = torch.empty(t.shape, device=t.device, dtype=BF16)
        n = t.shape[-1]
        return ops.copy2d(t, out, t.numel() // n, n, n, n)

    def _dhs(self, row0: int) -> torch.Tensor:
        B, S, D, n = self.B, self.S, self.cfg.llm.d, self.cfg.llm.n_layers
        if self._dHS is None or tuple(self._dHS.shape[1:3]) != (B, S - row0):
            self._dHS = torch.empty(n + 1, B, S - row0, D, device=self.device, dtype=BF16)
        ops.zero_(self._dHS)
        return self._dHS

    def loss_and_backward(self, pred, actions, gscale: float = 1.0, exchange: bool = True):
        """L1 loss (finetune.py:418) + backward into the flat grad buffer.  gscale scales the gradient (loss / grad-accumulation
        steps); exchange=False leaves the data-parallel exchange to the caller (non-boundary micro-steps)."""
        llm, head = self.llm, self.head
        B, S, Np = self.B, self.S, self.Np
        loss3, dpred = ops.l1_loss(pred, self._to_bf16(actions), True, gscale)
        row0 = self.live_row0()
        dHS = self._dhs(row0)
        head.backward(dpred, dHS, row0)
        aq_off = head.P.offsets["action_queries"][0]
        if self.reducer is not None and exchange:       # head/proprio grads are final: exchange them under the LLM backward
            self.reducer.reduce_async(head.P.grad, 0, aq_off)
        dX0 = llm.backward(dHS, B, S, row0, n_run=self.n_act)
        dq = ops.action_query_grad(dX0.contiguous(), self.pos0, Np, row0)
        ops.cast_f32_bf16(dq, out=head.P.g("action_queries"))
        if self.reducer is not None and exchange:
            self.reducer.reduce_async(head.P.grad, aq_off, None)
        return loss3

    def optimizer_step(self, lr: float, beta1=0.9, beta2=0.999, eps=1e-8, wd=0.01):
        """torch.optim.AdamW semantics on the flat trainable buffer (finetune.py:910, 1078-1082)."""
        self.step_count += 1
        P = self.head.P
        gscale = 1.0
        if self.reducer is not None:
            self.reducer.wait()
            gscale = self.reducer.grad_scale       # DDP averages: sum-all-reduce then 1/N, folded into AdamW
        ops.adamw_(P.data, P.grad, P.m, P.v, self.step_count, lr, beta1, beta2, eps, wd, gscale=gscale)
        self.head.dirty = True

    def train_step(self, batch, lr: float, noise=None):
        """One micro-batch; the optimizer steps on every ``ga``-th call (set_grad_accumulation)."""
        self.executed_steps += 1
        pred = self.forward(batch, noise, for_training=True)
        loss3 = self.loss_and_backward(pred, batch["actions"], 1.0 / self.ga, exchange=self.ga == 1)
        if self._accumulate():
            if self.ga > 1 and self.reducer is not None:
                self.reducer.reduce_async(self.head.P.grad, 0, None)
            self.optimizer_step(lr)
        return loss3

    # ---- pipelined two-stream schedule + hipGraph replay -----------------------------------------------------------
    # The action head is 3 % of the FLOPs but a chain of ~300 small dependent kernels (~9 ms when run alone): block i
    # only needs hidden_states[i+1], so its forward trails the LLM forward on a SECOND stream and its backward runs AHEAD
    # of the LLM backward (which needs dHS[i+1] from block i).  The head's kernels fill the idle CUs / tile-quantisation
    # tails of the LLM's large GEMMs instead of serialising with them.
    #
    # The step is cut into SEGMENTS, each living on exactly one stream ("M": vision/LLM, "H": head); segments are
    # ordered so that every event is recorded before it is waited on.  Eagerly a segment is a Python call under its
    # stream; captured, every segment is its own single-stream (linear) hipGraph and the cross-stream edges are plain
    # hipEvents between graph launches.  (One multi-stream hipGraph of the whole step was measured to serialise the two
    # backward chains in the runtime's graph executor - rocprofv3 trace, tools/timeline.py: LLM backward started only
    # after the head backward's last kernel - so the overlap is not left to it.)  Layer chunks are short next to the
    # forward->backward turn-around (little pipeline fill/drain) and longer elsewhere (fewer graph launches).
    def _ensure_streams(self):
        if getattr(self, "side", None) is None:
            self.side = torch.cuda.Stream()            # head stream (a high-priority stream measured 0.7 % slower on the step in round 1)
            self._cap_main = torch.cuda.Stream()       # capture stream of the "M" graphs (replayed on the current stream)
            self.vis_stream = torch.cuda.Stream()      # vision stage of the NEXT step (fills the backward's idle CUs)
            self._vstreams = [self.vis_stream] + [torch.cuda.Stream() for _ in range(max(0, len(self.vits) - 1))]   # one per backbone

    @staticmethod
    def _chunks(n: int, sizes) -> List[Tuple[int, int]]:
        """[lo, hi) layer ranges covering 0..n with the given chunk sizes (last size repeats / is clipped)."""
        out, lo, k = [], 0, 0
        while lo < n:
            sz = sizes[min(k, len(sizes) - 1)]
            out.append((lo, min(n, lo + sz)))
            lo, k = out[-1][1], k + 1
        return out

    def _prep_backward(self, batch):
        """Everything the backward needs that depends only on the batch (runs at the start of the step, off the
        forward->backward turn-around): live-row window, zeroed dHS, guard, scatter indices, bf16 targets."""
        row0 = self._row0 if self._row0 is not None else self.live_row0()
        self._row0_used = row0
        self._dhs(row0)
        self.head.prep_backward(self.pos1, self.Np, self.B, self.S, row0, self.pos0, self.cnt0)
        self._guard = self.head.guard if row0 else None          # NaN when a sample's action block starts before the frozen window
        self._actions_bf = self._to_bf16(batch["actions"])

    def _segments(self, batch, noise):
        """[(stream 'M'|'H', fn, wait_key|None, signal_key|None)] for everything after the vision stage."""
        cfg, llm, head = self.cfg, self.llm, self.head
        n_all, nb = cfg.llm.n_layers, cfg.num_blocks
        assert nb <= n_all
        n = self.n_act                                  # layers that reach the loss
        # long chunks at the bottom layers, single layers at the top: the head's last forward chunk and first backward
        # chunk (the serial turn-around) stay short; the backward walks the same ranges top-down
        fch = self._chunks(n, [4] * max(0, (n - 4) // 4) + [2, 1, 1]) if n >= 8 else self._chunks(n, [1])
        if os.environ.get("VLA_FWD_CHUNKS"):                  # A/B knob: "4,4,4,4,4,2,1,1"
            fch = self._chunks(n, [int(x) for x in os.environ["VLA_FWD_CHUNKS"].split(",")])
        segs = []

        def m_begin():
            mm = self._embed(batch)
            self._prep_backward(batch)
            llm.fwd_begin(self.B, self.S, mm, self._row0_used)

        def m_fwd(c, lo, hi, begin=False):
            def fn():
                if begin:
                    m_begin()
                for i in range(lo, hi):
                    llm.fwd_layer(i)
                if hi == n_all:
                    llm.fwd_final()
            return fn

        def h_fwd(c, lo, hi, last):
            def fn():
                if c == 0:
                    head.fwd_begin(llm.HS, self.pos1, batch["proprio"], self.Np, noise)
                    head.refresh_transposes()        # (the W^T operands of the head's backward: rebuilt here, beside the LLM forward, instead of
                                                     #  at bwd_begin - 0.13 ms on the forward -> backward turn-around that the LLM backward waits for: step -0.09 ms same box)
                for i in range(lo, min(hi, nb)):
                    head.fwd_layer(i)
                if last:
                    pred = head.fwd_end()
                    self._loss3, dpred = ops.l1_loss(pred, self._actions_bf, True, 1.0 / self.ga)
                    if self._guard is not None:
                        ops.add_scalar_f32_(self._loss3, self._guard)
                    head.bwd_begin(dpred, self._row0_used)
            return fn

        def h_bwd(lo, hi):
            def fn():
                for i in range(min(hi, nb) - 1, lo - 1, -1):
                    head.bwd_layer(i, self._dHS)
            return fn

        def m_bwd(lo, hi, first, last):
            def fn():
                if first:
                    llm.bwd_begin(self._dHS, self._row0_used, n)
                for i in range(hi - 1, lo - 1, -1):
                    llm.bwd_layer(i, self._dHS)
                if last:
                    dq = ops.action_query_grad(llm.bwd_result().contiguous(), self.pos0, self.Np, self._row0_used)
                    ops.cast_f32_bf16(dq, out=head.P.g("action_queries"))
            return fn

        # ONE whole-batch forward pipeline.  (Round 1 ran the forward as two half-batch pipelines on two streams - with the 128-row
        # GEMM, two workgroups per CU, the chains filled each other's tail rounds: -0.5 ms.  With the persistent 256 x 256 kernel,
        # one workgroup per CU and 6.5 rounds of gate/up tiles per whole-batch launch, one pipeline measured 25.27-25.34 against
        # 25.61-25.75 ms and the two-pipeline form left the tree in round 4: DESIGN section 5b.)
        for c, (lo, hi) in enumerate(fch):
            segs.append(("M", m_fwd(c, lo, hi, begin=c == 0), None, ("f", c)))
            segs.append(("H", h_fwd(c, lo, hi, c == len(fch) - 1), ("f", c), None))
        for k, (lo, hi) in enumerate(reversed(fch)):
            segs.append(("H", h_bwd(lo, hi), None, ("b", k)))
            segs.append(("M", m_bwd(lo, hi, k == 0, k == len(fch) - 1), ("b", k), None))
        segs.append(("H", head.bwd_end, None, ("end", 0)))      # the caller joins on this event (head gradients final)
        return segs

    def _stream_of(self, name: str, main):
        return main if name == "M" else self.side if name == "H" else self._vstreams[int(name[1:])]

    def _run_segments(self, segs, graphs=None, timeline=None, hooks=None):
        """Enqueue the segments [(stream 'M'|'H'|'V<j>', fn|None, wait key | [keys] | None, signal key | None)] in order.
        timeline: optional list that receives (stream, index, start_event, end_event) per segment (timing events).
        hooks: {segment index: fn(event)} called right after that segment was enqueued, with an event recorded behind it.
        Returns the dict of recorded events."""
        main = torch.cuda.current_stream()
        for name in {sg[0] for sg in segs} - {"M"}:
            self._stream_of(name, main).wait_stream(main)        # fork (inputs / previous AdamW are ordered before them)
        ev = {}
        for k, (st, fn, wait, signal) in enumerate(segs):
            stream = self._stream_of(st, main)
            with torch.cuda.stream(stream):
                for w in ([] if wait is None else wait if isinstance(wait, list) else [wait]):
                    stream.wait_event(ev[w])
                if timeline is not None:
                    t0 = torch.cuda.Event(enable_timing=True)
                    t0.record(stream)
                if fn is None:
                    pass
                elif graphs is None:
                    fn()
                else:
                    graphs[k].replay()
                if timeline is not None:
                    t1 = torch.cuda.Event(enable_timing=True)
                    t1.record(stream)
                    timeline.append((st, k, t0, t1))
                if signal is not None:
                    ev[signal] = torch.cuda.Event()
                    ev[signal].record(stream)
                if hooks is not None and k in hooks:
                    hev = ev[signal] if signal is not None else torch.cuda.Event()
                    if signal is None:
                        hev.record(stream)
                    hooks[k](hev)
        return ev

    def _capture_segments(self, segs, pools):
        """One linear hipGraph per segment, captured on a stream of the segment's kind with that kind's memory pool (graphs
        sharing a pool replay strictly in capture order on ONE stream, so the allocator's reuse of freed capture-time
        temporaries stays race-free while the streams overlap)."""
        graphs = []
        for st, fn, _, _ in segs:
            if fn is None:
                graphs.append(None)
                continue
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=pools.setdefault(st, torch.cuda.graph_pool_handle()),
                                  stream=self._cap_main if st == "M" else self._stream_of(st, None), capture_error_mode="thread_local"):
                fn()
            graphs.append(g)
        return graphs

    def _fwd_bwd(self, batch, noise, vision: bool = True):
        """Eager run of the two-stream schedule (vision stage first unless the vision graph already ran)."""
        self.executed_steps += 1
        if vision:
            self._vision(batch)
        torch.cuda.current_stream().wait_event(self._run_segments(self._segments(batch, noise))[("end", 0)])   # join
        return self._loss3

    # Data-parallel schedule of the captured step.  The gradient exchange of step k (one bucketed RCCL all-reduce of the
    # flat gradient buffer on its own stream) is NOT waited for at the end of step k: step k+1 first replays the vision
    # graph - ViT + projector, ~25 % of the step, which read no trainable tensor - and only then waits for the exchange,
    # applies AdamW and replays the rest.  Same arithmetic as synchronous DP (every use of a parameter sees the updated
    # value); the all-reduce is hidden under the next step's ViT instead of under a backward tail that the live-row
    # LLM backward has made too short to hide it.  `flush()` applies the last pending update.
    def capture(self, batch: Dict[str, torch.Tensor], noise: Optional[torch.Tensor] = None, warmup: int = 2,
                conservative_rows: bool = False):
        """``batch``/``noise`` become the static input buffers: copy new data INTO them before each replay.
        The live-row window of the LLM backward is frozen here: from the capture batch's first action-query row, or - with
        ``conservative_rows`` - from the first text row (valid for ANY later batch of the same shape: the action block
        cannot start before tok0 + patches + one prompt token)."""
        self._ensure_streams()
        self._static_batch, self._static_noise = batch, noise
        self._row0 = None
        self._vision_and_embed(batch)                # masks of the capture batch -> the frozen live-row window
        self._row0 = self.live_row0()
        if conservative_rows and not self.full_llm_backward:
            self._row0 = min(self._row0, (self.cfg.n_patches + 1) // 32 * 32)
        for _ in range(warmup):                      # allocate every buffer / set kernel attributes outside capture
            self.head.dirty = True
            self._fwd_bwd(batch, noise)
        torch.cuda.synchronize()
        self.head.dirty = True
        # one memory pool per stream: graphs sharing a pool are replayed strictly in capture order on ONE stream, so the
        # allocator's reuse of freed capture-time temporaries stays race-free while the two streams overlap
        pools = {"V": torch.cuda.graph_pool_handle()}
        # vision stage: reads the staged pixels of the NEXT batch (stage_next_pixels), writes self.patches
        self._next_px = batch["pixel_values"].clone()
        self._px_stage = batch["pixel_values"].clone()
        self._g_vis = torch.cuda.CUDAGraph()
        # thread_local: other host threads (the RCCL watchdog of a multi-rank job) may touch the HIP runtime meanwhile
        with torch.cuda.graph(self._g_vis, pool=pools["V"], stream=self.vis_stream, capture_error_mode="thread_local"):
            self._vision(dict(batch, pixel_values=self._px_stage))
        self._vis_ev = None
        self._segs = self._segments(batch, noise)
        self._graphs = self._capture_segments(self._segs, pools)
        torch.cuda.synchronize()
        self._pending_lr = None
        # the vision stage of step k+1 starts behind this forward segment of step k: it runs under the backward, whose two
        # dependent kernel chains leave most CUs idle.  Default: the third-last forward segment (the last two hold one LLM
        # layer each and are latency-bound with the head trailing them); same-box sweep at B = 32: last 31.52 ms/step,
        # second-last 31.33, third-last 31.14, fourth-last 31.35.  VLA_VIS_AFTER overrides.
        m_fwd = [k for k, sg in enumerate(self._segs) if sg[0] == "M" and sg[3] is not None and sg[3][0] == "f"]
        # (fourth-last forward segment: 26.51 vs 26.64-26.69 ms for the third-last with the one-pipeline forward, same box)
        self._vis_after = m_fwd[min(len(m_fwd) - 1, max(0, int(os.environ.get("VLA_VIS_AFTER", len(m_fwd) - 4))))]
        self._launch_vision()                        # vision stage of the FIRST step (the pixels given to capture)

    def stage_next_pixels(self, pixel_values: torch.Tensor):
        """Captured mode: pixels of the batch AFTER the one the next train_step_graphed() call trains on (its vision stage
        runs during that call).  Without staging, the pixels given to capture() are reused."""
        if getattr(self, "_px_copied", None) is not None:
            torch.cuda.current_stream().wait_event(self._px_copied)   # the vision stream has taken its copy of the old pixels
        self._next_px.copy_(pixel_values)

    def _launch_vision(self, after_event=None):
        """Enqueue the vision stage (ViT + projector -> self.patches) of the staged pixels on the vision stream."""
        V, cur = self.vis_stream, torch.cuda.current_stream()
        if after_event is not None:
            V.wait_event(after_event)           # self.patches of the running step has been consumed (_embed)
        else:
            V.wait_stream(cur)
        tl = getattr(self, "_timeline", None)
        with torch.cuda.stream(V):
            if tl is not None:
                t0 = torch.cuda.Event(enable_timing=True)
                t0.record(V)
            n = self._next_px.numel()
            ops.copy2d(self._next_px, self._px_stage, 1, n, n, n)     # (native copy: no ATen / runtime copy kernel on the step)
            self._px_copied = torch.cuda.Event()
            self._px_copied.record(V)
            self._g_vis.replay()
            self._vis_ev = torch.cuda.Event(enable_timing=tl is not None)
            self._vis_ev.record(V)
            if tl is not None:
                tl.append(("V", -1, t0, self._vis_ev))

    def train_step_pipelined(self, batch, lr: float, noise=None):
        """Eager (un-captured) run of the two-stream schedule."""
        self._ensure_streams()
        loss3 = self._fwd_bwd(batch, noise)
        if self.reducer is not None:
            self.reducer.reduce_async(self.head.P.grad, 0, None)
        self.optimizer_step(lr)
        return loss3

    def train_step_graphed(self, lr: float):
        """Replay of the captured step on the static buffers.  The parameter update of THIS step (RCCL exchange +
        AdamW) is left pending and applied inside the next call, after that step's vision graph - or by flush()."""
        cur = torch.cuda.current_stream()
        self.executed_steps += 1
        self.flush(join=False)
        cur.wait_event(self._vis_ev)           # patches of THIS step (computed during the previous call)
        self._h_end = self._run_segments(self._segs, self._graphs, getattr(self, "_timeline", None),
                                         hooks={self._vis_after: lambda ev: self._launch_vision(ev)})[("end", 0)]
        cur.wait_event(self._px_copied)        # later writes to the staging source are ordered behind the vision copy
        if self.ga > 1:                        # gradient accumulation: join, fold, update only on the boundary micro-step
            cur.wait_event(self._h_end)
            if not self._accumulate():
                return self._loss3
            self._h_end = torch.cuda.Event()
            self._h_end.record(cur)            # the summed gradient is final on the current stream
        if self.reducer is not None:
            aq_off = self.head.P.offsets["action_queries"][0]
            # head / proprio gradients are final when the head stream ends: their exchange starts there, underneath the
            # rest of the LLM backward and the next step's vision stage; the action-query gradient follows the LLM backward
            # (the action queries go first on the exchange stream: the LLM backward ends before the head's tail does, and
            # the next step's LLM stream only waits for them - flush())
            ev_aq = self.reducer.reduce_async(self.head.P.grad, aq_off, None)
            ev_head = self.reducer.reduce_async(self.head.P.grad, 0, aq_off, after_event=self._h_end)
            self._reduced = (ev_aq, ev_head)
        self._pending_lr = lr
        return self._loss3

    def flush(self, join: bool = True):
        """Apply the pending parameter update of the last train_step_graphed (no-op if none).  The update is two AdamW
        launches: the action queries (64 x D, the only trainable tensor the LLM stream reads, in _embed) on the current
        stream, everything else (head + proprio projector, 99.97 % of the bytes) on the head stream, behind that stream's
        last backward kernel and ahead of its first forward kernel of the next step - the 0.5 ms of optimiser traffic
        runs beside the first LLM forward segment instead of in front of it.  ``join`` (default) makes the current stream
        wait for the head-stream half too, so that callers may read any parameter afterwards (checkpoints, evaluation);
        the captured step passes False: its head segments run on the head stream anyway."""
        if getattr(self, "_pending_lr", None) is None:
            return
        lr, self._pending_lr = self._pending_lr, None
        cur, P = torch.cuda.current_stream(), self.head.P
        aq_off = P.offsets["action_queries"][0]
        assert aq_off + rup(math.prod(P.offsets["action_queries"][1]), 8) == P.numel, "action_queries must close the flat buffer"
        self.step_count += 1
        gscale = 1.0
        if self.reducer is not None:
            ev_aq, ev_head = getattr(self, "_reduced", None) or (None, None)
            if ev_aq is not None and ev_head is not None:
                cur.wait_event(ev_aq)                 # the LLM stream needs the action queries only ...
                self.side.wait_event(ev_head)         # ... the 437 MB head exchange is joined by the head stream
                self.reducer._pending = False
            else:
                self.reducer.wait(cur, self.side)
            self._reduced = None
            gscale = self.reducer.grad_scale
        if self.ga > 1:                                # accumulated gradient was assembled on the current stream
            self.side.wait_event(self._h_end)
        with torch.cuda.stream(self.side):
            ops.adamw_(P.data[:aq_off], P.grad[:aq_off], P.m[:aq_off], P.v[:aq_off], self.step_count, lr, gscale=gscale)
            side_done = torch.cuda.Event()
            side_done.record()
        ops.adamw_(P.data[aq_off:], P.grad[aq_off:], P.m[aq_off:], P.v[aq_off:], self.step_count, lr, gscale=gscale)
        if join:
            cur.wait_event(side_done)
        self.head.dirty = True
