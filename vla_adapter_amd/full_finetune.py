"""Full fine-tune step (BASELINE.json configs[3]: "full-backbone unfreeze (ViT + 0.5B LLM + adapter)").

Reference behaviour: with ``use_lora=False`` every parameter of the VLM keeps ``requires_grad=True``
(vla-scripts/finetune.py:846-849) and AdamW runs over all of them plus the action head and the proprio projector
(:903-910); autograd then needs the full-sequence backward through the Qwen2 stack, the projector and the ViT.

What this module adds to ``engine.VLAEngine`` (which covers the adapter-only step):
  * every VLM tensor becomes a view into ONE flat bf16 buffer (+ flat grad / AdamW state): the fused layouts the kernels
    read (q|k|v stacked, gate/up interleaved, zero-padded ViT MLP, K-padded patch embedding) ARE the parameters - they are row
    permutations / zero paddings of the reference tensors, and AdamW is elementwise, so updating them is updating the
    reference tensors (padding rows have zero weight and zero gradient and stay zero);
  * a training forward that keeps what the backward needs (ViT: per-block inputs, LayerNorm outputs and statistics, qkv,
    attention output and log-sum-exp, GELU pre-activations; LLM: the two RMSNorm outputs and the SwiGLU product per layer);
  * the backward: dX chain of the adapter-only path over ALL rows, plus dW = dY^T.X for every Linear (NT GEMMs on
    transposed operands), bias / LayerNorm / RMSNorm / pos-embed gradients in one fp32 accumulator cast once, ViT attention
    backward (same kernel as the LLM's), GELU backward, patch-embed dW, embedding-table gradient;
  * one AdamW launch over the VLM buffer, one over the head buffer; W^T operands of the dX products rebuilt after the update.
Not covered: LayerScale / prefix-token backbones (DINOv2) - their folded LayerScale is a product of parameters, not a layout.
Single stream, eager or one linear hipGraph: the step is GEMM-bound end to end (3x the forward FLOPs), unlike the adapter-only
step whose backward is a chain of small kernels.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, List, Optional, Tuple

import torch

from . import engine as E
from . import ops
from .ops import ACT_GELU, ACT_GELU_TANH, ACT_NONE, ACT_SWIGLU, BF16

rup = E.rup


class _Slot:
    """One trainable tensor: where it lives (dict / attribute) and its name in the flat buffer."""

    def __init__(self, name, holder, key, vector: bool):
        self.name, self.holder, self.key, self.vector = name, holder, key, vector

    def get(self):
        return self.holder[self.key] if isinstance(self.holder, dict) else getattr(self.holder, self.key)

    def set(self, t):
        if isinstance(self.holder, dict):
            self.holder[self.key] = t
        else:
            setattr(self.holder, self.key, t)


class FullFinetune:
    def __init__(self, eng: E.VLAEngine):
        cfg = eng.cfg
        assert not getattr(eng, "fp8_frozen", False), "full fine-tune trains the backbone weights: the fp8 frozen-weight path does not apply"
        assert len(eng.vits) == 1 and cfg.n_img == 1, "full fine-tune path: single-backbone, single-image configuration (BASELINE configs[3])"
        vc = cfg.vit[0]
        if vc.layerscale or vc.n_prefix:
            raise NotImplementedError("full fine-tune of LayerScale / prefix-token backbones (DINOv2) is not built")
        assert cfg.llm.dh == 64, "fused RoPE backward needs head dim 64"
        self.eng, self.cfg, self.dev = eng, cfg, eng.device
        self.vit, self.llm, self.head = eng.vits[0], eng.llm, eng.head
        eng.full_llm_backward = True
        # ---- adopt every VLM tensor into one flat buffer: matrices first, then the vector section (fp32-accumulated grads)
        slots: List[_Slot] = []
        v = self.vit
        slots.append(_Slot("vit.wpe", v, "wpe", False))
        for i, b in enumerate(v.blocks):
            for k in ("wqkv", "wproj", "w1", "w2"):
                slots.append(_Slot(f"vit.{i}.{k}", b, k, False))
        for k in eng.proj:
            if k.endswith("weight"):
                slots.append(_Slot("proj." + k, eng.proj, k, False))
        for i, L in enumerate(self.llm.layers):
            for k in ("wqkv", "wo", "wgu", "wd"):
                slots.append(_Slot(f"llm.{i}.{k}", L, k, False))
        slots.append(_Slot("llm.embed", self.llm, "embed", False))
        slots += [_Slot("vit.bpe", v, "bpe", True), _Slot("vit.pos", v, "pos", True)]
        for i, b in enumerate(v.blocks):
            for k in ("n1w", "n1b", "bqkv", "bproj", "n2w", "n2b", "b1", "b2"):
                slots.append(_Slot(f"vit.{i}.{k}", b, k, True))
        for k in eng.proj:
            if k.endswith("bias"):
                slots.append(_Slot("proj." + k, eng.proj, k, True))
        for i, L in enumerate(self.llm.layers):
            for k in ("n1", "n2", "bqkv"):
                slots.append(_Slot(f"llm.{i}.{k}", L, k, True))
        slots.append(_Slot("llm.norm", self.llm, "norm", True))
        self.slots = slots
        self.P = E.FlatParams([(s.name, tuple(s.get().shape)) for s in slots], self.dev)
        for s in slots:
            self.P.view(s.name).copy_(s.get())
            s.set(self.P.view(s.name))
        self.vec_off = self.P.offsets["vit.bpe"][0]
        self.acc32 = torch.zeros(self.P.numel - self.vec_off, device=self.dev, dtype=torch.float32)
        # W^T operands of the ViT / projector dX products (the LLM's already exist: frozen-path dX)
        z = lambda r, c: torch.empty(r, c, device=self.dev, dtype=BF16)
        for b in v.blocks:
            b["wqkvT"], b["wprojT"], b["w1T"], b["w2T"] = z(v.cfg.d, 3 * v.cfg.d), z(v.cfg.d, v.cfg.d), z(v.cfg.d, v.mlp_pad), z(v.mlp_pad, v.cfg.d)
        self.projT = {k: z(w.shape[1], w.shape[0]) for k, w in eng.proj.items() if k.endswith("weight")}
        self.refresh_transposes()
        self.step_count = 0
        self._key = None

    # ------------------------------------------------------------------------------------------------ bookkeeping
    def G(self, name):
        return self.P.g(name)

    def A(self, name):
        """fp32 accumulator of a vector-section parameter."""
        off, shape = self.P.offsets[name]
        return self.acc32[off - self.vec_off:off - self.vec_off + math.prod(shape)].view(shape)

    def refresh_transposes(self):
        for b in self.vit.blocks:
            for k in ("wqkv", "wproj", "w1", "w2"):
                ops.transpose(b[k], out=b[k + "T"])
        for k, t in self.projT.items():
            ops.transpose(self.eng.proj[k], out=t)
        for L in self.llm.layers:
            for k in ("wqkv", "wo", "wgu", "wd"):
                ops.transpose(L[k], out=L[k + "T"])

    def reference_named_gradients(self) -> Dict[str, torch.Tensor]:
        """Gradients under the reference's state-dict names (fused layouts undone) - for parity tests / checkpoints."""
        cfg, out = self.cfg, {}
        c, v = cfg.llm, self.vit
        H, KV, dh, I, D = c.heads, c.kv_heads, c.dh, c.inter, c.d
        for i in range(c.n_layers):
            p = f"language_model.model.layers.{i}."
            gq = self.G(f"llm.{i}.wqkv")
            out[p + "self_attn.q_proj.weight"], out[p + "self_attn.k_proj.weight"], out[p + "self_attn.v_proj.weight"] = gq[:H * dh], gq[H * dh:(H + KV) * dh], gq[(H + KV) * dh:]
            gb = self.G(f"llm.{i}.bqkv")
            out[p + "self_attn.q_proj.bias"], out[p + "self_attn.k_proj.bias"], out[p + "self_attn.v_proj.bias"] = gb[:H * dh], gb[H * dh:(H + KV) * dh], gb[(H + KV) * dh:]
            out[p + "self_attn.o_proj.weight"] = self.G(f"llm.{i}.wo")
            ggu = self.G(f"llm.{i}.wgu").view(I // 16, 2, 16, D)
            out[p + "mlp.gate_proj.weight"], out[p + "mlp.up_proj.weight"] = ggu[:, 0].reshape(I, D), ggu[:, 1].reshape(I, D)
            out[p + "mlp.down_proj.weight"] = self.G(f"llm.{i}.wd")
            out[p + "input_layernorm.weight"], out[p + "post_attention_layernorm.weight"] = self.G(f"llm.{i}.n1"), self.G(f"llm.{i}.n2")
        out["language_model.model.norm.weight"], out["language_model.model.embed_tokens.weight"] = self.G("llm.norm"), self.G("llm.embed")
        pre = "vision_backbone.featurizer."
        P_ = v.cfg.patch
        out[pre + "patch_embed.proj.weight"] = self.G("vit.wpe")[:, :3 * P_ * P_].reshape(v.cfg.d, 3, P_, P_)
        out[pre + "patch_embed.proj.bias"], out[pre + "pos_embed"] = self.G("vit.bpe"), self.G("vit.pos").reshape(1, -1, v.cfg.d)
        for i in range(len(v.blocks)):
            q = f"{pre}blocks.{i}."
            g = lambda k: self.G(f"vit.{i}.{k}")
            out[q + "norm1.weight"], out[q + "norm1.bias"], out[q + "norm2.weight"], out[q + "norm2.bias"] = g("n1w"), g("n1b"), g("n2w"), g("n2b")
            out[q + "attn.qkv.weight"], out[q + "attn.qkv.bias"] = g("wqkv"), g("bqkv")
            out[q + "attn.proj.weight"], out[q + "attn.proj.bias"] = g("wproj"), g("bproj")
            out[q + "mlp.fc1.weight"], out[q + "mlp.fc1.bias"] = g("w1")[:v.cfg.mlp], g("b1")[:v.cfg.mlp]
            out[q + "mlp.fc2.weight"], out[q + "mlp.fc2.bias"] = g("w2")[:, :v.cfg.mlp], g("b2")
        for k in self.eng.proj:
            out["projector." + k] = self.G("proj." + k)
        return out

    # ------------------------------------------------------------------------------------------------ buffers
    def _alloc(self, B: int, S: int):
        if self._key == (B, S):
            return
        v, c, dev = self.vit, self.cfg.llm, self.dev
        e = lambda *s, dt=BF16: torch.empty(*s, device=dev, dtype=dt)
        nb, d, T = len(v.blocks), v.cfg.d, v.cfg.n_patches
        Mv = B * T
        self.vX = e(nb + 1, Mv, d)                       # block inputs; vX[nb] = output of the last useful block
        self.vH1, self.vH2, self.vA, self.vXm = e(nb, Mv, d), e(nb, Mv, d), e(nb, Mv, d), e(nb, Mv, d)
        self.vS1, self.vS2 = e(nb, Mv, 2, dt=torch.float32), e(nb, Mv, 2, dt=torch.float32)
        self.vQKV, self.vLSE = e(nb, Mv, 3 * d), e(nb, B, v.cfg.heads, T, dt=torch.float32)
        self.vMpre, self.vMact = e(nb, Mv, v.mlp_pad), e(nb, Mv, v.mlp_pad)
        self.vcols = None
        n, D, I = c.n_layers, c.d, c.inter
        M = B * S
        self.N1, self.N2, self.Hs = e(n, M, D), e(n, M, D), e(n, M, I)
        # gradient scratch
        self.g_d, self.g_big, self.g_mid = e(Mv, d), e(Mv, v.mlp_pad), e(Mv, 3 * d)
        self.pj_pre, self.pj_act = {}, {}
        self._key = (B, S)

    # ------------------------------------------------------------------------------------------------ forward
    def _vit_forward(self, pixels):
        v, cfg = self.vit, self.vit.cfg
        B, T, d = pixels.shape[0], cfg.n_patches, cfg.d
        nb = len(v.blocks)
        self.vcols = ops.im2col_patch(pixels, 0, cfg.patch, v.kpe)
        ops.gemm_nt(self.vcols, v.wpe, bias=v.bpe, residual=v.pos, res_mod=T, out=self.vX[0])
        act = ACT_GELU_TANH if cfg.gelu_tanh else ACT_GELU
        dh = d // cfg.heads
        for i, b in enumerate(v.blocks):
            x = self.vX[i]
            ops.N.check(ops._lib().vla_layernorm_fwd(ops._st(), ops._p(x), ops._p(b["n1w"]), ops._p(b["n1b"]), ops._p(self.vH1[i]), ops._p(self.vS1[i]),
                                                     x.shape[0], d, d, d, cfg.eps), "layernorm_fwd")
            qkv = ops.gemm_nt(self.vH1[i], b["wqkv"], bias=b["bqkv"], out=self.vQKV[i]).view(B, T, 3 * d)
            dsc = ops._attn_desc(qkv[:, :, :d], qkv[:, :, d:2 * d], qkv[:, :, 2 * d:], self.vA[i].view(B, T, d), self.vLSE[i], None, False,
                                 dh ** -0.5, cfg.heads, cfg.heads, dh)
            ops.N.check(ops._lib().vla_attn_fwd(ops._st(), C.byref(dsc)), "attn_fwd")
            ops.gemm_nt(self.vA[i], b["wproj"], bias=b["bproj"], residual=x, out=self.vXm[i])
            xm = self.vXm[i]
            ops.N.check(ops._lib().vla_layernorm_fwd(ops._st(), ops._p(xm), ops._p(b["n2w"]), ops._p(b["n2b"]), ops._p(self.vH2[i]), ops._p(self.vS2[i]),
                                                     xm.shape[0], d, d, d, cfg.eps), "layernorm_fwd")
            ops.gemm_nt(self.vH2[i], b["w1"], bias=b["b1"], out=self.vMpre[i])          # pre-activation kept for the GELU backward
            if cfg.gelu_tanh:
                ops.gemm_nt(self.vH2[i], b["w1"], bias=b["b1"], act=act, out=self.vMact[i])
            else:
                ops.N.check(ops._lib().vla_gelu_fwd(ops._st(), ops._p(self.vMpre[i]), ops._p(self.vMact[i]), self.vMpre[i].numel()), "gelu_fwd")
            ops.gemm_nt(self.vMact[i], b["w2"], bias=b["b2"], residual=xm, out=self.vX[i + 1])
        return self.vX[nb]

    def _proj_forward(self, feats):
        eng, cfg = self.eng, self.cfg
        pj = eng.proj
        self.pj_in = feats
        self.pj_pre["fc1"] = ops.gemm_nt(feats, pj["fc1.weight"], bias=pj["fc1.bias"])
        self.pj_act["fc1"] = ops.gelu_fwd(self.pj_pre["fc1"])
        dst = eng.patches.view(-1, cfg.llm.d)
        if cfg.fused:
            self.pj_pre["fc2"] = ops.gemm_nt(self.pj_act["fc1"], pj["fc2.weight"], bias=pj["fc2.bias"])
            self.pj_act["fc2"] = ops.gelu_fwd(self.pj_pre["fc2"])
            ops.gemm_nt(self.pj_act["fc2"], pj["fc3.weight"], bias=pj["fc3.bias"], out=dst)
        else:
            ops.gemm_nt(self.pj_act["fc1"], pj["fc2.weight"], bias=pj["fc2.bias"], out=dst)

    def _llm_forward(self, B, S, kmask):
        llm, c = self.llm, self.cfg.llm
        D, H, KV, dh = c.d, c.heads, c.kv_heads, c.dh
        llm.fwd_begin(B, S, kmask, 0)
        for i, L in enumerate(llm.layers):
            x = llm.HS[i].view(-1, D)
            llm._rms(x, L["n1"], self.N1[i], llm.R1[i])
            qkv = llm.QKV[i]
            ops.gemm_nt(self.N1[i], L["wqkv"], bias=L["bqkv"], out=qkv, rope=(1, llm.cos, llm.sin, S, dh, (H + KV) * dh))
            llm._attn_fwd(qkv.view(B, S, -1), i, 0, B, S)
            x1 = llm.X1[i]
            ops.gemm_nt(llm.AO[i], L["wo"], residual=x, out=x1)
            llm._rms(x1, L["n2"], self.N2[i], llm.R2[i])
            ops.gemm_nt(self.N2[i], L["wgu"], act=ACT_SWIGLU, out=llm.GU[i], out2=self.Hs[i])
            ops.gemm_nt(self.Hs[i], L["wd"], residual=x1, out=llm.HS[llm.out_slot(i)].view(-1, D))
        llm.fwd_final()

    def forward(self, batch: Dict[str, torch.Tensor], noise: Optional[torch.Tensor] = None):
        eng, cfg = self.eng, self.cfg
        eng._vision_begin(batch)
        B, S = eng.B, eng.S
        self._alloc(B, S)
        feats = self._vit_forward(batch["pixel_values"])
        self._proj_forward(feats)
        mm = eng._embed(batch)
        self._batch = batch
        self._llm_forward(B, S, mm)
        return self.head.forward(self.llm.HS, eng.pos1, batch["proprio"], eng.Np, noise)

    # ------------------------------------------------------------------------------------------------ backward
    def _dw(self, dy2d, x2d, out, a_group=None):
        """out[N, K] = dY[M, N]^T . X[M, K]  (both transposed to M-contiguous operands, M padded to 64)."""
        Mp = rup(dy2d.shape[0], 64)
        ops.gemm_nt(ops.transpose(dy2d, ld_out=Mp), ops.transpose(x2d, ld_out=Mp), out=out)

    def _llm_backward(self, dHS):
        llm, c, B, S = self.llm, self.cfg.llm, self.eng.B, self.eng.S
        n, D, H, KV, dh, I = c.n_layers, c.d, c.heads, c.kv_heads, c.dh, c.inter
        M = B * S
        lib, st, p = ops._lib(), ops._st, ops._p
        llm.bwd_begin(dHS, 0)
        # final norm weight: dy = dHS[n], x = raw output of the last layer
        ops.N.check(lib.vla_rmsnorm_dw(st(), p(dHS[n].view(M, D)), p(llm.HS[n + 1].view(M, D)), p(llm.RF), p(self.A("llm.norm")), M, D), "rmsnorm_dw")
        for i in range(n - 1, -1, -1):
            L, d, other = llm.layers[i], llm._d, llm._other
            if i < n - 1:
                ops.add_(d, dHS[i + 1].view(M, D))
            d_gu, d_n = llm.d_gu[:M], llm.d_n[:M]
            self._dw(d, self.Hs[i], self.G(f"llm.{i}.wd"))
            ops.gemm_swiglu_bwd(d, L["wdT"], llm.GU[i], out=d_gu)
            self._dw(d_gu, self.N2[i], self.G(f"llm.{i}.wgu"))
            ops.gemm_nt(d_gu, L["wguT"], out=d_n)
            ops.N.check(lib.vla_rmsnorm_dw(st(), p(d_n), p(llm.X1[i]), p(llm.R2[i]), p(self.A(f"llm.{i}.n2")), M, D), "rmsnorm_dw")
            d1 = ops.rmsnorm_bwd(d_n, llm.X1[i], L["n2"], llm.R2[i], dres=d, out=other)
            self._dw(d1, llm.AO[i], self.G(f"llm.{i}.wo"))
            dao = ops.gemm_nt(d1, L["woT"], out=d_n)
            q, k, v = llm._attn_views(llm.QKV[i].view(B, S, -1))
            d_qkv = llm.d_qkv[:M]
            dq, dk, dv = llm._attn_views(d_qkv.view(B, S, -1))
            ops.attn_bwd(dao.view(B, S, -1), q, k, v, llm.AO[i].view(B, S, -1), llm.LSE[i], H, KV, dh, True, llm.kmask, dq=dq, dk=dk, dv=dv,
                         rope=(llm.cos, llm.sin))
            self._dw(d_qkv, self.N1[i], self.G(f"llm.{i}.wqkv"))
            ops.colsum_(d_qkv, self.A(f"llm.{i}.bqkv"))
            ops.gemm_nt(d_qkv, L["wqkvT"], out=d_n)
            ops.N.check(lib.vla_rmsnorm_dw(st(), p(d_n), p(llm.HS[i].view(M, D)), p(llm.R1[i]), p(self.A(f"llm.{i}.n1")), M, D), "rmsnorm_dw")
            d_new = ops.rmsnorm_bwd(d_n, llm.HS[i].view(M, D), L["n1"], llm.R1[i], dres=d1, out=d)
            llm._d, llm._other = d_new, d1
        return llm.bwd_result()           # [B, S, D]: gradient w.r.t. inputs_embeds

    def _proj_backward(self, dX0):
        eng, cfg = self.eng, self.cfg
        B, S, Np, D = eng.B, eng.S, eng.Np, cfg.llm.d
        pj = eng.proj
        # gradient of the projected patches = rows 1..Np of every sequence of dX0, compacted by the native strided copy
        dp = torch.empty(B * Np, D, device=self.dev, dtype=BF16)
        for b in range(B):            # B strided row blocks: one native copy each (off the GEMM-bound critical path)
            ops.copy2d(dX0[b, 1], dp[b * Np], Np, D, D, D)
        last = "fc3" if cfg.fused else "fc2"
        prev = "fc2" if cfg.fused else "fc1"
        self._dw(dp, self.pj_act[prev], self.G(f"proj.{last}.weight"))
        ops.colsum_(dp, self.A(f"proj.{last}.bias"))
        dh_ = ops.gemm_nt(dp, self.projT[f"{last}.weight"])
        d_pre = ops.gelu_bwd(dh_, self.pj_pre[prev])
        if cfg.fused:
            self._dw(d_pre, self.pj_act["fc1"], self.G("proj.fc2.weight"))
            ops.colsum_(d_pre, self.A("proj.fc2.bias"))
            dh_ = ops.gemm_nt(d_pre, self.projT["fc2.weight"])
            d_pre = ops.gelu_bwd(dh_, self.pj_pre["fc1"])
        self._dw(d_pre, self.pj_in, self.G("proj.fc1.weight"))
        ops.colsum_(d_pre, self.A("proj.fc1.bias"))
        return ops.gemm_nt(d_pre, self.projT["fc1.weight"])          # d features [B*Np, vis_dim]

    def _vit_backward(self, dfeat):
        v, cfg = self.vit, self.vit.cfg
        B, T, d = self.eng.B, cfg.n_patches, cfg.d
        nb = len(v.blocks)
        dh = d // cfg.heads
        lib, st, p = ops._lib(), ops._st, ops._p
        dx = dfeat                                      # gradient w.r.t. vX[nb]
        for i in range(nb - 1, -1, -1):
            b = v.blocks[i]
            g = lambda k: self.G(f"vit.{i}.{k}")
            a = lambda k: self.A(f"vit.{i}.{k}")
            # x_out = x_mid + fc2(gelu(fc1(LN2(x_mid))))
            self._dw(dx, self.vMact[i], g("w2"))
            ops.colsum_(dx, a("b2"))
            dm = ops.gemm_nt(dx, b["w2T"], out=self.g_big)
            if cfg.gelu_tanh:
                raise NotImplementedError("tanh-GELU backward")
            dpre = ops.gelu_bwd(dm, self.vMpre[i])
            self._dw(dpre, self.vH2[i], g("w1"))
            ops.colsum_(dpre, a("b1"))
            dh2 = ops.gemm_nt(dpre, b["w1T"], out=self.g_d)
            dxm = torch.empty_like(dx)
            ops.N.check(lib.vla_layernorm_bwd(st(), p(dh2), p(self.vXm[i]), p(b["n2w"]), p(self.vS2[i]), p(dxm), p(a("n2w")), p(a("n2b")),
                                              dx.shape[0], d, d, d, d), "layernorm_bwd")
            ops.add_(dxm, dx)                           # residual
            # x_mid = x_in + proj(attn(qkv(LN1(x_in))))
            self._dw(dxm, self.vA[i], g("wproj"))
            ops.colsum_(dxm, a("bproj"))
            da = ops.gemm_nt(dxm, b["wprojT"], out=self.g_d)
            qkv = self.vQKV[i].view(B, T, 3 * d)
            dqkv = self.g_mid.view(B, T, 3 * d)
            ops.attn_bwd(da.view(B, T, d), qkv[:, :, :d], qkv[:, :, d:2 * d], qkv[:, :, 2 * d:], self.vA[i].view(B, T, d), self.vLSE[i],
                         cfg.heads, cfg.heads, dh, False, None, dq=dqkv[:, :, :d], dk=dqkv[:, :, d:2 * d], dv=dqkv[:, :, 2 * d:])
            self._dw(self.g_mid, self.vH1[i], g("wqkv"))
            ops.colsum_(self.g_mid, a("bqkv"))
            dh1 = ops.gemm_nt(self.g_mid, b["wqkvT"], out=self.g_d)
            dxi = torch.empty_like(dx)
            ops.N.check(lib.vla_layernorm_bwd(st(), p(dh1), p(self.vX[i]), p(b["n1w"]), p(self.vS1[i]), p(dxi), p(a("n1w")), p(a("n1b")),
                                              dx.shape[0], d, d, d, d), "layernorm_bwd")
            ops.add_(dxi, dxm)
            dx = dxi
        # patch embedding: x0 = cols . Wpe^T + bpe + pos
        self._dw(dx, self.vcols, self.G("vit.wpe"))
        ops.colsum_(dx, self.A("vit.bpe"))
        ops.N.check(lib.vla_colsum_bf16(st(), p(dx), p(self.A("vit.pos")), B, T * d, T * d, 1, 0, 0), "colsum(pos)")   # sum over the batch

    def backward(self, pred, actions, gscale: float = 1.0):
        eng, head, llm = self.eng, self.head, self.llm
        B, S, Np = eng.B, eng.S, eng.Np
        ops.zero_(self.acc32)
        ops.zero_(self.G("llm.embed"))
        loss3, dpred = ops.l1_loss(pred, eng._to_bf16(actions), True, gscale)
        dHS = eng._dhs(0)
        head.backward(dpred, dHS, 0)
        dX0 = self._llm_backward(dHS).contiguous()
        dq = ops.action_query_grad(dX0, eng.pos0, Np, 0)
        ops.cast_f32_bf16(dq, out=head.P.g("action_queries"))
        ops.N.check(ops._lib().vla_embed_grad(ops._st(), ops._p(dX0), ops._p(self._batch["input_ids"]), ops._p(eng.qidx0), ops._p(self.G("llm.embed")),
                                              B, self._batch["input_ids"].shape[1], Np, self.cfg.llm.d, self.cfg.llm.vocab), "embed_grad")
        dfeat = self._proj_backward(dX0)
        self._vit_backward(dfeat)
        ops.cast_f32_bf16(self.acc32, out=self.P.grad[self.vec_off:])       # every bias / norm / pos-embed gradient in one cast
        return loss3

    # ------------------------------------------------------------------------------------------------ update
    def optimizer_step(self, lr: float, beta1=0.9, beta2=0.999, eps=1e-8, wd=0.01, refresh: bool = True):
        self.step_count += 1
        gscale = 1.0
        red = self.eng.reducer
        if red is not None:
            red.reduce_async(self.P.grad, 0, None)
            red.reduce_async(self.head.P.grad, 0, None)
            red.wait()
            gscale = red.grad_scale
        P, HP = self.P, self.head.P
        ops.adamw_(P.data, P.grad, P.m, P.v, self.step_count, lr, beta1, beta2, eps, wd, gscale=gscale)
        ops.adamw_(HP.data, HP.grad, HP.m, HP.v, self.step_count, lr, beta1, beta2, eps, wd, gscale=gscale)
        self.head.dirty = True
        if refresh:
            self.refresh_transposes()

    def train_step(self, batch, lr: float, noise=None):
        pred = self.forward(batch, noise)
        loss3 = self.backward(pred, batch["actions"])
        self.optimizer_step(lr)
        return loss3

    # ---- hipGraph replay: the ~3000 launches of a step cost more host time than GPU time when issued from Python ------------
    def capture(self, batch: Dict[str, torch.Tensor], noise: Optional[torch.Tensor] = None, warmup: int = 2):
        """Forward + backward as ONE linear hipGraph on the static ``batch`` / ``noise`` buffers (copy new data into them before
        each replay); AdamW stays outside (host-side bias corrections), the W^T rebuild is a second small graph."""
        self._cap_stream = torch.cuda.Stream()
        for _ in range(warmup):
            self.head.dirty = True
            self.backward(self.forward(batch, noise), batch["actions"])
        torch.cuda.synchronize()
        self.head.dirty = True                       # the head's own W^T / padded-operand refresh becomes part of the graph
        pool = torch.cuda.graph_pool_handle()
        self._g_step = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._g_step, pool=pool, stream=self._cap_stream, capture_error_mode="thread_local"):
            self._loss3 = self.backward(self.forward(batch, noise), batch["actions"])
        self._g_t = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._g_t, pool=pool, stream=self._cap_stream, capture_error_mode="thread_local"):
            self.refresh_transposes()
        torch.cuda.synchronize()

    def train_step_graphed(self, lr: float):
        self._g_step.replay()
        self.optimizer_step(lr, refresh=False)
        self._g_t.replay()
        return self._loss3
