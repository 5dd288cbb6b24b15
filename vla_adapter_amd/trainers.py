"""Backbone-training steps: LoRA (the reference's documented recipe, README.md:254-274 / vla-scripts/finetune.py:832-844) and the
full fine-tune (``use_lora=False``: every VLM parameter keeps ``requires_grad``, :846-849, 903-910) through ANY backbone the
engine runs - one or two ViTs (SigLIP; DINOv2 with cls + register prefix tokens and LayerScale as its own parameter,
modeling_prismatic.py:58-66), one or two images per sample (the fused layout of :196-237), Qwen2.5-0.5B or -1.5B geometry.

``engine.VLAEngine`` covers the adapter-only step (frozen backbones, live-row backward).  This module adds, for both modes, ONE
training forward that keeps what the backward needs and ONE explicit backward; what differs between the modes is what a
Linear does with its weight (the ``_lin`` / ``_lin_bwd`` pair):

  full   y = x W^T + b through the base GEMM's fused epilogues; backward dx = dy W (NT GEMM on the kept W^T) and dW = dy^T x as a
         TN GEMM on dy and x as they lie in memory (vla_gemm_bf16_tn: no operand transposes), bias gradient by column sums.
         Every VLM tensor is a view into ONE flat bf16 buffer in the fused layouts the kernels read (q|k|v stacked, gate/up
         interleaved, zero-padded ViT MLP, K-padded patch embedding): they are row permutations / zero paddings of the
         reference tensors and AdamW is elementwise, so updating them IS updating the reference tensors.
  lora   peft ``LoraConfig(r, lora_alpha=2r, target_modules="all-linear", init_lora_weights="gaussian")``:
         y = x W^T + b + 2 (x A^T) B^T.  The low-rank branch runs INSIDE the base product (the GEMM's K extension):
         t = 2 x A_cat^T (one skinny GEMM), then y = [x | t] . [W | B_blk]^T in one fp32 accumulator - the base GEMM keeps its
         bias / RoPE / SwiGLU / residual epilogue and no read-modify-write pass over y exists.  Backward alike:
         dt = 2 dy B_blk (skinny), dx = [dy | dt] . [W^T | A_cat^T]^T, dA_cat = dt^T x and dB_j = dy_j^T t_j as TN GEMMs.
         One rounding of y instead of peft's three (base output, branch, sum): closer to the fp32 result than the reference's
         own bf16 arithmetic; the oracle's LORA_FUSED switch restates it.  PARITY UNPINNED either way (peft is not importable).
         Parameters are stored per pair under peft's names in one flat buffer; a fused base Linear (q|k|v, gate/up) carries
         several pairs: A_cat is a view of adjacent A's, B_blk the dense block matrix (rebuilt after every update).
         Ranks are zero-padded to a multiple of 64 (the GEMM's K granule); padding has zero value and zero gradient.

Frozen in LoRA mode, as under peft: patch embedding (Conv2d), position embedding, cls / register tokens, LayerScale, norms,
biases, the token embedding.  The action head, the proprio projector and the action queries train in both modes (engine.Head).
"""
from __future__ import annotations

import ctypes as C
import math
import os
from typing import Dict, List, Optional, Tuple

import torch

from . import engine as E
from . import ops
from .ops import ACT_GELU, ACT_GELU_TANH, ACT_NONE, ACT_SWIGLU, BF16

rup = E.rup


# ------------------------------------------------------------------------------------------------ LoRA pairs of one fused Linear
class LoraLinear:
    """LoRA pairs of one fused base Linear W [n_out, k_in].  projs: [(peft leaf name, ("range", lo, hi) | ("group16", offset, n_rows))].
    k_real / n_real: the Linear's true in / out width when the engine pads it (ViT MLP 4304 -> 4352): padding columns of A and
    padding rows of B are zero and stay zero (zero gradient: their activations / output gradients are zero)."""

    def __init__(self, name: str, n_out: int, k_in: int, projs, r: int, k_real: Optional[int] = None, n_real: Optional[int] = None):
        self.name, self.n_out, self.k_in, self.projs, self.r = name, n_out, k_in, projs, r
        self.k_real, self.n_real = k_real or k_in, n_real or n_out
        self.rp = rup(r, 64)
        self.Rr = self.rp * len(projs)

    def spec(self):
        out = [(f"{self.name}.{p}.lora_A", (self.rp, self.k_in)) for p, _ in self.projs]
        return out, [(f"{self.name}.{p}.lora_B", (self._rows(d), self.rp)) for p, d in self.projs]

    @staticmethod
    def _rows(d):
        return d[2] - d[1] if d[0] == "range" else d[2]

    def bind(self, P: E.FlatParams, device):
        self.P = P
        a0 = P.offsets[f"{self.name}.{self.projs[0][0]}.lora_A"][0]
        self.A_cat = P.data[a0:a0 + self.Rr * self.k_in].view(self.Rr, self.k_in)
        self.gA_cat = P.grad[a0:a0 + self.Rr * self.k_in].view(self.Rr, self.k_in)
        self.A_catT = torch.empty(self.k_in, self.Rr, device=device, dtype=BF16)
        self.single = len(self.projs) == 1 and self.projs[0][1][0] == "range"
        if self.single:
            self.B_blk = P.view(f"{self.name}.{self.projs[0][0]}.lora_B")
        else:
            self.B_blk = torch.zeros(self.n_out, self.Rr, device=device, dtype=BF16)
        self.B_blkT = torch.empty(self.Rr, self.n_out, device=device, dtype=BF16)

    def init_(self, gen):
        """init_lora_weights="gaussian": A ~ N(0, (1/r)^2), B = 0; rank padding and width padding zero."""
        for p, _ in self.projs:
            A = self.P.view(f"{self.name}.{p}.lora_A")
            A.zero_()
            A[:self.r, :self.k_real] = (torch.randn(self.r, self.k_real, generator=gen, device=A.device) / self.r).to(BF16)
            self.P.view(f"{self.name}.{p}.lora_B").zero_()

    def refresh(self):
        """Derived operands after a parameter change: B_blk (dense block matrix), A_cat^T, B_blk^T."""
        if not self.single:
            for j, (p, d) in enumerate(self.projs):
                Bj = self.P.view(f"{self.name}.{p}.lora_B")
                dst = self.B_blk[d[1], j * self.rp:]
                if d[0] == "range":
                    ops.copy2d(Bj, dst, Bj.shape[0], self.rp, self.rp, self.Rr)
                else:               # 16-row groups every 32 rows (gate / up interleave), starting at row d[1]
                    ops.copy2d(Bj, dst, Bj.shape[0], self.rp, self.rp, self.Rr, d_group=(16, 32 * self.Rr))
        ops.transpose(self.A_cat, out=self.A_catT)
        ops.transpose(self.B_blk, out=self.B_blkT)

    def grads(self, dy2d, x2d, t2d, dt2d, emit, dropped: bool = False):
        """dA_cat = dt^T x (one TN product into the flat gradient), dB_j = dy_j^T t_j per pair (the columns of dy that belong to
        pair j: a range, or 16-column groups every 32 for gate / up).  emit(a, b, out, alpha, a_cols) launches or collects them.
        dropped (lora_dropout > 0): x2d is [pairs, M, K], pair j's own dropped input - one dA product per pair."""
        if dropped:
            for j in range(len(self.projs)):
                emit(dt2d[:, j * self.rp:(j + 1) * self.rp], x2d[j], self.gA_cat[j * self.rp:(j + 1) * self.rp])
        else:
            emit(dt2d, x2d, self.gA_cat)
        for j, (p, d) in enumerate(self.projs):
            gB = self.P.g(f"{self.name}.{p}.lora_B")
            tj = t2d[:, j * self.rp:(j + 1) * self.rp]
            if d[0] == "range":
                emit(dy2d[:, d[1]:d[2]], tj, gB)
            else:
                emit(dy2d, tj, gB, 1.0, (d[2], 16, 32, d[1]))


# ------------------------------------------------------------------------------------------------ shared training core
class _Slot:
    """One trainable tensor of the full fine-tune: where it lives (dict / attribute) and its name in the flat buffer."""

    def __init__(self, name, holder, key, vector: bool):
        self.name, self.holder, self.key, self.vector = name, holder, key, vector

    def get(self):
        return self.holder[self.key] if isinstance(self.holder, dict) else getattr(self.holder, self.key)

    def set(self, t):
        if isinstance(self.holder, dict):
            self.holder[self.key] = t
        else:
            setattr(self.holder, self.key, t)


class BackboneTrainer:
    """Training forward / backward through ViT(s) -> projector -> Qwen2 stack -> action head, shared by both modes."""

    mode = "?"

    def __init__(self, eng: E.VLAEngine):
        cfg = eng.cfg
        assert not getattr(eng, "fp8_frozen", False), "engine.enable_fp8_frozen() is the adapter-only forward's opt-in; LoRA has its own (LoRAFinetune(fp8=True))"
        self.eng, self.cfg, self.dev = eng, cfg, eng.device
        self.vits, self.llm, self.head = eng.vits, eng.llm, eng.head
        for v in self.vits:
            assert not v.cfg.gelu_tanh, "tanh-GELU backward is not built (no backbone of the reference uses it)"
            v.fold_layerscale(False)                 # LayerScale acts as its own (trainable or frozen) parameter from here on
        eng.full_llm_backward = True
        self.step_count = 0
        self._key = None
        # test hook: {("llm", i) | ("vit", j, i): {}} -> the backward fills "d_out" (gradient w.r.t. the layer's output) and "d_in"
        # (w.r.t. its input) with clones, so that ONE layer's dX / dW can be checked against the oracle's autograd of that layer
        # on the run's own activations (tests/test_layer_gradients_gpu.py)
        self.taps: Optional[dict] = None
        # granularity of the data-parallel exchange: gradient ranges are handed over every `exchange_layers` LLM layers (4 x 30 MB
        # of bf16 gradients at the 0.5B geometry: large messages for the point-to-point xGMI links) / `exchange_blocks` ViT blocks
        self.exchange_layers, self.exchange_blocks = 4, 7
        self.n_active = min(cfg.llm.n_layers, cfg.num_blocks)      # LLM layers that reach the loss (see _segments)
        self.ga, self._micro, self._gacc = 1, 0, None             # gradient accumulation (set_grad_accumulation)
        self.objective = "l1"
        self.overlap_update = not os.environ.get("VLA_NO_UPDATE_OVERLAP")      # AdamW range by range under the backward (_run)
        # Streams of the step schedule (_segments): the caller's stream carries the dX chain, `gstream` everything that only feeds a
        # parameter gradient, `hstream` the action head.  VLA_TRAINER_STREAMS=1: everything in line, 2: no separate head stream.
        nstreams = int(os.environ.get("VLA_TRAINER_STREAMS", "3"))
        self.gstream = torch.cuda.Stream() if nstreams > 1 else None
        self.hstream = torch.cuda.Stream() if nstreams > 2 else None
        # two vision backbones (DINOv2 + SigLIP: the documented recipe, BASELINE configs[4]) are independent of each other in both directions: the
        # second one's forward and backward run on a stream of their own beside the first's ("V" segments; frozen forwards of 32 images each:
        # 15.0 ms one after the other, 13.2 ms side by side).  VLA_SERIAL_BACKBONES=1: one after the other on the chain, as before.
        self.vstream = torch.cuda.Stream() if nstreams > 2 and len(self.vits) == 2 and not os.environ.get("VLA_SERIAL_BACKBONES") else None
        self.group_tn = not os.environ.get("VLA_NO_GROUPED_TN")          # (A/B knob)
        self._deferred = []
        self._refreshed, self._rgraphs = set(), {}      # derived operands rebuilt behind a range's AdamW in this step; their graphs (captured step)

    # ---- mode hooks ---------------------------------------------------------------------------------------------
    def _lin(self, key, x, W, bias=None, **kw):
        raise NotImplementedError

    def _lin_bwd(self, key, dy, x, WT, out=None, swiglu_gu=None):
        raise NotImplementedError

    trains_vectors = False       # norms / biases / LayerScale / pos-embed / prefix tokens / patch embedding / token embedding

    def A(self, name):
        return None              # fp32 accumulator of a vector parameter (full mode)

    # ---- buffers ------------------------------------------------------------------------------------------------
    def _alloc(self, B: int, S: int):
        if self._key == (B, S):
            return
        cfg, dev = self.cfg, self.dev
        e = lambda *s, dt=BF16: torch.empty(*s, device=dev, dtype=dt)
        self.V = []
        Bv = B * cfg.n_img
        for v in self.vits:
            vc = v.cfg
            nb, d, T = len(v.blocks), vc.d, vc.n_patches + vc.n_prefix
            Mv = Bv * T
            st = dict(X=e(nb + 1, Mv, d), H1=e(nb, Mv, d), H2=e(nb, Mv, d), A=e(nb, Mv, d), Xm=e(nb, Mv, d),
                      S1=e(nb, Mv, 2, dt=torch.float32), S2=e(nb, Mv, 2, dt=torch.float32), QKV=e(nb, Mv, 3 * d),
                      LSE=e(nb, Bv, vc.heads, T, dt=torch.float32), Mpre=e(nb, Mv, v.mlp_pad), Mact=e(nb, Mv, v.mlp_pad),
                      g_d=e(Mv, d), dxa=e(Mv, d),
                      # dY of every Linear, kept PER BLOCK: the weight-gradient products read them on the second stream, any time
                      # after the dX chain has moved on (2.2 GB at batch 16 for SigLIP: nothing on a 288 GB part)
                      G_fc2=e(nb, Mv, d), G_pre=e(nb, Mv, v.mlp_pad), G_proj=e(nb, Mv, d), G_qkv=e(nb, Mv, 3 * d))
            if vc.layerscale:                        # pre-scale outputs of proj / fc2: the LayerScale gradient needs them; the
                st["PA"], st["PM"] = e(nb, Mv, d), e(nb, Mv, d)      # residual-stream gradients get slots of their own (G_* hold the scaled dY)
                st["DX"], st["DXM"] = e(nb, Mv, d), e(nb, Mv, d)
            if vc.n_prefix:
                st["pe"] = e(Bv * vc.n_patches, d)
            self.V.append(st)
        c = cfg.llm
        n, D, I = c.n_layers, c.d, c.inter
        M = B * S
        self.N1, self.N2, self.Hs = e(n, M, D), e(n, M, D), e(n, M, I)
        W_ = (c.heads + 2 * c.kv_heads) * c.dh
        # per-layer dY of the four Linears (residual-stream gradient = dY of down_proj, d1 = dY of o_proj): 3.4 GB at batch 16
        self.G_res, self.G_d1, self.G_gu, self.G_qkv = e(n, M, D), e(n, M, D), e(n, M, 2 * I), e(n, M, W_)
        self.d_last = e(M, D)
        if self.trains_vectors:       # dY of the two RMSNorms per layer: their weight gradients run on the gradient stream too (0.5 GB)
            self.G_n1, self.G_n2 = e(n, M, D), e(n, M, D)
        self.dfeats = e(B * cfg.n_patches, cfg.vis_dim)
        self.dp = e(B * cfg.n_patches, D)
        self.pj = {}
        self._key = (B, S)

    # ---- forward ------------------------------------------------------------------------------------------------
    def _ln(self, x, w, b, y, st, eps):
        d = x.shape[1]
        ops.N.check(ops._lib().vla_layernorm_fwd(ops._st(), ops._p(x), ops._p(w), ops._p(b), ops._p(y), ops._p(st), x.shape[0], d, d, d, eps), "layernorm_fwd")

    def _rms(self, x, w, out, rstd):
        self.llm._rms(x, w, out, rstd)

    def _ln_bwd(self, dy, x, w, st, dx, dw, db):
        d = x.shape[1]
        ops.N.check(ops._lib().vla_layernorm_bwd(ops._st(), ops._p(dy), ops._p(x), ops._p(w), ops._p(st), ops._p(dx), ops._p(dw), ops._p(db),
                                                 x.shape[0], d, d, d, d), "layernorm_bwd")

    def _stacked_pixels(self, j: int, px: torch.Tensor):
        """Images of backbone j stacked along the batch: ([n_img * B, C, H, W] tensor, first channel) - engine._vision_backbone."""
        cfg, nbk = self.cfg, len(self.cfg.vit)
        if cfg.n_img == 1:
            return px, 3 * j
        return torch.cat([px[:, 3 * (im * nbk + j):3 * (im * nbk + j) + 3] for im in range(cfg.n_img)], 0).contiguous(), 0

    def _vit_forward(self, j: int, px: torch.Tensor):
        v, st, cfg = self.vits[j], self.V[j], self.cfg
        vc = v.cfg
        stacked, c0 = self._stacked_pixels(j, px)
        Bv, Np, T, d = stacked.shape[0], vc.n_patches, vc.n_patches + vc.n_prefix, vc.d
        X = st["X"]
        st["cols"] = ops.im2col_patch(stacked, c0, vc.patch, v.kpe)
        if vc.n_prefix:
            ops.gemm_nt(st["cols"], v.wpe, bias=v.bpe, residual=v.pos, res_mod=Np, out=st["pe"])
            x3 = X[0].view(Bv, T, d)
            ops.copy_rows3d(st["pe"], x3[0, vc.n_prefix:], Bv, Np, d, Np * d, d, T * d, d)
            ops.copy_rows3d(v.prefix, x3, Bv, vc.n_prefix, d, 0, d, T * d, d)               # cls + register tokens, broadcast over the batch
        else:
            ops.gemm_nt(st["cols"], v.wpe, bias=v.bpe, residual=v.pos, res_mod=Np, out=X[0])
        dh = d // vc.heads
        for i, b in enumerate(v.blocks):
            x, k = X[i], f"vit{j}.{i}."
            self._ln(x, b["n1w"], b["n1b"], st["H1"][i], st["S1"][i], vc.eps)
            qkv = self._lin(k + "qkv", st["H1"][i], b["wqkv"], b["bqkv"], out=st["QKV"][i]).view(Bv, T, 3 * d)
            dsc = ops._attn_desc(qkv[:, :, :d], qkv[:, :, d:2 * d], qkv[:, :, 2 * d:], st["A"][i].view(Bv, T, d), st["LSE"][i], None, False,
                                 dh ** -0.5, vc.heads, vc.heads, dh)
            ops.N.check(ops._lib().vla_attn_fwd(ops._st(), C.byref(dsc)), "attn_fwd")
            if vc.layerscale:
                self._lin(k + "proj", st["A"][i], b["wproj"], b["bproj"], out=st["PA"][i])
                ops.layerscale_fwd(st["PA"][i], b["ls1"], x, out=st["Xm"][i])
            else:
                self._lin(k + "proj", st["A"][i], b["wproj"], b["bproj"], out=st["Xm"][i], residual=x)
            xm = st["Xm"][i]
            self._ln(xm, b["n2w"], b["n2b"], st["H2"][i], st["S2"][i], vc.eps)
            self._lin(k + "fc1", st["H2"][i], b["w1"], b["b1"], out=st["Mpre"][i])          # pre-activation kept for the GELU backward
            ops.N.check(ops._lib().vla_gelu_fwd(ops._st(), ops._p(st["Mpre"][i]), ops._p(st["Mact"][i]), st["Mpre"][i].numel()), "gelu_fwd")
            if vc.layerscale:
                self._lin(k + "fc2", st["Mact"][i], b["w2"], b["b2"], out=st["PM"][i])
                ops.layerscale_fwd(st["PM"][i], b["ls2"], xm, out=X[i + 1])
            else:
                self._lin(k + "fc2", st["Mact"][i], b["w2"], b["b2"], out=X[i + 1], residual=xm)
        # patch features (prefix tokens dropped, no final norm) -> this backbone's column block of the fused feature buffer
        feats, B, npi = self.eng.feats, self.eng.B, vc.n_patches
        vis, col = cfg.vis_dim, sum(u.cfg.d for u in self.vits[:j])
        out3 = X[len(v.blocks)].view(Bv, T, d)
        for im in range(cfg.n_img):
            ops.copy_rows3d(out3[im * B, vc.n_prefix:], feats[0, im * npi:, col:], B, npi, d, T * d, d, feats.shape[1] * vis, vis)

    def _proj_forward(self):
        eng, cfg, pj = self.eng, self.cfg, self.eng.proj
        feats = eng.feats.view(-1, cfg.vis_dim)
        P = self.pj
        P["in"] = feats
        P["pre1"] = self._lin("proj.fc1", feats, pj["fc1.weight"], pj["fc1.bias"])
        P["act1"] = ops.gelu_fwd(P["pre1"])
        dst = eng.patches.view(-1, cfg.llm.d)
        if cfg.fused:
            P["pre2"] = self._lin("proj.fc2", P["act1"], pj["fc2.weight"], pj["fc2.bias"])
            P["act2"] = ops.gelu_fwd(P["pre2"])
            self._lin("proj.fc3", P["act2"], pj["fc3.weight"], pj["fc3.bias"], out=dst)
        else:
            self._lin("proj.fc2", P["act1"], pj["fc2.weight"], pj["fc2.bias"], out=dst)

    def _llm_fwd_layers(self, lo: int, hi: int):
        llm, c = self.llm, self.cfg.llm
        B, S = self.eng.B, self.eng.S
        D, H, KV, dh = c.d, c.heads, c.kv_heads, c.dh
        for i in range(lo, hi):
            L = llm.layers[i]
            x, k = llm.HS[i].view(-1, D), f"llm.{i}."
            self._rms(x, L["n1"], self.N1[i], llm.R1[i])
            qkv = llm.QKV[i]
            if dh in (64, 128):   # RoPE in the projection's epilogue (the LoRA delta is already inside the accumulator)
                self._lin(k + "qkv", self.N1[i], L["wqkv"], L["bqkv"], out=qkv, rope=(1, llm.cos, llm.sin, S, dh, (H + KV) * dh))
            else:
                self._lin(k + "qkv", self.N1[i], L["wqkv"], L["bqkv"], out=qkv)
                ops.rope_half_(qkv[:, :H * dh], llm.cos, llm.sin, S, H, dh)
                ops.rope_half_(qkv[:, H * dh:(H + KV) * dh], llm.cos, llm.sin, S, KV, dh)
            llm._attn_fwd(qkv.view(B, S, -1), i, 0, B, S)
            x1 = llm.X1[i]
            self._lin(k + "o", llm.AO[i], L["wo"], None, out=x1, residual=x)
            self._rms(x1, L["n2"], self.N2[i], llm.R2[i])
            self._lin(k + "gu", self.N2[i], L["wgu"], None, act=ACT_SWIGLU, out=llm.GU[i], out2=self.Hs[i])
            self._lin(k + "down", self.Hs[i], L["wd"], None, out=llm.HS[llm.out_slot(i)].view(-1, D), residual=x1)
        if hi == c.n_layers:
            llm.fwd_final()

    # ---- backward pieces (stateless between calls: every gradient that crosses a piece boundary lives in a per-layer slot) -----
    def _llm_bwd_layers(self, lo: int, hi: int):
        """dX chain through layers hi-1 .. lo.  Everything that only feeds a parameter gradient (dW / LoRA-pair TN products, bias
        column sums) is handed to _defer / _defer_tn and runs on the gradient stream."""
        llm, c, B, S = self.llm, self.cfg.llm, self.eng.B, self.eng.S
        n, D, H, KV, dh, I = c.n_layers, c.d, c.heads, c.kv_heads, c.dh, c.inter
        M = B * S
        lib, st, p = ops._lib(), ops._st, ops._p
        tv, dHS = self.trains_vectors, self._dHS
        if hi == n:                                        # backward of the final norm into the top layer's slot
            ops.rmsnorm_bwd(dHS[n].view(M, D), llm.HS[n + 1].view(M, D), llm.norm, llm.RF, out=self.G_res[n - 1])
            if tv:
                ops.N.check(lib.vla_rmsnorm_dw(st(), p(dHS[n].view(M, D)), p(llm.HS[n + 1].view(M, D)), p(llm.RF), p(self.A("llm.norm")), M, D), "rmsnorm_dw")
        for i in range(hi - 1, lo - 1, -1):
            L, k = llm.layers[i], f"llm.{i}."
            d = self.G_res[i]
            if i == self.n_active - 1 and i < n - 1:       # top ACTIVE layer below dead ones: its output gradient is the head's alone
                ops.copy2d(dHS[i + 1].view(M, D), d, M, D, D, D)
            elif i < n - 1:
                ops.add_(d, dHS[i + 1].view(M, D))
            tap = self.taps.get(("llm", i)) if self.taps is not None else None
            if tap is not None:
                tap["d_out"] = d.clone()
            d_gu = self._lin_bwd(k + "down", d, self.Hs[i], L["wdT"], out=self.G_gu[i], swiglu_gu=llm.GU[i])
            d_n = self._lin_bwd(k + "gu", d_gu, self.N2[i], L["wguT"], out=self.G_n2[i] if tv else llm.d_n[:M])
            if tv:
                self._defer(lambda dy=d_n, x=llm.X1[i], r=llm.R2[i], acc=self.A(k + "n2"):
                            ops.N.check(ops._lib().vla_rmsnorm_dw(ops._st(), ops._p(dy), ops._p(x), ops._p(r), ops._p(acc), M, D), "rmsnorm_dw"))
            d1 = ops.rmsnorm_bwd(d_n, llm.X1[i], L["n2"], llm.R2[i], dres=d, out=self.G_d1[i])
            dao = self._lin_bwd(k + "o", d1, llm.AO[i], L["woT"], out=llm.d_n[:M])
            q, kk, v = llm._attn_views(llm.QKV[i].view(B, S, -1))
            d_qkv = self.G_qkv[i]
            dq, dk, dv = llm._attn_views(d_qkv.view(B, S, -1))
            ops.attn_bwd(dao.view(B, S, -1), q, kk, v, llm.AO[i].view(B, S, -1), llm.LSE[i], H, KV, dh, True, llm.kmask, dq=dq, dk=dk, dv=dv,
                         rope=(llm.cos, llm.sin) if dh in (64, 128) else None)
            if dh not in (64, 128):
                ops.rope_half_(d_qkv[:, :H * dh], llm.cos, llm.sin, S, H, dh, sign=-1)
                ops.rope_half_(d_qkv[:, H * dh:(H + KV) * dh], llm.cos, llm.sin, S, KV, dh, sign=-1)
            d_n = self._lin_bwd(k + "qkv", d_qkv, self.N1[i], L["wqkvT"], out=self.G_n1[i] if tv else llm.d_n[:M])
            if tv:
                self._defer(lambda dy=d_qkv, acc=self.A(k + "bqkv"): ops.colsum_(dy, acc))
                self._defer(lambda dy=d_n, x=llm.HS[i].view(M, D), r=llm.R1[i], acc=self.A(k + "n1"):
                            ops.N.check(ops._lib().vla_rmsnorm_dw(ops._st(), ops._p(dy), ops._p(x), ops._p(r), ops._p(acc), M, D), "rmsnorm_dw"))
            d = ops.rmsnorm_bwd(d_n, llm.HS[i].view(M, D), L["n1"], llm.R1[i], dres=d1, out=self.G_res[i - 1] if i > 0 else self.d_last)
            if tap is not None:
                tap["d_in"] = d.clone()

    def _mid_backward(self):
        """Between the LLM and the vision backward: action-query gradient, token-embedding gradient, projector."""
        eng, head = self.eng, self.head
        dX0 = self.d_last.view(eng.B, eng.S, self.cfg.llm.d)          # gradient w.r.t. inputs_embeds
        dq = ops.action_query_grad(dX0, eng.pos0, eng.Np, 0)
        ops.cast_f32_bf16(dq, out=head.P.g("action_queries"))
        self._embed_backward(dX0)
        self._proj_backward(dX0)                                      # -> self.dfeats

    def _proj_backward(self, dX0):
        eng, cfg = self.eng, self.cfg
        B, S, Np, D = eng.B, eng.S, eng.Np, cfg.llm.d
        P, pjT = self.pj, self.projT
        dp = self.dp
        ops.copy_rows3d(dX0[0, 1], dp, B, Np, D, S * D, D, Np * D, D)          # rows 1..Np of every sequence: the projected patches
        tv = self.trains_vectors
        cs = lambda dy, name: self._defer(lambda: ops.colsum_(dy, self.A(name)))     # (every dY below lives until the next step)
        if cfg.fused:
            if tv:
                cs(dp, "proj.fc3.bias")
            dh2 = self._lin_bwd("proj.fc3", dp, P["act2"], pjT["fc3.weight"])
            dpre = P["dpre2"] = ops.gelu_bwd(dh2, P["pre2"])
            if tv:
                cs(dpre, "proj.fc2.bias")
            dh1 = self._lin_bwd("proj.fc2", dpre, P["act1"], pjT["fc2.weight"])
        else:
            if tv:
                cs(dp, "proj.fc2.bias")
            dh1 = self._lin_bwd("proj.fc2", dp, P["act1"], pjT["fc2.weight"])
        dpre1 = P["dpre1"] = ops.gelu_bwd(dh1, P["pre1"])
        if tv:
            cs(dpre1, "proj.fc1.bias")
        return self._lin_bwd("proj.fc1", dpre1, P["in"], pjT["fc1.weight"], out=self.dfeats)       # d features [B*Np, vis_dim]

    def _vit_grad_slots(self, j: int, i: int):
        """(gradient w.r.t. block i's output, w.r.t. its x_mid) of backbone j.  Without LayerScale these residual-stream gradients
        ARE the dY of fc2 / proj, so the chain walks the per-block dY slots; with it the scaled dY have slots of their own."""
        st = self.V[j]
        return (st["G_fc2"][i], st["G_proj"][i]) if not self.vits[j].cfg.layerscale else (st["DX"][i], st["DXM"][i])

    def _vit_bwd_begin(self, j: int):
        """Gradient w.r.t. the last useful block's output: this backbone's column block of d feats on the patch rows, zero on the prefix."""
        v, cfg = self.vits[j], self.cfg
        vc = v.cfg
        B, npi, T, d = self.eng.B, vc.n_patches, vc.n_patches + vc.n_prefix, vc.d
        Bv, nb = B * cfg.n_img, len(v.blocks)
        vis, col = cfg.vis_dim, sum(u.cfg.d for u in self.vits[:j])
        dx = self._vit_grad_slots(j, nb - 1)[0]
        if vc.n_prefix:
            ops.zero_(dx)
        dx3, df3 = dx.view(Bv, T, d), self.dfeats.view(B, cfg.n_patches, vis)
        for im in range(cfg.n_img):
            ops.copy_rows3d(df3[0, im * npi:, col:], dx3[im * B, vc.n_prefix:], B, npi, d, cfg.n_patches * vis, vis, T * d, d)

    def _vit_bwd_blocks(self, j: int, lo: int, hi: int):
        v, st, cfg = self.vits[j], self.V[j], self.cfg
        vc = v.cfg
        T, d = vc.n_patches + vc.n_prefix, vc.d
        Bv, dh = self.eng.B * cfg.n_img, d // vc.heads
        tv, ls = self.trains_vectors, vc.layerscale
        cs = lambda dy, acc: self._defer(lambda: ops.colsum_(dy, acc))
        for i in range(hi - 1, lo - 1, -1):
            b, k = v.blocks[i], f"vit{j}.{i}."
            a = (lambda n: self.A(k + n)) if tv else (lambda n: None)
            dx, dxm = self._vit_grad_slots(j, i)
            tap = self.taps.get(("vit", j, i)) if self.taps is not None else None
            if tap is not None:
                tap["d_out"] = dx.clone()
            # x_out = x_mid + ls2 * fc2(gelu(fc1(LN2(x_mid))))
            dh_ = ops.layerscale_bwd(dx, st["PM"][i] if tv else None, b["ls2"], a("ls2"), out=st["G_fc2"][i]) if ls else dx
            if tv:
                cs(dh_, a("b2"))
            dm = self._lin_bwd(k + "fc2", dh_, st["Mact"][i], b["w2T"], out=st["G_pre"][i])
            ops.N.check(ops._lib().vla_gelu_bwd(ops._st(), ops._p(dm), ops._p(st["Mpre"][i]), ops._p(dm), dm.numel()), "gelu_bwd")   # in place
            dpre = dm
            if tv:
                cs(dpre, a("b1"))
            dh2 = self._lin_bwd(k + "fc1", dpre, st["H2"][i], b["w1T"], out=st["g_d"])
            self._ln_bwd(dh2, st["Xm"][i], b["n2w"], st["S2"][i], dxm, a("n2w"), a("n2b"))
            ops.add_(dxm, dx)                            # residual
            # x_mid = x_in + ls1 * proj(attn(qkv(LN1(x_in))))
            da_ = ops.layerscale_bwd(dxm, st["PA"][i] if tv else None, b["ls1"], a("ls1"), out=st["G_proj"][i]) if ls else dxm
            if tv:
                cs(da_, a("bproj"))
            da = self._lin_bwd(k + "proj", da_, st["A"][i], b["wprojT"], out=st["g_d"])
            qkv = st["QKV"][i].view(Bv, T, 3 * d)
            g_mid = st["G_qkv"][i]
            dqkv = g_mid.view(Bv, T, 3 * d)
            ops.attn_bwd(da.view(Bv, T, d), qkv[:, :, :d], qkv[:, :, d:2 * d], qkv[:, :, 2 * d:], st["A"][i].view(Bv, T, d), st["LSE"][i],
                         vc.heads, vc.heads, dh, False, None, dq=dqkv[:, :, :d], dk=dqkv[:, :, d:2 * d], dv=dqkv[:, :, 2 * d:])
            if tv:
                cs(g_mid, a("bqkv"))
            dh1 = self._lin_bwd(k + "qkv", g_mid, st["H1"][i], b["wqkvT"], out=st["g_d"])
            if i == 0 and not tv:
                return                                   # below block 0 everything is frozen (Conv2d patch embedding, pos_embed, tokens)
            dxi = self._vit_grad_slots(j, i - 1)[0] if i > 0 else st["dxa"]
            self._ln_bwd(dh1, st["X"][i], b["n1w"], st["S1"][i], dxi, a("n1w"), a("n1b"))
            ops.add_(dxi, dxm)
            if tap is not None:
                tap["d_in"] = dxi.clone()

    def _vit_bwd_end(self, j: int):
        """Full fine-tune: patch embedding x0 = cols . Wpe^T + bpe + pos on the patch rows; cls / register tokens on the prefix rows."""
        v, st, cfg = self.vits[j], self.V[j], self.cfg
        vc = v.cfg
        npi, T, d = vc.n_patches, vc.n_patches + vc.n_prefix, vc.d
        Bv = self.eng.B * cfg.n_img
        dx = st["dxa"]
        lib, p = ops._lib(), ops._p
        if vc.n_prefix:
            dx3 = dx.view(Bv, T, d)
            dpe = st["pe"]
            ops.copy_rows3d(dx3[0, vc.n_prefix:], dpe, Bv, npi, d, T * d, d, npi * d, d)
            ops.N.check(lib.vla_colsum_bf16(ops._st(), p(dx), p(self.A(f"vit{j}.prefix")), Bv, vc.n_prefix * d, T * d, 1, 0, 0), "colsum(prefix)")
        else:
            dpe = dx
        self._defer_tn(dpe, st["cols"], self.G(f"vit{j}.wpe"))
        ops.colsum_(dpe, self.A(f"vit{j}.bpe"))
        ops.N.check(lib.vla_colsum_bf16(ops._st(), p(dpe), p(self.A(f"vit{j}.pos")), Bv, npi * d, npi * d, 1, 0, 0), "colsum(pos)")   # sum over the batch

    # ---- the step as a list of single-stream segments ---------------------------------------------------------------
    # Three streams.  "M": vision, LLM forward, the dX chain of the backward.  "H": the action head - 3 % of the FLOPs but a chain
    # of ~600 small dependent kernels: its forward trails the LLM forward chunk by chunk, its backward runs ahead of the LLM
    # backward (layer i needs d hidden_states[i + 1] from block i), its batched dW products fill the LLM backward's gaps.  "G":
    # everything that only feeds a parameter gradient (the dW / LoRA-pair TN products of a piece as ONE grouped launch, bias column
    # sums), one piece behind the chain: a dW product or a chain GEMM alone fills 270-530 tiles of a 512-slot chip, together they
    # fill each other's tail rounds.  Segments are ordered so that every event is recorded before it is waited for; eagerly a
    # segment is a Python call under its stream, captured it is its own linear hipGraph (engine.VLAEngine uses the same scheme for
    # the adapter-only step).  A segment's `ranges` are the flat gradient ranges that are FINAL when it ends: the data-parallel
    # exchange of each starts right there, underneath the rest of the backward (vla-scripts/finetune.py:215-227, 869: DDP's
    # bucketed all-reduce overlapped with backward; BASELINE configs[3] "grad-bucket overlap").
    def _segments(self, batch, noise, gscale: float = 1.0, actions=None):
        if self.objective == "token_ce":
            return self._segments_ce(batch, gscale)
        eng, cfg, llm, head = self.eng, self.cfg, self.llm, self.head
        n, nb = cfg.llm.n_layers, cfg.num_blocks
        # LLM layers above the head's last block (Qwen2.5-1.5B: 28 layers, 24 head blocks - action_heads.py:117-118 reads
        # hidden_states[1..24]) never reach the loss: the reference computes them and throws the result away; autograd hands their
        # parameters all-zero gradients, so AdamW moves them by the weight-decay factor 1 - lr wd alone, which a bf16 parameter does
        # not see (it rounds to 1 for lr wd < 2^-9).  Here they are neither computed nor touched (tests/test_full_finetune_gpu.py).
        # LLM layers per segment: `exchange_layers`, but 2, 1, 1 at the top - the forward -> backward turn-around (last head blocks, loss, first
        # head-backward blocks) is what the LLM backward waits for: with four-layer segments the M stream idled 1.3 ms there (tools/trainer_timeline.py)
        el, na = self.exchange_layers, self.n_active
        lch = E.VLAEngine._chunks(na, [el] * max(0, (na - 4) // el) + [2, 1, 1]) if na >= 8 and not os.environ.get("VLA_UNIFORM_CHUNKS") else E.VLAEngine._chunks(na, [el])
        two = self.gstream is not None
        segs = []

        def add(st, fn, wait=None, signal=None, ranges=None):
            segs.append((st, fn, wait, signal, ranges))

        def grads(after, signal, ranges):
            """The gradient work the segment signalling `after` has deferred (+ hand-over of the finished ranges)."""
            if two:
                add("G", self._flush_work, after, signal, ranges)
            else:                                   # single stream: nothing was deferred; the ranges are final where the chain stands
                add("M", None, None, signal, ranges)
            return signal

        side = self.vstream is not None                 # the second backbone on its own stream (forward and backward)

        def f_pre():
            eng._vision_begin(batch)
            self._alloc(eng.B, eng.S)
            self._begin_forward()

        def f_front():
            if not side:
                f_pre()
            for j in range(1 if side else len(self.vits)):
                self._vit_forward(j, batch["pixel_values"])

        def f_front2():
            self._proj_forward()
            self._mm = eng._embed(batch)
            self._batch = batch
            llm.fwd_begin(eng.B, eng.S, self._mm, 0)
            self._begin_backward()                  # (fp32 accumulators / sparse embedding gradient: zero before anything adds to them)
            self._dHS = eng._dhs(0)                 # zeroed hidden-state gradients: the head's backward scatters into them
        if side:
            add("M", f_pre, None, ("pre", 0))
            add("V", lambda: self._vit_forward(1, batch["pixel_values"]), ("pre", 0), ("vf", 1))
        add("M", f_front, None, None)
        add("M", f_front2, ("vf", 1) if side else None, ("front", 0))
        for c, (lo, hi) in enumerate(lch):
            add("M", lambda lo=lo, hi=hi: self._llm_fwd_layers(lo, hi), None, ("f", c))

            def h_fwd(c=c, lo=lo, hi=hi):
                if c == 0:
                    head.fwd_begin(llm.HS, eng.pos1, batch["proprio"], eng.Np, noise)
                    head.refresh_transposes()        # (W^T operands of the head's backward: beside the LLM forward, not on the turn-around)
                for i in range(lo, min(hi, nb)):
                    head.fwd_layer(i)
                if c == len(lch) - 1:
                    self._pred = head.fwd_end()
            add("H", h_fwd, ("f", c), None)
        self._n_forward = len(segs)                 # forward() = these; backward() = the rest

        def h_loss():
            tgt = eng._to_bf16(batch["actions"] if actions is None else actions)
            self._loss3, dpred = ops.l1_loss(self._pred, tgt, True, gscale)
            head.prep_backward(eng.pos1, eng.Np, eng.B, eng.S, 0)
            head.bwd_begin(dpred, 0)
        add("H", h_loss, None, None)
        aq_off = head.P.offsets["action_queries"][0]
        gsig = []
        for k, (lo, hi) in enumerate(reversed(lch)):
            wait = None
            if min(hi, nb) > lo:                    # (layers above the head's last block receive no gradient from it)
                add("H", lambda lo=lo, hi=hi: [head.bwd_layer(i, self._dHS) for i in range(min(hi, nb) - 1, lo - 1, -1)], None, ("b", k))
                wait = ("b", k)
            add("M", lambda lo=lo, hi=hi: self._llm_bwd_layers(lo, hi), wait, ("m", k))
            gsig.append(grads(("m", k), ("g", k), self._ranges("llm", lo, hi - 1)))
        add("H", head.bwd_end, None, ("hend", 0), [(head.P.grad, 0, aq_off)])
        add("M", self._mid_backward, None, ("mid", 0))
        gsig.append(grads(("mid", 0), ("gmid", 0), [(head.P.grad, aq_off, head.P.numel)] + self._ranges("embed") + self._ranges("proj")))
        vsegs = []
        for j, v in enumerate(self.vits):
            vch = self._vit_chunks(j)
            for q, (lo, hi) in enumerate(reversed(vch)):
                def v_bwd(j=j, lo=lo, hi=hi, first=(q == 0)):
                    if first:
                        self._vit_bwd_begin(j)
                    self._vit_bwd_blocks(j, lo, hi)
                    if lo == 0 and self.trains_vectors:
                        self._vit_bwd_end(j)
                vsegs.append((j, q, lo, hi, v_bwd))
        if side:                                    # the two backbones' segments alternate in the list (each followed by its gradient work): both
            a, b = [x for x in vsegs if x[0] == 0], [x for x in vsegs if x[0] == 1]     # chains start right behind the projector's backward
            vsegs = [x for pair in zip(a, b) for x in pair] + a[len(b):] + b[len(a):]
        for j, q, lo, hi, v_bwd in vsegs:
            on_side = side and j == 1
            add("V" if on_side else "M", v_bwd, ("mid", 0) if on_side and q == 0 else None, ("v", j, q))
            gsig.append(grads(("v", j, q), ("gv", j, q), self._ranges("vit", lo, hi - 1, j)))
        if self.trains_vectors:                     # the tail casts the ONE fp32 buffer every piece's bias / norm sums met in
            add("M", self._end_backward, [sg for sg in gsig if two], ("end", 0), self._ranges("tail"))
        return segs

    # ---- token cross-entropy objective: the native trainer of prismatic/training/strategies/base_strategy.py:257-417 ----
    # PrismaticVLM.forward(..., labels=) -> HF shifted causal-LM loss (prismatic/models/vlms/prismatic.py:312-481), loss.backward().
    # No action queries, no action head: every LLM layer and the final norm are live, lm_head = the tied embedding table.  Only the
    # rows that can carry a label are projected: the L - 1 text rows of every sequence (the patch rows' targets are IGNORE_INDEX by
    # construction, :411-422) - 1520 of 5632 rows at batch 16 - so the three vocabulary-wide products (logits, d hidden, d lm_head)
    # cost 27 % of their full-sequence form.  softmax - onehot is formed in place on the logits (vla_token_ce_bwd).
    def _segments_ce(self, batch, gscale: float = 1.0):
        eng, cfg, llm = self.eng, self.cfg, self.llm
        n, D, V = cfg.llm.n_layers, cfg.llm.d, cfg.llm.vocab
        assert getattr(llm, "lm_head", None) is None, "token-CE training: tied lm_head only (Qwen2.5-0.5B / 1.5B tie_word_embeddings)"
        lch = E.VLAEngine._chunks(n, [self.exchange_layers])
        two = self.gstream is not None
        segs = []

        def add(st, fn, wait=None, signal=None, ranges=None):
            segs.append((st, fn, wait, signal, ranges))

        def grads(after, signal, ranges):
            if two:
                add("G", self._flush_work, after, signal, ranges)
            else:
                add("M", None, None, signal, ranges)
            return signal

        def f_front():
            eng._vision_begin(batch)
            self._alloc(eng.B, eng.S)
            self._begin_forward()
            for j in range(len(self.vits)):
                self._vit_forward(j, batch["pixel_values"])
            self._proj_forward()
            self._mm = eng._embed(batch, action_queries=False)
            self._batch = batch
            llm.fwd_begin(eng.B, eng.S, self._mm, 0)
            self._begin_backward()
            self._dHS = eng._dhs(0)
        add("M", f_front, None, ("front", 0))
        for c, (lo, hi) in enumerate(lch):
            add("M", lambda lo=lo, hi=hi: self._llm_fwd_layers(lo, hi), None, ("f", c))
        self._n_forward = len(segs)

        def f_ce():
            B, S, Np = eng.B, eng.S, eng.Np
            Lm = S - Np - 1                                          # text rows that predict a token: sequence rows Np .. S - 2
            if getattr(self, "_ce_key", None) != (B, S):
                e = lambda *sh, dt=BF16: torch.empty(*sh, device=self.dev, dtype=dt)
                self.ce_h, self.ce_dh, self.ce_logits = e(B * Lm, D), e(B * Lm, D), e(B * Lm, V)
                self.ce_out = torch.zeros(2, device=self.dev, dtype=torch.float32)
                self._ce_key = (B, S)
            ops.copy_rows3d(llm.HS[n][0, Np], self.ce_h, B, Lm, D, S * D, D, Lm * D, D)
            ops.gemm_nt(self.ce_h, llm.embed, out=self.ce_logits, split_k=0)
            tgt = self._ce_tgt = batch["labels"][:, 1:].contiguous().view(-1)
            ops.zero_(self.ce_out)
            lib, st, p = ops._lib(), ops._st, ops._p
            ops.N.check(lib.vla_token_ce(st(), p(self.ce_logits), V, p(tgt), B * Lm, V, p(self.ce_out)), "token_ce")
            loss = self.ce_out[0:1] / self.ce_out[1:2]
            self._loss3 = torch.cat([loss, loss, loss])              # (same three-slot shape the L1 path logs)
            ops.N.check(lib.vla_token_ce_bwd(st(), p(self.ce_logits), V, p(tgt), B * Lm, V, p(self.ce_out), gscale, p(self.ce_logits), V), "token_ce_bwd")
            ops.gemm_nt(self.ce_logits, self.lmT, out=self.ce_dh, split_k=0)                      # d hidden = dlogits . W_lm
            if self.trains_vectors:                                  # d lm_head = dlogits^T . hidden, added to the tied table's gradient in the tail
                self._defer_tn(self.ce_logits, self.ce_h, self.g_lm)
            ops.copy_rows3d(self.ce_dh, self._dHS[n][0, Np], B, Lm, D, Lm * D, D, S * D, D)
        add("M", f_ce, None, ("ce", 0))
        gsig = [grads(("ce", 0), ("gce", 0), [])] if self.trains_vectors else []       # (LoRA: lm_head is not adapted - nothing deferred here)
        for k, (lo, hi) in enumerate(reversed(lch)):
            add("M", lambda lo=lo, hi=hi: self._llm_bwd_layers(lo, hi), None, ("m", k))
            gsig.append(grads(("m", k), ("g", k), self._ranges("llm", lo, hi - 1)))

        def f_mid():
            dX0 = self.d_last.view(eng.B, eng.S, D)
            self._embed_backward(dX0)
            self._proj_backward(dX0)
        add("M", f_mid, None, ("mid", 0))
        gsig.append(grads(("mid", 0), ("gmid", 0), self._ranges("proj")))
        for j, v in enumerate(self.vits):
            vch = self._vit_chunks(j)
            for q, (lo, hi) in enumerate(reversed(vch)):
                def v_bwd(j=j, lo=lo, hi=hi, first=(q == 0)):
                    if first:
                        self._vit_bwd_begin(j)
                    self._vit_bwd_blocks(j, lo, hi)
                    if lo == 0 and self.trains_vectors:
                        self._vit_bwd_end(j)
                add("M", v_bwd, None, ("v", j, q))
                gsig.append(grads(("v", j, q), ("gv", j, q), self._ranges("vit", lo, hi - 1, j)))
        if self.trains_vectors:
            def f_tail():
                ops.add_(self.G("llm.embed"), self.g_lm)             # tied weights: lookup gradient + lm_head gradient (one bf16 add, as autograd accumulates)
                self._end_backward()
            add("M", f_tail, [sg for sg in gsig if two], ("end", 0), self._ranges("embed") + self._ranges("tail"))
        return segs

    def set_objective(self, objective: str):
        """"l1" (default: action head + L1 regression, vla-scripts/finetune.py) or "token_ce" (the native VLM / VLA trainer's token
        cross-entropy, base_strategy.py:257-417).  Before capture()."""
        assert objective in ("l1", "token_ce") and getattr(self, "_graphs", None) is None
        self.objective = objective
        if objective == "token_ce":
            self.n_active = self.cfg.llm.n_layers                    # every layer and the final norm reach this loss
            V, D = self.llm.embed.shape
            self.lmT = torch.empty(D, V, device=self.dev, dtype=BF16)
            self.g_lm = torch.zeros(V, D, device=self.dev, dtype=BF16) if self.trains_vectors else None
            self._refresh_objective()

    def _refresh_objective(self):
        if self.objective == "token_ce":
            ops.transpose(self.llm.embed, out=self.lmT)              # W_lm^T operand of d hidden = dlogits . W_lm (the tied table moves every step)

    def _flush_work(self):
        work, self._deferred = self._deferred, []
        self._run_work(work)

    def _stream(self, name: str, main):
        return main if name == "M" else ((self.hstream or main) if name == "H" else (self.vstream or main) if name == "V" else (self.gstream or main))

    def _run(self, segs, graphs=None, exchange: bool = True, update=None):
        """Enqueue the segments in order (eagerly, or as replays of their captured graphs); events cross the streams; a segment's
        finished gradient ranges go to the exchange behind an event of their own.  The caller's stream joins the others at the end.
        update=(lr, betas, eps, wd): AdamW runs RANGE BY RANGE on the gradient stream as soon as a range is final (and exchanged) -
        the backward reads only the W^T copies of a weight, never the parameter itself, so the 15 GB of optimiser traffic of a full
        fine-tune hides under the rest of the backward instead of trailing it (torch's optimizer.step() after loss.backward(),
        vla-scripts/finetune.py:1078-1082: same arithmetic, every use of a parameter in the NEXT forward sees the updated value)."""
        main = torch.cuda.current_stream()
        for st in (self.hstream, self.gstream, self.vstream):
            if st is not None:
                st.wait_stream(main)                 # fork: inputs / the previous update are ordered before everything
        ev = {}
        for k, (st, fn, wait, signal, ranges) in enumerate(segs):
            stream = self._stream(st, main)
            with torch.cuda.stream(stream):
                for w in ([] if wait is None else wait if isinstance(wait, list) else [wait]):
                    stream.wait_event(ev[w])
                tl = getattr(self, "_timeline", None)           # (tools/trainer_timeline.py: timing events around every segment)
                if tl is not None and fn is not None:
                    t0 = torch.cuda.Event(enable_timing=True)
                    t0.record(stream)
                if fn is not None:
                    if graphs is None:
                        fn()
                    elif graphs[k] is not None:
                        graphs[k].replay()
                if tl is not None and fn is not None:
                    t1 = torch.cuda.Event(enable_timing=True)
                    t1.record(stream)
                    tl.append((st, k, t0, t1))
                if signal is not None or ranges:
                    e = torch.cuda.Event()
                    e.record(stream)
                    if signal is not None:
                        ev[signal] = e
                    if ranges and exchange:
                        self._exchange(ranges, after_event=e)
                    if ranges and update is not None:
                        self._update_ranges(ranges, e, update, k if graphs is not None else None)
        for st in (self.hstream, self.gstream, self.vstream):
            if st is not None:
                main.wait_stream(st)

    def _update_ranges(self, ranges, final_event, update, seg_index=None):
        """AdamW over the ranges a segment has finished, then - round 4 - the derived operands of exactly those parameters (W^T copies of
        a full fine-tune, A_cat^T / B_blk / B_blk^T of LoRA pairs): the layer's own backward, their only reader in this step, is over
        when its range is final, so the rebuild (200-350 launches, 0.8-1.5 ms at the end of the step before) hides under the rest of the
        backward like the update itself.  seg_index: captured step - the range's rebuild is a small graph of its own."""
        lr, beta1, beta2, eps, wd = update
        red = self.eng.reducer
        st = self.gstream or torch.cuda.current_stream()
        with torch.cuda.stream(st):
            st.wait_event(final_event)
            gscale = 1.0
            if red is not None:                      # the range's collectives were just queued on the exchange stream: wait for them
                st.wait_stream(red.stream)
                gscale = red.grad_scale
            for buf, lo, hi in ranges:
                P = self.P if buf is self.P.grad else self.head.P
                if hi > lo:
                    ops.adamw_(P.data[lo:hi], P.grad[lo:hi], P.m[lo:hi], P.v[lo:hi], self.step_count, lr, beta1, beta2, eps, wd, gscale=gscale)
            if seg_index is None:
                for fn in self._refresh_pieces_in(ranges):
                    fn()
            elif seg_index in self._rgraphs:
                self._rgraphs[seg_index].replay()

    # ---- derived operands, piecewise
    def _refresh_pieces(self):
        """[(lo, hi, fn)]: fn rebuilds the operands derived from the parameters at flat offsets [lo, hi) of self.P (none: refresh() does all)."""
        return []

    def _refresh_pieces_in(self, ranges, mark: bool = True):
        spans = [(lo, hi) for buf, lo, hi in ranges if buf is self.P.grad and hi > lo]
        out = []
        for i, (lo, hi, fn) in enumerate(self._refresh_pieces()):
            if any(a <= lo and hi <= b for a, b in spans):
                out.append(fn)
                if mark:
                    self._refreshed.add(i)
        return out

    def _refresh_rest(self):
        """What no finished range covered in this step (eager path): the remaining pieces, or everything when the mode has no pieces."""
        pieces = self._refresh_pieces()
        if not pieces:
            return self.refresh()
        for i, (_, _, fn) in enumerate(pieces):
            if i not in self._refreshed:
                fn()
        self._refresh_extra()
        self._refreshed = set()

    def _refresh_extra(self):
        pass                     # (derived operands that belong to no parameter range: the token-CE objective's W_lm^T)

    def _run_inline(self, segs):
        """The same pieces one after the other on the current stream, gradient work in line (forward() / backward())."""
        g, h, v, self.gstream, self.hstream, self.vstream = self.gstream, self.hstream, self.vstream, None, None, None
        try:
            for _, fn, _, _, _ in segs:
                if fn is not None:
                    fn()
        finally:
            self.gstream, self.hstream, self.vstream = g, h, v

    def forward(self, batch: Dict[str, torch.Tensor], noise: Optional[torch.Tensor] = None):
        """Training forward on the current stream (keeps what backward() needs) -> predicted actions [B, chunk, action_dim]."""
        segs = self._segments(batch, noise)
        self._run_inline(segs[:self._n_forward])
        self._fwd_args = (batch, noise)
        return self._pred

    def backward(self, pred, actions, gscale: float = 1.0):
        """Whole backward of the last forward() on the current stream (tests, eager debugging); train_step runs the same pieces on
        three streams."""
        batch, noise = self._fwd_args
        segs = self._segments(batch, noise, gscale, actions=actions)
        self._run_inline(segs[self._n_forward:])
        return self._loss3

    def _begin_backward(self):
        pass

    def _begin_forward(self):
        pass                     # (first launch of a step: LoRA with dropout bumps its mask counter here)

    def _vit_chunks(self, j: int):
        """Blocks per backward segment of backbone j: `exchange_blocks`, but 1, 2, 4 at the BOTTOM of the last backbone - the blocks the
        backward reaches last: the weight gradients, update and operand rebuild of the final segment trail the chain with nothing beside
        them (1.8 ms behind a seven-block segment in the full fine-tune, tools/trainer_timeline.py)."""
        nbj = len(self.vits[j].blocks)
        last = j == len(self.vits) - 1 and nbj >= 14 and not os.environ.get("VLA_UNIFORM_CHUNKS")
        return E.VLAEngine._chunks(nbj, [1, 2, 4, self.exchange_blocks] if last else [self.exchange_blocks])

    def _end_backward(self):
        pass

    def _embed_backward(self, dX0):
        pass

    def _ranges(self, kind, lo=0, hi=0, j=0):
        """[(flat gradient buffer, first element, end element)] of a parameter group (mode-specific)."""
        return []

    def _span(self, first: str, last: str):
        a = self.P.offsets[first][0]
        off, shape = self.P.offsets[last]
        return (self.P.grad, a, off + rup(math.prod(shape), 8))

    def _defer(self, fn):
        """Gradient-only work (reads a persistent dY / activation, writes a parameter gradient): run now, or - two streams - at
        the end of the current piece on the gradient stream."""
        if self.gstream is None:
            fn()
        else:
            self._deferred.append(fn)

    def _defer_tn(self, a, b, out, alpha: float = 1.0, a_cols=None):
        """A weight-gradient product out = alpha a^T b.  Two streams: collected, and all products of a piece go out as ONE grouped
        launch over their common tile list (vla_gemm_bf16_tn_grouped) - alone a dW product leaves up to half the chip idle in its
        tail round."""
        if self.gstream is None or not self.group_tn:
            self._defer(lambda: ops.gemm_tn(a, b, out=out, alpha=alpha, a_cols=a_cols))
            return
        if not self._deferred or not isinstance(self._deferred[0], list):
            self._deferred.insert(0, [])             # ONE problem list per piece (the products are independent of the other work)
        self._deferred[0].append(ops.tn_problem(a, b, out, alpha, a_cols))

    @staticmethod
    def _run_work(work):
        for w in work:
            if isinstance(w, list):
                ops.gemm_tn_grouped(w)
            else:
                w()

    def _exchange(self, ranges, after_event=None):
        red = self.eng.reducer
        if red is not None:
            for buf, lo, hi in ranges:
                if hi > lo:
                    red.reduce_async(buf, lo, hi, after_event=after_event)

    # ---- update / capture ----------------------------------------------------------------------------------------------
    def set_grad_accumulation(self, n: int):
        """vla-scripts/finetune.py:1039-1042, 1078-1082: loss / n on every micro-batch, gradients summed over n micro-batches (in
        bf16, as autograd accumulates ``.grad``), one optimizer step per n.  The data-parallel exchange then runs once, on the
        summed gradient of the boundary micro-step (the reference's DDP all-reduces on every micro-step: same result).  Call
        before capture(): the captured loss kernel carries the 1 / n."""
        assert n >= 1 and getattr(self, "_graphs", None) is None, "set_grad_accumulation() before capture()"
        self.ga, self._micro = int(n), 0
        self._gacc = (torch.zeros_like(self.P.grad), torch.zeros_like(self.head.P.grad)) if n > 1 else None

    def _accumulate(self) -> bool:
        """Fold the micro-step's gradients into the accumulators; True on the boundary micro-step (the grad buffers then hold the sums)."""
        if self.ga == 1:
            return True
        for acc, g in zip(self._gacc, (self.P.grad, self.head.P.grad)):
            if self._micro == 0:
                n = g.numel()
                ops.copy2d(g, acc, 1, n, n, n)
            else:
                ops.add_(acc, g)
        self._micro += 1
        if self._micro < self.ga:
            return False
        self._micro = 0
        for acc, g in zip(self._gacc, (self.P.grad, self.head.P.grad)):
            n = g.numel()
            ops.copy2d(acc, g, 1, n, n, n)
        red = self.eng.reducer
        if red is not None:                          # one exchange, of the sums
            red.reduce_async(self.P.grad, 0, None)
            red.reduce_async(self.head.P.grad, 0, None)
        return True

    def train_step(self, batch, lr: float, noise=None):
        """One micro-step, launched eagerly on the three streams (with a reducer and no accumulation: every gradient range goes
        to the exchange as soon as it is final); the optimizer steps on every ``ga``-th call."""
        if self.ga == 1 and self.overlap_update:
            self.step_count += 1
            self._run(self._segments(batch, noise), update=(lr, 0.9, 0.999, 1e-8, 0.01))
            self._after_update(refresh=True)
            return self._loss3
        self._run(self._segments(batch, noise, 1.0 / self.ga), exchange=self.ga == 1)
        if self._accumulate():
            self.optimizer_step(lr)
        return self._loss3

    def capture(self, batch: Dict[str, torch.Tensor], noise: Optional[torch.Tensor] = None, warmup: int = 2):
        """The step on the static ``batch`` / ``noise`` buffers (copy new data into them before each replay) as one linear hipGraph
        per segment of the schedule (_segments), replayed on the segment's stream with plain events between them.  The host hands
        a segment's finished gradient ranges to the exchange stream between two replays, so the collectives of a captured
        multi-rank step run underneath the remaining backward exactly as in the eager step (collectives themselves are not
        captured).  AdamW stays outside (host-side bias corrections); the derived-operand rebuild is a last small graph.  A step is
        3000-5000 launches: issued from Python they cost more host time than GPU time."""
        for _ in range(warmup):                      # allocate every buffer / set kernel attributes outside the capture
            self.head.dirty = True
            self._run(self._segments(batch, noise, 1.0 / self.ga), exchange=False)
        torch.cuda.synchronize()
        self.head.dirty = True                       # the head's own W^T / padded-operand refresh becomes part of its graphs
        self._segs = self._segments(batch, noise, 1.0 / self.ga)
        # one memory pool and one capture stream per stream kind: graphs sharing a pool are replayed strictly in capture order on
        # ONE stream, so the allocator's reuse of freed capture-time temporaries stays race-free while the streams overlap
        pools = {k: torch.cuda.graph_pool_handle() for k in "MHGV"}
        caps = {k: torch.cuda.Stream() for k in "MHGV"}
        self._graphs = []
        for st, fn, _, _, _ in self._segs:
            if fn is None:
                self._graphs.append(None)
                continue
            kind = st if self._stream(st, None) is not None else "M"      # (single-stream mode: everything is an "M" graph)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=pools[kind], stream=caps[kind], capture_error_mode="thread_local"):
                fn()
            self._graphs.append(g)
        # derived operands: per finished range a small graph behind that range's AdamW (overlapped update only), the rest at the step's end
        self._rgraphs, covered, rpool = {}, set(), torch.cuda.graph_pool_handle()
        if self.ga == 1 and self.overlap_update and self._refresh_pieces():
            for k, (st, fn, _, _, ranges) in enumerate(self._segs):
                if not ranges:
                    continue
                before = set(self._refreshed)
                fns = self._refresh_pieces_in(ranges)
                covered |= self._refreshed - before
                if fns:
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, pool=rpool, stream=caps["G"], capture_error_mode="thread_local"):     # (a pool of their own: they
                        for f_ in fns:                                                                               #  replay between the G graphs)
                            f_()
                    self._rgraphs[k] = g
            self._refreshed = set()
        rest = [f_ for i, (_, _, f_) in enumerate(self._refresh_pieces()) if i not in covered]
        has_extra = type(self)._refresh_extra is not BackboneTrainer._refresh_extra
        self._g_r = torch.cuda.CUDAGraph() if (not self._rgraphs or rest or has_extra) else None      # (nothing left: LoRA - every pair lies in a range)
        if self._g_r is not None:
            with torch.cuda.graph(self._g_r, pool=pools["M"], stream=caps["M"], capture_error_mode="thread_local"):
                if self._rgraphs:
                    for f_ in rest:
                        f_()
                    self._refresh_extra()
                else:
                    self.refresh()
        torch.cuda.synchronize()

    def train_step_graphed(self, lr: float):
        if self.ga == 1 and self.overlap_update:
            self.step_count += 1
            self._run(self._segs, self._graphs, update=(lr, 0.9, 0.999, 1e-8, 0.01))
            self._after_update(refresh=False)
            if self._g_r is not None:
                self._g_r.replay()
            return self._loss3
        self._run(self._segs, self._graphs, exchange=self.ga == 1)
        if self._accumulate():
            self.optimizer_step(lr, refresh=False)
            if self._g_r is not None:
                self._g_r.replay()
        return self._loss3

    def _after_update(self, refresh: bool):
        if self.eng.reducer is not None:
            self.eng.reducer._pending = False        # every collective was joined range by range (_update_ranges)
        self.head.dirty = True
        if refresh:
            self._refresh_rest()                     # (eager: the ranges' pieces ran behind their AdamW; captured: the caller replays _g_r)

    def _adam_ranges(self):
        return [(0, self.P.numel)]

    def _exchange_and_scale(self) -> float:
        """Join the data-parallel exchange the backward started range by range; returns the 1/N scale folded into AdamW."""
        red = self.eng.reducer
        if red is None:
            return 1.0
        red.wait()
        return red.grad_scale

    def optimizer_step(self, lr: float, beta1=0.9, beta2=0.999, eps=1e-8, wd=0.01, refresh: bool = True):
        self.step_count += 1
        gscale = self._exchange_and_scale()
        P, HP = self.P, self.head.P
        for lo, hi in self._adam_ranges():            # (parameters of dead LLM layers: see _segments)
            ops.adamw_(P.data[lo:hi], P.grad[lo:hi], P.m[lo:hi], P.v[lo:hi], self.step_count, lr, beta1, beta2, eps, wd, gscale=gscale)
        ops.adamw_(HP.data, HP.grad, HP.m, HP.v, self.step_count, lr, beta1, beta2, eps, wd, gscale=gscale)
        self.head.dirty = True
        if refresh:
            self.refresh()


# ------------------------------------------------------------------------------------------------ full fine-tune
class FullFinetune(BackboneTrainer):
    mode = "full"
    trains_vectors = True

    def __init__(self, eng: E.VLAEngine):
        super().__init__(eng)
        cfg = self.cfg
        slots: List[_Slot] = []
        # ---- adopt every VLM tensor into one flat buffer: matrices first (per parameter group contiguous: ViT blocks, projector,
        #      LLM layers - the data-parallel exchange starts per group), then the vector section (fp32-accumulated gradients)
        for j, v in enumerate(self.vits):
            slots.append(_Slot(f"vit{j}.wpe", v, "wpe", False))
            for i, b in enumerate(v.blocks):
                for k in ("wqkv", "wproj", "w1", "w2"):
                    slots.append(_Slot(f"vit{j}.{i}.{k}", b, k, False))
        for k in eng.proj:
            if k.endswith("weight"):
                slots.append(_Slot("proj." + k, eng.proj, k, False))
        for i, L in enumerate(self.llm.layers):
            for k in ("wqkv", "wo", "wgu", "wd"):
                slots.append(_Slot(f"llm.{i}.{k}", L, k, False))
        slots.append(_Slot("llm.embed", self.llm, "embed", False))
        first_vec = len(slots)
        for j, v in enumerate(self.vits):
            slots += [_Slot(f"vit{j}.bpe", v, "bpe", True), _Slot(f"vit{j}.pos", v, "pos", True)]
            if v.cfg.n_prefix:
                slots.append(_Slot(f"vit{j}.prefix", v, "prefix", True))
            for i, b in enumerate(v.blocks):
                for k in ("n1w", "n1b", "bqkv", "bproj", "n2w", "n2b", "b1", "b2") + (("ls1", "ls2") if v.cfg.layerscale else ()):
                    slots.append(_Slot(f"vit{j}.{i}.{k}", b, k, True))
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
        self.vec_off = self.P.offsets[slots[first_vec].name][0]
        self.acc32 = torch.zeros(self.P.numel - self.vec_off, device=self.dev, dtype=torch.float32)
        # W^T operands of the ViT / projector dX products (the LLM's already exist: frozen-path dX)
        z = lambda r, c: torch.empty(r, c, device=self.dev, dtype=BF16)
        for v in self.vits:
            for b in v.blocks:
                b["wqkvT"], b["wprojT"], b["w1T"], b["w2T"] = z(v.cfg.d, 3 * v.cfg.d), z(v.cfg.d, v.cfg.d), z(v.cfg.d, v.mlp_pad), z(v.mlp_pad, v.cfg.d)
        self.projT = {k: z(w.shape[1], w.shape[0]) for k, w in eng.proj.items() if k.endswith("weight")}
        self.refresh()

    vit = property(lambda self: self.vits[0])

    # ---- bookkeeping
    def G(self, name):
        return self.P.g(name)

    def A(self, name):
        """fp32 accumulator of a vector-section parameter."""
        off, shape = self.P.offsets[name]
        return self.acc32[off - self.vec_off:off - self.vec_off + math.prod(shape)].view(shape)

    _KEYMAP = {"qkv": "wqkv", "proj": "wproj", "fc1": "w1", "fc2": "w2", "o": "wo", "gu": "wgu", "down": "wd"}

    def _gname(self, key: str) -> str:
        part, leaf = key.rsplit(".", 1)
        return f"{part}.{leaf}.weight" if part == "proj" else f"{part}.{self._KEYMAP[leaf]}"

    def refresh(self):
        """W^T operands of the dX products, rebuilt after every update."""
        for v in self.vits:
            for b in v.blocks:
                for k in ("wqkv", "wproj", "w1", "w2"):
                    ops.transpose(b[k], out=b[k + "T"])
        for k, t in self.projT.items():
            ops.transpose(self.eng.proj[k], out=t)
        for L in self.llm.layers:
            for k in ("wqkv", "wo", "wgu", "wd"):
                ops.transpose(L[k], out=L[k + "T"])
        self._refresh_objective()

    refresh_transposes = refresh

    def _refresh_pieces(self):
        """One piece per weight matrix: its W^T copy (the flat offsets of the matrix in self.P)."""
        if getattr(self, "_pieces", None) is None:
            out = []

            def add(name, holder, k):
                lo, n = self.P.offsets[name][0], holder[k].numel()
                out.append((lo, lo + n, lambda h=holder, k=k: ops.transpose(h[k], out=h[k + "T"])))
            for j, v in enumerate(self.vits):
                for i, b in enumerate(v.blocks):
                    for k in ("wqkv", "wproj", "w1", "w2"):
                        add(f"vit{j}.{i}.{k}", b, k)
            for k, t in self.projT.items():
                lo = self.P.offsets["proj." + k][0]
                out.append((lo, lo + t.numel(), lambda k=k, t=t: ops.transpose(self.eng.proj[k], out=t)))
            for i, L in enumerate(self.llm.layers):
                for k in ("wqkv", "wo", "wgu", "wd"):
                    add(f"llm.{i}.{k}", L, k)
            self._pieces = out
        return self._pieces

    def _refresh_extra(self):
        self._refresh_objective()

    # ---- the Linear of this mode
    def _lin(self, key, x, W, bias=None, **kw):
        return ops.gemm_nt(x, W, bias=bias, **kw)

    def _lin_bwd(self, key, dy, x, WT, out=None, swiglu_gu=None):
        """dx = dy W on the chain; dW = dy^T x (TN GEMM on dy and x as they lie) handed to the gradient stream: dy is the layer's
        persistent dY slot, x a saved activation - both stay untouched until the next step."""
        self._defer_tn(dy, x, self.G(self._gname(key)))
        if swiglu_gu is not None:            # down_proj: dGU = swiglu'(GU) * (dy W_down) in the dX GEMM's epilogue
            return ops.gemm_swiglu_bwd(dy, WT, swiglu_gu, out=out)
        return ops.gemm_nt(dy, WT, out=out)

    def _embed_backward(self, dX0):
        ids, q0, G, B, Np = self._batch["input_ids"], self.eng.qidx0, self.G("llm.embed"), self.eng.B, self.eng.Np
        self._defer(lambda: ops.N.check(ops._lib().vla_embed_grad(ops._st(), ops._p(dX0), ops._p(ids), ops._p(q0), ops._p(G), B, ids.shape[1], Np,
                                                                  self.cfg.llm.d, self.cfg.llm.vocab), "embed_grad"))      # gradient-only: off the dX chain

    def _adam_ranges(self):
        na, n = self.n_active, self.cfg.llm.n_layers
        if na == n:
            return [(0, self.P.numel)]
        off = lambda name: self.P.offsets[name][0]          # dead: matrices and vectors of layers na .. n-1, the final norm
        return [(0, off(f"llm.{na}.wqkv")), (off("llm.embed"), off(f"llm.{na}.n1"))]

    def _begin_backward(self):
        ops.zero_(self.acc32)
        ops.zero_(self.G("llm.embed"))

    def _end_backward(self):
        ops.cast_f32_bf16(self.acc32, out=self.P.grad[self.vec_off:])       # every bias / norm / LayerScale / pos-embed / token gradient in one cast

    def _ranges(self, kind, lo=0, hi=0, j=0):
        if kind == "llm":
            return [self._span(f"llm.{lo}.wqkv", f"llm.{hi}.wd")]
        if kind == "vit":
            return [self._span(f"vit{j}.{lo}.wqkv", f"vit{j}.{hi}.w2")]
        if kind == "embed":
            return [self._span("llm.embed", "llm.embed")]
        if kind == "proj":
            names = ["proj." + k for k in self.eng.proj if k.endswith("weight")]
            return [self._span(names[0], names[-1])]
        # tail: the patch-embedding matrices (final after each backbone's last block) and the vector section - without the norms /
        # biases of dead LLM layers and the final norm when layers lie above the head's last block (they close the section): the
        # reference hands them all-zero gradients and never moves them, exactly what _adam_ranges does on the un-overlapped path
        na, n = self.n_active, self.cfg.llm.n_layers
        vec_end = self.P.numel if na == n else self.P.offsets[f"llm.{na}.n1"][0]
        return [self._span(f"vit{j}.wpe", f"vit{j}.wpe") for j in range(len(self.vits))] + [(self.P.grad, self.vec_off, vec_end)]

    def reference_named_gradients(self) -> Dict[str, torch.Tensor]:
        """Gradients under the reference's state-dict names (fused layouts undone) - for parity tests / checkpoints."""
        cfg, out = self.cfg, {}
        c = cfg.llm
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
        names = ["vision_backbone.featurizer."] + (["vision_backbone.fused_featurizer."] if cfg.fused else [])
        for j, (pre, v) in enumerate(zip(names, self.vits)):
            vc, P_ = v.cfg, v.cfg.patch
            out[pre + "patch_embed.proj.weight"] = self.G(f"vit{j}.wpe")[:, :3 * P_ * P_].reshape(vc.d, 3, P_, P_)
            out[pre + "patch_embed.proj.bias"], out[pre + "pos_embed"] = self.G(f"vit{j}.bpe"), self.G(f"vit{j}.pos").reshape(1, -1, vc.d)
            if vc.n_prefix:
                gp = self.G(f"vit{j}.prefix")
                out[pre + "cls_token"] = gp[:1].reshape(1, 1, vc.d)
                if vc.n_prefix > 1:
                    out[pre + "reg_token"] = gp[1:].reshape(1, vc.n_prefix - 1, vc.d)
            for i in range(len(v.blocks)):
                q = f"{pre}blocks.{i}."
                g = lambda k: self.G(f"vit{j}.{i}.{k}")
                out[q + "norm1.weight"], out[q + "norm1.bias"], out[q + "norm2.weight"], out[q + "norm2.bias"] = g("n1w"), g("n1b"), g("n2w"), g("n2b")
                out[q + "attn.qkv.weight"], out[q + "attn.qkv.bias"] = g("wqkv"), g("bqkv")
                out[q + "attn.proj.weight"], out[q + "attn.proj.bias"] = g("wproj"), g("bproj")
                out[q + "mlp.fc1.weight"], out[q + "mlp.fc1.bias"] = g("w1")[:vc.mlp], g("b1")[:vc.mlp]
                out[q + "mlp.fc2.weight"], out[q + "mlp.fc2.bias"] = g("w2")[:, :vc.mlp], g("b2")
                if vc.layerscale:
                    out[q + "ls1.scale_factor"], out[q + "ls2.scale_factor"] = g("ls1"), g("ls2")
        for k in self.eng.proj:
            out["projector." + k] = self.G("proj." + k)
        return out


# ------------------------------------------------------------------------------------------------ LoRA
class LoRAFinetune(BackboneTrainer):
    mode = "lora"
    trains_vectors = False

    def __init__(self, eng: E.VLAEngine, rank: int = 32, seed: int = 0, fp8: bool = False, fp8_backward: Optional[bool] = None,
                 dropout: float = 0.0):
        """dropout: peft's ``lora_dropout`` (vla-scripts/finetune.py:110, 832-840; every shipped script uses 0.0): each wrapped module drops
        ITS OWN copy of the input before lora_A - q / k / v of one fused projection draw three masks over the same x.  With p > 0 a pair's
        t_j = 2 dropout_j(x) A_j^T comes from its own dropped input (kept for dA_j = dt_j^T dropout_j(x)), the base product keeps t B^T
        inside its accumulator, and the input gradient leaves the single accumulator: dx = dy W + sum_j mask_j * (dt_j A_j) / (1 - p)
        (one K = r product and one masked accumulation per pair; down_proj's SwiGLU backward then runs as its own pass).  Masks are
        counter-based (vla_dropout_bf16: seed, pair, step) - regenerated in the backward, fresh on every replay of a captured step;
        torch's Philox stream is not reproduced (statistical parity only; tests hand the generated masks to the oracle).  Not with fp8.
        fp8: BASELINE configs[4]'s "fp8 MFMA weight path" where it belongs - under LoRA every base weight is frozen, so every base
        product runs on OCP e4m3 operands (weights quantised once, one scale per output channel; activations per row: inside the
        norm that produces them, or by one pass over the producer's output) while the rank-r branch stays bf16 INSIDE the same
        accumulator (the GEMM's K extension on the dequantised base product).  fp8_backward (default: as fp8): the dX products
        dy W likewise, on e4m3 W^T (one scale per input channel) and row-quantised dy; the adapter gradients dA / dB read the bf16
        activations and bf16 dy as before.  The reference has no fp8 code (it runs bf16 everywhere): PARITY UNPINNED; the oracle
        restates this arithmetic (oracle.FP8 registry + FP8_BWD).  A Linear whose contraction length is not a multiple of 128
        (plumbing-size configs) keeps bf16."""
        super().__init__(eng)
        cfg, self.rank = self.cfg, rank
        self.fp8 = bool(fp8)
        self.fp8_backward = self.fp8 if fp8_backward is None else (bool(fp8_backward) and self.fp8)
        self.dropout = float(dropout)
        assert 0.0 <= self.dropout < 1.0 and not (self.dropout > 0 and self.fp8), "lora_dropout in [0, 1); not together with the fp8 base products"
        self._drop_seed = (int(seed) * 0x9E3779B97F4A7C15 + 0x5851F42D4C957F2D) & (2 ** 64 - 1)
        self._drop_step = torch.zeros(1, dtype=torch.int32, device=eng.device)      # bumped by the first launch of every step (in the graph)
        self.Yd, self._ubuf = {}, {}          # dropout: dropped inputs per Linear [pairs, M, K] (kept for dA); u = dt_j A_j scratch per shape
        c = cfg.llm
        H, KV, dh, I, D = c.heads, c.kv_heads, c.dh, c.inter, c.d
        L: Dict[str, LoraLinear] = {}
        pre = "base_model.model."
        names = ["vision_backbone.featurizer."] + (["vision_backbone.fused_featurizer."] if cfg.fused else [])
        for j, (vn, v) in enumerate(zip(names, self.vits)):
            d = v.cfg.d
            for i in range(len(v.blocks)):
                q = f"{pre}{vn}blocks.{i}."
                L[f"vit{j}.{i}.qkv"] = LoraLinear(q + "attn", 3 * d, d, [("qkv", ("range", 0, 3 * d))], rank)
                L[f"vit{j}.{i}.proj"] = LoraLinear(q + "attn", d, d, [("proj", ("range", 0, d))], rank)
                L[f"vit{j}.{i}.fc1"] = LoraLinear(q + "mlp", v.mlp_pad, d, [("fc1", ("range", 0, v.mlp_pad))], rank, n_real=v.cfg.mlp)
                L[f"vit{j}.{i}.fc2"] = LoraLinear(q + "mlp", d, v.mlp_pad, [("fc2", ("range", 0, d))], rank, k_real=v.cfg.mlp)
        for k, w in eng.proj.items():
            if k.endswith("weight"):
                n = k.split(".")[0]
                L[f"proj.{n}"] = LoraLinear(pre + "projector", w.shape[0], w.shape[1], [(n, ("range", 0, w.shape[0]))], rank)
        for i in range(c.n_layers):
            q = f"{pre}language_model.model.layers.{i}."
            L[f"llm.{i}.qkv"] = LoraLinear(q + "self_attn", (H + 2 * KV) * dh, D, [("q_proj", ("range", 0, H * dh)), ("k_proj", ("range", H * dh, (H + KV) * dh)),
                                                                                ("v_proj", ("range", (H + KV) * dh, (H + 2 * KV) * dh))], rank)
            L[f"llm.{i}.o"] = LoraLinear(q + "self_attn", D, H * dh, [("o_proj", ("range", 0, D))], rank)
            L[f"llm.{i}.gu"] = LoraLinear(q + "mlp", 2 * I, D, [("gate_proj", ("group16", 0, I)), ("up_proj", ("group16", 16, I))], rank)
            L[f"llm.{i}.down"] = LoraLinear(q + "mlp", D, I, [("down_proj", ("range", 0, D))], rank)
        self.L = L
        spec = []                      # per fused Linear: its A's (adjacent: A_cat is a view), then its B's - a layer's pairs are one range
        for l in L.values():
            a, b = l.spec()
            spec += a + b
        self.P = E.FlatParams(spec, self.dev)
        gen = torch.Generator(device=self.dev).manual_seed(seed)
        for l in L.values():
            l.bind(self.P, self.dev)
            l.init_(gen)
        # W^T operands of the ViT / projector dX products (the LLM's exist already); the base weights are frozen: built once
        for v in self.vits:
            for b in v.blocks:
                for k in ("wqkv", "wproj", "w1", "w2"):
                    b[k + "T"] = ops.transpose(b[k])
        self.projT = {k: ops.transpose(w) for k, w in eng.proj.items() if k.endswith("weight")}
        self.T, self.DT = {}, {}              # t = 2 x A^T per LoRA Linear (kept for dB), dt = 2 dy B scratch per shape
        self.Q, self.QT = {}, {}              # fp8: key -> (e4m3 codes, fp32 scales) of W [out, in] / of W^T [in, out]
        self._qbuf, self._xq = {}, None       # row-quantised activations: scratch per shape; (data_ptr, codes, scales) of the last norm output
        if self.fp8:
            for key in L:
                holder, wk = self._base(key)
                W, WT = holder[wk], self._baseT(key)
                if W.shape[1] % 128 == 0:
                    self.Q[key] = ops.quant_fp8_rows(W)
                if self.fp8_backward and WT.shape[1] % 128 == 0:
                    self.QT[key] = ops.quant_fp8_rows(WT)
        self.refresh()

    vit = property(lambda self: self.vits[0])

    def refresh(self):
        for l in self.L.values():
            l.refresh()

    def _refresh_pieces(self):
        """One piece per wrapped Linear: B_blk, A_cat^T, B_blk^T from its pairs (adjacent in the flat buffer: first A to last B)."""
        if getattr(self, "_pieces", None) is None:
            out = []
            for l in self.L.values():
                lo = self.P.offsets[f"{l.name}.{l.projs[0][0]}.lora_A"][0]
                off, shape = self.P.offsets[f"{l.name}.{l.projs[-1][0]}.lora_B"]
                out.append((lo, off + int(torch.Size(shape).numel()), l.refresh))
            self._pieces = out
        return self._pieces

    def fp8_keys(self):
        """(forward keys, backward keys) of the Linears that run on e4m3 base operands (tests register the same set with the oracle)."""
        return sorted(self.Q), sorted(self.QT)

    def _baseT(self, key: str):
        part, *rest = key.split(".")
        if part.startswith("vit"):
            return self.vits[int(part[3:])].blocks[int(rest[0])][{"qkv": "wqkvT", "proj": "wprojT", "fc1": "w1T", "fc2": "w2T"}[rest[1]]]
        if part == "proj":
            return self.projT[rest[0] + ".weight"]
        return self.llm.layers[int(rest[0])][{"qkv": "wqkvT", "o": "woT", "gu": "wguT", "down": "wdT"}[rest[1]]]

    # ---- fp8: row-quantised activations
    def _qscratch(self, rows: int, cols: int, slot: str):
        k = (rows, cols, slot)
        b = self._qbuf.get(k)
        if b is None:
            b = self._qbuf[k] = (torch.empty(rows, cols, device=self.dev, dtype=torch.uint8), torch.empty(rows, device=self.dev, dtype=torch.float32))
        return b

    def _quant(self, x, slot: str):
        """(codes, scales) of the rows of x: taken from the norm that just produced x, else one pass over x (vla_quant_fp8_rows)."""
        if self._xq is not None and self._xq[0] == (x.data_ptr(), tuple(x.shape)):
            hit, self._xq = self._xq, None               # (consumed: the scratch is overwritten by the next norm of this shape)
            return hit[1], hit[2]
        q, s_ = self._qscratch(x.shape[0], x.shape[1], slot)
        return ops.quant_fp8_rows(x, q, s_)

    def _ln(self, x, w, b, y, st, eps):
        if not self.fp8 or x.shape[1] % 128:
            return super()._ln(x, w, b, y, st, eps)
        q, s_ = self._qscratch(x.shape[0], x.shape[1], "n")
        ops.layernorm_fwd_q8(x, w, b, eps, q, s_, y=y, stats=st)          # the row is still in registers: bit-identical to norm + quantise
        self._xq = ((y.data_ptr(), tuple(y.shape)), q, s_)

    def _rms(self, x, w, out, rstd):
        if not self.fp8 or x.shape[1] % 128:
            return super()._rms(x, w, out, rstd)
        q, s_ = self._qscratch(x.shape[0], x.shape[1], "n")
        ops.rmsnorm_fwd_q8(x, w, self.cfg.llm.eps, q, s_, rstd=rstd, y=out)
        self._xq = ((out.data_ptr(), tuple(out.shape)), q, s_)

    def _alloc(self, B, S):
        if self._key != (B, S):
            self.T, self.DT, self.Yd, self._ubuf = {}, {}, {}, {}
        super()._alloc(B, S)

    def _begin_forward(self):
        if self.dropout > 0:
            ops.inc_i32_(self._drop_step)

    def drop_seed(self, key: str, j: int) -> int:
        """Mask key of pair j of the wrapped Linear `key` (with the step counter: vla_dropout_bf16's (seed, step))."""
        return (self._drop_seed + (list(self.L).index(key) * 8 + j + 1) * 0xC2B2AE3D27D4EB4F) & (2 ** 64 - 1)

    # ---- the Linear of this mode: base product with the low-rank branch inside its accumulator (GEMM K extension)
    def _lin(self, key, x, W, bias=None, **kw):
        l = self.L[key]
        t = self.T.get(key)
        if t is None or t.shape[0] != x.shape[0]:
            t = self.T[key] = torch.empty(x.shape[0], l.Rr, device=self.dev, dtype=BF16)
        if self.dropout > 0:                                        # every pair drops its own copy of x (kept: dA_j needs it)
            Y = self.Yd.get(key)
            if Y is None or Y.shape[1] != x.shape[0]:
                Y = self.Yd[key] = torch.empty(len(l.projs), x.shape[0], x.shape[1], device=self.dev, dtype=BF16)
            for j in range(len(l.projs)):
                ops.dropout(x, Y[j], self.dropout, self.drop_seed(key, j), self._drop_step)
                ops.gemm_nt(Y[j], l.A_cat[j * l.rp:(j + 1) * l.rp], alpha=2.0, out=t[:, j * l.rp:(j + 1) * l.rp])
        else:
            ops.gemm_nt(x, l.A_cat, alpha=2.0, out=t)               # t = 2 x A_cat^T   (alpha / r = 2)
        wq = self.Q.get(key)
        if wq is not None:                                          # e4m3 base operands, bf16 low-rank branch, one accumulator
            xq, xs = self._quant(x, "f")
            return ops.gemm_nt(xq, wq[0], bias=bias, fp8=(xs, wq[1]), ext=(t, l.B_blk), **kw)
        return ops.gemm_nt(x, W, bias=bias, ext=(t, l.B_blk), **kw)

    def _lin_bwd(self, key, dy, x, WT, out=None, swiglu_gu=None):
        l = self.L[key]
        M = dy.shape[0]
        dt = self.DT.get(key)                                       # one per Linear: the pair gradients read it on the gradient stream
        if dt is None or dt.shape[0] != M:
            dt = self.DT[key] = torch.empty(M, l.Rr, device=self.dev, dtype=BF16)
        ops.gemm_nt(dy, l.B_blkT, alpha=2.0, out=dt)                # dt = 2 dy B_blk
        if self.dropout > 0:
            return self._lin_bwd_dropout(key, l, dy, dt, WT, out, swiglu_gu)
        l.grads(dy, x, self.T[key], dt, self._defer_tn)             # dA_cat, dB_j: gradient-only work
        wq = self.QT.get(key)
        if wq is not None:                                          # dx = Q(dy) Q(W^T)^T + dt A_cat
            dq, ds = self._quant(dy, "b")
            if swiglu_gu is not None:
                return ops.gemm_swiglu_bwd(dq, wq[0], swiglu_gu, out=out, ext=(dt, l.A_catT), fp8=(ds, wq[1]))
            return ops.gemm_nt(dq, wq[0], out=out, fp8=(ds, wq[1]), ext=(dt, l.A_catT))
        if swiglu_gu is not None:
            return ops.gemm_swiglu_bwd(dy, WT, swiglu_gu, out=out, ext=(dt, l.A_catT))
        return ops.gemm_nt(dy, WT, out=out, ext=(dt, l.A_catT))     # dx = dy W + dt A_cat

    def _lin_bwd_dropout(self, key, l, dy, dt, WT, out, swiglu_gu):
        """lora_dropout > 0: dA_j reads pair j's own dropped input; dx = dy W + sum_j mask_j * (dt_j A_j) / (1 - p) - the low-rank share
        passes back through each pair's mask, so it cannot ride in the base product's accumulator."""
        M, rp = dy.shape[0], l.rp
        l.grads(dy, self.Yd[key], self.T[key], dt, self._defer_tn, dropped=True)
        if swiglu_gu is not None:                                   # down_proj: dH first, its SwiGLU backward as a pass of its own
            dx = self._uscratch(M, WT.shape[0], "h")
            ops.gemm_nt(dy, WT, out=dx)
        else:
            dx = ops.gemm_nt(dy, WT, out=out)
        u = self._uscratch(M, WT.shape[0], "u")
        for j in range(len(l.projs)):
            ops.gemm_nt(dt[:, j * rp:(j + 1) * rp], l.A_catT[:, j * rp:(j + 1) * rp], out=u)        # u = dt_j A_j   (K = r)
            ops.dropout_bwd_add_(dx, u, self.dropout, self.drop_seed(key, j), self._drop_step)
        if swiglu_gu is not None:
            return ops.swiglu_bwd(dx, swiglu_gu, out=out)
        return dx

    def _uscratch(self, rows: int, cols: int, slot: str):
        k = (rows, cols, slot)
        b = self._ubuf.get(k)
        if b is None:
            b = self._ubuf[k] = torch.empty(rows, cols, device=self.dev, dtype=BF16)
        return b

    def _adam_ranges(self):
        na = self.n_active
        if na == self.cfg.llm.n_layers:
            return [(0, self.P.numel)]
        l = self.L[f"llm.{na}.qkv"]                          # the pairs of dead layers close the flat buffer
        return [(0, self.P.offsets[f"{l.name}.{l.projs[0][0]}.lora_A"][0])]

    def _ranges(self, kind, lo=0, hi=0, j=0):
        first = lambda key: f"{self.L[key].name}.{self.L[key].projs[0][0]}.lora_A"
        last = lambda key: f"{self.L[key].name}.{self.L[key].projs[-1][0]}.lora_B"
        if kind == "llm":
            return [self._span(first(f"llm.{lo}.qkv"), last(f"llm.{hi}.down"))]
        if kind == "vit":
            return [self._span(first(f"vit{j}.{lo}.qkv"), last(f"vit{j}.{hi}.fc2"))]
        if kind == "proj":
            keys = [k for k in self.L if k.startswith("proj.")]
            return [self._span(first(keys[0]), last(keys[-1]))]
        return []                       # embed / tail: frozen under LoRA

    # ---- adapters in and out
    def lora_state_dict(self) -> Dict[str, torch.Tensor]:
        """peft's adapter key layout ('....q_proj.lora_A.weight' [r, in], '....lora_B.weight' [out, r]); rank padding and the
        ViT MLP's width padding removed."""
        out = {}
        for l in self.L.values():
            for p, d in l.projs:
                A, B = self.P.view(f"{l.name}.{p}.lora_A")[:l.r, :l.k_real], self.P.view(f"{l.name}.{p}.lora_B")[:l.n_real, :l.r]
                out[f"{l.name}.{p}.lora_A.weight"], out[f"{l.name}.{p}.lora_B.weight"] = A, B
        return out

    def load_lora_state_dict(self, sd: Dict[str, torch.Tensor]):
        """Inverse of lora_state_dict (``lora_adapter/adapter_model.safetensors``: resume, offline merge).  Every pair must be
        present with this trainer's rank; paddings are re-zeroed."""
        for l in self.L.values():
            for p, d in l.projs:
                A, B = self.P.view(f"{l.name}.{p}.lora_A"), self.P.view(f"{l.name}.{p}.lora_B")
                a, b = sd[f"{l.name}.{p}.lora_A.weight"], sd[f"{l.name}.{p}.lora_B.weight"]
                assert tuple(a.shape) == (l.r, l.k_real) and tuple(b.shape) == (min(l.n_real, B.shape[0]), l.r), \
                    f"{l.name}.{p}: adapter shapes {tuple(a.shape)} / {tuple(b.shape)} do not match rank {l.r}"
                A.zero_()
                B.zero_()
                A[:l.r, :l.k_real].copy_(a.to(self.dev, BF16))
                B[:b.shape[0], :l.r].copy_(b.to(self.dev, BF16))
        self.refresh()

    def merged_weights(self) -> Dict[str, torch.Tensor]:
        """W + 2 B A per target under the engine's fused names (merge_lora_weights_and_save.py / finetune.py:579-601 merge the
        adapter into a fresh bf16 base): fp32 product, one rounding."""
        out = {}
        for key, l in self.L.items():
            holder, wk = self._base(key)
            delta = 2.0 * (l.B_blk.float() @ l.A_cat.float())
            out[key] = (holder[wk].float() + delta).to(BF16)
        return out

    def _base(self, key: str):
        part, *rest = key.split(".")
        if part.startswith("vit"):
            return self.vits[int(part[3:])].blocks[int(rest[0])], {"qkv": "wqkv", "proj": "wproj", "fc1": "w1", "fc2": "w2"}[rest[1]]
        if part == "proj":
            return self.eng.proj, rest[0] + ".weight"
        return self.llm.layers[int(rest[0])], {"qkv": "wqkv", "o": "wo", "gu": "wgu", "down": "wd"}[rest[1]]
