"""LoRA fine-tune step (the reference's default recipe: vla-scripts/finetune.py:832-844).

peft ``LoraConfig(r, lora_alpha=2r, lora_dropout, target_modules="all-linear", init_lora_weights="gaussian")`` wraps every
``nn.Linear`` of the VLM except the output head: ``y = W x + b + (alpha / r) * B(A x)`` with ``A ~ N(0, 1/r^2)`` (std 1/r) of
shape [r, in], ``B = 0`` of shape [out, r]; the base weights are frozen, ``action_queries`` is re-enabled, the action head
and the proprio projector (separate modules) train in full.  Targets here: timm ``attn.qkv / attn.proj / mlp.fc1 / mlp.fc2``
of every useful ViT block (the patch embedding is a Conv2d: not "linear"), the projector's ``fc1 / fc2 (/ fc3)``, Qwen2's
``q / k / v / o / gate / up / down`` (``embed_tokens`` is an Embedding, ``lm_head`` the excluded output layer).
PARITY UNPINNED: peft is not importable here; checked against ``oracle.lora_linear`` autograd (tests/test_lora_gpu.py).

MI355X-first layout.  The kernels read FUSED base weights (q|k|v stacked; gate/up interleaved in 16-row groups), so a fused
base Linear carries P LoRA pairs.  Parameters are stored per pair, exactly as peft names them (``...q_proj.lora_A/B``), in one
flat buffer: the A's of one fused Linear are adjacent, so ``A_cat [P r, in]`` is a VIEW; the dense block matrix
``B_blk [out, P r]`` (pair j's B in its own rows x columns, zeros elsewhere) is a derived operand rebuilt after every update.
Forward  t = x A_cat^T (one skinny GEMM), y += 2 t B_blk^T (epilogue-accumulate GEMM on the base output).
Backward dt = 2 dy B_blk, dA_cat = dt^T x (one GEMM into the flat gradient), dB_j = 2 dy_j^T t_j per pair (the rows of
dy^T that belong to pair j: a contiguous range, or 16-row groups every 32 rows for gate / up - the GEMM's row-group
addressing), dx += dt A_cat.  Ranks are zero-padded to a multiple of 64 (the GEMM's K granule): padded rows / columns have
zero value and zero gradient and stay zero under AdamW.
RoPE and SwiGLU can no longer live in the base GEMM's epilogue (the low-rank delta must be added to the pre-activation first):
stand-alone ``vla_rope_half`` / ``vla_swiglu_fwd`` / ``vla_swiglu_bwd`` take over.  The backward is the full-sequence dX chain
of full_finetune.py without the base dW products.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, List, Optional, Tuple

import torch

from . import engine as E
from . import ops
from .ops import BF16

rup = E.rup


class LoraLinear:
    """LoRA pairs of one fused base Linear W [n_out, k_in].  projs: [(peft name, ("range", lo, hi) | ("group16", offset, n_rows))]."""

    def __init__(self, name: str, n_out: int, k_in: int, projs, r: int):
        self.name, self.n_out, self.k_in, self.projs, self.r = name, n_out, k_in, projs, r
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
        """init_lora_weights="gaussian": A ~ N(0, (1/r)^2), B = 0 (padding rows / columns zero)."""
        for p, _ in self.projs:
            A = self.P.view(f"{self.name}.{p}.lora_A")
            A.zero_()
            A[:self.r] = (torch.randn(self.r, self.k_in, generator=gen, device=A.device) / self.r).to(BF16)
            self.P.view(f"{self.name}.{p}.lora_B").zero_()

    def refresh(self):
        """Derived operands after a parameter change: B_blk (dense block matrix), A_cat^T, B_blk^T."""
        if not self.single:
            for j, (p, d) in enumerate(self.projs):
                Bj = self.P.view(f"{self.name}.{p}.lora_B")
                dst = self.B_blk[d[1], j * self.rp:] if d[0] == "range" else self.B_blk[d[1], j * self.rp:]
                if d[0] == "range":
                    ops.copy2d(Bj, dst, Bj.shape[0], self.rp, self.rp, self.Rr)
                else:               # 16-row groups every 32 rows (gate / up interleave), starting at row d[1]
                    ops.copy2d(Bj, dst, Bj.shape[0], self.rp, self.rp, self.Rr, d_group=(16, 32 * self.Rr))
        ops.transpose(self.A_cat, out=self.A_catT)
        ops.transpose(self.B_blk, out=self.B_blkT)

    # y (base output, bf16) += 2 * (x A^T) B^T ; returns t for the backward
    def forward(self, x2d, y2d, t_out=None):
        t = ops.gemm_nt(x2d, self.A_cat, out=t_out)
        ops.gemm_nt(t, self.B_blk, alpha=2.0, residual=y2d, out=y2d)
        return t

    def backward(self, dy2d, x2d, t, dx2d):
        """Gradients of the pairs into the flat grad buffer; dx2d += contribution of the low-rank branch."""
        M = dy2d.shape[0]
        Mp = rup(M, 64)
        dt = ops.gemm_nt(dy2d, self.B_blkT, alpha=2.0)                       # [M, Rr]
        dtT, xT = ops.transpose(dt, ld_out=Mp), ops.transpose(x2d, ld_out=Mp)
        ops.gemm_nt(dtT, xT, out=self.gA_cat)                                # dA_cat = dt^T x
        dyT, tT = ops.transpose(dy2d, ld_out=Mp), ops.transpose(t, ld_out=Mp)
        for j, (p, d) in enumerate(self.projs):
            gB = self.P.g(f"{self.name}.{p}.lora_B")
            tj = tT[j * self.rp:(j + 1) * self.rp]
            if d[0] == "range":
                ops.gemm_nt(dyT[d[1]:d[2]], tj, alpha=2.0, out=gB)
            else:
                ops.gemm_nt(dyT[d[1]:d[1] + d[2]], tj, alpha=2.0, out=gB, a_group=(16, 32 * Mp))     # [n_j rows] by row-group addressing
        ops.gemm_nt(dt, self.A_catT, residual=dx2d, out=dx2d)                # dx += dt A_cat


class LoRAFinetune:
    def __init__(self, eng: E.VLAEngine, rank: int = 32, seed: int = 0):
        cfg = eng.cfg
        assert not getattr(eng, "fp8_frozen", False), "LoRA trains the backbone weights: the fp8 frozen-weight path does not apply"
        assert len(eng.vits) == 1 and cfg.n_img == 1, "LoRA path: single-backbone, single-image configuration"
        vc = cfg.vit[0]
        if vc.layerscale or vc.n_prefix:
            raise NotImplementedError("LoRA through LayerScale / prefix-token backbones (DINOv2) is not built")
        self.eng, self.cfg, self.dev, self.rank = eng, cfg, eng.device, rank
        self.vit, self.llm, self.head = eng.vits[0], eng.llm, eng.head
        eng.full_llm_backward = True
        c, v = cfg.llm, self.vit
        H, KV, dh, I, D, d = c.heads, c.kv_heads, c.dh, c.inter, c.d, v.cfg.d
        L: Dict[str, LoraLinear] = {}
        pre = "base_model.model."
        for i in range(len(v.blocks)):
            q = f"{pre}vision_backbone.featurizer.blocks.{i}."
            L[f"vit.{i}.qkv"] = LoraLinear(q + "attn", 3 * d, d, [("qkv", ("range", 0, 3 * d))], rank)
            L[f"vit.{i}.proj"] = LoraLinear(q + "attn", d, d, [("proj", ("range", 0, d))], rank)
            L[f"vit.{i}.fc1"] = LoraLinear(q + "mlp", v.mlp_pad, d, [("fc1", ("range", 0, v.mlp_pad))], rank)
            L[f"vit.{i}.fc2"] = LoraLinear(q + "mlp", d, v.mlp_pad, [("fc2", ("range", 0, d))], rank)
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
        specA, specB = [], []
        for l in L.values():
            a, b = l.spec()
            specA += a
            specB += b
        self.P = E.FlatParams(specA + specB, self.dev)
        gen = torch.Generator(device=self.dev).manual_seed(seed)
        for l in L.values():
            l.bind(self.P, self.dev)
            l.init_(gen)
        # W^T operands of the ViT / projector dX products (the LLM's exist already)
        z = lambda r_, c_: torch.empty(r_, c_, device=self.dev, dtype=BF16)
        for b in v.blocks:
            for k in ("wqkv", "wproj", "w1", "w2"):
                b[k + "T"] = ops.transpose(b[k])
        self.projT = {k: ops.transpose(w) for k, w in eng.proj.items() if k.endswith("weight")}
        self.refresh()
        self.step_count = 0
        self._key = None

    def refresh(self):
        for l in self.L.values():
            l.refresh()

    def lora_state_dict(self) -> Dict[str, torch.Tensor]:
        """peft's adapter key layout ('....q_proj.lora_A.weight' [r, in], '....lora_B.weight' [out, r]); rank padding and the
        ViT MLP's width padding removed."""
        out, v = {}, self.vit
        for l in self.L.values():
            for p, d in l.projs:
                A, B = self.P.view(f"{l.name}.{p}.lora_A")[:l.r], self.P.view(f"{l.name}.{p}.lora_B")[:, :l.r]
                if p == "fc1":
                    B = B[:v.cfg.mlp]
                if p == "fc2":
                    A = A[:, :v.cfg.mlp]
                out[f"{l.name}.{p}.lora_A.weight"], out[f"{l.name}.{p}.lora_B.weight"] = A, B
        return out

    # ------------------------------------------------------------------------------------------------ buffers
    def _alloc(self, B: int, S: int):
        if self._key == (B, S):
            return
        v, c, dev = self.vit, self.cfg.llm, self.dev
        e = lambda *s, dt=BF16: torch.empty(*s, device=dev, dtype=dt)
        nb, d, T = len(v.blocks), v.cfg.d, v.cfg.n_patches
        Mv, rp = B * T, rup(self.rank, 64)
        self.vX = e(nb + 1, Mv, d)
        self.vH1, self.vH2, self.vA, self.vXm = e(nb, Mv, d), e(nb, Mv, d), e(nb, Mv, d), e(nb, Mv, d)
        self.vS1, self.vS2 = e(nb, Mv, 2, dt=torch.float32), e(nb, Mv, 2, dt=torch.float32)
        self.vQKV, self.vLSE = e(nb, Mv, 3 * d), e(nb, B, v.cfg.heads, T, dt=torch.float32)
        self.vMpre, self.vMact = e(nb, Mv, v.mlp_pad), e(nb, Mv, v.mlp_pad)
        n, D, I = c.n_layers, c.d, c.inter
        M = B * S
        self.N1, self.N2, self.Hs = e(n, M, D), e(n, M, D), e(n, M, I)
        self.T = {}                                      # saved t = x A^T per LoRA Linear
        self.g_d, self.g_big, self.g_mid = e(Mv, d), e(Mv, v.mlp_pad), e(Mv, 3 * d)
        self.d_h = e(M, I)
        self.pj = {}
        self._key = (B, S)

    def _lin(self, key: str, x, W, bias, out=None):
        """Base Linear (frozen) + its LoRA branch; remembers t for the backward."""
        y = ops.gemm_nt(x, W, bias=bias, out=out)
        self.T[key] = self.L[key].forward(x, y)
        return y

    # ------------------------------------------------------------------------------------------------ forward
    def _ln(self, x, w, b, y, st, eps):
        d = x.shape[1]
        ops.N.check(ops._lib().vla_layernorm_fwd(ops._st(), ops._p(x), ops._p(w), ops._p(b), ops._p(y), ops._p(st), x.shape[0], d, d, d, eps), "layernorm_fwd")

    def forward(self, batch: Dict[str, torch.Tensor], noise: Optional[torch.Tensor] = None):
        eng, cfg, v, llm = self.eng, self.cfg, self.vit, self.llm
        eng._vision_begin(batch)
        B, S = eng.B, eng.S
        self._alloc(B, S)
        vc = v.cfg
        T, d = vc.n_patches, vc.d
        cols = ops.im2col_patch(batch["pixel_values"], 0, vc.patch, v.kpe)
        ops.gemm_nt(cols, v.wpe, bias=v.bpe, residual=v.pos, res_mod=T, out=self.vX[0])
        dhv = d // vc.heads
        for i, b in enumerate(v.blocks):
            x = self.vX[i]
            self._ln(x, b["n1w"], b["n1b"], self.vH1[i], self.vS1[i], vc.eps)
            qkv = self._lin(f"vit.{i}.qkv", self.vH1[i], b["wqkv"], b["bqkv"], out=self.vQKV[i]).view(B, T, 3 * d)
            dsc = ops._attn_desc(qkv[:, :, :d], qkv[:, :, d:2 * d], qkv[:, :, 2 * d:], self.vA[i].view(B, T, d), self.vLSE[i], None, False,
                                 dhv ** -0.5, vc.heads, vc.heads, dhv)
            ops.N.check(ops._lib().vla_attn_fwd(ops._st(), C.byref(dsc)), "attn_fwd")
            xm = self._lin(f"vit.{i}.proj", self.vA[i], b["wproj"], b["bproj"], out=self.vXm[i])
            ops.add_(xm, x)
            self._ln(xm, b["n2w"], b["n2b"], self.vH2[i], self.vS2[i], vc.eps)
            self._lin(f"vit.{i}.fc1", self.vH2[i], b["w1"], b["b1"], out=self.vMpre[i])
            ops.N.check(ops._lib().vla_gelu_fwd(ops._st(), ops._p(self.vMpre[i]), ops._p(self.vMact[i]), self.vMpre[i].numel()), "gelu_fwd")
            xo = self._lin(f"vit.{i}.fc2", self.vMact[i], b["w2"], b["b2"], out=self.vX[i + 1])
            ops.add_(xo, xm)
        # projector
        pj, feats = eng.proj, self.vX[len(v.blocks)]
        self.pj["in"] = feats
        self.pj["pre1"] = self._lin("proj.fc1", feats, pj["fc1.weight"], pj["fc1.bias"])
        self.pj["act1"] = ops.gelu_fwd(self.pj["pre1"])
        dst = eng.patches.view(-1, cfg.llm.d)
        if cfg.fused:
            self.pj["pre2"] = self._lin("proj.fc2", self.pj["act1"], pj["fc2.weight"], pj["fc2.bias"])
            self.pj["act2"] = ops.gelu_fwd(self.pj["pre2"])
            self._lin("proj.fc3", self.pj["act2"], pj["fc3.weight"], pj["fc3.bias"], out=dst)
        else:
            self._lin("proj.fc2", self.pj["act1"], pj["fc2.weight"], pj["fc2.bias"], out=dst)
        mm = eng._embed(batch)
        self._batch = batch
        # LLM
        c = cfg.llm
        D, H, KV, dh = c.d, c.heads, c.kv_heads, c.dh
        llm.fwd_begin(B, S, mm, 0)
        for i, Lw in enumerate(llm.layers):
            x = llm.HS[i].view(-1, D)
            llm._rms(x, Lw["n1"], self.N1[i], llm.R1[i])
            qkv = self._lin(f"llm.{i}.qkv", self.N1[i], Lw["wqkv"], Lw["bqkv"], out=llm.QKV[i])
            ops.rope_half_(qkv[:, :H * dh], llm.cos, llm.sin, S, H, dh)                    # RoPE after the low-rank delta
            ops.rope_half_(qkv[:, H * dh:(H + KV) * dh], llm.cos, llm.sin, S, KV, dh)
            llm._attn_fwd(qkv.view(B, S, -1), i, 0, B, S)
            x1 = self._lin(f"llm.{i}.o", llm.AO[i], Lw["wo"], None, out=llm.X1[i])
            ops.add_(x1, x)
            llm._rms(x1, Lw["n2"], self.N2[i], llm.R2[i])
            gu = self._lin(f"llm.{i}.gu", self.N2[i], Lw["wgu"], None, out=llm.GU[i])
            ops.swiglu_fwd(gu, out=self.Hs[i])
            xo = self._lin(f"llm.{i}.down", self.Hs[i], Lw["wd"], None, out=llm.HS[llm.out_slot(i)].view(-1, D))
            ops.add_(xo, x1)
        llm.fwd_final()
        return self.head.forward(llm.HS, eng.pos1, batch["proprio"], eng.Np, noise)

    # ------------------------------------------------------------------------------------------------ backward
    def _lin_bwd(self, key: str, dy, x, WT, dx_out=None):
        """dx = dy W (frozen base) + LoRA branch; LoRA gradients into the flat buffer."""
        dx = ops.gemm_nt(dy, WT, out=dx_out)
        self.L[key].backward(dy, x, self.T[key], dx)
        return dx

    def backward(self, pred, actions, gscale: float = 1.0):
        eng, head, llm, v, cfg = self.eng, self.head, self.llm, self.vit, self.cfg
        B, S, Np = eng.B, eng.S, eng.Np
        c = cfg.llm
        n, D, H, KV, dh, I = c.n_layers, c.d, c.heads, c.kv_heads, c.dh, c.inter
        M = B * S
        loss3, dpred = ops.l1_loss(pred, eng._to_bf16(actions), True, gscale)
        dHS = eng._dhs(0)
        head.backward(dpred, dHS, 0)
        llm.bwd_begin(dHS, 0)
        for i in range(n - 1, -1, -1):
            Lw, d, other = llm.layers[i], llm._d, llm._other
            if i < n - 1:
                ops.add_(d, dHS[i + 1].view(M, D))
            dh_ = self._lin_bwd(f"llm.{i}.down", d, self.Hs[i], Lw["wdT"], dx_out=self.d_h[:M])
            d_gu = ops.swiglu_bwd(dh_, llm.GU[i], out=llm.d_gu[:M])
            d_n = self._lin_bwd(f"llm.{i}.gu", d_gu, self.N2[i], Lw["wguT"], dx_out=llm.d_n[:M])
            d1 = ops.rmsnorm_bwd(d_n, llm.X1[i], Lw["n2"], llm.R2[i], dres=d, out=other)
            dao = self._lin_bwd(f"llm.{i}.o", d1, llm.AO[i], Lw["woT"], dx_out=llm.d_n[:M])
            q, k, vv = llm._attn_views(llm.QKV[i].view(B, S, -1))
            d_qkv = llm.d_qkv[:M]
            dq, dk, dv = llm._attn_views(d_qkv.view(B, S, -1))
            ops.attn_bwd(dao.view(B, S, -1), q, k, vv, llm.AO[i].view(B, S, -1), llm.LSE[i], H, KV, dh, True, llm.kmask, dq=dq, dk=dk, dv=dv,
                         rope=(llm.cos, llm.sin))
            d_n = self._lin_bwd(f"llm.{i}.qkv", d_qkv, self.N1[i], Lw["wqkvT"], dx_out=llm.d_n[:M])
            d_new = ops.rmsnorm_bwd(d_n, llm.HS[i].view(M, D), Lw["n1"], llm.R1[i], dres=d1, out=d)
            llm._d, llm._other = d_new, d1
        dX0 = llm.bwd_result().contiguous()
        dq = ops.action_query_grad(dX0, eng.pos0, Np, 0)
        ops.cast_f32_bf16(dq, out=head.P.g("action_queries"))
        # projector
        dp = torch.empty(B * Np, D, device=self.dev, dtype=BF16)
        for b in range(B):
            ops.copy2d(dX0[b, 1], dp[b * Np], Np, D, D, D)
        if cfg.fused:
            dh2 = self._lin_bwd("proj.fc3", dp, self.pj["act2"], self.projT["fc3.weight"])
            dpre = ops.gelu_bwd(dh2, self.pj["pre2"])
            dh1 = self._lin_bwd("proj.fc2", dpre, self.pj["act1"], self.projT["fc2.weight"])
        else:
            dh1 = self._lin_bwd("proj.fc2", dp, self.pj["act1"], self.projT["fc2.weight"])
        dpre1 = ops.gelu_bwd(dh1, self.pj["pre1"])
        dx = self._lin_bwd("proj.fc1", dpre1, self.pj["in"], self.projT["fc1.weight"])
        # ViT
        vc = v.cfg
        T, d_ = vc.n_patches, vc.d
        dhv = d_ // vc.heads
        lib, st, p = ops._lib(), ops._st, ops._p
        for i in range(len(v.blocks) - 1, -1, -1):
            b = v.blocks[i]
            dm = self._lin_bwd(f"vit.{i}.fc2", dx, self.vMact[i], b["w2T"], dx_out=self.g_big)
            dpre = ops.gelu_bwd(dm, self.vMpre[i])
            dh2 = self._lin_bwd(f"vit.{i}.fc1", dpre, self.vH2[i], b["w1T"], dx_out=self.g_d)
            dxm = torch.empty_like(dx)
            ops.N.check(lib.vla_layernorm_bwd(st(), p(dh2), p(self.vXm[i]), p(b["n2w"]), p(self.vS2[i]), p(dxm), None, None, dx.shape[0], d_, d_, d_, d_),
                        "layernorm_bwd")
            ops.add_(dxm, dx)
            da = self._lin_bwd(f"vit.{i}.proj", dxm, self.vA[i], b["wprojT"], dx_out=self.g_d)
            qkv = self.vQKV[i].view(B, T, 3 * d_)
            dqkv = self.g_mid.view(B, T, 3 * d_)
            ops.attn_bwd(da.view(B, T, d_), qkv[:, :, :d_], qkv[:, :, d_:2 * d_], qkv[:, :, 2 * d_:], self.vA[i].view(B, T, d_), self.vLSE[i],
                         vc.heads, vc.heads, dhv, False, None, dq=dqkv[:, :, :d_], dk=dqkv[:, :, d_:2 * d_], dv=dqkv[:, :, 2 * d_:])
            dh1 = self._lin_bwd(f"vit.{i}.qkv", self.g_mid, self.vH1[i], b["wqkvT"], dx_out=self.g_d)
            if i == 0:
                break                                    # below block 0 everything is frozen (Conv2d patch embedding, pos_embed)
            dxi = torch.empty_like(dx)
            ops.N.check(lib.vla_layernorm_bwd(st(), p(dh1), p(self.vX[i]), p(b["n1w"]), p(self.vS1[i]), p(dxi), None, None, dx.shape[0], d_, d_, d_, d_),
                        "layernorm_bwd")
            ops.add_(dxi, dxm)
            dx = dxi
        return loss3

    # ------------------------------------------------------------------------------------------------ update
    def optimizer_step(self, lr: float, beta1=0.9, beta2=0.999, eps=1e-8, wd=0.01):
        self.step_count += 1
        gscale, red = 1.0, self.eng.reducer
        if red is not None:
            red.reduce_async(self.P.grad, 0, None)
            red.reduce_async(self.head.P.grad, 0, None)
            red.wait()
            gscale = red.grad_scale
        P, HP = self.P, self.head.P
        ops.adamw_(P.data, P.grad, P.m, P.v, self.step_count, lr, beta1, beta2, eps, wd, gscale=gscale)
        ops.adamw_(HP.data, HP.grad, HP.m, HP.v, self.step_count, lr, beta1, beta2, eps, wd, gscale=gscale)
        self.head.dirty = True
        self.refresh()

    def train_step(self, batch, lr: float, noise=None):
        pred = self.forward(batch, noise)
        loss3 = self.backward(pred, batch["actions"])
        self.optimizer_step(lr)
        return loss3

    # ---- hipGraph replay (as FullFinetune.capture): a LoRA step is ~4000 launches, more host time than GPU time from Python ------
    def capture(self, batch: Dict[str, torch.Tensor], noise: Optional[torch.Tensor] = None, warmup: int = 2):
        """Forward + backward as ONE linear hipGraph on the static ``batch`` / ``noise`` buffers (copy new data into them before
        each replay); AdamW stays outside (host-side bias corrections), the derived-operand rebuild is a second small graph."""
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
        self._g_r = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._g_r, pool=pool, stream=self._cap_stream, capture_error_mode="thread_local"):
            self.refresh()
        torch.cuda.synchronize()

    def train_step_graphed(self, lr: float, beta1=0.9, beta2=0.999, eps=1e-8, wd=0.01):
        assert self.eng.reducer is None, "captured LoRA step: single process (the exchange of train_step is not in the graph)"
        self._g_step.replay()
        self.step_count += 1
        P, HP = self.P, self.head.P
        ops.adamw_(P.data, P.grad, P.m, P.v, self.step_count, lr, beta1, beta2, eps, wd)
        ops.adamw_(HP.data, HP.grad, HP.m, HP.v, self.step_count, lr, beta1, beta2, eps, wd)
        self._g_r.replay()
        return self._loss3

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
        if part == "vit":
            return self.vit.blocks[int(rest[0])], {"qkv": "wqkv", "proj": "wproj", "fc1": "w1", "fc2": "w2"}[rest[1]]
        if part == "proj":
            return self.eng.proj, rest[0] + ".weight"
        return self.llm.layers[int(rest[0])], {"qkv": "wqkv", "o": "wo", "gu": "wgu", "down": "wd"}[rest[1]]
