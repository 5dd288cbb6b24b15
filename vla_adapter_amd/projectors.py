"""Mirror of prismatic/models/projectors.py:6-24 on the native kernels (forward; training goes through VLAEngine)."""
from typing import Dict

import torch

from . import ops
from .ops import ACT_NONE, BF16


class ProprioProjector:
    """fc1 (proprio_dim -> llm_dim), GELU(erf), fc2 (llm_dim -> llm_dim).  State-dict keys: fc1/fc2 .weight/.bias."""

    def __init__(self, llm_dim: int, proprio_dim: int, device="cuda"):
        self.llm_dim, self.proprio_dim, self.device = llm_dim, proprio_dim, device
        z = lambda *s: torch.zeros(*s, device=device, dtype=BF16)
        self.params = {"fc1.weight": z(llm_dim, proprio_dim), "fc1.bias": z(llm_dim), "fc2.weight": z(llm_dim, llm_dim), "fc2.bias": z(llm_dim)}

    def state_dict(self) -> Dict[str, torch.Tensor]:
        return {k: v.clone() for k, v in self.params.items()}

    def load_state_dict(self, sd: Dict[str, torch.Tensor]):
        for k in self.params:
            self.params[k] = sd[k.replace("module.", "")].to(self.device, BF16).contiguous()   # DDP prefix stripped (finetune.py:132-154)
        self._version = getattr(self, "_version", 0) + 1

    def forward(self, proprio: torch.Tensor) -> torch.Tensor:
        B = proprio.shape[0]
        x = torch.zeros(B, 64, device=self.device, dtype=BF16)                 # K padded to the GEMM's 64
        x[:, :self.proprio_dim] = proprio.reshape(B, -1).to(BF16)
        w1 = torch.zeros(self.llm_dim, 64, device=self.device, dtype=BF16)
        w1[:, :self.proprio_dim] = self.params["fc1.weight"]
        h = ops.gelu_fwd(ops.gemm_nt(x, w1, bias=self.params["fc1.bias"], act=ACT_NONE))
        return ops.gemm_nt(h, self.params["fc2.weight"], bias=self.params["fc2.bias"])

    __call__ = forward
