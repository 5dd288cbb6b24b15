"""Mirror of prismatic/models/action_heads.py:21-81 (L1RegressionActionHead) on the native engine.

``predict_action`` keeps the reference signature: it takes the regrouped ``[B, 25, K+64, D]`` tensor built at
vla-scripts/finetune.py:396-409 (or modeling_prismatic.py:848-862).  Internally that tensor is viewed as a stack of
per-layer "sequences" of K task rows followed by the 64 action rows, which is exactly the in-place layout
engine.Head reads from the LLM's hidden-state buffer during training (no regroup copy on the hot path).
"""
from typing import Dict, Optional

import torch

from . import engine as E
from .ops import BF16


class L1RegressionActionHead:
    def __init__(self, input_dim=4096, hidden_dim=4096, action_dim=7, num_task_tokens=512, use_pro_version=False, device="cuda",
                 num_blocks: int = 24):
        assert input_dim == hidden_dim, "the reference always passes llm_dim for both (finetune.py:884-896)"
        self.num_task_tokens, self.action_dim, self.hidden_dim = num_task_tokens, action_dim, hidden_dim
        cfg = E.VLACfg(llm=E.LLMCfg(d=hidden_dim), num_blocks=num_blocks, action_dim=action_dim, pro=bool(use_pro_version))
        self.head = E.Head(cfg, device)
        self.device = device

    # reference state-dict layout ('model.mlp_resnet_blocks.N....'); checkpoints: action_head--{step}_checkpoint.pt
    def state_dict(self) -> Dict[str, torch.Tensor]:
        return self.head.head_state_dict()

    def load_state_dict(self, sd: Dict[str, torch.Tensor], proprio_sd: Optional[Dict[str, torch.Tensor]] = None):
        sd = {k.replace("module.", "", 1) if k.startswith("module.") else k: v for k, v in sd.items()}
        pp = proprio_sd if proprio_sd is not None else {k: v for k, v in self.head.proprio_views().items()}
        self.head.load_state_dicts(sd, pp)
        self._version = getattr(self, "_version", 0) + 1          # predict_action reloads the engine's copy when this moves

    def predict_action(self, actions_hidden_states: torch.Tensor, proprio=None, proprio_projector=None, phase="Inference",
                       noise: Optional[torch.Tensor] = None) -> torch.Tensor:
        """actions_hidden_states [B, n_states >= num_blocks+1, K+64, D] -> actions [B, chunk, action_dim] (bf16).
        phase == "Training" draws the N(0, 0.02^2) input perturbation of action_heads.py:14-17, 69-72 unless ``noise``
        ([chunk, action_dim*D]) is given."""
        assert proprio is not None and proprio_projector is not None, "the reference head dereferences both (action_heads.py:53-54)"
        B, _, KA, D = actions_hidden_states.shape
        K = self.num_task_tokens
        assert KA == K + E.NUM_TOKENS, f"expected {K}+64 rows per layer, got {KA}"
        if proprio_projector is not None and hasattr(proprio_projector, "params"):
            for k, v in self.head.proprio_views().items():
                v.copy_(proprio_projector.params[k])
            self.head.dirty = True
        hs = actions_hidden_states.to(BF16).permute(1, 0, 2, 3).contiguous()          # [n, B, K+64, D]
        pos1 = torch.arange(E.NUM_TOKENS, device=hs.device, dtype=torch.int32)[None].expand(B, -1).contiguous()
        if phase == "Training" and noise is None:
            noise = torch.randn(self.head.cfg.chunk, self.action_dim * D, device=hs.device) * 0.02
        return self.head.forward(hs, pos1, proprio.reshape(B, -1), K, noise if phase == "Training" or noise is not None else None)
