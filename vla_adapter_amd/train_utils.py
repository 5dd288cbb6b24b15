"""Mirror of prismatic/training/train_utils.py:8-41 (the two action masks).  On CUDA int64 labels the union the hot
path consumes comes from the native ``vla_action_mask`` kernel (see engine.VLAEngine.forward); these functions keep
the reference's names/semantics for logging code that wants the two masks separately (index arithmetic only)."""
import torch

from .constants import ACTION_DIM, ACTION_TOKEN_BEGIN_IDX, IGNORE_INDEX


def get_current_action_mask(token_ids: torch.Tensor) -> torch.Tensor:
    cumsum = torch.cumsum(token_ids != IGNORE_INDEX, dim=1)
    return ((1 <= cumsum) & (cumsum <= ACTION_DIM)) & (token_ids > ACTION_TOKEN_BEGIN_IDX)


def get_next_actions_mask(token_ids: torch.Tensor) -> torch.Tensor:
    cumsum = torch.cumsum(token_ids != IGNORE_INDEX, dim=1)
    return (cumsum > ACTION_DIM) & (token_ids > ACTION_TOKEN_BEGIN_IDX)


def all_actions_positions(labels: torch.Tensor, shift: int = 0):
    """Native path: (qidx, pos, count) of the union mask on labels[:, shift:] (int32 device tensors, no host sync)."""
    from . import ops
    return ops.action_mask(labels.contiguous(), shift)
