"""LoRA fine-tune step (vla-scripts/finetune.py:832-844): see ``trainers.LoRAFinetune``.
(Import location of rounds 1-2, kept for callers.)"""
from .trainers import LoRAFinetune, LoraLinear  # noqa: F401
