"""Full fine-tune step (BASELINE.json configs[3]; vla-scripts/finetune.py:846-849, 903-910): see ``trainers.FullFinetune``.
(Import location of rounds 1-2, kept for callers.)"""
from .trainers import FullFinetune, _Slot  # noqa: F401
