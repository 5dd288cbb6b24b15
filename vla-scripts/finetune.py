#!/usr/bin/env python3
"""Entry point with the reference's launch line (run_finetune_libero_object_1gpu.sh:36-59):

    torchrun --standalone --nnodes 1 --nproc-per-node K vla-scripts/finetune.py --batch_size 32 --max_steps 100 ...

Every ``FinetuneConfig`` field of the reference (finetune.py:66-128) is accepted as ``--flag value``.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from vla_adapter_amd.finetune import finetune, parse_args  # noqa: E402

if __name__ == "__main__":
    out = finetune(parse_args())
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"done: {out['steps']} steps in {out['seconds']:.1f} s on {out['world']} GPU(s)")
