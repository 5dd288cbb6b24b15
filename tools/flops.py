#!/usr/bin/env python3
"""CLI wrapper: prints the algorithmic FLOPs/sample of the hot path (see vla_adapter_amd/flops.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vla_adapter_amd import flops, engine
for k, v in flops.step_flops_per_sample(engine.config2()).items():
    print(f"{k:16s} {v / 1e9:10.1f} GF/sample")
