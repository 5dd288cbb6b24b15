#!/usr/bin/env python3
"""Per-tensor table of the HIP head against a reference-run fixture (tests/golden/head_bf16_<case>.npz): native vs the
reference's bf16 run, vs its fp32 run, and the reference's own bf16-vs-fp32 gap.  python tools/diag_head_fixture.py <case>"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_gen as GG  # noqa: E402
import test_head_fixture_gpu as T  # noqa: E402

case = sys.argv[1] if len(sys.argv) > 1 else "orig1_d128_kt64"
z, pred, grads, taps, dx = T._run_native(case)
rel = T.rel
print(f"{case}: actions native-vs-ref16 {rel(pred, z['out_bf16']):.3e}  native-vs-fp32 {rel(pred, z['out_fp32']):.3e}  ref16-vs-fp32 {rel(torch.as_tensor(z['out_bf16']), z['out_fp32']):.3e}")
for k in GG.grad_keys(case) + GG.weight_grad_rows(case) + ["proprio.fc2.bias"]:
    r16, r32 = torch.as_tensor(z[f"g_bf16.{k}"]), torch.as_tensor(z[f"g_fp32.{k}"])
    g = grads[k].float().cpu().reshape(-1, r16.shape[-1])[:r16.reshape(-1, r16.shape[-1]).shape[0]].reshape(r16.shape)
    print(f"{k:46s} |g| {r32.norm().item():.3e}  native-ref16 {rel(g, r16):.3e}  native-fp32 {rel(g, r32):.3e}  ref16-fp32 {rel(r16, r32):.3e}")
print("dx", rel(dx, z["dx_bf16"]), rel(dx, z["dx_fp32"]), rel(torch.as_tensor(z["dx_bf16"]), z["dx_fp32"]))
