"""Two ranks on ONE GPU through gloo (RCCL refuses two ranks per device): functional rehearsal of the captured
data-parallel step - deferred update, early head-gradient exchange, vision lead - ending with identical parameters
on both ranks (tools/ddp_rehearsal.py asserts it)."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("algo", ["allreduce", "rs_ag"])
def test_two_rank_captured_step_keeps_ranks_in_sync(algo):
    """Ranks end bit-identical AND equal to a single-process run on the summed gradients (tools/ddp_rehearsal.py), with the
    exchange as bucketed all-reduce or as reduce-scatter + all-gather."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, VLA_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", VLA_DDP_ALGO=algo)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), os.path.join("tools", "ddp_rehearsal.py")],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.count("ranks-in-sync-ok") == 2, r.stdout[-3000:]


def test_single_rank_rccl_exchange_is_the_identity():
    """The RCCL path itself on one GPU: a one-rank "nccl" group with the reducer forced to issue its collectives between
    the captured segment graphs (tools/rccl_single_rank.py) - same losses and parameters as without the exchange."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join("tools", "rccl_single_rank.py")], cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "rccl-single-rank-ok" in r.stdout, r.stdout[-3000:]


@pytest.mark.parametrize("mode,captured", [("full", "0"), ("full", "1"), ("lora", "1")])
def test_two_rank_backbone_trainers_exchange_during_the_backward(mode, captured):
    """Full fine-tune (BASELINE configs[3]: "grad-bucket overlap") and LoRA on the plumbing-size DINOv2 + SigLIP geometry, two gloo
    ranks on one GPU: gradient ranges go to the exchange as the backward finishes them (eager and as a chain of captured segment
    graphs); ranks end bit-identical and equal to the single-process run on the summed gradients (tools/ddp_rehearsal_trainers.py)."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, VLA_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", VLA_TRAINER=mode, VLA_CAPTURED=captured)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), os.path.join("tools", "ddp_rehearsal_trainers.py")],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.count("ranks-in-sync-ok") == 2, r.stdout[-3000:]


@pytest.mark.parametrize("algo", ["allreduce", "rs_ag"])
def test_bench_two_ranks_prints_the_contract_line(algo):
    """The driver's scaling command (`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`) with N = 2 on ONE GPU
    through gloo: the plumbing the first real RCCL run depends on - rendezvous from the environment, per-rank batches, the barrier /
    max-over-ranks timing, rank 0 printing ONE JSON line whose `value` is the whole-job rate.  Small batch: this
    checks the contract, not a throughput."""
    import json
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, VLA_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    B, K = 4, 3
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), "bench.py", "--gpus", "2", "--steps", str(K), "--warmup", "1", "--batch", str(B),
                        "--no-cpu-baseline", "--no-full-backward", "--ddp-algo", algo],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, f"rank 0 prints exactly one JSON line, got {len(lines)}: {r.stdout[-2000:]}"
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == K and d["warmup"] == 1 and d["unit"] == "samples/s" and d["scaling"] == "weak"
    assert d["config"]["global_batch"] == 2 * B and d["config"]["per_gpu_batch"] == B and d["config"]["parallelism"] == "dp2"
    assert algo in d["config"]["gradient_exchange"]
    assert abs(d["value"] - 2 * B / (d["ms_per_step"] * 1e-3)) <= 0.01 * d["value"], "value = samples of ALL ranks / max-over-ranks time"
    assert abs(d["samples_per_s_per_gpu"] * 2 - d["value"]) <= 0.01 * d["value"]
    assert d["roofline"]["frac"] > 0 and d["cpu_baseline"] is None
