"""The N > 1 execution path of the training step on the driver's one-GPU box, under RCCL.

``bench.py --mode train --force-dist-path`` launched by ``torch.distributed.run`` with ONE rank: ``init_process_group("nccl")``
(RCCL on ROCm), thread-local graph capture of forward + backward + gradient pack, ``StaticGradAllReducer.all_reduce`` through
the communicator, fused AdamW -- reference call sites main.py:94-103,144 (accelerate's DDP wrapper), util/engine.py:58.
The step must report ``backend: nccl`` and the same loss as the single-process step (same seeds, same data, one rank:
the all-reduce is an identity).  Two ranks on one device are refused by RCCL ("duplicate GPU"), so the cross-process
exchange itself stays covered by the 2-rank gloo tests (tests/test_data_parallel_cpu.py).
"""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _last_json(text):
    for line in reversed(text.strip().splitlines()):
        line = line.strip()
        if line.startswith("{") and line.endswith("}"):
            return json.loads(line)
    raise AssertionError("no JSON line in:\n" + text[-2000:])


def test_training_step_through_a_one_rank_rccl_group_matches_the_single_process_step():
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    common = [os.path.join(ROOT, "bench.py"), "--gpus", "1", "--mode", "train", "--steps", "3", "--warmup", "1"]
    solo = subprocess.run([sys.executable] + common, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert solo.returncode == 0, solo.stderr[-2000:]
    a = _last_json(solo.stdout)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    dist = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1",
                           "--master-addr", "127.0.0.1", "--master-port", str(port)] + common + ["--force-dist-path"],
                          capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert dist.returncode == 0, dist.stderr[-2000:]
    b = _last_json(dist.stdout)
    # ... and the eager step (no hipGraph): the replayed forward + backward must train exactly like it.  (Until round 4 it
    # did not: a hipMemsetAsync in the MSDA backward launcher and the framework's multi-block reductions -- which clear their
    # semaphores with one -- are not reproduced by a replayed graph on this stack; both are gone from the step.)
    eager = subprocess.run([sys.executable] + common + ["--no-graph"], capture_output=True, text=True, timeout=600, cwd=ROOT,
                           env=env)
    assert eager.returncode == 0, eager.stderr[-2000:]
    c = _last_json(eager.stdout)
    assert c["config"]["execution"].startswith("eager")
    assert b["config"]["backend"] == "nccl" and b["config"]["world_size"] == 1 and b["n_gpus"] == 1
    assert "hipGraph" in b["config"]["execution"] or "graph" in b["config"]["execution"].lower()
    assert a["config"]["backend"].startswith("none")
    # same data, same seeds, one rank: the reduced gradients are the local ones -- the loss of the third step from the
    # initial state agrees to fp32 reassociation on all three execution paths
    for other in (b, c):
        assert abs(a["loss"] - other["loss"]) <= 1e-4 * max(1.0, abs(a["loss"])), (a["loss"], b["loss"], c["loss"])
    # the timed steps start from the initial state: three AdamW updates at lr 1e-4 move the loss a long way (5.38 -> 3.55
    # on the benchmark's data), so agreement to 1e-4 is agreement of every gradient that matters
    assert a["loss"] < 4.0
    assert b["config"]["grad_bytes"] == a["config"]["grad_bytes"] > 0
