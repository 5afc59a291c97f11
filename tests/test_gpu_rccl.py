"""The N > 1 protocol on real RCCL.  The CPU suite covers the logic with world_size-2 gloo processes
(tests/test_shard_gloo.py, tests/test_train.py); what gloo cannot show is RCCL itself: the `nccl` backend bound to a device,
its dtype table (no 16-bit integers: the trajectory gather sends bytes), collectives on the GPU's streams.  A one-GPU box
gives one rank per process group -- every collective still runs through ProcessGroupNCCL -- and the driver's multi-GPU
run uses exactly these calls with N ranks."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _torchrun(script_args, extra_env=None, timeout=300):
    env = dict(os.environ)
    env.update(extra_env or {})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port())] + script_args
    return subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)


def test_collectives_on_rccl():
    r = _torchrun([os.path.join(ROOT, "tests", "rccl_worker.py")])
    assert r.returncode == 0 and "RCCL_OK world=1" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


def test_bench_protocol_on_rccl():
    """bench.py exactly as the driver launches it for N > 1 (torch.distributed.run, rank from the environment), with the
    process group forced on for the single rank: init_process_group("nccl", device_id), both barriers, max-over-ranks."""
    r = _torchrun([os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--trees", "512", "--skip-extras"],
                  {"MZ_BENCH_FORCE_DIST": "1"})
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 1 and out["scaling"] == "weak" and out["value"] > 0 and out["metric"] == "mcts_simulations_per_sec"
