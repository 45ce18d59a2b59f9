"""The multi-GPU path ON HARDWARE, through the C ABI (hsrans_comm_create / hsrans_sharded_create / hsrans_decode_sharded): runs on every
visible GPU — on a one-GPU box with a world of one rank (the communicator is made on RCCL, the decode and status paths run; there
is nobody to exchange with), on a multi-GPU box with the real exchange.

N = min(visible GPUs, 8) ranks, one process per GPU over RCCL (torch.distributed backend "nccl"), started through
`python -m torch.distributed.run` — the launcher the driver uses for bench.py — as a child process.  Every rank runs
tests/rccl_worker.py: ShardedDecoder legs none / all / root with parts 1 and 4 on mt_, raw and block_ streams, every receiver's
bytes compared with the CPU oracle.  The gloo twin (tests/test_sharded_gloo.py) covers the same host logic without GPUs; what
only this test can see is the exchange's ordering against the decode kernels on RCCL's stream (sharded.pipelined_gather)."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.gpu
def test_sharded_decode_over_rccl_on_every_visible_gpu():
    n = torch.cuda.device_count()  # (counting devices does not initialise the GPU in this process)
    if n < 1:
        pytest.fail("GPU test selected but no GPU is visible")
    n = min(n, 8)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "rccl_worker.py")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, cwd=ROOT, env=env)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["world"] == n and d["backend"] == "nccl" and d["failed"] == [] and d["checks"] == 3 * 4 * 2 * 3


def test_the_rccl_test_is_wired_to_the_device_count():
    """CPU-side check that the hardware test exists, is marked gpu and keys on the number of visible devices (so that the first
    multi-GPU box meets sharded.pipelined_gather in pytest, not in the bench)."""
    src = open(__file__).read()
    assert "torch.cuda.device_count()" in src and "@pytest.mark.gpu" in src and "rccl_worker.py" in src
    worker = open(os.path.join(ROOT, "tests", "rccl_worker.py")).read()
    assert 'init_process_group("nccl"' in worker and "ShardedDecoder" in worker and "Oracle" in worker and "uses_c_abi" in worker
