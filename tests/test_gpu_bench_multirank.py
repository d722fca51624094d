"""bench.py's N > 1 path as real PROCESSES (torch.distributed.run), on the one GPU of the test
box: every rank on device 0, gloo with the halo rows staged through host memory
(AMHIP_BENCH_ONE_GPU=1 -- RCCL refuses two ranks per device; on a multi-GPU node the same code
runs over backend "nccl" = RCCL).  --verify gathers the cloud and every window on rank 0 and
compares with ONE full-map DSM there: windows == the single-GPU result."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(nproc, workload, extra=()):
    env = dict(os.environ, AMHIP_BENCH_ONE_GPU="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", str(nproc), "--workload", workload,
           "--steps", "2", "--warmup", "1", "--verify"] + list(extra)
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       universal_newlines=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_two_ranks_strip_of_windows():
    d = _run(2, "small")
    assert d["n_gpus"] == 2 and "REHEARSAL" in d["data"]
    v = d["verify"]
    assert v["windows"] == 2 and v["pass"], v
    assert d["config"]["parallelism"].startswith("one map, 2 x 1 windows")


def test_four_ranks_two_by_two_windows_with_diagonal_neighbours():
    d = _run(4, "small4")
    v = d["verify"]
    assert v["windows"] == 4 and v["pass"], v
    assert d["scaling"] == "strong" and d["config"]["parallelism"].startswith("one map, 2 x 2 windows")
