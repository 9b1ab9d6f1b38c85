"""GPU: bench.py's multi-process path end to end -- two ranks sharing the one GPU of the test box over gloo
(ALPRO_DIST_BACKEND=gloo; RCCL refuses two ranks on one device).  Exercises what the N>1 driver run exercises: rendezvous,
parameter broadcast, the differentiable feature all-gather inside forward, the flat gradient all-reduce, the untimed
per-kernel pass on EVERY rank (a rank-0-only extra step would hang in its collectives) and a clean exit."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("how", ["torchrun", "plain"])
def test_bench_two_ranks_gloo_shared_gpu(how):
    """how = torchrun: the driver's command line (python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...); how = plain:
    `python bench.py --gpus N ...` with no rendezvous environment -- bench.py re-executes itself under torch.distributed.run."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, ALPRO_DIST_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    tail = [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--batch", "2", "--frames", "2", "--steps", "1", "--warmup", "1"]
    if how == "torchrun":
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port)] + tail
    else:
        cmd = [sys.executable] + tail
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]          # exactly one JSON line, from rank 0
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["config"]["per_gpu_batch"] == 2
    assert "cpu_baseline" not in d                          # reported at N=1 only
    # the N > 1 line diagnoses its own gradient exchange -- one logged exchange per timed step on rank 0, the part that
    # went on the wire from inside backward (the overlapped exchange is the default), what was left for synchronize(), the exposed time
    x = d["exchange"]
    assert x is not None and x["exchanges"] == d["steps"] == 1
    assert x["overlap_backward"] is True and x["ranges_on_wire_early"] >= 1 and x["bytes_on_wire_early"] > 0
    assert x["bytes_on_wire_early"] + x["bytes_at_synchronize"] >= 4 * 230e6      # the whole flat gradient buffer (234 M trained parameters, fp32 wire) went out
    assert x["comm_exposed_ms"] >= 0.0 and x["comm_exposed_ms_max"] >= x["comm_exposed_ms"]
    assert x["cu_budget_while_in_flight"] in (None, 240)   # 256 - ALPRO_RCCL_CU_RESERVE while all-reduces launched from backward are in flight
    assert d["world_size"] == 2 and d["dist_backend"] == "gloo"
