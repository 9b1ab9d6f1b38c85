"""GPU: data parallelism of the HIP path.  Two ranks share the one GPU of the test box over gloo
(RCCL refuses two ranks on one device): 2 ranks x B pairs must give the same VTC loss AND the same averaged gradients as 1 rank x 2B
pairs (run_pretrain_sparse.py:595-648 semantics; all-gather gradient in its exact "sum" mode), with the gradient exchange of the
second step overlapped with backward (FlatAdamW.overlap_backward: BERT + heads go on the wire before the ViT backward starts)."""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "dist_gpu_worker.py")


def _run(world, out_base, B, wire, backend="gloo", force=False):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(world):
        env = dict(os.environ, ALPRO_DIST_BACKEND=backend, ALPRO_FORCE_COLLECTIVES="1" if force else "0", RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, WORKER, "%s.%d.pt" % (out_base, r), str(B), wire], cwd=ROOT, env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=900)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
    print("".join(o for o in outs if "dist_backend" in o)[-400:])
    return [torch.load("%s.%d.pt" % (out_base, r)) for r in range(world)]


@pytest.mark.parametrize("wire,tol", [("fp32", 2e-4), ("bf16", 4e-2)])  # bf16 wire: each rank's value and the sum are rounded to 8 bits
def test_two_ranks_equal_one_rank_with_twice_the_batch(tmp_path, wire, tol):
    B = 2
    two = _run(2, str(tmp_path / "w2"), B, wire)
    one = _run(1, str(tmp_path / "w1"), B, "fp32")[0]
    assert two[0]["names"] == one["names"] == two[1]["names"]
    assert abs(two[0]["loss"] - one["loss"]) < 1e-5 * max(1.0, abs(one["loss"])), (two[0]["loss"], one["loss"])
    for r in two:
        assert r["on_wire_early"] >= 2, "the exchange did not start before backward returned (%d ranges)" % r["on_wire_early"]
        rel = (r["norms"] - one["norms"]).abs() / one["norms"].clamp_min(1e-6)
        for i, n in enumerate(one["names"]):   # key.bias gradients are exactly 0 in exact arithmetic (softmax shift invariance): fp32 noise only
            if n.endswith("attention.self.key.bias"):
                assert float(r["norms"][i]) < 1e-5
                rel[i] = 0.0
        worst = int(rel.argmax())
        assert float(rel.max()) < tol, (one["names"][worst], float(r["norms"][worst]), float(one["norms"][worst]))
        for n, g in one["grads"].items():
            err = float((r["grads"][n] - g).abs().max())
            assert err <= tol * max(float(g.abs().max()), 1e-6), (n, err, float(g.abs().max()))
    for n in one["grads"]:   # both ranks hold the same averaged gradient
        assert torch.equal(two[0]["grads"][n], two[1]["grads"][n]), n


@pytest.mark.parametrize("wire,tol", [("fp32", 2e-4), ("bf16", 2e-2)])
def test_one_rank_nccl_runs_every_collective(tmp_path, wire, tol):
    """RCCL had never executed this code.  A ONE-rank `nccl` (= RCCL) process group with ALPRO_FORCE_COLLECTIVES=1
    sends the training step through every collective branch the 8-GPU run takes -- broadcast_parameters, the differentiable feature
    all-gather (all_gather_into_tensor forward, reduce_scatter_tensor backward), FlatAdamW's async all_reduce handles on RCCL's
    stream launched from inside backward (grads_final) and `_finish_exchange`'s h.wait() stream hand-over, fp32 and bf16 wire -- and
    must reproduce the plain single-process step.  Since round 4 every reduction of the step has a fixed order (the CLS-row gradient's
    frame terms, dgamma / dbeta / bias column sums go through the reduction workspace, the squared norm and the embedding scatters
    likewise: alpro_amd.hip.set_deterministic), so with the fp32 wire the forced-collective step must be BITWISE equal to the plain one
    (one-rank all-gather / reduce-scatter / all-reduce are copies); the bf16 wire rounds the gradients once on the way."""
    B = 2
    forced = _run(1, str(tmp_path / "nccl1"), B, wire, backend="nccl", force=True)[0]
    plain = _run(1, str(tmp_path / "plain"), B, "fp32")[0]
    assert forced["backend"] == "nccl" and plain["backend"] == "none"
    assert forced["on_wire_early"] >= 2 and plain["on_wire_early"] == 0
    assert forced["names"] == plain["names"]
    assert abs(forced["loss"] - plain["loss"]) <= 1e-6 * max(1.0, abs(plain["loss"]))
    rel = (forced["norms"] - plain["norms"]).abs() / plain["norms"].clamp_min(1e-6)
    for i, n in enumerate(plain["names"]):
        if n.endswith("attention.self.key.bias"):
            rel[i] = 0.0
    assert float(rel.max()) < tol, (plain["names"][int(rel.argmax())], float(rel.max()))
    for n, g in plain["grads"].items():
        err = float((forced["grads"][n] - g).abs().max())
        assert err <= tol * max(float(g.abs().max()), 1e-6) + 1e-9, (n, err, float(g.abs().max()))
    if wire == "fp32":
        assert forced["loss"] == plain["loss"]
        assert torch.equal(forced["norms"], plain["norms"]), [n for i, n in enumerate(plain["names"]) if forced["norms"][i] != plain["norms"][i]][:8]
        for n, g in plain["grads"].items():
            assert torch.equal(forced["grads"][n], g), n


def test_two_runs_of_the_same_step_are_bitwise_equal(tmp_path):
    """two processes run the same two training steps (forward, hand-written backward, clip + AdamW) from the same
    seed; the loss, every parameter-gradient norm and the kept gradients must agree bit for bit -- the default path has no
    order-dependent fp32 reduction left."""
    a = _run(1, str(tmp_path / "a"), 2, "fp32")[0]
    b = _run(1, str(tmp_path / "b"), 2, "fp32")[0]
    assert a["loss"] == b["loss"]
    assert torch.equal(a["norms"], b["norms"]), [n for i, n in enumerate(a["names"]) if a["norms"][i] != b["norms"][i]][:8]
    for n, g in a["grads"].items():
        assert torch.equal(b["grads"][n], g), n
