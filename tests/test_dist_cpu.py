"""CPU, world_size 2 over gloo: the data-parallel pieces of the path (alpro_amd/dist.py) -- differentiable
all-gather (forward order = rank order; backward = reduce across ranks + own slice, "average" = Horovod 0.19's
HorovodAllgather.backward used at alpro_models.py:110-111, "sum" = the exact full-batch gradient), bucketed gradient all-reduce,
parameter broadcast, and the VTC loss identity "2 ranks x B pairs == 1 rank x 2B pairs" that holds in "sum" mode."""
import os
import sys
import socket

import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _vtc(video_feat, text_feat, temp, rank, allgather):
    b = video_feat.shape[0]
    gv, gt = allgather(video_feat), allgather(text_feat)
    sim_v2t = video_feat @ gt.t() / temp
    sim_t2v = text_feat @ gv.t() / temp
    tgt = torch.zeros_like(sim_v2t)
    tgt[:, b * rank:b * (rank + 1)] = torch.eye(b)
    lv = -torch.sum(torch.log_softmax(sim_v2t, 1) * tgt, 1).mean()
    lt = -torch.sum(torch.log_softmax(sim_t2v, 1) * tgt, 1).mean()
    return (lv + lt) / 2


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from alpro_amd import dist
    dist.init(backend="gloo")
    assert dist.size() == world and dist.rank() == rank and dist.local_rank() == rank
    torch.manual_seed(0)
    B, D = 3, 16
    full_v = torch.nn.functional.normalize(torch.randn(world * B, D), dim=-1)
    full_t = torch.nn.functional.normalize(torch.randn(world * B, D), dim=-1)
    v = full_v[rank * B:(rank + 1) * B].clone().requires_grad_(True)
    t = full_t[rank * B:(rank + 1) * B].clone().requires_grad_(True)
    # forward: gathered rows in rank order
    g = dist.allgather(v)
    assert torch.allclose(g, full_v)
    # loss averaged over ranks == single-process loss on the concatenated batch; in "sum" mode the same holds for the gradients
    assert dist.allgather_grad_mode() == "average"          # the default follows the reference's Horovod
    dist.set_allgather_grad_mode("sum")
    loss = _vtc(v, t, 0.07, rank, dist.allgather)
    loss.backward()
    fv, ft = full_v.clone().requires_grad_(True), full_t.clone().requires_grad_(True)
    ref = _vtc(fv, ft, 0.07, 0, lambda x: x)
    ref.backward()
    lsum = loss.detach().clone()
    torch.distributed.all_reduce(lsum)
    assert torch.allclose(lsum / world, ref.detach(), atol=1e-6), (lsum / world, ref)
    # d(mean over ranks of loss_r)/dv = fv.grad rows: each rank's backward already sums the other ranks' contributions
    assert torch.allclose(v.grad / world, fv.grad[rank * B:(rank + 1) * B], atol=1e-6)
    assert torch.allclose(t.grad / world, ft.grad[rank * B:(rank + 1) * B], atol=1e-6)
    # "average" (Horovod 0.19 HorovodAllgather.backward = allreduce(grad, average=True).narrow(own rows)): the gradient that reaches
    # a rank's features through the GATHERED copies is the mean over ranks; the direct path (features as queries) is untouched.
    # Restated per rank with the gathered tensors as independent leaves.
    dist.set_allgather_grad_mode("average")
    v.grad = t.grad = None
    _vtc(v, t, 0.07, rank, dist.allgather).backward()
    direct_v = direct_t = None
    via_gv, via_gt = torch.zeros_like(full_v), torch.zeros_like(full_t)
    for q in range(world):
        vq = full_v[q * B:(q + 1) * B].clone().requires_grad_(True)
        tq = full_t[q * B:(q + 1) * B].clone().requires_grad_(True)
        gv, gt = full_v.clone().requires_grad_(True), full_t.clone().requires_grad_(True)
        leaves = iter([gv, gt])                               # allgather order: video, then text (alpro_models.py:110-111)
        _vtc(vq, tq, 0.07, q, lambda x: next(leaves)).backward()
        via_gv += gv.grad
        via_gt += gt.grad
        if q == rank:
            direct_v, direct_t = vq.grad, tq.grad
    assert torch.allclose(v.grad, direct_v + via_gv[rank * B:(rank + 1) * B] / world, atol=1e-6)
    assert torch.allclose(t.grad, direct_t + via_gt[rank * B:(rank + 1) * B] / world, atol=1e-6)
    # bucketed gradient all-reduce (average), skipping parameters without gradients
    ps = [torch.nn.Parameter(torch.zeros(5, 7)), torch.nn.Parameter(torch.zeros(11)), torch.nn.Parameter(torch.zeros(3))]
    ps[0].grad = torch.full((5, 7), float(rank + 1))
    ps[1].grad = torch.arange(11, dtype=torch.float32) * (rank + 1)
    sent = dist.allreduce_grads_(ps, bucket_bytes=64)
    assert sent == (35 + 11) * 4
    assert torch.allclose(ps[0].grad, torch.full((5, 7), 1.5)) and torch.allclose(ps[1].grad, torch.arange(11, dtype=torch.float32) * 1.5)
    assert ps[2].grad is None
    # a parameter that received a gradient on ONE rank only (a head only some batches use) takes part on every rank -- zeros from the others,
    # as the reference's zero_none_grad would have filled in -- instead of shifting the positional buckets; stride-0 placeholders
    # (alpro_amd.optim.zero_none_grad) count as "no gradient"
    from alpro_amd.optim import placeholder_grad
    qs = [torch.nn.Parameter(torch.zeros(6)), torch.nn.Parameter(torch.zeros(9)), torch.nn.Parameter(torch.zeros(4))]
    qs[0].grad = torch.full((6,), float(rank + 1))
    if rank == 1:
        qs[1].grad = torch.full((9,), 8.0)
    else:
        qs[1].grad = placeholder_grad(qs[1])
    sent = dist.allreduce_grads_(qs, bucket_bytes=1 << 20)
    assert sent == (6 + 9) * 4 and qs[2].grad is None
    assert torch.allclose(qs[0].grad, torch.full((6,), 1.5)) and torch.allclose(qs[1].grad, torch.full((9,), 4.0)) and any(qs[1].grad.stride())
    # broadcast of parameters from rank 0
    lin = torch.nn.Linear(4, 4)
    with torch.no_grad():
        lin.weight.fill_(float(rank + 7))
    dist.broadcast_parameters(lin)
    assert float(lin.weight[0, 0]) == 7.0
    dist.barrier()
    _hvd_facade_checks(rank, world)
    _amp_without_synchronize_checks(rank, world)
    _gradient_accumulation_checks(rank, world)
    out.put((rank, "ok"))


def _gradient_accumulation_checks(rank, world):
    """the unmodified drivers call optimizer.synchronize() after EVERY micro-step (run_pretrain_sparse.py:596-601;
    gradient_accumulation_steps = 2 in config_release/msrvtt_qa.json, msvd_qa.json, pretrain_prompter.json).  Each call has to exchange what
    the backward before it added: after two micro-steps every rank must hold avg(g1) + avg(g2), with nothing left in flight -- before the flat
    buffers exist (first step: separate gradient tensors), after (_build), and with ranges launched from inside backward (overlap)."""
    import alpro_amd.compat
    if alpro_amd.compat.PATH not in sys.path:
        sys.path.insert(0, alpro_amd.compat.PATH)
    from horovod import torch as hvd
    from alpro_amd import dist, hip
    from alpro_amd.optim import FlatAdamW
    hip.set_option = lambda *a, **k: None
    hip.set_stream_option = lambda *a, **k: None
    mean = lambda f: sum(f(r) for r in range(world)) / world     # noqa: E731

    def micro(ps, k, final=None):
        """one micro-step through autograd: d loss / d p = (rank + 1) * k for ps[0], 10 * that for ps[1]"""
        loss = (ps[0].sum() + 10.0 * ps[1].sum()) * float((rank + 1) * k)
        loss.backward()
        if final is not None:
            dist.grads_final(params=final)

    for built, overlap in ((False, False), (True, False), (True, True)):
        ps = [torch.nn.Parameter(torch.zeros(5, 7)), torch.nn.Parameter(torch.zeros(11))]
        opt = FlatAdamW(ps, lr=0.0, overlap_backward=overlap)
        opt.record_exchange = True
        fac = hvd.DistributedOptimizer(opt)
        if built:
            for p in ps:
                p.grad = torch.zeros_like(p)
            assert opt._build()
        micro(ps, 1, final=[ps[0]] if overlap else None)
        assert (len(opt._inflight) >= 1) == overlap
        fac.synchronize()
        want1 = mean(lambda r: float(r + 1))
        assert torch.allclose(ps[0].grad, torch.full((5, 7), want1)) and torch.allclose(ps[1].grad, torch.full((11,), 10 * want1))
        fac.synchronize()                                        # a repeated call without a backward in between changes nothing
        assert torch.allclose(ps[0].grad, torch.full((5, 7), want1))
        micro(ps, 2, final=[ps[0]] if overlap else None)         # second micro-step accumulates on top of the averaged first
        fac.synchronize()
        want2 = want1 + mean(lambda r: 2.0 * (r + 1))
        assert torch.allclose(ps[0].grad, torch.full((5, 7), want2)), (built, overlap, ps[0].grad.flatten()[:2], want2)
        assert torch.allclose(ps[1].grad, torch.full((11,), 10 * want2))
        assert not opt._inflight and not opt._reduced and opt._pre_synced == "avg" and opt._sync_is_current()
        if built:   # the exchange diagnostics the N > 1 bench line prints: one record per finished exchange
            st = opt.exchange_stats()
            assert st["exchanges"] == 2 and st["overlap_backward"] is overlap and st["comm_exposed_ms"] >= 0.0 and st["timer"].startswith("host clock")
            assert st["ranges_on_wire_early"] == (1.0 if overlap else 0.0) and st["bytes_on_wire_early"] == (36 * 4 if overlap else 0)   # ps[0]: 35 -> 36 elements (16-byte views)
            assert st["bytes_on_wire_early"] + st["bytes_at_synchronize"] == (36 + 12) * 4
        micro(ps, 3)                                             # a micro-step NOT followed by synchronize(): step() must notice
        assert not opt._sync_is_current()
        opt.zero_grad()
        assert opt._pre_synced is None
    # local gradients on top of an exchanged SUM cannot be repaired by a collective: loud, not silent
    ps = [torch.nn.Parameter(torch.zeros(3))]
    opt = FlatAdamW(ps, lr=0.0, overlap_backward=False)
    ps[0].sum().backward()
    opt.synchronize(average=False)
    ps[0].sum().backward()
    try:
        opt.synchronize(average=False)
        raise AssertionError("expected a RuntimeError")
    except RuntimeError as e:
        assert "already exchanged SUM" in str(e)
    dist.barrier()


def _amp_without_synchronize_checks(rank, world):
    """`with amp.scale_loss(loss, opt) as s: s.backward()` on a bare FlatAdamW -- no optimizer.synchronize() inside the
    block, the usage config.check_backward_precision's message advertises.  The all-reduces launched from inside backward (grads_final) are
    still writing the flat gradient buffer when the block exits; unscale_ must finish the exchange (remaining ranges out, handles waited
    for, the 16-bit wire copy back) BEFORE it multiplies by 1/S, and tell step() that the sums are already there.  Before the fix the early
    range came out scaled-then-overwritten (wire copy) or raced, the late range unscaled-then-summed: both wrong by a factor S."""
    from alpro_amd import amp, config as rt, dist, hip
    from alpro_amd.optim import FlatAdamW
    hip.set_option = lambda *a, **k: None          # (the CU reservation talks to the GPU library)
    prev = rt.compute_dtype()
    rt.set_compute_dtype("fp16")
    try:
        for wire in (None, torch.bfloat16):
            ps = [torch.nn.Parameter(torch.zeros(64, 33)), torch.nn.Parameter(torch.zeros(130))]
            opt = FlatAdamW(ps, lr=0.0, overlap_backward=True, wire_dtype=wire)
            for p in ps:
                p.grad = torch.zeros_like(p)
            assert opt._build()
            S = amp.scaler_for(opt).to("cpu").loss_scale()      # (attached by the constructor under fp16 operands: 2^16)
            assert S == 65536.0
            loss = torch.zeros((), requires_grad=True)
            with amp.scale_loss(loss, opt) as scaled:
                assert float(scaled) == 0.0 and opt._grads_scaled
                with torch.no_grad():               # what a hand-written backward leaves behind: S * dL/dw in the flat views ...
                    ps[0].grad.fill_(S * (rank + 1))
                    ps[1].grad.fill_(S * 10 * (rank + 1))
                dist.grads_final(params=[ps[0]])    # ... and the first parameter's range already on the wire
                assert len(opt._inflight) >= 1 and len(opt._reduced) >= 1
            assert not opt._inflight and not opt._reduced and opt._pre_synced == "sum" and opt._grads_scaled is False
            tot = sum(r + 1 for r in range(world))
            assert torch.equal(ps[0].grad, torch.full((64, 33), float(tot))), (wire, ps[0].grad.flatten()[:3])
            assert torch.equal(ps[1].grad, torch.full((130,), 10.0 * tot)), (wire, ps[1].grad[:3])
            # the same through the hvd.DistributedOptimizer facade.  Its own _synced flag knows nothing of the exchange
            # unscale_ just finished; the synchronize() its step() (or the driver) issues next must NOT all-reduce the flat buffer a second
            # time -- only the averaging is still owed -- and a further synchronize() changes nothing
            import alpro_amd.compat
            if alpro_amd.compat.PATH not in sys.path:
                sys.path.insert(0, alpro_amd.compat.PATH)
            from horovod import torch as hvd
            fac = hvd.DistributedOptimizer(opt)
            fac.synchronize()
            assert opt._pre_synced == "avg"
            assert torch.equal(ps[0].grad, torch.full((64, 33), float(tot) / world)) and torch.equal(ps[1].grad, torch.full((130,), 10.0 * tot / world))
            fac.synchronize()
            assert torch.equal(ps[0].grad, torch.full((64, 33), float(tot) / world))
            assert not opt._inflight and not opt._reduced          # (step() itself needs the GPU kernels: tests/test_dist_gpu.py)
            opt._pre_synced = None
    finally:
        rt.set_compute_dtype(prev)


def _hvd_facade_checks(rank, world):
    """The horovod.torch / apex.amp stand-ins (alpro_amd/compat) driven the way run_pretrain_sparse.py:432-441,596-648 drives
    them: DistributedOptimizer(...).synchronize() -> clip on AVERAGED gradients -> `with skip_synchronize(): step()`."""
    import sys
    import alpro_amd.compat
    sys.path.insert(0, alpro_amd.compat.PATH)
    from horovod import torch as hvd
    from apex import amp
    assert hvd.__file__.startswith(alpro_amd.compat.PATH)
    hvd.init()
    assert hvd.size() == world and hvd.rank() == rank
    t = torch.full((4,), float(rank + 1))
    assert torch.allclose(hvd.allreduce(t), torch.full((4,), 1.5)) and float(t[0]) == rank + 1     # out-of-place mean
    hvd.allreduce_(t, average=False)
    assert torch.allclose(t, torch.full((4,), 3.0))
    b = torch.full((2,), float(rank))
    hvd.broadcast_(b, root_rank=1)
    assert float(b[0]) == 1.0
    torch.manual_seed(100 + rank)                 # different initial weights per rank
    model = torch.nn.Linear(3, 2)
    opt = hvd.DistributedOptimizer(torch.optim.SGD(model.parameters(), lr=0.5, momentum=0.9), named_parameters=model.named_parameters(),
                                   compression=hvd.Compression.none)
    hvd.broadcast_parameters(model.state_dict(), root_rank=0)
    hvd.broadcast_optimizer_state(opt, root_rank=0)
    model, opt = amp.initialize(model, opt, enabled=0, opt_level="O1")
    w0 = model.weight.detach().clone()
    x = torch.full((1, 3), float(rank + 1))
    loss = model(x).sum()
    with amp.scale_loss(loss, opt, delay_unscale=False) as scaled:
        scaled.backward()
        opt.synchronize()
    assert torch.allclose(model.weight.grad, torch.full((2, 3), 1.5))                             # mean of 1 and 2
    torch.nn.utils.clip_grad_norm_(amp.master_params(opt), 100.0)
    with opt.skip_synchronize():
        opt.step()
        opt.zero_grad()
    assert torch.allclose(model.weight.detach(), w0 - 0.5 * 1.5)
    both = hvd.allgather(model.weight.detach().reshape(1, -1))
    assert torch.allclose(both[0], both[1])                                                       # replicas stay identical
    # step() without an explicit synchronize() exchanges by itself
    model(x).sum().backward()
    opt.step()
    both = hvd.allgather(model.weight.detach().reshape(1, -1))
    assert torch.allclose(both[0], both[1])
    assert amp.state_dict() == {}


def test_world2_gloo():
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0, "rank process failed (exit %s)" % p.exitcode
    got = sorted(q.get(timeout=5) for _ in range(2))
    assert got == [(0, "ok"), (1, "ok")]
