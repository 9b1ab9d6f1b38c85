"""One-minute smoke of the 8-phase GEMM's tile walks on a GPU box (run under `timeout`; a hang here must not take the rest of a call with it)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from alpro_amd import hip  # noqa: E402

hip.load()
g = torch.Generator(device="cuda").manual_seed(0)
for (M, N, K) in ((256 * 200, 768, 768), (256 * 196 + 32, 768, 256), (256 * 700, 512, 768), (15168, 768, 768)):
    A = torch.randn(M, K, device="cuda", generator=g).half()
    W = (torch.randn(N, K, device="cuda", generator=g) * 0.05).half()
    ref = (A.float() @ W.float().t())
    outs = {}
    for sched in (0, 1, 1, 1):
        with hip.option("gemm_sched", sched), hip.option("gemm_tile", 256):
            o = hip.gemm(A, W)
        torch.cuda.synchronize()
        outs.setdefault(sched, o)
        assert torch.equal(o, outs[sched]), ("repeat differs", M, N, K, sched)
    err = float((outs[1].float() - ref).abs().max())
    print("M=%d N=%d K=%d: static == ticket: %s, max err vs fp32 matmul %.3e" % (M, N, K, torch.equal(outs[0], outs[1]), err), flush=True)
    assert torch.equal(outs[0], outs[1]) and err < 0.05 * (K / 768) ** 0.5 + 0.02
print("sched smoke ok")
