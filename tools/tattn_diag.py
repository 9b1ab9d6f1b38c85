"""GPU diagnostic: where does alpro_gemm_qkv_tattn differ from alpro_gemm + alpro_attn_temporal_fwd?  python tools/tattn_diag.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from alpro_amd import hip
hip.load()
dt = torch.float16
H, T, K = 12, 8, 768
for M in (768, 256, 32, 512 + 96):
    torch.manual_seed(M)
    a = torch.randn(M, K, device="cuda").to(dt)
    w = (torch.randn(3 * H * 64, K, device="cuda") * 0.07).to(dt)
    b = torch.randn(3 * H * 64, device="cuda")
    two = hip.attn_temporal(hip.gemm(a, w, bias=b), T, H, 0.125).float()
    for rep in range(3):
        out = hip.gemm_qkv_tattn(a, w, b, T, H, 0.125).float()
        bad = (out - two).abs() > 0.05
        nb = int(bad.sum())
        print("M=%d rep %d: %d bad of %d" % (M, rep, nb, bad.numel()))
        if nb:
            rows = bad.any(1).nonzero().flatten().tolist()
            heads = bad.view(M, H, 64).any(2).any(0).nonzero().flatten().tolist()
            ds = bad.view(M, H, 64).any(1).any(0).nonzero().flatten().tolist()
            print("   rows", rows[:40], "... n=%d" % len(rows))
            print("   heads", heads, " d", ds[:70])
            r0 = rows[0]
            h0 = bad.view(M, H, 64)[r0].any(1).nonzero().flatten().tolist()[0]
            print("   sample row %d head %d:\n   fused %s\n   two   %s" % (r0, h0, out.view(M, H, 64)[r0, h0, :16].tolist(), two.view(M, H, 64)[r0, h0, :16].tolist()))
    if M == 768:
        o3, qkv3, lse3 = hip.gemm_qkv_tattn(a, w, b, T, H, 0.125, want_qkv=True)
        g2 = hip.gemm(a, w, bias=b)
        print("   training form: out bad %d, qkv max diff %.3e, lse max diff %.3e" % (int(((o3.float() - two).abs() > 0.05).sum()), (qkv3.float() - g2.float()).abs().max().item(),
              (lse3 - hip.attn_temporal(g2, T, H, 0.125, want_lse=True)[1]).abs().max().item()))
