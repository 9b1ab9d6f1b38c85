import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from alpro_amd import hip
hip.load()
dt = torch.bfloat16
shapes = [("qkv wgrad", 100416, 2304, 768), ("proj wgrad", 100416, 768, 768), ("fc1 wgrad", 100416, 3072, 768), ("fc2 wgrad", 100416, 768, 3072),
                      ("bert qkv slice", 15168, 768, 768), ("bert ffn", 15168, 3072, 768), ("bert neg ffn", 30336, 3072, 768)]
if len(sys.argv) > 1: shapes = [shapes[int(i)] for i in sys.argv[1].split(",")]
for name, M, N, K in shapes:
    a = torch.randn(M, N, device="cuda").to(dt); b = torch.randn(M, K, device="cuda").to(dt)
    c = torch.zeros(N, K, device="cuda")
    for _ in range(3): hip.gemm_tn_acc(a, b, c)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): hip.gemm_tn_acc(a, b, c)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print("%-16s M=%6d N=%4d K=%4d  %.3f ms  %.0f TF" % (name, M, N, K, ms, 2.0 * M * N * K / ms / 1e9))
