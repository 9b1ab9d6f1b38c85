"""Phase timeline of ONE workgroup of the key-owned attention backward (measurement build only):
    python -m alpro_amd.build --ablations && ALPRO_HIP_LIB=alpro_amd/lib/libalpro_hip_ablate.so python tools/attn_bwd_stamps.py [block]
Stamps (shader clock): 0 start, 1 loads issued, 2 loads landed, 3 after barrier, then per round r (0, 1): 4+4r main pass done,
5+4r after barrier, 6+4r dQ pass done, 7+4r after barrier; 12 end.
Persistent kernel (ALPRO_ATTN_BWD=2): steps 80, 81, 82 of the workgroup, 5 stamps each: step start, own copies landed, after the barrier,
copies issued, step work done."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from alpro_amd import hip
lib = hip.load()
lib.alpro_debug_attn_bwd_stamps.argtypes = [ctypes.c_int, ctypes.c_void_p]
batch, L, H = 512, 197, 12
dt = torch.float16
qkv = torch.randn(batch * L, 3 * H * 64, device="cuda").to(dt)
out, lse = hip.attn(qkv, batch, L, H, 0.125, want_lse=True)
do = torch.randn_like(out)
for _ in range(3):
    hip.attn_bwd(qkv, out, do, lse, batch, L, H, 0.125)
torch.cuda.synchronize()
alias = int(os.environ.get("ATTN_ALIAS", "0"))   # > 0: all units alias the first `alias` units (inputs L2 resident; results wrong, timing only)
if alias:
    assert lib.alpro_debug_attn_bwd_alias(alias) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        hip.attn_bwd(qkv, out, do, lse, batch, L, H, 0.125)
    e0.record()
    for _ in range(10):
        hip.attn_bwd(qkv, out, do, lse, batch, L, H, 0.125)
    e1.record(); torch.cuda.synchronize()
    print("alias %d: %.3f ms per launch" % (alias, e0.elapsed_time(e1) / 10))
for block in ([int(sys.argv[1])] if len(sys.argv) > 1 else [100, 3000, 5000]):
    assert lib.alpro_debug_attn_bwd_stamps(block, None) == 0
    hip.attn_bwd(qkv, out, do, lse, batch, L, H, 0.125)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 128)()
    assert lib.alpro_debug_attn_bwd_stamps(-1, buf) == 0
    t0 = min(buf[w * 16] for w in range(8))
    print("block %d (cycles since the first wave started)" % block)
    for w in range(8):
        print("  wave %d: " % w + " ".join("%6d" % (buf[w * 16 + i] - t0) for i in range(16)))
