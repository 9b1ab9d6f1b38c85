import torch, sys
sys.path.insert(0, '.')
from alpro_amd import hip
hip.load()
K = 768
for dt in (torch.float32, torch.bfloat16):
    a = torch.zeros(130, K); a[torch.arange(130), torch.arange(130)] = 1.0
    w = ((torch.arange(200 * K, dtype=torch.float32).reshape(200, K) % 251) - 125.0)
    out = hip.gemm(a.to(dt).cuda(), w.to(dt).cuda(), out_dtype=torch.float32).cpu()
    ref = w[:, :130].T.contiguous()
    print(dt, 'equal', torch.equal(out, ref), 'maxerr', (out - ref).abs().max().item())
    if not torch.equal(out, ref):
        bad = (out != ref).nonzero()
        print('n bad', len(bad), bad[:10].tolist())
        print(out[:4, :8]); print(ref[:4, :8])
        # where does out[0, :] come from? search ref values
        a2 = torch.zeros(1, K); a2[0, 5] = 1.0
        o2 = hip.gemm(a2.to(dt).cuda(), w.to(dt).cuda(), out_dtype=torch.float32).cpu()
        print('row e5 ->', o2[0, :6], 'expect', w[:6, 5])
        for kk in (0, 1, 4, 8, 31, 32, 64, 100):
            a2 = torch.zeros(1, K); a2[0, kk] = 1.0
            o2 = hip.gemm(a2.to(dt).cuda(), w.to(dt).cuda(), out_dtype=torch.float32).cpu()
            # find which k column of w matches
            match = [(k2) for k2 in range(K) if torch.equal(o2[0], w[:, k2])]
            print('unit k', kk, 'matches w col', match[:4], 'nonzero', int((o2 != 0).sum()))
