"""hip.gemm on strided / offset operand views against torch.matmul (float64 on the CPU)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from piano_a2s_amd import hip
dev = torch.device("cuda:0")
torch.manual_seed(0)
for (M, N, K, ldb, boff, ldc, coff, hop) in [(201, 60, 512, 120, 0, 120, 0, 160), (201, 60, 512, 120, 60, 120, 60, 160), (201, 64, 512, 128, 0, 128, 0, 160),
                                             (201, 60, 512, 60, 0, 60, 0, 160), (201, 960, 512, 960, 0, 960, 0, 160), (201, 60, 32, 120, 60, 960, 540, 5),
                                             (201, 60, 512, 120, 0, 960, 420, 160)]:
    plen = (M - 1) * hop + K + 8
    plen += (-plen) % 4
    a = torch.randn(plen, device=dev)
    b = torch.randn(K, ldb, device=dev)
    c = torch.zeros(M, ldc, device=dev)
    hip.gemm(a, hop, 1, b, ldb, 1, c, ldc, M, N, K, batch=1, bsA=plen, bsB=0, bsC=M * ldc, b_off=boff, c_off=coff)
    torch.cuda.synchronize()
    A = torch.stack([a[i * hop:i * hop + K] for i in range(M)]).double().cpu()
    want = A @ b[:, boff:boff + N].double().cpu()
    got = c[:, coff:coff + N].double().cpu()
    untouched = float(c.abs().sum() - c[:, coff:coff + N].abs().sum())
    print((M, N, K, ldb, boff, ldc, coff, hop), "rel err %.3e" % float((got - want).abs().max() / want.abs().max()), "outside written:", untouched)
