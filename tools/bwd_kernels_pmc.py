"""Two launches of each big ConvStack-backward kernel at the training shapes (for rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes): two-term
weight gradients, two-term data gradients with the BatchNorm-statistics epilogue, the BatchNorm-backward apply, the Linear's three GEMMs.
usage: python tools/bwd_kernels_pmc.py [B]   -- prints the algorithmic tensor sizes in KB for comparison"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from piano_a2s_amd import hip  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
T, F = 1201, 480
dev = torch.device("cuda:0")
L = hip.lib()
kb = lambda c: B * T * c * F * 4 // 1024
print(f"tensor KB: 20 channels {kb(20)}, 40 channels {kb(40)}")
for ci, co in ((20, 20), (20, 40), (40, 40)):
    x = torch.randn(B, T, ci, F, device=dev)
    dy = torch.randn(B, T, co, F, device=dev) * 1e-4
    dW = torch.zeros(co, ci, 3, 3, device=dev)
    scale, shift = torch.rand(ci, device=dev) + 0.5, torch.randn(ci, device=dev) * 0.1
    nbytes = L.a2s_conv3x3_wgrad_workspace_bytes(ci, co)
    ws = torch.empty(nbytes // 4, device=dev)
    amax = hip.absmax(dy)
    for _ in range(2):
        hip.check(L.a2s_conv3x3_wgrad_scaled(hip.stream(), hip._p(dy), hip._p(x), hip._p(scale), hip._p(shift), hip._p(dW), hip._p(ws), C.c_size_t(nbytes),
                                             B, T, F, ci, co, hip._p(amax)), "wgrad")
    # data gradient: dy (co channels) -> g (ci channels), statistics epilogue over yl (ci channels)
    g = torch.empty(B, T, ci, F, device=dev)
    yl = torch.randn(B, T, ci, F, device=dev)
    w = torch.randn(co, ci, 3, 3, device=dev) * 0.05
    mean, invstd = torch.zeros(ci, device=dev), torch.ones(ci, device=dev)
    part = torch.empty(L.a2s_conv3x3_stat_blocks(B, T, F, co), ci, 2, device=dev)
    cws = hip.conv_workspace(co, dev)
    for _ in range(2):
        hip.check(L.a2s_conv3x3_dgrad_bnstats_scaled(hip.stream(), hip._p(dy), hip._p(w), hip._p(g), hip._p(yl), hip._p(mean), hip._p(invstd), hip._p(scale), hip._p(shift),
                                                     hip._p(part), B, T, F, co, ci, hip._p(cws), hip._p(amax)), "dgrad")
    torch.cuda.synchronize()
    del x, dy, g, yl
rows, Cc = B * T, 40
g = torch.randn(rows, Cc, F, device=dev); x = torch.randn(rows, Cc, F, device=dev)
mean, invstd = torch.zeros(Cc, device=dev), torch.ones(Cc, device=dev)
scale, shift = torch.ones(Cc, device=dev), torch.zeros(Cc, device=dev)
dg, db = torch.zeros(Cc, device=dev), torch.zeros(Cc, device=dev)
part = torch.empty(L.a2s_bn_bwd_partial_floats(C.c_long(rows), Cc, F), dtype=torch.float32, device=dev)
c12 = torch.empty(2 * Cc, device=dev); amax = torch.zeros(1, device=dev)
for _ in range(2):
    hip.check(L.a2s_bn_bwd_amax(hip.stream(), hip._p(g), hip._p(x), hip._p(mean), hip._p(invstd), hip._p(scale), hip._p(shift), None, hip.f32(1.0),
                                hip._p(dg), hip._p(db), hip._p(g), hip._p(part), hip._p(c12), C.c_long(rows), Cc, F, hip._p(amax)), "bn_bwd")
torch.cuda.synchronize()
