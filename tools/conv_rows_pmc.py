"""Three launches of each row-streaming convolution at the training shapes (for rocprofv3 --pmc passes): conv4 forward (40 -> 40,
conv3x3_rows16), conv3 forward (20 -> 40, rows16), conv2 forward (20 -> 20, conv3x3_rows), conv4 data gradient with the
BatchNorm-backward statistics epilogue, and the row-streaming weight gradients (conv3x3_wgrad_rows 40 -> 40, 20 -> 40, 20 -> 20).
usage: python tools/conv_rows_pmc.py [B]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from piano_a2s_amd import hip  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
T, F = 1201, 480
dev = torch.device("cuda:0")
L = hip.lib()
for ci, co, flip in ((40, 40, 0), (20, 40, 0), (20, 20, 0), (40, 40, 1)):
    x = torch.randn(B, T, ci, F, device=dev) * (1e-4 if flip else 1.0)
    y = torch.empty(B, T, co, F, device=dev)
    w = torch.randn((ci, co, 3, 3) if flip else (co, ci, 3, 3), device=dev) * 0.05
    scale, shift = torch.rand(ci, device=dev) + 0.5, torch.randn(ci, device=dev) * 0.1
    partial = torch.empty(L.a2s_conv3x3_stat_blocks(B, T, F, ci), co, 2, device=dev)
    cws = hip.conv_workspace(ci, dev)
    in_amax = x.abs().amax(dim=(0, 1, 3)).contiguous()
    out_amax = torch.zeros(co, device=dev)
    xmax = hip.absmax(x)
    yl = torch.randn(B, T, co, F, device=dev) if flip else None
    bn = [torch.randn(co, device=dev) * 0.1, torch.rand(co, device=dev) + 0.5, torch.rand(co, device=dev) + 0.5, torch.randn(co, device=dev) * 0.1]
    for _ in range(3):
        if flip:
            hip.check(L.a2s_conv3x3_dgrad_bnstats_scaled(hip.stream(), hip._p(x), hip._p(w), hip._p(y), hip._p(yl), hip._p(bn[0]), hip._p(bn[1]), hip._p(bn[2]),
                                                         hip._p(bn[3]), hip._p(partial), B, T, F, ci, co, hip._p(cws), hip._p(xmax)), "dgrad")
        else:
            hip.conv3x3_forward(x, w, y, scale, shift, partial, cws, in_amax, out_amax)
    torch.cuda.synchronize()
    del x, y, yl
for ci, co in ((40, 40), (20, 40), (20, 20)):
    x = torch.randn(B, T, ci, F, device=dev)
    dy = torch.randn(B, T, co, F, device=dev) * 1e-4
    scale, shift = torch.rand(ci, device=dev) + 0.5, torch.randn(ci, device=dev) * 0.1
    dW = torch.zeros(co, ci, 3, 3, device=dev)
    ws = torch.empty(L.a2s_conv3x3_wgrad_workspace_bytes(ci, co) // 4, device=dev)
    bound = hip.act_bound(scale, shift, x.abs().amax(dim=(0, 1, 3)).contiguous())
    dmax = hip.absmax(dy)
    for _ in range(3):
        hip.conv3x3_wgrad(dy, x, scale, shift, dW, ws, dmax, bound)
    torch.cuda.synchronize()
    del x, dy
