"""Three launches of each split-operand convolution at the training shapes (for rocprofv3 --pmc passes).  usage: python tools/conv_split_pmc.py [B] [mode]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from piano_a2s_amd import hip  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 3
T, F = 1201, 480
dev = torch.device("cuda:0")
L = hip.lib()
L.a2s_debug_set(b"conv_bf16x3", mode)
for ci, co, flip in ((40, 40, 0), (20, 20, 0)):
    x = torch.randn(B, T, ci, F, device=dev)
    y = torch.empty(B, T, co, F, device=dev)
    w = torch.randn(co, ci, 3, 3, device=dev) * 0.05
    scale, shift = torch.rand(ci, device=dev) + 0.5, torch.randn(ci, device=dev) * 0.1
    partial = torch.empty(L.a2s_conv3x3_stat_blocks(B, T, F, ci), co, 2, device=dev)
    cws = hip.conv_workspace(ci, dev)
    for _ in range(3):
        hip.check(L.a2s_conv3x3(hip.stream(), hip._p(x), hip._p(w), hip._p(y), hip._p(scale), hip._p(shift), hip._p(partial), B, T, F, ci, co, flip, hip._p(cws)), "conv")
    torch.cuda.synchronize()
