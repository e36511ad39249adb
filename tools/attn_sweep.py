import ctypes as C, sys, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from piano_a2s_amd import hip
B = int(sys.argv[1]); T, H = 1201, 256
L = hip.lib(); dev = torch.device("cuda:0")
keys = torch.exp(2 * torch.randn(B, T, H, device=dev) * 0.5); enc = torch.randn(B, T, 2 * H, device=dev)
q = torch.randn(B, H, device=dev) * 0.5; v = torch.randn(H, device=dev) * 0.3
ctx = torch.empty(B, 2 * H, device=dev); attw = torch.empty(B, T, device=dev); ws = hip.attn_workspace(B, T, H, dev)
dctx = torch.randn(B, 2 * H, device=dev); dq = torch.empty(B, H, device=dev); ds = torch.empty(B, T, device=dev)
def fwd(): hip.check(L.a2s_attn_step_fwd(hip.stream(), hip._p(keys), hip._p(enc), hip._p(q), C.c_long(H), hip._p(v), hip._p(ctx), C.c_long(2*H), C.c_void_p(0), C.c_long(0), hip._p(attw), B, T, H, C.c_void_p(0), 0, hip._p(ws)), "f")
def bwd(): hip.check(L.a2s_attn_step_bwd(hip.stream(), hip._p(keys), hip._p(enc), hip._p(q), C.c_long(H), hip._p(v), hip._p(attw), hip._p(ctx), C.c_long(2*H), hip._p(dctx), C.c_long(2*H), C.c_void_p(0), C.c_long(0), C.c_void_p(0), C.c_long(0), hip._p(dq), C.c_long(H), hip._p(ds), B, T, H, hip._p(ws)), "b")
for name, fn in (("fwd", fwd), ("bwd", bwd)):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(40): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 40 * 1e3
    print(name, "B", B, round(us, 1), "us", round(B * T * 768 * 4 / us / 1e3, 1), "GB/s")
