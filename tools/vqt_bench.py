"""VQT front end alone (for rocprofv3 --kernel-trace --stats): python tools/vqt_bench.py [batch] [iters]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from piano_a2s_amd import synthetic, vqt
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    dev = torch.device("cuda:0")
    wave = synthetic.make_waveforms(B, 5, device=dev)
    front = vqt.VQT(dev)
    front(wave)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(iters):
        out = front(wave)
    torch.cuda.synchronize()
    ms = (time.time() - t0) / iters * 1e3
    flops = sum(2.0 * 2 * (o["hi"] - o["lo"]) * o["n_fft"] for o in front.octaves) * out.shape[2] * B
    print(f"B={B}: {ms:.2f} ms per batch, {B / ms * 1e3:.0f} clips/s, {flops / ms / 1e9:.2f} TFLOP/s algorithmic; octaves (hop, n_fft): "
          + " ".join(f"({o['hop']},{o['n_fft']})" for o in front.octaves) + f"; decimator taps {front.dec_taps}")


if __name__ == "__main__":
    main()
