#!/usr/bin/env python3
"""Greedy-decode throughput (BASELINE.json configs[4]): batched 12 s clips -> Kern tokens on one MI355X, eval mode, procedural
weights with <eos> bias so decoding terminates at data-dependent steps.  Prints one JSON line (clips/s, tokens/s, decode steps)."""
import json
import sys
import time

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from piano_a2s_amd import engine, spec, synthetic

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda:0")
cfg = spec.default_cfg()
S = {k: v.to(dev) for k, v in spec.procedural_state(cfg, 2032, eos_bias=2.5, lively="token").items()}
batch = synthetic.make_batch(B, cfg, 77, spectrogram="ridges", full_tail=0.0)
x = batch[0].to(dev)
eng = engine.Engine(cfg)
for it in range(3):
    torch.cuda.synchronize(); t0 = time.time()
    with torch.no_grad():
        ts, key, up, lo = eng.forward(S, x, inference=True)
    torch.cuda.synchronize(); dt = time.time() - t0
steps = sum(b["staff"][k][2]["steps"] for b in eng.saved["bars"] for k in ("up", "lo"))
launched = sum(b["staff"][k][2]["launched"] for b in eng.saved["bars"] for k in ("up", "lo"))
tokens = int((up.abs().sum(-1) > 0).sum() + (lo.abs().sum(-1) > 0).sum())
print(json.dumps({"metric": "greedy decode clips/s", "batch": B, "seconds": round(dt, 4), "clips_per_s": round(B / dt, 2),
                  "decoded_token_rows_per_s": round(tokens / dt), "executed_steps": steps, "launched_steps": launched}))
