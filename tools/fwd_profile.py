"""Forward-only timing probe (round-1 development aid): training-mode teacher-forced forward at full model size."""
import sys, time, random
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from piano_a2s_amd import engine, spec, synthetic

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda:0")
cfg = spec.default_cfg()
S = {k: v.to(dev) for k, v in spec.procedural_state(cfg, 1).items()}
batch = synthetic.make_batch(B, cfg, 1234, full_tail=0.0)
spec_d = batch[0].to(dev); gt = [b.to(dev) for b in batch[1:7]]
eng = engine.Engine(cfg)
random.seed(1234)
for it in range(2):
    torch.cuda.synchronize(); t0 = time.time()
    eng.forward(S, spec_d, inference=False, ground_truth=gt, teacher_forcing_ratio=0.7, training=True, dropout=True)
    torch.cuda.synchronize(); t1 = time.time()
    steps = sum(b["staff"][k][2]["steps"] for b in eng.saved["bars"] for k in ("up", "lo"))
    print(f"iter {it}: forward B={B} {t1 - t0:.3f}s decode steps={steps} mem={torch.cuda.max_memory_allocated() / 2**30:.2f} GiB", flush=True)
