import cProfile, pstats, random, sys, os, io
import torch
sys.path.insert(0, os.getcwd())
import models
from piano_a2s_amd import spec, synthetic, train
dev = torch.device("cuda:0")
cfg = spec.default_cfg()
torch.manual_seed(1)
m = models.ScoreTranscription(**cfg).to(dev); m.train()
step = train.TrainStep(m)
b = synthetic.make_batch(256, cfg, 1234, full_tail=0.01)
b = [t.to(dev) if torch.is_tensor(t) else t for t in b]
for k in range(3):
    step(b, 0.7, rng=random.Random(100 + k)); torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
step(b, 0.7, rng=random.Random(103))
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue()[:6000])
