"""Per-rank vs synchronised BatchNorm at world size 1 (bench.py's data_parallel_settings block alone): python tools/dp_settings.py [batch] [steps]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import bench
    import models
    from piano_a2s_amd import spec
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    dev = torch.device("cuda:0")
    cfg = spec.default_cfg()
    torch.manual_seed(1234)
    model = models.ScoreTranscription(**cfg).to(dev)
    model.train()
    print(json.dumps(bench.dp_settings_block(model, cfg, B, dev, 0.01, steps=steps)))


if __name__ == "__main__":
    main()
