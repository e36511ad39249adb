"""How well-conditioned are the full-size ConvStack gradient norms?  The golden full-size case (tests/golden/g2_full.*, 2 clips, teacher
forcing 1.0, no dropout) is evaluated with the fp32-input MFMA convolutions on the original spectrogram and on copies perturbed by
one unit in the last place (x * (1 +- 2^-23), random signs), and with the split-operand convolutions; the table shows the relative
deviation of every ConvStack gradient norm from the fixture.  usage: python tools/grad_conditioning.py"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from piano_a2s_amd import engine, engine_bwd, hip, spec, synthetic  # noqa: E402
from test_gpu_backward import _loss_grads  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    gd = os.path.join(ROOT, "tests", "golden")
    data = np.load(os.path.join(gd, "g2_full.npz"))
    meta = json.load(open(os.path.join(gd, "g2_full.json")))
    cfg = spec.default_cfg()
    st = spec.procedural_state(cfg, meta["weights_seed"], eos_bias=meta["eos_bias"], lively=meta["lively"])
    kw = dict(meta["batch_kwargs"])
    kw["upper_range"], kw["lower_range"] = tuple(kw["upper_range"]), tuple(kw["lower_range"])
    batch = synthetic.make_batch(2, cfg, meta["batch_seed"], **kw)
    S = {k: v.to(dev) for k, v in st.items()}
    gt = [b.to(dev) for b in batch[1:7]]
    names = [k for k in meta["grad_names"] if k.startswith("convstack.")]
    ref = dict(zip(meta["grad_names"], data["train_tf1.gradnorms"]))
    L = hip.lib()

    def run(x, mode):
        hip.check(L.a2s_debug_set(b"conv_bf16x3", mode), "debug_set")
        eng = engine.Engine(cfg)
        outs = eng.forward(S, x, inference=False, ground_truth=gt, teacher_forcing_ratio=1.0, training=True, dropout=False)
        _, gouts = _loss_grads(outs, batch, dev)
        G = engine_bwd.backward(eng, S, gouts)
        torch.cuda.synchronize()
        return {k: abs(float(G[k].double().norm()) - ref[k]) / ref[k] for k in names}

    x0 = batch[0].to(dev)
    cols = {"fp32": run(x0, 0)}
    g = torch.Generator(device="cpu").manual_seed(0)
    for i in range(4):
        sign = (torch.randint(0, 2, x0.shape, generator=g).float() * 2 - 1).to(dev)
        cols[f"fp32 ulp{i}"] = run(x0 * (1 + sign * 2.0 ** -23), 0)
    cols["split dgrad"] = run(x0, 2)
    cols["split all"] = run(x0, 3)
    hip.check(L.a2s_debug_set(b"conv_bf16x3", 3), "debug_set")
    print(f"{'relative deviation of |grad| from the fixture':46s}" + "".join(f"{c:>12s}" for c in cols))
    for k in names:
        print(f"{k:46s}" + "".join(f"{cols[c][k]:12.2e}" for c in cols))
    print(f"{'max':46s}" + "".join(f"{max(cols[c].values()):12.2e}" for c in cols))


if __name__ == "__main__":
    main()
