"""Dropout on the HIP path (reference F.dropout sites: models.py:541 p=0.2 on the ConvStack output, :239 p=0.1 on the bar token,
:391 p=0.1 on every note token): the masks the engine draws have the right keep probability, the forward applies keep / (1 - p)
scaling, and the BACKWARD reuses the very same masks -- shown end to end by handing the engine's masks to the oracle (its F.dropout
calls replaced by the recorded masks, in the reference's call order) and comparing outputs, loss and every gradient."""
import math
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def test_dropout_masks_statistics_scaling_and_reuse_in_backward(dev):
    from oracle import model_ref, recipe_ref
    from piano_a2s_amd import engine, engine_bwd, spec, synthetic
    cfg = spec.default_cfg(freq_bins=24, conv_feature_size=32, hidden_size=32, max_length=(12, 8))
    st = spec.procedural_state(cfg, 11, eos_bias=3.0, lively=True)
    batch = synthetic.make_batch(6, cfg, 5, frames=41, upper_range=(3, 10), lower_range=(2, 7), full_tail=0.1)
    S = {k: v.to(dev) for k, v in st.items()}
    gt = [b.to(dev) for b in batch[1:7]]
    torch.manual_seed(99)
    eng = engine.Engine(cfg)
    outs = eng.forward(S, batch[0].to(dev), inference=False, ground_truth=gt, teacher_forcing_ratio=0.5, training=True, dropout=True, rng=random.Random(3))
    torch.cuda.synchronize()
    sv = eng.saved
    # ---- (1) the masks, in the reference's F.dropout call order: ConvStack output, then per bar [bar token, upper steps, lower steps]
    conv_mask = sv["conv"]["drop"].view(6, 41, 32).cpu().float()
    masks = [(conv_mask, 0.2)]
    token_masks = []
    for b in sv["bars"]:
        masks.append((b["keep"].cpu().float().unsqueeze(1), 0.1))
        for name in ("up", "lo"):
            s = b["staff"][name][2]
            for t in range(s["steps"]):
                m = s["drop"][t].cpu().float().unsqueeze(1)
                masks.append((m, 0.1))
                token_masks.append(m)
    # keep probabilities (4 sigma of a binomial)
    for m, p, what in ((conv_mask, 0.2, "ConvStack output"), (torch.cat([x.flatten() for x in token_masks]), 0.1, "note tokens"),
                       (torch.cat([b["keep"].cpu().flatten() for b in sv["bars"]]), 0.1, "bar tokens")):
        n = m.numel()
        assert abs(float(m.mean()) - (1 - p)) <= 4 * math.sqrt(p * (1 - p) / n), f"{what}: keep rate {float(m.mean()):.4f} over {n} elements, expected {1 - p}"
        assert set(np.unique(m.numpy()).tolist()) <= {0.0, 1.0}
    # ---- (2) forward scaling: the encoder's input is relu(bn(z)) * mask / 0.8, exactly zero where dropped
    mean, invstd, scale, shift = (t.cpu() for t in sv["conv"]["out_bn"])
    z = sv["conv"]["z"].cpu()
    expect = torch.relu(z * scale + shift) * conv_mask.view(-1, 32) / 0.8
    got = sv["enc"]["layers"][0]["in"].cpu()
    assert torch.equal(got == 0, expect == 0) and float((got - expect).abs().max()) <= 1e-6 * float(expect.abs().max())
    # ---- (3) the oracle with the SAME masks: outputs, loss, all gradients
    P, Bf = spec.split_state(st)
    P = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    Bf = {k: v.clone() for k, v in Bf.items()}
    queue = list(masks)

    def replay(x, p, training, enabled):
        m, p_rec = queue.pop(0)
        assert p == p_rec and tuple(m.shape) == tuple(x.shape), (p, p_rec, m.shape, x.shape)
        return x * m / (1.0 - p)
    orig = model_ref._dropout
    model_ref._dropout = replay
    try:
        ref = model_ref.forward(P, Bf, cfg, batch[0], inference=False, ground_truth=[batch[i] for i in range(1, 7)], teacher_forcing_ratio=0.5,
                                training=True, rng=random.Random(3), dropout=True)
    finally:
        model_ref._dropout = orig
    assert not queue, "the engine drew more masks than the reference has dropout calls"
    for name, o, r in zip(("ts", "key", "up", "lo"), outs, ref):
        rel = ((o.cpu() - r.detach()).abs() / r.detach().abs().clamp(min=1.0)).max()          # north-star bar: 1e-4 relative
        assert float(rel) <= 1e-4, (name, float(rel))
    losses = recipe_ref.objectives(ref, (batch[1], batch[2], batch[3], batch[5]))
    losses[0].backward()
    leaves = [o.detach().clone().requires_grad_(True) for o in outs]
    mine = recipe_ref.objectives(leaves, tuple(b.to(dev) for b in (batch[1], batch[2], batch[3], batch[5])))
    mine[0].backward()
    assert abs(float(mine[0]) - float(losses[0])) <= 1e-4 * abs(float(losses[0]))
    G = engine_bwd.backward(eng, S, [l.grad for l in leaves])
    torch.cuda.synchronize()
    bad = []
    for k, p in P.items():
        ref_g = p.grad.double()
        err = float((G[k].cpu().double() - ref_g).abs().max()) / max(float(ref_g.abs().max()), 1e-12)
        if err > 2e-4:
            bad.append((k, err))
    assert not bad, f"gradients differ from the oracle run on the same dropout masks: {bad[:6]}"
