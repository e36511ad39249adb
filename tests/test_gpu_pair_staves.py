"""The two NoteDecoders of a segment issued by one host loop with ONE attention sweep per decode step for both staves while both run
(csrc/a2s_seq.hip: a2s_note_decoder_fwd_pair / attn_fwd_split256_pair -- the encoder outputs are read once per clip and step instead of once per
staff; round 6) against the two independent step loops (a2s_debug_set("attn_pair", 0)): the whole fused training step, loss terms, gradient
norm, updated parameters; the library-side launch counter proves which path ran.  Reference: models.py:261-275 (both decode_notes calls read the
same encoder_outputs)."""
import random

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.mark.parametrize("B,tf,groups", [(80, 1.0, False), (120, 0.6, False), (128, 0.7, True)])
def test_one_sweep_for_both_staves_equals_two_sweeps(dev, B, tf, groups):
    import models
    from piano_a2s_amd import hip, spec, synthetic, train
    L = hip.lib()
    if groups:          # a long-clip group beside the bulk group (as tests/test_gpu_step_ordering.py makes one)
        cfg = spec.default_cfg(freq_bins=48, max_length=(40, 24))
        batch = synthetic.make_batch(B, cfg, 63, frames=61, upper_range=(3, 12), lower_range=(2, 8), full_tail=0.0, full_rows=((3, 1, "up"), (8, 3, "lo")))
    else:
        cfg = spec.default_cfg(freq_bins=48, max_length=(24, 14))
        batch = synthetic.make_batch(B, cfg, 63, frames=61, upper_range=(4, 22), lower_range=(3, 12), full_tail=0.05)
    dbatch = [t.to(dev) if torch.is_tensor(t) else t for t in batch]
    torch.manual_seed(11)
    init = models.ScoreTranscription(**cfg).state_dict()
    prev = L.a2s_debug_get(b"attn_pair")
    res = []
    try:
        for pair in (0, 1):
            hip.check(L.a2s_debug_set(b"attn_pair", pair), "debug_set")
            m = models.ScoreTranscription(**cfg)
            m.load_state_dict(init)
            m = m.to(dev).train()
            step = train.TrainStep(m, dropout=False, **(dict(group_plan={"step_cost": 4.0, "min_gain": 0.0}) if groups else dict(clip_groups=False)))
            n0, nb0 = L.a2s_debug_get(b"attn_pair_launches"), L.a2s_debug_get(b"attn_pair_bwd_launches")
            losses = step(dbatch, tf, rng=random.Random(7))
            torch.cuda.synchronize()
            res.append((losses[:, 0].double().cpu(), step.opt.ctl.double().cpu(), step.flat.double().cpu(), L.a2s_debug_get(b"attn_pair_launches") - n0,
                        len(step._last[2]), L.a2s_debug_get(b"attn_pair_bwd_launches") - nb0))
            del step, m
    finally:
        hip.check(L.a2s_debug_set(b"attn_pair", prev), "debug_set")
    (l0, c0, p0, n_off, g0, nb_off), (l1, c1, p1, n_on, g1, nb_on) = res
    assert n_off == 0 and n_on > 0, f"pair sweeps launched: {n_off} with the switch off, {n_on} with it on"
    assert nb_off == 0 and nb_on == n_on, f"backward pair sweeps: {nb_off} / {nb_on} against {n_on} forward"
    assert g0 == g1 and (g0 >= 2) == groups
    assert torch.isfinite(l1).all() and float(c1[2]) == 1.0
    assert torch.allclose(l0, l1, rtol=2e-6, atol=0), (l0, l1)
    assert abs(float(c0[0]) - float(c1[0])) <= 2e-5 * float(c0[0]), (c0, c1)
    assert float((p0 - p1).abs().max()) <= 5e-6 * float(p0.abs().max())
