"""Host logic of the recipe on CPU (no GPU, no SpeechBrain, no hyperpyyaml): the yaml subset loader, the scheduler / checkpointer /
metrics restatements, and the whole pretrain.py / finetune.py plumbing driven end to end with the oracle standing in for the HIP model."""
import io
import json
import os

import pytest
import torch
import yaml

from piano_a2s_amd import metrics, sb_compat
from piano_a2s_amd.hyperyaml import load_hyperpyyaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_yaml_subset_semantics():
    text = """
seed: 7
__s: !apply:torch.manual_seed [!ref <seed>]
a: 12
b: 100
frames: !ref <a> * <b> + 1
name: !ref run.<seed>
root: !PLACEHOLDER
out: !ref <root>/<seed>/<name>
lens: (398, 189)
opt: !name:torch.optim.Adadelta
  lr: !ref <a>
  rho: 0.95
loss: !new:torch.nn.NLLLoss
  ignore_index: 147
lin: !new:torch.nn.Linear [3, 4]
mods:
  m: !ref <lin>
lst: !new:torch.nn.ModuleList
  - [!ref <lin>]
"""
    with pytest.raises(ValueError):
        load_hyperpyyaml(io.StringIO(text))                       # PLACEHOLDER not overridden
    hp = load_hyperpyyaml(io.StringIO(text), "root: /w\nb: 10")
    assert hp["frames"] == 121 and hp["name"] == "run.7" and hp["out"] == "/w/7/run.7" and hp["lens"] == (398, 189)
    assert hp["loss"].ignore_index == 147 and hp["mods"]["m"] is hp["lin"] and hp["lst"][0] is hp["lin"]
    opt = hp["opt"](hp["lin"].parameters())
    assert isinstance(opt, torch.optim.Adadelta) and opt.param_groups[0]["lr"] == 12 and opt.param_groups[0]["rho"] == 0.95


def test_repository_hparams_resolve():
    for f, ov in (("pretrain.yaml", {"workspace": "/w", "soundfont_folder": "/s"}), ("finetune.yaml", {"workspace": "/w", "asap_folder": "/a", "mv2h_bin": "/m"})):
        hp = load_hyperpyyaml(open(os.path.join(ROOT, "hparams", f)), ov)
        assert hp["max_frame_num"] == 1201 and hp["max_length"] == (398, 189)
        m = hp["transcription"]
        assert m.cfg["freq_bins"] == 480 and hp["modules"]["transcription"] is m and hp["model"][0] is m
        assert sum(p.numel() for p in m.parameters()) == 16_358_675                        # SURVEY: reference parameter count
        assert hp["loss_score"].ignore_index == 147
    assert "counter" not in hp["checkpointer"].recoverables                                # finetune restarts the epoch counter


def test_newbob_and_epoch_counter():
    s = sb_compat.NewBobScheduler(initial_value=1.0, improvement_threshold=0.0025, annealing_factor=0.8, patient=0)
    assert s(0.50) == (1.0, 1.0)                 # first call: nothing to compare with
    assert s(0.40) == (1.0, 1.0)                 # improved by 20 %
    old, new = s(0.3995)                         # improved by 0.125 % < 0.25 % -> anneal
    assert (old, round(new, 6)) == (1.0, 0.8)
    c = sb_compat.EpochCounter(3)
    assert list(c) == [1, 2, 3] and c.state_dict() == {"current": 3}


def test_wer_and_unpad():
    assert metrics.word_error_rate("a b c d", "a x c") == 0.5                              # 1 substitution + 1 deletion over 4 words
    assert metrics.word_error_rate("a b", "a b") == 0.0 and metrics.word_error_rate("a", "a b c") == 2.0
    assert metrics.unpad([5, 6, 146, 147, 146]).tolist() == [5, 6] and metrics.unpad([1, 2]).tolist() == [1, 2]
    inv = {1: "4", 2: "c", 3: "e"}
    wer, per = metrics.corpus_wer({"x": [[1, 2], [1, 3]]}, {"x": [[1, 2], [1, 2]]}, inv)
    assert per["x"] == pytest.approx(1 / 5)            # reference text "4 c \n = \n 4 c" = 5 whitespace-separated words, 1 substituted


def test_checkpointer_keeps_best_and_finetune_seeding(tmp_path):
    lin = torch.nn.Linear(2, 2)
    ck = sb_compat.Checkpointer(str(tmp_path / "pre" / "save"), {"model": lin, "counter": sb_compat.EpochCounter(5)})
    for i, wer in enumerate((0.9, 0.4, 0.7)):
        with torch.no_grad():
            lin.weight.fill_(float(i))
        ck.save_and_keep_only(meta={"loss": 1.0, "WER": wer, "unixtime": 1000.0 + 100 * i}, min_keys=["WER"])
    kept = os.listdir(tmp_path / "pre" / "save")
    assert len(kept) == 1
    assert yaml.safe_load(open(tmp_path / "pre" / "save" / kept[0] / "CKPT.yaml"))["WER"] == 0.4
    assert sorted(os.listdir(tmp_path / "pre" / "save" / kept[0])) == ["CKPT.yaml", "counter.ckpt", "model.ckpt"]
    with torch.no_grad():
        lin.weight.fill_(9.0)
    ck.recover_if_possible(min_key="WER")
    assert float(lin.weight[0, 0]) == 1.0                                                    # the WER-0.4 weights are back
    import finetune
    finetune.seed_from_pretraining(str(tmp_path / "pre"), str(tmp_path / "fine"))
    assert yaml.safe_load(open(tmp_path / "fine" / "save" / kept[0] / "CKPT.yaml"))["WER"] == 100


SMALL = ["--hidden_size=32", "--conv_feature_size=32", "--bins_per_octave=24", "--n_octaves=1", "--max_length=(12, 8)",
         "--synthetic_frames=41", "--synthetic_lengths=[[3, 10], [2, 7]]", "--batch_size=1", "--number_of_epochs=1",
         "--transcription=!new:tests.oracle_module.OracleTranscription {freq_bins: 24, conv_feature_size: 32, hidden_size: 32, max_length: [12, 8]}"]


@pytest.mark.timeout(900)
def test_pretrain_then_finetune_plumbing(tmp_path):
    """`python pretrain.py hparams/pretrain.yaml ...` then `python finetune.py hparams/finetune.yaml ...` on CPU: 2 train clips,
    1 epoch, batch 1; the oracle plays the model.  Checks the artefacts the reference recipe leaves behind."""
    import finetune
    import pretrain
    ws = str(tmp_path)
    brain = pretrain.main([os.path.join(ROOT, "hparams", "pretrain.yaml"), "--device=cpu", f"--workspace={ws}", "--soundfont_folder=/none",
                           "--synthetic_clips=2"] + SMALL)
    out = os.path.join(ws, "1234", "pretrain.epr")
    log = open(os.path.join(out, "train_log.txt")).read()
    assert "epoch: 1" in log and "WER" in log and "teacher_forcing_ratio" in log
    assert abs(brain.train_stats["teacher_forcing_ratio"] - 0.7 * 0.99) < 1e-12                # decay ** epoch with epoch = 1
    ck = os.listdir(os.path.join(out, "save"))
    assert len(ck) == 1 and {"model.ckpt", "scheduler.ckpt", "normalizer.ckpt", "counter.ckpt", "CKPT.yaml"} <= set(os.listdir(os.path.join(out, "save", ck[0])))
    res = os.listdir(os.path.join(out, "results", "test"))
    rec = json.load(open(os.path.join(out, "results", "test", res[0])))
    assert len(rec["pred"]) == 5 and len(rec["pred"][0]) == 4 and isinstance(rec["pred"][0][1], str) and -6 <= rec["pred"][0][0] <= 7
    assert {"wer_upper", "wer_lower", "key_f1", "time_f1", "style", "soundfont", "composer", "target_path"} <= set(rec)
    assert res[0].startswith("0~")                                                             # id = "<version>~<name>"
    fbrain = finetune.main([os.path.join(ROOT, "hparams", "finetune.yaml"), "--device=cpu", f"--workspace={ws}", "--asap_folder=/none",
                            "--mv2h_bin=/none", "--synthetic_clips=2"] + SMALL)
    fout = os.path.join(ws, "1234", "finetune.epr")
    assert fbrain.teacher_forcing_ratio == 0.0 and "teacher_forcing_ratio" not in fbrain.train_stats
    metas = [yaml.safe_load(open(os.path.join(fout, "save", d, "CKPT.yaml"))) for d in os.listdir(os.path.join(fout, "save"))]
    assert len(metas) == 1 and metas[0]["WER"] < 100                                           # the seeded WER=100 checkpoint was replaced


def test_speechbrain_format_checkpoint_directory_recovers(tmp_path):
    """A save/ directory as SpeechBrain 0.5.15 writes it for the reference recipe (pretrain.yaml:110-116): torch-pickled state dicts
    for modules / scheduler, but the EPOCH COUNTER as a plain-text integer (its own saver hook), CKPT.yaml with `end-of-epoch`."""
    d = tmp_path / "save" / "CKPT+2024-10-08+12-00-00+00"
    d.mkdir(parents=True)
    lin = torch.nn.ModuleList([torch.nn.Linear(2, 2)])
    torch.save({"0.weight": torch.full((2, 2), 3.0), "0.bias": torch.zeros(2)}, d / "model.ckpt")
    torch.save({"hyperparam_value": 0.64, "metric_values": [0.9, 0.8], "current_patient": 0}, d / "scheduler.ckpt")
    (d / "counter.ckpt").write_text("7")
    (d / "CKPT.yaml").write_text("WER: 0.8\nend-of-epoch: true\nloss: 1.5\nunixtime: 1728388800.0\n")
    counter, sched = sb_compat.EpochCounter(30), sb_compat.NewBobScheduler(1.0, annealing_factor=0.8)
    ck = sb_compat.Checkpointer(str(tmp_path / "save"), {"model": lin, "scheduler": sched, "counter": counter})
    path, meta = ck.recover_if_possible()
    assert meta["WER"] == 0.8 and counter.current == 7 and sched.hyperparam_value == 0.64 and float(lin[0].weight[0, 0]) == 3.0
    assert next(counter) == 8                                          # training resumes with epoch 8
    # and what is written here is readable by SpeechBrain: text counter, pickled dicts
    ck.save_checkpoint(meta={"WER": 0.5, "unixtime": 1728388900.0}, name="CKPT+x")
    assert (tmp_path / "save" / "CKPT+x" / "counter.ckpt").read_text() == "8"
    assert yaml.safe_load((tmp_path / "save" / "CKPT+x" / "CKPT.yaml").read_text())["end-of-epoch"] is True
    assert set(torch.load(tmp_path / "save" / "CKPT+x" / "scheduler.ckpt")) == {"hyperparam_value", "metric_values", "current_patient"}
    # a mid-epoch checkpoint re-runs the interrupted epoch (SpeechBrain: current = saved - 1)
    (d / "CKPT.yaml").write_text("WER: 0.1\nend-of-epoch: false\nunixtime: 1728389900.0\n")
    c2 = sb_compat.EpochCounter(30)
    sb_compat.Checkpointer(str(tmp_path / "save"), {"counter": c2}).recover_if_possible(min_key="WER")
    assert c2.current == 6
    # round-1 checkpoints of this repository pickled the counter: still readable
    torch.save({"current": 4}, d / "counter.ckpt")
    sb_compat.Checkpointer(str(tmp_path / "save"), {"counter": c2}).recover_if_possible(min_key="WER")
    assert c2.current == 3


def test_fused_adadelta_state_is_torch_adadelta_state():
    """optimizer.ckpt interchange: FusedAdadelta.state_dict() loads into torch.optim.Adadelta and the other way round (the
    reference's Brain checkpoints `optimizer` = torch.optim.Adadelta.state_dict())."""
    from piano_a2s_amd.spec import flat_layout
    from piano_a2s_amd.train import FusedAdadelta
    torch.manual_seed(0)
    params = [torch.nn.Parameter(torch.randn(3, 5)), torch.nn.Parameter(torch.randn(7)), torch.nn.Parameter(torch.randn(2, 2, 3))]
    ref = torch.optim.Adadelta(params, lr=0.8, rho=0.95, eps=1e-8)
    for _ in range(2):
        for p in params:
            p.grad = torch.randn_like(p)
        ref.step()
    offs, total = flat_layout([p.numel() for p in params])
    layout = [(o, tuple(p.shape)) for o, p in zip(offs, params)]
    fused = FusedAdadelta(torch.zeros(total), lr=1.0, layout=layout)
    fused.load_state_dict(ref.state_dict())
    assert fused.lr == 0.8 and fused.steps == 2
    for (o, shape), p in zip(layout, params):
        n = p.numel()
        assert torch.equal(fused.square_avg[o:o + n].view(shape), ref.state[p]["square_avg"])
        assert torch.equal(fused.acc_delta[o:o + n].view(shape), ref.state[p]["acc_delta"])
    assert float(fused.square_avg.sum()) == pytest.approx(float(sum(ref.state[p]["square_avg"].sum() for p in params)))    # padding stays zero
    fresh = torch.optim.Adadelta(params, lr=1.0)
    fresh.load_state_dict(fused.state_dict())
    for p in params:
        assert torch.equal(fresh.state[p]["square_avg"], ref.state[p]["square_avg"]) and torch.equal(fresh.state[p]["acc_delta"], ref.state[p]["acc_delta"])
    assert fresh.param_groups[0]["lr"] == 0.8
    import io
    buf = io.BytesIO()
    torch.save(fused.state_dict(), buf)                               # what Checkpointer does
    buf.seek(0)
    again = FusedAdadelta(torch.zeros(total), layout=layout)
    again.load_state_dict(torch.load(buf))
    assert torch.equal(again.square_avg, fused.square_avg) and torch.equal(again.acc_delta, fused.acc_delta)


def test_clip_group_planner():
    """train.plan_clip_groups: no cut for a homogeneous minibatch; the clips holding full-length bars become the second group when
    there are a few of them; inside the first group the clips with the longest rows come first (stable), the second keeps its order."""
    import numpy as np
    from piano_a2s_amd import spec, synthetic
    from piano_a2s_amd.train import plan_clip_groups
    cfg = spec.default_cfg()

    def untils(tail, seed):
        b = synthetic.make_batch(256, cfg, seed, frames=4, full_tail=tail)
        iu, il = torch.arange(1, 399), torch.arange(1, 190)
        return ((b[3] != 147).long() * iu).amax(-1).numpy(), ((b[5] != 147).long() * il).amax(-1).numpy()

    up, lo = untils(0.0, 3)
    order, n_main = plan_clip_groups(up, lo)
    assert n_main == 256 and order.tolist() == list(range(256))
    up, lo = untils(0.01, 3)
    order, n_main = plan_clip_groups(up, lo)
    assert 0 < 256 - n_main <= 128 and sorted(order.tolist()) == list(range(256))
    longest = np.maximum(up.max(1), lo.max(1))
    main = order[:n_main].tolist()
    assert all(longest[a] > longest[b] or (longest[a] == longest[b] and a < b) for a, b in zip(main[:-1], main[1:]))
    assert order[n_main:].tolist() == sorted(order[n_main:].tolist())
    assert up[order[:n_main]].max() <= 121 and all(up[c].max() == 398 for c in order[n_main:])
