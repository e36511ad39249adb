"""pretrain.py end to end on the MI355X with the HIP model (reduced dims, synthetic clips): the recipe's fit_batch takes the fused
HIP step, validation/test decode greedily, metrics/checkpoint/results are written; a second epoch must not increase the training loss
on this tiny overfit-able set (sanity that the optimizer really updates the parameters the forward uses)."""
import json
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_pretrain_on_gpu(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import pretrain
    args = [os.path.join(ROOT, "hparams", "pretrain.yaml"), "--device=cuda:0", f"--workspace={tmp_path}", "--soundfont_folder=/none",
            "--synthetic_clips=8", "--hidden_size=32", "--conv_feature_size=32", "--bins_per_octave=24", "--n_octaves=1", "--max_length=(12, 8)",
            "--synthetic_frames=41", "--synthetic_lengths=[[3, 10], [2, 7]]", "--batch_size=4", "--number_of_epochs=3"]
    brain = pretrain.main(args)
    assert brain._fused, "the recipe must run the fused HIP training step on the GPU"
    out = os.path.join(str(tmp_path), "1234", "pretrain.epr")
    all_lines = [l for l in open(os.path.join(out, "train_log.txt")).read().splitlines() if l.startswith("epoch")]
    assert len(all_lines) == 4 and all_lines[-1].startswith("epoch: None")      # 3 epochs (train+valid) + the final TEST evaluation
    lines = all_lines[:3]
    train_loss = [float(l.split("train loss: ")[1].split(",")[0]) for l in lines]
    assert train_loss[-1] < train_loss[0], f"training loss did not go down: {train_loss}"
    assert len(os.listdir(os.path.join(out, "save"))) == 1
    rec = json.load(open(os.path.join(out, "results", "test", os.listdir(os.path.join(out, "results", "test"))[0])))
    assert len(rec["pred"]) == 5


def test_pretrain_with_online_vqt(tmp_path):
    """waveform -> GPU VQT (480 bins) -> model: full-width front-end, reduced model behind it, 1 epoch."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import pretrain
    args = [os.path.join(ROOT, "hparams", "pretrain.yaml"), "--device=cuda:0", f"--workspace={tmp_path}", "--soundfont_folder=/none",
            "--synthetic_clips=4", "--online_vqt=True", "--hidden_size=32", "--conv_feature_size=32", "--max_length=(12, 8)",
            "--synthetic_frames=101", "--synthetic_lengths=[[3, 10], [2, 7]]", "--batch_size=2", "--number_of_epochs=1"]
    brain = pretrain.main(args)
    assert brain._fused and brain.modules.transcription.cfg["freq_bins"] == 480
    assert os.path.exists(os.path.join(str(tmp_path), "1234", "pretrain.epr", "train_log.txt"))


SMALL = ["--hidden_size=32", "--conv_feature_size=32", "--bins_per_octave=24", "--n_octaves=1", "--max_length=(12, 8)", "--max_frame_num=41"]


def test_pretrain_then_finetune_from_on_disk_corpora(tmp_path):
    """BASELINE.json configs[3] on one GPU: pretrain.py on a rendered-corpus folder, then finetune.py on an ASAP-shaped folder, both read
    through the on-disk dataset classes (TrainDataset / TestDataset / ASAPDataset) with the HIP model and the fused step.
    Checks what reference finetune.py:44,251-262 define: fixed teacher-forcing ratio 0.6, training starts from the copied pre-training
    checkpoint (weights AND Adadelta accumulators), the seeded WER = 100 record is replaced, the test split doubles as validation split."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import yaml
    import finetune
    import pretrain
    from piano_a2s_amd import spec, train
    from tests import disk_corpus
    cfg = spec.default_cfg(freq_bins=24, conv_feature_size=32, hidden_size=32, max_length=(12, 8))
    ws = str(tmp_path)
    kw = dict(upper_range=(3, 10), lower_range=(2, 7), full_tail=0.1)
    disk_corpus.write_rendered_corpus(os.path.join(ws, "feature.score"), cfg, "train", range(10), 3, 41, seed=1, soundfonts=("pianoA",), **kw)
    for split, seed in (("valid", 2), ("test", 3)):
        disk_corpus.write_rendered_corpus(os.path.join(ws, "feature.score"), cfg, split, [0], 2, 41, seed=seed, **kw)
    disk_corpus.write_asap_corpus(os.path.join(ws, "feature.asap"), cfg, "train", 6, 41, seed=4, **kw)
    asap_test = disk_corpus.write_asap_corpus(os.path.join(ws, "feature.asap"), cfg, "test", 3, 41, seed=5, **kw)
    common = ["--device=cuda:0", f"--workspace={ws}", "--midi_syn=score", "--batch_size=2"] + SMALL
    pbrain = pretrain.main([os.path.join(ROOT, "hparams", "pretrain.yaml"), "--soundfont_folder=/none", "--number_of_epochs=2"] + common)
    assert pbrain._fused
    pout = os.path.join(ws, "1234", "pretrain.score")
    pck = os.listdir(os.path.join(pout, "save"))
    assert len(pck) == 1 and "optimizer.ckpt" in os.listdir(os.path.join(pout, "save", pck[0]))
    popt = torch.load(os.path.join(pout, "save", pck[0], "optimizer.ckpt"), map_location="cpu")
    assert len(popt["state"]) == 83 and float(popt["state"][0]["square_avg"].abs().sum()) > 0, "the fused step's Adadelta accumulators must be in the checkpoint"
    pmodel = torch.load(os.path.join(pout, "save", pck[0], "model.ckpt"), map_location="cpu")
    rec = json.load(open(os.path.join(pout, "results", "test", sorted(os.listdir(os.path.join(pout, "results", "test")))[0])))
    assert rec["style"] == "pop" and rec["soundfont"] in ("pianoA", "pianoB") and rec["target_path"].endswith(os.path.join("test", "0", "target", "Chunk00.pkl"))

    seen = {"tf": [], "first": None}
    orig_call = train.TrainStep.__call__

    def spy(self, batch, tf, *a, **k):
        if seen["first"] is None:           # state the very first fine-tuning step starts from
            seen["first"] = (self.flat.detach().cpu().clone(), self.opt.square_avg.detach().cpu().clone(), {n: p.detach().cpu().clone() for n, p in self.model.named_parameters()})
        seen["tf"].append(tf)
        return orig_call(self, batch, tf, *a, **k)
    train.TrainStep.__call__ = spy
    try:
        fbrain = finetune.main([os.path.join(ROOT, "hparams", "finetune.yaml"), "--asap_folder=/none", "--mv2h_bin=/none", "--number_of_epochs=1"] + common)
    finally:
        train.TrainStep.__call__ = orig_call
    assert fbrain._fused and fbrain.finetune
    assert seen["tf"] and all(tf == 0.6 for tf in seen["tf"]), seen["tf"]                       # finetune.py:44 / finetune.yaml:43
    assert len(seen["tf"]) == 3                                                                  # 6 ASAP train clips / batch 2, 1 epoch
    _, sq0, params0 = seen["first"]
    for n, p in params0.items():
        assert torch.equal(p, pmodel["0." + n]), f"fine-tuning did not start from the pre-trained {n}"
    assert float(sq0.abs().sum()) > 0, "fine-tuning must continue the pre-training Adadelta accumulators (optimizer.ckpt)"
    fout = os.path.join(ws, "1234", "finetune.score")
    metas = [yaml.safe_load(open(os.path.join(fout, "save", d, "CKPT.yaml"))) for d in os.listdir(os.path.join(fout, "save"))]
    assert len(metas) == 1 and metas[0]["WER"] < 100                                             # the seeded WER = 100 record was replaced
    res = sorted(os.listdir(os.path.join(fout, "results", "test")))
    assert [r[:-5] for r in res] == sorted(asap_test)                                            # result ids = ASAP clip names (finetune.py clip id)
    assert sorted(os.listdir(os.path.join(fout, "results", "valid"))) == res                     # valid split == test split (finetune.py:262)
    rec = json.load(open(os.path.join(fout, "results", "test", res[0])))
    assert rec["target_path"] == os.path.join(ws, "feature.asap", "test", "target", res[0][:-5] + ".pkl") and "style" not in rec
    summary = json.load(open(os.path.join(fout, "run_summary.json")))
    assert summary["fused_hip_step"] and summary["optimizer_steps"] == 3 and summary["world_size"] == 1


def test_checkpoint_resume_continues_the_uninterrupted_run(tmp_path):
    """Save -> new process state -> recover -> next step == the uninterrupted run: model, BatchNorm buffers and the fused step's Adadelta
    accumulators all travel through the SpeechBrain-layout checkpoint directory.  (The step is deterministic up to the order of a few
    float atomics -- embedding-gradient scatter -- so "equal" means within that run-to-run spread, 1e-6 of the largest parameter; the
    control shows what the round-1 behaviour -- accumulators silently restarted from zero -- does to the same step: > 20x that bar, 1.4e-4 measured.)"""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import random
    import models
    from piano_a2s_amd import sb_compat, spec, synthetic, train
    dev = torch.device("cuda:0")
    cfg = spec.default_cfg(freq_bins=24, conv_feature_size=32, hidden_size=32, max_length=(12, 8))
    st = spec.procedural_state(cfg, 11, eos_bias=3.0, lively=True)
    batches = [[t.to(dev) if torch.is_tensor(t) else t for t in synthetic.make_batch(3, cfg, s, frames=41, upper_range=(3, 10), lower_range=(2, 7))] for s in (5, 6, 7)]

    def fresh():
        m = models.ScoreTranscription(**cfg)
        m.load_state_dict(st)
        m = m.to(dev).train()
        return m, train.TrainStep(m, dropout=False)

    m, step = fresh()
    for i, b in enumerate(batches):
        step(b, 0.7, rng=random.Random(i))
    torch.cuda.synchronize()
    want = (step.flat.cpu().clone(), step.opt.square_avg.cpu().clone(), step.opt.acc_delta.cpu().clone(), {k: v.cpu().clone() for k, v in m.named_buffers()})

    m, step = fresh()
    for i, b in enumerate(batches[:2]):
        step(b, 0.7, rng=random.Random(i))
    ck = sb_compat.Checkpointer(str(tmp_path / "save"), {"model": torch.nn.ModuleList([m]), "optimizer": step.opt})
    ck.save_and_keep_only(meta={"WER": 0.5}, min_keys=["WER"])
    scale = float(want[0].abs().max())
    for with_optimizer in (True, False):
        m2, step2 = fresh()                                   # "new process": fresh weights, zero accumulators
        rec = {"model": torch.nn.ModuleList([m2])}
        if with_optimizer:
            rec["optimizer"] = step2.opt
        assert sb_compat.Checkpointer(str(tmp_path / "save"), rec).recover_if_possible(device="cuda:0") is not None
        assert step2.opt.steps == (2 if with_optimizer else 0)
        step2(batches[2], 0.7, rng=random.Random(2))
        torch.cuda.synchronize()
        err = float((step2.flat.cpu() - want[0]).abs().max()) / scale
        if with_optimizer:
            assert err <= 1e-6, f"parameters after resume differ from the uninterrupted run by {err:.3e} of the largest parameter"
            for name, got, ref in (("square_avg", step2.opt.square_avg, want[1]), ("acc_delta", step2.opt.acc_delta, want[2])):
                assert float((got.cpu() - ref).abs().max()) <= 1e-3 * float(ref.abs().max()), name
            for k, v in m2.named_buffers():
                assert torch.allclose(v.cpu().float(), want[3][k].float(), rtol=1e-6, atol=1e-7), k
        else:
            assert err > 2e-5, f"control: without the optimizer state the resumed step should visibly differ (got {err:.3e})"


def test_nonfinite_loss_skips_the_update_and_patience_aborts(tmp_path):
    """SpeechBrain check_gradients semantics on the fused path (reference pretrain.py:126): a non-finite loss leaves the parameters and the
    Adadelta state untouched, is counted, and the run aborts once `nonfinite_patience` is exceeded."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import pretrain
    from piano_a2s_amd import train
    args = [os.path.join(ROOT, "hparams", "pretrain.yaml"), "--device=cuda:0", f"--workspace={tmp_path}", "--soundfont_folder=/none",
            "--synthetic_clips=8", "--synthetic_frames=41", "--synthetic_lengths=[[3, 10], [2, 7]]", "--batch_size=2", "--number_of_epochs=1",
            "--nonfinite_patience=2"] + SMALL
    orig_call = train.TrainStep.__call__
    log = []

    def poisoned(self, batch, tf, *a, **k):
        with torch.no_grad():                                  # an infinite bias makes every time-signature log-probability NaN
            self.model.get_parameter("decoder.time_sig_out.4.bias").fill_(float("inf"))
        before = self.flat.detach().clone()
        out = orig_call(self, batch, tf, *a, **k)
        log.append((torch.equal(before, self.flat), float(self.opt.ctl[2]), float(self.opt.square_avg.abs().sum())))
        return out
    train.TrainStep.__call__ = poisoned
    try:
        with pytest.raises(ValueError, match="patience"):
            pretrain.main(args)
    finally:
        train.TrainStep.__call__ = orig_call
    assert len(log) == 3 and all(unchanged and applied == 0.0 and sq == 0.0 for unchanged, applied, sq in log), log


def test_pretrain_under_torchrun_single_rank(tmp_path):
    """`python -m torch.distributed.run --nproc-per-node 1 pretrain.py ...`: launcher -> ddp_init_group (RCCL, world 1) ->
    DistributedSampler -> fused step with its gradient all-reduces -> rank-0 checkpoint/results, end to end on the one GPU of the box
    (BASELINE.json configs[2] with N = 1; the launcher is a child process started before anything here touches the GPU runtime)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import subprocess
    import sys
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr=127.0.0.1", "--master-port=29631",
           os.path.join(ROOT, "pretrain.py"), os.path.join(ROOT, "hparams", "pretrain.yaml"), f"--workspace={tmp_path}", "--soundfont_folder=/none",
           "--synthetic_clips=8", "--synthetic_frames=41", "--synthetic_lengths=[[3, 10], [2, 7]]", "--batch_size=2", "--number_of_epochs=1"] + SMALL
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    out = os.path.join(str(tmp_path), "1234", "pretrain.epr")
    summary = json.load(open(os.path.join(out, "run_summary.json")))
    assert summary["fused_hip_step"] and summary["world_size"] == 1 and summary["backend"] == "nccl"
    assert summary["optimizer_steps"] == 4 and summary["gradient_allreduces"] == 2 * 4           # two overlapped slices per step (train.GradientExchange)
    assert len(os.listdir(os.path.join(out, "save"))) == 1


def test_resume_keeps_the_annealed_learning_rate(tmp_path, monkeypatch):
    """NewBob anneals the lr at the end of every evaluation here (threshold forced); the annealed value must be what optimizer.ckpt saves
    and what the fused step trains with after a resume (reference: the torch optimizer is the recoverable and carries its lr).  ADVICE
    r2: the fused path used to restart every resumed run -- and every finetune seeded from a pretraining save/ -- at the yaml's lr = 1.
    The WER is forced to fall from evaluation to evaluation so that the checkpointer's keep-best rule keeps the LATEST checkpoint."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import pretrain
    from piano_a2s_amd import metrics, sb_compat, train
    if pretrain.sb is not sb_compat:
        pytest.skip("SpeechBrain installed: its own scheduler class is in use")
    orig_init = sb_compat.NewBobScheduler.__init__

    def always_anneal(self, *a, **k):
        orig_init(self, *a, **k)
        self.improvement_threshold = 10.0                      # no evaluation improves the WER tenfold: anneal after each but the first
    monkeypatch.setattr(sb_compat.NewBobScheduler, "__init__", always_anneal)
    calls = [0]
    orig_wer = metrics.corpus_wer

    def falling_wer(*a, **k):
        calls[0] += 1
        return 1.0 / calls[0], orig_wer(*a, **k)[1]
    monkeypatch.setattr(metrics, "corpus_wer", falling_wer)
    seen = []
    orig_call = train.TrainStep.__call__

    def spy(self, batch, tf, *a, **k):
        seen.append(float(self.opt.lr))
        return orig_call(self, batch, tf, *a, **k)
    monkeypatch.setattr(train.TrainStep, "__call__", spy)
    args = [os.path.join(ROOT, "hparams", "pretrain.yaml"), "--device=cuda:0", f"--workspace={tmp_path}", "--soundfont_folder=/none",
            "--synthetic_clips=4", "--synthetic_frames=41", "--synthetic_lengths=[[3, 10], [2, 7]]", "--batch_size=2"] + SMALL
    # evaluations: after epoch 1 (lr 1 -> 1: first metric), epoch 2 (1 -> .8), epoch 3 (.8 -> .64), the final TEST stage (.64 -> .512;
    # the reference's on_stage_end anneals and checkpoints there too, pretrain.py:179-186)
    pretrain.main(args + ["--number_of_epochs=3"])
    per_epoch = 2
    assert seen == [1.0] * (2 * per_epoch) + [0.8] * per_epoch, seen
    brain = pretrain.main(args + ["--number_of_epochs=4"])    # a new run in the same workspace: recovers, trains epoch 4 only
    resumed = seen[3 * per_epoch:]
    assert len(resumed) == per_epoch and all(abs(v - 0.8 ** 3) < 1e-12 for v in resumed), f"lr after resume {resumed}, expected 0.8^3 (saved by the last evaluation)"
    assert abs(brain.optimizer.param_groups[0]["lr"] - 0.8 ** 5) < 1e-12 and abs(brain._fused.opt.lr - 0.8 ** 5) < 1e-12
