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
