#!/usr/bin/env python3
"""Fine-tuning entry point (real-audio clips, ASAP layout): ``python finetune.py hparams/finetune.yaml --workspace=... ...``.

As the reference's finetune.py (:230-294): starts from the pre-training checkpoints by copying ``<pretrain>/save`` into the new
output folder and resetting their recorded WER so that the first fine-tuned epoch is kept; fixed teacher-forcing ratio; the test
split doubles as validation split."""
import os
import shutil
import sys

from piano_a2s_amd.recipe import ASR, sb, write_run_summary
from utilities import load, save

try:
    from hyperpyyaml import load_hyperpyyaml
except Exception:  # noqa: BLE001
    from piano_a2s_amd.hyperyaml import load_hyperpyyaml


class FinetuneASR(ASR):
    finetune = True


def seed_from_pretraining(pretrained_output_folder, output_folder):
    src, dst = os.path.join(pretrained_output_folder, "save"), os.path.join(output_folder, "save")
    if os.path.isdir(src):
        shutil.copytree(src, dst, dirs_exist_ok=True)
    if os.path.isdir(dst):
        for folder in os.listdir(dst):
            meta_file = os.path.join(dst, folder, "CKPT.yaml")
            if os.path.exists(meta_file):
                meta = load(meta_file)
                meta["WER"] = 100                     # any fine-tuned epoch beats it
                save(meta, meta_file)


def main(argv):
    hparams_file, run_opts, overrides = sb.parse_arguments(argv)
    sb.utils.distributed.ddp_init_group(run_opts)
    with open(hparams_file) as fin:
        hparams = load_hyperpyyaml(fin, overrides)
    sb.create_experiment_directory(experiment_directory=hparams["output_folder"], hyperparams_to_save=hparams_file, overrides=overrides)
    # rank 0 copies, every rank waits: the others must not look for checkpoints in a half-copied save/ (the reference runs the cp
    # synchronously on every rank, finetune.py:251-258); Brain.on_fit_start then broadcasts rank 0's recovered parameters
    sb.utils.distributed.run_on_main(seed_from_pretraining, args=[hparams["pretrained_output_folder"], hparams["output_folder"]])

    n_syn = int(hparams.get("synthetic_clips", 0) or 0)
    if n_syn:
        from datasets.syn import SyntheticClips
        cfg = hparams["transcription"].cfg
        syn = dict(frames=int(hparams.get("synthetic_frames") or hparams["max_frame_num"]))
        if hparams.get("synthetic_lengths"):
            syn.update(upper_range=tuple(hparams["synthetic_lengths"][0]), lower_range=tuple(hparams["synthetic_lengths"][1]))
        train_set = SyntheticClips(cfg, n_syn, seed=hparams["seed"], **syn)
        valid_set = test_set = SyntheticClips(cfg, max(1, n_syn // 8), seed=hparams["seed"] + 10_000, **syn)
    else:
        from datasets.asap import ASAPDataset
        train_set = ASAPDataset(hparams, "train", run_opts["device"])
        valid_set = test_set = ASAPDataset(hparams, "test", run_opts["device"])

    brain = FinetuneASR(modules=hparams["modules"], opt_class=hparams["opt_class"], hparams=hparams, run_opts=run_opts,
                        checkpointer=hparams["checkpointer"])
    brain.fit(brain.hparams.epoch_counter, train_set, valid_set,
              train_loader_kwargs=hparams["train_dataloader_opts"], valid_loader_kwargs=hparams["valid_dataloader_opts"])
    brain.evaluate(test_set, test_loader_kwargs=hparams["test_dataloader_opts"], min_key="WER")
    write_run_summary(brain, hparams)
    return brain


if __name__ == "__main__":
    main(sys.argv[1:])
