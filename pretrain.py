#!/usr/bin/env python3
"""Pre-training entry point: ``python pretrain.py hparams/pretrain.yaml --workspace=... --soundfont_folder=...`` or
``torchrun --nproc_per_node=N pretrain.py hparams/pretrain.yaml ...`` (one process per GPU, RCCL).

Same command line, yaml keys and recipe hooks as the reference's pretrain.py (:251-305); the recipe class lives in
piano_a2s_amd/recipe.py.  ``--synthetic_clips=N`` trains on N seeded synthetic clips instead of a rendered corpus."""
import sys

from piano_a2s_amd.recipe import ASR, sb, write_run_summary

try:
    from hyperpyyaml import load_hyperpyyaml
except Exception:  # noqa: BLE001
    from piano_a2s_amd.hyperyaml import load_hyperpyyaml


def main(argv):
    hparams_file, run_opts, overrides = sb.parse_arguments(argv)
    sb.utils.distributed.ddp_init_group(run_opts)
    with open(hparams_file) as fin:
        hparams = load_hyperpyyaml(fin, overrides)
    sb.create_experiment_directory(experiment_directory=hparams["output_folder"], hyperparams_to_save=hparams_file, overrides=overrides)

    n_syn = int(hparams.get("synthetic_clips", 0) or 0)
    if n_syn:
        from datasets.syn import SyntheticClips, SyntheticWaveClips
        if hparams.get("online_vqt"):                      # raw waveforms -> GPU VQT -> model (instead of cached spectrograms)
            SyntheticClips = SyntheticWaveClips
        cfg = hparams["transcription"].cfg
        syn = dict(frames=int(hparams.get("synthetic_frames") or hparams["max_frame_num"]))
        if hparams.get("synthetic_lengths"):
            syn.update(upper_range=tuple(hparams["synthetic_lengths"][0]), lower_range=tuple(hparams["synthetic_lengths"][1]))
        train_set = SyntheticClips(cfg, n_syn, seed=hparams["seed"], **syn)
        valid_set = SyntheticClips(cfg, max(1, n_syn // 8), seed=hparams["seed"] + 10_000, **syn)
        test_set = SyntheticClips(cfg, max(1, n_syn // 8), seed=hparams["seed"] + 20_000, **syn)
    else:
        from datasets.syn import TestDataset, TrainDataset
        test_versions = range(4) if hparams["midi_syn"] == "epr" else [0]      # score + 3 composers' renderings for "epr"
        train_set = TrainDataset(hparams, "train", run_opts["device"], range(10))
        valid_set = TestDataset(hparams, "valid", run_opts["device"], test_versions)
        test_set = TestDataset(hparams, "test", run_opts["device"], test_versions)

    brain = ASR(modules=hparams["modules"], opt_class=hparams["opt_class"], hparams=hparams, run_opts=run_opts,
                checkpointer=hparams["checkpointer"])
    brain.fit(brain.hparams.epoch_counter, train_set, valid_set,
              train_loader_kwargs=hparams["train_dataloader_opts"], valid_loader_kwargs=hparams["valid_dataloader_opts"])
    brain.evaluate(test_set, test_loader_kwargs=hparams["test_dataloader_opts"], min_key="WER")
    write_run_summary(brain, hparams)
    return brain


if __name__ == "__main__":
    main(sys.argv[1:])
