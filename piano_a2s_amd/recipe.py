"""The training recipe: a Brain subclass with the reference's hooks (reference pretrain.py:31-214 / finetune.py:30-193).

Shared by pretrain.py and finetune.py (they differ in the teacher-forcing schedule, the clip id and the result record).
`fit_batch` keeps the reference's semantics (forward, 4-term NLL, backward, check_gradients, optimizer step, zero_grad) but, when the
transcription module is the HIP model, runs them as the fused step of piano_a2s_amd.train (loss terms stay on the device)."""
import os

import numpy as np
import torch

try:                                    # the real thing when present, the compat slice otherwise
    import speechbrain as sb
except Exception:                       # noqa: BLE001
    from piano_a2s_amd import sb_compat as sb

from data_processing.humdrum import LabelsMultiple
from piano_a2s_amd import metrics
from utilities import load, mkdirs, save

labels = LabelsMultiple(extended=True)


def _to_device(batch, device):
    return [t.to(device, non_blocking=True) if torch.is_tensor(t) else t for t in batch]


_VQT = {}


def _features(batch, device):
    """Online front-end: when the loader yields raw 16 kHz waveforms (B, N) instead of cached spectrograms
    (B, 1, T, F), the VQT runs on the GPU in front of the model (piano_a2s_amd.vqt; the reference caches librosa features offline)."""
    batch = _to_device(batch, device)
    if torch.is_tensor(batch[0]) and batch[0].dim() == 2:
        from piano_a2s_amd.vqt import VQT
        if device not in _VQT:
            _VQT[device] = VQT(torch.device(device))
        batch[0] = _VQT[device](batch[0])
    return batch


class ASR(sb.Brain):
    finetune = False

    # ------------------------------------------------------------------ forward / objective
    def compute_forward(self, batch, stage):
        batch = _features(batch, self.device)
        spectrogram, ts_t, key_t, up_t, up_len, lo_t, lo_len = batch[:7]
        if stage == sb.Stage.TRAIN:
            return self.modules.transcription(spectrogram=spectrogram, inference=False,
                                              ground_truth=[ts_t, key_t, up_t, up_len, lo_t, lo_len],
                                              teacher_forcing_ratio=self.teacher_forcing_ratio, device=self.device)
        return self.modules.transcription(spectrogram=spectrogram, inference=True, ground_truth=None,
                                          teacher_forcing_ratio=0., device=self.device)

    def compute_objectives(self, predictions, batch, stage):
        batch = _to_device(batch, self.device)
        _, ts_t, key_t, up_t, _, lo_t, _, names, versions = batch
        ts_o, key_o, up_o, lo_o = predictions
        hp = self.hparams
        time_loss = hp.loss_time_sig(ts_o.permute(0, 2, 1), ts_t)
        key_loss = hp.loss_key(key_o.permute(0, 2, 1), key_t)
        flat = lambda o, t: (o.reshape(o.shape[0] * o.shape[1], -1, o.shape[3]).permute(0, 2, 1), t.reshape(t.shape[0] * t.shape[1], -1))
        upper_loss = hp.loss_score(*flat(up_o, up_t))
        lower_loss = hp.loss_score(*flat(lo_o, lo_t))
        self._record_losses(time_loss, key_loss, upper_loss, lower_loss)
        if stage != sb.Stage.TRAIN:
            self._record_predictions(predictions, (ts_t, key_t, up_t, lo_t), names, versions)
        return time_loss + key_loss + upper_loss + lower_loss

    def _record_losses(self, *terms):
        for store, t in zip((self.time_losses, self.key_losses, self.upper_losses, self.lower_losses), terms):
            store.append(float(t.detach()))

    def _clip_id(self, name, version):
        return name if self.finetune else "~".join([str(int(version)), name])

    def _record_predictions(self, predictions, targets, names, versions):
        ts_o, key_o, up_o, lo_o = predictions
        ts_t, key_t, up_t, lo_t = targets
        up_ids, lo_ids = up_o.argmax(-1).cpu().numpy(), lo_o.argmax(-1).cpu().numpy()
        ts_ids, key_ids = ts_o.argmax(-1).cpu().numpy(), key_o.argmax(-1).cpu().numpy()
        up_t, lo_t, ts_t, key_t = up_t.cpu().numpy(), lo_t.cpu().numpy(), ts_t.cpu().numpy(), key_t.cpu().numpy()
        for b, name in enumerate(names):
            cid = self._clip_id(name, versions[b])
            self.upper_pred[cid] = [metrics.unpad(r).tolist() for r in up_ids[b]]
            self.upper_target[cid] = [metrics.unpad(r).tolist() for r in up_t[b]]
            self.lower_pred[cid] = [metrics.unpad(r).tolist() for r in lo_ids[b]]
            self.lower_target[cid] = [metrics.unpad(r).tolist() for r in lo_t[b]]
            self.key_pred[cid], self.key_target[cid] = key_ids[b].tolist(), key_t[b].tolist()
            self.time_sig_pred[cid], self.time_sig_target[cid] = ts_ids[b].tolist(), ts_t[b].tolist()

    # ------------------------------------------------------------------ training step
    def _fused_step(self):
        """The fused HIP step needs the HIP model, Adadelta and the yaml's NLL objective; otherwise fall back to the generic path."""
        if getattr(self, "_fused", None) is None:
            model = self.modules.transcription
            opt = getattr(self, "optimizer", None)
            if opt is None:
                return False              # no optimizer yet (evaluate() before fit()): decide once there is one, do not cache "no"
            self._fused = False
            if hasattr(model, "flatten_") and isinstance(opt, torch.optim.Adadelta) and str(self.device).startswith("cuda"):
                from piano_a2s_amd import train
                g = opt.param_groups[0]
                self._fused = train.TrainStep(model, lr=g["lr"], rho=g["rho"], eps=g["eps"], max_grad_norm=self.max_grad_norm)
        return self._fused

    def init_optimizers(self):
        """As sb.Brain.init_optimizers (the optimizer becomes the checkpointer's `optimizer` recoverable) -- but when the fused HIP step
        is the one that trains, ITS Adadelta accumulators are the optimizer state: FusedAdadelta is registered instead of the idle
        torch object (same optimizer.ckpt format), before on_fit_start recovers, so a resumed run continues with the saved
        square_avg / acc_delta instead of silently restarting them from zero."""
        super().init_optimizers()
        fused = self._fused_step()
        if fused and self.checkpointer is not None:
            self.checkpointer.add_recoverable("optimizer", fused.opt)

    def on_fit_start(self):
        """Recovery restores the learning rate with the optimizer state (optimizer.ckpt holds the NewBob-annealed lr, as in the
        reference, whose torch optimizer IS the recoverable).  When the fused step trains, the recoverable is fused.opt: its recovered
        lr is the truth, and the idle torch optimizer (which update_learning_rate also addresses) is brought in line with it -- without
        this, the first epoch after a resume, or after finetune.py's copy of the pretraining save/, ran at the yaml's initial lr."""
        super().on_fit_start()
        fused = self._fused_step()
        if fused and self.optimizer is not None:
            for g in self.optimizer.param_groups:
                g["lr"] = fused.opt.lr

    def fit_batch(self, batch):
        fused = self._fused_step()
        if not fused:
            return super().fit_batch(batch)
        fused(_features(batch, self.device), self.teacher_forcing_ratio)
        *terms, applied = fused.report()                              # one small D2H per step (the reference does four)
        self._record_losses(*[torch.tensor(t) for t in terms])
        loss = torch.tensor(sum(terms))
        if not applied:
            # the device skipped the update (non-finite loss on some rank or non-finite gradient norm): SpeechBrain's check_gradients
            # counts these and gives up after `nonfinite_patience` of them (reference pretrain.py:126)
            self.nonfinite_count += 1
            if self.nonfinite_count > self.nonfinite_patience:
                raise ValueError("Loss is not finite and patience is exhausted. To debug, wrap `fit()` with autograd's "
                                 "`detect_anomaly()`.")
        return loss

    def evaluate_batch(self, batch, stage):
        with torch.no_grad():
            predictions = self.compute_forward(batch, stage=stage)
            loss = self.compute_objectives(predictions, batch, stage=stage)
        return loss.detach()

    # ------------------------------------------------------------------ stage hooks
    def on_stage_start(self, stage, epoch):
        self.time_losses, self.key_losses, self.upper_losses, self.lower_losses = [], [], [], []
        if stage != sb.Stage.TRAIN:
            self.upper_pred, self.upper_target, self.lower_pred, self.lower_target = {}, {}, {}, {}
            self.key_pred, self.key_target, self.time_sig_pred, self.time_sig_target = {}, {}, {}, {}
            for split in ("valid", "test"):
                mkdirs(os.path.join(self.hparams.output_folder, "results", split))
        self.time_sig_list = load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                               "data_processing", "metadata", "time_signature_list.json"))
        if stage != sb.Stage.TRAIN:
            self.teacher_forcing_ratio = 0.
        elif self.finetune:
            self.teacher_forcing_ratio = self.hparams.teacher_forcing_ratio
        else:                                                          # exponential decay per epoch (pretrain.py:151)
            self.teacher_forcing_ratio = self.hparams.teacher_forcing_ratio * self.hparams.teacher_forcing_decay ** epoch

    def on_stage_end(self, stage, stage_loss, epoch):
        stats = {"loss": stage_loss, "time_loss": np.mean(self.time_losses), "key_loss": np.mean(self.key_losses),
                 "upper_loss": np.mean(self.upper_losses), "lower_loss": np.mean(self.lower_losses)}
        if not self.finetune:
            stats["teacher_forcing_ratio"] = self.teacher_forcing_ratio
        if stage == sb.Stage.TRAIN:
            self.train_stats = stats
            return
        if not hasattr(self, "train_stats"):
            self.train_stats = {"loss": -1}
        inv = labels.labels_map_inv
        wer_up, wer_up_d = metrics.corpus_wer(self.upper_pred, self.upper_target, inv)
        wer_lo, wer_lo_d = metrics.corpus_wer(self.lower_pred, self.lower_target, inv)
        key_f1, key_f1_d = metrics.corpus_f1(self.key_pred, self.key_target)
        time_f1, time_f1_d = metrics.corpus_f1(self.time_sig_pred, self.time_sig_target)
        stats.update(key_f1=key_f1, time_f1=time_f1, WER_upper=wer_up, WER_lower=wer_lo, WER=(wer_up + wer_lo) / 2)
        old_lr, new_lr = self.hparams.lr_annealing(stats["WER"])
        sb.nnet.schedulers.update_learning_rate(self.optimizer, new_lr)
        fused = self._fused_step()
        if fused:                                                      # the lr the fused step uses AND the one optimizer.ckpt saves below
            sb.nnet.schedulers.update_learning_rate(fused.opt, new_lr)
        self.hparams.train_logger.log_stats(stats_meta={"epoch": epoch, "lr": old_lr}, train_stats=self.train_stats, valid_stats=stats)
        self.checkpointer.save_and_keep_only(meta={"loss": stats["loss"], "WER": stats["WER"]}, min_keys=["WER"])
        self.last_stats = stats
        split = "test" if stage == sb.Stage.TEST else "valid"
        for cid in self.upper_pred:
            pred = [[self.key_pred[cid][i] - 6, self.time_sig_list[self.time_sig_pred[cid][i]], self.lower_pred[cid][i], self.upper_pred[cid][i]]
                    for i in range(len(self.upper_pred[cid]))]
            record = {"pred": pred, "wer_upper": wer_up_d[cid], "wer_lower": wer_lo_d[cid], "key_f1": key_f1_d[cid], "time_f1": time_f1_d[cid]}
            record.update(self._clip_record(cid, split))
            save(record, os.path.join(self.hparams.output_folder, "results", split, f"{cid}.json"))

    def _clip_record(self, cid, split):
        ff = self.hparams.feature_folder
        if self.finetune:
            return {"target_path": os.path.join(ff, "test", "target", f"{cid}.pkl")}
        version, chunk, soundfont = (cid.split("~") + ["", ""])[:3]
        info_path = os.path.join(ff, split, version, "info", f"{chunk}.json")
        composer = load(info_path).get("composer") if os.path.exists(info_path) else None
        return {"style": "classical" if chunk[:1].islower() else "pop", "soundfont": soundfont, "composer": composer,
                "target_path": os.path.join(ff, split, version, "target", f"{chunk}.pkl")}


def write_run_summary(brain, hparams):
    """<output_folder>/run_summary.json (rank 0): how the run executed -- which training step, how many ranks over which backend, how many
    optimizer steps and gradient all-reduces.  Not part of the reference's artefacts; it is what lets a launch under torchrun be
    checked end to end (tests/test_gpu_recipe.py)."""
    import torch.distributed as dist
    if not sb.utils.distributed.if_main_process():
        return
    fused = getattr(brain, "_fused", None)
    inited = dist.is_available() and dist.is_initialized()
    save({"fused_hip_step": bool(fused), "world_size": dist.get_world_size() if inited else 1, "backend": dist.get_backend() if inited else None,
          "optimizer_steps": int(getattr(brain, "step", 0)), "gradient_allreduces": int(fused.collectives) if fused else 0,
          "nonfinite_steps": int(getattr(brain, "nonfinite_count", 0)), "device": str(brain.device)},
         os.path.join(hparams["output_folder"], "run_summary.json"))
