"""The slice of SpeechBrain 0.5.15 that the piano-a2s recipes touch, for environments without SpeechBrain.

The reference recipes subclass ``sb.Brain`` and call a handful of helpers (reference pretrain.py:12-14,31,253-305; hook list in
SURVEY.md 8b).  SpeechBrain is a third-party dependency (environment.yaml:105) that is not under /root/reference and not
installed here, so this is a behavioural restatement FROM ITS PUBLISHED INTERFACE, anchored on the call sites above; where the
recipe's own files define the behaviour (hooks, hparams keys) those are followed exactly.  When SpeechBrain is importable the
recipes use it instead (see pretrain.py).
"""
import argparse
import datetime
import enum
import os
import shutil
import sys
import time
import types

import torch
import torch.distributed as dist
import yaml


class Stage(enum.Enum):
    TRAIN = 1
    VALID = 2
    TEST = 3


def if_main_process():
    return int(os.environ.get("RANK", "0")) == 0


def run_on_main(func, args=None, kwargs=None):
    if if_main_process():
        func(*(args or ()), **(kwargs or {}))
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def parse_arguments(arg_list):
    """-> (hparams_file, run_opts, overrides).  Unknown ``--key=value`` / ``--key value`` pairs become YAML overrides."""
    ap = argparse.ArgumentParser()
    ap.add_argument("param_file")
    ap.add_argument("--device", default=None)
    ap.add_argument("--distributed_launch", action="store_true")
    ap.add_argument("--distributed_backend", default="nccl")
    ap.add_argument("--max_grad_norm", type=float, default=5.0)
    ap.add_argument("--nonfinite_patience", type=int, default=3)
    ap.add_argument("--debug", action="store_true")
    ap.add_argument("--debug_batches", type=int, default=2)
    ap.add_argument("--debug_epochs", type=int, default=2)
    known, rest = ap.parse_known_args(arg_list)
    overrides = []        # kept as YAML TEXT: values may carry hyperpyyaml tags (!new:, !ref ...) that only the hparams loader knows
    i = 0
    while i < len(rest):
        tok = rest[i]
        if not tok.startswith("--"):
            raise ValueError(f"cannot parse override '{tok}'")
        if "=" in tok:
            k, v = tok[2:].split("=", 1)
            i += 1
        else:
            k, v = tok[2:], rest[i + 1]
            i += 2
        overrides.append(f"{k}: {v}")
    run_opts = {k: v for k, v in vars(known).items() if k != "param_file"}
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if run_opts["device"] is None:
        run_opts["device"] = f"cuda:{local_rank}" if torch.cuda.is_available() else "cpu"
    elif run_opts["device"] == "cuda" and "LOCAL_RANK" in os.environ:
        run_opts["device"] = f"cuda:{local_rank}"
    return known.param_file, run_opts, "\n".join(overrides)


def _set_cuda_device(device):
    """torch.cuda.set_device needs an index: a bare 'cuda' (single-process `--device cuda`, or a Brain built without run_opts) means the
    current device."""
    d = torch.device(device)
    torch.cuda.set_device(d.index if d.index is not None else torch.cuda.current_device())


def ddp_init_group(run_opts):
    """One process per GPU: initialise the process group whenever the torchrun environment is present (the README launches
    with torchrun but without --distributed_launch)."""
    if "RANK" in os.environ and "WORLD_SIZE" in os.environ and not dist.is_initialized():      # torchrun's environment, any world size
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = run_opts.get("distributed_backend", "nccl")
        if not str(run_opts.get("device", "cpu")).startswith("cuda"):
            backend = "gloo"
        if str(run_opts.get("device", "")).startswith("cuda"):
            _set_cuda_device(run_opts["device"])
        dist.init_process_group(backend=backend)


def create_experiment_directory(experiment_directory, hyperparams_to_save=None, overrides=None, **_):
    if if_main_process():
        os.makedirs(experiment_directory, exist_ok=True)
        if hyperparams_to_save is not None:
            shutil.copy(hyperparams_to_save, os.path.join(experiment_directory, "hyperparams.yaml"))
            if overrides:
                with open(os.path.join(experiment_directory, "hyperparams.yaml"), "a") as f:
                    f.write("\n# overrides\n" + (overrides if isinstance(overrides, str) else yaml.safe_dump(overrides)))
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


# ----------------------------------------------------------------------------- small recoverable objects
class EpochCounter:
    def __init__(self, limit):
        self.current, self.limit = 0, int(limit)

    def __iter__(self):
        return self

    def __next__(self):
        if self.current < self.limit:
            self.current += 1
            return self.current
        raise StopIteration

    # SpeechBrain's EpochCounter checkpoints itself as a PLAIN-TEXT integer (its @mark_as_saver / @mark_as_loader hooks), not as a
    # pickled dict: `_save` / `_recover` below are the hooks Checkpointer looks for, so that a save/ directory written by SpeechBrain
    # (the reference's pretrain run, the README's published checkpoints) recovers here and vice versa.
    def _save(self, path):
        with open(path, "w") as f:
            f.write(str(self.current))

    def _recover(self, path, end_of_epoch=True, device=None):
        try:
            with open(path) as f:
                value = int(f.read().strip())
        except (UnicodeDecodeError, ValueError):          # a counter.ckpt written by round 1 of this repository: torch.save({"current": n})
            value = int(torch.load(path, map_location="cpu")["current"])
        self.current = value if end_of_epoch else value - 1

    def state_dict(self):
        return {"current": self.current}

    def load_state_dict(self, sd):
        self.current = int(sd["current"])


class InputNormalization:
    """Declared and checkpointed by the recipe but never applied (pretrain.yaml:80-81)."""

    def __init__(self, norm_type="global", **_):
        self.norm_type, self.count = norm_type, 0

    def state_dict(self):
        return {"norm_type": self.norm_type, "count": self.count}

    def load_state_dict(self, sd):
        self.count = sd.get("count", 0)


class NewBobScheduler:
    """lr <- lr * annealing_factor when the relative improvement of the metric is below the threshold (patient epochs)."""

    def __init__(self, initial_value, annealing_factor=0.5, improvement_threshold=0.0025, patient=0):
        self.hyperparam_value = initial_value
        self.annealing_factor, self.improvement_threshold, self.patient = annealing_factor, improvement_threshold, patient
        self.metric_values, self.current_patient = [], patient

    def __call__(self, metric_value):
        old = new = self.hyperparam_value
        if self.metric_values:
            prev = self.metric_values[-1]
            improvement = 0.0 if prev == 0 else (prev - metric_value) / prev
            if improvement < self.improvement_threshold:
                if self.current_patient == 0:
                    new = old * self.annealing_factor
                    self.current_patient = self.patient
                else:
                    self.current_patient -= 1
        self.metric_values.append(metric_value)
        self.hyperparam_value = new
        return old, new

    def state_dict(self):
        return {"hyperparam_value": self.hyperparam_value, "metric_values": self.metric_values, "current_patient": self.current_patient}

    def load_state_dict(self, sd):
        self.hyperparam_value, self.metric_values, self.current_patient = sd["hyperparam_value"], sd["metric_values"], sd["current_patient"]


def update_learning_rate(optimizer, new_lr, param_group=None):
    if hasattr(optimizer, "param_groups"):
        for g in optimizer.param_groups:
            g["lr"] = new_lr
    if hasattr(optimizer, "lr"):
        optimizer.lr = new_lr


class FileTrainLogger:
    def __init__(self, save_file, **_):
        self.save_file = save_file

    @staticmethod
    def _fmt(stats, prefix):
        out = []
        for k, v in (stats or {}).items():
            v = float(v) if hasattr(v, "__float__") else v
            out.append(f"{prefix}{k}: {v:.4g}" if isinstance(v, float) else f"{prefix}{k}: {v}")
        return ", ".join(out)

    def log_stats(self, stats_meta, train_stats=None, valid_stats=None, test_stats=None, verbose=False):
        parts = [self._fmt(stats_meta, ""), self._fmt(train_stats, "train "), self._fmt(valid_stats, "valid "), self._fmt(test_stats, "test ")]
        line = " - ".join(p for p in parts if p)
        if if_main_process():
            os.makedirs(os.path.dirname(os.path.abspath(self.save_file)), exist_ok=True)
            with open(self.save_file, "a") as f:
                f.write(line + "\n")
        if verbose:
            print(line)


class Checkpointer:
    """save/<CKPT+timestamp>/{<name>.ckpt ..., CKPT.yaml} -- the directory layout finetune.py copies and edits
    (reference finetune.py:251-258) and the README's published checkpoints use."""

    def __init__(self, checkpoints_dir, recoverables=None, **_):
        self.checkpoints_dir = checkpoints_dir
        self.recoverables = dict(recoverables or {})

    def add_recoverable(self, name, obj):
        self.recoverables[name] = obj

    def _list(self):
        if not os.path.isdir(self.checkpoints_dir):
            return []
        out = []
        for d in sorted(os.listdir(self.checkpoints_dir)):
            meta_f = os.path.join(self.checkpoints_dir, d, "CKPT.yaml")
            if d.startswith("CKPT") and os.path.exists(meta_f):
                with open(meta_f) as f:
                    out.append((os.path.join(self.checkpoints_dir, d), yaml.safe_load(f) or {}))
        return out

    def save_checkpoint(self, meta=None, name=None):
        meta = dict(meta or {})
        meta.setdefault("unixtime", time.time())
        meta.setdefault("end-of-epoch", True)
        stamp = datetime.datetime.fromtimestamp(meta["unixtime"]).strftime("%Y-%m-%d+%H-%M-%S") + "+00"
        path = os.path.join(self.checkpoints_dir, name or f"CKPT+{stamp}")
        if if_main_process():
            os.makedirs(path, exist_ok=True)
            for n, obj in self.recoverables.items():
                target = os.path.join(path, f"{n}.ckpt")
                if hasattr(obj, "_save"):                 # the object's own saver hook (SpeechBrain: @mark_as_saver)
                    obj._save(target)
                else:
                    torch.save(obj.state_dict() if hasattr(obj, "state_dict") else obj, target)
            with open(os.path.join(path, "CKPT.yaml"), "w") as f:
                yaml.safe_dump({k: (float(v) if hasattr(v, "__float__") and not isinstance(v, bool) else v) for k, v in meta.items()}, f)
        return path

    def save_and_keep_only(self, meta=None, min_keys=(), max_keys=(), num_to_keep=1, **_):
        self.save_checkpoint(meta)
        if not if_main_process():
            return
        ck = self._list()
        keep = set()
        for key in min_keys:
            c = [x for x in ck if key in x[1]]
            keep.update(p for p, _ in sorted(c, key=lambda x: x[1][key])[:num_to_keep])
        for key in max_keys:
            c = [x for x in ck if key in x[1]]
            keep.update(p for p, _ in sorted(c, key=lambda x: -x[1][key])[:num_to_keep])
        if not min_keys and not max_keys:
            keep.update(p for p, _ in sorted(ck, key=lambda x: -x[1].get("unixtime", 0))[:num_to_keep])
        for p, _ in ck:
            if p not in keep:
                shutil.rmtree(p, ignore_errors=True)

    def find_checkpoint(self, min_key=None, max_key=None):
        ck = self._list()
        if not ck:
            return None
        if min_key is not None:
            c = [x for x in ck if min_key in x[1]]
            return min(c, key=lambda x: x[1][min_key]) if c else None
        if max_key is not None:
            c = [x for x in ck if max_key in x[1]]
            return max(c, key=lambda x: x[1][max_key]) if c else None
        return max(ck, key=lambda x: x[1].get("unixtime", 0))

    def recover_if_possible(self, min_key=None, max_key=None, device=None, **_):
        found = self.find_checkpoint(min_key, max_key)
        if found is None:
            return None
        path, meta = found
        for n, obj in self.recoverables.items():
            f = os.path.join(path, f"{n}.ckpt")
            if not os.path.exists(f):
                continue
            if hasattr(obj, "_recover"):                  # the object's own loader hook (SpeechBrain: @mark_as_loader)
                obj._recover(f, end_of_epoch=bool(meta.get("end-of-epoch", True)), device=device)
            elif hasattr(obj, "load_state_dict"):
                obj.load_state_dict(torch.load(f, map_location=device or "cpu"))
        return path, meta


# ----------------------------------------------------------------------------- Brain
class Brain:
    """Training-loop skeleton with the hook names the recipe overrides (compute_forward, compute_objectives, fit_batch,
    evaluate_batch, on_stage_start, on_stage_end) and the attributes it reads (modules, hparams, device, optimizer, checkpointer)."""

    def __init__(self, modules=None, opt_class=None, hparams=None, run_opts=None, checkpointer=None):
        run_opts = dict(run_opts or {})
        self.device = run_opts.get("device", "cuda" if torch.cuda.is_available() else "cpu")
        if str(self.device).startswith("cuda") and torch.cuda.is_available():
            # liba2s_hip.so launches on torch's CURRENT stream of the CURRENT device: make the run's device current (SpeechBrain's
            # Brain.__init__ does the same), or `--device cuda:1` would launch on a device-0 stream with device-1 pointers
            _set_cuda_device(self.device)
        self.max_grad_norm = float(run_opts.get("max_grad_norm", 5.0))
        self.nonfinite_patience = int(run_opts.get("nonfinite_patience", 3))
        self.debug, self.debug_batches, self.debug_epochs = bool(run_opts.get("debug")), int(run_opts.get("debug_batches", 2)), int(run_opts.get("debug_epochs", 2))
        self.modules = torch.nn.ModuleDict(modules or {}).to(self.device)
        self.opt_class = opt_class
        self.hparams = types.SimpleNamespace(**(hparams or {}))
        self.checkpointer = checkpointer
        self.nonfinite_count = 0
        self.step = 0
        self.avg_train_loss = 0.0
        self.optimizer = None
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1

    # --- hooks the recipe overrides
    def compute_forward(self, batch, stage):
        raise NotImplementedError

    def compute_objectives(self, predictions, batch, stage):
        raise NotImplementedError

    def on_stage_start(self, stage, epoch=None):
        pass

    def on_stage_end(self, stage, stage_loss, epoch=None):
        pass

    def on_fit_start(self):
        self.init_optimizers()
        if self.checkpointer is not None:
            self.checkpointer.recover_if_possible(device=self.device)
        self.sync_replicas()

    def sync_replicas(self):
        """What DDP's constructor does for the reference (SpeechBrain wraps modules in DDP in on_fit_start): every replica starts
        from rank 0's parameters and buffers, whatever each rank's own initialisation or checkpoint recovery produced."""
        if self.world > 1:
            with torch.no_grad():
                for t in list(self.modules.parameters()) + list(self.modules.buffers()):
                    dist.broadcast(t.data, src=0)

    def init_optimizers(self):
        if self.opt_class is not None and self.optimizer is None:
            self.optimizer = self.opt_class(self.modules.parameters())
            if self.checkpointer is not None and hasattr(self.optimizer, "state_dict"):
                self.checkpointer.add_recoverable("optimizer", self.optimizer)

    def check_gradients(self, loss):
        """Non-finite loss: skip the update (abort after `nonfinite_patience` in a row-free count); else clip the global L2
        norm of all module parameters' gradients to max_grad_norm."""
        if not torch.isfinite(loss):
            self.nonfinite_count += 1
            if self.nonfinite_count > self.nonfinite_patience:
                raise ValueError("Loss is not finite and patience is exhausted.")
            return False
        torch.nn.utils.clip_grad_norm_((p for p in self.modules.parameters()), self.max_grad_norm)
        return True

    def fit_batch(self, batch):
        outputs = self.compute_forward(batch, Stage.TRAIN)
        loss = self.compute_objectives(outputs, batch, Stage.TRAIN)
        loss.backward()
        if self.check_gradients(loss):
            self.optimizer.step()
        self.optimizer.zero_grad()
        return loss.detach()

    def evaluate_batch(self, batch, stage):
        out = self.compute_forward(batch, stage=stage)
        loss = self.compute_objectives(out, batch, stage=stage)
        return loss.detach()

    @staticmethod
    def update_average(loss, avg, step):
        loss = float(loss)
        return avg if not torch.isfinite(torch.tensor(loss)) else avg + (loss - avg) / step

    def make_dataloader(self, dataset, stage, **loader_kwargs):
        if isinstance(dataset, torch.utils.data.DataLoader):
            return dataset
        sampler = None
        if stage == Stage.TRAIN and dist.is_available() and dist.is_initialized():
            sampler = torch.utils.data.distributed.DistributedSampler(dataset, shuffle=loader_kwargs.pop("shuffle", False))
            loader_kwargs["shuffle"] = False
        if str(self.device).startswith("cuda"):
            # batches are collated into PINNED host buffers, so the trainer's one `.to(device, non_blocking=True)` per batch tensor is a
            # single asynchronous DMA (the reference moves every ITEM to the device inside Dataset.__getitem__, syn.py:113)
            loader_kwargs.setdefault("pin_memory", True)
        return torch.utils.data.DataLoader(dataset, sampler=sampler, **loader_kwargs)

    def fit(self, epoch_counter, train_set, valid_set=None, train_loader_kwargs=None, valid_loader_kwargs=None, progressbar=None):
        train_loader = self.make_dataloader(train_set, Stage.TRAIN, **(train_loader_kwargs or {}))
        valid_loader = self.make_dataloader(valid_set, Stage.VALID, **(valid_loader_kwargs or {})) if valid_set is not None else None
        self.on_fit_start()
        for epoch in epoch_counter:
            self.on_stage_start(Stage.TRAIN, epoch)
            self.modules.train()
            if hasattr(train_loader.sampler, "set_epoch"):
                train_loader.sampler.set_epoch(epoch)
            self.avg_train_loss, n = 0.0, 0
            for batch in train_loader:
                self.step += 1
                n += 1
                loss = self.fit_batch(batch)
                self.avg_train_loss = self.update_average(loss, self.avg_train_loss, n)
                if self.debug and n >= self.debug_batches:
                    break
            self.on_stage_end(Stage.TRAIN, self.avg_train_loss, epoch)
            if valid_loader is not None:
                self._eval_loop(valid_loader, Stage.VALID, epoch)
            if self.debug and epoch >= self.debug_epochs:
                break

    def _eval_loop(self, loader, stage, epoch):
        self.on_stage_start(stage, epoch)
        self.modules.eval()
        avg, n = 0.0, 0
        for batch in loader:
            n += 1
            avg = self.update_average(self.evaluate_batch(batch, stage=stage), avg, n)
            if self.debug and n >= self.debug_batches:
                break
        self.on_stage_end(stage, avg, epoch)
        return avg

    def evaluate(self, test_set, max_key=None, min_key=None, progressbar=None, test_loader_kwargs=None):
        loader = self.make_dataloader(test_set, Stage.TEST, **(test_loader_kwargs or {}))
        if self.checkpointer is not None:
            self.checkpointer.recover_if_possible(min_key=min_key, max_key=max_key, device=self.device)
        return self._eval_loop(loader, Stage.TEST, None)


# namespaces so that `sb.nnet.schedulers.update_learning_rate`, `sb.utils.distributed.ddp_init_group` resolve
nnet = types.SimpleNamespace(schedulers=types.SimpleNamespace(update_learning_rate=update_learning_rate, NewBobScheduler=NewBobScheduler))
utils = types.SimpleNamespace(distributed=types.SimpleNamespace(ddp_init_group=ddp_init_group, run_on_main=run_on_main, if_main_process=if_main_process),
                              checkpoints=types.SimpleNamespace(Checkpointer=Checkpointer),
                              epoch_loop=types.SimpleNamespace(EpochCounter=EpochCounter),
                              train_logger=types.SimpleNamespace(FileTrainLogger=FileTrainLogger))
