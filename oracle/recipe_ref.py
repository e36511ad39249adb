"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Never imported by the product package.

CPU restatement of the recipe-side arithmetic of the training step around the model:
  * the 4-term NLL objective          -- reference pretrain.py:56-88 (finetune.py:55-79)
  * gradient check + norm clipping    -- SpeechBrain 0.5.15 ``Brain.check_gradients`` (third-party,
    not under /root/reference; pinned version environment.yaml:105): finite check of the loss, then
    ``clip_grad_norm_(parameters, max_grad_norm=5.0)``; restated from its published behaviour and
    anchored on the call site pretrain.py:126
  * Adadelta(lr=1, rho=0.95, eps=1e-8) -- reference hparams/pretrain.yaml:44-47 -> torch.optim.Adadelta
  * the target padding contract        -- reference datasets/syn.py:46-74

Parity status: PINNED for the loss (fixtures from the reference model + torch.nn.NLLLoss run in the
build container) and for clip+Adadelta (fixtures from torch.nn.utils.clip_grad_norm_ and
torch.optim.Adadelta, the very objects the reference instantiates).  See tests/golden/make_golden.py.
"""
import torch

PAD = 147
EOS = 146


def nll_mean(logp, target, ignore_index=None):
    """torch.nn.NLLLoss(reduction='mean'[, ignore_index]) on (..., C) log-probs / (...) targets."""
    C = logp.shape[-1]
    lp = logp.reshape(-1, C)
    tg = target.reshape(-1)
    if ignore_index is None:
        keep = torch.ones_like(tg, dtype=torch.bool)
    else:
        keep = tg != ignore_index
    picked = lp.gather(1, tg.clamp(0, C - 1).unsqueeze(1)).squeeze(1)
    return -(picked * keep.to(lp.dtype)).sum() / keep.sum().to(lp.dtype)


def objectives(predictions, targets):
    """reference ASR.compute_objectives: returns (total, time, key, upper, lower) losses.

    predictions = (ts (B,5,7), key (B,5,14), upper (B,5,U,V), lower (B,5,L,V)) log-probs;
    targets = (ts (B,5), key (B,5), upper (B,5,U), lower (B,5,L)) int64.
    The score terms ignore <pad>=147 (pretrain.yaml:53-54) and average over non-pad targets.
    """
    ts_o, key_o, up_o, lo_o = predictions
    ts_t, key_t, up_t, lo_t = targets
    time_loss = nll_mean(ts_o, ts_t)
    key_loss = nll_mean(key_o, key_t)
    upper_loss = nll_mean(up_o, up_t, ignore_index=PAD)
    lower_loss = nll_mean(lo_o, lo_t, ignore_index=PAD)
    return time_loss + key_loss + upper_loss + lower_loss, time_loss, key_loss, upper_loss, lower_loss


def clip_grad_norm(grads, max_norm=5.0):
    """torch.nn.utils.clip_grad_norm_ semantics (L2): returns (total_norm, clipped grads)."""
    total = torch.sqrt(sum((g.detach().double() ** 2).sum() for g in grads)).float()
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    return total, [g * coef for g in grads]


def adadelta_step(p, g, square_avg, acc_delta, lr=1.0, rho=0.95, eps=1e-8):
    """torch.optim.Adadelta single-tensor update (weight_decay=0); tensors updated in place."""
    square_avg.mul_(rho).addcmul_(g, g, value=1 - rho)
    std = square_avg.add(eps).sqrt_()
    delta = acc_delta.add(eps).sqrt_().div_(std).mul_(g)
    acc_delta.mul_(rho).addcmul_(delta, delta, value=1 - rho)
    p.add_(delta, alpha=-lr)


def train_step(params, grads, state, loss_value, max_grad_norm=5.0, lr=1.0, rho=0.95, eps=1e-8):
    """check_gradients + optimizer.step of reference ASR.fit_batch (pretrain.py:125-128).

    Non-finite loss: the step is skipped (returns False).  ``state`` maps name -> (square_avg, acc_delta).
    """
    if not bool(torch.isfinite(torch.as_tensor(loss_value))):
        return False
    names = list(params.keys())
    _, clipped = clip_grad_norm([grads[n] for n in names], max_grad_norm)
    for n, g in zip(names, clipped):
        if n not in state:
            state[n] = (torch.zeros_like(params[n]), torch.zeros_like(params[n]))
        adadelta_step(params[n], g, state[n][0], state[n][1], lr, rho, eps)
    return True


def pad_measure(measure, max_length):
    """reference SyntheticDataset.pad_single_measure, datasets/syn.py:67-74."""
    out = torch.full((max_length,), PAD, dtype=torch.long)
    measure = list(measure)[:max_length]
    if measure:
        out[:len(measure)] = torch.tensor(measure, dtype=torch.long)
    if len(measure) < max_length:
        out[len(measure)] = EOS
    return out


def pad_score(score, max_length):
    """reference SyntheticDataset.pad_score, datasets/syn.py:60-65: -> ((bars,max_length) i64, (bars,) i64)."""
    padded = torch.stack([pad_measure(m, max_length) for m in score])
    lengths = torch.tensor([min(len(m), max_length) for m in score], dtype=torch.long)
    return padded, lengths


def pad_spectrogram(spec, max_frames):
    """reference SyntheticDataset.pad_spectrogram, datasets/syn.py:46-58: (T,F) -> (1,max_frames,F) f32."""
    spec = torch.as_tensor(spec).float()
    out = torch.zeros((max_frames, spec.shape[-1]))
    n = min(spec.shape[0], max_frames)
    out[:n] = spec[:n]
    return out.unsqueeze(0)
