"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Never imported by the product package.

CPU restatement (PyTorch fp32, explicit arithmetic) of the reference's transcription model,
``/root/reference/models.py``, *as written*: same operation sequence, same per-step
``repeat``+``cat``+``Linear(4H->H)`` attention, same Python-``random`` draw protocol, same
host-side EOS bookkeeping.  Nothing here is hoisted, fused or re-associated: this file is
what the HIP path is compared against and what ``bench.py`` times as ``cpu_baseline``.

Parity status: PINNED.  The reference is a Python module that imports in the build
container; ``tests/golden/make_golden.py`` runs it (read-only, never copied) and commits its
inputs/outputs as fixtures; ``tests/test_oracle_golden.py`` checks this restatement against
those fixtures (outputs, loss terms, gradients, running statistics, RNG draw counts).

Parameters are passed as a flat dict keyed by the reference's ``state_dict`` names
(e.g. ``convstack.conv1.weight``, ``decoder.upper_decoder.gru.weight_ih_l0``).
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg may import this.
"""
import random as _py_random

import torch
import torch.nn.functional as F

# vocabulary constants of LabelsMultiple(extended=True) -- reference models.py:9-12
VOCAB_SIZE = 173
SOS = 145
EOS = 146
PAD = 147


# ----------------------------------------------------------------------------- primitives
def _dropout(x, p, training, enabled):
    """reference: F.dropout(x, p, training=self.training) (models.py:239,391,541)."""
    if enabled and training and p > 0.0:
        return F.dropout(x, p=p, training=True)
    return x


def batch_norm(x, P, B, name, training, channel_dim=1, eps=1e-5, momentum=0.1):
    """nn.BatchNorm{1,2}d semantics (reference models.py:499-505).

    training: normalise with the batch mean and *biased* variance over every dim except
    ``channel_dim``; update running_mean / running_var (*unbiased* variance, momentum 0.1)
    and num_batches_tracked in ``B`` in place.  eval: normalise with the running statistics.
    """
    w, b = P[name + ".weight"], P[name + ".bias"]
    dims = [d for d in range(x.dim()) if d != channel_dim]
    shape = [1] * x.dim()
    shape[channel_dim] = -1
    if training:
        n = x.numel() // x.shape[channel_dim]
        mean = x.mean(dim=dims)
        var = ((x - mean.view(shape)) ** 2).mean(dim=dims)
        with torch.no_grad():
            B[name + ".running_mean"].mul_(1 - momentum).add_(momentum * mean.detach())
            B[name + ".running_var"].mul_(1 - momentum).add_(momentum * var.detach() * n / max(n - 1, 1))
            B[name + ".num_batches_tracked"].add_(1)
    else:
        mean, var = B[name + ".running_mean"], B[name + ".running_var"]
    xhat = (x - mean.view(shape)) / torch.sqrt(var.view(shape) + eps)
    return xhat * w.view(shape) + b.view(shape)


def gru_cell(x, h, w_ih, w_hh, b_ih, b_hh):
    """One PyTorch GRU step, gate packing [r; z; n] (torch.nn.GRU docs; reference uses nn.GRU).

    r = s(W_ir x + b_ir + W_hr h + b_hr); z likewise; n = tanh(W_in x + b_in + r*(W_hn h + b_hn));
    h' = (1 - z) * n + z * h.
    """
    gi = x @ w_ih.t() + b_ih
    gh = h @ w_hh.t() + b_hh
    H = h.shape[-1]
    r = torch.sigmoid(gi[..., :H] + gh[..., :H])
    z = torch.sigmoid(gi[..., H:2 * H] + gh[..., H:2 * H])
    n = torch.tanh(gi[..., 2 * H:] + r * gh[..., 2 * H:])
    return (1 - z) * n + z * h


def gru_direction(x, P, prefix, suffix, lengths=None, reverse=False, h0=None):
    """Run one direction of one nn.GRU layer over (B, T, I).  Returns (outputs (B,T,H), h_n (B,H)).

    With ``lengths`` (B,) the semantics are those of pack_padded_sequence(enforce_sorted=False):
    item b only sees its first lengths[b] steps; the reverse direction starts at step
    lengths[b]-1; h_n is the state after the item's own last step.
    """
    w_ih, w_hh = P[f"{prefix}.weight_ih_{suffix}"], P[f"{prefix}.weight_hh_{suffix}"]
    b_ih, b_hh = P[f"{prefix}.bias_ih_{suffix}"], P[f"{prefix}.bias_hh_{suffix}"]
    Bn, T, _ = x.shape
    H = w_hh.shape[1]
    h = x.new_zeros(Bn, H) if h0 is None else h0
    outs = [None] * T
    steps = range(T - 1, -1, -1) if reverse else range(T)
    for t in steps:
        hn = gru_cell(x[:, t], h, w_ih, w_hh, b_ih, b_hh)
        if lengths is not None:
            live = (lengths > t).to(x.dtype).unsqueeze(1)
            hn = live * hn + (1 - live) * h
        h = hn
        outs[t] = h
    return torch.stack(outs, dim=1), h


# ----------------------------------------------------------------------------- model parts
def convstack_forward(x, P, B, training, dropout=True):
    """reference ConvStack.forward, models.py:523-543.  x: (B,1,T,F) -> (B,T,C)."""
    for i in (1, 2, 3, 4):
        x = F.conv2d(x, P[f"convstack.conv{i}.weight"], None, stride=1, padding=1)
        x = torch.relu(batch_norm(x, P, B, f"convstack.bn{i}", training))
    x = x.transpose(1, 2).flatten(2)                         # (B, T, 40*F), index = c*F + f
    x = x @ P["convstack.out.weight"].t()                    # Linear, no bias
    x = batch_norm(x.transpose(1, 2), P, B, "convstack.out_bn", training).transpose(1, 2)
    x = torch.relu(x)
    return _dropout(x, 0.2, training, dropout)


def encoder_forward(x, P):
    """reference Encoder.forward, models.py:75-82: 2-layer bi-GRU + shared fc/tanh bridge."""
    f0, hf0 = gru_direction(x, P, "encoder.gru", "l0")
    r0, hr0 = gru_direction(x, P, "encoder.gru", "l0_reverse", reverse=True)
    x1 = torch.cat([f0, r0], dim=2)
    f1, hf1 = gru_direction(x1, P, "encoder.gru", "l1")
    r1, hr1 = gru_direction(x1, P, "encoder.gru", "l1_reverse", reverse=True)
    out = torch.cat([f1, r1], dim=2)
    fc_w, fc_b = P["encoder.fc.weight"], P["encoder.fc.bias"]
    h1 = torch.tanh(torch.cat([hf0, hr0], dim=1) @ fc_w.t() + fc_b)
    h2 = torch.tanh(torch.cat([hf1, hr1], dim=1) @ fc_w.t() + fc_b)
    return out, torch.cat([h1, h2], dim=1).unsqueeze(0)      # (B,T,2H), (1,B,2H)


def attention(hidden, enc, P, prefix):
    """reference AttentionLayer.forward, models.py:452-461 -- as written (no key hoisting)."""
    T = enc.shape[1]
    hrep = hidden.transpose(0, 1).repeat(1, T, 1)            # (B,T,2H)
    cat = torch.cat((hrep, enc), dim=2)                      # (B,T,4H)
    energy = torch.tanh(cat @ P[prefix + ".attn.weight"].t() + P[prefix + ".attn.bias"])
    score = (energy @ P[prefix + ".v.weight"].t()).squeeze(2)
    return F.softmax(score, dim=1)                           # (B,T)


def _staff_token(ids, lengths, P):
    """reference get_staff_token_from_{gt,probs}, models.py:164-189: packed bi-GRU final states."""
    emb = F.embedding(ids, P["decoder.note_emb.weight"])
    lengths = lengths.to(torch.long).cpu()
    if int(lengths.min()) <= 0:
        raise RuntimeError("Length of all samples has to be greater than 0")   # pack_padded_sequence
    Tmax = int(lengths.max())
    emb = emb[:, :Tmax]
    _, hf = gru_direction(emb, P, "decoder.staff_emb", "l0", lengths=lengths)
    _, hr = gru_direction(emb, P, "decoder.staff_emb", "l0_reverse", lengths=lengths, reverse=True)
    return torch.cat([hf, hr], dim=1).unsqueeze(1)           # (B,1,2*staff_emb)


def decode_notes(enc, hidden, P, prefix, max_steps, inference, gt, tf_ratio, training, rng, dropout):
    """reference NoteDecoder.decode_notes, models.py:366-420 (prefix = decoder.{upper,lower}_decoder)."""
    if inference:
        assert tf_ratio == 0 and gt is None
    Bn = enc.shape[0]
    emb_w = P[prefix + ".embedding.weight"]
    token = F.embedding(torch.full((Bn, 1), SOS, dtype=torch.long), emb_w)
    probs = [None] * max_steps
    eos_seen = torch.zeros(Bn)
    lengths = torch.full((Bn,), max_steps, dtype=torch.long)
    for t in range(max_steps):
        if eos_seen.sum() == Bn:
            break
        token = _dropout(token, 0.1, training, dropout)
        a = attention(hidden, enc, P, prefix + ".attn").unsqueeze(1)
        context = torch.bmm(a, enc)
        x = torch.cat([token, context], dim=2)
        h = gru_cell(x[:, 0], hidden[0], P[prefix + ".gru.weight_ih_l0"], P[prefix + ".gru.weight_hh_l0"],
                     P[prefix + ".gru.bias_ih_l0"], P[prefix + ".gru.bias_hh_l0"])
        hidden = h.unsqueeze(0)
        out = torch.cat([h.unsqueeze(1), context], dim=-1)
        logits = out @ P[prefix + ".out.weight"].t() + P[prefix + ".out.bias"]
        prob = F.log_softmax(logits, dim=-1)
        probs[t] = prob.squeeze(1)
        teacher_force = rng.random() < tf_ratio               # drawn on every executed step
        if (not inference) and teacher_force:
            token = F.embedding(gt[:, t].unsqueeze(1), emb_w)
        else:
            token = F.embedding(torch.argmax(prob, dim=-1), emb_w)
        am = torch.argmax(prob, dim=-1)[:, 0]
        for b in range(Bn):
            hit = (gt[b, t] == EOS) if gt is not None else (am[b] == EOS)
            if bool(hit):
                eos_seen[b] = 1
                lengths[b] = t + 1
    zero = enc.new_zeros(Bn, VOCAB_SIZE)
    score = torch.stack([p if p is not None else zero for p in probs], dim=1)
    return score, lengths


def decoder_forward(enc, hidden, P, cfg, inference, ground_truth, tf_ratio, training, rng, dropout):
    """reference HierarchicalDecoder.decode_bars, models.py:191-316."""
    if inference:
        assert tf_ratio == 0 and ground_truth is None
    Bn = enc.shape[0]
    if ground_truth is not None:
        ts_gt, key_gt, up_gt, up_len_gt, lo_gt, lo_len_gt = ground_truth
    sos_eos = torch.tensor([[SOS, EOS]], dtype=torch.long).repeat(Bn, 1)
    staff0 = _staff_token(sos_eos, torch.full((Bn,), 2), P)
    ts_tok = F.embedding(torch.full((Bn, 1), cfg["num_time_sig"], dtype=torch.long), P["decoder.time_sig_emb.weight"])
    key_tok = F.embedding(torch.full((Bn, 1), cfg["num_keys"], dtype=torch.long), P["decoder.key_emb.weight"])
    token = torch.cat([staff0, staff0, ts_tok, key_tok], dim=-1)

    def head(x, name):
        for i in (0, 2, 4):
            x = x @ P[f"decoder.{name}.{i}.weight"].t() + P[f"decoder.{name}.{i}.bias"]
            if i != 4:
                x = torch.relu(x)
        return F.log_softmax(x, dim=-1)

    ts_outs, key_outs, up_outs, lo_outs = [], [], [], []
    U, L = cfg["max_length"]
    for bar in range(cfg["max_bars"]):
        token = _dropout(token, 0.1, training, dropout)
        a = attention(hidden, enc, P, "decoder.attn").unsqueeze(1)
        context = torch.bmm(a, enc)
        x = torch.cat([token, context], dim=2)
        h = gru_cell(x[:, 0], hidden[0], P["decoder.gru.weight_ih_l0"], P["decoder.gru.weight_hh_l0"],
                     P["decoder.gru.bias_ih_l0"], P["decoder.gru.bias_hh_l0"])
        hidden = h.unsqueeze(0)
        bar_summary = h.unsqueeze(1)
        gt_u = up_gt[:, bar, :] if ground_truth is not None else None
        gt_l = lo_gt[:, bar, :] if ground_truth is not None else None
        tf = tf_ratio if ground_truth is not None else 0.0
        up_probs, up_len = decode_notes(enc, bar_summary.transpose(0, 1), P, "decoder.upper_decoder", U,
                                        inference, gt_u, tf, training, rng, dropout)
        lo_probs, lo_len = decode_notes(enc, bar_summary.transpose(0, 1), P, "decoder.lower_decoder", L,
                                        inference, gt_l, tf, training, rng, dropout)
        up_outs.append(up_probs)
        lo_outs.append(lo_probs)
        head_in = torch.cat([bar_summary.squeeze(1), context.squeeze(1)], dim=1)
        ts_lp = head(head_in, "time_sig_out")
        key_lp = head(head_in, "key_out")
        ts_outs.append(ts_lp)
        key_outs.append(key_lp)
        teacher_force = rng.random() < tf_ratio               # one draw per bar, after both staves
        if teacher_force and not inference:
            up_tok = _staff_token(up_gt[:, bar, :], up_len_gt[:, bar], P)
            lo_tok = _staff_token(lo_gt[:, bar, :], lo_len_gt[:, bar], P)
            ts_tok = F.embedding(ts_gt[:, bar], P["decoder.time_sig_emb.weight"]).unsqueeze(1)
            key_tok = F.embedding(key_gt[:, bar], P["decoder.key_emb.weight"]).unsqueeze(1)
        else:
            up_tok = _staff_token(torch.argmax(up_probs, dim=-1), up_len, P)
            lo_tok = _staff_token(torch.argmax(lo_probs, dim=-1), lo_len, P)
            ts_tok = F.embedding(torch.argmax(ts_lp, dim=-1), P["decoder.time_sig_emb.weight"]).unsqueeze(1)
            key_tok = F.embedding(torch.argmax(key_lp, dim=-1), P["decoder.key_emb.weight"]).unsqueeze(1)
        token = torch.cat([up_tok, lo_tok, ts_tok, key_tok], dim=-1)
    return (torch.stack(ts_outs, dim=1), torch.stack(key_outs, dim=1),
            torch.stack(up_outs, dim=1), torch.stack(lo_outs, dim=1))


def default_cfg(**kw):
    """Constructor defaults of hparams/pretrain.yaml (reference pretrain.yaml:16-19,70-75,84-96)."""
    cfg = dict(in_channels=1, freq_bins=480, conv_feature_size=256, hidden_size=256, max_bars=5,
               num_time_sig=7, num_keys=14, max_length=(398, 189), note_emb_size=16, staff_emb_size=32,
               time_sig_emb_size=5, key_emb_size=8)
    cfg.update(kw)
    return cfg


def forward(P, B, cfg, spectrogram, inference=True, ground_truth=None, teacher_forcing_ratio=0.0,
            training=False, rng=_py_random, dropout=True):
    """reference ScoreTranscription.forward, models.py:26-51.

    ``P`` parameters, ``B`` BatchNorm buffers (updated in place when ``training``), both keyed by
    state_dict names.  ``training`` mirrors ``module.training`` (BN batch statistics, dropout).
    Returns the four log-probability tensors (B,5,7) (B,5,14) (B,5,U,173) (B,5,L,173).
    """
    conv = convstack_forward(spectrogram, P, B, training, dropout)
    enc, hidden = encoder_forward(conv, P)
    return decoder_forward(enc, hidden, P, cfg, inference, ground_truth, teacher_forcing_ratio,
                           training, rng, dropout)
