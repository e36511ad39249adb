"""Data-parallel path on CPU: 2 processes, gloo backend, 127.0.0.1 rendezvous.

The multi-GPU design has exactly two collectives: broadcast of the flat parameter buffer at start and one
SUM all-reduce of the flat gradient buffer per step, divided by the world size.  Here each rank computes REAL gradients of the
transcription model for its own minibatch (oracle model at reduced size -- the HIP model cannot run without a GPU), then the
product functions piano_a2s_amd.train.broadcast_parameters / average_gradients run over gloo; the parent checks that
  * after the broadcast both ranks hold rank 0's parameters,
  * after the exchange both ranks hold the same gradient, equal to the mean of the two per-rank gradients computed without
    any collective (DDP semantics: each rank contributes the gradient of its own minibatch mean).
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

SMALL = dict(freq_bins=24, conv_feature_size=32, hidden_size=32, max_length=(12, 8))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rank_gradient(rank, seed_params):
    """Flat gradient of the oracle training objective on rank `rank`'s minibatch (2 clips)."""
    from oracle import model_ref, recipe_ref
    from piano_a2s_amd import spec, synthetic
    cfg = spec.default_cfg(**SMALL)
    P, B = spec.split_state(spec.procedural_state(cfg, seed_params, eos_bias=3.0, lively=True))
    P = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    batch = synthetic.make_batch(2, cfg, 100 + rank, frames=41, upper_range=(3, 10), lower_range=(2, 7), full_tail=0.0)
    gt = [batch[i] for i in range(1, 7)]
    outs = model_ref.forward(P, {k: v.clone() for k, v in B.items()}, cfg, batch[0], inference=False, ground_truth=gt,
                             teacher_forcing_ratio=1.0, training=True, dropout=False)
    recipe_ref.objectives(outs, (batch[1], batch[2], batch[3], batch[5]))[0].backward()
    return torch.cat([p.grad.reshape(-1) for p in P.values()])


def _worker(rank, world, port, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    from piano_a2s_amd import spec, train
    cfg = spec.default_cfg(**SMALL)
    # replicas start different, then rank 0's parameters are broadcast
    P, _ = spec.split_state(spec.procedural_state(cfg, 11 + rank))
    flat_p = torch.cat([v.reshape(-1) for v in P.values()])
    train.broadcast_parameters(flat_p, src=0)
    g_local = _rank_gradient(rank, 11)
    g = train.average_gradients(g_local.clone(), world)
    # the overlapped form the fused step uses: slices announced as the backward pass finishes them (later parameters first)
    ex = train.GradientExchange(world)
    g2 = g_local.clone()
    cut = g2.numel() // 3
    ex.slice_ready(g2, cut, g2.numel())
    ex.slice_ready(g2, 0, cut)
    g2 = ex.finish(g2)
    # the loss gate of the fused step: the loss is one more word behind the gradients (engine_bwd.Backward.flat_full) and travels inside the
    # FIRST slice's all-reduce, so every rank sees the same (mean) loss and takes the same skip-the-update decision with two collectives
    # per step, not three.  Rank 1 of the second exchange holds a non-finite loss.
    gate = []
    for poison in (False, True):
        full = torch.cat([g_local, torch.tensor([1.5 + rank if not (poison and rank == 1) else float("inf")])])
        ex = train.GradientExchange(world)
        ex.slice_ready(full, cut, full.numel())          # decoder + encoder slice + the loss word
        ex.slice_ready(full, 0, cut)                     # ConvStack slice
        full = ex.finish(full)
        assert ex.issued == 2
        gate.append(float(full[-1]))
        if not poison:
            assert torch.equal(full[:-1], g), "the extra word must not change the reduced gradient"
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), params=flat_p.numpy(), g_local=g_local.numpy(), g=g.numpy(), g2=g2.numpy(), gate=np.array(gate))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_gradient_exchange(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r = [np.load(tmp_path / f"rank{i}.npz") for i in range(world)]
    from piano_a2s_amd import spec
    cfg = spec.default_cfg(**SMALL)
    P0, _ = spec.split_state(spec.procedural_state(cfg, 11))
    ref_p = torch.cat([v.reshape(-1) for v in P0.values()]).numpy()
    assert np.array_equal(r[0]["params"], ref_p) and np.array_equal(r[1]["params"], ref_p), "broadcast from rank 0"
    assert np.array_equal(r[0]["g"], r[1]["g"]), "both ranks must hold the identical reduced gradient"
    assert np.array_equal(r[0]["g2"], r[0]["g"]) and np.array_equal(r[1]["g2"], r[0]["g"]), "slice-wise overlapped exchange == one all-reduce"
    mean = (r[0]["g_local"].astype(np.float64) + r[1]["g_local"].astype(np.float64)) / 2
    assert np.abs(r[0]["g"] - mean).max() <= 1e-6 * max(1.0, np.abs(mean).max())
    assert np.abs(r[0]["g_local"] - r[1]["g_local"]).max() > 0, "ranks must have seen different minibatches"
    assert r[0]["gate"][0] == r[1]["gate"][0] == 2.0, "mean of the ranks' losses 1.5 and 2.5, identical on both ranks"
    assert not np.isfinite(r[0]["gate"][1]) and not np.isfinite(r[1]["gate"][1]), "one rank's non-finite loss must gate the update on every rank"
