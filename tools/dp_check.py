"""Data-parallel correctness on ONE GPU: two processes (gloo backend, both on cuda:0) each take half of a minibatch through the fused
training step with synchronised BatchNorm and the overlapped gradient exchange; the parameters they end up with must be identical on
both ranks and equal to ONE process stepping on the whole minibatch (the loss is a mean over rows, so the halves must hold the same
number of counted targets for exact equality: the lengths are mirrored between the halves).
usage: python tools/dp_check.py [--local-bn]

--local-bn: the DEFAULT data-parallel setting instead (per-rank BatchNorm statistics, plain DDP semantics): the two ranks' parameters must be
identical and equal to ONE process that computes the gradient of each half separately, averages the two and applies the fused clip + Adadelta
(which is what DDP's all-reduce-mean does to per-replica gradients)."""
import os
import random
import sys
import tempfile

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CFG = dict(max_length=(14, 9), max_bars=3)


def make(dev):
    import models
    from piano_a2s_amd import spec, synthetic
    cfg = spec.default_cfg(**CFG)
    st = spec.procedural_state(cfg, 5, eos_bias=1.0, lively="token")
    half = synthetic.make_batch(3, cfg, 21, frames=61, upper_range=(2, 13), lower_range=(2, 8), full_tail=0.0, spectrogram="ridges")
    other = synthetic.make_batch(3, cfg, 22, frames=61, upper_range=(2, 13), lower_range=(2, 8), full_tail=0.0, spectrogram="ridges")
    # same targets in both halves (so every loss term has the same row count per rank), different spectrograms
    other = [other[0]] + list(half[1:])
    m = models.ScoreTranscription(**cfg)
    m.load_state_dict(st)
    return cfg, m.to(dev), half, other


def worker(rank, world, port, outdir, sync_bn=True):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    from piano_a2s_amd import train
    cfg, m, half, other = make(dev)
    m.train()
    step = train.TrainStep(m, dropout=False, sync_bn=sync_bn)
    mine = half if rank == 0 else other
    batch = [t.to(dev) if torch.is_tensor(t) else t for t in mine]
    losses = step(batch, 1.0, rng=random.Random(3))
    torch.cuda.synchronize()
    torch.save({"flat": step.flat.cpu(), "losses": losses.cpu()}, os.path.join(outdir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def main():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    world = 2
    outdir = tempfile.mkdtemp()
    local_bn = "--local-bn" in sys.argv
    mp.spawn(worker, args=(world, port, outdir, not local_bn), nprocs=world, join=True)
    r = [torch.load(os.path.join(outdir, f"rank{i}.pt")) for i in range(world)]
    same = float((r[0]["flat"] - r[1]["flat"]).abs().max())
    if local_bn:
        # one process: gradient of each half on its own (per-rank statistics), mean of the two, ONE fused clip + Adadelta
        dev = torch.device("cuda:0")
        from piano_a2s_amd import train
        grads = []
        for part in ("half", "other"):
            cfg, m, half, other = make(dev)
            m.train()
            st = train.TrainStep(m, dropout=False)
            st.keep_grads = True
            mine = half if part == "half" else other
            st([t.to(dev) if torch.is_tensor(t) else t for t in mine], 1.0, rng=random.Random(3))
            torch.cuda.synchronize()
            grads.append({k: v.clone() for k, v in st.last_grads.items()})
        cfg, m, half, other = make(dev)
        flat = m.flatten_()
        layout = m.flat_layout()
        g = torch.zeros_like(flat)
        for (off, shape), name in zip(layout, [k for k, _ in m.named_parameters()]):
            n = grads[0][name].numel()
            g[off:off + n] = ((grads[0][name] + grads[1][name]) / 2).reshape(-1)
        opt = train.FusedAdadelta(flat, layout=layout)
        opt.step(g, torch.ones(1, device=dev), zero_grad=False)
        torch.cuda.synchronize()
        ref = flat.cpu()
        err = float((r[0]["flat"] - ref).abs().max() / ref.abs().max())
        print(f"ranks identical: max |p0 - p1| = {same:.3e};  per-rank BatchNorm, 2 ranks vs averaged single-process gradients: max rel parameter error {err:.3e}")
        assert same == 0.0 and err < 1e-5
        return
    # one process, the whole minibatch
    dev = torch.device("cuda:0")
    from piano_a2s_amd import train
    cfg, m, half, other = make(dev)
    m.train()
    step = train.TrainStep(m, dropout=False)
    whole = [torch.cat([a, b]).to(dev) if torch.is_tensor(a) else a + b for a, b in zip(half, other)]
    losses = step(whole, 1.0, rng=random.Random(3))
    torch.cuda.synchronize()
    ref = step.flat.cpu()
    err = float((r[0]["flat"] - ref).abs().max() / ref.abs().max())
    print(f"ranks identical: max |p0 - p1| = {same:.3e};  2 ranks x 3 clips vs 1 process x 6 clips: max rel parameter error {err:.3e}")
    print("loss terms  dp:", [round(float(x), 6) for x in (r[0]["losses"][:, 0] + r[1]["losses"][:, 0]) / 2], " single:", [round(float(x), 6) for x in losses[:, 0].cpu()])
    assert same == 0.0 and err < 2e-5


if __name__ == "__main__":
    main()
