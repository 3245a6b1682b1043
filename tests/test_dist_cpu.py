"""Multi-process (gloo, world_size 2) test of the data-parallel gradient exchange used by the N>1 bench path."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from protopformer_amd.engine import GradSync
    torch.manual_seed(100 + rank)
    g = torch.randn(1000)
    mine = g.clone()
    sync = GradSync(g, [0, 128, 640, 1000], use_side_stream=False)
    # chunks become ready in backward order (last chunk first)
    for c in (2, 1, 0):
        sync.chunk_ready(c)
    scale = sync.finish()
    gathered = [torch.zeros(1000) for _ in range(world)]
    dist.all_gather(gathered, mine)
    expect = sum(gathered)
    ok = torch.allclose(g, expect, atol=1e-6) and abs(scale - 1.0 / world) < 1e-12
    # the non-finite guard is the loss summed over the ranks: one bad rank makes EVERY rank's guard non-finite (same decision everywhere)
    good = sync.reduce_guard(torch.tensor(1.5 + rank)); sync.finish()
    ok = ok and abs(float(good) - sum(1.5 + r for r in range(world))) < 1e-6
    bad = sync.reduce_guard(torch.tensor(float("nan") if rank == 1 else 2.0)); sync.finish()
    ok = ok and not bool(torch.isfinite(bad).all()) and bad.data_ptr() == good.data_ptr()
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_grad_sync_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]


def test_grad_sync_single_process_is_noop():
    from protopformer_amd.engine import GradSync
    g = torch.arange(10.0)
    s = GradSync(g, [0, 5, 10], use_side_stream=False)
    s.chunk_ready(1); s.chunk_ready(0)
    assert s.finish() == 1.0 and torch.equal(g, torch.arange(10.0))
    loss = torch.tensor(3.0)
    assert s.reduce_guard(loss).data_ptr() == loss.data_ptr() and s.launched == 0          # single rank: the loss itself, no collective
