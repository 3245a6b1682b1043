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
    # bf16 wire format (round 6): same chunks, each summand and the sum rounded once to bf16 -> the fp32 buffer holds the widened result,
    # within 2 bf16 roundings (2 x 2^-9 relative) of the fp32 exchange, and EXACTLY bf16(bf16(a) + bf16(b)) for two ranks
    g16 = mine.clone()
    s16 = GradSync(g16, [0, 128, 640, 1000], use_side_stream=False, payload="bf16")
    for c in (2, 1, 0):
        s16.chunk_ready(c)
    ok = ok and s16.finish() == 1.0 / world and s16.payload == "bf16" and s16.launched == 3
    exact = sum(t.bfloat16().float() for t in gathered).bfloat16().float()
    ok = ok and torch.equal(g16, exact)
    ok = ok and bool(((g16 - expect).abs() <= 2 ** -7 * expect.abs() + 2 ** -8 * max(t.abs().max() for t in gathered)).all())
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


def test_chunk_plan_orders_by_readiness():
    """make_grad_sync's chunk boundaries (engine.chunk_plan, pure): the chunk that backward completes LAST -- [embedding + block 0] -- is the
    smallest block chunk, the first to go is [final norm + heads]; four chunks (every collective costs launch time)."""
    from protopformer_amd.engine import chunk_plan, readiness_cuts
    assert readiness_cuts(12) == [1, 6] and readiness_cuts(24) == [1, 12] and readiness_cuts(2) == [1] and readiness_cuts(1) == []
    ents, off = [("features.cls_token", 0), ("features.pos_embed", 8), ("features.patch_embed.proj.weight", 100)], 1000
    for i in range(12):
        ents += [(f"features.blocks.{i}.norm1.weight", off), (f"features.blocks.{i}.attn.qkv.weight", off + 8)]
        off += 1000
    ents += [("features.norm.weight", off), ("prototype_vectors", off + 16)]
    bounds, block_chunk = chunk_plan(ents, off + 500)
    assert bounds == [0, 2000, 7000, 13000, 13500]
    assert block_chunk == {1: 1, 6: 2}                            # block i done -> the chunk that starts at block i is complete
    sizes = [b - a for a, b in zip(bounds, bounds[1:])]
    assert sizes[0] == min(sizes[:3])                             # the chunk ready last is the smallest of the block chunks
    b4, bc4 = chunk_plan(ents, off + 500, n_chunks=4)             # the pre-round-6 equal partition, kept for A/B
    assert b4 == [0, 5000, 9000, 13000, 13500] and bc4 == {4: 1, 8: 2}
    bg, bcg = chunk_plan(ents, off + 500, cuts=[1, 2, 4, 8])      # an explicit (geometric) partition
    assert bg == [0, 2000, 3000, 5000, 9000, 13000, 13500] and bcg == {1: 1, 2: 2, 4: 3, 8: 4}
    b2, bc2 = chunk_plan(ents, off + 500, cuts=[6, 0, 12, 99])    # out-of-range cuts are dropped
    assert b2 == [0, 7000, 13000, 13500] and bc2 == {6: 1}
