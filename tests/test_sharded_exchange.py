"""The N > 1 path without GPUs: RowShardExchange (routing of ids to the ranks that own their
table rows, rows back, gradients out) on CPU tensors with the gloo backend, world_size 2 and 3.
The owner-side arithmetic that libxdfm.so does on a GPU is played here by plain torch indexing,
so what is checked is the exchange itself: every token gets the row of its id, every gradient
reaches the owner of its id, in an order that does not depend on timing."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from recommenders_amd.models.deeprec.models._sharded import RowShardExchange, local_rows


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, n_rows, dim, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(1)
        table = torch.randn(n_rows, dim)              # the full table, same on every rank
        shard = table[rank::world].clone()            # what this rank owns
        assert shard.shape[0] == local_rows(n_rows, rank, world)
        rng = np.random.default_rng(100 + rank)
        n = 500 + 37 * rank                           # ragged: ranks hold different token counts
        ids = torch.from_numpy(rng.integers(0, n_rows, n))
        ids[:50] = 3                                  # a hot id
        if rank == world - 1:
            ids = ids[ids % world != 0]               # this rank asks owner 0 for nothing
            n = int(ids.numel())
        ex = RowShardExchange()
        plan = ex.route(ids, n_rows)
        # every distinct id is asked for once: one slot per distinct row, grouped by owner
        assert plan.n_send == len(torch.unique(ids)) and sum(plan.send_counts) == plan.n_send
        # owner side: local row -> row, in the order the requests arrived
        assert int(plan.recv_rows.max()) < shard.shape[0]
        rows_back = ex.to_source(plan, shard[plan.recv_rows])
        got = rows_back[plan.inverse]                 # token order
        assert torch.equal(got, table[ids])
        # the sorted order is by (owner, row), equal ids in token order (stable)
        order = plan.order.long()
        dest = (ids % world)[order]
        assert torch.equal(dest, torch.sort(dest).values)
        key = (ids % world) * ((n_rows + world - 1) // world) + ids // world
        assert torch.equal(order, torch.argsort(key, stable=True))
        assert torch.equal(plan.inverse[order], plan.slot_sorted.long())
        # gradients out: the tokens of one row pre-reduced at the source, one record per slot, summed at
        # the owner in arrival order
        grads = torch.randn(n, dim, dtype=torch.float64)
        pre = torch.zeros(plan.n_send, dim, dtype=torch.float64).index_add_(0, plan.inverse, grads)
        recv_g = ex.to_owner(plan, pre)
        acc = torch.zeros(shard.shape[0], dim, dtype=torch.float64)
        acc.index_add_(0, plan.recv_rows, recv_g)
        # reference: every rank's contributions gathered the slow way
        all_ids = [None] * world
        all_g = [None] * world
        dist.all_gather_object(all_ids, ids)
        dist.all_gather_object(all_g, grads)
        ref = torch.zeros(n_rows, dim, dtype=torch.float64)
        for i, g in zip(all_ids, all_g):
            ref.index_add_(0, i, g)
        assert torch.allclose(acc, ref[rank::world], rtol=1e-12, atol=1e-12)
        # arrival order = (source rank, source token order): reproducible
        counts = plan.recv_counts
        assert sum(counts) == plan.n_recv and len(counts) == world
        # ---- the same exchange with requests of a fixed number of slots per owner: no sizes travel, the
        # unused slots name the owner's scratch row (kept past its real rows) and carry zero gradients
        gmax = plan.gmax
        assert gmax == max(max(np.bincount((np.unique(i.numpy()) % world), minlength=world)) for i in all_ids)
        cap = ex.grow_capacity(gmax)
        assert cap % 64 == 0 and cap >= gmax * 1.25
        pad_row = (n_rows + world - 1) // world        # one index every rank may use as its scratch row
        fp = ex.route_fixed(ids, n_rows, pad_row)
        assert ex.fits(fp) and fp.gmax == gmax and fp.fixed and fp.n_send == fp.n_recv == world * (cap + 1)
        shard_p = torch.zeros(pad_row + 1, dim)
        shard_p[:shard.shape[0]] = shard
        assert int(fp.recv_rows.max()) <= pad_row
        rows_back = ex.to_source(fp, shard_p[fp.recv_rows])
        assert torch.equal(rows_back[fp.inverse], table[ids])
        assert torch.equal(fp.inverse[fp.order.long()], fp.slot_sorted.long())
        pre = torch.zeros(fp.n_send, dim, dtype=torch.float64).index_add_(0, fp.inverse, grads)
        recv_g = ex.to_owner(fp, pre)
        acc = torch.zeros(pad_row + 1, dim, dtype=torch.float64)
        acc.index_add_(0, fp.recv_rows, recv_g)
        assert torch.allclose(acc[:shard.shape[0]], ref[rank::world], rtol=1e-12, atol=1e-12)
        assert not acc[pad_row].any()                  # the scratch row got zeros only
        # a request too small for the batch says so on every rank (the caller then routes with sizes)
        ex.cap = 8
        small = ex.route_fixed(ids, n_rows, pad_row)
        assert not ex.fits(small) and small.gmax == gmax
        out.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        out.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_row_shard_exchange_gloo(world):
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, 1001, 6, out)) for r in range(world)]
    for p in procs:
        p.start()
    res = [out.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(r, "ok") for r in range(world)], res


def test_single_rank_exchange_is_identity():
    ex = RowShardExchange()
    ids = torch.tensor([5, 1, 5, 9])
    plan = ex.route(ids, 10)
    # one rank: nothing travels, but duplicates still collapse to one slot per distinct row
    assert plan.n_send == plan.n_recv == 3 and plan.recv_rows.tolist() == [1, 5, 9]
    assert plan.inverse.tolist() == [1, 0, 1, 2] and plan.order.tolist() == [1, 0, 2, 3]
    x = torch.arange(6.0).reshape(3, 2)
    assert ex.to_owner(plan, x) is x and ex.to_source(plan, x) is x
    assert [local_rows(10, r, 4) for r in range(4)] == [3, 3, 2, 2]
