"""ShardedXDeepFMEngine on one GPU: with world size 1 the exchange is the identity, so a step
must reproduce the plain engine (same kernels, rows fetched through the owner path) and the
oracle.  The cross-rank routing itself is covered on CPU by tests/test_sharded_exchange.py;
real multi-GPU runs are the driver's scaling bench."""
import numpy as np
import pytest
import torch

from oracle import xdeepfm_oracle as X
from recommenders_amd.models.deeprec.models._engine import XDeepFMEngine
from recommenders_amd.models.deeprec.models._sharded import ShardedXDeepFMEngine
from tests.util import make_hp, random_batch

pytestmark = pytest.mark.gpu


def _upload(eng, b):
    return eng.upload(b["labels"], b["dnn_feat_indices"], b["dnn_feat_values"], b["dnn_feat_weights"])


@pytest.mark.parametrize("virtual_world", [1, 3])
@pytest.mark.parametrize("optimizer", ["adam", "adagrad"])
def test_sharded_step_equals_plain_engine_and_oracle(optimizer, virtual_world):
    """virtual_world = 3: tokens are partitioned as for 3 owners (all of them this process), so
    the kernels see the permuted token order and the inverse map of a real multi-rank step."""
    hp = make_hp(optimizer=optimizer, learning_rate=0.01, FEATURE_COUNT=300, cross_l2=0.01, cross_l1=0.001)
    rng = np.random.default_rng(5)
    p32 = X.init_params(hp, seed=9)
    plain = XDeepFMEngine(hp, embedding_update="sparse_lazy")
    shard = ShardedXDeepFMEngine(hp, virtual_world=virtual_world)
    assert shard.world == 1 and shard.N == 300
    plain.load_state({k: v.clone() for k, v in p32.items()})
    shard.load_global_state({k: v.clone() for k, v in p32.items()})
    p64 = {k: v.double() for k, v in p32.items()}
    st = X.new_opt_state(p64, hp)
    for step in range(3):
        b = random_batch(rng, hp, 64 if step != 1 else 33)
        l0, d0 = plain.train_step(_upload(plain, b))
        l1, d1 = shard.train_step(_upload(shard, b))
        o_loss, o_dl, _, _ = X.train_step(p64, st, b, hp, mode="sparse_lazy")
        assert abs(float(d1) - o_dl) <= 1e-5 * abs(o_dl)
        assert abs(float(l1) - o_loss) <= 1e-5 * abs(o_loss)
        # same arithmetic in the same order: the two engines agree to the bit on the forward
        assert float(d0) == float(d1)
    torch.cuda.synchronize()
    shard.check_errors()
    table, lin = shard.gather_global_table()
    tol = 2e-4 if optimizer == "adam" else 2e-5
    for got, name in ((table, "embedding"), (lin.view(-1, 1), "lin_w")):
        exp = p64[name].numpy()
        err = np.abs(got.cpu().numpy() - exp).max()
        assert err <= tol * np.abs(exp).max(), (name, err)
    for k in shard.dense_shapes:
        exp = p64[k].numpy().reshape(shard.dense_shapes[k])
        err = np.abs(shard.views[k].cpu().numpy() - exp).max()
        assert err <= tol * max(np.abs(exp).max(), 1e-30), (k, err)
    # plain engine took the same steps
    assert np.abs(plain.table.cpu().numpy() - table.cpu().numpy()).max() <= 1e-6


def _free_port():
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _two_rank_worker(rank, world, port, optimizer, out):
    """One rank of a real multi-process step.  The ranks share cuda:0 (a one-GPU box) and talk
    over gloo with host-staged exchanges; everything else is the code of an N-GPU run."""
    import os

    import torch.distributed as dist

    from oracle import ffm_iterator_oracle as O
    from tests.util import random_features

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        hp = make_hp(optimizer=optimizer, learning_rate=0.01, FEATURE_COUNT=301, cross_l2=0.01, cross_l1=0.001)
        F, N = hp.FIELD_COUNT, hp.FEATURE_COUNT
        p32 = X.init_params(hp, seed=9)
        eng = ShardedXDeepFMEngine(hp, seed=7)
        assert eng.world == world and eng.exchange.host_staged
        eng.load_global_state({k: v.clone() for k, v in p32.items()})
        p64 = {k: v.double() for k, v in p32.items()}
        st = X.new_opt_state(p64, hp)
        rng = np.random.default_rng(21)  # every rank draws the same global batches
        per = 48
        steps = []
        for step in range(3):
            labels, feats = random_features(rng, per * world, F, N)
            for toks in feats[:8]:
                toks.append([0, 5, 1.0])  # a row that several ranks touch in one step
            mine = O.convert(labels[rank * per:(rank + 1) * per], feats[rank * per:(rank + 1) * per], F, N)
            steps.append((labels, feats, _upload(eng, mine)))
        for step, (labels, feats, dev_batch) in enumerate(steps):
            # the next batch's ids are routed beside this step (prefetch_route), except the last one's
            nxt = steps[step + 1][2].ids if step + 1 < len(steps) else None
            loss, dloss = eng.train_step(dev_batch, next_ids=nxt)
            assert (eng._route_pre is not None) == (nxt is not None)
            tot = torch.stack([loss, dloss]).double().cpu()
            dist.all_reduce(tot)
            o_loss, o_dl, _, _ = X.train_step(p64, st, O.convert(labels, feats, F, N), hp, mode="sparse_lazy")
            assert abs(float(tot[1]) - o_dl) <= 2e-5 * abs(o_dl), (step, float(tot[1]), o_dl)
            assert abs(float(tot[0]) - o_loss) <= 2e-5 * abs(o_loss), (step, float(tot[0]), o_loss)
        torch.cuda.synchronize()
        eng.check_errors()
        table, lin = eng.gather_global_table()
        tol = 2e-4 if optimizer == "adam" else 2e-5
        for got, name in ((table, "embedding"), (lin.view(-1, 1), "lin_w")):
            exp = p64[name].numpy()
            err = np.abs(got.cpu().numpy() - exp).max()
            assert err <= tol * np.abs(exp).max(), (name, err)
        for k in eng.dense_shapes:
            exp = p64[k].numpy().reshape(eng.dense_shapes[k])
            err = np.abs(eng.views[k].cpu().numpy() - exp).max()
            assert err <= tol * max(np.abs(exp).max(), 1e-30), (k, err)
        out.put((rank, "ok"))
    except Exception:  # noqa: BLE001
        import traceback

        out.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("optimizer", ["adam", "adagrad"])
def test_two_ranks_sharing_the_gpu_match_the_oracle_on_the_global_batch(optimizer):
    """world_size 2, one process per rank: ids to their owners, rows back, gradient rows to the
    owners, owner-side update, dense all-reduce -- against the oracle stepping on the
    concatenated batch.  (RCCL needs one GPU per rank; the driver's scaling bench is that run.)"""
    import torch.multiprocessing as mp

    world = 2
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_two_rank_worker, args=(r, world, port, optimizer, out)) for r in range(world)]
    for p in procs:
        p.start()
    res = [out.get(timeout=110) for _ in procs]
    for p in procs:
        p.join(timeout=30)
    assert sorted(res) == [(r, "ok") for r in range(world)], res
