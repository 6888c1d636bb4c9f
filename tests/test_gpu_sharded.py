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
    plain = XDeepFMEngine(hp, embedding_update="sparse_rows")
    shard = ShardedXDeepFMEngine(hp, virtual_world=virtual_world)
    assert shard.world == 1 and shard.n_local == 300 and shard.pad_row == 300 and shard.N == 301  # + scratch row
    plain.load_state({k: v.clone() for k, v in p32.items()})
    shard.load_global_state({k: v.clone() for k, v in p32.items()})
    p64 = {k: v.double() for k, v in p32.items()}
    st = X.new_opt_state(p64, hp)
    batches = [random_batch(rng, hp, 64 if step not in (1, 4) else 33) for step in range(6)]
    dev = [_upload(shard, b) for b in batches]
    for step, b in enumerate(batches):
        l0, d0 = plain.train_step(_upload(plain, b))
        # steps 0-2 announce their successor (routing one step ahead), 3 does not, 4 does
        nxt = dev[step + 1] if (step + 1 < len(dev) and step != 3) else None
        if step == 2 and virtual_world == 1:
            # a request too small for the batch that was routed ahead with it: found out (without a wait:
            # the answer travelled to the host a step ago), the batch goes with sized requests instead and
            # the capacity grows
            assert shard._route_pre is not None and shard._route_pre[1].fixed
            shard._route_pre[1].cap = 1
        l1, d1 = shard.train_step(dev[step], next_batch=nxt)
        # the reference's semantics: every row moves every step (rows are caught up at their owner)
        o_loss, o_dl, _, _ = X.train_step(p64, st, b, hp, mode="dense_exact")
        assert abs(float(d1) - o_dl) <= 1e-5 * abs(o_dl)
        assert abs(float(l1) - o_loss) <= 1e-5 * abs(o_loss)
        # same rows read, same forward arithmetic: the two engines agree to the bit on the data loss
        assert float(d0) == float(d1)
    torch.cuda.synchronize()
    shard.check_errors()
    if virtual_world == 1:
        # the first batch finds the capacity (sized), the forced miss is the other sized step; the rest moved
        # fixed-size requests and never asked the host for sizes
        assert (shard.sized_steps, shard.fixed_steps) == (2, 4) and shard.exchange.cap >= 64
    else:
        assert shard.fixed_steps == 0  # (virtual owners are a sized-request test vehicle)
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
    # the scratch row took zero gradients only
    assert not shard.table[shard.n_local:].any()
    # plain engine took the same steps
    assert np.abs(plain.state_dict()["embedding"].cpu().numpy() - table.cpu().numpy()).max() <= 1e-6


@pytest.mark.parametrize("embedding_update", ["sparse_rows", "dense_exact", "sparse_rows+bn", "sparse_rows+dropout",
                                              "sparse_rows+bf16"])
def test_whole_step_graph_equals_eager_exchanges(embedding_update):
    """Batches of one shape, each announcing its successor: once both routing generations have replayed their
    local graphs the whole step -- exchanges, owner-side update, the next batch's routing -- is one graph per
    generation (_StepGraph).  Bit for bit the engine whose exchanges stay eager, through a batch that does not
    fit its request (found out from the tagged answer of the routing inside the previous graph; that batch
    goes with sized requests) and a step without an announced successor."""
    embedding_update, _, variant = embedding_update.partition("+")
    extra = dict(embed_l2=0.01, layer_l2=0.01) if embedding_update == "dense_exact" else {}
    if variant == "bn":  # moving statistics move once per step, captured or not
        extra.update(enable_BN=True, loss="cross_entropy_loss")
    if variant == "dropout":  # masks are drawn from the device-side step count
        extra.update(user_dropout=True, dropout=[0.3, 0.2])
    dtype = "bf16" if variant == "bf16" else None
    hp = make_hp(optimizer="adam", learning_rate=0.01, FEATURE_COUNT=300, cross_l2=0.01, **extra)
    rng = np.random.default_rng(15)
    p32 = X.init_params(hp, seed=9)
    whole = ShardedXDeepFMEngine(hp, seed=3, embedding_update=embedding_update, compute_dtype=dtype)
    eager = ShardedXDeepFMEngine(hp, seed=3, embedding_update=embedding_update, compute_dtype=dtype)
    eager.graph_step = False
    assert whole.graph_step
    for e in (whole, eager):
        e.load_global_state({k: v.clone() for k, v in p32.items()})
    # (one token per field: every batch has the same number of tokens, as the graph asks)
    batches = [random_batch(rng, hp, 64, multi_p=0.0, missing_p=0.0) for _ in range(14)]
    dev_w = [_upload(whole, b) for b in batches]
    dev_e = [_upload(eager, b) for b in batches]
    used = []
    for k in range(len(batches)):
        last = k + 1 == len(batches) or k == 9  # step 9 announces nothing: step 10 routes its own batch
        forced = k == 7 and whole._route_pre is not None and whole._route_pre[1].fit_tagged is not None
        if forced:
            plan = whole._route_pre[1]
            keep, plan.cap = plan.cap, 1  # "does not fit"
        before = whole.graph_steps
        lw = whole.train_step(dev_w[k], next_batch=None if last else dev_w[k + 1])
        le = eager.train_step(dev_e[k], next_batch=None if last else dev_e[k + 1])
        if forced:
            plan.cap = keep
            assert whole.graph_steps == before  # that batch went with sized requests
        used.append(whole.graph_steps - before)
        assert float(lw[0]) == float(le[0]) and float(lw[1]) == float(le[1]), (k, used)
    torch.cuda.synchronize()
    whole.check_errors()
    assert sum(used) >= 5 and eager.graph_steps == 0, used
    assert used[7] == 0 and used[9] == 0 and used[13] == 0
    assert whole.sized_steps == 2  # the first batch and the forced miss
    a, b = whole.state_dict(), eager.state_dict()
    for key in a:
        if torch.is_tensor(a[key]):
            assert torch.equal(a[key], b[key]), key
    assert whole.step == eager.step == len(batches)


def test_sharded_batch_norm_without_peers_equals_plain_engine():
    """enable_BN with nothing split: the plain normalisation kernels, inside the captured local graph."""
    hp = make_hp(optimizer="adam", learning_rate=0.01, FEATURE_COUNT=300, enable_BN=True, loss="cross_entropy_loss")
    rng = np.random.default_rng(8)
    p32 = X.init_params(hp, seed=9)
    plain = XDeepFMEngine(hp, embedding_update="sparse_rows")
    shard = ShardedXDeepFMEngine(hp)
    assert shard.graph_local
    plain.load_state({k: v.clone() for k, v in p32.items()})
    shard.load_global_state({k: v.clone() for k, v in p32.items()})
    for step in range(4):
        b = random_batch(rng, hp, 64)
        _, d0 = plain.train_step(_upload(plain, b))
        _, d1 = shard.train_step(_upload(shard, b))
        assert float(d0) == float(d1), step
    assert shard._local_graphs and all(g.graph is not None for g in shard._local_graphs.values())
    a, b_ = plain.state_dict(), shard.state_dict()
    for k in a:
        if k.startswith("bnstat_"):
            assert torch.equal(a[k], b_[k]), k


def test_device_partition_equals_its_torch_restatement():
    """xdfm_route_keys + the radix sort + xdfm_route_uniq against RowShardExchange._partition_host (the
    torch form the gloo rehearsal on CPU tensors runs): bit-exact, incl. hot ids, an owner without
    a row and ragged sizes around the kernels' 1024-position tiles."""
    from recommenders_amd.models.deeprec.models._sharded import RowShardExchange

    ex = RowShardExchange()
    rng = np.random.default_rng(3)
    for n, n_rows, G in ((1, 10, 1), (1023, 50, 3), (1024, 1000, 8), (1025, 7, 8), (50000, 1 << 20, 8),
                         (319488, 1 << 25, 8), (4000, 1000, 4)):
        ids = rng.integers(0, n_rows, n)
        ids[rng.random(n) < 0.3] = ids[0]
        if G == 4:
            ids = ids[ids % 4 != 2]  # owner 2 gets nothing
        ids = torch.from_numpy(ids)
        host = ex._partition_host(ids, n_rows, G)
        devp = ex._partition_device(ids.cuda(), n_rows, G)
        torch.cuda.synchronize()
        n_uniq = int(host[4][G])
        assert torch.equal(host[4], devp[4].cpu()), (n, n_rows, G)
        for h, d, name in zip(host[:3], devp[:3], ("order", "slot_sorted", "inverse")):
            assert torch.equal(h, d.cpu()), (name, n, n_rows, G)
        assert torch.equal(host[3], devp[3][:n_uniq].cpu())
        # the request in fixed-capacity form: kernel against its torch restatement, for a capacity that
        # holds the batch and (values of the slots that exist) for one that does not
        counts_ref = None
        nt = int(ids.numel())
        for cap in (max(64, nt), 3):
            hp_ = ex._pad(host[3], host[1], host[2], host[4], nt, G, cap, 12345)
            dp_ = ex._pad(devp[3], devp[1], devp[2], devp[4], nt, G, cap, 12345)
            torch.cuda.synchronize()
            assert torch.equal(hp_[3], dp_[3].cpu()), (n, G, cap)
            counts_ref = hp_[3]
            if int(hp_[3][G]) <= cap:
                for h, d, name in zip(hp_[:3], dp_[:3], ("rows_padded", "slot_sorted_p", "inverse_p")):
                    assert torch.equal(h, d.cpu()), (name, n, n_rows, G, cap)
        assert int(counts_ref[:G].sum()) == n_uniq


def _free_port():
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _two_rank_worker(rank, world, port, optimizer, loss, out):
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
        bn = loss == "batch_norm"
        dense = loss == "dense_exact"
        if dense:
            # the notebook's kind of configuration (xdeepfm_criteo.ipynb:142-163: all three l2 = 0.01) plus the
            # l1 terms: the table's regularisers and the dense-exact optimizer are shard-local
            hp = make_hp(optimizer=optimizer, learning_rate=0.01, FEATURE_COUNT=301, cross_l2=0.01, cross_l1=0.001,
                         embed_l2=0.01, layer_l2=0.01, embed_l1=0.001, layer_l1=0.001)
        elif bn:
            # cross_entropy_loss: see test_batch_norm_training_matches_oracle (logits of +-10 under log_loss)
            hp = make_hp(optimizer=optimizer, learning_rate=0.01, FEATURE_COUNT=301, cross_l2=0.01,
                         loss="cross_entropy_loss", enable_BN=True)
        else:
            hp = make_hp(optimizer=optimizer, learning_rate=0.01, FEATURE_COUNT=301, cross_l2=0.01, cross_l1=0.001,
                         loss=loss)
        F, N = hp.FIELD_COUNT, hp.FEATURE_COUNT
        p32 = X.init_params(hp, seed=9)
        eng = ShardedXDeepFMEngine(hp, seed=7, embedding_update="dense_exact" if dense else "sparse_rows")
        assert eng.world == world and eng.exchange.host_staged
        assert eng.catchup_on == (optimizer == "adam" and not dense)
        eng.load_global_state({k: v.clone() for k, v in p32.items()})
        p64 = {k: v.double() for k, v in p32.items()}
        st = X.new_opt_state(p64, hp)
        rng = np.random.default_rng(21)  # every rank draws the same global batches
        per = 48
        steps = []
        for step in range(5):
            labels, feats = random_features(rng, per * world, F, N)
            for toks in feats[:8]:
                toks.append([0, 5, 1.0])  # a row that several ranks touch in one step
            mine = O.convert(labels[rank * per:(rank + 1) * per], feats[rank * per:(rank + 1) * per], F, N)
            steps.append((labels, feats, _upload(eng, mine)))
        for step, (labels, feats, dev_batch) in enumerate(steps):
            # the next batch's ids are routed beside this step (prefetch_route), except the last one's
            nxt = steps[step + 1][2] if step + 1 < len(steps) else None
            loss, dloss = eng.train_step(dev_batch, next_batch=nxt)
            assert (eng._route_pre is not None) == (nxt is not None)
            tot = torch.stack([loss, dloss]).double().cpu()
            dist.all_reduce(tot)
            o_loss, o_dl, _, _ = X.train_step(p64, st, O.convert(labels, feats, F, N), hp, mode="dense_exact")
            assert abs(float(tot[1]) - o_dl) <= 2e-5 * abs(o_dl), (step, float(tot[1]), o_dl)
            assert abs(float(tot[0]) - o_loss) <= 2e-5 * abs(o_loss), (step, float(tot[0]), o_loss)
        torch.cuda.synchronize()
        eng.check_errors()
        # the first batch found the capacity with sized requests, every later one moved fixed-size messages
        assert (eng.sized_steps, eng.fixed_steps) == (1, len(steps) - 1)
        table, lin = eng.gather_global_table()
        tol = 2e-4 if optimizer == "adam" else 2e-5
        for got, name in ((table, "embedding"), (lin.view(-1, 1), "lin_w")):
            exp = p64[name].numpy()
            err = np.abs(got.cpu().numpy() - exp).max()
            assert err <= tol * np.abs(exp).max(), (name, err)
        for k in eng.dense_shapes:
            if bn and k.startswith("dnn_b_") and k != "dnn_b_out":
                continue  # zero gradient in front of a batch norm: Adam walks on rounding noise
            exp = p64[k].numpy().reshape(eng.dense_shapes[k])
            err = np.abs(eng.views[k].cpu().numpy() - exp).max()
            assert err <= (5e-4 if bn else tol) * max(np.abs(exp).max(), 1e-30), (k, err)
        assert bool(eng.bn_state) == bn
        for k, t in eng.bn_state.items():  # moving statistics of the GLOBAL batch, the same on every rank
            if k.startswith("bnstat_dnn_mm_"):
                continue  # follows the noise-walked bias
            exp = p64[k].numpy().reshape(t.shape)
            err = np.abs(t.cpu().numpy() - exp).max()
            assert err <= 5e-4 * max(np.abs(exp).max(), 1e-30), (k, err)
        out.put((rank, "ok"))
    except Exception:  # noqa: BLE001
        import traceback

        out.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("optimizer,loss", [("adam", "log_loss"), ("adagrad", "log_loss"), ("adam", "square_loss"),
                                            ("adam", "batch_norm"), ("adam", "dense_exact")])
def test_two_ranks_sharing_the_gpu_match_the_oracle_on_the_global_batch(optimizer, loss):
    """world_size 2, one process per rank: ids to their owners, rows back, gradient rows to the
    owners, owner-side update, dense all-reduce -- against the oracle stepping on the
    concatenated batch.  "dense_exact" = embedding_update="dense_exact" with embed / layer / cross l1 and l2 all on
    (the owners stream their rows with the regularisers folded in, the ranks' loss shares add up).  square_loss (a root of a mean over the GLOBAL batch) adds one scalar all-reduce
    between the forward and the backward pass; "batch_norm" = enable_BN, whose statistics are those of the global
    batch: one all-reduce per normalisation and pass.  (RCCL needs one GPU per rank; the driver's scaling bench is that run.)"""
    import torch.multiprocessing as mp

    world = 2
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_two_rank_worker, args=(r, world, port, optimizer, loss, out)) for r in range(world)]
    for p in procs:
        p.start()
    res = [out.get(timeout=110) for _ in procs]
    for p in procs:
        p.join(timeout=30)
    assert sorted(res) == [(r, "ok") for r in range(world)], res


def _rccl_worker(port, bn, out):
    """One rank over the nccl (= RCCL) backend with every collective of the N-GPU step issued for
    real (an exchange with oneself): the calls, dtypes and split lists of the multi-GPU path."""
    import faulthandler
    import os
    import sys

    import torch.distributed as dist

    faulthandler.dump_traceback_later(90, exit=True, file=sys.stderr)  # a rank that hangs names its line and ends
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        extra = dict(enable_BN=True, loss="cross_entropy_loss") if bn else {}
        hp = make_hp(optimizer="adam", learning_rate=0.01, FEATURE_COUNT=300, **extra)
        p32 = X.init_params(hp, seed=9)
        eng = ShardedXDeepFMEngine(hp, seed=7, force_collectives=True)
        assert eng.graph_local == (not bn)
        assert eng.exchange.collective and not eng.exchange.host_staged
        eng.load_global_state({k: v.clone() for k, v in p32.items()})
        p64 = {k: v.double() for k, v in p32.items()}
        st = X.new_opt_state(p64, hp)
        rng = np.random.default_rng(31)
        batches = [random_batch(rng, hp, 64, multi_p=0.0, missing_p=0.0) for _ in range(4 if bn else 10)]
        dev = [_upload(eng, b) for b in batches]
        for k, b in enumerate(batches):
            if k == 2 and not bn:
                eng._route_pre[1].cap = 1  # one forced miss: that batch goes with sized requests over RCCL
            loss, dloss = eng.train_step(dev[k], next_batch=dev[k + 1] if k + 1 < len(dev) else None)
            o_loss, o_dl, _, _ = X.train_step(p64, st, b, hp, mode="dense_exact")
            assert abs(float(dloss) - o_dl) <= 2e-5 * abs(o_dl), (k, float(dloss), o_dl)
        torch.cuda.synchronize()
        eng.check_errors()
        assert (eng.sized_steps, eng.fixed_steps) == ((1, 3) if bn else (2, 8))
        # the later steps ran as one graph each, the all-to-alls and the all-reduce captured with the kernels
        # (and eager collectives -- gather_global_table below -- still work behind them)
        assert eng.graph_steps >= (0 if bn else 3), eng.graph_steps
        table, _ = eng.gather_global_table()
        err = np.abs(table.cpu().numpy() - p64["embedding"].numpy()).max()
        assert err <= (5e-4 if bn else 2e-4) * np.abs(p64["embedding"].numpy()).max(), err
        eng.close()  # (captured collectives are released before the group is destroyed)
        out.put("ok")
    except Exception:  # noqa: BLE001
        import traceback

        out.put(traceback.format_exc())
    finally:
        dist.destroy_process_group()
        faulthandler.cancel_dump_traceback_later()


@pytest.mark.parametrize("bn", [False, True])
def test_collectives_of_the_sharded_step_run_over_rccl(bn):
    """bn: enable_BN, whose moments / sums all-reduces then run over RCCL too."""
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(_free_port(), bn, out))
    p.start()
    res = out.get(timeout=110)
    p.join(timeout=30)
    assert res == "ok", res
    assert p.exitcode == 0, "the rank did not end (exit code %r)" % (p.exitcode,)
