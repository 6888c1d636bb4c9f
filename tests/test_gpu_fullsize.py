"""The bench workload at BASELINE configs[2]'s full sizes (39 fields, 2^25 rows, dim 16, batch 8192):
too large for the CPU oracle, so the checks are properties that hold at any size -- the gather against
torch's own indexing (bit for bit), a checksum of the sparse update against the per-token
contributions, untouched rows, run-to-run determinism, graph replay against eager launches."""
import types

import numpy as np
import pytest
import torch

import bench
from recommenders_amd.models.deeprec.models._engine import XDeepFMEngine

pytestmark = pytest.mark.gpu

N, B, F, D = 1 << 25, 8192, 39, 16


@pytest.fixture(scope="module")
def setup():
    args = types.SimpleNamespace(features=N, dim=D, batch=B, workload="criteo")
    hp = bench.workload_hparams(args)
    batches = bench.synth_batches(hp, 2, seed=77)
    return hp, batches


def _dev(hb):
    return [torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in (
        hb["labels"].reshape(-1), hb["dnn_feat_indices"], hb["dnn_feat_values"], hb["dnn_feat_weights"])]


def test_gather_fm_and_linear_at_full_size(setup):
    hp, batches = setup
    eng = XDeepFMEngine(hp, seed=5, embedding_update="sparse_rows")
    labels, idx, ids, w = _dev(batches[0])
    db = eng.make_batch(labels, idx, ids, w)
    eng.forward(db, with_loss=False, with_grad=False)
    torch.cuda.synchronize()
    eng.check_errors()
    # one token per row: out = w * table[id], one fp32 multiply -> bit-exact against torch's gather
    exp = eng.table[ids] * w[:, None]
    got = eng.ws["emb_out"][:B].reshape(B * F, D)
    assert torch.equal(got.view(torch.int32), exp.view(torch.int32))
    # FM sum and linear term: sums of the same rows (order of the fp32 sums differs: tolerance)
    s = exp.reshape(B, F, D).double().sum(1)
    assert torch.allclose(eng.ws["fm_sum"][:B].double(), s, rtol=1e-5, atol=1e-6)
    fm = 0.5 * (s * s - (exp.reshape(B, F, D).double() ** 2).sum(1)).sum(1)
    assert torch.allclose(eng.ws["fm_out"][:B].double(), fm, rtol=1e-4, atol=1e-6)
    lin = (eng.lin_w[ids].double() * w.double()).reshape(B, F).sum(1)
    assert torch.allclose(eng.ws["lin_out"][:B].double(), lin, rtol=1e-5, atol=1e-6)


def test_sparse_sgd_update_checksums_and_untouched_rows(setup):
    """SGD makes the update linear in the gradient: the column sums of (table_after - table_before)
    must equal -lr * the column sums of the per-token contributions, and only rows whose id occurs in
    the batch may change (13 hot ids take 8192 tokens each: the long-segment path)."""
    hp, batches = setup
    hp = bench.workload_hparams(types.SimpleNamespace(features=N, dim=D, batch=B, workload="criteo"))
    hp.optimizer = "sgd"
    hp.learning_rate = 0.5
    eng = XDeepFMEngine(hp, seed=5, embedding_update="sparse_rows")
    labels, idx, ids, w = _dev(batches[1])
    before, lin_before = eng.table.clone(), eng.lin_w.clone()
    db = eng.make_batch(labels, idx, ids, w)
    eng.plan_async(db)
    eng.forward(db, with_loss=True, with_grad=True)
    eng.backward(db)
    torch.cuda.synchronize()
    # per-token contribution c_j = w_j*d_emb[row_j] + g[b_j]*w_j*(fm_sum[b_j] - w_j*table[id_j]), l_j = g[b_j]*w_j
    d_emb = eng.ws["d_emb"][:B].double().reshape(B * F, D)
    if eng.d_emb2() is not None:
        d_emb = d_emb + eng.d_emb2()[:B].double().reshape(B * F, D)
    g = eng.ws["g_logit"][:B].double().repeat_interleave(F)
    wd = w.double()
    fm_sum = eng.ws["fm_sum"][:B].double().repeat_interleave(F, dim=0)
    contrib = wd[:, None] * d_emb + (g * wd)[:, None] * (fm_sum - wd[:, None] * before[ids].double())
    lin_contrib = g * wd
    eng.update(db)
    torch.cuda.synchronize()
    eng.check_errors()
    # the id sort behind the segment sum: keys ascending, tokens a permutation that carries the ids,
    # equal keys in token order (stable)
    sk, st_ = db.sorted_ids[:db.nnz].long(), db.sorted_tok[:db.nnz].long()
    assert bool((sk[1:] >= sk[:-1]).all())
    assert torch.equal(torch.sort(st_).values, torch.arange(db.nnz, device="cuda"))
    assert torch.equal(ids[st_], sk)
    same = sk[1:] == sk[:-1]
    assert bool((st_[1:][same] > st_[:-1][same]).all())
    delta = (eng.table.double() - before.double())
    exp = -hp.learning_rate * contrib.sum(0)
    got = delta.sum(0)
    scale = (hp.learning_rate * contrib.abs().sum(0)).clamp_min(1e-30)
    assert float(((got - exp).abs() / scale).max()) < 1e-5
    lin_delta = (eng.lin_w.double() - lin_before.double()).sum()
    assert abs(float(lin_delta + hp.learning_rate * lin_contrib.sum())) < 1e-5 * float(
        hp.learning_rate * lin_contrib.abs().sum())
    # rows outside the batch are bit-identical
    touched = torch.zeros(N, dtype=torch.bool, device="cuda")
    touched[ids] = True
    changed = (delta != 0).any(1)
    assert not bool((changed & ~touched).any())
    assert int(changed.sum()) > 0.9 * int(touched.sum())  # (a zero gradient row is possible, rare)


def test_full_size_step_is_deterministic_and_graph_equals_eager(setup):
    hp, batches = setup
    engines = [XDeepFMEngine(hp, seed=5, embedding_update="sparse_rows") for _ in range(3)]
    a, b, c = engines
    losses = []
    for eng in (a, b):
        for hb in batches:
            losses.append(float(eng.train_step(eng.make_batch(*_dev(hb)))[0]))
    gs = c.capture_step(*_dev(batches[0]))  # runs batch 0 once while warming up
    l1 = float(gs.run(*_dev(batches[1]))[0])
    torch.cuda.synchronize()
    for eng in engines:
        eng.check_errors()
    assert losses[:2] == losses[2:] and l1 == losses[1]
    # rows are compared as the reference's dense optimizer would have left them (state_dict flushes the
    # steps rows still owe; the graph engine has caught some up ahead of time, row dates differ)
    sa, sb, sc = a.state_dict(), b.state_dict(), c.state_dict()
    for key in ("embedding", "lin_w", "slot/table_m", "slot/table_v", "slot/lin_m", "slot/lin_v"):
        assert torch.equal(sa[key], sb[key]) and torch.equal(sa[key], sc[key]), key
    assert torch.equal(a.dense, b.dense) and torch.equal(a.dense, c.dense)
    assert np.isfinite(losses).all()
