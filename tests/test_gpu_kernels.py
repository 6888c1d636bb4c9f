"""GPU parity of every C-ABI entry point against the CPU oracle (oracle/), on seeded ragged
inputs.  Bars (BASELINE.json north_star): embedding gather bit-exact; everything else fp32
within 1e-5 relative to the tensor's scale; integer work (AUC pair counts, sort plans) exact."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import xdeepfm_oracle as X
from recommenders_amd import _lib
from recommenders_amd.models.deeprec import deeprec_utils as U
from recommenders_amd.models.deeprec.models._engine import XDeepFMEngine, _ptr
from tests.util import make_hp, random_batch

pytestmark = pytest.mark.gpu

RTOL = 1e-5  # the fp32 tolerance north_star states


def close(got, exp, rtol=RTOL, what=""):
    got = np.asarray(got, dtype=np.float64)
    exp = np.asarray(exp, dtype=np.float64)
    assert got.shape == exp.shape, (what, got.shape, exp.shape)
    scale = max(float(np.abs(exp).max()), 1e-30)
    err = float(np.abs(got - exp).max())
    assert err <= rtol * scale, "%s: max err %.3e vs scale %.3e (rel %.2e)" % (what, err, scale, err / scale)


def load_engine(hp, params, mode="dense_exact"):
    eng = XDeepFMEngine(hp, embedding_update=mode)
    eng.load_state({k: v.detach().clone() for k, v in params.items()})
    return eng


def upload(eng, b):
    return eng.upload(b["labels"], b["dnn_feat_indices"], b["dnn_feat_values"], b["dnn_feat_weights"])


@pytest.mark.parametrize("dim", [10, 16, 7, 64])
def test_embedding_gather_bit_exact(dim):
    rng = np.random.default_rng(dim)
    hp = make_hp(dim=dim, FEATURE_COUNT=5000, FIELD_COUNT=39, use_CIN_part=False, use_DNN_part=False)
    B = 257
    b = random_batch(rng, hp, B, multi_p=0.2, missing_p=0.15, numeric_fields=13)
    p = X.init_params(hp, seed=1)
    eng = load_engine(hp, p)
    db = upload(eng, b)
    eng.forward(db, with_loss=False)
    torch.cuda.synchronize()
    eng.check_errors()
    got = eng.ws["emb_out"][:B].cpu().numpy().reshape(B * hp.FIELD_COUNT, dim)
    exp = X.embedding_lookup_sparse_np(p["embedding"].numpy(), b["dnn_feat_indices"], b["dnn_feat_values"],
                                       b["dnn_feat_weights"], B * hp.FIELD_COUNT)
    assert got.dtype == np.float32 and np.array_equal(got.view(np.uint32), exp.view(np.uint32))
    # linear and FM side outputs of the same kernel
    _, aux = X.forward({k: v.double() for k, v in p.items()}, b, hp)
    close(eng.ws["lin_out"][:B].cpu().numpy() + p["lin_b"].numpy(), aux["linear"].numpy()[:, 0], what="linear")
    close(eng.ws["fm_out"][:B].cpu().numpy(), aux["fm"].numpy()[:, 0], what="fm")


def test_embedding_gather_edge_cases():
    hp = make_hp(dim=8, FEATURE_COUNT=50, FIELD_COUNT=3, use_CIN_part=False, use_DNN_part=False,
                 use_FM_part=False, use_Linear_part=True)
    p = X.init_params(hp, seed=1)
    eng = load_engine(hp, p)
    # a batch whose last samples have no tokens at all (trailing empty rows -> zeros)
    from oracle import ffm_iterator_oracle as O
    b = O.convert([1.0, 0.0, 1.0], [[[0, 3, 1.0], [2, 49, 0.5]], [], []], 3, 50)
    db = upload(eng, b)
    eng.forward(db, with_loss=False)
    got = eng.ws["emb_out"][:3].cpu().numpy().reshape(9, 8)
    exp = X.embedding_lookup_sparse_np(p["embedding"].numpy(), b["dnn_feat_indices"], b["dnn_feat_values"],
                                       b["dnn_feat_weights"], 9)
    assert np.array_equal(got, exp) and not got[3:].any()
    eng.check_errors()
    # feature id out of range is reported, not read
    bad = O.convert([1.0], [[[0, 50, 1.0]]], 3, 50)
    eng.forward(upload(eng, bad), with_loss=False)
    with pytest.raises(ValueError, match="feature index"):
        eng.check_errors()
    # field out of range (row beyond B*F) is reported by the CSR builder
    bad = O.convert([1.0], [[[5, 1, 1.0]]], 3, 50)
    upload(eng, bad)
    with pytest.raises(ValueError, match="field index"):
        eng.check_errors()


CONFIGS = {
    "all_parts": dict(),
    "notebook_like": dict(FIELD_COUNT=39, dim=10, FEATURE_COUNT=3000, cross_layer_sizes=[20, 10],
                          layer_sizes=[20, 20], activation=["relu", "relu"], use_FM_part=False),
    "cin_only_odd": dict(FIELD_COUNT=5, dim=6, cross_layer_sizes=[10, 6, 7], use_Linear_part=False,
                         use_FM_part=False, use_DNN_part=False),
    "dnn_acts": dict(layer_sizes=[9, 33, 5], activation=["sigmoid", "elu", "identity"],
                     dropout=[0.0, 0.0, 0.0], use_CIN_part=False),
    "wide_dim": dict(FIELD_COUNT=6, dim=64, cross_layer_sizes=[34, 40], layer_sizes=[70], activation=["relu"]),
    # the bench's shapes (same kernel instantiations: 16-byte paths, 10 f tiles, 7 pair tiles per wave)
    "criteo_shape": dict(FIELD_COUNT=39, dim=16, FEATURE_COUNT=5000, cross_layer_sizes=[20, 10],
                         layer_sizes=[20, 20], activation=["relu", "relu"]),
    # _build_fast_CIN (fast_CIN_d > 0): oracle restated from xDeepFM.py:295-451, parity unpinned
    # (no fixture of the reference covers this branch and TensorFlow is absent)
    "fast_cin": dict(FIELD_COUNT=9, dim=6, fast_CIN_d=3, cross_layer_sizes=[10, 6, 5], layer_sizes=[8],
                     activation=["relu"], dropout=[0.0]),
    "fast_cin_criteo_shape": dict(FIELD_COUNT=39, dim=16, FEATURE_COUNT=5000, fast_CIN_d=4,
                                  cross_layer_sizes=[20, 10], layer_sizes=[20], activation=["relu"],
                                  dropout=[0.0]),
    # cross_activation between the CIN product and its split (xDeepFM.py:242, :402)
    "cross_act_cin": dict(cross_activation="sigmoid", FIELD_COUNT=9, dim=8, cross_layer_sizes=[12, 8, 6]),
    "cross_act_fast": dict(cross_activation="tanh", FIELD_COUNT=9, dim=6, fast_CIN_d=2,
                           cross_layer_sizes=[10, 6], layer_sizes=[8], activation=["relu"], dropout=[0.0]),
    # BASELINE configs[4] "wide" shapes (dim 64, CIN [256,256,256], DNN [1024,512]) at a small batch:
    # H = 128 maps do not fit the matrix-core CIN tiles, the generic kernels serve these layers
    "wide_shape": dict(FIELD_COUNT=39, dim=64, FEATURE_COUNT=4000, cross_layer_sizes=[256, 256, 256],
                       layer_sizes=[1024, 512], activation=["relu", "relu"], init_value=0.02),
    # softmax (base_model.py:343-344) as the CIN's cross_activation and as a DNN activation
    # (pooled softmax channels add up to logits near +12: cross_entropy_loss, which is computed from
    # the logit, instead of log_loss's log(1 - p + eps), which keeps four digits there in fp32)
    "softmax_acts": dict(cross_activation="softmax", FIELD_COUNT=9, dim=8, cross_layer_sizes=[12, 8, 6],
                         layer_sizes=[10, 7], activation=["softmax", "relu"], loss="cross_entropy_loss"),
    # more than 40 fields: the input-gradient kernel's first form and the wide staging of the filter gradient
    # dim 32: four 8-row stages per sample in the filter gradient's hidden | pooled cut
    "dim32": dict(FIELD_COUNT=6, dim=32, FEATURE_COUNT=400, cross_layer_sizes=[20, 6], layer_sizes=[12],
                  activation=["relu"], dropout=[0.0]),
    "many_fields": dict(FIELD_COUNT=45, dim=4, FEATURE_COUNT=2000, cross_layer_sizes=[8, 4],
                        layer_sizes=[16], activation=["relu"], dropout=[0.0]),
}


@pytest.mark.parametrize("name", list(CONFIGS))
@pytest.mark.parametrize("loss", ["log_loss", "cross_entropy_loss", "square_loss"])
def test_forward_and_gradients_match_oracle(name, loss):
    if loss != "log_loss" and name not in ("all_parts", "cin_only_odd"):
        pytest.skip("loss variants are covered on two configs")
    rng = np.random.default_rng(11)
    hp = make_hp(**{"loss": loss, **CONFIGS[name]})
    B = 97
    b = random_batch(rng, hp, B)
    p32 = X.init_params(hp, seed=7)
    for k in p32:
        if k.endswith("_b") or "_b_" in k:
            p32[k] = p32[k] + 0.05  # biases away from zero so their role is tested
    eng = load_engine(hp, p32)
    db = upload(eng, b)
    eng.forward(db, with_loss=True, with_grad=True)
    eng.backward(db)
    # table / lin_w gradients through the scatter mode of the sorted-key segment sum
    eng.plan(db)
    use_lin, use_fm = bool(hp.use_Linear_part), bool(hp.use_FM_part)
    opt = eng._opt_args()
    _lib.check(eng.lib.xdfm_embed_bwd(
        1, _ptr(eng.table), None, None, None, None, None, eng.table_stride, eng.lin_stride, _ptr(eng.grad_table),
        _ptr(eng.grad_lin) if use_lin else None, eng.N, eng.D, eng.F, db.B, _ptr(db.sorted_ids),
        _ptr(db.sorted_tok), _ptr(db.tok_row), _ptr(db.weights), db.nnz, _ptr(eng.ws["d_emb"]),
        _ptr(eng.ws["g_logit"]) if (use_lin or use_fm) else None,
        _ptr(eng.ws["fm_sum"]) if use_fm else None, 1 if use_fm else 0, ctypes.byref(opt),
        _ptr(eng.ws["pieces"]), _ptr(eng.d_emb2()), 0, None, eng._stream()))
    torch.cuda.synchronize()
    eng.check_errors()

    # oracle in float64: the yardstick both fp32 implementations are measured against
    p64 = {k: v.double().requires_grad_(True) for k, v in p32.items()}
    logit, aux = X.forward(p64, b, hp)
    pred = X.pred_from_logit(logit, hp)
    dl = X.data_loss(logit, pred, torch.as_tensor(b["labels"]).double(), hp)
    dl.backward()

    close(eng.ws["logit"][:B].cpu().numpy(), logit.detach().numpy()[:, 0], what="logit")
    close(eng.ws["pred"][:B].cpu().numpy(), pred.detach().numpy()[:, 0], what="pred")
    close(eng.data_loss.cpu().numpy()[0], dl.item(), what="data_loss")
    for k in eng.dense_shapes:
        close(eng.grad_views[k].cpu().numpy(), p64[k].grad.numpy().reshape(eng.dense_shapes[k]), what="d " + k)
    close(eng.grad_table.cpu().numpy(), p64["embedding"].grad.numpy(), what="d embedding")
    if use_lin:
        close(eng.grad_lin.cpu().numpy(), p64["lin_w"].grad.numpy()[:, 0], what="d lin_w")
    # also: the fp32 oracle is no closer to fp64 than the kernels are required to be
    l32, _ = X.forward(p32, b, hp)
    close(l32.numpy()[:, 0], logit.detach().numpy()[:, 0], rtol=5 * RTOL, what="oracle fp32 vs fp64")


@pytest.mark.parametrize("fast_d", [0, 2])
@pytest.mark.parametrize("optimizer", ["adam", "adagrad", "sgd", "rmsprop", "adadelta", "ftrl", "padagrad", "pgd"])
@pytest.mark.parametrize("mode", ["dense_exact", "sparse_rows", "sparse_lazy"])
def test_training_steps_match_oracle(optimizer, mode, fast_d):
    """dense_exact and sparse_rows (row updates + catch-up of the steps a row missed) against the
    oracle's reference semantics (every row moves every step); sparse_lazy against the oracle's
    restatement of lazy row updates."""
    if fast_d and (optimizer != "adam"):
        pytest.skip("the fast-CIN filters go through the same update path; one optimizer is enough")
    reg = dict(embed_l2=0.01, layer_l2=0.01, cross_l2=0.01, embed_l1=0.001, layer_l1=0.001, cross_l1=0.001)
    if mode != "dense_exact":
        reg = dict(cross_l2=0.01, cross_l1=0.001)
    reg["fast_CIN_d"] = fast_d
    hp = make_hp(optimizer=optimizer, learning_rate=0.01, FEATURE_COUNT=300, **reg)
    oracle_mode = "sparse_lazy" if mode == "sparse_lazy" else "dense_exact"
    rng = np.random.default_rng(5)
    p32 = X.init_params(hp, seed=9)
    eng = load_engine(hp, p32, mode)
    p64 = {k: v.double() for k, v in p32.items()}
    st = X.new_opt_state(p64, hp)
    for step in range(4):
        b = random_batch(rng, hp, 64 if step != 2 else 33)
        db = upload(eng, b)
        loss, dloss = eng.train_step(db)
        o_loss, o_dl, _, _ = X.train_step(p64, st, b, hp, mode=oracle_mode)
        close(float(dloss), o_dl, what="data loss step %d" % step)
        close(float(loss), o_loss, what="total loss step %d" % step)
    torch.cuda.synchronize()
    eng.check_errors()
    sd = eng.state_dict()
    # a few Adam steps amplify fp32 rounding of tiny gradients by up to 1/sqrt(v): allow 2e-4 there
    # (rmsprop / adadelta / ftrl divide by roots of small accumulators in the same way)
    tol = 2e-4 if optimizer in ("adam", "rmsprop", "adadelta") else 2e-5
    if optimizer == "ftrl":
        tol = 2e-3  # p = -linear / quadratic with `linear` a cancelling sum (g - sigma*p): fp32 vs the fp64 oracle
    for k, v in p64.items():
        close(sd[k].cpu().numpy().reshape(v.shape), v.numpy(), rtol=tol, what="param " + k)


@pytest.mark.parametrize("optimizer", ["adam", "rmsprop", "adadelta"])
def test_sparse_rows_with_long_gaps_equal_reference_dense_updates(optimizer):
    """Rows that stay out of the batches for many steps (beyond the 24 missed steps the catch-up
    applies one by one: the closed-form remainder) against the oracle's dense-exact optimizer, where
    every row moves every step; 2e-4 per parameter.  A replayed step graph with
    the next batch's sort running one step ahead must give the same bits as eager steps."""
    from oracle import ffm_iterator_oracle as O
    from tests.util import random_features

    # no cross_l2 here: its gradient W/||W|| makes Adam oscillate filters around 0 with amplitude ~lr,
    # which separates ANY fp32 run from the fp64 oracle over 130 steps (the dense path alike)
    hp = make_hp(optimizer=optimizer, learning_rate=0.005, FEATURE_COUNT=6000, batch_size=32)
    rng = np.random.default_rng(11)
    p32 = X.init_params(hp, seed=4)
    eng = load_engine(hp, p32, "sparse_rows")
    lazy = load_engine(hp, p32, "sparse_lazy")
    p64 = {k: v.double() for k, v in p32.items()}
    st = X.new_opt_state(p64, hp)
    steps = 130
    batches = []
    for k in range(steps):
        labels, feats = random_features(rng, 32, hp.FIELD_COUNT, hp.FEATURE_COUNT, multi_p=0.0, missing_p=0.0)
        # a handful of rows come back after long and short absences; 70 -> 129 is a gap of 58 steps late
        # enough in training for the closed-form remainder to carry 34 of them
        if k in (0, 3, 5, 30, 70, 129):
            for f, tok in enumerate(feats[0]):
                tok[1] = 10 + f
        batches.append(O.convert(labels, feats, hp.FIELD_COUNT, hp.FEATURE_COUNT))
    devb = [upload(eng, b) for b in batches]
    for k in range(steps):
        eng.train_step(devb[k], next_batch=devb[k + 1] if k + 1 < steps else None)
        lazy.train_step(upload(lazy, batches[k]))
        X.train_step(p64, st, batches[k], hp, mode="dense_exact")
    torch.cuda.synchronize()
    eng.check_errors()
    sd, sd_lazy = eng.state_dict(), lazy.state_dict()
    for k, v in p64.items():
        got = sd[k].cpu().numpy().reshape(v.shape)
        assert np.abs(got - v.numpy()).max() <= 2e-4, (k, np.abs(got - v.numpy()).max())
    if optimizer == "adam":  # the lazy update is measurably a different algorithm here
        d = np.abs(sd_lazy["embedding"].cpu().numpy() - p64["embedding"].numpy()).max()
        assert d > 2e-3, d
    # scoring after training reads flushed rows: predictions against the oracle's
    pred = eng.forward(devb[-1], with_loss=False).cpu().numpy()
    close(pred, X.predict(p64, batches[-1], hp).numpy(), rtol=2e-3, what="pred after training")


def test_graphed_sparse_rows_step_equals_eager():
    """A replayed step graph (its id sort runs inside the graph, beside the forward pass; rows are aged
    where they are read) gives the same bits as eager steps: through run_packed(pack(...))."""
    hp = make_hp(optimizer="adam", learning_rate=0.02, FEATURE_COUNT=2000, batch_size=48)
    rng = np.random.default_rng(12)
    p32 = X.init_params(hp, seed=6)
    batches = [random_batch(rng, hp, 48, multi_p=0.0, missing_p=0.0) for _ in range(9)]

    def dev(b):
        return [torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in (
            b["labels"].reshape(-1), b["dnn_feat_indices"], b["dnn_feat_values"], b["dnn_feat_weights"])]

    eager, graph = load_engine(hp, p32, "sparse_rows"), load_engine(hp, p32, "sparse_rows")
    eager.train_step(upload(eager, batches[0]))
    gs = graph.capture_step(*dev(batches[0]))  # runs batch 0 once while warming up
    devs = [dev(b) for b in batches]
    packed = [gs.pack(*devs[k]) for k in range(9)]
    for k in range(1, 9):
        le = float(eager.train_step(upload(eager, batches[k]))[0])
        lg = float(gs.run_packed(packed[k])[0])
        assert le == lg, (k, le, lg)
    sa, sb = eager.state_dict(), graph.state_dict()
    assert sa["step"] == sb["step"] == 9
    for k in sa:
        if torch.is_tensor(sa[k]):
            assert torch.equal(sa[k], sb[k]), k


def test_graphed_step_through_run_and_run_batch_equals_eager():
    """The other two ways into a captured step: run() with four arrays and pack_batch() / run_batch() from
    DeviceBatches (what fit() uses; the identities and successor arguments of earlier versions are accepted
    and ignored).  Same bits as eager steps."""
    hp = make_hp(optimizer="adam", learning_rate=0.02, FEATURE_COUNT=2000, batch_size=48)
    rng = np.random.default_rng(13)
    p32 = X.init_params(hp, seed=8)
    batches = [random_batch(rng, hp, 48, multi_p=0.0, missing_p=0.0) for _ in range(9)]

    def dev(b):
        return [torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in (
            b["labels"].reshape(-1), b["dnn_feat_indices"], b["dnn_feat_values"], b["dnn_feat_weights"])]

    eager, graph = load_engine(hp, p32, "sparse_rows"), load_engine(hp, p32, "sparse_rows")
    eager.train_step(upload(eager, batches[0]))
    devs = [dev(b) for b in batches]
    dbs = [upload(graph, b) for b in batches]
    from recommenders_amd.models.deeprec.models._engine import GraphedStep

    gs = GraphedStep(graph, dbs[0].labels, None, dbs[0].ids, dbs[0].weights, csr=(dbs[0].row_ptr, dbs[0].tok_row))
    keys = [object() for _ in batches]
    for k in range(1, 9):
        le = float(eager.train_step(upload(eager, batches[k]))[0])
        if k < 5:  # four arrays
            lg = float(gs.run(*devs[k], next_ids=devs[k + 1][2] if k % 2 else None)[0])
        else:      # device batches
            pb = gs.pack_batch(dbs[k])
            lg = float(gs.run_batch(pb, keys[k], keys[k + 1] if k + 1 < 9 else None)[0])
        assert le == lg, (k, le, lg)
    sa, sb = eager.state_dict(), graph.state_dict()
    assert sa["step"] == sb["step"] == 9
    for k in sa:
        if torch.is_tensor(sa[k]):
            assert torch.equal(sa[k], sb[k]), k


def test_folded_step_for_fewer_hardware_queues_replays_as_a_chain(monkeypatch):
    """GPU_MAX_HW_QUEUES below 4 (hipGraphLaunch of the branched step died inside the runtime at 2, round 2):
    the engine folds its step onto one stream, the capture is a chain.  Here the variable is set after
    the runtime came up (it changes what the engine does, not the runtime): the folded capture must replay
    to the bits of folded eager steps, and to fp32 rounding of the branched engine."""
    hp = make_hp(optimizer="adam", learning_rate=0.02, FEATURE_COUNT=2000, batch_size=48)
    rng = np.random.default_rng(14)
    p32 = X.init_params(hp, seed=9)
    batches = [random_batch(rng, hp, 48, multi_p=0.0, missing_p=0.0) for _ in range(6)]

    def dev(b):
        return [torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in (
            b["labels"].reshape(-1), b["dnn_feat_indices"], b["dnn_feat_values"], b["dnn_feat_weights"])]

    branched = load_engine(hp, p32, "sparse_rows")
    monkeypatch.setenv("GPU_MAX_HW_QUEUES", "2")
    eager, graph = load_engine(hp, p32, "sparse_rows"), load_engine(hp, p32, "sparse_rows")
    assert eager.folded and graph.folded and not branched.folded
    assert not graph.overlap_dnn and graph.serial_plan and not graph.split_dw
    for eng in (branched, eager):
        eng.train_step(upload(eng, batches[0]))
    gs = graph.capture_step(*dev(batches[0]))
    for k in range(1, 6):
        lb = float(branched.train_step(upload(branched, batches[k]))[0])
        le = float(eager.train_step(upload(eager, batches[k]))[0])
        lg = float(gs.run(*dev(batches[k]))[0])
        assert le == lg, (k, le, lg)
        assert abs(le - lb) <= 1e-5 * abs(lb), (k, le, lb)
    sa, sb, sc = eager.state_dict(), graph.state_dict(), branched.state_dict()
    for k in sa:
        if torch.is_tensor(sa[k]):
            assert torch.equal(sa[k], sb[k]), k
            close(sa[k].cpu().numpy(), sc[k].cpu().numpy(), rtol=2e-4, what="folded vs branched " + k)


def _dense_adam_absent(p, m, v, s, G, lr, b1=float(np.float32(0.9)), b2=float(np.float32(0.999)),
                       eps=float(np.float32(1e-8))):
    """TF1 Adam on a row with a zero gradient for steps s+1 .. s+G (float64, step by step, with the
    fp32 constants TF's kernel holds): base_model.py:290 + the dense table gradient of SURVEY.md 8a A2."""
    p, m, v = (np.asarray(x, dtype=np.float64).copy() for x in (p, m, v))
    for j in range(s + 1, s + G + 1):
        m *= b1
        v *= b2
        p -= lr * np.sqrt(1 - b2 ** j) / (1 - b1 ** j) * m / (np.sqrt(v) + eps)
    return p, m, v


@pytest.mark.parametrize("lr", [0.002, 0.05])
def test_row_ageing_equals_dense_adam_step_by_step(lr):
    """Kernel level: rows that are G steps behind, read through xdfm_embed_gather with an xdfm_row_age
    (aged in registers, nothing written) and written back by xdfm_embed_catchup, against Adam's dense update
    applied step by step in float64 -- gaps from 1 to thousands of steps, rows last touched in the first steps
    of training (where the bias corrections bend) and late, second moments from 1e-4 down to below eps^2, at the
    notebook's learning rate and at one 25 times larger.  A catch-up split in two (20 steps, then 10) must land
    where one of 30 steps does."""
    D, N = 16, 4096
    hp = make_hp(optimizer="adam", learning_rate=lr, FEATURE_COUNT=N, dim=D, use_CIN_part=False, use_DNN_part=False,
                 use_FM_part=False)
    rng = np.random.default_rng(3)
    cases = [(s, G) for s in (1, 3, 50, 63, 500, 16383, 16384, 40000)
             for G in (1, 5, 24, 30, 100, 255, 256, 300, 2000)]
    rows = np.arange(len(cases)) * 7 + 5
    for t_now_extra in (0,):
        for s_fixed in sorted({s + G for s, G in cases}):
            idx = [i for i, (s, G) in enumerate(cases) if s + G == s_fixed]
            eng = XDeepFMEngine(hp, embedding_update="sparse_rows")
            eng.load_state({"step": s_fixed}, strict=False)  # all rows current for s_fixed
            p0 = rng.normal(0, 0.1, (len(idx), D)).astype(np.float32)
            mag = 10.0 ** rng.uniform(-9, -2, (len(idx), D))
            m0 = (rng.choice([-1.0, 1.0], (len(idx), D)) * mag * rng.uniform(0.05, 1.0, (len(idx), D))).astype(np.float32)
            v0 = (mag * mag * rng.uniform(0.001, 1.0, (len(idx), D))).astype(np.float32)
            lw0 = rng.normal(0, 0.1, len(idx)).astype(np.float32)
            lm0, lv0 = m0[:, 0].copy(), v0[:, 0].copy()
            r = torch.as_tensor(rows[idx]).cuda()
            eng.table[r] = torch.as_tensor(p0).cuda()
            eng.table_m[r] = torch.as_tensor(m0).cuda()
            eng.table_v[r] = torch.as_tensor(v0).cuda()
            eng.lin_w[r] = torch.as_tensor(lw0).cuda()
            eng.lin_m[r] = torch.as_tensor(lm0).cuda()
            eng.lin_v[r] = torch.as_tensor(lv0).cuda()
            eng.row_step[r] = torch.as_tensor([cases[i][0] for i in idx], dtype=torch.int32).cuda()
            before = eng._records.clone()
            out = torch.zeros((len(idx), D), device="cuda")
            lout = torch.zeros((len(idx),), device="cuda")
            _lib.check(eng.lib.xdfm_embed_gather(_ptr(eng.table), _ptr(eng.lin_w), eng.table_stride, eng.lin_stride,
                                                 eng.N, D, _ptr(r), len(idx), _ptr(out), _ptr(lout), 0,
                                                 _ptr(eng.err_flag), eng._age(0), eng._stream()))
            torch.cuda.synchronize()
            assert torch.equal(before, eng._records)  # the gather writes nothing
            eng._rows_stale = True
            eng.flush_rows()
            torch.cuda.synchronize()
            for n, i in enumerate(idx):
                s, G = cases[i]
                pe, me, ve = _dense_adam_absent(p0[n], m0[n], v0[n], s, G, lr)
                le, _, _ = _dense_adam_absent(lw0[n], lm0[n], lv0[n], s, G, lr)
                drift = np.abs(pe - p0[n]).max()
                tol = 3e-7 * max(1.0, np.abs(pe).max()) + 2e-5 * drift
                got = out[n].cpu().numpy()
                assert np.abs(got - pe).max() <= tol, (lr, s, G, np.abs(got - pe).max(), drift)
                assert abs(float(lout[n]) - le) <= tol, (lr, s, G, "lin")
                # written back: the same bits as the gather formed, slots decayed, row dated
                assert np.array_equal(eng.table[rows[i]].cpu().numpy(), got)
                np.testing.assert_allclose(eng.table_m[rows[i]].cpu().numpy(), me, rtol=2e-5, atol=1e-37)
                np.testing.assert_allclose(eng.table_v[rows[i]].cpu().numpy(), ve, rtol=2e-5, atol=1e-37)
                assert int(eng.row_step[rows[i]]) == s_fixed
            del eng
    # split invariance: 30 steps at once against 20 + 10
    one = XDeepFMEngine(hp, embedding_update="sparse_rows")
    two = XDeepFMEngine(hp, embedding_update="sparse_rows")
    s = 700
    p0 = rng.normal(0, 0.1, (8, D)).astype(np.float32)
    m0 = (10.0 ** rng.uniform(-7, -3, (8, D))).astype(np.float32)
    v0 = (m0 * m0 * 0.3).astype(np.float32)
    r = torch.arange(8).cuda() * 3
    for eng, stops in ((one, (s + 30,)), (two, (s + 20, s + 30))):
        eng.load_state({"step": s}, strict=False)
        eng.table[r], eng.table_m[r], eng.table_v[r] = (torch.as_tensor(x).cuda() for x in (p0, m0, v0))
        for t in stops:
            eng.opt_state.copy_(torch.tensor([0.9 ** t, 0.999 ** t, 0.0, float(t)]))
            eng._init_drift_tab()
            eng._rows_stale = True
            eng.flush_rows()
    torch.cuda.synchronize()
    d = (one.table[r] - two.table[r]).abs().max().item()
    moved = (one.table[r].cpu() - torch.as_tensor(p0)).abs().max().item()
    assert d <= 2e-7 + 2e-6 * moved, (d, moved)


@pytest.mark.parametrize("name", ["all_parts", "criteo_shape", "wide_shape"])
def test_pooled_channels_summed_over_d_first_equal_the_layer_kernels(name):
    """The pooled channels of a CIN layer through xdfm_cin_pair_sum + a plain product (the sum over the
    embedding dimension taken first) against the layer kernels carrying every channel: another
    association of the same fp32 sums, 1e-5 apart at most."""
    hp = make_hp(**CONFIGS[name])
    rng = np.random.default_rng(4)
    b = random_batch(rng, hp, 70)
    p32 = X.init_params(hp, seed=3)
    outs = []
    for split in (True, False):
        eng = load_engine(hp, p32)
        eng.pool_split = split
        db = upload(eng, b)
        eng.forward(db, with_loss=True, with_grad=True)
        eng.backward(db)
        torch.cuda.synchronize()
        assert bool(eng._split) == split
        d_emb = eng.ws["d_emb"][:70].clone()
        if eng.d_emb2() is not None:
            d_emb = d_emb + eng.d_emb2()[:70]
        outs.append((eng.ws["logit"][:70].clone(), eng.dense_grad.clone(), d_emb))
    for x, y, what in zip(outs[0], outs[1], ("logit", "dense gradients", "d emb_out")):
        close(x.cpu().numpy(), y.cpu().numpy(), rtol=2e-5, what=what)


@pytest.mark.parametrize("name", ["all_parts", "criteo_shape", "many_fields"])
def test_fused_last_layer_equals_the_layer_kernels(name):
    """A last CIN layer (every channel pooled) through xdfm_cin_pooled_fwd / xdfm_cin_pooled_bwd (sum over
    the embedding dimension first, f32 matrix cores) against the layer kernels: another association of
    the same fp32 sums, 2e-5 apart at most."""
    hp = make_hp(**CONFIGS[name])
    rng = np.random.default_rng(5)
    b = random_batch(rng, hp, 70)
    p32 = X.init_params(hp, seed=6)
    outs = []
    for fused in (True, False):
        eng = load_engine(hp, p32)
        eng.pooled_layer = fused
        db = upload(eng, b)
        eng.forward(db, with_loss=True, with_grad=True)
        eng.backward(db)
        torch.cuda.synchronize()
        assert bool(eng._fused_pool) == fused, name
        d_emb = eng.ws["d_emb"][:70].clone()
        if eng.d_emb2() is not None:
            d_emb = d_emb + eng.d_emb2()[:70]
        outs.append((eng.ws["logit"][:70].clone(), eng.dense_grad.clone(), d_emb))
    for x, y, what in zip(outs[0], outs[1], ("logit", "dense gradients", "d emb_out")):
        close(x.cpu().numpy(), y.cpu().numpy(), rtol=2e-5, what=what)


@pytest.mark.parametrize("B,F,H,D,L,acc", [(5, 6, 6, 8, 7, 0), (37, 7, 5, 12, 9, 1), (100, 39, 10, 16, 10, 0),
                                           (70, 12, 12, 8, 16, 1), (33, 64, 16, 4, 8, 1), (1000, 33, 16, 16, 12, 0),
                                           (0, 8, 8, 8, 8, 0)])
def test_cin_pooled_entry_points_match_float64(B, F, H, D, L, acc):
    """xdfm_cin_pooled_fwd / _bwd through the C-ABI against the same sums in float64 (ragged tiles:
    fields, maps, columns and channels short of the 16 x 16 x 16 x L' the kernels compute on; a strided
    pooled array whose other columns must stay untouched; d_x0 written or added to)."""
    lib = _lib.lib
    g = torch.Generator(device="cuda").manual_seed(1)
    x0 = torch.randn(B, F, D, device="cuda", generator=g)
    xk = torch.randn(B, H, D, device="cuda", generator=g)
    # the filter (and its gradient) as the leading columns of a wider array when B is odd: row stride ldw > L
    ldw = L + 3 if B % 2 else 0
    w_full = torch.randn(F * H, ldw or L, device="cuda", generator=g) * 0.1
    w = w_full[:, :L]
    off, stride = 3, L + 5
    pooled = torch.full((B, stride), 7.0, device="cuda")
    d_pooled = torch.randn(B, stride, device="cuda", generator=g)
    n = ctypes.c_size_t()
    _lib.check(lib.xdfm_cin_pooled_workspace_floats(B, F, H, D, L, ctypes.byref(n)))
    assert n.value > 0
    work = torch.empty(n.value, device="cuda")
    _lib.check(lib.xdfm_cin_pooled_fwd(_ptr(x0), _ptr(xk), _ptr(w_full), B, F, H, D, L, ldw, _ptr(pooled), stride, off,
                                       None))
    base = torch.randn(B, F, D, device="cuda", generator=g)
    d_x0 = base.clone() if acc else torch.full((B, F, D), 9.0, device="cuda")
    d_xk = torch.full((B, H, D), 9.0, device="cuda")
    d_w_full = torch.full((F * H, ldw or L), 5.0, device="cuda")
    d_w = d_w_full[:, :L]
    _lib.check(lib.xdfm_cin_pooled_bwd(_ptr(x0), _ptr(xk), _ptr(w_full), _ptr(d_pooled), stride, off, B, F, H, D, L, ldw,
                                       _ptr(d_xk), _ptr(d_x0), acc, _ptr(d_w_full), _ptr(work), None))
    torch.cuda.synchronize()
    if B == 0:
        assert bool((d_w == 0).all())
        return
    a = x0.double().requires_grad_(True)
    c = xk.double().requires_grad_(True)
    W = w.double().reshape(F, H, L).clone().requires_grad_(True)
    y = torch.einsum("bfd,bhd,fhn->bn", a, c, W)
    y.backward(d_pooled[:, off:off + L].double())

    def rel(got, exp):
        return float((got.double() - exp).abs().max() / exp.abs().max())

    assert rel(pooled[:, off:off + L], y.detach()) < 2e-6
    assert bool((pooled[:, :off] == 7).all()) and bool((pooled[:, off + L:] == 7).all())
    assert rel(d_x0, a.grad + (base.double() if acc else 0)) < 2e-6
    assert rel(d_xk, c.grad) < 2e-6
    assert rel(d_w.reshape(F, H, L), W.grad) < 2e-6
    assert bool((d_w_full[:, L:] == 5).all())  # columns past L belong to somebody else


def test_cin_pooled_refuses_what_it_does_not_serve():
    lib = _lib.lib
    n = ctypes.c_size_t(1)
    for shape in [(8, 65, 8, 8, 8), (8, 8, 17, 8, 8), (8, 8, 8, 10, 8), (8, 8, 8, 20, 8), (8, 8, 8, 8, 17),
                  (8, 64, 16, 16, 16)]:
        _lib.check(lib.xdfm_cin_pooled_workspace_floats(*shape, ctypes.byref(n)))
        assert n.value == 0, shape
    x = torch.zeros(8 * 65 * 8, device="cuda")
    with pytest.raises(_lib.XdfmError):
        _lib.check(lib.xdfm_cin_pooled_fwd(_ptr(x), _ptr(x), _ptr(x), 8, 65, 8, 8, 8, 0, _ptr(x), 8, 0, None))


def test_dropout_training_matches_oracle_given_the_masks():
    """user_dropout (base_model.py:335-336, :377): the masks come from the engine's own generator
    (TF's stream is not reproducible); with the same masks the oracle must take the same steps.
    Eval / infer feed keep = 1 (base_model.py:400, :414)."""
    hp = make_hp(user_dropout=True, dropout=[0.3, 0.5], layer_sizes=[32, 16], activation=["relu", "tanh"],
                 optimizer="adam", learning_rate=0.01, FEATURE_COUNT=300, layer_l2=0.01)
    rng = np.random.default_rng(8)
    p32 = X.init_params(hp, seed=4)
    eng = load_engine(hp, p32)
    p64 = {k: v.double() for k, v in p32.items()}
    st = X.new_opt_state(p64, hp)
    seen = []
    for step in range(3):
        B = 256 if step != 1 else 77
        b = random_batch(rng, hp, B)
        loss, dloss = eng.train_step(upload(eng, b))
        masks = [eng.ws["drop_%d" % k][:B].cpu().numpy().astype(np.float64) for k in range(2)]
        for k, m in enumerate(masks):
            keep = 1.0 - hp.dropout[k]
            assert set(np.unique(m.astype(np.float32))) <= {np.float32(0.0), np.float32(1.0 / keep)}
            assert abs((m != 0).mean() - keep) < 0.05
        seen.append(masks[0][:77] != 0)
        o_loss, o_dl, _, _ = X.train_step(p64, st, b, hp, dropout_masks=masks)
        close(float(dloss), o_dl, what="data loss step %d" % step)
        close(float(loss), o_loss, what="total loss step %d" % step)
    assert (seen[0] != seen[1]).any() and (seen[1] != seen[2]).any()  # a new mask every step
    sd = eng.state_dict()
    for k, v in p64.items():
        close(sd[k].cpu().numpy().reshape(v.shape), v.numpy(), rtol=2e-4, what="param " + k)
    # scoring: no dropout
    b = random_batch(rng, hp, 50)
    eng.forward(upload(eng, b), with_loss=False, with_grad=False)
    logit, _ = X.forward(p64, b, hp)
    close(eng.ws["logit"][:50].cpu().numpy(), logit.numpy()[:, 0], rtol=2e-4, what="eval logit")


@pytest.mark.parametrize("name", ["all_parts", "criteo_shape", "cross_act_cin"])
def test_filters_packed_ahead_change_no_bit(name):
    """xdfm_cin_pack + xdfm_cin_fwd_packed / xdfm_cin_bwd_packed against the per-call packing of
    xdfm_cin_fwd / xdfm_cin_bwd: same kernels on the same images, so equality is exact."""
    hp = make_hp(**CONFIGS[name])
    rng = np.random.default_rng(3)
    b = random_batch(rng, hp, 70)
    p32 = X.init_params(hp, seed=2)
    outs = []
    for prepack in (True, False):
        eng = load_engine(hp, p32)
        eng.prepack = prepack
        eng.pool_split = False  # (the pooled channels' own path needs the packed images: off on both sides)
        db = upload(eng, b)
        eng.forward(db, with_loss=True, with_grad=True)
        eng.backward(db)
        torch.cuda.synchronize()
        assert any(k.startswith("img_f_") for k in eng.ws) == prepack
        # the gradient w.r.t. emb_out: one array when the last CIN layer folded the DNN tower's part in
        # (packed path), two summands otherwise -- the same association either way
        d_emb = eng.ws["d_emb"][:70].clone()
        if eng.d_emb2() is not None:
            d_emb = d_emb + eng.d_emb2()[:70]
        outs.append((eng.ws["logit"][:70].clone(), eng.dense_grad.clone(), d_emb))
    for x, y in zip(*outs):
        assert torch.equal(x, y)


@pytest.mark.parametrize("inner", [1, 8])
def test_batch_norm_kernels_match_torch(inner):
    """xdfm_bn_fwd / xdfm_bn_bwd against torch's batch_norm in float64 (training statistics, moving
    averages with momentum 0.95 and the biased variance, inference with the moving statistics)."""
    import torch.nn.functional as Fn

    B, C = 333, 12
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, C, inner, generator=g) * 2 + 0.5
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    mm, mv = torch.randn(C, generator=g), torch.rand(C, generator=g) + 0.5
    dy = torch.randn(B, C, inner, generator=g)
    lib = _lib.lib
    dev = "cuda"
    xd, gd, bd, mmd, mvd, dyd = (t.to(dev).contiguous() for t in (x, gamma, beta, mm.clone(), mv.clone(), dy))
    stats = torch.zeros(2 * C, device=dev)
    y = torch.empty_like(xd)
    _lib.check(lib.xdfm_bn_fwd(_ptr(xd), B, C, inner, _ptr(gd), _ptr(bd), _ptr(mmd), _ptr(mvd), 0.95, 1e-4, 1, 3,
                               _ptr(stats), _ptr(y), None))
    dx, dg, db = torch.empty_like(xd), torch.empty(C, device=dev), torch.empty(C, device=dev)
    _lib.check(lib.xdfm_bn_bwd(_ptr(xd), _ptr(y), _ptr(dyd), B, C, inner, _ptr(gd), _ptr(stats), 3, _ptr(dx),
                               _ptr(dg), _ptr(db), None))
    torch.cuda.synchronize()
    x64 = x.double().requires_grad_(True)
    g64, b64 = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    rm, rv = mm.double().clone(), mv.double().clone()
    ref = torch.tanh(Fn.batch_norm(x64, None, None, g64, b64, True, 0.0, 1e-4))  # act 3 = XDFM_ACT_TANH
    ref.backward(dy.double())
    mean = x.double().mean((0, 2))
    var = x.double().var((0, 2), unbiased=False)
    rm -= (rm - mean) * 0.05
    rv -= (rv - var) * 0.05
    close(y.cpu().numpy(), ref.detach().numpy(), what="bn y")
    close(dx.cpu().numpy(), x64.grad.numpy(), rtol=5e-5, what="bn dx")
    close(dg.cpu().numpy(), g64.grad.numpy(), rtol=5e-5, what="bn dgamma")
    close(db.cpu().numpy(), b64.grad.numpy(), rtol=5e-5, what="bn dbeta")
    close(mmd.cpu().numpy(), rm.numpy(), what="moving mean")
    close(mvd.cpu().numpy(), rv.numpy(), what="moving variance")
    # inference: the (just moved) moving statistics, which stay as they are
    keep = mmd.clone()
    _lib.check(lib.xdfm_bn_fwd(_ptr(xd), B, C, inner, _ptr(gd), _ptr(bd), _ptr(mmd), _ptr(mvd), 0.95, 1e-4, 0, 0,
                               _ptr(stats), _ptr(y), None))
    torch.cuda.synchronize()
    ref = Fn.batch_norm(x.double(), rm, rv, gamma.double(), beta.double(), False, 0.0, 1e-4)
    close(y.cpu().numpy(), ref.numpy(), what="bn inference")
    assert torch.equal(keep, mmd)


@pytest.mark.parametrize("variant", ["plain", "fast_cin", "cross_act_dropout"])
def test_batch_norm_training_matches_oracle(variant):
    """enable_BN (xDeepFM.py:234-240, :394-400, :497-503): three steps on batch statistics, the
    moving statistics, then scoring on the moving statistics.  Parity unpinned (the oracle
    restates tf.layers.batch_normalization from its documented non-fused behaviour)."""
    # cross_entropy_loss: normalised activations drive the logits to +-10, where log_loss's
    # log(1 - p + eps) (base_model.py:251-257) loses three digits in fp32 on either side
    kw = dict(enable_BN=True, optimizer="adam", learning_rate=0.01, FEATURE_COUNT=300, layer_l2=0.01,
              cross_l2=0.01, loss="cross_entropy_loss")
    if variant == "fast_cin":
        kw["fast_CIN_d"] = 2
    if variant == "cross_act_dropout":
        kw.update(cross_activation="tanh", user_dropout=True, dropout=[0.2, 0.3])
    hp = make_hp(**kw)
    rng = np.random.default_rng(15)
    p32 = X.init_params(hp, seed=6)
    for k in p32:
        if k.startswith("bn_"):
            p32[k] = p32[k] + 0.1 * torch.randn(p32[k].shape, generator=torch.Generator().manual_seed(1))
    eng = load_engine(hp, p32)
    p64 = {k: v.double() for k, v in p32.items()}
    st = X.new_opt_state(p64, hp)
    n_layers = len(hp.layer_sizes)
    for step in range(3):
        B = 128 if step != 1 else 77
        b = random_batch(rng, hp, B)
        loss, dloss = eng.train_step(upload(eng, b))
        masks = None
        if hp.user_dropout:
            masks = [eng.ws["drop_%d" % k][:B].cpu().numpy().astype(np.float64) for k in range(n_layers)]
        o_loss, o_dl, _, _ = X.train_step(p64, st, b, hp, dropout_masks=masks)
        close(float(dloss), o_dl, rtol=2e-5, what="data loss step %d" % step)
        close(float(loss), o_loss, rtol=2e-5, what="total loss step %d" % step)
    sd = eng.state_dict()
    for k, v in p64.items():
        if (k.startswith("dnn_b_") and k != "dnn_b_out") or k.startswith("bnstat_dnn_mm_"):
            # a bias in front of a batch norm has a zero gradient: Adam turns its rounding noise
            # (1e-17 here, 1e-9 there) into +-lr steps that the normalisation removes again (the
            # moving mean of that layer's input follows the bias)
            continue
        close(sd[k].cpu().numpy().reshape(v.shape), v.numpy(), rtol=5e-4, what="param " + k)
    # scoring on the moving statistics, from the engine's own state (the noise-walked biases are
    # not normalised away here: the moving mean follows them at 5% per step only)
    b = random_batch(rng, hp, 50)
    eng.forward(upload(eng, b), with_loss=False, with_grad=False)
    own = {k: sd[k].detach().cpu().double().reshape(v.shape) for k, v in p64.items()}
    logit, _ = X.forward(own, b, hp)
    close(eng.ws["logit"][:50].cpu().numpy(), logit.numpy()[:, 0], what="eval logit")
    after = eng.state_dict()
    assert all(torch.equal(after[k], sd[k]) for k in after if k.startswith("bnstat_"))  # scoring moves nothing


def test_clip_by_norm_matches_oracle():
    hp = make_hp(is_clip_norm=1, max_grad_norm=0.05, optimizer="sgd", learning_rate=0.1, FEATURE_COUNT=200,
                 embed_l2=0.01)
    rng = np.random.default_rng(6)
    p32 = X.init_params(hp, seed=3)
    eng = load_engine(hp, p32)
    p64 = {k: v.double() for k, v in p32.items()}
    st = X.new_opt_state(p64, hp)
    b = random_batch(rng, hp, 64)
    eng.train_step(upload(eng, b))
    X.train_step(p64, st, b, hp)
    sd = eng.state_dict()
    for k, v in p64.items():
        close(sd[k].cpu().numpy().reshape(v.shape), v.numpy(), rtol=2e-5, what="param " + k)


def test_backward_is_deterministic_with_heavy_duplicates():
    """13 'numeric' fields collapse to 13 hot rows: thousands of tokens per row.  Two runs must
    be bit identical (no float atomics) and equal the oracle."""
    hp = make_hp(FIELD_COUNT=39, dim=16, FEATURE_COUNT=100000, use_CIN_part=False, use_FM_part=False,
                 layer_sizes=[8], activation=["relu"], optimizer="adam")
    rng = np.random.default_rng(8)
    from recommenders_amd.synth import criteo_shape_ids
    from oracle import ffm_iterator_oracle as O
    B = 2048
    ids = criteo_shape_ids(rng, B, hp.FEATURE_COUNT)
    vals = np.where(np.arange(39)[None, :] < 13, np.round(rng.random((B, 39)), 3), 1.0)
    feats = [[[f, int(ids[i, f]), float(vals[i, f])] for f in range(39)] for i in range(B)]
    b = O.convert(list(rng.integers(0, 2, B).astype(float)), feats, 39, hp.FEATURE_COUNT)
    p32 = X.init_params(hp, seed=2)
    outs = []
    for _ in range(2):
        eng = load_engine(hp, p32, "sparse_rows")
        eng.train_step(upload(eng, b))
        torch.cuda.synchronize()
        outs.append((eng.table.cpu().numpy().copy(), eng.lin_w.cpu().numpy().copy()))
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    p64 = {k: v.double() for k, v in p32.items()}
    X.train_step(p64, X.new_opt_state(p64, hp), b, hp, mode="sparse_rows")
    close(outs[0][0], p64["embedding"].numpy(), rtol=2e-4, what="table after 1 sparse Adam step")


def test_eval_auc_logloss_matches_cal_metric(golden):
    hp = make_hp(use_CIN_part=False, use_DNN_part=False, use_FM_part=False)
    eng = XDeepFMEngine(hp)
    cases = [c for c in golden("cal_metric_golden.json") if c["metrics"][0] == "auc"]
    rng = np.random.default_rng(3)
    n = 200000
    y = (rng.random(n) < 0.3).astype(np.float32)
    p = np.round(rng.random(n), 3).astype(np.float32)  # many ties
    p[:10] = [0.0, 1.0, 0.0, 1.0, 0.5, 0.5, 1e-12, 1 - 1e-7, 0.25, 0.75]
    cases.append({"labels": y.tolist(), "preds": p.tolist(), "expected": None})
    # edge cases of the tie handling: every score equal; two samples; negative and signed-zero scores
    # (regression outputs) whose order the 32-bit keys must keep; a file larger than one sort tile
    cases.append({"labels": [1, 0, 1, 0, 0], "preds": [0.5] * 5, "expected": None})
    cases.append({"labels": [0, 1], "preds": [0.9, 0.1], "expected": None})
    cases.append({"labels": [0, 1, 1, 0, 1, 0], "preds": [-0.0, 0.0, -1.5, -2.5, 3.0, 2.0], "expected": None})
    big = 1500000
    cases.append({"labels": (rng.random(big) < 0.25).astype(np.float32).tolist(),
                  "preds": rng.random(big).astype(np.float32).tolist(), "expected": None})
    for c in cases:
        labels = np.asarray(c["labels"], dtype=np.float32)
        preds = np.asarray(c["preds"], dtype=np.float32)
        auc, ll, n_pos = eng.eval_metrics(torch.from_numpy(preds).cuda(), torch.from_numpy(labels).cuda())
        ref = U.cal_metric(list(labels), list(preds), ["auc", "logloss"])
        assert n_pos == float(labels.sum())
        assert round(auc, 4) == ref["auc"]
        # (scores outside (0, 1) are clipped on both sides alike; the rounded values can differ in the last
        # digit only when the sum sits on a rounding boundary)
        assert abs(ll - U._binary_logloss(labels, np.clip(preds.astype(np.float64), 10e-12, 1 - 10e-12))) < 1e-9
        assert round(ll, 4) == ref["logloss"]
        if c["expected"]:
            assert round(auc, 4) == c["expected"]["auc"] and round(ll, 4) == c["expected"]["logloss"]
        # unrounded: the AUC is an exact integer ratio
        exact = U._binary_auc(labels, preds)
        assert abs(auc - exact) < 1e-15


@pytest.mark.parametrize("dropout", [False, True, "bn"])
def test_graph_replay_equals_eager_steps(dropout):
    """A step captured into a HIP graph (device-side Adam scalars, side-stream sort joined
    inside the capture) must reproduce the eager launches bit for bit.  With dropout the masks
    are a function of (seed, device-side step count, element): replays draw the eager masks."""
    hp = make_hp(optimizer="adam", learning_rate=0.01, FEATURE_COUNT=400, cross_l2=0.01,
                 user_dropout=(dropout is True), dropout=[0.2, 0.4], enable_BN=(dropout == "bn"))
    rng = np.random.default_rng(12)
    p32 = X.init_params(hp, seed=4)
    from oracle import ffm_iterator_oracle as O

    def fixed_shape_batch():
        B, F, N = 64, hp.FIELD_COUNT, hp.FEATURE_COUNT
        feats = [[[f, int(rng.integers(0, N)), float(np.float32(rng.random()))] for f in range(F)] for _ in range(B)]
        return O.convert(list(rng.integers(0, 2, B).astype(float)), feats, F, N)

    batches = [fixed_shape_batch() for _ in range(3)]
    eager = load_engine(hp, p32, "sparse_rows")
    graph = load_engine(hp, p32, "sparse_rows")

    def dev(b):
        return [torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in (
            b["labels"].reshape(-1), b["dnn_feat_indices"], b["dnn_feat_values"], b["dnn_feat_weights"])]

    losses_e = [float(eager.train_step(eager.make_batch(*dev(b)))[0]) for b in batches]
    gs = graph.capture_step(*dev(batches[0]))          # runs batch 0 once while warming up
    # one batch through the four-array entry, one through the packed single-copy entry
    losses_g = [float(gs.run(*dev(batches[1]))[0]), float(gs.run_packed(gs.pack(*dev(batches[2])))[0])]
    torch.cuda.synchronize()
    graph.check_errors()
    assert losses_g == losses_e[1:]
    # state_dict() brings every row up to date first: a replay without an announced next batch
    # prepares whatever its next-ids buffer holds (rows caught up a step early, taken back later),
    # so the raw records may differ in how far single rows have been brought forward
    se, sg = eager.state_dict(), graph.state_dict()
    for k in ("embedding", "lin_w", "slot/table_m", "slot/table_v", "slot/lin_m"):
        assert torch.equal(se[k], sg[k]), k
    assert torch.equal(eager.dense, graph.dense)
    assert all(torch.equal(eager.bn_state[k], graph.bn_state[k]) for k in eager.bn_state)  # moving statistics
    assert graph.state_dict()["step"] == 3 and abs(graph.state_dict()["b1p"] - 0.9 ** 3) < 1e-6
