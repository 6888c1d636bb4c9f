"""BASELINE configs[1]: xDeepFM at the quick-start notebook's configuration
(/root/reference/examples/00_quick_start/xdeepfm_criteo.ipynb:142-163 -- FIELD_COUNT 39,
FEATURE_COUNT 2 300 000, dim 10, CIN [20, 10], DNN [20, 20], linear + CIN + DNN, all three l2 = 0.01,
Adam 0.002, batch 4096) end to end through XDeepFMModel.fit() on Criteo-sample-shaped libffm text
(synth.write_criteo_like, S2 of SURVEY.md 8(d): the real dac_sample needs the network), against the
CPU oracle trained from the same initial tensors on the same files.  north_star: AUC and logloss
within 1e-3 of the reference path.

The second test is the same data with embed_l2 = layer_l2 = 0 through embedding_update="sparse_rows"
(row updates + catch-up, what bench.py runs) against the oracle's DENSE-exact Adam: the reference
moves every row every step, and the row-update path must land on the same numbers."""
import numpy as np
import pytest

from oracle import ffm_iterator_oracle as O
from oracle import xdeepfm_oracle as X
from recommenders_amd.models.deeprec.deeprec_utils import cal_metric, prepare_hparams
from recommenders_amd.models.deeprec.io.iterator import FFMTextIterator
from recommenders_amd.models.deeprec.models.xDeepFM import XDeepFMModel
from recommenders_amd.synth import write_criteo_like

pytestmark = pytest.mark.gpu

N, F = 2300000, 39


def notebook_hparams(**kw):
    cfg = dict(
        model_type="xDeepFM", data_format="ffm", FEATURE_COUNT=N, FIELD_COUNT=F, dim=10, layer_sizes=[20, 20],
        cross_layer_sizes=[20, 10], activation=["relu", "relu"], dropout=[0.0, 0.0], loss="log_loss",
        method="classification", metrics=["auc", "logloss"], use_Linear_part=True, use_FM_part=False,
        use_CIN_part=True, use_DNN_part=True, init_method="tnormal", init_value=0.1, learning_rate=0.002,
        optimizer="adam", embed_l2=0.01, layer_l2=0.01, cross_l2=0.01, batch_size=4096, epochs=3,
        show_step=200000, save_model=False, save_epoch=2)
    cfg.update(kw)
    return prepare_hparams(None, **cfg)


@pytest.fixture(scope="module")
def files(tmp_path_factory):
    d = tmp_path_factory.mktemp("criteo_like")
    # configs[1]'s size: 100 000 rows = 80 000 train (20 steps of 4096 per epoch, 60 optimizer steps over the
    # 3 epochs), 10 000 valid, 10 000 test (the 80/10/10 split of SURVEY.md 8(d) S2)
    return write_criteo_like(str(d / "c"), rows=100000, feature_count=N, seed=42)


def _oracle_fit(hp, params, train, valid, mode):
    st = X.new_opt_state(params, hp)
    batches = list(O.batches(train, hp.batch_size, F, N))
    for _ in range(hp.epochs):
        for arrays, _, _ in batches:
            X.train_step(params, st, arrays, hp, mode=mode)
    preds, labels = [], []
    for arrays, _, _ in O.batches(valid, hp.batch_size, F, N):
        preds.extend(X.predict(params, arrays, hp).numpy().tolist())
        labels.extend(arrays["labels"][:, 0].tolist())
    return cal_metric(labels, preds, ["auc", "logloss"]), np.asarray(preds)


def _gpu_fit(hp, params, train, valid, out):
    model = XDeepFMModel(hp, FFMTextIterator, seed=1)
    model.engine.load_state({k: v.clone() for k, v in params.items()})
    model.fit(train, valid)
    got = model.run_eval(valid)
    model.predict(valid, out)
    return got, np.asarray([float(v) for v in open(out).read().split()])


def test_notebook_configuration_dense_exact_adam_matches_oracle(files, tmp_path):
    train, valid, _ = files
    hp = notebook_hparams()  # embedding_update defaults to dense_exact: reference semantics
    params = X.init_params(hp, seed=21)
    got, preds = _gpu_fit(hp, params, train, valid, str(tmp_path / "p.txt"))
    exp, exp_preds = _oracle_fit(hp, params, train, valid, "dense_exact")
    assert abs(got["auc"] - exp["auc"]) <= 1e-3, (got, exp)
    assert abs(got["logloss"] - exp["logloss"]) <= 1e-3, (got, exp)
    assert np.abs(preds - exp_preds).max() <= 2e-3


def test_sparse_rows_with_catch_up_lands_on_the_dense_exact_reference(files, tmp_path):
    train, valid, _ = files
    hp = notebook_hparams(embed_l2=0.0, layer_l2=0.0, embedding_update="sparse_rows")
    params = X.init_params(hp, seed=22)
    got, preds = _gpu_fit(hp, params, train, valid, str(tmp_path / "p.txt"))
    exp, exp_preds = _oracle_fit(hp, params, train, valid, "dense_exact")
    assert abs(got["auc"] - exp["auc"]) <= 1e-3, (got, exp)
    assert abs(got["logloss"] - exp["logloss"]) <= 1e-3, (got, exp)
    assert np.abs(preds - exp_preds).max() <= 2e-3
