"""prepare_hparams / HParams / cal_metric against fixtures generated from the reference's own
deeprec_utils (tests/golden/gen_golden.py).  Mirrors the reference's
tests/unit/recommenders/models/test_deeprec_utils.py:17-47."""
import math

import numpy as np
import pytest

from recommenders_amd.models.deeprec import deeprec_utils as U


def test_prepare_hparams_matches_reference(golden):
    cases = golden("hparams_golden.json")
    assert len(cases) >= 8
    for case in cases:
        kw = case["kwargs"]
        if case["ok"]:
            hp = U.prepare_hparams(None, **kw)
            assert hp.values() == case["values"], kw
        else:
            with pytest.raises(Exception) as exc:
                U.prepare_hparams(None, **kw)
            assert type(exc.value).__name__ == case["error"], kw
            assert str(exc.value) == case["message"], kw


def test_hparams_attributes_and_yaml(tmp_path):
    # reference test_prepare_hparams / test_load_yaml_file: FEATURE_COUNT, data_format, dim present
    y = tmp_path / "xDeepFM.yaml"
    y.write_text(
        "data:\n  FEATURE_COUNT: 1000\n  FIELD_COUNT: 10\n  data_format: ffm\n"
        "model:\n  method: classification\n  model_type: xDeepFM\n  dim: 10\n  layer_sizes: [20, 20]\n"
        "  activation: [relu, relu]\n  dropout: [0.0, 0.0]\n  cross_layer_sizes: [20, 10]\n"
        "  use_Linear_part: true\n  use_CIN_part: true\n  use_DNN_part: true\n"
        "train:\n  loss: log_loss\n  learning_rate: 0.001\n  epochs: 2\n  batch_size: 128\n"
        "info:\n  metrics: [auc, logloss]\n  save_model: false\n"
    )
    config = U.load_yaml(str(y))
    assert config is not None
    hp = U.prepare_hparams(str(y), learning_rate=0.01)
    assert hasattr(hp, "FEATURE_COUNT") and hasattr(hp, "data_format") and hasattr(hp, "dim")
    assert hp.learning_rate == 0.01 and hp.batch_size == 128 and hp.use_CIN_part is True
    assert "metrics" in hp.values()
    with pytest.raises(FileNotFoundError):
        U.load_yaml(str(tmp_path / "missing.yaml"))
    bad = tmp_path / "bad.yaml"
    bad.write_text("a: [1\n")
    with pytest.raises(IOError):
        U.load_yaml(str(bad))


def test_cal_metric_matches_reference(golden):
    for case in golden("cal_metric_golden.json"):
        res = U.cal_metric(case["labels"], case["preds"], case["metrics"])
        assert set(res) == set(case["expected"])
        for k, v in case["expected"].items():
            assert math.isclose(float(res[k]), v, rel_tol=0, abs_tol=1e-12), (k, res[k], v)


def test_cal_metric_docstring_example_and_errors():
    # cal_metric([0,1,1,0],[.1,.9,.4,.35],["auc","logloss"]) on the reference
    res = U.cal_metric([0, 1, 1, 0], [0.1, 0.9, 0.4, 0.35], ["auc", "logloss"])
    assert res == {"auc": 1.0, "logloss": 0.3894}
    with pytest.raises(ValueError):
        U.cal_metric([0, 1], [0.1, 0.2], ["nope"])
    with pytest.raises(ValueError):
        U.cal_metric([1, 1], [0.1, 0.2], ["auc"])
    # preds of exactly 0 / 1 are clipped, not infinite
    r = U.cal_metric([0.0, 1.0], [np.float32(1.0), np.float32(0.0)], ["logloss"])
    assert np.isfinite(r["logloss"]) and r["logloss"] > 20


def test_notebook_hparams_equal_the_values_the_reference_printed(golden, tmp_path):
    """configs[1]: the quick-start notebook printed its HParams (xdeepfm_criteo.ipynb, the
    prepare_hparams cell).  The yaml it read is a downloaded resource; an equivalent one is rebuilt
    here from the printed values of the keys such a file sets -- every other key of the printed
    dict must come out of this package's defaults and the cell's keyword arguments."""
    import yaml

    fx = golden("notebook_hparams.json")
    printed, kwargs = fx["printed_values"], fx["kwargs"]
    sections = {
        "data": ["data_format"],
        "model": ["method", "model_type", "dim", "activation", "dropout", "use_FM_part"],
        "train": ["init_method", "loss", "optimizer", "enable_BN"],
        "show": ["show_step", "save_model", "save_epoch", "metrics"],
        "info": ["load_model_name"],
    }
    doc = {sec: {k: printed[k] for k in keys} for sec, keys in sections.items()}
    path = tmp_path / "xDeepFM.yaml"
    path.write_text(yaml.safe_dump(doc))
    hp = U.prepare_hparams(str(path), **kwargs)
    got = hp.values()
    assert set(got) == set(printed), (set(got) ^ set(printed))
    for k, v in printed.items():
        assert got[k] == v and type(got[k]) is type(v), (k, got[k], v)
    assert "HParams object with values " + repr(printed) == str(hp) or got == printed
