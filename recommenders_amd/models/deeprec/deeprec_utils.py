"""Hyper-parameter handling and evaluation metrics for the xDeepFM path.

Drop-in for the names the reference exposes in
``recommenders/models/deeprec/deeprec_utils.py`` that the xDeepFM path uses:

* ``flat_config``      (reference :21-35)
* ``check_type``       (reference :38-131)
* ``check_nn_config``  (reference :134-283)
* ``load_yaml``        (reference :286-302)
* ``HParams``          (reference :305-339)
* ``create_hparams``   (reference :342-404)
* ``prepare_hparams``  (reference :407-427)
* ``cal_metric``       (reference :514-602)

The behaviour (accepted keys, defaults, error types and messages) follows the
reference; the implementation is table driven and has no sklearn dependency:
AUC is the tie-aware rank statistic (equal to sklearn's trapezoidal
``roc_auc_score``), logloss is computed in float64 after the reference's
``[1e-11, 1 - 1e-11]`` clip.  ``XDeepFMModel.run_eval`` does not come through
here for ``auc``/``logloss``: it uses the fused HIP evaluation kernel
(``xdfm_eval_auc_logloss``) and only rounds on the host.
"""

import numpy as np
import yaml

__all__ = [
    "flat_config",
    "check_type",
    "check_nn_config",
    "load_yaml",
    "HParams",
    "create_hparams",
    "prepare_hparams",
    "cal_metric",
]


def flat_config(config):
    """Merge the sections (data/model/train/info) of a yaml config into one dict.

    Later sections win on duplicate keys, as in reference :30-35.
    """
    merged = {}
    for section in config.keys():
        merged.update(config[section].items())
    return merged


# key -> required python type.  Reference :48-131 (the duplicated entries of the
# reference's int list collapse here).
_TYPED_KEYS = (
    (
        int,
        "int",
        (
            "word_size entity_size doc_size history_size FEATURE_COUNT FIELD_COUNT dim "
            "epochs batch_size show_step save_epoch PAIR_NUM DNN_FIELD_NUM "
            "attention_layer_sizes n_user n_item n_user_attr n_item_attr "
            "item_embedding_dim cate_embedding_dim user_embedding_dim max_seq_length "
            "hidden_size T L n_v n_h kernel_size min_seq_length attention_size "
            "train_num_ngs"
        ).split(),
    ),
    (
        float,
        "float",
        "init_value learning_rate embed_l2 embed_l1 layer_l2 layer_l1 mu".split(),
    ),
    (
        str,
        "str",
        (
            "train_file eval_file test_file infer_file method load_model_name "
            "infer_model_name loss optimizer init_method attention_activation "
            "user_vocab item_vocab cate_vocab"
        ).split(),
    ),
    (list, "list", "layer_sizes activation dropout att_fcn_layer_sizes dilations".split()),
)


def check_type(config):
    """Raise ``TypeError`` when a typed key holds a value of another type."""
    for pytype, label, keys in _TYPED_KEYS:
        for key in keys:
            if key in config and not isinstance(config[key], pytype):
                raise TypeError("Parameters {0} must be {1}".format(key, label))


def check_nn_config(f_config):
    """Validate a flat config: required keys, data format, value types.

    Raises ``ValueError`` for a missing key or a wrong ``data_format`` and
    ``TypeError`` (from :func:`check_type`) for a wrongly typed value.
    A config without ``model_type`` raises ``KeyError`` like the reference.
    """
    model_type = f_config["model_type"]
    # Reference quirk, kept: the FM / LR / DKN / xDeepFM required-key lists are selected by a first if/elif
    # chain (:143-186) and then overwritten by a second, independent chain over the sequential models
    # (:187-262) whose final ``else`` resets the list to [].  Net effect on this path: xDeepFM has NO required
    # keys; a missing one surfaces later (``KeyError`` on ``data_format`` below, ``AttributeError`` in the
    # model).  The sequential models' own lists are not restated here: those models are out of scope.
    if model_type in ("exDeepFM", "xDeepFM"):
        if f_config["data_format"] != "ffm":
            raise ValueError(
                "For xDeepFM model, data format must be 'ffm', but your set is {0}".format(
                    f_config["data_format"]
                )
            )
    check_type(f_config)


def load_yaml(filename):
    """Load a yaml file with the safe loader.

    ``FileNotFoundError`` propagates; any other failure becomes ``IOError``.
    """
    try:
        with open(filename, "r") as f:
            return yaml.load(f, yaml.SafeLoader)
    except FileNotFoundError:
        raise
    except Exception:
        raise IOError("load {0} error!".format(filename))


class HParams:
    """Attribute bag of hyper-parameters (reference :305-339)."""

    def __init__(self, hparams_dict):
        for val in hparams_dict.values():
            if not isinstance(val, (int, float, str, list)):
                raise ValueError(
                    "Hyperparameter value {} should be integer, float, string or list.".format(
                        val
                    )
                )
        self._values = hparams_dict
        for name, val in hparams_dict.items():
            setattr(self, name, val)

    def __repr__(self):
        return "HParams object with values {}".format(self._values.__repr__())

    def values(self):
        """The underlying dict (used for key-membership tests by the model)."""
        return self._values


_DEFAULTS = {
    # dkn
    "use_entity": True,
    "use_context": True,
    # model
    "cross_activation": "identity",
    "user_dropout": False,
    "dropout": [0.0],
    "attention_dropout": 0.0,
    "load_saved_model": False,
    "fast_CIN_d": 0,
    "use_Linear_part": False,
    "use_FM_part": False,
    "use_CIN_part": False,
    "use_DNN_part": False,
    # train
    "init_method": "tnormal",
    "init_value": 0.01,
    "embed_l2": 0.0,
    "embed_l1": 0.0,
    "layer_l2": 0.0,
    "layer_l1": 0.0,
    "cross_l2": 0.0,
    "cross_l1": 0.0,
    "reg_kg": 0.0,
    "learning_rate": 0.001,
    "lr_rs": 1,
    "lr_kg": 0.5,
    "kg_training_interval": 5,
    "max_grad_norm": 2,
    "is_clip_norm": 0,
    "dtype": 32,
    "optimizer": "adam",
    "epochs": 10,
    "batch_size": 1,
    "enable_BN": False,
    # show info
    "show_step": 1,
    "save_model": True,
    "save_epoch": 5,
    "write_tfevents": False,
    # sequential
    "train_num_ngs": 4,
    "need_sample": True,
    "embedding_dropout": 0.0,
    "EARLY_STOP": 100,
    # caser
    "min_seq_length": 1,
    # sum
    "slots": 5,
    "cell": "SUM",
}


def create_hparams(flags):
    """Defaults (reference :351-402) overlaid with ``flags``."""
    values = {k: (list(v) if isinstance(v, list) else v) for k, v in _DEFAULTS.items()}
    values.update(flags)
    return HParams(values)


def prepare_hparams(yaml_file=None, **kwargs):
    """yaml (optional) + keyword overrides -> checked :class:`HParams`."""
    config = flat_config(load_yaml(yaml_file)) if yaml_file is not None else {}
    for name, value in kwargs.items():
        config[name] = value
    check_nn_config(config)
    return create_hparams(config)


# ----------------------------------------------------------------------------
# metrics
# ----------------------------------------------------------------------------
def _binary_auc(labels, preds):
    """Area under the ROC curve with tied scores grouped (Mann-Whitney U with
    average ranks == trapezoidal ROC).  float64."""
    y = np.asarray(labels).reshape(-1)
    s = np.asarray(preds, dtype=np.float64).reshape(-1)
    classes = np.unique(y)
    if classes.size != 2:
        raise ValueError(
            "Only one class present in y_true. ROC AUC score is not defined in that case."
        )
    pos = y == classes[1]
    n_pos = int(pos.sum())
    n_neg = int(y.size - n_pos)
    order = np.argsort(s, kind="mergesort")
    s_sorted = s[order]
    pos_sorted = pos[order]
    # group boundaries of equal scores
    boundary = np.flatnonzero(np.diff(s_sorted)) + 1
    starts = np.concatenate(([0], boundary))
    ends = np.concatenate((boundary, [s.size]))
    cum_pos = np.concatenate(([0], np.cumsum(pos_sorted, dtype=np.int64)))
    pos_in = cum_pos[ends] - cum_pos[starts]
    neg_in = (ends - starts) - pos_in
    neg_before = starts - cum_pos[starts]
    # pairs (pos, neg) with score(pos) > score(neg), ties count one half
    u2 = int(np.sum(2 * pos_in * neg_before + pos_in * neg_in))  # 2*U, exact integer
    return u2 / (2.0 * n_pos * n_neg)


def _binary_logloss(labels, preds):
    y = np.asarray(labels, dtype=np.float64).reshape(-1)
    p = np.asarray(preds, dtype=np.float64).reshape(-1)
    return float(-np.mean(y * np.log(p) + (1.0 - y) * np.log(1.0 - p)))


def _mrr_score(y_true, y_score):
    order = np.argsort(y_score)[::-1]
    y_true = np.take(y_true, order)
    return np.sum(y_true / (np.arange(len(y_true)) + 1)) / np.sum(y_true)


def _dcg_score(y_true, y_score, k=10):
    k = min(np.shape(y_true)[-1], k)
    order = np.argsort(y_score)[::-1]
    y_true = np.take(y_true, order[:k])
    return np.sum((2**y_true - 1) / np.log2(np.arange(len(y_true)) + 2))


def _ndcg_score(y_true, y_score, k=10):
    return _dcg_score(y_true, y_score, k) / _dcg_score(y_true, y_true, k)


def _hit_score(y_true, y_score, k=10):
    truth = set(np.where(np.asarray(y_true) == 1)[0].tolist())
    for idx in np.argsort(y_score)[::-1][:k]:
        if int(idx) in truth:
            return 1
    return 0


def _at_list(metric):
    parts = metric.split("@")
    if len(parts) > 1:
        return [int(tok) for tok in parts[1].split(";")]
    return [1, 2]


def cal_metric(labels, preds, metrics):
    """Compute the requested metrics (reference :514-602).

    ``auc`` and ``logloss`` are the xDeepFM metrics; ``rmse``, ``acc``, ``f1`` are the other pointwise
    ones, ``mean_mrr``, ``ndcg@..``, ``hit@..``, ``group_auc`` the grouped ranking metrics (lists of
    per-group labels / predictions, reference :558-599).  Values are rounded to 4 decimals exactly where the
    reference rounds.  Like the reference, ``logloss`` rebinds ``preds`` to the clipped
    list for the metrics that follow it, and ``acc``/``f1`` threshold a fresh array.
    """
    res = {}
    for metric in metrics:
        if metric == "auc":
            res["auc"] = round(_binary_auc(labels, preds), 4)
        elif metric == "rmse":
            mse = float(
                np.mean(
                    (np.asarray(labels, dtype=np.float64) - np.asarray(preds, dtype=np.float64))
                    ** 2
                )
            )
            res["rmse"] = np.sqrt(round(mse, 4))
        elif metric == "logloss":
            # float() first: the reference's clip is meant to compare in float64 (its
            # pinned numpy<1.24 does); under numpy>=2 a float32 pred of exactly 1.0
            # would slip through ``min(p, 1.0 - 10e-12)`` (weak-scalar comparison).
            preds = [max(min(float(p), 1.0 - 10e-12), 10e-12) for p in preds]
            res["logloss"] = round(_binary_logloss(labels, preds), 4)
        elif metric in ("acc", "f1"):
            hard = (np.asarray(preds) >= 0.5).astype(np.float64)
            truth = np.asarray(labels).reshape(-1)
            if metric == "acc":
                res["acc"] = round(float(np.mean(truth == hard.reshape(-1))), 4)
            else:
                tp = float(np.sum((hard.reshape(-1) == 1) & (truth == 1)))
                fp = float(np.sum((hard.reshape(-1) == 1) & (truth != 1)))
                fn = float(np.sum((hard.reshape(-1) != 1) & (truth == 1)))
                denom = 2 * tp + fp + fn
                res["f1"] = round(2 * tp / denom if denom > 0 else 0.0, 4)
        elif metric == "mean_mrr":
            res["mean_mrr"] = round(
                np.mean([_mrr_score(l, p) for l, p in zip(labels, preds)]), 4
            )
        elif metric.startswith("ndcg"):
            for k in _at_list(metric):
                res["ndcg@{0}".format(k)] = round(
                    np.mean([_ndcg_score(l, p, k) for l, p in zip(labels, preds)]), 4
                )
        elif metric.startswith("hit"):
            for k in _at_list(metric):
                res["hit@{0}".format(k)] = round(
                    np.mean([_hit_score(l, p, k) for l, p in zip(labels, preds)]), 4
                )
        elif metric == "group_auc":
            res["group_auc"] = round(
                np.mean([_binary_auc(l, p) for l, p in zip(labels, preds)]), 4
            )
        else:
            raise ValueError("Metric {0} not defined".format(metric))
    return res
