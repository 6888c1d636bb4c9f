"""Generates the golden fixtures of tests/golden/ from the REFERENCE itself.

Run in the build container only (needs /root/reference):

    python tests/golden/gen_golden.py

* hparams_golden.json / cal_metric_golden.json: produced by importing the reference's
  ``recommenders.models.deeprec.deeprec_utils`` directly (it imports here as is).
* ffm_iterator_golden.json: produced by running the reference's ``FFMTextIterator``
  VERBATIM.  That module does ``import tensorflow`` only to create eight placeholders and
  to open the file (``tf.io.gfile.GFile``); TensorFlow is not installed, so a placeholder-
  only stand-in module is registered first.  No value in the fixture flows through the
  stand-in: the arrays come from the reference's own pure-Python parsing (iterator.py:90-219).
  The first case is the 3-line file whose outputs SURVEY.md section 8(c)(3) records.

The fixtures are data (inputs + expected outputs); no reference source is stored.
"""
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")


def _tf_stand_in():
    tf = types.ModuleType("tensorflow")
    tf.float32, tf.int64, tf.int32, tf.bool = "float32", "int64", "int32", "bool"

    class _Graph:
        def as_default(self):
            return self

        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

    class _PH:
        def __init__(self, dtype, shape=None, name=None):
            self.name = name

    tf.Graph = _Graph
    tf.compat = types.SimpleNamespace(v1=types.SimpleNamespace(placeholder=_PH))
    tf.io = types.SimpleNamespace(gfile=types.SimpleNamespace(GFile=open))
    sys.modules["tensorflow"] = tf
    return tf


def gen_hparams_and_metrics():
    from recommenders.models.deeprec import deeprec_utils as R

    cases = []
    base = dict(model_type="xDeepFM", data_format="ffm", FEATURE_COUNT=1000, FIELD_COUNT=10, dim=10,
                layer_sizes=[20, 20], cross_layer_sizes=[20, 10], activation=["relu", "relu"],
                dropout=[0.0, 0.0], loss="log_loss", method="classification")
    kwargs_list = [
        base,
        dict(base, learning_rate=0.002, embed_l2=0.01, metrics=["auc", "logloss"], epochs=3),
        dict(model_type="xDeepFM"),
        dict(model_type="xDeepFM", data_format="dkn"),
        dict(base, embed_l2=0),
        dict(base, FEATURE_COUNT=1.5),
        dict(base, layer_sizes=20),
        dict(base, loss=3),
        dict(base, bad={"a": 1}),
        dict(),
    ]
    for kw in kwargs_list:
        try:
            hp = R.prepare_hparams(None, **kw)
            cases.append({"kwargs": kw, "ok": True, "values": hp.values()})
        except Exception as e:  # noqa: BLE001
            cases.append({"kwargs": kw, "ok": False, "error": type(e).__name__, "message": str(e)})
    with open(os.path.join(HERE, "hparams_golden.json"), "w") as f:
        json.dump(cases, f, indent=1, sort_keys=True)

    rng = np.random.default_rng(7)
    mcases = []
    for n in (8, 64, 1000):
        y = rng.integers(0, 2, n).astype(np.float32)
        p = rng.random(n).astype(np.float32)
        p[::5] = p[0]  # ties
        y[0], y[1] = 0.0, 1.0
        res = R.cal_metric(list(y), list(p), ["auc", "logloss", "rmse", "acc", "f1"])
        mcases.append({"labels": y.tolist(), "preds": [float(v) for v in p],
                       "metrics": ["auc", "logloss", "rmse", "acc", "f1"],
                       "expected": {k: float(v) for k, v in res.items()}})
    # the docstring-style grouped call (deeprec_utils.py:529-530 shows the format)
    gl = [rng.integers(0, 2, 6).astype(float) for _ in range(5)]
    gp = [rng.random(6) for _ in range(5)]
    for g in gl:
        g[0], g[1] = 1.0, 0.0
    res = R.cal_metric(gl, gp, ["ndcg@2;4;6", "group_auc", "mean_mrr", "hit@1;3"])
    mcases.append({"labels": [g.tolist() for g in gl], "preds": [g.tolist() for g in gp],
                   "metrics": ["ndcg@2;4;6", "group_auc", "mean_mrr", "hit@1;3"],
                   "expected": {k: float(v) for k, v in res.items()}})
    with open(os.path.join(HERE, "cal_metric_golden.json"), "w") as f:
        json.dump(mcases, f)


def gen_iterator():
    _tf_stand_in()
    from recommenders.models.deeprec.io.iterator import FFMTextIterator

    class HP:
        pass

    rng = np.random.default_rng(11)
    files = {}
    files["survey_8c3"] = (
        "1 1:1:1.0 2:4:0.5 2:5:0.25 3:9:1\n0 1:2:1 3:10:2.0%imp7\n1 2:6:1 1:3:1\n", 3, 10, 2)
    # ragged: multi-hot fields, missing fields, unsorted fields, extra spaces, partial last batch
    lines = []
    F, N = 6, 40
    for i in range(11):
        toks = []
        for f in rng.permutation(F)[: rng.integers(1, F + 1)]:
            for _ in range(rng.integers(1, 4) if rng.random() < 0.3 else 1):
                val = "1" if rng.random() < 0.5 else "%.4f" % rng.random()
                toks.append("%d:%d:%s" % (f + 1, rng.integers(1, N + 1), val))
        line = "%d %s" % (rng.integers(0, 2), " ".join(toks))
        if i % 4 == 1:
            line = line.replace(" ", "  ", 1)  # an empty word
        if i % 3 == 0:
            line += "%%q%d" % i
        lines.append(line)
    lines[5] = "0"  # a sample without any feature
    files["ragged"] = ("\n".join(lines) + "\n", F, N, 4)
    files["no_trailing_newline_crlf"] = ("1 1:1:1\r\n0 2:2:0.5\r\n1 1:3:2e-1 2:1:.5", 2, 5, 2)
    files["first_batch_empty_tokens"] = ("1\n0\n1 1:1:1\n", 2, 5, 2)
    out = {}
    for name, (text, F, N, bs) in files.items():
        path = "/tmp/_golden_%s.txt" % name
        with open(path, "w", newline="") as f:
            f.write(text)
        hp = HP()
        hp.FEATURE_COUNT, hp.FIELD_COUNT, hp.batch_size = N, F, bs
        it = FFMTextIterator(hp, sys.modules["tensorflow"].Graph())
        batches = []
        for feed, imp, n in it.load_data_from_file(path):
            by_name = {k.name: v for k, v in feed.items()}
            batches.append({"n": n, "impression_ids": imp,
                            "arrays": {k: {"dtype": str(v.dtype), "shape": list(v.shape),
                                           "data": v.reshape(-1).tolist()} for k, v in by_name.items()}})
        out[name] = {"text": text, "FIELD_COUNT": F, "FEATURE_COUNT": N, "batch_size": bs,
                     "batches": batches}
    with open(os.path.join(HERE, "ffm_iterator_golden.json"), "w") as f:
        json.dump(out, f)


def gen_notebook_hparams():
    """The HParams the quick-start notebook PRINTED when its authors ran it
    (examples/00_quick_start/xdeepfm_criteo.ipynb, the cell that calls prepare_hparams): an output
    the reference holds, taken as text from the notebook's JSON, together with that cell's keyword
    arguments.  The yaml it read (a downloaded resource) is not in the repository; the test rebuilds
    an equivalent one from the printed values."""
    import ast

    with open("/root/reference/examples/00_quick_start/xdeepfm_criteo.ipynb") as f:
        nb = json.load(f)
    for cell in nb["cells"]:
        if cell["cell_type"] != "code" or "prepare_hparams(yaml_file" not in "".join(cell["source"]):
            continue
        text = "".join("".join(o.get("text", "")) for o in cell.get("outputs", []))
        at = text.index("HParams object with values ") + len("HParams object with values ")
        values = ast.literal_eval(text[at:].strip())
        kwargs = dict(FEATURE_COUNT=2300000, FIELD_COUNT=39, cross_l2=0.01, embed_l2=0.01, layer_l2=0.01,
                      learning_rate=0.002, batch_size=4096, epochs=10, cross_layer_sizes=[20, 10],
                      init_value=0.1, layer_sizes=[20, 20], use_Linear_part=True, use_CIN_part=True,
                      use_DNN_part=True)
        with open(os.path.join(HERE, "notebook_hparams.json"), "w") as out:
            json.dump({"source": "examples/00_quick_start/xdeepfm_criteo.ipynb (printed output)",
                       "kwargs": kwargs, "printed_values": values}, out, indent=1)
        return
    raise RuntimeError("notebook cell not found")


if __name__ == "__main__":
    gen_hparams_and_metrics()
    gen_notebook_hparams()
    gen_iterator()
    print("golden fixtures written to", HERE)
