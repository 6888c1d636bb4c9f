#!/usr/bin/env python
"""How far do row updates drift from the reference's dense-exact Adam?  BASELINE configs[1] shape
(notebook configuration, S2 data: 100 000 Criteo-sample-shaped rows, 80/10/10 split) with
embed_l2 = layer_l2 = 0, trained on the GPU from the same initial tensors in three modes:
  dense_exact  the reference's semantics (every row moves every step)
  sparse_rows  row updates + catch-up of the missed steps (what bench.py runs)
  sparse_lazy  row updates only (absent rows stand still)
Prints validation AUC / logloss per epoch and the final deltas against dense_exact."""
import os
import sys
import tempfile

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import xdeepfm_oracle as X  # noqa: E402  (initial tensors only)
from recommenders_amd.models.deeprec.deeprec_utils import prepare_hparams  # noqa: E402
from recommenders_amd.models.deeprec.io.iterator import FFMTextIterator  # noqa: E402
from recommenders_amd.models.deeprec.models.xDeepFM import XDeepFMModel  # noqa: E402
from recommenders_amd.synth import write_criteo_like  # noqa: E402


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    d = tempfile.mkdtemp()
    train, valid, test = write_criteo_like(os.path.join(d, "c"), rows=rows, feature_count=2300000)
    res = {}
    for mode in ("dense_exact", "sparse_rows", "sparse_lazy"):
        hp = prepare_hparams(
            None, model_type="xDeepFM", data_format="ffm", FEATURE_COUNT=2300000, FIELD_COUNT=39, dim=10,
            layer_sizes=[20, 20], cross_layer_sizes=[20, 10], activation=["relu", "relu"], dropout=[0.0, 0.0],
            loss="log_loss", method="classification", metrics=["auc", "logloss"], use_Linear_part=True,
            use_FM_part=False, use_CIN_part=True, use_DNN_part=True, init_method="tnormal", init_value=0.1,
            learning_rate=0.002, optimizer="adam", embed_l2=0.0, layer_l2=0.0, cross_l2=0.01, batch_size=4096,
            epochs=1, show_step=200000, save_model=False, embedding_update=mode)
        params = X.init_params(hp, seed=21)
        model = XDeepFMModel(hp, FFMTextIterator, seed=1)
        model.engine.load_state(params)
        for e in range(epochs):
            model.fit(train, valid)
        eng = model.engine
        preds, labels = [], []
        for feed, _, _ in model.iterator.load_data_from_file(test):
            b = model._batch_from_feed(feed)
            preds.append(eng.forward(b, with_loss=False).clone())
            labels.append(b.labels)
        auc, logloss, _ = eng.eval_metrics(torch.cat(preds), torch.cat(labels))
        res[mode] = (auc, logloss, torch.cat(preds).cpu().numpy())
        print("%-12s test auc %.6f logloss %.6f after %d epochs" % (mode, auc, logloss, epochs), flush=True)
    ref = res["dense_exact"]
    for mode in ("sparse_rows", "sparse_lazy"):
        r = res[mode]
        print("%-12s vs dense_exact: d auc %+.2e  d logloss %+.2e  max |d pred| %.2e" % (
            mode, r[0] - ref[0], r[1] - ref[1], np.abs(r[2] - ref[2]).max()))


if __name__ == "__main__":
    main()
