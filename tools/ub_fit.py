#!/usr/bin/env python
"""End-to-end rate of XDeepFMModel.fit() on Criteo-shaped libffm TEXT (the reader, the H2D copies and
the training step in the loop), next to the resident-batch rate of bench.py (GPU box):
    python tools/ub_fit.py"""
import io
import os
import re
import sys
import time
from contextlib import redirect_stdout

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from recommenders_amd.models.deeprec.deeprec_utils import prepare_hparams  # noqa: E402
from recommenders_amd.models.deeprec.io.iterator import FFMTextIterator  # noqa: E402
from recommenders_amd.models.deeprec.models.xDeepFM import XDeepFMModel  # noqa: E402
from recommenders_amd.synth import criteo_shape_ids  # noqa: E402


def write(path, n, seed):
    rng = np.random.default_rng(seed)
    ids = criteo_shape_ids(rng, n, 1 << 25)
    vals = rng.random((n, 13))
    labels = rng.random(n) < 0.25
    with open(path, "w") as f:
        for i in range(n):
            toks = ["%d:%d:%s" % (j + 1, ids[i, j] + 1, ("%.4f" % vals[i, j]) if j < 13 else "1") for j in range(39)]
            f.write("%d %s\n" % (labels[i], " ".join(toks)))
    return path


if __name__ == "__main__":
    B, n_train = 8192, 8192 * 24
    train = write("/tmp/ub_fit_train.ffm", n_train, 0)
    valid = write("/tmp/ub_fit_valid.ffm", 8192 * 2, 1)
    print("train file: %d lines, %.1f MB" % (n_train, os.path.getsize(train) / 1e6), flush=True)
    def run(epochs):
        hp = prepare_hparams(
            None, model_type="xDeepFM", data_format="ffm", FEATURE_COUNT=1 << 25, FIELD_COUNT=39, dim=16,
            layer_sizes=[20, 20], cross_layer_sizes=[20, 10], activation=["relu", "relu"], dropout=[0.0, 0.0],
            loss="log_loss", method="classification", metrics=["auc", "logloss"], use_Linear_part=True,
            use_FM_part=True, use_CIN_part=True, use_DNN_part=True, init_method="tnormal", init_value=0.1,
            learning_rate=0.002, optimizer="adam", embed_l2=0.0, layer_l2=0.0, cross_l2=0.01, batch_size=B,
            epochs=epochs, save_model=False, show_step=1000000, embedding_update="sparse_rows")
        model = XDeepFMModel(hp, FFMTextIterator, seed=3)
        buf = io.StringIO()
        t0 = time.perf_counter()
        with redirect_stdout(buf):
            model.fit(train, valid)
        return time.perf_counter() - t0, buf.getvalue().strip().splitlines()[-1]

    per_epoch = n_train + 8192 * 2  # every epoch trains on the train file and scores the valid file
    run(1)  # warm the page cache and the library
    t1, _ = run(1)
    t10, last = run(10)
    print(last)
    print("fit(), 1 epoch : %.3f s = %.0f lines/s (text parsed: reader + H2D + step / scoring + metrics)"
          % (t1, per_epoch / t1))
    later = (t10 - t1) / 9
    print("fit(), 10 epochs: %.3f s; epochs 2..10: %.4f s each = %.0f lines/s (parsed batches and their device "
          "copies kept by the iterator, XDFM_FFM_CACHE_MB)" % (t10, later, per_epoch / later))
