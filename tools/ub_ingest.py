#!/usr/bin/env python
"""Host-side ingest rate of FFMTextIterator on Criteo-shaped libffm text (39 tokens per line):
lines/s through the native reader with 1 and with the default number of parsing threads."""
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def measure():
    from recommenders_amd.models.deeprec.deeprec_utils import prepare_hparams
    from recommenders_amd.models.deeprec.io.iterator import FFMTextIterator

    path, N, B = sys.argv[2], 1 << 25, 8192
    hp = prepare_hparams(None, model_type="xDeepFM", data_format="ffm", FEATURE_COUNT=N, FIELD_COUNT=39, dim=16,
                         layer_sizes=[20], cross_layer_sizes=[20], activation=["relu"], dropout=[0.0],
                         loss="log_loss", method="classification", metrics=["auc"], batch_size=B, save_model=False)
    it = FFMTextIterator(hp, None)
    best = 0.0
    for _ in range(3):
        t0, cnt = time.perf_counter(), 0
        for _feed, _, c in it.load_data_from_file(path):
            cnt += c
        best = max(best, cnt / (time.perf_counter() - t0))
    print("threads=%s  %.0f lines/s  %.1f MB/s" % (os.environ.get("XDFM_PARSE_THREADS", "default"), best,
                                                   best * os.path.getsize(path) / cnt / 1e6))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--measure":
        measure()
        sys.exit(0)
    from recommenders_amd.synth import criteo_shape_ids

    path = "/tmp/criteo_shape_ub.ffm"
    n = 8192 * 16
    rng = np.random.default_rng(0)
    ids = criteo_shape_ids(rng, n, 1 << 25)
    vals = rng.random((n, 13))
    with open(path, "w") as f:
        for i in range(n):
            toks = ["%d:%d:%s" % (j + 1, ids[i, j] + 1, ("%.4f" % vals[i, j]) if j < 13 else "1") for j in range(39)]
            f.write("%d %s\n" % (i & 1, " ".join(toks)))
    print("file: %d lines, %.1f MB" % (n, os.path.getsize(path) / 1e6))
    for t in ("1", "4", "16", None):
        env = dict(os.environ)
        env["XDFM_FFM_CACHE_MB"] = "0"  # the reader itself: no kept batches
        if t is None:
            env.pop("XDFM_PARSE_THREADS", None)
        else:
            env["XDFM_PARSE_THREADS"] = t
        subprocess.run([sys.executable, os.path.abspath(__file__), "--measure", path], env=env, check=True)
