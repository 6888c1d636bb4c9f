"""Shared helpers of the test-suite: hparams, random ragged batches, oracle glue."""
import numpy as np

from oracle import ffm_iterator_oracle as it_oracle
from recommenders_amd.models.deeprec.deeprec_utils import prepare_hparams

BASE = dict(
    model_type="xDeepFM", data_format="ffm", FEATURE_COUNT=500, FIELD_COUNT=7, dim=8,
    layer_sizes=[20, 12], cross_layer_sizes=[12, 8, 6], activation=["relu", "tanh"],
    dropout=[0.0, 0.0], loss="log_loss", method="classification", metrics=["auc", "logloss"],
    use_Linear_part=True, use_FM_part=True, use_CIN_part=True, use_DNN_part=True,
    init_value=0.1, learning_rate=0.01, batch_size=64, epochs=1, save_model=False,
    embed_l2=0.0, layer_l2=0.0, cross_l2=0.0, show_step=1000000,
)


def make_hp(**kw):
    cfg = dict(BASE)
    cfg.update(kw)
    return prepare_hparams(None, **cfg)


def random_features(rng, B, F, N, multi_p=0.15, missing_p=0.1, numeric_fields=2):
    """Python lists in the layout parser_one_line produces (0-based field / feature)."""
    labels, feats = [], []
    for _ in range(B):
        toks = []
        for f in rng.permutation(F) if rng.random() < 0.2 else range(F):
            f = int(f)
            if rng.random() < missing_p:
                continue
            reps = int(rng.integers(2, 4)) if rng.random() < multi_p else 1
            for _ in range(reps):
                val = float(np.float32(rng.random())) if f < numeric_fields else 1.0
                toks.append([f, int(rng.integers(0, N)), val])
        labels.append(float(rng.integers(0, 2)))
        feats.append(toks)
    return labels, feats


def random_batch(rng, hp, B, **kw):
    labels, feats = random_features(rng, B, hp.FIELD_COUNT, hp.FEATURE_COUNT, **kw)
    return it_oracle.convert(labels, feats, hp.FIELD_COUNT, hp.FEATURE_COUNT)


def write_libffm(path, labels, feats, imp_every=0):
    with open(path, "w") as f:
        for i, (lab, toks) in enumerate(zip(labels, feats)):
            line = "%d %s" % (int(lab), " ".join("%d:%d:%r" % (t[0] + 1, t[1] + 1, t[2]) for t in toks))
            if imp_every and i % imp_every == 0:
                line += "%%imp%d" % i
            f.write(line.rstrip() + "\n")
    return path
