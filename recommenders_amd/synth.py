"""Deterministic synthetic libffm data (inputs S1-S3).

The reference downloads its sample data at run time (deeprec_utils.py:430-444); offline
there is none, so tests, the smoke run and bench.py use these generators.  They are NOT
the reference's files and nothing claims their metrics equal the notebook's.
"""

import numpy as np

__all__ = ["write_plumbing_sample", "write_criteo_like", "criteo_field_layout", "criteo_shape_ids"]


def write_plumbing_sample(path, n=2000, field_count=10, feature_count=1000, seed=42):
    """S1: small ragged libffm file: one feature per field, ~10 % of the lines with a 2- or
    3-hot field, ~10 % with a missing non-final field, 3 numeric fields with U(0,1) values."""
    rng = np.random.default_rng(seed)
    per_field = feature_count // field_count
    with open(path, "w") as f:
        for i in range(n):
            toks = []
            multi = rng.random() < 0.1
            missing = rng.random() < 0.1
            skip = int(rng.integers(0, field_count - 1)) if missing else -1
            hot = int(rng.integers(0, field_count)) if multi else -1
            for fld in range(field_count):
                if fld == skip:
                    continue
                reps = int(rng.integers(2, 4)) if fld == hot else 1
                for _ in range(reps):
                    feat = fld * per_field + int(rng.integers(0, per_field)) + 1
                    val = "%.4f" % rng.random() if fld < 3 else "1"
                    toks.append("%d:%d:%s" % (fld + 1, feat, val))
            line = "%d %s" % (int(rng.random() < 0.5), " ".join(toks))
            if i % 7 == 0:
                line += "%%imp%d" % (i // 7)
            f.write(line + "\n")
    return path


def criteo_field_layout(feature_count, n_numeric=13, n_categorical=26):
    """Start offset and cardinality of every field: numeric fields own one feature id each,
    categorical fields split the remaining ids with geometrically spread cardinalities."""
    rest = feature_count - n_numeric
    w = np.geomspace(1.0, 4000.0, n_categorical)
    card = np.maximum((w / w.sum() * rest).astype(np.int64), 2)
    card[-1] += rest - card.sum()
    cards = np.concatenate([np.ones(n_numeric, dtype=np.int64), card])
    starts = np.concatenate([[0], np.cumsum(cards)[:-1]])
    return starts, cards


def criteo_shape_ids(rng, n_rows, feature_count, n_numeric=13, n_categorical=26):
    """[n_rows, 39] 0-based feature ids: log-uniform (Zipf-like, exponent ~1) inside each
    categorical field's range; the numeric fields collapse to one hot id each."""
    starts, cards = criteo_field_layout(feature_count, n_numeric, n_categorical)
    u = rng.random((n_rows, n_numeric + n_categorical))
    rank = np.floor(np.exp(u * np.log(cards.astype(np.float64)))).astype(np.int64) - 1
    rank = np.clip(rank, 0, cards - 1)
    return starts[None, :] + rank


def write_criteo_like(prefix, rows=100000, feature_count=2300000, seed=42, split=(0.8, 0.1, 0.1)):
    """S2: Criteo-sample-shaped libffm text (39 fields = 13 numeric + 26 categorical) with
    labels from a planted logistic model.  Writes <prefix>_train/_valid/_test, returns paths."""
    rng = np.random.default_rng(seed)
    n_num, n_cat = 13, 26
    ids = criteo_shape_ids(rng, rows, feature_count, n_num, n_cat)
    num_vals = np.round(np.log1p(rng.lognormal(1.0, 1.5, (rows, n_num))) / 10.0, 4)
    effect = np.random.default_rng(seed + 1).normal(0.0, 0.45, feature_count)
    num_coef = np.random.default_rng(seed + 2).normal(0.0, 1.5, n_num)
    logit = -1.1 + effect[ids[:, n_num:]].sum(1) + (num_vals * num_coef).sum(1)
    # a pairwise term so that the interaction parts have something to find
    logit += 0.8 * effect[ids[:, n_num]] * effect[ids[:, n_num + 1]] / 0.45
    labels = (rng.random(rows) < 1.0 / (1.0 + np.exp(-logit))).astype(np.int64)
    bounds = np.cumsum([int(rows * s) for s in split])
    names = ["train", "valid", "test"]
    paths, lo = [], 0
    for name, hi in zip(names, bounds):
        path = "%s_%s" % (prefix, name)
        with open(path, "w") as f:
            for i in range(lo, hi):
                toks = ["%d:%d:%.4f" % (k + 1, ids[i, k] + 1, num_vals[i, k]) for k in range(n_num)]
                toks += ["%d:%d:1" % (k + 1, ids[i, k] + 1) for k in range(n_num, n_num + n_cat)]
                f.write("%d %s\n" % (labels[i], " ".join(toks)))
        paths.append(path)
        lo = hi
    return paths
