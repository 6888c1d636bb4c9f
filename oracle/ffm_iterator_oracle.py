"""Oracle for FFMTextIterator (test infrastructure only; see oracle/__init__.py).

Pure-Python restatement of recommenders/models/deeprec/io/iterator.py:
``parse_line`` follows :90-116, ``batches`` follows :118-150 and ``convert`` follows
:152-219.  Slow on purpose (per-token Python), meant for small cases.
"""

import numpy as np


def parse_line(line, col_sep=" ", id_sep="%"):
    """iterator.py:100-116"""
    parts = line.strip().split(id_sep)
    imp = parts[1].strip() if len(parts) == 2 else 0
    cols = parts[0].strip().split(col_sep)
    label = float(cols[0])
    toks = []
    for word in cols[1:]:
        if word.strip() == "":
            continue
        piece = word.split(":")
        toks.append((int(piece[0]) - 1, int(piece[1]) - 1, float(piece[2])))
    return label, toks, imp


def convert(labels, features, field_count, feature_count):
    """iterator.py:163-219 -> dict of the eight arrays."""
    n = len(labels)
    fm_idx, fm_val = [], []
    entries = []  # (row, col, feature, value) in file order
    width = -1
    for i in range(n):
        seen = {}
        for (field, feat, val) in features[i]:
            fm_idx.append([i, feat])
            fm_val.append(val)
            seen[field] = seen[field] + 1 if field in seen else 0
            entries.append((i * field_count + field, seen[field], feat, val))
            width = max(width, seen[field])
    width += 1
    # python's sorted() is stable: ties keep file order (:198-201)
    order = sorted(range(len(entries)), key=lambda k: (entries[k][0], entries[k][1]))
    out = {
        "labels": np.asarray([[v] for v in labels], dtype=np.float32),
        "fm_feat_indices": np.asarray(fm_idx, dtype=np.int64),
        "fm_feat_values": np.asarray(fm_val, dtype=np.float32),
        "fm_feat_shape": np.asarray([n, feature_count], dtype=np.int64),
        "dnn_feat_indices": np.asarray([[e[0], e[1]] for e in entries], dtype=np.int64)[order],
        "dnn_feat_values": np.asarray([e[2] for e in entries], dtype=np.int64)[order],
        "dnn_feat_weights": np.asarray([e[3] for e in entries], dtype=np.float32)[order],
        "dnn_feat_shape": np.asarray([n * field_count, width], dtype=np.int64),
    }
    return out


def batches(path, batch_size, field_count, feature_count, col_sep=" ", id_sep="%"):
    """iterator.py:127-150 -> yields (arrays, impression_ids, count)."""
    labels, feats, imps = [], [], []
    with open(path, "r") as fh:
        for line in fh:
            lab, toks, imp = parse_line(line, col_sep, id_sep)
            labels.append(lab)
            feats.append(toks)
            imps.append(imp)
            if len(labels) == batch_size:
                yield convert(labels, feats, field_count, feature_count), imps, batch_size
                labels, feats, imps = [], [], []
    if labels:
        yield convert(labels, feats, field_count, feature_count), imps, len(labels)
