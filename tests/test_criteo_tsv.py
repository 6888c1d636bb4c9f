"""Criteo TSV -> DataFrame -> libffm -> FFMTextIterator on a small synthetic TSV in the Criteo DAC
layout (reference recommenders/datasets/criteo.py:22-26: label, 13 integer columns, 26 categorical
columns, tab-separated, empty cells for missing values).  The real dataset is a download the build
has no access to; the reference's own tests for this module only check the frame's shape
(tests/smoke/recommenders/dataset/test_criteo.py), which the first test mirrors."""
import io
import os
import tarfile

import numpy as np
import pandas as pd
import pytest

from oracle import ffm_iterator_oracle as O
from recommenders_amd.datasets import criteo
from recommenders_amd.models.deeprec.io.iterator import FFMTextIterator
from tests.util import make_hp


def write_tsv(path, rows=60, seed=0):
    rng = np.random.default_rng(seed)
    lines = []
    for _ in range(rows):
        ints = ["" if rng.random() < 0.2 else str(int(rng.integers(-2, 500))) for _ in range(13)]
        cats = ["" if rng.random() < 0.15 else "%08x" % int(rng.integers(0, 40)) for _ in range(26)]
        lines.append("\t".join([str(int(rng.integers(0, 2)))] + ints + cats))
    with open(path, "w") as f:
        f.write("\n".join(lines) + "\n")
    return lines


def test_header_and_frame(tmp_path):
    assert criteo.DEFAULT_HEADER[0] == "label" and len(criteo.DEFAULT_HEADER) == 40
    assert criteo.DEFAULT_HEADER[1] == "int00" and criteo.DEFAULT_HEADER[13] == "int12"
    assert criteo.DEFAULT_HEADER[14] == "cat00" and criteo.DEFAULT_HEADER[-1] == "cat25"
    lines = write_tsv(str(tmp_path / "dac_sample.txt"))
    df = criteo.load_pandas_df(size="sample", local_cache_path=str(tmp_path))
    assert df.shape == (len(lines), 40) and list(df.columns) == criteo.DEFAULT_HEADER
    first = lines[0].split("\t")
    assert int(df["label"][0]) == int(first[0])
    assert (df["cat03"][0] == first[17]) or (first[17] == "" and pd.isna(df["cat03"][0]))
    # the extracted layout of the reference (<cache>/dac/dac_sample.txt) and a direct file path
    os.makedirs(tmp_path / "b" / "dac")
    os.replace(tmp_path / "dac_sample.txt", tmp_path / "b" / "dac" / "dac_sample.txt")
    assert criteo.load_pandas_df(local_cache_path=str(tmp_path / "b")).equals(df)
    assert criteo.load_pandas_df(local_cache_path=str(tmp_path / "b" / "dac" / "dac_sample.txt")).equals(df)
    with pytest.raises(FileNotFoundError, match="does not download"):
        criteo.load_pandas_df(local_cache_path=str(tmp_path / "empty"))
    with pytest.raises(ValueError):
        criteo.load_pandas_df(size="huge", local_cache_path=str(tmp_path))


def test_archive_is_extracted_and_traversal_refused(tmp_path):
    write_tsv(str(tmp_path / "dac_sample.txt"), rows=5)
    arc = tmp_path / "cache" / "dac_sample.tar.gz"
    os.makedirs(arc.parent)
    with tarfile.open(arc, "w:gz") as tar:
        tar.add(str(tmp_path / "dac_sample.txt"), arcname="dac_sample.txt")
    assert criteo.download_criteo("sample", str(arc.parent)) == str(arc)
    df = criteo.load_pandas_df(local_cache_path=str(arc.parent))
    assert df.shape == (5, 40) and os.path.isfile(arc.parent / "dac" / "dac_sample.txt")
    bad = tmp_path / "bad.tar.gz"
    with tarfile.open(bad, "w:gz") as tar:
        info = tarfile.TarInfo("../escape.txt")
        info.size = 1
        tar.addfile(info, io.BytesIO(b"x"))
    with pytest.raises(Exception, match="Path Traversal"):
        criteo.extract_criteo("sample", str(bad), path=str(tmp_path / "out"))
    assert not os.path.exists(tmp_path / "escape.txt")


def test_frame_through_the_converter_feeds_the_iterator(tmp_path):
    lines = write_tsv(str(tmp_path / "dac_sample.txt"), rows=50, seed=3)
    df = criteo.load_pandas_df(local_cache_path=str(tmp_path))
    out = str(tmp_path / "criteo.ffm")
    conv_df, conv = criteo.to_libffm(df, filepath=out)
    params = conv.get_params()
    assert params["field count"] == 39 and params["file path"] == out
    # 13 numeric fields of one feature each, then one feature per distinct value of every category
    cats = [set(l.split("\t")[14 + j] for l in lines) for j in range(26)]
    assert params["feature count"] == 13 + sum(len(c) for c in cats)
    text = open(out).read().splitlines()
    assert len(text) == 50
    first = lines[0].split("\t")
    words = text[0].split(" ")
    assert words[0] == first[0] and len(words) == 40
    for j in range(13):  # field j+1, feature j+1, the cell (0 when missing)
        assert words[1 + j] == "%d:%d:%s" % (j + 1, j + 1, first[1 + j] or "0")
    assert all(w.split(":")[0] == str(14 + j) and w.endswith(":1") for j, w in enumerate(words[14:]))
    # and the file reads back through the product's iterator exactly as through the oracle's
    hp = make_hp(FEATURE_COUNT=params["feature count"], FIELD_COUNT=39, batch_size=16)
    it = FFMTextIterator(hp, None)
    got = list(it.load_data_from_file(out))
    exp = list(O.batches(out, 16, 39, params["feature count"]))
    assert len(got) == len(exp) == 4
    for (feed, _, n), (arrays, _, en) in zip(got, exp):
        assert n == en
        for k in ("labels", "dnn_feat_indices", "dnn_feat_values", "dnn_feat_weights"):
            assert np.array_equal(feed[getattr(it, k)], arrays[k]), k
        assert int(feed[it.dnn_feat_values].max()) < params["feature count"]


def test_hashed_writer_streams_ragged_lines(tmp_path):
    lines = write_tsv(str(tmp_path / "train.txt"), rows=70, seed=5)
    out = str(tmp_path / "hashed.ffm")
    N = 1000
    assert criteo.hashed_libffm(str(tmp_path / "train.txt"), out, feature_count=N, chunk_rows=32) == 70
    text = open(out).read().splitlines()
    assert len(text) == 70
    for src, line in zip(lines, text):
        cells = src.split("\t")
        words = line.split(" ")
        assert words[0] == cells[0]
        want = []
        for j in range(1, 40):
            if cells[j] == "":
                continue  # missing cell: no token
            if j <= 13:
                want.append("%d:%d:%s" % (j, j, cells[j]))
            else:
                fid = criteo.hashed_feature(j, cells[j], N)
                assert 14 <= fid <= N
                want.append("%d:%d:1" % (j, fid))
        assert words[1:] == want
    # same value in another field hashes elsewhere (the field is part of the key), deterministically
    assert criteo.hashed_feature(14, "abc", N) == criteo.hashed_feature(14, "abc", N)
    assert len({criteo.hashed_feature(f, "abc", 1 << 20) for f in range(14, 40)}) > 20
    hp = make_hp(FEATURE_COUNT=N, FIELD_COUNT=39, batch_size=32)
    it = FFMTextIterator(hp, None)
    got = list(it.load_data_from_file(out))
    exp = list(O.batches(out, 32, 39, N))
    assert [g[2] for g in got] == [32, 32, 6]
    for (feed, _, n), (arrays, _, en) in zip(got, exp):
        for k in ("labels", "dnn_feat_indices", "dnn_feat_values", "dnn_feat_weights", "fm_feat_indices"):
            assert np.array_equal(feed[getattr(it, k)], arrays[k]), k
