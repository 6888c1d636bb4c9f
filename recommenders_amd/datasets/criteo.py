"""Criteo DAC (display advertising challenge) TSV -> DataFrame -> libffm text: the data format on the
input side of the xDeepFM hot path.

Mirror of `recommenders.datasets.criteo` (reference recommenders/datasets/criteo.py:18-61, :126-182):
the same header (label, int00..int12, cat00..cat25), `load_pandas_df(size, local_cache_path, header)`
and `extract_criteo(size, compressed_file, path)`.  Differences, on purpose:

  * nothing is downloaded (there is no network where this package is built and run):
    `local_cache_path` must hold `dac_sample.txt` / `train.txt`, the `dac/` directory the reference
    extracts to, or the `.tar.gz` the reference downloads; otherwise FileNotFoundError says what
    to put where.  `download_criteo` only finds such a local archive.
  * the Spark loader (`load_spark_df`, `get_spark_schema`) is outside the path and absent.

Added for the path itself (the reference's notebook reads a file somebody converted before):

  * `to_libffm(df, filepath)`: the frame through `LibffmConverter` (13 numeric fields of one feature
    each, 26 categorical fields with one feature per distinct value), missing cells filled the way
    a frame must be for the converter's dtype check (integers 0, categories "");
  * `hashed_libffm(tsv_path, out_path, feature_count)`: a streaming writer for files too large for a
    value dictionary -- numeric field j is feature j with the cell as value, categorical cells are
    hashed (CRC-32 of "<field>=<value>") into the remaining ids with value 1, missing cells are
    skipped (FFMTextIterator takes ragged lines).
"""
import os
import tarfile
import zlib

import numpy as np
import pandas as pd

from recommenders_amd.datasets.pandas_df_utils import LibffmConverter

N_INT, N_CAT = 13, 26
DEFAULT_HEADER = (
    ["label"]
    + ["int{0:02d}".format(i) for i in range(N_INT)]
    + ["cat{0:02d}".format(i) for i in range(N_CAT)]
)
_FILENAME = {"sample": "dac_sample.txt", "full": "train.txt"}
_ARCHIVE = {"sample": "dac_sample.tar.gz", "full": "10082655"}


def _check_size(size):
    if size not in _FILENAME:
        raise ValueError("size must be 'sample' or 'full', not {0!r}".format(size))


def download_criteo(size="sample", work_directory="."):
    """Path of the Criteo archive inside ``work_directory`` (reference :126-138 downloads it; this
    build never touches the network and only finds a file that is already there)."""
    _check_size(size)
    for name in (_ARCHIVE[size], "dac.tar.gz", "dac_sample.tar.gz"):
        path = os.path.join(work_directory, name)
        if os.path.isfile(path):
            return path
    raise FileNotFoundError(
        "no Criteo archive in {0}: this package does not download; place {1} (or the extracted {2}) "
        "there".format(work_directory, _ARCHIVE[size], _FILENAME[size])
    )


def extract_criteo(size, compressed_file, path=None):
    """Extract the Criteo tar (reference :141-182); returns the path of the extracted text file.
    Members that would land outside the target directory are refused."""
    _check_size(size)
    extracted_dir = os.path.join(os.path.dirname(compressed_file), "dac") if path is None else path
    root = os.path.abspath(extracted_dir)
    with tarfile.open(compressed_file) as tar:
        members = tar.getmembers()
        for member in members:
            target = os.path.abspath(os.path.join(extracted_dir, member.name))
            if os.path.commonpath([root, target]) != root or member.issym() or member.islnk():
                raise Exception("Attempted Path Traversal in Tar File")
        tar.extractall(extracted_dir, members)
    return os.path.join(extracted_dir, _FILENAME[size])


def _find_local(size, local_cache_path):
    _check_size(size)
    base = "." if local_cache_path is None else str(local_cache_path)
    if os.path.isfile(base):
        if tarfile.is_tarfile(base):
            return extract_criteo(size, base)
        return base
    for cand in (os.path.join(base, _FILENAME[size]), os.path.join(base, "dac", _FILENAME[size])):
        if os.path.isfile(cand):
            return cand
    return extract_criteo(size, download_criteo(size, base))


def load_pandas_df(size="sample", local_cache_path=None, header=DEFAULT_HEADER):
    """The Criteo DAC dataset as a ``pandas.DataFrame`` (reference :29-61): tab-separated, no header
    line, ``<label> <13 integer features> <26 categorical features>``; empty cells are NaN."""
    filepath = _find_local(size, local_cache_path)
    return pd.read_csv(filepath, sep="\t", header=None, names=list(header))


def to_libffm(df, filepath=None, col_rating="label"):
    """A Criteo frame -> libffm through `LibffmConverter`: returns ``(converted_frame, converter)``;
    ``converter.get_params()`` gives FIELD_COUNT / FEATURE_COUNT for the hparams.  Integer columns
    with missing cells (float64 with NaN after `read_csv`) become int64 with 0, missing categories
    the empty string (one more category of that field)."""
    out = {}
    for name in df.columns:
        col = df[name]
        if name == col_rating:
            out[name] = col.astype(np.int64) if not col.isna().any() else col
        elif col.dtype == object:
            out[name] = col.fillna("").astype(object)
        elif np.issubdtype(col.dtype, np.floating) and bool(np.all(np.isnan(col.values) | (col.values == np.floor(col.values)))):
            out[name] = col.fillna(0).astype(np.int64)
        else:
            out[name] = col.fillna(0)
    clean = pd.DataFrame(out, index=df.index, columns=list(df.columns))
    converter = LibffmConverter(filepath=filepath).fit(clean, col_rating=col_rating)
    return converter.transform(clean), converter


def hashed_feature(field, value, feature_count, n_numeric=N_INT):
    """1-based feature id of a categorical cell: CRC-32 of ``"<field>=<value>"`` folded into the ids
    that follow the numeric fields' own (``n_numeric + 1 .. feature_count``)."""
    space = int(feature_count) - n_numeric
    if space <= 0:
        raise ValueError("feature_count must exceed the number of numeric fields")
    return n_numeric + 1 + zlib.crc32(("%d=%s" % (field, value)).encode("utf-8")) % space


def hashed_libffm(tsv_path, out_path, feature_count, chunk_rows=200000, n_numeric=N_INT, n_categorical=N_CAT):
    """Stream a Criteo TSV into libffm text with hashed categorical ids; returns the number of
    lines written.  Field numbers are 1-based in column order.  Numeric field j: ``j:j:<cell>``
    (integers print without a decimal point); categorical field j: ``j:<hashed id>:1``; empty cells
    give no token."""
    names = ["label"] + ["c%d" % i for i in range(n_numeric + n_categorical)]
    lines = 0
    reader = pd.read_csv(tsv_path, sep="\t", header=None, names=names, dtype=str, keep_default_na=False,
                         chunksize=int(chunk_rows), quoting=3)
    with open(out_path, "w") as out:
        for chunk in reader:
            cols = [chunk["label"].values]
            for j in range(1, n_numeric + n_categorical + 1):
                cell = chunk[names[j]].values
                tok = np.empty(len(cell), dtype=object)
                present = cell != ""
                if j <= n_numeric:
                    tok[present] = [" %d:%d:%s" % (j, j, v) for v in cell[present]]
                else:
                    uniq, inv = np.unique(cell[present], return_inverse=True)
                    ids = np.array([" %d:%d:1" % (j, hashed_feature(j, v, feature_count, n_numeric)) for v in uniq],
                                   dtype=object)
                    tok[present] = ids[inv] if len(uniq) else []
                tok[~present] = ""
                cols.append(tok)
            rows = cols[0].astype(object)
            for tok in cols[1:]:
                rows = rows + tok
            out.write("\n".join(rows))
            out.write("\n")
            lines += len(rows)
    return lines
