"""libffm text input for xDeepFM: drop-in ``FFMTextIterator``.

Mirrors the public surface of the reference's
``recommenders/models/deeprec/io/iterator.py``:

* ``FFMTextIterator(hparams, graph, col_spliter=" ", ID_spliter="%")``  (reference :50)
* ``.parser_one_line(line)``                                            (reference :90-116)
* ``.load_data_from_file(infile)`` -> yields ``(feed_dict, impression_ids, n)`` (:118-150)
* ``._convert_data(labels, features)``                                  (reference :152-219)
* ``.gen_feed_dict(data_dict)``                                         (reference :221-241)
* the eight placeholder attributes ``labels, fm_feat_indices, ...``     (reference :67-88)

There is no TensorFlow here: the placeholder attributes are inert ``Placeholder``
handles that only serve as ``feed_dict`` keys, and ``graph`` is accepted and ignored.
``load_data_from_file`` is backed by the C++ reader in libxdfm.so
(``xdfm_ffm_open/next/fill``) whenever both separators are single characters; the
arrays it yields have the reference's dtypes, shapes and ordering.
"""

import abc
import ctypes
import os

import numpy as np

from recommenders_amd import _lib

__all__ = ["BaseIterator", "FFMTextIterator", "Placeholder", "FeedDict"]


class Placeholder(object):
    """Inert stand-in for ``tf.compat.v1.placeholder``: a named feed_dict key."""

    __slots__ = ("name", "dtype", "shape")

    def __init__(self, dtype, shape, name):
        self.name = name
        self.dtype = np.dtype(dtype)
        self.shape = shape

    def __repr__(self):
        return "<Placeholder {0} {1} {2}>".format(self.name, self.dtype, self.shape)


class FeedDict(dict):
    """A plain dict (placeholder -> numpy array) that can also carry the CSR
    offsets the C++ reader computed (``row_ptr``), so the model does not have
    to rebuild them on the device."""

    row_ptr = None


class BaseIterator(object):
    """Abstract base iterator class"""

    @abc.abstractmethod
    def parser_one_line(self, line):
        pass

    @abc.abstractmethod
    def load_data_from_file(self, infile):
        pass

    @abc.abstractmethod
    def _convert_data(self, labels, features):
        pass

    @abc.abstractmethod
    def gen_feed_dict(self, data_dict):
        pass


class FFMTextIterator(BaseIterator):
    """Data loader for FFM format based models, such as xDeepFM.

    The file is streamed batch by batch, so large files can be used.
    """

    def __init__(self, hparams, graph=None, col_spliter=" ", ID_spliter="%"):
        self.feature_cnt = hparams.FEATURE_COUNT
        self.field_cnt = hparams.FIELD_COUNT
        self.col_spliter = col_spliter
        self.ID_spliter = ID_spliter
        self.batch_size = hparams.batch_size
        self.graph = graph
        self._parsed = {}        # (path, size, mtime, batch size, F, N) -> [(feed_dict, impressions, n)]
        self._parsed_bytes = {}

        self.labels = Placeholder(np.float32, [None, 1], "label")
        self.fm_feat_indices = Placeholder(np.int64, [None, 2], "fm_feat_indices")
        self.fm_feat_values = Placeholder(np.float32, [None], "fm_feat_values")
        self.fm_feat_shape = Placeholder(np.int64, [None], "fm_feat_shape")
        self.dnn_feat_indices = Placeholder(np.int64, [None, 2], "dnn_feat_indices")
        self.dnn_feat_values = Placeholder(np.int64, [None], "dnn_feat_values")
        self.dnn_feat_weights = Placeholder(np.float32, [None], "dnn_feat_weights")
        self.dnn_feat_shape = Placeholder(np.int64, [None], "dnn_feat_shape")

    # ------------------------------------------------------------------ text
    def parser_one_line(self, line):
        """Parse one line into ``(label, [[field, feature, value], ...], impression_id)``.

        Field and feature indices become 0-based; the impression id is ``0`` when the
        line has no ``%id`` suffix and the stripped string otherwise.
        """
        impression_id = 0
        words = line.strip().split(self.ID_spliter)
        if len(words) == 2:
            impression_id = words[1].strip()
        cols = words[0].strip().split(self.col_spliter)
        label = float(cols[0])
        features = []
        for word in cols[1:]:
            if not word.strip():
                continue
            field, feature, value = word.split(":")[:3]
            features.append([int(field) - 1, int(feature) - 1, float(value)])
        return label, features, impression_id

    def load_data_from_file(self, infile):
        """Yield ``(feed_dict, impression_id_list, data_size)`` per mini-batch.

        Batches hold ``batch_size`` samples except a final partial one; nothing is
        shuffled.  Malformed text raises ``ValueError`` (with the line number).
        """
        if len(self.col_spliter) == 1 and len(self.ID_spliter) == 1:
            yield from self._load_cached(infile)
        else:
            yield from self._load_python(infile)

    def _load_cached(self, infile):
        """The native reader behind a cache of parsed batches: `fit()` reads its files once per
        epoch, and parsing text is an order of magnitude slower than the training step, so the
        batches of a file that fits the budget (XDFM_FFM_CACHE_MB of host memory, default 4096,
        0 = off) are kept and yielded again while the file is unchanged (same size and mtime).
        The feed dicts themselves are kept (their arrays read-only), so whatever the model attached
        to them (the batch in device memory, its sort plan) is reused as well; the model bounds
        that device memory by XDFM_FFM_CACHE_DEVICE_MB (base_model._batch_from_feed).
        Behind it, XDFM_FFM_CACHE_DIR names an on-disk cache of the same batches (_disk_cached)."""
        budget = int(os.environ.get("XDFM_FFM_CACHE_MB", "4096")) << 20
        if budget <= 0:
            yield from self._read_ahead(self._load_native(infile))
            return
        try:
            st = os.stat(infile)
        except OSError:
            yield from self._load_native(infile)  # raises the reader's own error
            return
        key = (os.path.abspath(str(infile)), st.st_size, st.st_mtime_ns, int(self.batch_size),
               int(self.field_cnt), int(self.feature_cnt), str(self.col_spliter), str(self.ID_spliter))
        hit = self._parsed.get(key)
        if hit is not None:
            yield from hit
            return
        kept, held, size = [], sum(self._parsed_bytes.values()), 0
        for item in self._disk_cached(infile, key):
            if kept is not None:
                size += sum(v.nbytes for v in item[0].values() if isinstance(v, np.ndarray))
                if held + size > budget:
                    kept = None  # too large: stream it, as the reference does
                else:
                    # handed out again by reference in later epochs: a caller that wrote into
                    # them would corrupt those epochs
                    for v in item[0].values():
                        if isinstance(v, np.ndarray):
                            v.setflags(write=False)
                    kept.append(item)
            yield item
        if kept is not None:
            for k in [k for k in self._parsed if k[0] == key[0]]:  # older versions of this file
                del self._parsed[k]
                del self._parsed_bytes[k]
            self._parsed[key] = kept
            self._parsed_bytes[key] = size

    # ---------------------------------------------------------- on-disk cache
    _DISK_VERSION = 1
    _DISK_KEYS = ("labels", "fm_feat_indices", "fm_feat_values", "fm_feat_shape", "dnn_feat_indices",
                  "dnn_feat_values", "dnn_feat_weights", "dnn_feat_shape")

    def _disk_path(self, key):
        root = os.environ.get("XDFM_FFM_CACHE_DIR", "")
        if not root:
            return None
        import hashlib

        tag = hashlib.sha1(repr((self._DISK_VERSION,) + tuple(key)).encode()).hexdigest()[:24]
        return os.path.join(root, "ffm_%s.npz" % tag)

    def _disk_cached(self, infile, key):
        """The parsed batches of a file from / to XDFM_FFM_CACHE_DIR (off when unset): a re-run whose
        file is unchanged (same path, size, mtime, batch size, FIELD_COUNT, FEATURE_COUNT, separators) parses
        no text at all, not even in its first epoch.  One uncompressed .npz per file, arrays written as
        the batches stream by and read back one batch at a time (nothing unpickled, allow_pickle off);
        the archive is renamed into place only when the whole file went through."""
        path = self._disk_path(key)
        if path is None:
            yield from self._read_ahead(self._load_native(infile))
            return
        if os.path.exists(path):
            # validated BEFORE the first batch is handed out (meta readable, every member of every
            # batch present and complete): a damaged archive is then parsed again from the start
            # instead of after some of its batches were delivered
            try:
                self._check_disk(path)
            except Exception as exc:  # damaged, truncated or foreign archive (zipfile / numpy errors)
                import warnings

                warnings.warn("ignoring the parsed-batch cache %s (%s)" % (path, exc))
            else:
                yield from self._read_ahead(self._load_disk(path))
                return
        import json
        import zipfile

        os.makedirs(os.path.dirname(path), exist_ok=True)
        tmp = "%s.tmp.%d" % (path, os.getpid())
        done = False
        try:
            with zipfile.ZipFile(tmp, "w", zipfile.ZIP_STORED, allowZip64=True) as zf:

                def put(name, arr):
                    with zf.open(name + ".npy", "w", force_zip64=True) as f:
                        np.lib.format.write_array(f, np.ascontiguousarray(arr), allow_pickle=False)

                count = 0
                for feed, imps, n in self._read_ahead(self._load_native(infile)):
                    res = {ph.name: v for ph, v in feed.items()}
                    for name, ph in zip(self._DISK_KEYS, self._placeholders()):
                        put("b%d_%s" % (count, name), res[ph.name])
                    if feed.row_ptr is not None:
                        put("b%d_row_ptr" % count, feed.row_ptr)
                    if any(i != 0 for i in imps):
                        put("b%d_imp" % count, np.frombuffer(json.dumps(imps).encode(), dtype=np.uint8))
                    count += 1
                    yield feed, imps, n
                put("meta", np.asarray([self._DISK_VERSION, count], dtype=np.int64))
                done = True
        finally:
            if done:
                os.replace(tmp, path)
            elif os.path.exists(tmp):
                os.remove(tmp)

    def _placeholders(self):
        return (self.labels, self.fm_feat_indices, self.fm_feat_values, self.fm_feat_shape,
                self.dnn_feat_indices, self.dnn_feat_values, self.dnn_feat_weights, self.dnn_feat_shape)

    def _check_disk(self, path):
        """Raises unless the archive holds meta and, for each of its batches, all eight arrays with
        sizes that match their zip entries (zip directory + .npy headers only: no array is read)."""
        import zipfile

        with zipfile.ZipFile(path) as zf:
            infos = {i.filename: i for i in zf.infolist()}
            with zf.open("meta.npy") as f:
                meta = np.lib.format.read_array(f, allow_pickle=False)
            version, count = (int(x) for x in meta)
            if version != self._DISK_VERSION:
                raise ValueError("cache version %d" % version)
            for i in range(count):
                for name in self._DISK_KEYS:
                    member = "b%d_%s.npy" % (i, name)
                    info = infos.get(member)
                    if info is None:
                        raise KeyError(member)
                    with zf.open(member) as f:
                        major, _ = np.lib.format.read_magic(f)
                        shape, _, dt = (np.lib.format.read_array_header_1_0 if major == 1
                                        else np.lib.format.read_array_header_2_0)(f)
                        if f.tell() + int(np.prod(shape, dtype=np.int64)) * dt.itemsize != info.file_size:
                            raise ValueError("%s is truncated" % member)

    def _load_disk(self, path):
        import json

        with np.load(path, allow_pickle=False) as z:
            version, count = (int(x) for x in z["meta"])
            if version != self._DISK_VERSION:
                raise ValueError("cache version %d" % version)
            names = set(z.files)
            for i in range(count):
                res = {name: z["b%d_%s" % (i, name)] for name in self._DISK_KEYS}
                feed = self.gen_feed_dict(res)
                if "b%d_row_ptr" % i in names:
                    feed.row_ptr = z["b%d_row_ptr" % i]
                n = int(res["labels"].shape[0])
                if "b%d_imp" % i in names:
                    imps = json.loads(z["b%d_imp" % i].tobytes().decode())
                else:
                    imps = [0] * n
                yield feed, imps, n

    @staticmethod
    def _read_ahead(gen, depth=2):
        """Run a batch generator in a thread, `depth` batches ahead of its consumer: the native
        reader parses (outside the GIL) while the caller uploads and enqueues the previous batch."""
        import queue
        import threading

        q = queue.Queue(maxsize=depth)
        stop = threading.Event()
        end = object()

        def put(item):
            while not stop.is_set():
                try:
                    q.put(item, timeout=0.1)
                    return True
                except queue.Full:
                    continue
            return False

        def work():
            try:
                for item in gen:
                    if not put((item, None)):
                        return
                put((end, None))
            except BaseException as exc:  # noqa: BLE001 -- handed to the consumer
                put((end, exc))
            finally:
                gen.close()

        t = threading.Thread(target=work, name="xdfm-ffm-reader", daemon=True)
        t.start()
        try:
            while True:
                item, exc = q.get()
                if item is end:
                    if exc is not None:
                        raise exc
                    return
                yield item
        finally:
            stop.set()
            t.join(timeout=5)

    def _load_python(self, infile):
        label_list, features_list, impression_id_list = [], [], []
        with open(infile, "r") as rd:
            for line in rd:
                label, features, impression_id = self.parser_one_line(line)
                features_list.append(features)
                label_list.append(label)
                impression_id_list.append(impression_id)
                if len(label_list) == self.batch_size:
                    res = self._convert_data(label_list, features_list)
                    yield self.gen_feed_dict(res), impression_id_list, self.batch_size
                    label_list, features_list, impression_id_list = [], [], []
            if label_list:
                res = self._convert_data(label_list, features_list)
                yield self.gen_feed_dict(res), impression_id_list, len(label_list)

    def _load_native(self, infile):
        lib = _lib.lib
        handle = ctypes.c_void_p()
        try:
            _lib.check(
                lib.xdfm_ffm_open(
                    str(infile).encode(),
                    int(self.field_cnt),
                    int(self.feature_cnt),
                    self.col_spliter.encode(),
                    self.ID_spliter.encode(),
                    ctypes.byref(handle),
                )
            )
        except _lib.XdfmError as exc:
            if exc.code == -5:
                raise FileNotFoundError(exc.message)
            raise ValueError(exc.message)
        info = _lib.BatchInfo()
        try:
            while True:
                rc = lib.xdfm_ffm_next(handle, int(self.batch_size), ctypes.byref(info))
                if rc == -6:  # XDFM_E_EOF
                    break
                if rc < 0:
                    raise ValueError(lib.xdfm_last_error().decode("utf-8", "replace"))
                yield self._fill_native(handle, info)
        finally:
            lib.xdfm_ffm_close(handle)

    def _fill_native(self, handle, info):
        n, nnz = int(info.n_rows), int(info.nnz)
        labels = np.empty((n, 1), dtype=np.float32)
        fm_idx = np.empty((nnz, 2), dtype=np.int64)
        fm_val = np.empty((nnz,), dtype=np.float32)
        dnn_idx = np.empty((nnz, 2), dtype=np.int64)
        dnn_val = np.empty((nnz,), dtype=np.int64)
        dnn_w = np.empty((nnz,), dtype=np.float32)
        imp_off = np.empty((n + 1,), dtype=np.int64)
        imp_chars = np.empty((max(int(info.imp_bytes), 1),), dtype=np.uint8)
        imp_present = np.empty((n,), dtype=np.uint8)
        row_ptr = np.empty((n * int(self.field_cnt) + 1,), dtype=np.int32)

        def p(a):
            return a.ctypes.data_as(ctypes.c_void_p)

        rc = _lib.lib.xdfm_ffm_fill(
            handle, p(labels), p(fm_idx), p(fm_val), p(dnn_idx), p(dnn_val), p(dnn_w),
            p(row_ptr), p(imp_off), p(imp_chars), p(imp_present),
        )  # fmt: skip
        if rc == -1:
            # a field index outside [1, FIELD_COUNT]: the arrays are still produced the
            # way the reference produces them (rows alias), only the CSR offsets are not.
            row_ptr = None
            _lib.check(
                _lib.lib.xdfm_ffm_fill(
                    handle, p(labels), p(fm_idx), p(fm_val), p(dnn_idx), p(dnn_val), p(dnn_w),
                    None, p(imp_off), p(imp_chars), p(imp_present),
                )  # fmt: skip
            )
        else:
            _lib.check(rc)
        if nnz == 0:
            # np.asarray([], dtype)[[]] in the reference has shape (0,), not (0, 2)
            fm_idx = fm_idx.reshape(0)
            dnn_idx = dnn_idx.reshape(0)
        if imp_present.any():
            raw = imp_chars.tobytes()
            impression_ids = [
                raw[imp_off[i] : imp_off[i + 1]].decode("utf-8", "replace") if imp_present[i] else 0
                for i in range(n)
            ]
        else:
            impression_ids = [0] * n
        res = {
            "labels": labels,
            "fm_feat_indices": fm_idx,
            "fm_feat_values": fm_val,
            "fm_feat_shape": np.asarray([n, self.feature_cnt], dtype=np.int64),
            "dnn_feat_indices": dnn_idx,
            "dnn_feat_values": dnn_val,
            "dnn_feat_weights": dnn_w,
            "dnn_feat_shape": np.asarray(
                [n * self.field_cnt, int(info.dnn_width)], dtype=np.int64
            ),
        }
        feed = self.gen_feed_dict(res)
        feed.row_ptr = row_ptr
        return feed, impression_ids, n

    # ------------------------------------------------------------- in-memory
    def _convert_data(self, labels, features):
        """Python lists -> the eight numpy arrays (vectorised; same result as the
        reference's nested loops + ``sorted``)."""
        n = len(labels)
        field_count = self.field_cnt
        counts = [len(f) for f in features]
        nnz = int(sum(counts))
        sample = np.repeat(np.arange(n, dtype=np.int64), counts)
        if nnz:
            flat = [tok for feats in features for tok in feats]
            field = np.asarray([t[0] for t in flat], dtype=np.int64)
            feat = np.asarray([t[1] for t in flat], dtype=np.int64)
            value64 = np.asarray([t[2] for t in flat], dtype=np.float64)
        else:
            field = feat = np.zeros((0,), dtype=np.int64)
            value64 = np.zeros((0,), dtype=np.float64)
        value = value64.astype(np.float32)

        # running count of (sample, field) in file order = column of the dnn view
        row = sample * field_count + field
        key_order = np.lexsort((np.arange(nnz), field, sample))  # stable by (sample, field)
        s_sample, s_field = sample[key_order], field[key_order]
        new_group = np.ones(nnz, dtype=bool)
        if nnz:
            new_group[1:] = (s_sample[1:] != s_sample[:-1]) | (s_field[1:] != s_field[:-1])
        group_start = np.maximum.accumulate(np.where(new_group, np.arange(nnz), 0))
        col = np.empty(nnz, dtype=np.int64)
        col[key_order] = np.arange(nnz) - group_start
        width = int(col.max()) + 1 if nnz else 0

        order = np.lexsort((col, row))  # stable sort by (row, col)

        res = {}
        if nnz:
            res["fm_feat_indices"] = np.stack([sample, feat], axis=1)
            res["dnn_feat_indices"] = np.stack([row, col], axis=1)[order]
        else:
            res["fm_feat_indices"] = np.zeros((0,), dtype=np.int64)
            res["dnn_feat_indices"] = np.zeros((0,), dtype=np.int64)
        res["fm_feat_values"] = value
        res["fm_feat_shape"] = np.asarray([n, self.feature_cnt], dtype=np.int64)
        res["labels"] = np.asarray(labels, dtype=np.float32).reshape(n, 1)
        res["dnn_feat_values"] = feat[order]
        res["dnn_feat_weights"] = value[order]
        res["dnn_feat_shape"] = np.asarray([n * field_count, width], dtype=np.int64)
        return res

    def gen_feed_dict(self, data_dict):
        """Map the placeholder handles to the arrays of ``data_dict``."""
        feed = FeedDict()
        feed[self.labels] = data_dict["labels"]
        feed[self.fm_feat_indices] = data_dict["fm_feat_indices"]
        feed[self.fm_feat_values] = data_dict["fm_feat_values"]
        feed[self.fm_feat_shape] = data_dict["fm_feat_shape"]
        feed[self.dnn_feat_indices] = data_dict["dnn_feat_indices"]
        feed[self.dnn_feat_values] = data_dict["dnn_feat_values"]
        feed[self.dnn_feat_weights] = data_dict["dnn_feat_weights"]
        feed[self.dnn_feat_shape] = data_dict["dnn_feat_shape"]
        return feed
