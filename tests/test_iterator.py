"""FFMTextIterator: oracle and product against fixtures produced by the reference's own iterator
(tests/golden/gen_golden.py), and product against oracle on synthetic ragged files.
The C++ reader lives in libxdfm.so but is host code: no GPU needed."""
import numpy as np
import pytest

from oracle import ffm_iterator_oracle as O
from recommenders_amd.models.deeprec.io.iterator import FFMTextIterator
from recommenders_amd.synth import write_plumbing_sample
from tests.util import make_hp

KEYS = ["labels", "fm_feat_indices", "fm_feat_values", "fm_feat_shape", "dnn_feat_indices",
        "dnn_feat_values", "dnn_feat_weights", "dnn_feat_shape"]
PH = {"labels": "label"}


def _expect(arr):
    return np.asarray(arr["data"], dtype=arr["dtype"]).reshape(arr["shape"])


def _check_batch(got, exp_arrays, ctx):
    for k in KEYS:
        e = _expect(exp_arrays[PH.get(k, k)])
        g = np.asarray(got[k])
        assert g.dtype == e.dtype, (ctx, k, g.dtype, e.dtype)
        assert g.shape == e.shape, (ctx, k, g.shape, e.shape)
        assert np.array_equal(g, e), (ctx, k)


def _write(tmp_path, name, text):
    p = tmp_path / (name + ".txt")
    with open(p, "w", newline="") as f:
        f.write(text)
    return str(p)


def test_oracle_matches_reference_fixtures(golden, tmp_path):
    cases = golden("ffm_iterator_golden.json")
    assert "survey_8c3" in cases
    for name, case in cases.items():
        path = _write(tmp_path, name, case["text"])
        out = list(O.batches(path, case["batch_size"], case["FIELD_COUNT"], case["FEATURE_COUNT"]))
        assert len(out) == len(case["batches"])
        for (arrays, imps, n), exp in zip(out, case["batches"]):
            assert n == exp["n"] and imps == exp["impression_ids"]
            _check_batch(arrays, exp["arrays"], name)


@pytest.mark.parametrize("native", [True, False])
def test_product_matches_reference_fixtures(golden, tmp_path, native):
    cases = golden("ffm_iterator_golden.json")
    for name, case in cases.items():
        path = _write(tmp_path, name, case["text"])
        hp = make_hp(FEATURE_COUNT=case["FEATURE_COUNT"], FIELD_COUNT=case["FIELD_COUNT"],
                     batch_size=case["batch_size"])
        it = FFMTextIterator(hp, None)
        gen = it._load_native(path) if native else it._load_python(path)
        out = list(gen)
        assert len(out) == len(case["batches"]), name
        for (feed, imps, n), exp in zip(out, case["batches"]):
            assert isinstance(feed, dict)
            assert n == exp["n"] and list(imps) == exp["impression_ids"], name
            got = {k: feed[getattr(it, k)] for k in KEYS}
            _check_batch(got, exp["arrays"], name)


def test_survey_vector_literal(tmp_path):
    """The outputs SURVEY.md 8(c)(3) records for the reference's iterator, typed in by hand."""
    path = _write(tmp_path, "s", "1 1:1:1.0 2:4:0.5 2:5:0.25 3:9:1\n0 1:2:1 3:10:2.0%imp7\n1 2:6:1 1:3:1\n")
    hp = make_hp(FEATURE_COUNT=10, FIELD_COUNT=3, batch_size=2)
    it = FFMTextIterator(hp, None)
    (f1, imp1, n1), (f2, imp2, n2) = list(it.load_data_from_file(path))
    assert (n1, imp1) == (2, [0, "imp7"]) and (n2, imp2) == (1, [0])
    assert f1[it.fm_feat_indices].tolist() == [[0, 0], [0, 3], [0, 4], [0, 8], [1, 1], [1, 9]]
    assert f1[it.fm_feat_values].tolist() == [1, 0.5, 0.25, 1, 1, 2]
    assert f1[it.fm_feat_shape].tolist() == [2, 10]
    assert f1[it.dnn_feat_indices].tolist() == [[0, 0], [1, 0], [1, 1], [2, 0], [3, 0], [5, 0]]
    assert f1[it.dnn_feat_values].tolist() == [0, 3, 4, 8, 1, 9]
    assert f1[it.dnn_feat_weights].tolist() == [1, 0.5, 0.25, 1, 1, 2]
    assert f1[it.dnn_feat_shape].tolist() == [6, 2]
    assert f1[it.labels].tolist() == [[1], [0]]
    assert f2[it.fm_feat_indices].tolist() == [[0, 5], [0, 2]]
    assert f2[it.dnn_feat_indices].tolist() == [[0, 0], [1, 0]]
    assert f2[it.dnn_feat_values].tolist() == [2, 5]
    assert f2[it.dnn_feat_shape].tolist() == [3, 1]
    assert f1.row_ptr.tolist() == [0, 1, 3, 4, 5, 5, 6]


def test_native_reader_vs_oracle_on_plumbing_sample(tmp_path):
    path = write_plumbing_sample(str(tmp_path / "s1.txt"), n=700, field_count=10, feature_count=1000)
    hp = make_hp(FEATURE_COUNT=1000, FIELD_COUNT=10, batch_size=128)
    it = FFMTextIterator(hp, None)
    got = list(it.load_data_from_file(path))
    exp = list(O.batches(path, 128, 10, 1000))
    assert len(got) == len(exp) == 6 and got[-1][2] == 700 - 5 * 128
    for (feed, imps, n), (arrays, eimps, en) in zip(got, exp):
        assert n == en and list(imps) == eimps
        for k in KEYS:
            g, e = feed[getattr(it, k)], arrays[k]
            assert g.dtype == e.dtype and g.shape == e.shape and np.array_equal(g, e), k
        # the CSR offsets are the row histogram of the dnn view
        rows = arrays["dnn_feat_indices"][:, 0]
        ptr = np.concatenate([[0], np.cumsum(np.bincount(rows, minlength=n * 10))])
        assert np.array_equal(feed.row_ptr, ptr)


@pytest.mark.parametrize("threads", ["1", "5"])
def test_native_reader_pieces_parsed_on_several_threads_equal_the_oracle(tmp_path, monkeypatch, threads):
    """Batches large enough to be split between parsing threads (a piece is at least 256 lines):
    ragged lines, fields out of order inside a line (the dnn view then needs its stable sort),
    multi-hot fields, impression ids on some lines, a partial last batch, and a file whose last
    batches come sorted -- each piece writes its own stretch of the arrays and of the CSR offsets."""
    from tests.util import random_features, write_libffm

    monkeypatch.setenv("XDFM_PARSE_THREADS", threads)
    monkeypatch.setenv("XDFM_FFM_CACHE_MB", "0")
    rng = np.random.default_rng(7)
    F, N, B = 9, 300, 1500
    labels, feats = random_features(rng, 3 * B + 700, F, N, multi_p=0.3, missing_p=0.2)
    for toks in feats[2 * B:]:  # the tail of the file in field order: no sort needed there
        toks.sort(key=lambda t: t[0])
    path = write_libffm(str(tmp_path / "big.txt"), labels, feats, imp_every=3)
    hp = make_hp(FEATURE_COUNT=N, FIELD_COUNT=F, batch_size=B)
    it = FFMTextIterator(hp, None)
    got = list(it.load_data_from_file(path))
    exp = list(O.batches(path, B, F, N))
    assert len(got) == len(exp) == 4
    for (feed, imps, n), (arrays, eimps, en) in zip(got, exp):
        assert n == en and list(imps) == eimps
        for k in KEYS:
            g, e = feed[getattr(it, k)], arrays[k]
            assert g.dtype == e.dtype and g.shape == e.shape and np.array_equal(g, e), k
        rows = arrays["dnn_feat_indices"][:, 0]
        ptr = np.concatenate([[0], np.cumsum(np.bincount(rows, minlength=n * F))])
        assert np.array_equal(feed.row_ptr, ptr)
    # an error inside a later piece is reported with its line number
    with open(path, "a") as f:
        f.write("1 2:3:x\n")
    with pytest.raises(ValueError, match="line %d" % (3 * B + 701)):
        list(FFMTextIterator(hp, None).load_data_from_file(path))


def test_parser_one_line_and_errors(tmp_path):
    hp = make_hp(FEATURE_COUNT=10, FIELD_COUNT=3, batch_size=2)
    it = FFMTextIterator(hp, None)
    assert it.parser_one_line("1 1:2:0.5 3:4:1%abc \n") == (1.0, [[0, 1, 0.5], [2, 3, 1.0]], "abc")
    assert it.parser_one_line("0  2:2:1e-1") == (0.0, [[1, 1, 0.1]], 0)
    assert it.parser_one_line("1 1:1:1%a%b")[2] == 0  # two separators: no impression id
    assert O.parse_line("1 1:2:0.5 3:4:1%abc \n") == (1.0, [(0, 1, 0.5), (2, 3, 1.0)], "abc")
    for bad in ("x 1:1:1\n", "1 1:1\n", "1 a:1:1\n", "1 1:1:zz\n"):
        p = _write(tmp_path, "bad", bad)
        with pytest.raises(ValueError):
            list(it.load_data_from_file(p))
        with pytest.raises(ValueError):
            list(it._load_python(p))
    with pytest.raises(FileNotFoundError):
        list(it.load_data_from_file(str(tmp_path / "nope.txt")))
    empty = _write(tmp_path, "empty", "")
    assert list(it.load_data_from_file(empty)) == []


def test_field_out_of_range_aliases_like_reference(tmp_path):
    # iterator.py does not validate: field >= F aliases into later rows; arrays must still match
    p = _write(tmp_path, "alias", "1 5:1:1 1:2:1\n0 1:3:1\n")
    hp = make_hp(FEATURE_COUNT=10, FIELD_COUNT=3, batch_size=2)
    it = FFMTextIterator(hp, None)
    (feed, _, n), = list(it.load_data_from_file(p))
    (arrays, _, _), = list(O.batches(p, 2, 3, 10))
    for k in KEYS:
        assert np.array_equal(feed[getattr(it, k)], arrays[k]), k
    assert feed.row_ptr is None or len(feed.row_ptr) == n * 3 + 1


def test_parsed_batches_are_kept_while_the_file_is_unchanged(tmp_path, monkeypatch):
    """Second pass over an unchanged file: the very same feed dicts (nothing is parsed again);
    a rewritten file is parsed again; XDFM_FFM_CACHE_MB=0 streams every time."""
    import os

    from recommenders_amd.models.deeprec.io.iterator import FFMTextIterator
    from tests.util import make_hp, random_features, write_libffm

    hp = make_hp(batch_size=16)
    rng = np.random.default_rng(3)
    labels, feats = random_features(rng, 40, hp.FIELD_COUNT, hp.FEATURE_COUNT)
    path = write_libffm(str(tmp_path / "a.ffm"), labels, feats)
    it = FFMTextIterator(hp, None)
    first = list(it.load_data_from_file(path))
    second = list(it.load_data_from_file(path))
    assert [n for (_, _, n) in first] == [16, 16, 8]
    assert all(a[0] is b[0] for a, b in zip(first, second)) and len(first) == len(second)
    # same content as an iterator that never saw the file
    fresh = list(FFMTextIterator(hp, None).load_data_from_file(path))
    for (fa, ia, na), (fb, ib, nb) in zip(second, fresh):
        assert na == nb and ia == ib
        for (ka, va), (kb, vb) in zip(sorted(fa.items(), key=lambda kv: kv[0].name),
                                      sorted(fb.items(), key=lambda kv: kv[0].name)):
            assert ka.name == kb.name and np.array_equal(va, vb)
    # the file changes: parsed again
    labels2, feats2 = random_features(rng, 20, hp.FIELD_COUNT, hp.FEATURE_COUNT)
    write_libffm(path, labels2, feats2)
    os.utime(path, ns=(1, 1))
    third = list(it.load_data_from_file(path))
    assert [n for (_, _, n) in third] == [16, 4] and third[0][0] is not first[0][0]
    monkeypatch.setenv("XDFM_FFM_CACHE_MB", "0")
    it2 = FFMTextIterator(hp, None)
    a = list(it2.load_data_from_file(path))
    b = list(it2.load_data_from_file(path))
    assert a[0][0] is not b[0][0] and not it2._parsed


def test_parsed_batches_on_disk_serve_a_second_run_without_parsing(tmp_path, monkeypatch):
    """XDFM_FFM_CACHE_DIR: a second iterator (a re-run) over an unchanged file gets the same arrays,
    impression ids and CSR offsets from the archive and never opens the text; a changed file is
    parsed again; an interrupted first pass leaves no archive behind; a damaged archive is ignored."""
    import os

    from tests.util import random_features, write_libffm

    cache = tmp_path / "cache"
    monkeypatch.setenv("XDFM_FFM_CACHE_DIR", str(cache))
    hp = make_hp(batch_size=16)
    rng = np.random.default_rng(5)
    labels, feats = random_features(rng, 40, hp.FIELD_COUNT, hp.FEATURE_COUNT)
    path = write_libffm(str(tmp_path / "a.ffm"), labels, feats, imp_every=3)

    def same(xs, ys):
        assert len(xs) == len(ys)
        for (fa, ia, na), (fb, ib, nb) in zip(xs, ys):
            assert na == nb and ia == ib
            assert (fa.row_ptr is None) == (fb.row_ptr is None)
            assert fa.row_ptr is None or np.array_equal(fa.row_ptr, fb.row_ptr)
            for (ka, va), (kb, vb) in zip(sorted(fa.items(), key=lambda kv: kv[0].name),
                                          sorted(fb.items(), key=lambda kv: kv[0].name)):
                assert ka.name == kb.name and va.dtype == vb.dtype and va.shape == vb.shape
                assert np.array_equal(va, vb)

    # a pass that stops early writes nothing
    gen = FFMTextIterator(hp, None).load_data_from_file(path)
    next(gen)
    gen.close()
    assert not cache.exists() or not os.listdir(cache)
    first = list(FFMTextIterator(hp, None).load_data_from_file(path))
    files = os.listdir(cache)
    assert len(files) == 1 and files[0].endswith(".npz")
    it = FFMTextIterator(hp, None)

    def no_text(_):
        raise AssertionError("the text was parsed again")

    monkeypatch.setattr(it, "_load_native", no_text)
    second = list(it.load_data_from_file(path))
    same(first, second)
    assert any(i != 0 for (_, imps, _) in second for i in imps)
    # the file changes: parsed again, a second archive
    labels2, feats2 = random_features(rng, 20, hp.FIELD_COUNT, hp.FEATURE_COUNT)
    write_libffm(path, labels2, feats2)
    third = list(FFMTextIterator(hp, None).load_data_from_file(path))
    assert [n for (_, _, n) in third] == [16, 4] and len(os.listdir(cache)) == 2
    # a damaged archive is ignored (with a warning) and the text parsed
    for f in os.listdir(cache):
        with open(cache / f, "wb") as fh:
            fh.write(b"not a zip")
    with pytest.warns(UserWarning):
        again = list(FFMTextIterator(hp, None).load_data_from_file(path))
    same(third, again)
    # a TRUNCATED archive (its directory gone, or one member cut short) is found out before the first
    # batch is handed out: every batch is delivered exactly once, from the text
    files = sorted(os.listdir(cache))
    for f in files:
        os.remove(cache / f)
    full = list(FFMTextIterator(hp, None).load_data_from_file(path))
    (arch,) = os.listdir(cache)
    blob = open(cache / arch, "rb").read()
    with open(cache / arch, "wb") as fh:
        fh.write(blob[: len(blob) * 2 // 3])
    with pytest.warns(UserWarning):
        cut = list(FFMTextIterator(hp, None).load_data_from_file(path))
    same(full, cut)
    # another separator is another key: no archive of the " " / "%" file is taken for it
    other = FFMTextIterator(hp, None, col_spliter=" ", ID_spliter="#")
    before = set(os.listdir(cache))
    list(other.load_data_from_file(path))
    assert len(set(os.listdir(cache)) - before) == 1


@pytest.mark.parametrize("newline", ["\n", "\r\n", "\r"])
@pytest.mark.parametrize("threads", ["1", "4"])
def test_one_pass_lines_and_their_fallbacks_equal_the_oracle(tmp_path, monkeypatch, newline, threads):
    """The reader takes a canonical line (single separators, plain decimals of at most 15 digits, fields in
    range, no impression id) in one pass over its bytes and everything else through the general path; line
    ends come from an index built by all threads unless the file holds a '\\r'.  Every kind of line that must
    fall back, between canonical ones, in files of each newline convention, the last line without one."""
    monkeypatch.setenv("XDFM_PARSE_THREADS", threads)
    monkeypatch.setenv("XDFM_FFM_CACHE_MB", "0")
    rng = np.random.default_rng(11)
    F, N, B = 6, 1000, 700
    odd = [
        "1 1:2:3  4:5:6",                      # two separators in a row (an empty word)
        "0 1:7:1.5 %imp42",                    # an impression id
        "1 +1:8:2 2:9:-0.25",                  # a signed index
        "0 1:1_0:1 2:3:1e-2",                  # underscore, exponent
        "1 2:4:0.1234567890123456 3:5:1",      # 16 significant digits: not the short form
        "0 3:6:.5 4:7:5. 5:8:00012.50",        # bare fractions, leading zeros
        "1 6:9:1:extra 1:2:3",                 # a fourth component (ignored)
        "  0 1:2:3 ",                          # spaces around the line
        "1 2:2:1 2:3:1 2:4:1 1:9:1",           # multi-hot field, fields out of order
        "0 1:999999999999999999:0",            # an 18-digit feature id
        "1.0 1:2:3\t",                         # a tab at the end (stripped like any whitespace)
        "-0 1:2:-0",                           # negative zeros
        "1",                                   # no tokens at all
    ]
    lines = []
    for i in range(2 * B + 123):
        if i % 7 == 3:
            lines.append(odd[(i // 7) % len(odd)])
        else:
            toks = ["%d:%d:%s" % (f + 1, rng.integers(1, N + 1), "1" if f > 2 else "%.4f" % rng.random())
                    for f in range(F) if rng.random() > 0.15]
            lines.append("%d %s" % (i & 1, " ".join(toks)))
    path = str(tmp_path / "mix.txt")
    with open(path, "w", newline="") as f:
        f.write(newline.join(lines))  # (no newline after the last line)
    hp = make_hp(FEATURE_COUNT=N, FIELD_COUNT=F, batch_size=B)
    it = FFMTextIterator(hp, None)
    got = list(it.load_data_from_file(path))
    exp = list(O.batches(path, B, F, N))
    assert len(got) == len(exp) == 3 and sum(g[2] for g in got) == len(lines)
    for (feed, imps, n), (arrays, eimps, en) in zip(got, exp):
        assert n == en and list(imps) == eimps
        for k in KEYS:
            g, e = feed[getattr(it, k)], arrays[k]
            assert g.dtype == e.dtype and g.shape == e.shape and np.array_equal(g, e), k
    # a field outside [1, FIELD_COUNT] inside an otherwise canonical line aliases like the reference
    with open(path, "a", newline="") as f:
        f.write(newline + "1 1:2:3 %d:4:5 0:6:7" % (F + 2))
    it2 = FFMTextIterator(hp, None)
    got = list(it2.load_data_from_file(path))
    exp = list(O.batches(path, B, F, N))
    for k in KEYS:
        assert np.array_equal(got[-1][0][getattr(it2, k)], exp[-1][0][k]), k
