"""The hand-written stable LSD radix sort behind xdfm_embed_plan (csrc/radix_sort.hip), through the
C ABI, against numpy's stable argsort -- bit-exact, as for all index work.  Cases: every digit split
(1, 2, 3 and 4 passes), ragged sizes around the 2048-key sub-tile, blocks that loop over several
sub-tiles (n > 2048 * 512), heavy duplicates (the 13 hot ids of Criteo-shaped data), out-of-range ids
(clamped, flagged elsewhere)."""
import ctypes

import numpy as np
import pytest
import torch

from recommenders_amd import _lib
from recommenders_amd.synth import criteo_shape_ids

pytestmark = pytest.mark.gpu


def _plan(ids, N):
    n = int(ids.numel())
    sz = ctypes.c_size_t()
    _lib.check(_lib.lib.xdfm_embed_plan_workspace_bytes(n, N, ctypes.byref(sz)))
    ws = torch.empty((sz.value,), dtype=torch.uint8, device="cuda")
    ws.fill_(0xAB)  # nothing in the workspace has to be initialised
    sk = torch.full((max(n, 1),), -7, dtype=torch.int32, device="cuda")
    st = torch.full((max(n, 1),), -7, dtype=torch.int32, device="cuda")
    _lib.check(_lib.lib.xdfm_embed_plan(ids.data_ptr(), n, N, sk.data_ptr(), st.data_ptr(), ws.data_ptr(),
                                        ws.numel(), torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    return sk[:n].cpu().numpy(), st[:n].cpu().numpy()


CASES = [
    (1, 1), (5, 3), (64, 2), (2047, 500), (2048, 1 << 25), (2049, 1000), (100000, 600), (4097, 1 << 9),
    (70000, (1 << 18) + 1), (319488, 1 << 25), (250001, 1 << 31), (2048 * 512 + 777, 1 << 27),
]


@pytest.mark.parametrize("n,N", CASES)
def test_plan_equals_numpy_stable_argsort(n, N):
    rng = np.random.default_rng(n * 31 + N % 1000)
    ids = rng.integers(0, N, n, dtype=np.int64)
    if n >= 5000:  # a few very hot ids and a run of equal neighbours
        ids[rng.random(n) < 0.3] = ids[0]
        ids[n // 2:n // 2 + 3000] = ids[1]
    sk, st = _plan(torch.from_numpy(ids).cuda(), N)
    order = np.argsort(ids, kind="stable")
    assert np.array_equal(st, order.astype(np.int32))
    assert np.array_equal(sk.astype(np.int64) & 0xFFFFFFFF, ids[order])


def test_plan_on_criteo_shaped_batch_and_clamped_ids():
    N, B = 1 << 25, 8192
    ids = criteo_shape_ids(np.random.default_rng(5), B, N).astype(np.int64).reshape(-1)
    sk, st = _plan(torch.from_numpy(ids).cuda(), N)
    order = np.argsort(ids, kind="stable")
    assert np.array_equal(st, order.astype(np.int32)) and np.array_equal(sk, ids[order].astype(np.int32))
    bad = ids.copy()
    bad[::1000] = -5
    bad[1::1000] = N + 17
    sk, st = _plan(torch.from_numpy(bad).cuda(), N)
    clamped = np.clip(bad, 0, N - 1)
    order = np.argsort(clamped, kind="stable")
    assert np.array_equal(st, order.astype(np.int32)) and np.array_equal(sk, clamped[order].astype(np.int32))


def test_plan_rejects_a_short_workspace():
    ids = torch.zeros(4096, dtype=torch.int64, device="cuda")
    out = torch.empty(4096, dtype=torch.int32, device="cuda")
    ws = torch.empty(4096, dtype=torch.uint8, device="cuda")
    rc = _lib.lib.xdfm_embed_plan(ids.data_ptr(), 4096, 100, out.data_ptr(), out.data_ptr(), ws.data_ptr(),
                                  ws.numel(), torch.cuda.current_stream().cuda_stream)
    assert rc == -3
