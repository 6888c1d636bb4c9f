#!/usr/bin/env python
"""How fast can this GPU gather random 64-byte rows?  torch.index_select on the bench's id
distribution and on uniform ids, next to xdfm_embed_fwd (GPU box)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recommenders_amd.synth import criteo_shape_ids  # noqa: E402

dev = torch.device("cuda:0")
N, D, B, F = 1 << 25, 16, 8192, 39
table = torch.rand(N, D, device=dev)
rng = np.random.default_rng(0)
ids_z = torch.from_numpy(criteo_shape_ids(rng, B, N).astype(np.int64).reshape(-1)).to(dev)
ids_u = torch.randint(0, N, (B * F,), device=dev)


def timeit(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for name, ids in (("criteo-shape ids", ids_z), ("uniform ids", ids_u)):
    out = torch.empty(ids.numel(), D, device=dev)
    us = timeit(lambda: torch.index_select(table, 0, ids, out=out))
    nbytes = ids.numel() * (2 * D * 4 + 8)
    print("%-18s index_select %.1f us  %.0f GB/s (rows in + rows out + ids), unique rows %d"
          % (name, us, nbytes / us / 1e3, ids.unique().numel()))
# contiguous copy of the same volume for scale
src = torch.rand(B * F, D, device=dev)
dst = torch.empty_like(src)
us = timeit(lambda: dst.copy_(src))
print("contiguous copy of the same rows %.1f us  %.0f GB/s" % (us, 2 * src.numel() * 4 / us / 1e3))
big = torch.rand(1 << 28, device=dev)
big2 = torch.empty_like(big)
us = timeit(lambda: big2.copy_(big), reps=10)
print("1 GiB copy %.1f us  %.0f GB/s" % (us, 2 * big.numel() * 4 / us / 1e3))
