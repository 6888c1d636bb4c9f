#!/usr/bin/env python
"""xdfm_embed_fwd alone on the bench workload with COLD rows: 64 distinct Criteo-shaped batches
(8192 x 39 tokens) over a 2^25-row table kept as 256-byte records, cycled so that the rows touched
between two uses of a batch (1.6 GB) exceed the 256 MB Infinity Cache.  Prints us per launch and
the algorithmic GB/s ((8 D + 12) bytes per token)."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recommenders_amd import _lib  # noqa: E402
from recommenders_amd.synth import criteo_shape_ids  # noqa: E402

N, D, B, F, NB = 1 << 25, 16, 8192, 39, 64
dev = torch.device("cuda:0")
rec = torch.rand(N, 64, device=dev)
rng = np.random.default_rng(0)
ids = [torch.from_numpy(criteo_shape_ids(rng, B, N).astype(np.int64).reshape(-1)).to(dev) for _ in range(NB)]
w = torch.rand(B * F, device=dev)
row_ptr = torch.arange(B * F + 1, dtype=torch.int32, device=dev)
emb, lin, fm, fms = (torch.empty(B * F * D, device=dev), torch.empty(B, device=dev), torch.empty(B, device=dev),
                     torch.empty(B * D, device=dev))
err = torch.zeros(1, dtype=torch.int32, device=dev)
st = torch.cuda.current_stream().cuda_stream


def run(k):
    _lib.check(_lib.lib.xdfm_embed_fwd(rec.data_ptr(), rec[:, D].data_ptr(), 64, 64, N, D, F, B, row_ptr.data_ptr(),
                                       ids[k % NB].data_ptr(), w.data_ptr(), B * F, emb.data_ptr(), lin.data_ptr(),
                                       fm.data_ptr(), fms.data_ptr(), err.data_ptr(), st))


for k in range(NB):
    run(k)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 4 * NB
e0.record()
for k in range(reps):
    run(k)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / reps * 1e3
print("embed_fwd, cold rows (64 batches): %.2f us per launch, %.0f GB/s algorithmic = %.3f of 8 TB/s"
      % (us, B * F * (8 * D + 12) / us / 1e3, B * F * (8 * D + 12) / us / 1e3 / 8000))
e0.record()
for k in range(reps):
    run(0)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / reps * 1e3
print("embed_fwd, one batch over and over (rows in the Infinity Cache): %.2f us per launch, %.0f GB/s" % (
    us, B * F * (8 * D + 12) / us / 1e3))
assert int(err.item()) == 0
