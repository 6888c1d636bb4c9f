#!/usr/bin/env python
"""The embedding kernels alone on the bench workload with COLD rows: 64 distinct Criteo-shaped batches
(8192 x 39 tokens) over a 2^25-row table kept as 256-byte records, cycled so that the rows touched
between two uses of a batch (1.6 GB) exceed the 256 MB Infinity Cache.  The table is brought into a
trained state first (64 steps: every touched row has slots and a row_step up to 64 steps old), then
xdfm_embed_fwd is timed without and with row ageing and xdfm_embed_bwd (mode 0) with it.
Prints us per launch and the algorithmic GB/s."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from recommenders_amd import _lib  # noqa: E402
from recommenders_amd.models.deeprec.models._engine import XDeepFMEngine, _ptr  # noqa: E402


class A(object):
    features, dim, batch, workload, dtype = 1 << 25, 16, 8192, "criteo", "fp32"


NB = int(os.environ.get("UB_BATCHES", "64"))
hp = bench.workload_hparams(A)
dev = torch.device("cuda:0")
eng = XDeepFMEngine(hp, device=dev, seed=7, embedding_update="sparse_rows")
host = bench.synth_batches(hp, NB, seed=1234)
uniq = float(np.mean([len(np.unique(hb["dnn_feat_values"])) for hb in host]))
batches = [eng.upload(hb["labels"], hb["dnn_feat_indices"], hb["dnn_feat_values"], hb["dnn_feat_weights"]) for hb in host]
for b in batches:
    eng.train_step(b)
torch.cuda.synchronize()
eng.check_errors()
B, F, D, nnz = hp.batch_size, hp.FIELD_COUNT, hp.dim, batches[0].nnz
ws, st = eng.ws, eng._stream()


def fwd(b, age):
    _lib.check(_lib.lib.xdfm_embed_fwd(_ptr(eng.table), _ptr(eng.lin_w), eng.table_stride, eng.lin_stride, eng.N, D, F, B,
                                       _ptr(b.row_ptr), _ptr(b.ids), _ptr(b.weights), b.nnz, _ptr(ws["emb_out"]),
                                       _ptr(ws["lin_out"]), _ptr(ws["fm_out"]), _ptr(ws["fm_sum"]), _ptr(eng.err_flag),
                                       eng._age(0) if age else None, st))


def bwd(b, age):
    eng._advance_step()  # (rows updated by the call are dated one step on: every call sees aged rows again)
    opt = eng._opt_args()
    _lib.check(_lib.lib.xdfm_embed_bwd(0, _ptr(eng.table), _ptr(eng.table_m), _ptr(eng.table_v), _ptr(eng.lin_w),
                                       _ptr(eng.lin_m), _ptr(eng.lin_v), eng.table_stride, eng.lin_stride, None, None,
                                       eng.N, D, F, B, _ptr(b.sorted_ids), None, _ptr(b.sorted_row), _ptr(b.sorted_w),
                                       b.nnz, _ptr(ws["d_emb"]), _ptr(ws["g_logit"]), _ptr(ws["fm_sum"]), 1,
                                       ctypes.byref(opt), _ptr(ws["pieces"]), None, 0, eng._age(-1) if age else None, st))


def timed(fn, label, alg_bytes):
    for b in batches:
        fn(b)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 3 * NB
    e0.record()
    for k in range(reps):
        fn(batches[k % NB])
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    print("%-46s %7.2f us per call, %5.0f GB/s algorithmic = %.3f of 8 TB/s" % (label, us, alg_bytes / us / 1e3,
                                                                               alg_bytes / us / 1e3 / 8000))


fwd_bytes = nnz * (8 * D + 12)
bwd_bytes = nnz * (4 * D + 12) + uniq * 6 * 4 * D
timed(lambda b: fwd(b, False), "embed_fwd, rows as stored", fwd_bytes)
timed(lambda b: fwd(b, True), "embed_fwd, rows aged in registers", fwd_bytes)
timed(lambda b: eng._advance_step(), "opt_advance alone (part of the two below)", 0)
timed(lambda b: bwd(b, True), "embed_bwd + long, rows aged in the update", bwd_bytes)
timed(lambda b: bwd(b, False), "embed_bwd + long, no ageing", bwd_bytes)
eng.check_errors()
