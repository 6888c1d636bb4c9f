#!/usr/bin/env python
"""Where the HOST's time goes in one row-sharded step on one GPU (bench workload): every step starts with the
device idle (a synchronize in front), so what is measured per section is the host's enqueue work, not waiting.
    python tools/ub_sharded_host.py [--rccl]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from recommenders_amd.models.deeprec.models._sharded import ShardedXDeepFMEngine  # noqa: E402


class A(object):
    features, dim, batch, workload, dtype = 1 << 25, 16, 8192, "criteo", "fp32"


rccl = "--rccl" in sys.argv
dev = torch.device("cuda:0")
if rccl:
    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
hp = bench.workload_hparams(A)
eng = ShardedXDeepFMEngine(hp, device=dev, seed=7, force_collectives=rccl)
host = bench.synth_batches(hp, 16, seed=1234)
batches = [eng.upload(hb["labels"], hb["dnn_feat_indices"], hb["dnn_feat_values"], hb["dnn_feat_weights"]) for hb in host]
acc = {}


def timed(name, fn):
    def wrap(*a, **k):
        t0 = time.perf_counter()
        r = fn(*a, **k)
        acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
        return r
    return wrap


for name in ("resolve_plan", "fetch_rows", "_owner_update", "_update_dense", "prefetch_route", "_advance_step", "join_late"):
    setattr(eng, name, timed(name, getattr(eng, name)))
eng.exchange.all_reduce = timed("all_reduce", eng.exchange.all_reduce)
eng.exchange.fits = timed("  fits (inside resolve_plan)", eng.exchange.fits)
eng.exchange.to_owner = timed("  to_owner (exchange)", eng.exchange.to_owner)
eng.exchange.to_source = timed("  to_source (exchange)", eng.exchange.to_source)
n = len(batches)
for i in range(6):
    eng.train_step(batches[i % n], next_batch=batches[(i + 1) % n])
torch.cuda.synchronize()
acc.clear()
steps, total = 40, 0.0
for i in range(6, 6 + steps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.train_step(batches[i % n], next_batch=batches[(i + 1) % n])
    total += time.perf_counter() - t0
torch.cuda.synchronize()
print("host time per step with an idle device in front: %.1f us (%s exchanges); fixed / sized steps %d / %d" % (
    total / steps * 1e6, "RCCL" if rccl else "no", eng.fixed_steps, eng.sized_steps))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print("  %-34s %7.1f us" % (k, v / steps * 1e6))
