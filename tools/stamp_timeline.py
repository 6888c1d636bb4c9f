#!/usr/bin/env python
"""Timeline of ONE replay of the captured training step on the bench workload, from device-side time stamps
(xdfm_debug_stamps: every kernel of the step writes the 100 MHz clock at its first block's start and its
last block's end).  Unlike a tracer this leaves the replay as it is: same graph, same launch, no signals.

The replay that is measured is the last of a run of 31 enqueued without a host synchronisation (the stamps
are cleared by a small copy in stream order right before it), so clocks and queues are in the steady state
of the bench loop, where the host is a step ahead.  Printed: start / end / duration in us relative to the earliest start, per kernel family
(a family launched several times per step shows its first start and last end, and the launch count is in
the name); then the step's span and the idle gaps on the chain of large kernels.

    python tools/stamp_timeline.py [--repeat R] [bench.py's workload flags]"""
import argparse
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from recommenders_amd import _lib  # noqa: E402
from recommenders_amd.models.deeprec.models._engine import XDeepFMEngine  # noqa: E402

NAMES = ["embed_fwd", "cin_fwd (layer 0)", "cin_pooled_fwd + head", "cin_pooled_bwd", "cin_bwd_dx2", "cin_bwd_dw",
         "embed_bwd", "embed_bwd_long", "dense_update", "dense_fwd x3", "dense_bwd_dx x3", "dense_bwd_dw + reduce",
         "sort hist/scatter x6", "plan_gather", "cin_pack_all", "loss_finish", "opt_advance", "dense_stats x2",
         "cin_dw_unpack", "loss_pair", "cin_pooled_dw_reduce", "route_* (next batch)", "gather_rows (owner)",
         "embed_bwd scatter (records)"]
SLOTS = 32

ap = argparse.ArgumentParser()
ap.add_argument("--repeat", type=int, default=3)
ap.add_argument("--batch", type=int, default=8192)
ap.add_argument("--dim", type=int, default=16)
ap.add_argument("--features", type=int, default=1 << 25)
ap.add_argument("--batches", type=int, default=16)
ap.add_argument("--sharded", action="store_true", help="the row-sharded engine's step (eager exchanges + local graph)")
ap.add_argument("--force-collectives", action="store_true")
args = ap.parse_args()
args.workload, args.dtype = "criteo", "fp32"

hp = bench.workload_hparams(args)
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
host = bench.synth_batches(hp, args.batches, seed=1234)
resident = [tuple(torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (
    hb["labels"].reshape(-1), hb["dnn_feat_indices"], hb["dnn_feat_values"], hb["dnn_feat_weights"])) for hb in host]
if args.sharded:
    from recommenders_amd.models.deeprec.models._sharded import ShardedXDeepFMEngine

    if args.force_collectives:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29534")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    eng = ShardedXDeepFMEngine(hp, device=dev, seed=7, force_collectives=args.force_collectives)
    made = [eng.make_batch(*r) for r in resident]

    def run_step(i):
        eng.train_step(made[i % len(made)], next_batch=made[(i + 1) % len(made)])
else:
    eng = XDeepFMEngine(hp, device=dev, seed=7, embedding_update="sparse_rows")
    graphed = eng.capture_step(*resident[0])
    packed = [graphed.pack(*r) for r in resident]

    def run_step(i):
        graphed.run_packed(packed[i % len(packed)])
for i in range(40):
    run_step(i)
torch.cuda.synchronize()
eng.check_errors()

init = torch.empty(2 * SLOTS, dtype=torch.int64)
init[0::2] = np.iinfo(np.int64).max  # (the stamps are far below 2^63: signed and unsigned order agree)
init[1::2] = 0
buf = init.to(dev)
init_dev = init.to(dev)
_lib.check(_lib.lib.xdfm_debug_stamps(ctypes.c_void_p(buf.data_ptr())))
spans = []
for r in range(args.repeat):
    # 30 replays ahead of the measured one, no host synchronisation in between: clocks are up and the host is
    # a step ahead when the stamps are cleared (a 512-byte copy in stream order) and the measured replay starts
    for i in range(30):
        run_step(40 + 31 * r + i)
    buf.copy_(init_dev, non_blocking=True)
    run_step(40 + 31 * r + 30)
    torch.cuda.synchronize()
    t = buf.cpu().numpy().astype(np.float64).reshape(SLOTS, 2) / 100.0  # us
    rows = [(t[i, 0], t[i, 1], NAMES[i]) for i in range(len(NAMES)) if t[i, 1] > 0]
    t0 = min(a for a, _, _ in rows)
    rows.sort()
    print("replay %d" % r)
    for a, b, n in rows:
        print("  %8.1f %8.1f %7.1f  %s" % (a - t0, b - t0, b - a, n))
    end = max(b for _, b, _ in rows)
    spans.append(end - t0)
    chain = [n for n in ("embed_fwd", "cin_fwd (layer 0)", "cin_pooled_fwd + head", "cin_pooled_bwd", "cin_bwd_dx2",
                         "embed_bwd", "embed_bwd_long", "dense_update", "loss_pair")]
    byname = {n: (a - t0, b - t0) for a, b, n in rows}
    prev, gaps = None, []
    for n in chain:
        if n not in byname:
            continue
        if prev is not None:
            gaps.append("%s -> %s %.1f" % (prev, n, byname[n][0] - byname[prev][1]))
        prev = n
    print("  span %.1f us; gaps on the chain: %s" % (end - t0, "; ".join(gaps)))
_lib.check(_lib.lib.xdfm_debug_stamps(None))
print("spans: " + " ".join("%.1f" % s for s in spans))
if args.sharded:
    eng.close()  # (captured collectives go before the process group does)
