#!/usr/bin/env python
"""Benchmark of the xDeepFM training hot path on MI355X (BASELINE.json metric:
"train samples/sec, xDeepFM Criteo-shape").

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[2], SURVEY.md 8(d) input S3): Criteo-shape synthetic, 39 fields
(13 numeric with one hot feature id each + 26 categorical, Zipf-like ids in per-field ranges),
FEATURE_COUNT = 2^25 hashed rows, dim 16, batch 8192 per GPU, linear + FM + CIN[20,10] + DNN[20,20],
log_loss, Adam with sparse row updates of the embedding table ("sparse_rows": rows are aged to the
reference's dense-exact values where they are read).  A step is one full training pass over one batch
whose arrays (with the lookup's CSR offsets, built once when the batch was packed) are already resident
in HBM: one copy of the packed batch into the captured step's input, forward, backward, id sort +
segment-sum + fused row Adam, dense Adam.  fp32 throughout.

N > 1 (launched by torch.distributed.run, one rank per GPU): the table is row-sharded over the
ranks (owner = id mod N) with RCCL all-to-all exchanges, dense gradients are all-reduced; each
rank keeps its own 8192-sample batch (weak scaling).

Prints ONE JSON line on rank 0.  `roofline` is for the kernel with the largest total time per step
(`roofline.rank` = 1; should the largest have no model, the line says which one that is in
`roofline.unmodelled_above` instead of silently moving on), its duration measured live with HIP events
on the launch stream (xdfm_prof_enable/collect); `cpu_baseline` is the CPU oracle (oracle/, torch CPU
fp32) timed on a bounded sample of the same workload on the host cores.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from recommenders_amd import _lib  # noqa: E402
from recommenders_amd.models.deeprec.deeprec_utils import prepare_hparams  # noqa: E402
from recommenders_amd.synth import criteo_shape_ids  # noqa: E402

# HBM bytes per launch of each kernel from rocprofv3 PMC passes over this workload (FETCH_SIZE doubled
# as MI355X_MICROARCH.md prescribes for gfx950, + WRITE_SIZE): written by
# `tools/gpu_pmc.sh <tag>` + `tools/pmc_summary.py ... --json profiles/traffic.json`
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "traffic.json")


def load_traffic():
    try:
        with open(TRAFFIC_FILE) as f:
            return json.load(f).get("kernels", {})
    except (OSError, ValueError):
        return {}


def traffic_of(name):
    """HBM bytes per launch of a kernel, by the name the library gives it (rocprofv3 prints every
    template argument, the library's label may leave trailing ones out)."""
    table = load_traffic()
    if name in table:
        return table[name]
    stem = name[:-1] if name.endswith(">") else name
    hits = [v for k, v in table.items() if k.startswith(stem + ",") or k.startswith(stem + "<")]
    return hits[0] if len(hits) == 1 else None


HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
F32_MATRIX_PEAK_TFLOPS = 157.3  # same guide: fp32 MFMA / vector peak
BF16_MATRIX_PEAK_TFLOPS = 2500.0  # same guide: bf16 MFMA, dense


def workload_hparams(args):
    wide = getattr(args, "workload", "criteo") == "wide"
    return prepare_hparams(
        None, model_type="xDeepFM", data_format="ffm", FEATURE_COUNT=args.features, FIELD_COUNT=39,
        dim=64 if wide else args.dim, layer_sizes=[1024, 512] if wide else [20, 20],
        cross_layer_sizes=[256, 256, 256] if wide else [20, 10], activation=["relu", "relu"],
        dropout=[0.0, 0.0], loss="log_loss", method="classification", metrics=["auc", "logloss"],
        use_Linear_part=True, use_FM_part=True, use_CIN_part=True, use_DNN_part=True,
        init_method="tnormal", init_value=0.1, learning_rate=0.002, optimizer="adam",
        embed_l2=0.0, layer_l2=0.0, cross_l2=0.01, batch_size=args.batch, save_model=False,
        embedding_update="sparse_rows", compute_dtype=getattr(args, "dtype", None) or "fp32")


def synth_batches(hp, n_batches, seed):
    """S3 batches as host arrays in the iterator's layout (single-hot: one token per field)."""
    rng = np.random.default_rng(seed)
    B, F, N = hp.batch_size, hp.FIELD_COUNT, hp.FEATURE_COUNT
    out = []
    rows = np.arange(B * F, dtype=np.int64)
    dnn_idx = np.stack([rows, np.zeros_like(rows)], axis=1)
    for _ in range(n_batches):
        ids = criteo_shape_ids(rng, B, N).astype(np.int64).reshape(-1)
        w = np.ones((B, F), dtype=np.float32)
        w[:, :13] = rng.random((B, 13), dtype=np.float32)
        labels = (rng.random(B) < 0.25).astype(np.float32)
        out.append({"labels": labels.reshape(B, 1), "dnn_feat_indices": dnn_idx, "dnn_feat_values": ids,
                    "dnn_feat_weights": w.reshape(-1),
                    "fm_feat_indices": np.stack([rows // F, ids], axis=1), "fm_feat_values": w.reshape(-1)})
    return out


def kernel_models(hp, nnz, uniq):
    """name -> (bound, [algorithmic bytes or flops of each launch of that kernel in one step])
    (DESIGN.md section 4).  Kernel names are the ones libxdfm.so launches for this
    workload; the template arguments follow the dispatch rules of csrc/cin_kernels.hip."""
    B, F, D = hp.batch_size, hp.FIELD_COUNT, hp.dim
    vec = 4 if D % 4 == 0 else (2 if D % 2 == 0 else 1)
    m = {
        "embed_fwd_kernel<%d, true>" % vec: ("hbm", [nnz * (8 * D + 12)]),
        "embed_fwd_kernel<%d, false>" % vec: ("hbm", [nnz * (8 * D + 12)]),
        # per token: grad row + sorted key/token + weight; per unique row: read+write of p, m, v
        "embed_bwd_kernel<%d, 0, true>" % vec: ("hbm", [nnz * (4 * D + 12) + uniq * 6 * 4 * D]),
        "embed_bwd_kernel<%d, 0, false>" % vec: ("hbm", [nnz * (4 * D + 12) + uniq * 6 * 4 * D]),
        # sorted order of (row, weight) per token: 4 B index in, 8 B gathered, 8 B out
        "plan_gather_kernel": ("hbm", [nnz * 20]),
    }
    # LSD radix sort of (id, token) pairs (csrc/radix_sort.hip): one pass per digit reads a pair and writes a
    # pair (16 B per token, SCRATCH traffic: the reference scatters unordered and has no such bytes), its
    # histogram pass reads the keys (4 B per token)
    bits = max(1, int(np.ceil(np.log2(max(hp.FEATURE_COUNT, 2)))))
    n_pass = (bits + 8) // 9  # csrc/radix_sort.hip: digits of at most 9 bits
    m["sort_scatter_kernel"] = ("hbm", [nnz * 16] * n_pass)
    m["sort_hist_kernel"] = ("hbm", [nnz * 4] * n_pass)

    def add(name, bound, work):
        m.setdefault(name, (bound, []))[1].append(work)

    H = F
    n_layers = len(hp.cross_layer_sizes)
    if hp.values().get("compute_dtype", "fp32") == "bf16":
        # bf16 layer kernels carry the hidden channels; algorithmic flops = the reference's pairs
        # (the masked layer's f < h only) x channels, per (sample, d) row
        rows = B * D

        def nks(K):
            n = (K + 15) // 16
            return 2 if n <= 2 else 3 if n <= 3 else 4 if n <= 4 else 6 if n <= 6 else 8

        def tiles(K):
            return 1 if K <= 32 else 2 if K <= 64 else 4

        for k, L in enumerate(hp.cross_layer_sizes):
            nh = L // 2 if k != n_layers - 1 else 0
            pairs = F * (F - 1) // 2 if k == 0 else F * H
            if nh:
                add("cin_bf16_fwd_kernel<%d>" % nks(H), "mfma16", 2.0 * rows * pairs * nh)
                add("cin_bf16_dx_kernel<%d>" % nks(nh), "mfma16", 2.0 * rows * pairs * nh)
                # filter gradient, as csrc/cin_bf16_dw_kernels.hip dispatches it: the masked layer's pair form,
                # the full 128 x 128 tile at D = 64 with loads that write LDS, else the staged form
                cols_ok = D % 2 == 0 and (B * D) % 64 == 0
                if k == 0 and F <= 64 and cols_ok and os.environ.get("XDFM_DW_TRI", "1") != "0":
                    dw_name = "cin_bf16_dw_tri_kernel<%d>" % tiles(nh)
                elif k != 0 and D == 64 and H == 128 and nh == 128 and os.environ.get("XDFM_DW_GLDS", "1") != "0":
                    dw_name = "cin_bf16_dw_glds_kernel"
                else:
                    dw_name = "cin_bf16_dw_kernel<%d, %d>" % (tiles(H), tiles(nh))
                add(dw_name, "mfma16", 2.0 * rows * pairs * nh)
            H = nh
        return m
    for k, L in enumerate(hp.cross_layer_sizes):
        masked = k == 0
        pairs = F * (F - 1) // 2 if masked else F * H  # masked pairs (f >= h) are never computed
        rows = B * D
        if k == n_layers - 1 and k > 0 and F <= 64 and H <= 16 and D <= 16 and D % 4 == 0 and L <= 16:
            # every channel pooled: one kernel each way, sum over d first, bounded by reading x0 / xk once
            # (DESIGN.md section 4): forward 4 B (F + H) D bytes; backward that + d_x0, d_xk written + d_x0 read
            fwd_bytes = 4.0 * B * (F + H) * D
            add("cin_pooled_fwd_head_kernel", "hbm", fwd_bytes)
            add("cin_pooled_fwd_kernel", "hbm", fwd_bytes)
            add("cin_pooled_bwd_kernel", "hbm", 2.0 * fwd_bytes + 4.0 * B * F * D)
            continue
        nt_total = (L + 15) // 16
        fwd_flops = 2.0 * rows * pairs * L + 2.0 * rows * min(F, H) * L
        # the masked layer of up to 40 fields runs the unrolled form (csrc/cin_mfma_kernels.hip, fwd_mfma)
        if masked and D % 4 == 0 and (F + 3) // 4 <= 10 and os.environ.get("XDFM_CIN_FWD_STATIC", "1") != "0":
            add("cin_fwd_static_kernel<%d>" % ((F + 3) // 4), "mfma", fwd_flops)
        else:
            add("cin_fwd2_mfma_kernel", "mfma", fwd_flops)
        nft, nht = (F + 3) // 4, (H + 3) // 4
        S = (L + 3) // 4
        nf = 3 if nft <= 3 else 10
        ss = 3 if S <= 3 else 5
        ntile = nft * (nft + 1) // 2 if masked else nft * nht
        pt = 7 if (ntile + 27) // 28 * 28 < (ntile + 31) // 32 * 32 else 8
        nt_w = 2 if nt_total >= 2 else 1
        dx_flops = 2.0 * rows * pairs * L + 4.0 * rows * pairs
        dw_flops = 2.0 * rows * pairs * L
        nh_k = L // 2 if k != n_layers - 1 else 0
        shh, p0 = (nh_k + 3) // 4, nh_k // 4
        cut = D % 4 == 0 and 0 < nh_k < L and ((ss == 5 and shh == 3 and p0 == 2) or (ss == 3 and shh == 2 and p0 == 1))
        add("cin_bwd_dx2_mfma_kernel<%d, %d, %d>" % (nf, ss, shh if cut else ss), "mfma", dx_flops)
        nh = L // 2 if k != n_layers - 1 else 0
        # hidden | pooled cut of the channel tiles (csrc/cin_mfma_kernels.hip, dw_body): a pooled channel's
        # products are summed over the sample's D rows first.  The ALGORITHMIC flops stay 2 rows pairs L.
        sd = D // 8 if (0 < nh <= 16 and 0 < L - nh <= 16 and D in (8, 16, 32)) else 0
        add("cin_bwd_dw_mfma_kernel<%d, %d, %d>" % (pt, 2 if sd else nt_w, sd), "mfma", dw_flops)
        H = L // 2 if k != n_layers - 1 else L
    return m


def run_cpu_baseline(hp, batches, budget_s=20.0):
    """The oracle's train_step (torch CPU fp32) on the first batches, in the reference's semantics
    (dense-exact Adam: every one of the 2^25 rows moves every step) and, for scale,
    with lazy row updates (the cheapest thing a CPU could do)."""
    from oracle import xdeepfm_oracle as X

    cores = min(16, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    shapes = X.param_shapes(hp)
    gen = torch.Generator().manual_seed(0)
    params = {}
    for name, shape in shapes.items():
        t = torch.empty(shape, dtype=torch.float32)
        if name.endswith("_b") or "_b_" in name:
            t.zero_()
        else:
            t.uniform_(-0.1, 0.1, generator=gen)
        params[name] = t
    st = X.new_opt_state(params, hp)

    def timed(mode, budget, max_steps):
        X.train_step(params, st, batches[0], hp, mode=mode)  # warm-up (allocator, threads)
        done, t0 = 0, time.perf_counter()
        while True:
            X.train_step(params, st, batches[(done + 1) % len(batches)], hp, mode=mode)
            done += 1
            el = time.perf_counter() - t0
            if el > budget or done >= max_steps:
                return done, el

    lazy_n, lazy_t = timed("sparse_lazy", 0.3 * budget_s, 16)
    dense_n, dense_t = timed("dense_exact", 0.7 * budget_s, 8)
    return {"value": dense_n * hp.batch_size / dense_t, "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": "%d training steps of the same batch-%d workload (oracle/xdeepfm_oracle.py, torch CPU fp32, "
                      "dense-exact Adam = the reference's semantics) after 1 warm-up step" % (dense_n, hp.batch_size),
            "lazy_rows_value": lazy_n * hp.batch_size / lazy_t,
            "lazy_rows_sample": "%d steps with lazy row updates (a different algorithm for absent rows; the "
                                "cheapest CPU form)" % lazy_n}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # (400 timed steps = 0.15 s on one GPU: the launch queue is in its steady state for most of them; at 100 the
    # line read 1 % lower)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=8192)
    ap.add_argument("--dim", type=int, default=16)
    ap.add_argument("--features", type=int, default=1 << 25)
    ap.add_argument("--workload", choices=["criteo", "wide"], default="criteo",
                    help="criteo = BASELINE configs[2] (the bench line); wide = configs[4]'s layer sizes "
                         "(dim 64, CIN [256,256,256], DNN [1024,512]) at --batch per GPU, a side measurement")
    ap.add_argument("--dtype", choices=["fp32", "bf16"], default=None,
                    help="arithmetic of the CIN / DNN products (default: fp32 for criteo, bf16 for wide)")
    ap.add_argument("--batches", type=int, default=64, help="distinct synthetic batches cycled through")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel eagerly in the timed region")
    ap.add_argument("--sharded", action="store_true",
                    help="one GPU, but through the row-sharded engine (the N > 1 code path without peers)")
    ap.add_argument("--force-collectives", action="store_true",
                    help="with --sharded on one GPU: run every exchange through RCCL anyway (a process group of "
                         "one rank): the launch cost of the collectives without peers, a side measurement")
    ap.add_argument("--profile-steps", type=int, default=20,
                    help="eager steps after the timed region with per-kernel HIP events")
    args = ap.parse_args()
    if args.dtype is None:
        args.dtype = "bf16" if args.workload == "wide" else "fp32"

    # stdout carries the one JSON line and nothing else: libraries write to file descriptor 1 too (RCCL
    # prints a version banner there when its first communicator comes up), so descriptor 1 is pointed
    # at stderr for the run and the line goes to the kept copy
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run with --nproc-per-node %d" % (args.gpus, args.gpus))
    # XDFM_DIST_BACKEND=gloo: rehearsal of the N > 1 code on a box with fewer GPUs than ranks
    # (ranks share devices, exchanges staged through the host) -- never a measurement
    backend = os.environ.get("XDFM_DIST_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    hp = workload_hparams(args)
    # distinct batches cycled: their touched rows (~98 k x 256 B each) must exceed the 256 MB Infinity
    # Cache, or the "HBM" rates are last-level-cache rates
    n_batches = args.batches
    host_batches = synth_batches(hp, n_batches, seed=1234 + rank)

    if world == 1 and not args.sharded:
        from recommenders_amd.models.deeprec.models._engine import XDeepFMEngine

        eng = XDeepFMEngine(hp, device=dev, seed=7, embedding_update="sparse_rows")
    else:
        from recommenders_amd.models.deeprec.models._sharded import ShardedXDeepFMEngine

        if world == 1 and args.force_collectives:
            import torch.distributed as dist1

            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            dist1.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        # (every rank's synthetic batches have one shape: the whole step may replay as one graph on all ranks)
        eng = ShardedXDeepFMEngine(hp, device=dev, seed=7, force_collectives=bool(world == 1 and args.force_collectives),
                                   uniform_batches=True)

    resident = []
    for hb in host_batches:
        resident.append(tuple(torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (
            hb["labels"].reshape(-1), hb["dnn_feat_indices"], hb["dnn_feat_values"], hb["dnn_feat_weights"])))
    nnz = int(resident[0][2].numel())
    uniq = float(np.mean([len(np.unique(hb["dnn_feat_values"])) for hb in host_batches]))

    made = {}

    def device_batch(k):
        # the resident batch as the engine's DeviceBatch (CSR offsets built once, like the packed batches of
        # the captured step)
        k %= n_batches
        if k not in made:
            made[k] = eng.make_batch(*resident[k])
        return made[k]

    def eager_step(i):
        if world > 1 or args.sharded:
            # the following batch is announced: its routing to the owners (ids out, owner-side sort) runs
            # beside this step
            return eng.train_step(device_batch(i), next_batch=device_batch(i + 1))
        b = device_batch(i)
        b.planned = b.planned_gather = False  # every step sorts its ids, as the captured step does
        return eng.train_step(b)

    graphed = None
    if getattr(eng, "folded", False):
        print("bench: GPU_MAX_HW_QUEUES=%s -- the step is folded onto one stream (its capture is a chain)"
              % os.environ.get("GPU_MAX_HW_QUEUES"), file=sys.stderr)
    if world == 1 and not args.no_graph and not args.sharded:
        # one GPU: the whole step (copy of the batch arrays into the graph's inputs aside) is one
        # HIP graph launch; N > 1 stays eager (split sizes of the all-to-all are host values)
        graphed = eng.capture_step(*resident[0])
        # the resident batches in the graph's input layout: handing one over is a single copy
        packed = [graphed.pack(*r) for r in resident]

    def step(i):
        if graphed is not None:
            return graphed.run_packed(packed[i % n_batches])
        return eager_step(i)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    fence()
    eng.check_errors()
    # HIP events cannot be recorded inside a graph replay (the sharded engine replays its local
    # part as a graph too)
    in_situ = graphed is None and not getattr(eng, "graph_local", False)
    if in_situ:
        _lib.check(_lib.lib.xdfm_prof_enable(1))
    fence()
    graph_steps_before = getattr(eng, "graph_steps", 0)
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss, _ = step(args.warmup + i)
    host_enqueue = time.perf_counter() - t0  # the host's share: below the step time, the device is the limit
    fence()
    elapsed = time.perf_counter() - t0
    graph_steps_timed = getattr(eng, "graph_steps", 0) - graph_steps_before
    prof_steps = args.steps
    if not in_situ:
        # same kernels, same batches, launched eagerly with the library's HIP events around
        # every launch (on the launch stream), right after the timed region
        final = float(loss)
        prof_steps = args.profile_steps
        _lib.check(_lib.lib.xdfm_prof_enable(1))
        if hasattr(eng, "graph_local"):
            eng.graph_local = False
        # per-kernel durations are taken with every kernel alone on the device: the side-stream
        # placements of the last filter gradient (beside the table update) and of the id sort (beside
        # the forward pass) are switched off here
        eng.split_dw = False
        eng.serial_plan = True
        if hasattr(eng, "overlap_dnn") and not args.sharded and world == 1:
            # ... and so is the DNN tower's stream (its kernels and the CIN's stretch each other to twice
            # their time when they share the chip: tools/stamp_timeline.py)
            eng.overlap_dnn = eng.early_reg = False
        for i in range(prof_steps):
            eager_step(args.warmup + args.steps + i)
        fence()
        loss = torch.tensor(final)
    _lib.check(_lib.lib.xdfm_prof_enable(0))
    buf = ctypes.create_string_buffer(1 << 16)
    _lib.check(_lib.lib.xdfm_prof_collect(buf, len(buf)))
    prof = json.loads(buf.value.decode())
    eng.check_errors()
    final_loss = float(loss)
    if not np.isfinite(final_loss) and not os.environ.get("XDFM_BENCH_ALLOW_NAN"):
        raise SystemExit("training diverged: loss %r" % final_loss)

    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        total_ms = sum(v["ms"] for v in prof.values())
        ranked = sorted(prof.items(), key=lambda kv: -kv[1]["ms"])
        models = kernel_models(hp, nnz, uniq)

        def rate(name, v):
            if name not in models:
                return None
            bound, works = models[name]
            per_step = v["n"] / prof_steps
            if abs(per_step - len(works)) > 0.26:  # (the first profiled step may sort one batch more)
                return None  # launched a different number of times than modelled: do not guess
            avg_s = v["ms"] / v["n"] * 1e-3
            work = sum(works) / len(works)
            if bound == "hbm":
                return bound, work / avg_s / 1e9, HBM_PEAK_GBS, "GB/s", avg_s
            if bound == "mfma16":
                return "mfma", work / avg_s / 1e12, BF16_MATRIX_PEAK_TFLOPS, "TFLOP/s", avg_s
            return bound, work / avg_s / 1e12, F32_MATRIX_PEAK_TFLOPS, "TFLOP/s", avg_s

        # the dominant kernel = the largest total time per step; kernels above it that have no model are
        # named, never skipped in silence
        roofline = None
        skipped = []
        for pos, (name, v) in enumerate(ranked, 1):
            r = rate(name, v)
            if r is None:
                skipped.append(name)
                continue
            bound, ach, peak, unit, avg_s = r
            roofline = {"kernel": name, "bound": bound, "achieved": round(ach, 3), "peak": peak, "unit": unit,
                        "frac": round(ach / peak, 4), "traffic": traffic_of(name), "avg_us": round(avg_s * 1e6, 2),
                        "launches_per_step": round(v["n"] / prof_steps, 2),
                        "share_of_kernel_time": round(v["ms"] / total_ms, 3), "rank": pos,
                        "unmodelled_above": skipped}
            break
        kernels = {}
        for name, v in ranked[:40]:
            entry = {"avg_us": round(v["ms"] / v["n"] * 1e3, 2), "per_step": round(v["n"] / prof_steps, 2)}
            r = rate(name, v)
            if r is not None:
                entry[r[3]] = round(r[1], 2)
                entry["frac"] = round(r[1] / r[2], 4)
            kernels[name] = entry
        line = {
            "metric": "train samples/sec, xDeepFM Criteo-shape", "value": round(world * hp.batch_size * args.steps / elapsed, 1),
            "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "host_enqueue_ms_per_step": round(host_enqueue / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16" if args.dtype == "bf16" else "f32", "data": "synthetic",
            "config": {"workload": ("BASELINE configs[2]: Criteo-shape synthetic, 39 fields, 2^25 rows, dim %d, "
                                    "batch %d per GPU, linear+FM+CIN[20,10]+DNN[20,20], sparse-row Adam"
                                    if args.workload == "criteo" else
                                    "SIDE MEASUREMENT, one GPU's share of BASELINE configs[4] (" + args.dtype +
                                    " products, fp32 accumulation): 39 fields, dim %d, batch %d per GPU, "
                                    "CIN[256,256,256]+DNN[1024,512]") % (hp.dim, hp.batch_size),
                       "FEATURE_COUNT": hp.FEATURE_COUNT, "FIELD_COUNT": 39, "dim": hp.dim,
                       "global_batch": world * hp.batch_size,
                       "parallelism": "single GPU" if world == 1 else "table row-sharded x%d (all-to-all) + dense all-reduce%s" % (
                           world, "" if backend == "nccl" else " [REHEARSAL over %s, ranks share GPUs: not a measurement]" % backend)},
            "roofline": roofline,
            "kernel_time_ms_per_step": round(total_ms / prof_steps, 4),
            "launch": (("hip graph replay" + (" (folded onto one stream: GPU_MAX_HW_QUEUES < 4)"
                                              if getattr(eng, "folded", False) else "")) if graphed is not None else
                       ("whole step as one hip graph per routing generation (%d of %d timed steps)"
                        % (graph_steps_timed, args.steps) if graph_steps_timed else
                        "exchanges eager, local forward/backward as a hip graph") if not in_situ else "eager"),
            "kernel_timing": "HIP events around every launch, %s" % (
                "in the timed region" if in_situ else "%d eager steps right after the timed graph replays" % prof_steps),
            "kernels": kernels,
            "final_loss": round(final_loss, 5),
        }
        if world == 1 and not args.no_cpu_baseline and args.workload == "criteo":
            line["cpu_baseline"] = run_cpu_baseline(hp, host_batches)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    if dist is not None:
        if hasattr(eng, "close"):
            eng.close()  # graphs that hold captured collectives go before the group does
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
