"""Multi-GPU xDeepFM step: embedding table row-sharded over the ranks, dense tower data-parallel.

The reference trains on one device in one process (no collective call
site exists in it), so this layout is new design, not a translation:

* one process per GPU, ``torch.distributed`` with the ``nccl`` backend (RCCL over xGMI);
* table row ``id`` (and its optimizer slots, and ``lin_w[id]``) lives on rank ``id % G`` at
  local row ``id // G`` -- the 13 hot "numeric" ids of Criteo-shaped data spread over ranks;
* forward: all-to-all of the batch's ids to their owners, owners gather rows with the same
  HIP gather kernel, all-to-all of the rows back (every GPU link carries one message);
* backward: per-token gradient rows go back the same way; the owner runs the deterministic
  sorted-key segment sum + fused sparse row optimizer on what it received (source-rank order
  is fixed, so the sum is reproducible);
* dense parameters are replicated, their gradients all-reduced (one flat buffer).

``RowShardExchange`` is the routing layer alone (pure index plumbing + collectives); it also
runs on CPU tensors with the ``gloo`` backend, which is how the N > 1 path is tested without
GPUs.  ``ShardedXDeepFMEngine`` composes it with the kernels of libxdfm.so.
"""

import ctypes

import numpy as np
import torch
import torch.distributed as dist

from recommenders_amd import _lib
from recommenders_amd.models.deeprec.models._engine import XDeepFMEngine, _ptr

__all__ = ["RowShardExchange", "RoutePlan", "ShardedXDeepFMEngine", "owner_of", "local_rows"]


def owner_of(ids, world):
    return ids % world


def local_rows(n_rows, rank, world):
    """Rows of a FEATURE_COUNT = n_rows table that rank holds (ids rank, rank+G, ...)."""
    return (n_rows - rank + world - 1) // world


class RoutePlan(object):
    """Routing state of one batch: partition order, split sizes, ids received as owner."""

    __slots__ = ("order", "inverse", "send_counts", "recv_counts", "recv_rows", "n_send", "n_recv")


class RowShardExchange(object):
    def __init__(self, group=None, virtual_world=1):
        """virtual_world > 1 (single process only, for tests): partition the tokens as if there
        were that many owners, with every "owner" being this process.  The exchanges are then
        identities, but the engine sees the permuted token order of a real multi-rank step."""
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.virtual_world = int(virtual_world) if self.world == 1 else 1
        # rehearsal mode: the gloo backend has no all-to-all for device tensors, so several ranks
        # sharing one GPU (tests on a one-GPU box) stage their exchanges through the host
        self.host_staged = self.world > 1 and dist.get_backend(group) == "gloo"

    def _a2a(self, out, x, out_splits=None, in_splits=None):
        if self.host_staged and x.is_cuda:
            h = torch.empty(out.shape, dtype=out.dtype)
            dist.all_to_all_single(h, x.cpu(), out_splits, in_splits, group=self.group)
            out.copy_(h)
        else:
            dist.all_to_all_single(out, x, out_splits, in_splits, group=self.group)

    def all_reduce(self, x):
        if self.host_staged and x.is_cuda:
            h = x.cpu()
            dist.all_reduce(h, group=self.group)
            x.copy_(h)
        else:
            dist.all_reduce(x, group=self.group)

    def broadcast(self, x, src):
        if self.host_staged and x.is_cuda:
            h = x.cpu()
            dist.broadcast(h, src=src, group=self.group)
            x.copy_(h)
        else:
            dist.broadcast(x, src=src, group=self.group)

    def route(self, ids):
        """ids: int64 [n] global feature ids of this rank's tokens.  Sends every id to its
        owner (as a local row index) and returns the RoutePlan."""
        G = self.world
        plan = RoutePlan()
        n = int(ids.numel())
        plan.n_send = n
        if G == 1 and self.virtual_world > 1:
            order = torch.argsort(ids % self.virtual_world, stable=True)
            inverse = torch.empty_like(order)
            inverse[order] = torch.arange(n, dtype=order.dtype, device=order.device)
            plan.order, plan.inverse = order, inverse
            plan.send_counts = plan.recv_counts = [n]
            plan.recv_rows = ids[order].contiguous()
            plan.n_recv = n
            return plan
        if G == 1:
            plan.order = plan.inverse = None
            plan.send_counts = plan.recv_counts = [n]
            plan.recv_rows = ids
            plan.n_recv = n
            return plan
        dest = ids % G
        # stable partition by destination: tokens of one destination keep their token order
        order = torch.argsort(dest, stable=True)
        send_counts = torch.bincount(dest, minlength=G)
        recv_counts = torch.empty_like(send_counts)
        self._a2a(recv_counts, send_counts)
        both = torch.stack([send_counts, recv_counts]).tolist()  # the step's one host sync: exchange sizes
        plan.send_counts = [int(v) for v in both[0]]
        plan.recv_counts = [int(v) for v in both[1]]
        plan.order = order
        inverse = torch.empty_like(order)
        inverse[order] = torch.arange(n, dtype=order.dtype, device=order.device)
        plan.inverse = inverse
        plan.n_recv = sum(plan.recv_counts)
        send_rows = torch.div(ids, G, rounding_mode="floor")[order].contiguous()
        plan.recv_rows = self.to_owner(plan, send_rows)
        return plan

    def to_owner(self, plan, x):
        """x: [n_send, ...] in partition order -> [n_recv, ...] grouped by source rank."""
        if self.world == 1:
            return x
        out = x.new_empty((plan.n_recv,) + tuple(x.shape[1:]))
        self._a2a(out, x.contiguous(), plan.recv_counts, plan.send_counts)
        return out

    def to_source(self, plan, y, out=None):
        """y: [n_recv, ...] in the order of plan.recv_rows -> [n_send, ...] in partition order
        (into `out` when given)."""
        if self.world == 1:
            if out is None:
                return y
            out.copy_(y)
            return out
        if out is None:
            out = y.new_empty((plan.n_send,) + tuple(y.shape[1:]))
        self._a2a(out, y.contiguous(), plan.send_counts, plan.recv_counts)
        return out


class ShardedXDeepFMEngine(XDeepFMEngine):
    """XDeepFMEngine whose table shard is 1/world of the rows.  `train_step(batch)` takes this
    rank's part of the global batch; all ranks must call it with equally sized batches."""

    def __init__(self, hp, device=None, seed=None, group=None, virtual_world=1):
        self.exchange = RowShardExchange(group, virtual_world=virtual_world)
        self.world, self.rank = self.exchange.world, self.exchange.rank
        if hp.loss == "square_loss":
            raise ValueError("square_loss (a root of a mean) cannot be split over ranks")
        if hp.enable_BN is True:
            raise NotImplementedError("batch statistics over a batch split between ranks are not built")
        self.global_rows = int(hp.FEATURE_COUNT)
        rows = local_rows(self.global_rows, self.rank, self.world)
        XDeepFMEngine.__init__(self, hp, device=device, seed=None if seed is None else seed + 7919 * self.rank,
                               embedding_update="sparse_lazy", table_rows=max(rows, 1))
        if self.world > 1:
            self.exchange.broadcast(self.dense, 0)  # replicated parameters start identical
        self.dropout_seed ^= (0x632BE59BD9B4E019 * self.rank) & (2 ** 64 - 1)  # ranks draw different masks
        self._consts = {}
        self._owner_ws = {}
        self._route_stream = None
        self._route_pre = None
        # forward + backward + per-token gradients have fixed shapes once the rows are fetched:
        # they replay as one HIP graph between the (variable-size) exchanges
        self.graph_local = True
        self._local_graphs = {}

    # ------------------------------------------------------------------ helpers
    def _arange32(self, n):
        t = self._consts.get("arange32")
        if t is None or t.numel() < n + 1:
            t = torch.arange(max(n + 1, 1024), dtype=torch.int32, device=self.device)
            self._consts["arange32"] = t
        return t

    def _ones(self, n):
        t = self._consts.get("ones")
        if t is None or t.numel() < n:
            t = torch.ones((max(n, 1024),), dtype=torch.float32, device=self.device)
            self._consts["ones"] = t
        return t

    def _owner_buf(self, name, numel, dtype=torch.float32):
        t = self._owner_ws.get(name)
        if t is None or t.numel() < numel or t.dtype != dtype:
            t = torch.empty((max(int(numel * 1.25), 1),), dtype=dtype, device=self.device)
            self._owner_ws[name] = t
        return t

    # -------------------------------------------------------------------- step
    def fetch_rows(self, batch, static=None):
        """Route the batch's ids, gather at the owners, bring the rows back.  Afterwards the
        batch reads its embeddings from the fetched rows (token j -> fetched row inverse[j]).
        static: a _LocalGraph whose buffers receive the rows instead of fresh tensors."""
        hp, lib, st = self.hp, self.lib, self._stream()
        D = self.D
        use_lin = bool(hp.use_Linear_part)
        plan = self._take_prefetched_route(batch.ids)
        if plan is None:
            plan = self.exchange.route(batch.ids)
        n_recv = plan.n_recv
        # [row | linear weight | pad] records: rows and weights come back in one all-to-all
        DP = D + 4 if use_lin else D
        comb = torch.empty((n_recv, DP), dtype=torch.float32, device=self.device)
        if n_recv:
            ev = self._mark("owner_gather")
            _lib.check(
                lib.xdfm_embed_gather(
                    _ptr(self.table), _ptr(self.lin_w) if use_lin else None, self.table_stride, self.lin_stride,
                    self.N, D, _ptr(plan.recv_rows), n_recv, _ptr(comb), _ptr(comb[:, D]) if use_lin else None,
                    DP if use_lin else 0, _ptr(self.err_flag), st,
                )
            )
            self._done(ev)
        batch.route = plan
        # owner-side sort of the received row ids: runs beside the forward pass
        self._owner_plan_async(batch)
        if static is not None:
            self.exchange.to_source(plan, comb, out=static.src_comb)
            static.src_ids.copy_(self._arange64(batch.nnz)[:batch.nnz] if plan.inverse is None else plan.inverse)
            return
        src = self.exchange.to_source(plan, comb)
        batch.src_table = src[:, :D]
        batch.src_lin = src[:, D] if use_lin else None
        if plan.inverse is None:
            batch.src_ids = self._arange64(batch.nnz)
        else:
            batch.src_ids = plan.inverse
        batch.src_N = max(batch.nnz, 1)

    def prefetch_route(self, ids):
        """Route the ids of the NEXT batch (ids to their owners, exchange sizes to the host) on a
        stream of its own while this step computes: the one host synchronisation of a step then
        waits for a few small kernels instead of for everything enqueued before it, and the host
        can enqueue the next step while the GPU still works on this one.  Every rank must call
        it at the same point of its step (it issues collectives)."""
        if ids is None or (self.world == 1 and self.exchange.virtual_world == 1):
            return
        if self._route_stream is None:
            self._route_stream = torch.cuda.Stream(device=self.device)
        rs = self._route_stream
        with torch.cuda.stream(rs):  # the ids of a later batch do not depend on this step
            plan = self.exchange.route(ids)
            done = torch.cuda.Event()
            done.record(rs)
        self._route_pre = (ids, plan, done)

    def _take_prefetched_route(self, ids):
        pre, self._route_pre = self._route_pre, None
        if pre is None or pre[0] is not ids:
            return None
        _, plan, done = pre
        main = torch.cuda.current_stream(self.device)
        main.wait_event(done)
        for t in (plan.order, plan.inverse, plan.recv_rows):
            if torch.is_tensor(t) and t.is_cuda:
                t.record_stream(main)  # allocated on the route stream, read on this one
        return plan

    def _arange64(self, n):
        t = self._consts.get("arange64")
        if t is None or t.numel() < n:
            t = torch.arange(max(n, 1024), dtype=torch.int64, device=self.device)
            self._consts["arange64"] = t
        return t

    def _owner_plan_async(self, batch):
        plan = batch.route
        n = plan.n_recv
        batch.owner_sorted_ids = torch.empty((max(n, 1),), dtype=torch.int32, device=self.device)
        batch.owner_sorted_tok = torch.empty((max(n, 1),), dtype=torch.int32, device=self.device)
        batch.owner_event = None
        if n == 0:
            return
        sz = ctypes.c_size_t()
        _lib.check(self.lib.xdfm_embed_plan_workspace_bytes(n, self.N, ctypes.byref(sz)))
        ws = self._owner_buf("plan_ws", sz.value, torch.uint8)
        main = torch.cuda.current_stream(self.device)
        if self._side_stream is None:
            self._side_stream = torch.cuda.Stream(device=self.device)
        ready = torch.cuda.Event()
        ready.record(main)
        self._side_stream.wait_event(ready)
        with torch.cuda.stream(self._side_stream):
            ev = self._mark("owner_plan")
            _lib.check(
                self.lib.xdfm_embed_plan(_ptr(plan.recv_rows), n, self.N, _ptr(batch.owner_sorted_ids),
                                         _ptr(batch.owner_sorted_tok), _ptr(ws), ws.numel(), self._stream())
            )
            self._done(ev)
            batch.owner_event = torch.cuda.Event()
            batch.owner_event.record(self._side_stream)

    def _grad_records(self, nnz):
        """[gradient row | linear gradient | pad] per token: one buffer, one all-to-all."""
        DP = self.D + 4 if self.hp.use_Linear_part else self.D
        return torch.empty((max(nnz, 1), DP), dtype=torch.float32, device=self.device)

    def _token_grads(self, batch, order32, send_c):
        """Per-token gradient records in partition order (segment i = token order[i]) into
        send_c; `batch` carries src_table, tok_row, weights."""
        hp, lib, ws, st = self.hp, self.lib, self.ws, self._stream()
        D, nnz = self.D, batch.nnz
        use_lin, use_fm = bool(hp.use_Linear_part), bool(hp.use_FM_part)
        if not nnz:
            return
        opt = self._opt_args()
        ev = self._mark("token_grads")
        _lib.check(
            lib.xdfm_embed_bwd(
                1, _ptr(batch.src_table), None, None, None, None, None, batch.src_table.stride(0), 0, _ptr(send_c),
                _ptr(send_c[:, D]) if use_lin else None, max(nnz, 1), D, self.F, batch.B,
                _ptr(self._arange32(nnz)), _ptr(order32), _ptr(batch.tok_row), _ptr(batch.weights), nnz,
                _ptr(ws["d_emb"]), _ptr(ws["g_logit"]) if (use_lin or use_fm) else None,
                _ptr(ws["fm_sum"]) if use_fm else None, 1 if use_fm else 0, ctypes.byref(opt),
                _ptr(ws["pieces"]), _ptr(self.d_emb2()), send_c.stride(0) if use_lin else 0, st,
            )
        )
        self._done(ev)

    def _owner_update(self, batch, send_c):
        """Gradient rows to their owners, then the owner's sorted-key segment sum over what
        arrived + fused sparse row optimizer."""
        hp, lib, st = self.hp, self.lib, self._stream()
        D, nnz = self.D, batch.nnz
        use_lin = bool(hp.use_Linear_part)
        plan = batch.route
        opt = self._opt_args()
        recv_c = self.exchange.to_owner(plan, send_c[:nnz])
        n = plan.n_recv
        if n:
            if batch.owner_event is not None:
                torch.cuda.current_stream(self.device).wait_event(batch.owner_event)
            pieces = self._owner_buf("pieces", 2 * (n // 64 + 1) * (D + 1))
            ev = self._mark("owner_update")
            _lib.check(
                lib.xdfm_embed_bwd(
                    0, _ptr(self.table), _ptr(self.table_m), _ptr(self.table_v),
                    _ptr(self.lin_w) if use_lin else None, _ptr(self.lin_m) if use_lin else None,
                    _ptr(self.lin_v) if use_lin else None, self.table_stride, self.lin_stride, None, None,
                    self.N, D, 1, n,
                    _ptr(batch.owner_sorted_ids), _ptr(batch.owner_sorted_tok), _ptr(self._arange32(n)),
                    _ptr(self._ones(n)), n, _ptr(recv_c), _ptr(recv_c[:, D]) if use_lin else None, None, 0,
                    ctypes.byref(opt), _ptr(pieces), None, recv_c.stride(0) if use_lin else 0, st,
                )
            )
            self._done(ev)

    def _update_embedding(self, batch):
        nnz = batch.nnz
        plan = batch.route
        send_c = self._grad_records(nnz)
        order32 = (self._arange32(nnz) if plan.order is None else plan.order.to(torch.int32))
        self._token_grads(batch, order32, send_c)
        self._owner_update(batch, send_c)
        return True

    # ------------------------------------------------------ graph of the local part
    def _local_graph(self, batch):
        key = (batch.B, batch.nnz)
        g = self._local_graphs.get(key)
        if g is None and len(self._local_graphs) < 4:
            g = _LocalGraph(self, batch)
            self._local_graphs[key] = g
        return g

    def train_step(self, batch, next_ids=None):
        """One optimisation step on this rank's share of the global batch.  Returns this
        rank's (total_loss share, data_loss share) as device scalars: their sum over the ranks
        is the global mean loss (regularisation counted once, on rank 0's share).
        next_ids: the ids tensor of the batch the next call will get (the same object as its
        batch.ids): its routing is started beside this step's compute (prefetch_route)."""
        self.mean_count = batch.B * self.world
        g = None
        if self.graph_local and self.kernel_events is None and batch.nnz > 0:
            g = self._local_graph(batch)  # None: too many shapes seen, stay eager
        if g is not None:
            self.fetch_rows(batch, static=g)
            g.load(batch)
            if g.graph is None:
                try:
                    g.capture()  # no collective in here: ranks may capture at different steps
                except Exception as exc:  # stay correct if this runtime cannot capture the segment
                    import warnings

                    warnings.warn("local HIP graph capture failed (%s); running eagerly" % (exc,))
                    self.graph_local = False
                    self._local_graphs.clear()
                    g.graph = None
            if g.graph is not None:
                g.graph.replay()
                self.prefetch_route(next_ids)
                self._advance_step()
                self._owner_update(batch, g.send_c)
            else:
                # the fetched rows already sit in g's buffers: finish this step without the graph
                batch.src_table, batch.src_lin, batch.src_ids = g.src_table, g.src_lin, g.src_ids
                batch.src_N = max(batch.nnz, 1)
                self.forward(batch, with_loss=True, with_grad=True)
                self.backward(batch)
                self.prefetch_route(next_ids)
                self._advance_step()
                self._update_embedding(batch)
        else:
            self.fetch_rows(batch)
            self.forward(batch, with_loss=True, with_grad=True)
            self.backward(batch)
            self.prefetch_route(next_ids)
            self._advance_step()
            self._update_embedding(batch)
        self.join_late()
        if self.world > 1:
            ev = self._mark("dense_allreduce")
            self.exchange.all_reduce(self.dense_grad)
            self._done(ev)
        self._update_dense(True)
        reg = self.reg_loss if self.rank == 0 else torch.zeros_like(self.reg_loss)
        return (self.data_loss + reg)[0], self.data_loss[0].clone()

    def predict_batch(self, batch):
        self.fetch_rows(batch)
        return self.forward(batch, with_loss=False)

    # ------------------------------------------------------- table import / export
    def load_global_state(self, sd):
        """Load a state dict keyed like XDeepFMEngine.state_dict() holding the FULL table."""
        G, r = self.world, self.rank
        local = dict(sd)
        for key in ("embedding", "lin_w", "slot/table_m", "slot/table_v", "slot/lin_m", "slot/lin_v"):
            if key in sd:
                t = torch.as_tensor(np.asarray(sd[key]) if not torch.is_tensor(sd[key]) else sd[key])
                local[key] = t[r::G]
        self.load_state(local)

    def gather_global_table(self):
        """Full [FEATURE_COUNT, D] table and lin_w on every rank (tests / export)."""
        G = self.world
        n = self.global_rows
        full = torch.zeros((n, self.D), dtype=torch.float32, device=self.device)
        lin = torch.zeros((n,), dtype=torch.float32, device=self.device) if self.lin_w is not None else None
        for r in range(G):
            rows = local_rows(n, r, G)
            buf = self.table[:rows].clone() if r == self.rank else torch.empty((rows, self.D), device=self.device)
            if G > 1:
                self.exchange.broadcast(buf, r)
            full[r::G] = buf
            if lin is not None:
                lb = self.lin_w[:rows].clone() if r == self.rank else torch.empty((rows,), device=self.device)
                if G > 1:
                    self.exchange.broadcast(lb, r)
                lin[r::G] = lb
        return full, lin


class _LocalGraph(object):
    """forward + backward + per-token gradients of one rank for batches of one shape, captured
    as a HIP graph over static buffers (the rows fetched from the owners land in src_table)."""

    def __init__(self, eng, batch):
        from recommenders_amd.models.deeprec.models._engine import DeviceBatch

        dev, nnz, B, D = eng.device, batch.nnz, batch.B, eng.D
        use_lin = bool(eng.hp.use_Linear_part)
        f32 = torch.float32
        self.eng = eng
        self.src_comb = torch.zeros((nnz, D + 4 if use_lin else D), dtype=f32, device=dev)
        self.src_table = self.src_comb[:, :D]
        self.src_lin = self.src_comb[:, D] if use_lin else None
        self.src_ids = torch.zeros((nnz,), dtype=torch.int64, device=dev)
        self.order32 = torch.zeros((nnz,), dtype=torch.int32, device=dev)
        self.send_c = torch.zeros((nnz, D + 4 if use_lin else D), dtype=f32, device=dev)
        sb = DeviceBatch()
        sb.B, sb.nnz = B, nnz
        sb.labels = torch.zeros_like(batch.labels)
        sb.weights = torch.zeros_like(batch.weights)
        sb.row_ptr = torch.zeros_like(batch.row_ptr)
        sb.tok_row = torch.zeros_like(batch.tok_row)
        sb.ids = None  # never read: the embeddings come from src_table
        sb.src_table, sb.src_lin, sb.src_ids, sb.src_N = self.src_table, self.src_lin, self.src_ids, max(nnz, 1)
        sb.sorted_ids = sb.sorted_tok = None
        sb.planned = True
        sb.plan_event = None
        self.batch = sb
        self.graph = None

    def capture(self):
        """A warm-up pass off the capture, then the capture (the inputs hold a real batch)."""
        dev = self.eng.device
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            self._run()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        graph = torch.cuda.CUDAGraph()
        # thread_local: the process group's watchdog thread may touch the runtime meanwhile
        with torch.cuda.graph(graph, capture_error_mode="thread_local"):
            self._run()
        self.graph = graph

    def _run(self):
        eng = self.eng
        eng.forward(self.batch, with_loss=True, with_grad=True)
        eng.backward(self.batch)
        eng._token_grads(self.batch, self.order32, self.send_c)
        eng.join_late()  # every fork is joined before a capture ends

    def load(self, batch):
        """Copy the per-batch arrays (and the partition order) into the graph's inputs."""
        sb, plan = self.batch, batch.route
        sb.labels.copy_(batch.labels)
        sb.weights.copy_(batch.weights)
        sb.row_ptr.copy_(batch.row_ptr)
        sb.tok_row.copy_(batch.tok_row)
        if plan.order is None:
            self.order32.copy_(self.eng._arange32(batch.nnz)[:batch.nnz])
        else:
            self.order32.copy_(plan.order)
