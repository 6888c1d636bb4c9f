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
import os
import time

import numpy as np
import torch
import torch.distributed as dist

from recommenders_amd import _lib
from recommenders_amd.models.deeprec.models._engine import BN_EPS, BN_MOMENTUM, XDeepFMEngine, _ptr

__all__ = ["RowShardExchange", "RoutePlan", "ShardedXDeepFMEngine", "owner_of", "local_rows"]


def owner_of(ids, world):
    return ids % world


def local_rows(n_rows, rank, world):
    """Rows of a FEATURE_COUNT = n_rows table that rank holds (ids rank, rank+G, ...)."""
    return (n_rows - rank + world - 1) // world


class RoutePlan(object):
    """Routing state of one batch.  Tokens are sorted by (owner, local row), duplicates side by side;
    every distinct row is one *slot* of the request sent to its owner.

    order[n_tok]        token index of every sorted position (stable: equal ids keep token order)
    slot_sorted[n_tok]  slot of every sorted position (ascending)
    inverse[n_tok]      slot of every token
    uniq_rows[n_send]   local row (at its owner) of every slot; slots of one owner are contiguous
    send_counts[G]      slots per owner;  recv_counts[G]: slots this rank is asked for, per source
    recv_rows[n_recv]   local rows this rank serves as owner, grouped by source rank
    owner_sorted_ids / owner_sorted_tok: the owner-side sort of recv_rows (engine)"""

    __slots__ = ("n_tok", "order", "slot_sorted", "inverse", "uniq_rows", "send_counts", "recv_counts",
                 "recv_rows", "n_send", "n_recv", "owner_sorted_ids", "owner_sorted_tok", "gen",
                 "fixed", "cap", "gmax", "fit_host", "fit_event", "fit_tagged")

    def __init__(self):
        # fixed: the request has G x cap slots whatever the batch holds (route_fixed): the exchanges
        # need no sizes; send_counts / recv_counts are then None (equal splits)
        self.fixed, self.cap, self.gmax, self.fit_host, self.fit_event = False, 0, None, None, None
        # (a plan routed inside a captured step: the key of the tagged answer fits() polls, see route_fixed)
        self.fit_tagged = None


class _Bufs(object):
    """Grow-only named device buffers: no allocation per step once the batch shape has been seen."""

    def __init__(self, device):
        self.device = device
        self.t = {}

    def get(self, name, numel, dtype):
        t = self.t.get(name)
        if t is None or t.numel() < numel or t.dtype != dtype:
            t = torch.empty((max(int(numel * 1.25), 16),), dtype=dtype, device=self.device)
            self.t[name] = t
        return t[:numel]


class _Captured(object):
    """A fixed launch sequence (kernels of libxdfm.so over buffers that stay where they are) run eagerly the
    first time and replayed as one HIP graph from then on: the host pays one launch instead of a dozen.
    `key` names everything the sequence depends on (sizes, buffer addresses); another key captures anew.
    A runtime that cannot capture the sequence keeps running it eagerly."""

    def __init__(self):
        self.graphs = {}
        self.failed = False

    def run(self, key, fn, device):
        if torch.cuda.is_current_stream_capturing():
            fn()  # part of a larger capture (the whole step): recorded in place
            return
        g = self.graphs.get(key)
        if g is not None:
            g.replay()
            return
        fn()
        if self.failed or os.environ.get("XDFM_SHARD_GRAPHS", "1") == "0":
            return
        try:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                fn()
            if len(self.graphs) >= 16:
                self.graphs.clear()
            self.graphs[key] = g
        except Exception as exc:  # noqa: BLE001
            import warnings

            warnings.warn("HIP graph capture of a routing sequence failed (%s); it stays eager" % (exc,))
            self.failed = True


def _hip_copy_to_host(pinned, src):
    """hipMemcpyAsync device -> pinned host on the current stream (inside a capture: a copy node; torch's own
    non_blocking copy also records an event for its host allocator, which a capture does not want)."""
    nbytes = src.numel() * src.element_size()
    assert pinned.numel() * pinned.element_size() == nbytes and src.is_contiguous() and pinned.is_contiguous()
    st = torch._C._cuda_getCurrentRawStream(src.device.index)
    err = _lib.hip_runtime().hipMemcpyAsync(pinned.data_ptr(), src.data_ptr(), nbytes, _lib.HIP_MEMCPY_DEVICE_TO_HOST, st)
    if err != 0:
        raise RuntimeError("hipMemcpyAsync to pinned memory failed (%d)" % err)


class RowShardExchange(object):
    def __init__(self, group=None, virtual_world=1, force_collectives=False):
        """virtual_world > 1 (single process only, for tests): partition the tokens as if there
        were that many owners, with every "owner" being this process.  The exchanges are then
        identities, but the engine sees the permuted token order of a real multi-rank step.
        force_collectives (tests): issue every collective even at world size 1 (an exchange with
        oneself), so that the RCCL calls of an N-GPU step run on a one-GPU box."""
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.virtual_world = int(virtual_world) if self.world == 1 else 1
        # rehearsal mode: the gloo backend has no all-to-all for device tensors, so several ranks
        # sharing one GPU (tests on a one-GPU box) stage their exchanges through the host
        self.host_staged = self.world > 1 and dist.get_backend(group) == "gloo"
        self.collective = self.world > 1 or bool(force_collectives)
        self._bufs = {}
        self._calls = 0
        # slots per owner of a fixed-capacity request (0: none chosen yet); the same on every rank
        # (it is derived from counts every rank has seen: grow_capacity)
        self.cap = 0
        self._route_graphs = _Captured()
        self._pinned = {}
        self.watchdog_quiet_s = float(os.environ.get("XDFM_SHARD_WATCHDOG_QUIET_S", "0.5"))
        self._fit_bufs = {}    # (generation, G) -> tagged-answer buffers of routings that run inside a step graph
        self._fit_expect = {}  # (generation, G) -> how often that routing has been replayed

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

    def _all_gather_small(self, x):
        """x: small int32 vector, same length on every rank -> [world, len] on the host."""
        if not self.collective:
            return x.cpu().reshape(1, -1)
        if self.host_staged or not x.is_cuda:
            h = x.cpu()
            out = torch.empty((self.world * h.numel(),), dtype=h.dtype)
            dist.all_gather_into_tensor(out, h, group=self.group)
            return out.reshape(self.world, -1)
        out = torch.empty((self.world * x.numel(),), dtype=x.dtype, device=x.device)
        dist.all_gather_into_tensor(out, x, group=self.group)
        return out.cpu().reshape(self.world, -1)

    # ------------------------------------------------------------- partition + de-duplication
    def _partition_host(self, ids, n_rows, G):
        """torch restatement of xdfm_route_keys / xdfm_embed_plan / xdfm_route_uniq for CPU tensors (the
        gloo rehearsal of the N > 1 path); on the GPU box it is checked against the kernels."""
        rpr = (n_rows + G - 1) // G
        idc = ids.clamp(0, n_rows - 1)
        keys = (idc % G) * rpr + torch.div(idc, G, rounding_mode="floor")
        order = torch.argsort(keys, stable=True)
        sk = keys[order]
        n = int(ids.numel())
        head = torch.ones((n,), dtype=torch.bool, device=ids.device)
        if n > 1:
            head[1:] = sk[1:] != sk[:-1]
        slot_sorted = (torch.cumsum(head.to(torch.int64), 0) - 1).to(torch.int32)
        inverse = torch.empty((n,), dtype=torch.int64, device=ids.device)
        inverse[order] = slot_sorted.to(torch.int64)
        uk = sk[head]
        first = torch.full((G + 1,), -1, dtype=torch.int32, device=ids.device)
        if n:
            owner = torch.div(uk, rpr, rounding_mode="floor")
            oh = torch.ones_like(owner, dtype=torch.bool)
            oh[1:] = owner[1:] != owner[:-1]
            first[owner[oh]] = torch.nonzero(oh).reshape(-1).to(torch.int32)
            first[G] = int(uk.numel())
        return order.to(torch.int32), slot_sorted, inverse, uk % rpr, first

    def _partition_device(self, ids, n_rows, G):
        from recommenders_amd.models.deeprec.models._engine import _ptr

        lib = _lib.lib
        n = int(ids.numel())
        dev = ids.device
        slot = self._calls & 1  # the next batch is routed while this one's plan is still in use
        bufs = self._bufs.setdefault((dev, slot), _Bufs(dev))
        st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        rpr = (n_rows + G - 1) // G
        sz = ctypes.c_size_t()
        _lib.check(lib.xdfm_embed_plan_workspace_bytes(n, G * rpr, ctypes.byref(sz)))
        ws = bufs.get("ws", sz.value, torch.uint8)
        _lib.check(lib.xdfm_route_scratch_ints(n, ctypes.byref(sz)))
        scratch = bufs.get("scratch", sz.value, torch.int32)
        keys = bufs.get("keys", max(n, 1), torch.int64)
        sk = bufs.get("sk", max(n, 1), torch.int32)
        order = bufs.get("order", max(n, 1), torch.int32)
        slot_sorted = bufs.get("slot_sorted", max(n, 1), torch.int32)
        inverse = bufs.get("inverse", max(n, 1), torch.int64)
        uniq_rows = bufs.get("uniq_rows", max(n, 1), torch.int64)
        first = bufs.get("first", G + 1, torch.int32)
        _lib.check(lib.xdfm_route_keys(_ptr(ids), n, n_rows, G, _ptr(keys), st))
        _lib.check(lib.xdfm_embed_plan(_ptr(keys), n, G * rpr, _ptr(sk), _ptr(order), _ptr(ws), ws.numel(), st))
        _lib.check(lib.xdfm_route_uniq(_ptr(sk), _ptr(order), n, n_rows, G, _ptr(slot_sorted), _ptr(inverse),
                                       _ptr(uniq_rows), _ptr(first), _ptr(scratch), st))
        return order[:n], slot_sorted[:n], inverse[:n], uniq_rows, first

    @staticmethod
    def capacity_for(gmax, factor=1.25):
        """Slots per owner that hold a request of at most gmax distinct rows per owner with `factor` of
        room, a multiple of 64."""
        want = int(np.ceil(max(int(gmax), 1) * float(factor)))
        return (want + 63) // 64 * 64

    def grow_capacity(self, gmax, factor=1.25):
        """Make room for gmax distinct rows per (source, owner) pair.  gmax must be a number every rank
        holds (the maximum over all ranks), so that the ranks agree on the capacity."""
        cap = self.capacity_for(gmax, factor)
        if cap > self.cap:
            self.cap = cap
        return self.cap

    def _pad(self, uniq_rows, slot_sorted, inverse, first, n, G, cap, pad_row):
        """xdfm_route_pad (device) or its torch restatement (CPU tensors): the request in G blocks of cap + 1
        slots (the last one of a block is its header, see include/xdfm.h)."""
        S = cap + 1
        dev = first.device
        if first.is_cuda:
            from recommenders_amd.models.deeprec.models._engine import _ptr

            bufs = self._bufs.setdefault((dev, self._calls & 1), _Bufs(dev))
            rows_p = bufs.get("rows_padded", G * S, torch.int64)
            ss_p = bufs.get("slot_sorted_p", max(n, 1), torch.int32)
            inv_p = bufs.get("inverse_p", max(n, 1), torch.int64)
            counts = bufs.get("counts", G + 1, torch.int32)
            st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            _lib.check(_lib.lib.xdfm_route_pad(_ptr(first), _ptr(uniq_rows), _ptr(slot_sorted), _ptr(inverse), n, G,
                                               cap, int(pad_row), _ptr(rows_p), _ptr(ss_p), _ptr(inv_p), _ptr(counts),
                                               st))
            return rows_p, ss_p[:n], inv_p[:n], counts
        f = first.tolist()
        start = [0] * (G + 1)
        nxt = 0 if f[G] < 0 else f[G]
        start[G] = nxt
        for g in range(G - 1, -1, -1):
            if f[g] >= 0 and f[G] >= 0:
                nxt = f[g]
            start[g] = nxt
        cnt = [start[g + 1] - start[g] for g in range(G)]
        rows_p = torch.full((G * S,), int(pad_row), dtype=torch.int64)
        for g in range(G):
            k = min(cnt[g], cap)
            rows_p[g * S:g * S + k] = uniq_rows[start[g]:start[g] + k]
            rows_p[g * S + cap] = int(pad_row) | ((max(cnt) if cnt else 0) << 32)
        st_t = torch.tensor(start[:G], dtype=torch.int64)

        def renumber(slot):
            slot = slot.to(torch.int64)
            g = torch.bucketize(slot, torch.tensor(start[1:G], dtype=torch.int64), right=True) if G > 1 else torch.zeros_like(slot)
            k = slot - st_t[g]
            return torch.where(k < cap, g * S + k, torch.zeros_like(k))

        counts = torch.tensor(cnt + [max(cnt) if cnt else 0], dtype=torch.int32)
        return rows_p, renumber(slot_sorted).to(torch.int32), renumber(inverse), counts

    def route_fixed(self, ids, n_rows, pad_row):
        """route() without a host round trip: the tokens are partitioned and de-duplicated as there, but the
        request to every owner has self.cap slots (the owner's scratch row pad_row in the unused ones), so
        the ids travel -- and the rows and gradients will -- in equally sized messages and nothing of the
        step waits for sizes.  Whether the batch FITS (at most cap distinct rows per owner on every rank) is
        answered by fits(plan), from a small all-gather whose result travels to the host on its own: asked a
        step later, it never waits.  A plan that does not fit must not be used (route() serves that batch)."""
        G = self.world
        if self.cap <= 0:
            raise RuntimeError("route_fixed needs a capacity (grow_capacity)")
        self._calls += 1
        plan = RoutePlan()
        n = int(ids.numel())
        plan.n_tok = n
        plan.fixed, plan.cap = True, self.cap
        in_capture = bool(ids.is_cuda and torch.cuda.is_current_stream_capturing())
        if ids.is_cuda and n > 0:
            # partition + padding are a dozen launches over this generation's buffers: replayed as one graph
            # (the ids are copied into a buffer of the generation's first; inside a step graph `ids` IS a
            # buffer that stays where it is and the launches are recorded in place)
            bufs = self._bufs.setdefault((ids.device, self._calls & 1), _Bufs(ids.device))
            if in_capture:
                ids_in = ids.reshape(-1)
            else:
                ids_in = bufs.get("ids_in", n, torch.int64)
                ids_in.copy_(ids)
            res = {}

            def seq():
                order, slot_sorted, inverse, uniq_rows, first = self._partition_device(ids_in, int(n_rows), G)
                res["v"] = (order,) + self._pad(uniq_rows, slot_sorted, inverse, first, n, G, self.cap, pad_row)

            seq_key = ("route", n, int(n_rows), G, self.cap, int(pad_row), self._calls & 1, ids_in.data_ptr())
            hit = self._route_out.get(seq_key) if hasattr(self, "_route_out") else None
            if in_capture:
                seq()
                hit = res["v"]
            elif hit is None:
                self._route_graphs.run(seq_key, seq, ids.device)
                if not hasattr(self, "_route_out"):
                    self._route_out = {}
                if len(self._route_out) >= 16:
                    self._route_out.clear()
                self._route_out[seq_key] = res["v"]
                hit = res["v"]
            else:
                self._route_graphs.run(seq_key, seq, ids.device)
            plan.order, rows_p, plan.slot_sorted, plan.inverse, counts = hit
        else:
            part = self._partition_device if ids.is_cuda else self._partition_host
            plan.order, slot_sorted, inverse, uniq_rows, first = part(ids, int(n_rows), G)
            rows_p, plan.slot_sorted, plan.inverse, counts = self._pad(uniq_rows, slot_sorted, inverse, first, n, G,
                                                                       self.cap, pad_row)
        plan.uniq_rows = rows_p
        plan.n_send = plan.n_recv = G * (self.cap + 1)
        plan.send_counts = plan.recv_counts = None
        # ids to their owners (equal splits); every block's header tells the owner its source's largest count
        raw = self.to_owner(plan, rows_p)
        plan.recv_rows, src_max = self._decode(raw, G, self.cap)
        if src_max.is_cuda and in_capture:
            # no event can tell the host that a part of a graph has run: the answer travels with a count of
            # this routing's replays in the high half of every word, and fits() reads until the count is the
            # one it expects (fit_buffers; 8-byte words of pinned memory, each written whole)
            key = (self._calls & 1, G)
            seq_hi, tagged, pin, _ = self.fit_buffers(key, src_max.device)
            seq_hi.add_(1 << 32)
            tagged.copy_(src_max)
            tagged.add_(seq_hi)
            _hip_copy_to_host(pin, tagged)
            plan.fit_host, plan.fit_tagged = None, key
        elif src_max.is_cuda:
            gen = self._calls & 1
            pin = self._pinned.get((gen, G))
            if pin is None:
                pin = self._pinned[(gen, G)] = torch.empty((G,), dtype=torch.int32, pin_memory=True)
            pin.copy_(src_max, non_blocking=True)
            plan.fit_host = pin
            plan.fit_event = torch.cuda.Event()
            plan.fit_event.record(torch.cuda.current_stream(src_max.device))
        else:
            plan.fit_host = src_max
        return plan

    def _decode(self, raw, G, cap):
        """xdfm_route_decode or its torch form: (row indices with the headers turned into scratch rows, every
        source's largest per-owner count)."""
        if raw.is_cuda:
            from recommenders_amd.models.deeprec.models._engine import _ptr

            bufs = self._bufs.setdefault((raw.device, self._calls & 1), _Bufs(raw.device))
            rows = bufs.get("recv_rows", G * (cap + 1), torch.int64)
            src_max = bufs.get("src_max", G, torch.int32)
            st = ctypes.c_void_p(torch.cuda.current_stream(raw.device).cuda_stream)
            _lib.check(_lib.lib.xdfm_route_decode(_ptr(raw), G, cap, _ptr(rows), _ptr(src_max), st))
            return rows, src_max
        rows = raw & 0xFFFFFFFF
        src_max = (raw.view(G, cap + 1)[:, cap] >> 32).to(torch.int32)
        return rows, src_max

    def fit_buffers(self, key, device):
        """(replay count << 32 on the device, tagged answer on the device, its pinned copy, numpy view of that)
        of the routing of generation key[0] inside a step graph.  Made outside the capture."""
        hit = self._fit_bufs.get(key)
        if hit is None:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("fit_buffers must exist before the capture")
            G = key[1]
            pin = torch.zeros((G,), dtype=torch.int64).pin_memory()
            hit = self._fit_bufs[key] = (torch.zeros((1,), dtype=torch.int64, device=device),
                                         torch.zeros((G,), dtype=torch.int64, device=device), pin, pin.numpy())
            self._fit_expect[key] = 0
        return hit

    def fits(self, plan):
        """Whether a fixed-capacity plan holds its batch on every rank (plan.gmax = the largest number of
        distinct rows any rank asks of any owner)."""
        if plan.fit_tagged is not None:
            view, want = self._fit_bufs[plan.fit_tagged][3], self._fit_expect[plan.fit_tagged]
            t0 = time.monotonic()
            spins = 0
            while True:
                v = view.copy()
                if bool(((v >> 32) == want).all()):
                    break
                spins += 1
                if spins % 256 == 0:
                    time.sleep(0)  # (let a reader thread of the caller's have the interpreter)
                    if time.monotonic() - t0 > 120.0:
                        raise RuntimeError("the routing inside the captured step did not report (replay %d)" % want)
            plan.gmax = int((v & 0xFFFFFFFF).max())
            return plan.gmax <= plan.cap
        if plan.fit_event is not None:
            plan.fit_event.synchronize()
        plan.gmax = int(plan.fit_host.max())
        return plan.gmax <= plan.cap

    def route(self, ids, n_rows):
        """ids: int64 [n] global feature ids of this rank's tokens; n_rows: FEATURE_COUNT.  Sorts the
        tokens by (owner, row), de-duplicates, sends every distinct row to its owner (as a local
        row index) and returns the RoutePlan.  One collective for the split sizes (+ one host
        synchronisation), one all-to-all for the rows."""
        G = self.world
        V = G if G > 1 else self.virtual_world
        self._calls += 1
        plan = RoutePlan()
        n = int(ids.numel())
        plan.n_tok = n
        part = self._partition_device if ids.is_cuda else self._partition_host
        plan.order, plan.slot_sorted, plan.inverse, uniq_rows, first = part(ids, int(n_rows), V)
        table = self._all_gather_small(first).tolist()  # the step's one host sync: exchange sizes

        def counts_of(row):
            starts = list(row) if row[V] >= 0 else [0] * (V + 1)  # [V] < 0: that rank holds no token
            for g in range(V - 1, -1, -1):  # an owner without a row starts where the next one does
                if starts[g] < 0:
                    starts[g] = starts[g + 1]
            return [starts[g + 1] - starts[g] for g in range(V)], starts

        mine, starts = counts_of(table[self.rank])
        plan.gmax = max(max(counts_of(row)[0]) for row in table)  # the same number on every rank
        plan.n_send = starts[V]
        plan.uniq_rows = uniq_rows[:plan.n_send]
        if not self.collective:
            plan.send_counts = plan.recv_counts = [plan.n_send]
            plan.n_recv = plan.n_send
            if V > 1:
                # every virtual owner is this process and the table is whole: ask for global ids
                owner = torch.empty((plan.n_send,), dtype=torch.int64, device=ids.device)
                for g in range(V):
                    owner[starts[g]:starts[g + 1]] = g
                plan.recv_rows = plan.uniq_rows * V + owner
            else:
                plan.recv_rows = plan.uniq_rows
            return plan
        plan.send_counts = mine
        plan.recv_counts = [counts_of(table[src])[0][self.rank] for src in range(G)]
        plan.n_recv = sum(plan.recv_counts)
        plan.recv_rows = self.to_owner(plan, plan.uniq_rows)
        return plan

    def to_owner(self, plan, x, out=None):
        """x: [n_send, ...] in slot order -> [n_recv, ...] grouped by source rank."""
        if not self.collective:
            if out is None:
                return x
            out.copy_(x)
            return out
        if out is None:
            out = self._exchange_buf("to_owner", x, plan.n_recv)
        self._a2a(out, x.contiguous(), plan.recv_counts, plan.send_counts)
        return out

    def to_source(self, plan, y, out=None):
        """y: [n_recv, ...] in the order of plan.recv_rows -> [n_send, ...] in slot order
        (into `out` when given)."""
        if not self.collective:
            if out is None:
                return y
            out.copy_(y)
            return out
        if out is None:
            out = self._exchange_buf("to_source", y, plan.n_send)
        self._a2a(out, y.contiguous(), plan.send_counts, plan.recv_counts)
        return out

    def _exchange_buf(self, name, like, rows):
        """Grow-only receive buffer of a call that named none (two generations: the routing of the next
        batch runs while this one's buffers are in use)."""
        width = 1
        for d in like.shape[1:]:
            width *= int(d)
        key = (like.device, "x", self._calls & 1)
        bufs = self._bufs.setdefault(key, _Bufs(like.device))
        flat = bufs.get("%s/%s/%d" % (name, like.dtype, width), max(rows * width, 1), like.dtype)
        return flat[:rows * width].view((rows,) + tuple(like.shape[1:]))


class ShardedXDeepFMEngine(XDeepFMEngine):
    """XDeepFMEngine whose table shard is 1/world of the rows.  `train_step(batch)` takes this
    rank's part of the global batch; all ranks must call it with equally sized batches."""

    def __init__(self, hp, device=None, seed=None, group=None, virtual_world=1, embedding_update="sparse_rows",
                 force_collectives=False, compute_dtype=None, uniform_batches=False):
        """uniform_batches: the caller's promise that every batch on every rank has the same number of samples
        and tokens (synthetic or padded data).  With more than one rank the whole-step graph (_StepGraph) needs
        it: a rank replays that graph when its own consecutive batches agree in shape, the graph issues the
        step's collectives in another order than the eager step does, and all ranks must take the same path."""
        self.exchange = RowShardExchange(group, virtual_world=virtual_world, force_collectives=force_collectives)
        self.world, self.rank = self.exchange.world, self.exchange.rank
        if embedding_update == "dense_exact" and hp.is_clip_norm:
            raise ValueError("the row-sharded dense-exact update does not clip (clip_by_norm of the table needs the "
                             "norm of the whole table's gradient)")
        self.global_rows = int(hp.FEATURE_COUNT)
        rows = local_rows(self.global_rows, self.rank, self.world)
        # one scratch row past the real ones, at the SAME local index on every rank (ranks hold
        # ceil(N / G) or one row less): the unused slots of a fixed-capacity request name it
        self.n_local = max(rows, 1)
        self.pad_row = max((self.global_rows + self.world - 1) // self.world, 1)
        XDeepFMEngine.__init__(self, hp, device=device, seed=None if seed is None else seed + 7919 * self.rank,
                               embedding_update=embedding_update, table_rows=self.pad_row + 1,
                               compute_dtype=compute_dtype)
        # requests of a fixed number of slots per owner: no sizes from the host, the step's exchanges move
        # equally sized messages (XDFM_SHARD_FIXED=0: sized requests, one host synchronisation per step)
        self.fixed = os.environ.get("XDFM_SHARD_FIXED", "1") != "0" and self.exchange.virtual_world == 1
        self.cap_factor = float(os.environ.get("XDFM_SHARD_CAP_FACTOR", "1.25"))
        self.fixed_steps = self.sized_steps = 0
        # dense_exact: the regularisation loss of this rank's rows (summed over the ranks by the caller)
        self.reg_table = torch.zeros((1,), dtype=torch.float32, device=self.device)
        if self.world > 1:
            self.exchange.broadcast(self.dense, 0)  # replicated parameters start identical
        self.dropout_seed ^= (0x632BE59BD9B4E019 * self.rank) & (2 ** 64 - 1)  # ranks draw different masks
        self._consts = {}
        self._owner_graphs = _Captured()
        self._owner_ws = _Bufs(self.device)
        self._route_stream = None
        self._route_pre = None
        self._gen_free = {}  # routing-buffer generation -> event after the last step that read it
        # forward + backward + per-slot gradients have fixed shapes once the rows are fetched:
        # they replay as one HIP graph between the (variable-size) exchanges
        self.graph_local = True
        # square_loss is a root of a mean over the GLOBAL batch: the ranks' sums of squares meet in an
        # all-reduce between the forward and the backward pass (_square_loss_global), so that span
        # cannot be one graph
        self._square = hp.loss == "square_loss"
        if self._square and self.exchange.collective:
            self.graph_local = False
        # enable_BN: the batch statistics are those of the global batch (the reference is one process,
        # xDeepFM.py:234-240, :497-503): every normalisation has one all-reduce in the forward pass
        # (moments) and one in the backward pass (sums), see _bn_fwd / _bn_bwd
        if self.use_bn and self.exchange.collective:
            self.graph_local = False
        self._local_graphs = {}
        # with fixed-capacity requests and the next batch announced, the WHOLE step (exchanges, owner-side
        # work and the next batch's routing included) replays as one HIP graph per routing generation
        # (_StepGraph; XDFM_SHARD_STEP_GRAPH=0: the local part only, exchanges eager)
        # More than one rank: opt-in (XDFM_SHARD_STEP_GRAPH=1 AND uniform_batches).  The captured collectives
        # have run at world size 1 only (a one-GPU box; RCCL refuses two ranks on one device), and a path that
        # could leave N ranks waiting for each other is not made the default on that evidence.
        want = os.environ.get("XDFM_SHARD_STEP_GRAPH", "1" if self.world == 1 else "0") != "0"
        self.graph_step = want and (self.world == 1 or bool(uniform_batches))
        self.step_route_at = int(os.environ.get("XDFM_SHARD_ROUTE_AT", "0"))
        self._step_graphs = {}
        self._after_step_graph = False
        self.graph_steps = 0
        if self.graph_step and self.exchange.collective:
            # a process that ends with captured collectives alive hangs in the process group's destructor
            # (seen with an engine held by a module global): drop the graphs before the interpreter goes
            import atexit
            import weakref

            ref = weakref.ref(self)

            def _close_at_exit():
                eng = ref()
                if eng is not None and eng._step_graphs:
                    eng.close()

            atexit.register(_close_at_exit)

    # ------------------------------------------------------------------ scratch row
    _ROW_KEYS = ("embedding", "lin_w", "slot/table_m", "slot/table_v", "slot/lin_m", "slot/lin_v")

    def _clear_scratch_row(self):
        """The rows past this rank's real ones (the scratch row at index pad_row, and the one before it on a
        rank that holds one row less) hold zeros in every array: they take zero gradients only, so they stay
        zero under every optimizer and add nothing to a regularisation loss."""
        from recommenders_amd.models.deeprec.models._engine import _SLOT_M_INIT

        r = self.n_local
        for t in (self.table, self.lin_w, self.table_v, self.lin_v):
            if t is not None:
                t[r:].zero_()
        for t in (self.table_m, self.lin_m):  # accumulators keep their starting value (Adagrad / FTRL 0.1, RMSProp 1)
            if t is not None:
                t[r:].fill_(_SLOT_M_INIT.get(self.opt_kind, 0.0))
        if self.row_step is not None:
            self.row_step[r:] = 0

    def init_parameters(self, seed=None):
        XDeepFMEngine.init_parameters(self, seed)
        self._clear_scratch_row()

    def load_state(self, sd, strict=True):
        """load_state of this rank's rows: arrays of n_local rows get the scratch row appended."""
        local = dict(sd)
        for key in self._ROW_KEYS:
            if key in local:
                t = torch.as_tensor(np.asarray(local[key]) if not torch.is_tensor(local[key]) else local[key])
                if t.shape[0] == self.n_local:
                    t = torch.cat([t, torch.zeros((self.N - self.n_local,) + tuple(t.shape[1:]), dtype=t.dtype,
                                                  device=t.device)])
                local[key] = t
        XDeepFMEngine.load_state(self, local, strict=strict)
        self._clear_scratch_row()

    def state_dict(self):
        """This rank's rows (without the scratch row) and the replicated parameters."""
        sd = XDeepFMEngine.state_dict(self)
        for key in self._ROW_KEYS:
            if key in sd:
                sd[key] = sd[key][:self.n_local]
        return sd

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
        return self._owner_ws.get(name, max(int(numel), 1), dtype)

    # -------------------------------------------------------------------- step
    def _prepare(self, batch, sized=False):
        """Route the batch's ids and sort, at this owner, the rows it is asked for (current stream).
        With a capacity chosen the request has a fixed number of slots and nothing here waits for the
        device; sized=True (or no capacity yet) sends exactly the distinct rows, which costs one host
        synchronisation for the sizes."""
        ex = self.exchange
        if self.fixed and ex.cap > 0 and not sized:
            plan = ex.route_fixed(batch.ids, self.global_rows, self.pad_row)
        else:
            plan = ex.route(batch.ids, self.global_rows)
            if self.fixed and plan.gmax is not None:
                ex.grow_capacity(plan.gmax, self.cap_factor)  # room for the batches to come
        n = plan.n_recv
        # two generations alive at a time: this step's and the prefetched next one's
        gen = plan.gen = self.exchange._calls & 1
        plan.owner_sorted_ids = self._owner_buf("osid%d" % gen, n, torch.int32)
        plan.owner_sorted_tok = self._owner_buf("ostk%d" % gen, n, torch.int32)
        if n:
            sz = ctypes.c_size_t()
            _lib.check(self.lib.xdfm_embed_plan_workspace_bytes(n, self.N, ctypes.byref(sz)))
            ws = self._owner_buf("plan_ws%d" % gen, sz.value, torch.uint8)

            def sort():
                _lib.check(
                    self.lib.xdfm_embed_plan(_ptr(plan.recv_rows), n, self.N, _ptr(plan.owner_sorted_ids),
                                             _ptr(plan.owner_sorted_tok), _ptr(ws), ws.numel(), self._stream())
                )

            if plan.fixed and self.kernel_events is None:
                # a fixed-capacity request arrives in the same buffer every other step: the sort's six launches
                # replay as one graph
                self._owner_graphs.run(("osort", n, self.N, gen, plan.recv_rows.data_ptr(),
                                        plan.owner_sorted_ids.data_ptr(), ws.data_ptr()), sort, self.device)
            else:
                ev = self._mark("owner_plan")
                sort()
                self._done(ev)
        return plan

    def resolve_plan(self, batch):
        """The routing of `batch`: the one prefetched for it, or made now; a fixed-capacity request that
        turns out not to hold the batch is replaced by a sized one."""
        plan = self._take_prefetched_route(batch)
        if plan is None:
            plan = self._prepare(batch)
        if plan.fixed and not self.exchange.fits(plan):
            # more distinct rows for some owner than a request holds: this batch goes with sized
            # requests, the ones to come get more room (every rank reads the same maximum: same decision)
            self.exchange.grow_capacity(plan.gmax, self.cap_factor)
            plan = self._prepare(batch, sized=True)
        if plan.fixed:
            self.fixed_steps += 1
        else:
            self.sized_steps += 1
        return plan

    def fetch_rows(self, batch, static=None, train=True, plan=None):
        """Route the batch's ids (unless that was prefetched), gather the requested rows at their owners (aged
        to the current step as they are read), bring them back.  Afterwards the batch reads its embeddings from the fetched rows
        (token j -> fetched row inverse[j]).  static: a _LocalGraph whose buffers receive the rows."""
        hp, lib, st = self.hp, self.lib, self._stream()
        D = self.D
        use_lin = bool(hp.use_Linear_part)
        if plan is None:
            plan = self.resolve_plan(batch)
        n_recv = plan.n_recv
        # [row | linear weight | pad] records: rows and weights come back in one all-to-all
        DP = D + 4 if use_lin else D
        comb = self._owner_buf("comb", n_recv * DP).view(n_recv, DP) if n_recv else torch.empty(
            (0, DP), dtype=torch.float32, device=self.device)
        if n_recv:
            ev = self._mark("owner_gather")
            _lib.check(
                lib.xdfm_embed_gather(
                    _ptr(self.table), _ptr(self.lin_w) if use_lin else None, self.table_stride, self.lin_stride,
                    self.N, D, _ptr(plan.recv_rows), n_recv, _ptr(comb), _ptr(comb[:, D]) if use_lin else None,
                    DP if use_lin else 0, _ptr(self.err_flag), self._age(0), st,
                )
            )
            self._done(ev)
        batch.route = plan
        if static is not None:
            self.exchange.to_source(plan, comb, out=static.src_comb[:plan.n_send])
            return
        src = self.exchange.to_source(plan, comb, out=self._owner_buf("src", plan.n_send * DP).view(plan.n_send, DP)
                                      if plan.n_send else None)
        batch.src_table = src[:, :D]
        batch.src_lin = src[:, D] if use_lin else None
        batch.src_ids = plan.inverse
        batch.src_N = max(plan.n_send, 1)

    def prefetch_route(self, next_batch):
        """Route the NEXT batch (ids to their owners, exchange sizes to the host, owner-side sort) on a
        stream of its own while this step computes: the one host synchronisation of a step then
        waits for a few small kernels instead of for everything enqueued before it, and the host
        can enqueue the next step while the GPU still works on this one.  Every rank must call
        it at the same point of its step (it issues collectives) and must hand exactly that batch
        to its next train_step."""
        if next_batch is None:
            return
        if self._route_stream is None:
            self._route_stream = torch.cuda.Stream(device=self.device)
        rs = self._route_stream
        ready = getattr(next_batch, "ready_event", None)
        if self._after_step_graph:
            # the last step was one graph on the current stream (no per-generation events): wait for all of it
            rs.wait_stream(torch.cuda.current_stream(self.device))
            self._after_step_graph = False
        if ready is not None:
            rs.wait_event(ready)  # the copy / kernels that produced the batch's arrays
        else:
            rs.wait_stream(torch.cuda.current_stream(self.device))
        free = self._gen_free.get((self.exchange._calls + 1) & 1)
        if free is not None:
            rs.wait_event(free)  # the step before this one still read that generation's buffers
        with torch.cuda.stream(rs):
            plan = self._prepare(next_batch)
            done = torch.cuda.Event()
            done.record(rs)
        self._route_pre = (next_batch, plan, done)

    def _take_prefetched_route(self, batch):
        pre, self._route_pre = self._route_pre, None
        if pre is None:
            return None
        if pre[0] is not batch:
            # routing issues collectives: a rank that silently routed again would fall out of step
            # with its peers
            raise RuntimeError("train_step got a different batch than the one announced as next_batch")
        _, plan, done = pre
        if done is not None:  # (None: routed inside the previous step's graph, on this stream)
            torch.cuda.current_stream(self.device).wait_event(done)
        return plan

    def _token_grads(self, batch, seg_ids, order32, send_c):
        """Per-slot gradient records (the tokens of one distinct row pre-reduced in token order) into
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
                _ptr(send_c[:, D]) if use_lin else None, max(int(send_c.shape[0]), 1), D, self.F, batch.B,
                _ptr(seg_ids), _ptr(order32), _ptr(batch.tok_row), _ptr(batch.weights), nnz,
                _ptr(ws["d_emb"]), _ptr(ws["g_logit"]) if (use_lin or use_fm) else None,
                _ptr(ws["fm_sum"]) if use_fm else None, 1 if use_fm else 0, ctypes.byref(opt),
                _ptr(ws["pieces"]), _ptr(self.d_emb2()), send_c.stride(0) if use_lin else 0, None, st,
            )
        )
        self._done(ev)

    def _owner_update(self, batch, send_c):
        """Gradient records to their owners, then the owner's sorted-key segment sum over what
        arrived (source-rank order: reproducible) + fused sparse row optimizer."""
        hp, lib, st = self.hp, self.lib, self._stream()
        D = self.D
        use_lin = bool(hp.use_Linear_part)
        plan = batch.route
        opt = self._opt_args()
        n = plan.n_recv
        DP = send_c.shape[1]
        recv_c = self.exchange.to_owner(plan, send_c[:plan.n_send],
                                        out=self._owner_buf("recv_c", n * DP).view(n, DP) if n else None)
        if n and not self.sparse:
            self._owner_update_dense(plan, recv_c, n)
        elif n:
            pieces = self._owner_buf("pieces", 2 * (n // 64 + 1) * (D + 2))
            ev = self._mark("owner_update")
            _lib.check(
                lib.xdfm_embed_bwd(
                    0, _ptr(self.table), _ptr(self.table_m), _ptr(self.table_v),
                    _ptr(self.lin_w) if use_lin else None, _ptr(self.lin_m) if use_lin else None,
                    _ptr(self.lin_v) if use_lin else None, self.table_stride, self.lin_stride, None, None,
                    self.N, D, 1, n,
                    _ptr(plan.owner_sorted_ids), _ptr(plan.owner_sorted_tok), _ptr(self._arange32(n)),
                    _ptr(self._ones(n)), n, _ptr(recv_c), _ptr(recv_c[:, D]) if use_lin else None, None, 0,
                    ctypes.byref(opt), _ptr(pieces), None, recv_c.stride(0) if use_lin else 0, self._age(-1), st,
                )
            )
            self._done(ev)
            if self.catchup_on:
                self._rows_stale = True

    def _owner_update_dense(self, plan, recv_c, n):
        """The reference's dense update at the owner (dense_exact; SURVEY.md 8(e): regularisers and the
        dense-exact optimizer are shard-local): the records that arrived are summed per row into this
        rank's dense gradient image (xdfm_embed_bwd mode 2), then apply_gradients streams this rank's rows
        with embed_l2 / embed_l1 (lin_w: layer_l2 / layer_l1) folded in, as base_model.py:121-149, :322 do
        for the whole table; the regularisation loss of these rows goes to reg_table (the ranks' values add
        up to the reference's term)."""
        hp, lib, st = self.hp, self.lib, self._stream()
        D = self.D
        use_lin = bool(hp.use_Linear_part)
        opt = self._opt_args()
        pieces = self._owner_buf("pieces", 2 * (n // 64 + 1) * (D + 2))
        ev = self._mark("owner_update")
        _lib.check(
            lib.xdfm_embed_bwd(
                2, None, None, None, None, None, None, 0, 0, _ptr(self.grad_table),
                _ptr(self.grad_lin) if use_lin else None, self.N, D, 1, n,
                _ptr(plan.owner_sorted_ids), _ptr(plan.owner_sorted_tok), _ptr(self._arange32(n)),
                _ptr(self._ones(n)), n, _ptr(recv_c), _ptr(recv_c[:, D]) if use_lin else None, None, 0,
                ctypes.byref(opt), _ptr(pieces), None, recv_c.stride(0) if use_lin else 0, None, st,
            )
        )
        one = np.asarray([0, self.N * D], dtype=np.int64)
        _lib.check(
            lib.xdfm_dense_update(
                _ptr(self.table), _ptr(self.grad_table), _ptr(self.table_m), _ptr(self.table_v), 1,
                one.ctypes.data_as(ctypes.c_void_p),
                np.asarray([hp.embed_l2], dtype=np.float32).ctypes.data_as(ctypes.c_void_p),
                np.asarray([hp.embed_l1], dtype=np.float32).ctypes.data_as(ctypes.c_void_p),
                None, ctypes.byref(opt), _ptr(self.stats), _ptr(self.reg_table), 0, st,
            )
        )
        if use_lin:
            one = np.asarray([0, self.N], dtype=np.int64)
            _lib.check(
                lib.xdfm_dense_update(
                    _ptr(self.lin_w), _ptr(self.grad_lin), _ptr(self.lin_m), _ptr(self.lin_v), 1,
                    one.ctypes.data_as(ctypes.c_void_p),
                    np.asarray([hp.layer_l2], dtype=np.float32).ctypes.data_as(ctypes.c_void_p),
                    np.asarray([hp.layer_l1], dtype=np.float32).ctypes.data_as(ctypes.c_void_p),
                    None, ctypes.byref(opt), _ptr(self.stats), _ptr(self.reg_table), 1, st,
                )
            )
        _lib.check(lib.xdfm_embed_clear_rows(_ptr(self.grad_table), _ptr(self.grad_lin) if use_lin else None, D,
                                             _ptr(plan.owner_sorted_ids), n, st))
        self._done(ev)

    def _update_embedding(self, batch):
        nnz = batch.nnz
        plan = batch.route
        DP = self.D + 4 if self.hp.use_Linear_part else self.D
        slots = max(plan.n_send if plan.fixed else nnz, 1)
        send_c = self._owner_buf("send_c", slots * DP).view(slots, DP)
        if plan.fixed:
            send_c.zero_()
        self._token_grads(batch, plan.slot_sorted, plan.order, send_c)
        self._owner_update(batch, send_c)
        return True

    # ------------------------------------------------------ graph of the local part
    def _local_graph(self, batch, plan):
        # rows arrive one per slot: as many as the batch has distinct rows at most (sized requests) or the
        # request's fixed number
        slots = plan.n_send if plan.fixed else batch.nnz
        # a fixed-capacity plan lives in its generation's buffers (the same addresses every other step): the
        # graph reads the partition order / slot maps there instead of from copies
        direct = bool(plan.fixed and plan.order.is_cuda)
        key = (batch.B, batch.nnz, slots, bool(plan.fixed)) + (
            (plan.gen, plan.order.data_ptr(), plan.slot_sorted.data_ptr(), plan.inverse.data_ptr()) if direct else ())
        g = self._local_graphs.get(key)
        if g is not None and g.ws_token != self._ws_token:
            # the workspace was re-allocated (a larger batch came through): its addresses are stale
            self._local_graphs.clear()
            g = None
        if g is None and len(self._local_graphs) < 8:
            g = _LocalGraph(self, batch, slots, bool(plan.fixed), plan if direct else None)
            self._local_graphs[key] = g
        return g

    def _step_graph(self, batch, next_batch, plan, g):
        """The _StepGraph that serves this step, or None: it needs a fixed-capacity plan in its generation's
        buffers, the next batch announced and of the same shape (with its arrays in one flat buffer:
        make_batch), and both generations' local graphs already captured and replayed (so every buffer the
        step touches exists)."""
        if not (self.graph_step and not self.exchange.host_staged and plan.fixed and g.direct
                and g.graph is not None and next_batch is not None
                and next_batch is not batch and next_batch.B == batch.B and next_batch.nnz == batch.nnz
                and getattr(next_batch, "flat", None) is not None and next_batch.flat.numel() == g.flat.numel()
                and getattr(batch, "flat", None) is not None and batch.flat.numel() == g.flat.numel()):
            return None
        other = None
        for h in self._local_graphs.values():
            if (h.direct and h.graph is not None and h.gen == 1 - g.gen and h.slots == g.slots
                    and h.batch.B == batch.B and h.batch.nnz == batch.nnz and h.ws_token == self._ws_token):
                other = h
        if other is None:
            return None
        key = (id(g), id(other), self.exchange.cap)
        sg = self._step_graphs.get(key)
        if sg is not None and sg.ws_token != self._ws_token:
            self._step_graphs.clear()
            sg = None
        if sg is None:
            if len(self._step_graphs) >= 8:
                return None
            sg = self._step_graphs[key] = _StepGraph(self, g, other, plan)
        return sg

    def close(self):
        """Drop the captured steps.  Call it before torch.distributed.destroy_process_group(): graphs that hold
        captured collectives keep the communicator busy, and destroying the group then waits for ever."""
        import gc

        torch.cuda.synchronize(self.device)
        for sg in self._step_graphs.values():
            sg.graph = None
        self._step_graphs.clear()
        self._route_pre = None
        gc.collect()
        torch.cuda.synchronize(self.device)

    def make_batch(self, labels, dnn_idx, ids, weights):
        """XDeepFMEngine.make_batch with the arrays the local graph reads (labels, weights, CSR offsets, token
        rows) as views of ONE flat buffer: handing the batch to the graph is one copy."""
        from recommenders_amd.models.deeprec.models._engine import DeviceBatch

        B, nnz = int(labels.numel()), int(ids.numel())
        lay, flat = _flat_layout(B, nnz, self.F, self.device)
        b = DeviceBatch()
        b.B, b.nnz = B, nnz
        b.flat = flat
        b.labels, b.weights, b.row_ptr, b.tok_row = (_flat_view(flat, lay, k) for k in ("labels", "weights", "row_ptr", "tok_row"))
        # (the ids too: a step graph routes the NEXT batch from the flat buffer it was handed, see _StepGraph)
        b.ids = _flat_view(flat, lay, "ids")[:nnz].view(ids.shape) if nnz and ids.dtype == torch.int64 else ids
        if b.ids is not ids:
            b.ids.copy_(ids)
        b.labels.copy_(labels.reshape(-1))
        if nnz:
            b.weights[:nnz].copy_(weights.reshape(-1))
        _lib.check(
            self.lib.xdfm_build_row_ptr(_ptr(dnn_idx), nnz, B * self.F, _ptr(b.row_ptr), _ptr(b.tok_row),
                                        _ptr(self.err_flag), self._stream())
        )
        b.sorted_ids = b.sorted_tok = b.sorted_row = b.sorted_w = None
        b.planned = b.planned_gather = False
        b.plan_event = None
        b.ready_event = torch.cuda.Event()
        b.ready_event.record(torch.cuda.current_stream(self.device))
        b.route = None
        b.src_table = b.src_lin = b.src_ids = None
        b.src_N = 0
        return b

    def train_step(self, batch, next_batch=None):
        """One optimisation step on this rank's share of the global batch.  Returns this
        rank's (total_loss share, data_loss share) as device scalars: their sum over the ranks
        is the global mean loss (regularisation counted once, on rank 0's share).
        next_batch: the batch the next call will get: its routing is started beside this step's
        compute (prefetch_route)."""
        # (square_loss: every rank first forms its own root mean square and gradient, see below)
        self.mean_count = batch.B if self._square else batch.B * self.world
        self._ensure_ws(batch.B, batch.nnz)
        g = None
        rmse = None
        plan = self.resolve_plan(batch)
        if self.graph_local and self.kernel_events is None and batch.nnz > 0:
            g = self._local_graph(batch, plan)  # None: too many shapes seen, stay eager
        sg = self._step_graph(batch, next_batch, plan, g) if g is not None else None
        if sg is not None:
            return sg.step(batch, next_batch)
        if g is not None:
            self.fetch_rows(batch, static=g, plan=plan)
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
                self._advance_step()
                self._owner_update(batch, g.send_c)
            else:
                # the fetched rows already sit in g's buffers: finish this step without the graph
                batch.src_table, batch.src_lin, batch.src_ids = g.src_table, g.src_lin, g.src_ids
                batch.src_N = max(g.slots, 1)
                self.forward(batch, with_loss=True, with_grad=True)
                rmse = self._square_loss_global(batch)
                self.backward(batch)
                self._advance_step()
                self._update_embedding(batch)
        else:
            self.fetch_rows(batch, plan=plan)
            self.forward(batch, with_loss=True, with_grad=True)
            rmse = self._square_loss_global(batch)
            self.backward(batch)
            self._advance_step()
            self._update_embedding(batch)
        self.join_late()
        if self.exchange.collective:
            ev = self._mark("dense_allreduce")
            self.exchange.all_reduce(self.dense_grad)
            self._done(ev)
        self._update_dense(True)
        # The next batch's routing ends in the step's one host synchronisation (the exchange sizes).  It is
        # started here, with everything of THIS step enqueued: while the host waits for those few kernels the
        # device still has this step's update to run (started right behind the local graph, the wait left
        # the device idle between the graph and the owner-side update: the step was host-bound).
        self.prefetch_route(next_batch)
        done = torch.cuda.Event()
        done.record(torch.cuda.current_stream(self.device))
        self._gen_free[batch.route.gen] = done
        if self._square and self.exchange.collective and rmse is not None:
            self.data_loss.copy_(rmse / self.world)  # shares that add up to the global root mean square
        # this rank's share: the data loss, the replicated parameters' regularisation on rank 0 only, and (dense-exact)
        # the terms of this rank's table rows
        if self.rank != 0:
            self.reg_loss.zero_()
        if not self.sparse:
            self.reg_loss.add_(self.reg_table)
        tot = torch.empty((2,), dtype=torch.float32, device=self.device)
        _lib.check(self.lib.xdfm_loss_total(_ptr(self._loss2), _ptr(tot), self._stream()))  # [data, data + reg]
        return tot[1], tot[0]

    def _bn_fwd(self, x, y, where, k, B, C, inner, act):
        """Batch normalisation over the GLOBAL batch: every rank's (count, mean, sum of squared deviations)
        per channel meet in one all-reduce (each rank fills its own row of a [world, 2C + 1] array of
        zeros, so the sum is exact and the same on all ranks), are combined in rank order
        (mean = sum n_r mean_r / n, M2 = sum M2_r + n_r (mean_r - mean)^2) and xdfm_bn_fwd_moments normalises
        with the result; the moving statistics move identically on every rank."""
        if not (self._training and self.exchange.collective):
            return XDeepFMEngine._bn_fwd(self, x, y, where, k, B, C, inner, act)
        st = self._stream()
        mom = self._owner_buf("bn_mom", 2 * C)
        _lib.check(self.lib.xdfm_bn_local_moments(_ptr(x), B, C, inner, _ptr(mom), st))
        allm = torch.zeros((self.world, 2 * C + 1), dtype=torch.float32, device=self.device)
        allm[self.rank, :2 * C] = mom[:2 * C]
        allm[self.rank, 2 * C] = float(B * inner)
        self.exchange.all_reduce(allm)
        n_r = allm[:, 2 * C:].double()
        mean_r, m2_r = allm[:, :C].double(), allm[:, C:2 * C].double()
        n = n_r.sum()
        mean = (n_r * mean_r).sum(0) / n
        m2 = (m2_r + n_r * (mean_r - mean) ** 2).sum(0)
        mean_var = torch.cat([mean, m2 / n]).float()
        _lib.check(
            self.lib.xdfm_bn_fwd_moments(_ptr(x), B, C, inner, _ptr(self.views["bn_%s_g_%d" % (where, k)]),
                                         _ptr(self.views["bn_%s_o_%d" % (where, k)]),
                                         _ptr(self.bn_state["bnstat_%s_mm_%d" % (where, k)]),
                                         _ptr(self.bn_state["bnstat_%s_mv_%d" % (where, k)]), BN_MOMENTUM, BN_EPS,
                                         _ptr(mean_var), act, _ptr(self.ws["bn_stats_%s_%d" % (where, k)]),
                                         _ptr(y), st)
        )

    def _bn_bwd(self, x, y, dy, where, k, B, C, inner, act):
        """This rank's sums are its share of the gamma / beta gradients (the dense all-reduce adds the
        shares); the gradient w.r.t. the input needs the sums over the global batch: one all-reduce."""
        if not self.exchange.collective:
            return XDeepFMEngine._bn_bwd(self, x, y, dy, where, k, B, C, inner, act)
        st = self._stream()
        stats = self.ws["bn_stats_%s_%d" % (where, k)]
        dgamma, dbeta = self.grad_views["bn_%s_g_%d" % (where, k)], self.grad_views["bn_%s_o_%d" % (where, k)]
        _lib.check(self.lib.xdfm_bn_bwd_sums(_ptr(x), _ptr(y), _ptr(dy), B, C, inner, _ptr(stats), act,
                                             _ptr(dgamma), _ptr(dbeta), st))
        tot = torch.cat([dgamma.reshape(-1), dbeta.reshape(-1)])
        self.exchange.all_reduce(tot)
        # equally sized batches on every rank (train_step's contract): the global count per channel
        n_total = B * inner * self.world
        _lib.check(self.lib.xdfm_bn_bwd_apply(_ptr(x), _ptr(y), _ptr(dy), B, C, inner, n_total,
                                              _ptr(self.views["bn_%s_g_%d" % (where, k)]), _ptr(stats), act,
                                              _ptr(tot), _ptr(tot[C:]), _ptr(dy), st))

    def _square_loss_global(self, batch):
        """square_loss over a batch split between ranks: loss = sqrt(sum over ALL samples of (p - y)^2 /
        B_global) and d loss / d logit_i = (p_i - y_i) p'_i / (B_global loss).  The forward pass has left this
        rank's per-block sums of squares in loss_scratch and the LOCAL gradient (p_i - y_i) p'_i /
        (B_local rmse_local) in g_logit (and, through the CIN head, in d_pooled, which is linear in it):
        one scalar all-reduce gives the global sum, and both arrays are scaled by
        (B_local rmse_local) / (B_global rmse_global).  Returns the global loss (a device scalar), or
        None when nothing is split."""
        if not (self._square and self.exchange.collective):
            return None
        B = batch.B
        blocks = min((B + 255) // 256, 1024)
        sse = self.loss_scratch[:blocks].sum().reshape(1)
        tot = sse.clone()
        self.exchange.all_reduce(tot)
        rmse_l = torch.sqrt(sse / B)
        rmse_g = torch.sqrt(tot / (B * self.world))
        factor = torch.where(rmse_g > 0, rmse_l / (self.world * rmse_g), torch.zeros_like(rmse_g))
        self.ws["g_logit"][:B].mul_(factor)
        if self.hp.use_CIN_part:
            self.ws["d_pooled"][:B].mul_(factor)
        return rmse_g

    def predict_batch(self, batch):
        self.fetch_rows(batch, train=False)
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
        self.flush_rows()
        G = self.world
        n = self.global_rows
        full = torch.zeros((n, self.D), dtype=torch.float32, device=self.device)
        lin = torch.zeros((n,), dtype=torch.float32, device=self.device) if self.lin_w is not None else None
        for r in range(G):
            rows = local_rows(n, r, G)
            buf = (self.table[:rows].contiguous().clone() if r == self.rank
                   else torch.empty((rows, self.D), device=self.device))
            if G > 1:
                self.exchange.broadcast(buf, r)
            full[r::G] = buf
            if lin is not None:
                lb = self.lin_w[:rows].clone() if r == self.rank else torch.empty((rows,), device=self.device)
                if G > 1:
                    self.exchange.broadcast(lb, r)
                lin[r::G] = lb
        return full, lin


def _flat_layout(B, nnz, F, device):
    """(name -> (byte offset, dtype, elements), a new flat buffer) for the per-batch arrays of the local graph."""
    lay, off = {}, 0
    for name, dt, n in (("labels", torch.float32, B), ("weights", torch.float32, max(nnz, 1)),
                        ("row_ptr", torch.int32, B * F + 1), ("tok_row", torch.int32, max(nnz, 1)),
                        ("ids", torch.int64, max(nnz, 1))):
        lay[name] = (off, dt, n)
        off += (n * (8 if dt == torch.int64 else 4) + 255) // 256 * 256
    return lay, torch.zeros((max(off, 256),), dtype=torch.uint8, device=device)


def _flat_view(flat, lay, name):
    off, dt, n = lay[name]
    return flat[off:off + n * (8 if dt == torch.int64 else 4)].view(dt)


class _StepGraph(object):
    """One rank's WHOLE step under fixed-capacity requests as one HIP graph: rows gathered for the peers and
    sent, the local forward / backward / gradient records (_LocalGraph._run), records to their owners, the
    owners' update, the all-reduce and update of the replicated parameters, the loss -- and, on a branch of
    its own, the routing of the batch that FOLLOWS (partition, request to the owners, owner-side sort) into the
    other generation's buffers.  A step then costs the host one copy (the next batch's flat buffer into the
    other generation's inputs), one graph launch and a look at a pinned word: whether the batch routed by the
    previous graph fits its requests (RowShardExchange.fits, tagged answer).

    There is one such graph per routing generation; each reads its own batch's arrays from its _LocalGraph's
    inputs (loaded one step earlier, when the batch was announced as next_batch) and the next batch's ids from
    the other one's.  Every rank captures the same sequence of collectives (RCCL orders them on the process
    group's own stream, which joins the capture); ranks run the same software on the same shapes, so they
    take or leave this path together."""

    def __init__(self, eng, lg, lg_next, plan):
        self.eng, self.lg, self.lg_next, self.plan = eng, lg, lg_next, plan
        self.plan_next = None
        self.graph = None
        self.ws_token = eng._ws_token
        self.tot = torch.zeros((2,), dtype=torch.float32, device=eng.device)
        self.fit_key = (1 - plan.gen, eng.exchange.world)
        eng.exchange.fit_buffers(self.fit_key, eng.device)
        if eng._route_stream is None:
            eng._route_stream = torch.cuda.Stream(device=eng.device)

    def capture(self):
        eng, ex = self.eng, self.eng.exchange
        keep = (eng.step, eng.b1p, eng.b2p, ex._calls, eng._rows_stale, eng._route_pre)
        if ex.collective and ex.watchdog_quiet_s > 0:
            # The process group's watchdog thread polls the events of the eager collectives it still lists; asked
            # about one while the group's stream is being captured, the runtime answers with an error that ends
            # the process.  With the device idle and a few of its polling periods gone by, its list is empty.
            torch.cuda.synchronize(eng.device)
            time.sleep(ex.watchdog_quiet_s)
        graph = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                self._run()
        except Exception as exc:  # noqa: BLE001
            # (a capture that failed leaves its streams capturing: there is no falling back inside this process)
            raise RuntimeError("HIP graph capture of the whole sharded step failed (%s); XDFM_SHARD_STEP_GRAPH=0 "
                               "keeps the exchanges eager" % (exc,))
        finally:
            eng.step, eng.b1p, eng.b2p, ex._calls, eng._rows_stale, eng._route_pre = keep
        self.graph = graph

    def _run(self):
        eng, ex, lg = self.eng, self.eng.exchange, self.lg
        cur = torch.cuda.current_stream(eng.device)
        rs = eng._route_stream
        at = eng.step_route_at

        def route_next():
            with torch.cuda.stream(rs):
                self.plan_next = eng._prepare(self.lg_next.batch)

        def fork_here(where):
            if at == where:
                rs.wait_stream(cur)
                route_next()

        eng.fetch_rows(lg.batch, static=lg, plan=self.plan)
        fork_here(0)  # (behind the rows' exchange: the request's collective is issued after it)
        if lg.fixed:
            lg.send_c.zero_()
        eng.forward(lg.batch, with_loss=True, with_grad=True)
        fork_here(1)
        # as in XDeepFMEngine.train_step: the step scalars move on beside the backward pass, and the weight
        # gradients nothing of the table's path reads are issued behind the gradient records
        if not eng.backward(lg.batch, advance=True, defer_late=True):
            eng._advance_step()
        eng._token_grads(lg.batch, lg.seg_ids, lg.order32, lg.send_c)
        eng.run_late_work()
        fork_here(2)
        eng._owner_update(lg.batch, lg.send_c)
        eng.join_late()
        if ex.collective:
            ex.all_reduce(eng.dense_grad)
        eng._update_dense(True)
        fork_here(3)
        cur.wait_stream(rs)
        if eng.rank != 0:
            eng.reg_loss.zero_()
        if not eng.sparse:
            eng.reg_loss.add_(eng.reg_table)
        _lib.check(eng.lib.xdfm_loss_total(_ptr(eng._loss2), _ptr(self.tot), eng._stream()))

    def step(self, batch, next_batch):
        eng, ex, lg = self.eng, self.eng.exchange, self.lg
        if lg.loaded is not batch:
            lg.load(batch)
        self.lg_next.load(next_batch)
        if self.graph is None:
            self.capture()
        self.graph.replay()
        # what the eager step's calls leave on the host
        eng.step += 1
        if eng.opt_kind == 1 or eng.use_dropout or eng.catchup_on:
            eng.b1p = np.float32(eng.b1p * np.float32(0.9))
            eng.b2p = np.float32(eng.b2p * np.float32(0.999))
        if eng.sparse and eng.catchup_on:
            eng._rows_stale = True
        ex._calls += 1
        ex._fit_expect[self.fit_key] += 1
        eng._route_pre = (next_batch, self.plan_next, None)
        eng._after_step_graph = True
        eng.graph_steps += 1
        batch.route = self.plan
        return self.tot[1], self.tot[0]


class _LocalGraph(object):
    """forward + backward + per-token gradients of one rank for batches of one shape, captured
    as a HIP graph over static buffers (the rows fetched from the owners land in src_table)."""

    def __init__(self, eng, batch, slots=None, fixed=False, plan=None):
        """slots: rows fetched from the owners = gradient records sent back (the batch's token count bounds
        them under sized requests; a fixed-capacity request has world x cap, used or not: the unused
        records must leave as zeros, so send_c is cleared inside the graph)."""
        from recommenders_amd.models.deeprec.models._engine import DeviceBatch

        dev, nnz, B, D = eng.device, batch.nnz, batch.B, eng.D
        use_lin = bool(eng.hp.use_Linear_part)
        f32 = torch.float32
        self.eng = eng
        self.slots = slots = int(slots) if slots is not None else nnz
        self.fixed = bool(fixed)
        self.src_comb = torch.zeros((slots, D + 4 if use_lin else D), dtype=f32, device=dev)
        self.src_table = self.src_comb[:, :D]
        self.src_lin = self.src_comb[:, D] if use_lin else None
        self.direct = plan is not None
        if self.direct:
            # the plan's own (generation) buffers: nothing to copy per step
            self.src_ids, self.order32, self.seg_ids = plan.inverse, plan.order, plan.slot_sorted
        else:
            self.src_ids = torch.zeros((nnz,), dtype=torch.int64, device=dev)
            self.order32 = torch.zeros((nnz,), dtype=torch.int32, device=dev)
            # the segment keys of the per-slot pre-reduction: a buffer of this graph's own (a captured
            # launch keeps the address it was recorded with)
            self.seg_ids = torch.zeros((nnz,), dtype=torch.int32, device=dev)
        self.ws_token = eng._ws_token
        self.gen = plan.gen if plan is not None else None
        self.loaded = None  # the batch whose arrays the inputs hold
        self.send_c = torch.zeros((slots, D + 4 if use_lin else D), dtype=f32, device=dev)
        sb = DeviceBatch()
        sb.B, sb.nnz = B, nnz
        self.lay, self.flat = _flat_layout(B, nnz, eng.F, dev)
        sb.labels, sb.weights, sb.row_ptr, sb.tok_row = (_flat_view(self.flat, self.lay, k)
                                                          for k in ("labels", "weights", "row_ptr", "tok_row"))
        # (not read by the local pass, whose embeddings come from src_table: a step graph routes the batch that
        # follows from here)
        sb.ids = _flat_view(self.flat, self.lay, "ids")[:nnz] if nnz else None
        sb.src_table, sb.src_lin, sb.src_ids, sb.src_N = self.src_table, self.src_lin, self.src_ids, max(slots, 1)
        sb.sorted_ids = sb.sorted_tok = None
        sb.planned = True
        sb.plan_event = None
        self.batch = sb
        self.graph = None

    def capture(self):
        """A warm-up pass off the capture, then the capture (the inputs hold a real batch).  The replay
        that follows does this batch's work once more: everything the pass writes is overwritten by it,
        except the moving statistics of enable_BN, which move with every forward pass -- they are put
        back after the warm-up."""
        dev = self.eng.device
        moving = {k: t.clone() for k, t in self.eng.bn_state.items()}
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            self._run()
        torch.cuda.current_stream(dev).wait_stream(side)
        for k, t in moving.items():
            self.eng.bn_state[k].copy_(t)
        torch.cuda.synchronize(dev)
        graph = torch.cuda.CUDAGraph()
        # thread_local: the process group's watchdog thread may touch the runtime meanwhile
        with torch.cuda.graph(graph, capture_error_mode="thread_local"):
            self._run()
        self.graph = graph

    def _run(self):
        eng = self.eng
        if self.fixed:
            self.send_c.zero_()  # the request's unused slots send zero gradients (to the owners' scratch rows)
        eng.forward(self.batch, with_loss=True, with_grad=True)
        eng.backward(self.batch)
        eng._token_grads(self.batch, self.seg_ids, self.order32, self.send_c)
        eng.join_late()  # every fork is joined before a capture ends

    def load(self, batch):
        """Copy the per-batch arrays (and the partition order) into the graph's inputs."""
        sb, plan = self.batch, batch.route
        self.loaded = batch
        flat = getattr(batch, "flat", None)
        if flat is not None and flat.numel() == self.flat.numel():
            self.flat.copy_(flat)  # labels, weights, CSR offsets, token rows in one copy
        else:
            sb.labels.copy_(batch.labels)
            sb.weights[:batch.weights.numel()].copy_(batch.weights)
            sb.row_ptr.copy_(batch.row_ptr)
            sb.tok_row[:batch.tok_row.numel()].copy_(batch.tok_row)
        if not self.direct:
            self.order32.copy_(plan.order)
            self.seg_ids.copy_(plan.slot_sorted)
            self.src_ids.copy_(plan.inverse)
