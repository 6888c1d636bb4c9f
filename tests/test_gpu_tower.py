"""The one-launch-per-pass tower (xdfm_tower_fwd / xdfm_tower_bwd_dx) and the multi-job weight gradients
(xdfm_dense_bwd_dw_multi) through the C ABI against a float64 restatement of xDeepFM.py:473-533 (xw_plus_b +
activation per layer, identity output), and against the per-layer entry points they replace."""
import ctypes

import numpy as np
import pytest
import torch

from recommenders_amd import _lib
from recommenders_amd.models.deeprec.models._engine import _ptr

pytestmark = pytest.mark.gpu
RTOL = 1e-5  # BASELINE.json north_star: fp32 within 1e-5 relative
ACTS = {0: lambda z: z, 1: torch.relu, 2: torch.sigmoid, 3: torch.tanh, 4: torch.nn.functional.elu}


def close(got, exp, what):
    exp = exp.detach().cpu().double()
    got = got.detach().cpu().double()
    scale = float(exp.abs().max()) or 1.0
    err = float((got - exp).abs().max()) / scale
    assert err <= RTOL, "%s: %.3g of the tensor's scale" % (what, err)


def make(B, K, sizes, acts, seed):
    g = torch.Generator().manual_seed(seed)
    dev = torch.device("cuda:0")
    x = torch.randn(B, K, generator=g).to(dev)
    ws, bs, last = [], [], K
    for n in sizes:
        ws.append((torch.randn(last, n, generator=g) / np.sqrt(last)).to(dev))
        bs.append((0.1 * torch.randn(n, generator=g)).to(dev))
        last = n
    w_out = (torch.randn(last, 1, generator=g) / np.sqrt(last)).to(dev)
    b_out = (0.1 * torch.randn(1, generator=g)).to(dev)
    up = torch.randn(B, generator=g).to(dev) / B
    return x, ws, bs, w_out, b_out, up


def reference(x, ws, bs, w_out, b_out, acts, up):
    x64 = x.double().requires_grad_(True)
    p = [(w.double().requires_grad_(True), b.double().requires_grad_(True)) for w, b in zip(ws, bs)]
    wo, bo = w_out.double().requires_grad_(True), b_out.double().requires_grad_(True)
    hs, h = [], x64
    for (w, b), a in zip(p, acts):
        h = ACTS[a](h @ w + b)
        h.retain_grad()
        hs.append(h)
    out = (h @ wo + bo)[:, 0]
    (out * up.double()).sum().backward()
    return hs, out, x64.grad, [(w.grad, b.grad) for w, b in p], (wo.grad, bo.grad)


@pytest.mark.parametrize("B,K,sizes,acts", [
    (8192, 624, [20, 20], [1, 1]),          # the bench workload's tower
    (100, 390, [32, 64, 7], [3, 2, 4]),     # widest first layer, three layers, rows that do not fill a block
    (37, 13, [5], [2]),                     # one hidden layer, K not a multiple of 4
    (300, 40, [16, 16, 16, 16], [1, 0, 1, 3]),
])
def test_tower_passes_match_float64_and_the_per_layer_calls(B, K, sizes, acts):
    lib = _lib.lib
    dev = torch.device("cuda:0")
    x, ws, bs, w_out, b_out, up = make(B, K, sizes, acts, seed=B + K)
    hs64, out64, dx64, dws64, dout64 = reference(x, ws, bs, w_out, b_out, acts, up)

    n = len(sizes)
    h = [torch.empty(B, s, device=dev) for s in sizes]
    d_h = [torch.full((B, s), float("nan"), device=dev) for s in sizes]
    out = torch.empty(B, device=dev)
    d = _lib.TowerDesc()
    d.n_hidden, d.in_dim = n, K
    for k in range(n):
        d.size[k], d.act[k] = sizes[k], acts[k]
        d.w[k], d.b[k], d.h[k], d.d_h[k] = ws[k].data_ptr(), bs[k].data_ptr(), h[k].data_ptr(), d_h[k].data_ptr()
    d.w_out, d.b_out = w_out.data_ptr(), b_out.data_ptr()
    _lib.check(lib.xdfm_tower_fwd(ctypes.byref(d), _ptr(x), B, _ptr(out), None))
    for k in range(n):
        close(h[k], hs64[k], "h[%d]" % k)
    close(out, out64, "out")

    # the per-layer entry points give the same activations (another association of the same sums)
    y, last = x, K
    for k in range(n):
        yk = torch.empty(B, sizes[k], device=dev)
        _lib.check(lib.xdfm_dense_fwd(_ptr(y), _ptr(ws[k]), _ptr(bs[k]), B, last, sizes[k], acts[k], _ptr(yk), None))
        close(h[k], yk, "h[%d] against xdfm_dense_fwd" % k)
        y, last = yk, sizes[k]

    for accumulate in (0, 1):
        base = torch.randn(B, K, device=dev) / B if accumulate else torch.full((B, K), float("nan"), device=dev)
        dx = base.clone()
        _lib.check(lib.xdfm_tower_bwd_dx(ctypes.byref(d), _ptr(up), B, _ptr(dx), accumulate, None))
        close(dx, dx64 + base.double() if accumulate else dx64, "dx (accumulate=%d)" % accumulate)
    for k in range(n):
        close(d_h[k], hs64[k].grad, "d_h[%d]" % k)

    # weight gradients of every layer in one launch, with two more jobs of other shapes beside them
    extra_x = torch.randn(B, 9, device=dev)
    jobs = (_lib.DwJob * (n + 3))()
    acts_in = [x] + h
    dw = [torch.empty_like(w) for w in ws] + [torch.empty_like(w_out), torch.empty(1, 1, device=dev),
                                               torch.empty(9, 1, device=dev)]
    db = [torch.empty_like(b) for b in bs] + [torch.empty(1, device=dev) for _ in range(3)]
    for k in range(n):
        J = jobs[k]
        J.x, J.y, J.dy, J.dw, J.db = (acts_in[k].data_ptr(), h[k].data_ptr(), d_h[k].data_ptr(), dw[k].data_ptr(),
                                      db[k].data_ptr())
        J.K, J.Nout, J.act = acts_in[k].shape[1], sizes[k], acts[k]
    J = jobs[n]
    J.x, J.y, J.dy, J.dw, J.db = h[-1].data_ptr(), out.data_ptr(), up.data_ptr(), dw[n].data_ptr(), db[n].data_ptr()
    J.K, J.Nout, J.act = sizes[-1], 1, 0
    J = jobs[n + 1]  # sum_b up: the bias row of a [1] -> [1] identity layer
    J.x = J.y = J.dy = up.data_ptr()
    J.dw, J.db, J.K, J.Nout, J.act = dw[n + 1].data_ptr(), db[n + 1].data_ptr(), 1, 1, 0
    J = jobs[n + 2]
    J.x, J.y, J.dy, J.dw, J.db = extra_x.data_ptr(), up.data_ptr(), up.data_ptr(), dw[n + 2].data_ptr(), db[n + 2].data_ptr()
    J.K, J.Nout, J.act = 9, 1, 0
    sz = ctypes.c_size_t()
    _lib.check(lib.xdfm_dense_bwd_dw_multi_workspace_floats(jobs, n + 3, B, ctypes.byref(sz)))
    part = torch.empty(max(sz.value, 1), device=dev)
    _lib.check(lib.xdfm_dense_bwd_dw_multi(jobs, n + 3, B, _ptr(part), None))
    torch.cuda.synchronize()
    for k in range(n):
        close(dw[k], dws64[k][0], "dw[%d]" % k)
        close(db[k], dws64[k][1], "db[%d]" % k)
    close(dw[n], dout64[0], "dw_out")
    close(db[n], dout64[1], "db_out")
    close(db[n + 1], up.double().sum().reshape(1), "sum of the upstream gradient")
    close(dw[n + 2], (extra_x.double().T @ up.double()).reshape(9, 1), "extra job dw")

    # and bit for bit what xdfm_dense_bwd gives for a layer when the row splits agree (single job)
    one = (_lib.DwJob * 1)()
    ctypes.memmove(one, ctypes.byref(jobs[0]), ctypes.sizeof(_lib.DwJob))
    dw1, db1 = torch.empty_like(ws[0]), torch.empty_like(bs[0])
    one[0].dw, one[0].db = dw1.data_ptr(), db1.data_ptr()
    _lib.check(lib.xdfm_dense_bwd_dw_multi_workspace_floats(one, 1, B, ctypes.byref(sz)))
    part1 = torch.empty(max(sz.value, 1), device=dev)
    _lib.check(lib.xdfm_dense_bwd_dw_multi(one, 1, B, _ptr(part1), None))
    _lib.check(lib.xdfm_dense_bwd_workspace_floats(B, K, sizes[0], ctypes.byref(sz)))
    part2 = torch.empty(max(sz.value, 1), device=dev)
    dw2, db2 = torch.empty_like(ws[0]), torch.empty_like(bs[0])
    _lib.check(lib.xdfm_dense_bwd(_ptr(x), _ptr(ws[0]), _ptr(h[0]), _ptr(d_h[0]), B, K, sizes[0], acts[0], None, 0,
                                  _ptr(dw2), _ptr(db2), _ptr(part2), None))
    torch.cuda.synchronize()
    assert torch.equal(dw1, dw2) and torch.equal(db1, db2)


def test_tower_refuses_what_it_does_not_serve():
    lib = _lib.lib
    d = _lib.TowerDesc()
    d.n_hidden, d.in_dim = 1, 8
    d.size[0] = 33
    assert lib.xdfm_tower_fwd(ctypes.byref(d), None, 4, None, None) != 0
    assert b"32" in lib.xdfm_last_error()
    d.n_hidden = 5
    assert lib.xdfm_tower_fwd(ctypes.byref(d), None, 4, None, None) != 0


def test_device_time_stamps_of_a_training_step():
    """xdfm_debug_stamps: with a buffer installed every kernel family of a step leaves start <= end stamps of the
    100 MHz clock, the gather starts before the table gradient ends, and nothing is written once the buffer is
    taken away again (measurement hook of tools/stamp_timeline.py)."""
    from recommenders_amd.models.deeprec.models._engine import XDeepFMEngine
    from tests.util import make_hp, random_batch

    lib = _lib.lib
    dev = torch.device("cuda:0")
    hp = make_hp(FEATURE_COUNT=5000, FIELD_COUNT=8, dim=8, batch_size=256, layer_sizes=[16, 16],
                 activation=["relu", "relu"], dropout=[0.0, 0.0], cross_layer_sizes=[8, 4])
    eng = XDeepFMEngine(hp, device=dev, seed=3, embedding_update="sparse_rows")
    b = random_batch(np.random.default_rng(0), hp, 256)
    db = eng.upload(b["labels"], b["dnn_feat_indices"], b["dnn_feat_values"], b["dnn_feat_weights"])
    eng.train_step(db)
    torch.cuda.synchronize()
    init = torch.empty(64, dtype=torch.int64)
    init[0::2] = np.iinfo(np.int64).max
    init[1::2] = 0
    buf = init.to(dev)
    _lib.check(lib.xdfm_debug_stamps(ctypes.c_void_p(buf.data_ptr())))
    try:
        db.planned = db.planned_gather = False
        eng.train_step(db)
        torch.cuda.synchronize()
    finally:
        _lib.check(lib.xdfm_debug_stamps(None))
    t = buf.cpu().numpy().reshape(32, 2)
    seen = [i for i in range(32) if t[i, 1] > 0]
    assert 0 in seen and 6 in seen and 8 in seen  # gather, table gradient, dense update (csrc/stamp.h)
    for i in seen:
        assert t[i, 0] <= t[i, 1] and t[i, 1] - t[i, 0] < 100_000_000  # under a second of the 100 MHz clock
    assert t[0, 0] < t[6, 1] <= t[8, 1]
    before = buf.clone()
    db.planned = db.planned_gather = False
    eng.train_step(db)
    torch.cuda.synchronize()
    assert torch.equal(buf, before)
