#!/usr/bin/env python
"""Micro-benchmark of xdfm_embed_plan (stable LSD radix sort of one batch's tokens by id) on
Criteo-shaped ids: 8192 x 39 tokens, 2^25 rows.  Prints us per sort (HIP events, 200 sorts)."""
import ctypes
import sys, os

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recommenders_amd import _lib  # noqa: E402
from recommenders_amd.synth import criteo_shape_ids  # noqa: E402


def main():
    N, B = 1 << 25, int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    ids = torch.from_numpy(criteo_shape_ids(np.random.default_rng(5), B, N).astype(np.int64).reshape(-1)).cuda()
    n = ids.numel()
    sz = ctypes.c_size_t()
    _lib.check(_lib.lib.xdfm_embed_plan_workspace_bytes(n, N, ctypes.byref(sz)))
    ws = torch.empty((sz.value,), dtype=torch.uint8, device="cuda")
    sk = torch.empty((n,), dtype=torch.int32, device="cuda")
    st = torch.empty((n,), dtype=torch.int32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream

    def run():
        _lib.check(_lib.lib.xdfm_embed_plan(ids.data_ptr(), n, N, sk.data_ptr(), st.data_ptr(), ws.data_ptr(),
                                            ws.numel(), stream))
    for _ in range(20):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    iters = 200
    e0.record()
    for _ in range(iters):
        run()
    e1.record()
    torch.cuda.synchronize()
    print("xdfm_embed_plan: %d tokens, %.2f us per sort (eager launches, back to back)" % (n, e0.elapsed_time(e1) / iters * 1e3))
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        stream = s.cuda_stream
        run()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for _ in range(10):
                run()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(20):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    print("xdfm_embed_plan: %.2f us per sort inside a HIP graph (10 sorts per replay)" % (e0.elapsed_time(e1) / 200 * 1e3))
    order = np.argsort(ids.cpu().numpy(), kind="stable")
    assert np.array_equal(st.cpu().numpy(), order.astype(np.int32))


if __name__ == "__main__":
    main()
