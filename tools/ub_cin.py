#!/usr/bin/env python
"""Micro-benchmark of the CIN entry points over a few shapes (GPU box):
    python tools/ub_cin.py
Times xdfm_cin_fwd / xdfm_cin_bwd alone (HIP events, 30 back-to-back calls)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recommenders_amd import _lib  # noqa: E402

lib = _lib.lib
dev = torch.device("cuda:0")


def ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def run(B, F, H, D, L, masked, nh, reps=30, bwd=True):
    g = torch.Generator(device=dev).manual_seed(1)
    x0 = torch.rand(B, F, D, device=dev, generator=g) - 0.5
    xk = x0 if masked else torch.rand(B, H, D, device=dev, generator=g) - 0.5
    w = torch.rand(F * H, L, device=dev, generator=g) - 0.5
    xnext = torch.zeros(B, max(nh, 1), D, device=dev)
    pooled = torch.zeros(B, max(L - nh, 1), device=dev)
    sz = ctypes.c_size_t()
    _lib.check(lib.xdfm_cin_fwd_workspace_floats(B, F, H, D, L, ctypes.byref(sz)))
    fws = sz.value
    _lib.check(lib.xdfm_cin_bwd_workspace_floats(B, F, H, D, L, ctypes.byref(sz)))
    ws = torch.zeros(max(fws, sz.value, 1), device=dev)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    d_xnext = torch.rand(B, max(nh, 1), D, device=dev, generator=g)
    d_pooled = torch.rand(B, max(L - nh, 1), device=dev, generator=g)
    d_x0 = torch.zeros_like(x0)
    d_xk = None if masked else torch.zeros_like(xk)
    d_w = torch.zeros_like(w)

    def fwd():
        _lib.check(lib.xdfm_cin_fwd(ptr(x0), ptr(xk), ptr(w), B, F, H, D, L, nh, masked, ptr(xnext), ptr(pooled),
                                    L - nh, 0, ptr(ws), st))

    def bw():
        _lib.check(lib.xdfm_cin_bwd(ptr(x0), ptr(xk), ptr(w), B, F, H, D, L, nh, masked, ptr(d_xnext), ptr(d_pooled),
                                    L - nh, 0, ptr(d_xk), ptr(d_x0), ptr(d_w), ptr(ws), st))

    out = []
    for fn in (fwd, bw) if bwd else (fwd,):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / reps * 1e3)
    return out


if __name__ == "__main__":
    print("B F H D L masked nh -> fwd_us bwd_us")
    shapes = [(8192, 39, 39, 16, L, 1, L // 2) for L in (4, 8, 20, 40)]
    shapes += [(8192, 39, 10, 16, L, 0, 0) for L in (4, 10, 20)]
    shapes += [(B, 39, 39, 16, 20, 1, 10) for B in (2048, 4096, 16384, 32768)]
    shapes += [(8192, 39, 39, 16, 20, 0, 10), (8192, 8, 8, 16, 20, 1, 10), (8192, 20, 20, 16, 20, 1, 10)]
    for sh in shapes:
        t = run(*sh, bwd="--fwd" not in sys.argv)
        print(*sh, "->", " ".join("%.1f" % v for v in t), flush=True)
