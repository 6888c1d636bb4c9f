"""ctypes binding of libxdfm.so (the C ABI declared in include/xdfm.h).

The library is built in-tree by ``__graft_entry__.build()`` /
``make -C recommenders_amd/csrc``.  There is no fallback: if the shared object is
missing or a symbol cannot be bound, importing a caller fails with
``XdfmLibraryError``.
"""

import ctypes
import os

# PyTorch-ROCm ships its own libamdhip64.so.7.  libxdfm.so must bind to THAT runtime instance
# (same device contexts, streams and allocations as torch), not to a second copy from
# /opt/rocm: importing torch first puts its runtime in the process, and the dynamic loader
# then satisfies libxdfm.so's NEEDED libamdhip64.so.7 by soname from the already loaded one.
import torch  # noqa: F401  (must precede the CDLL below)

__all__ = ["lib", "check", "XdfmLibraryError", "XdfmError", "OptArgs", "RowAge", "BatchInfo", "LIB_PATH"]

# XDFM_LIB_PATH: another build of the same library (kernel tuning builds beside the shipped one)
LIB_PATH = os.environ.get("XDFM_LIB_PATH") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libxdfm.so")


class XdfmLibraryError(ImportError):
    pass


class XdfmError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("libxdfm error {0}: {1}".format(code, message))
        self.code = code
        self.message = message


class OptArgs(ctypes.Structure):
    """struct xdfm_opt_args"""

    _fields_ = [
        ("kind", ctypes.c_int32),
        ("lr", ctypes.c_float),
        ("beta1", ctypes.c_float),
        ("beta2", ctypes.c_float),
        ("eps", ctypes.c_float),
        ("beta1_power", ctypes.c_float),
        ("beta2_power", ctypes.c_float),
        ("l2", ctypes.c_float),
        ("l1", ctypes.c_float),
        ("clip_norm", ctypes.c_float),
        ("reserved", ctypes.c_int32),
        ("opt_state", ctypes.c_void_p),
    ]


class RowAge(ctypes.Structure):
    """struct xdfm_row_age"""

    _fields_ = [
        ("row_step", ctypes.c_void_p),
        ("step_stride", ctypes.c_int64),
        ("slot_m", ctypes.c_void_p),
        ("slot_v", ctypes.c_void_p),
        ("lin_m", ctypes.c_void_p),
        ("lin_v", ctypes.c_void_p),
        ("drift_tab", ctypes.c_void_p),
        ("opt", ctypes.POINTER(OptArgs)),
        ("step_offset", ctypes.c_int32),
        ("reserved", ctypes.c_int32),
    ]


class DwJob(ctypes.Structure):
    """struct xdfm_dw_job"""

    _fields_ = [
        ("x", ctypes.c_void_p),
        ("y", ctypes.c_void_p),
        ("dy", ctypes.c_void_p),
        ("dw", ctypes.c_void_p),
        ("db", ctypes.c_void_p),
        ("K", ctypes.c_int32),
        ("Nout", ctypes.c_int32),
        ("act", ctypes.c_int32),
        ("reserved", ctypes.c_int32),
    ]


class TowerDesc(ctypes.Structure):
    """struct xdfm_tower_desc"""

    _fields_ = [
        ("n_hidden", ctypes.c_int32),
        ("in_dim", ctypes.c_int32),
        ("size", ctypes.c_int32 * 4),
        ("act", ctypes.c_int32 * 4),
        ("w", ctypes.c_void_p * 4),
        ("b", ctypes.c_void_p * 4),
        ("w_out", ctypes.c_void_p),
        ("b_out", ctypes.c_void_p),
        ("h", ctypes.c_void_p * 4),
        ("d_h", ctypes.c_void_p * 4),
    ]


class CinPackJob(ctypes.Structure):
    """struct xdfm_cin_pack_job"""

    _fields_ = [
        ("w", ctypes.c_void_p),
        ("F", ctypes.c_int32),
        ("H", ctypes.c_int32),
        ("L", ctypes.c_int32),
        ("masked", ctypes.c_int32),
        ("fwd_image", ctypes.c_void_p),
        ("bwd_image", ctypes.c_void_p),
        ("ldw", ctypes.c_int32),
        ("reserved", ctypes.c_int32),
    ]


class BatchInfo(ctypes.Structure):
    """struct xdfm_ffm_batch_info"""

    _fields_ = [
        ("n_rows", ctypes.c_int64),
        ("nnz", ctypes.c_int64),
        ("dnn_width", ctypes.c_int64),
        ("imp_bytes", ctypes.c_int64),
        ("first_line", ctypes.c_int64),
    ]


_P = ctypes.c_void_p
_I32 = ctypes.c_int32
_I64 = ctypes.c_int64
_SZ = ctypes.c_size_t
_PSZ = ctypes.POINTER(ctypes.c_size_t)

# name -> argument types, in the order of include/xdfm.h.  tests/test_abi.py checks this
# table against the header text, so a declaration cannot drift from its binding.
SIGNATURES = {
    "xdfm_abi_version": [],
    "xdfm_prof_enable": [_I32],
    "xdfm_prof_collect": [ctypes.c_char_p, _SZ],
    "xdfm_debug_stamps": [_P],
    "xdfm_ffm_open": [ctypes.c_char_p, _I64, _I64, ctypes.c_char, ctypes.c_char, ctypes.POINTER(_P)],
    "xdfm_ffm_next": [_P, _I64, ctypes.POINTER(BatchInfo)],
    "xdfm_ffm_fill": [_P] * 11,
    "xdfm_ffm_close": [_P],
    "xdfm_build_row_ptr": [_P, _I64, _I64, _P, _P, _P, _P],
    "xdfm_embed_fwd": [_P, _P, _I64, _I64, _I64, _I32, _I32, _I64, _P, _P, _P, _I64, _P, _P, _P, _P, _P,
                       ctypes.POINTER(RowAge), _P],
    "xdfm_embed_gather": [_P, _P, _I64, _I64, _I64, _I32, _P, _I64, _P, _P, _I64, _P, ctypes.POINTER(RowAge), _P],
    "xdfm_route_keys": [_P, _I64, _I64, _I32, _P, _P],
    "xdfm_route_pad": [_P, _P, _P, _P, _I64, _I32, _I64, _I64, _P, _P, _P, _P, _P],
    "xdfm_route_decode": [_P, _I32, _I64, _P, _P, _P],
    "xdfm_route_scratch_ints": [_I64, _PSZ],
    "xdfm_route_uniq": [_P, _P, _I64, _I64, _I32, _P, _P, _P, _P, _P, _P],
    "xdfm_embed_plan_workspace_bytes": [_I64, _I64, _PSZ],
    "xdfm_embed_plan": [_P, _I64, _I64, _P, _P, _P, _SZ, _P],
    "xdfm_embed_plan_static": [_P, _I64, _I64, _P, _P, _P],
    "xdfm_embed_plan_gather": [_P, _P, _P, _I64, _P, _P, _P],
    "xdfm_embed_bwd": [_I32, _P, _P, _P, _P, _P, _P, _I64, _I64, _P, _P, _I64, _I32, _I32, _I64, _P, _P, _P,
                       _P, _I64, _P, _P, _P, _I32, ctypes.POINTER(OptArgs), _P, _P, _I64, ctypes.POINTER(RowAge),
                       _P],
    "xdfm_embed_catchup": [_P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _I32, _P, _I64, ctypes.POINTER(RowAge), _P],
    "xdfm_embed_clear_rows": [_P, _P, _I32, _P, _I64, _P],
    "xdfm_opt_advance": [_P, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float, _P, _P],
    "xdfm_drift_tab_floats": [_PSZ],
    "xdfm_drift_tab_init": [_P, _P, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float, _P],
    "xdfm_dense_update": [_P, _P, _P, _P, _I32, _P, _P, _P, _P, ctypes.POINTER(OptArgs), _P, _P,
                          _I32, _P],
    "xdfm_dense_reg_loss": [_P, _I32, _P, _P, _P, _P, _P, _P, _I32, _P],
    "xdfm_cin_fwd_workspace_floats": [_I64, _I32, _I32, _I32, _I32, _PSZ],
    "xdfm_cin_fwd": [_P, _P, _P, _I64, _I32, _I32, _I32, _I32, _I32, _I32, _P, _P, _I32, _I32, _P, _P],
    "xdfm_cin_bwd_workspace_floats": [_I64, _I32, _I32, _I32, _I32, _PSZ],
    "xdfm_cin_bwd": [_P, _P, _P, _I64, _I32, _I32, _I32, _I32, _I32, _I32, _P, _P, _I32, _I32, _P,
                     _P, _P, _P, _P],
    "xdfm_cin_pack_floats": [_I64, _I32, _I32, _I32, _I32, _I32, _PSZ, _PSZ],
    "xdfm_cin_pack": [_P, _I32, _I64, _I32, _P],
    "xdfm_cin_fwd_packed": [_P, _P, _P, _I64, _I32, _I32, _I32, _I32, _I32, _I32, _P, _P, _I32, _I32, _P],
    "xdfm_cin_bwd_packed": [_P, _P, _P, _I64, _I32, _I32, _I32, _I32, _I32, _I32, _P, _P, _I32, _I32, _P,
                            _P, _P, _P, _I32, _P, _P, _I32, _P],
    "xdfm_cin_pair_count": [_I32, _I32, _I32, ctypes.POINTER(ctypes.c_int64)],
    "xdfm_cin_pair_sum": [_P, _P, _I64, _I32, _I32, _I32, _I32, _P, _P],
    "xdfm_cin_pair_bwd": [_P, _P, _P, _I64, _I32, _I32, _I32, _I32, _P, _I32, _P, _I32, _P],
    "xdfm_cin_pair_sum_bf16": [_P, _P, _I64, _I32, _I32, _I32, _I32, _P, _P],
    "xdfm_cin_pair_bwd_bf16": [_P, _P, _P, _I64, _I32, _I32, _I32, _I32, _P, _I32, _P, _I32, _P],
    "xdfm_cin_bf16_image_bytes": [_I32, _I32, _I32, _PSZ, _PSZ],
    "xdfm_cin_bf16_pack": [_P, _I32, _I32, _I32, _I32, _I32, _P, _P, _P],
    "xdfm_cin_bf16_fwd": [_P, _P, _P, _I64, _I32, _I32, _I32, _I32, _I32, _P, _P],
    "xdfm_cin_bf16_bwd_dx": [_P, _P, _P, _P, _I64, _I32, _I32, _I32, _I32, _I32, _P, _P, _I32, _P],
    "xdfm_cin_bf16_bwd_dw_workspace_floats": [_I64, _I32, _I32, _I32, _I32, _PSZ],
    "xdfm_cin_bf16_bwd_dw": [_P, _P, _P, _I64, _I32, _I32, _I32, _I32, _I32, _P, _I32, _P, _P],
    "xdfm_cin_pooled_workspace_floats": [_I64, _I32, _I32, _I32, _I32, _PSZ],
    "xdfm_cin_pooled_fwd": [_P, _P, _P, _I64, _I32, _I32, _I32, _I32, _I32, _P, _I32, _I32, _P],
    "xdfm_cin_pooled_fwd_head": [_P, _P, _P, _I64, _I32, _I32, _I32, _I32, _I32, _P, _I32, _P, _P, _P, _P, _P, _P,
                                 _I32, _P, _I64, _I32, _I32, _P, _P, _P, _P, _P, _P, _P, _P],
    "xdfm_loss_finish_partials": [_P, _I32, _I64, _P, _P],
    "xdfm_cin_pooled_bwd": [_P, _P, _P, _P, _I32, _I32, _I64, _I32, _I32, _I32, _I32, _I32, _P, _P, _I32, _P, _P, _P],
    "xdfm_cin_pool_pack": [_P, _I32, _I32, _I32, _I32, _I32, _P, _P],
    "xdfm_cin_pool_unpack": [_P, _I32, _I32, _I32, _I32, _I32, _P, _P],
    "xdfm_bn_fwd": [_P, _I64, _I32, _I32, _P, _P, _P, _P, ctypes.c_float, ctypes.c_float, _I32, _I32, _P, _P, _P],
    "xdfm_softmax_fwd": [_P, _I64, _I32, _I32, _P, _P],
    "xdfm_softmax_bwd": [_P, _P, _I64, _I32, _I32, _P, _P],
    "xdfm_bn_bwd": [_P, _P, _P, _I64, _I32, _I32, _P, _P, _I32, _P, _P, _P, _P],
    "xdfm_bn_local_moments": [_P, _I64, _I32, _I32, _P, _P],
    "xdfm_bn_fwd_moments": [_P, _I64, _I32, _I32, _P, _P, _P, _P, ctypes.c_float, ctypes.c_float, _P, _I32, _P, _P, _P],
    "xdfm_bn_bwd_sums": [_P, _P, _P, _I64, _I32, _I32, _P, _I32, _P, _P, _P],
    "xdfm_bn_bwd_apply": [_P, _P, _P, _I64, _I32, _I32, _I64, _P, _P, _I32, _P, _P, _P, _P],
    "xdfm_dense_fwd": [_P, _P, _P, _I64, _I32, _I32, _I32, _P, _P],
    "xdfm_dense_bf16_workspace_floats": [_I64, _I32, _I32, _PSZ],
    "xdfm_dense_bf16_fwd": [_P, _P, _P, _I64, _I32, _I32, _I32, _P, _P, _P],
    "xdfm_dense_bf16_bwd": [_P, _P, _P, _P, _I64, _I32, _I32, _I32, _P, _I32, _P, _P, _P, _P],
    "xdfm_dense_bwd_workspace_floats": [_I64, _I32, _I32, _PSZ],
    "xdfm_dense_bwd": [_P, _P, _P, _P, _I64, _I32, _I32, _I32, _P, _I32, _P, _P, _P, _P],
    "xdfm_dense_bwd_dw_multi_workspace_floats": [_P, _I32, _I64, _PSZ],
    "xdfm_dense_bwd_dw_multi": [_P, _I32, _I64, _P, _P],
    "xdfm_tower_fwd": [_P, _P, _I64, _P, _P],
    "xdfm_tower_bwd_dx": [_P, _P, _I64, _P, _I32, _P],
    "xdfm_cin_act_fwd": [_P, _I64, _I32, _I32, _I32, _I32, _P, _P, _I32, _I32, _P],
    "xdfm_cin_act_bwd": [_P, _P, _P, _I32, _I32, _I64, _I32, _I32, _I32, _I32, _P, _P],
    "xdfm_fast_cin_fwd": [_P, _P, _P, _P, _I64, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _P, _P, _I32, _I32, _P],
    "xdfm_fast_cin_bwd_workspace_floats": [_I64, _I32, _I32, _I32, _I32, _I32, _PSZ],
    "xdfm_fast_cin_bwd": [_P, _P, _P, _P, _I64, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _P, _P, _I32, _I32,
                          _P, _P, _P, _P, _P, _P],
    "xdfm_dropout_act_fwd": [_P, _P, _I64, _I32, ctypes.c_float, _I32, ctypes.c_uint64, _P, _P],
    "xdfm_dropout_act_bwd": [_P, _P, _P, _I64, _I32, _I32, _P, _P],
    "xdfm_cin_head_fwd": [_P, _P, _P, _I64, _I32, _P, _P],
    "xdfm_cin_head_bwd": [_P, _P, _P, _I64, _I32, _P, _P, _P, _P, _P],
    "xdfm_logit_loss": [_P, _P, _P, _P, _P, _P, _I64, _I64, _I32, _I32, _P, _P, _P, _P, _P, _P],
    "xdfm_loss_finish": [_P, _I64, _I64, _I32, _P, _P],
    "xdfm_loss_total": [_P, _P, _P],
    "xdfm_head_loss": [_P, _P, _P, _P, _P, _P, _P, _I32, _P, _I64, _I64, _I32, _I32, _P, _P, _P, _P, _P, _P, _P,
                       _P],
    "xdfm_eval_workspace_bytes": [_I64, _PSZ],
    "xdfm_eval_auc_logloss": [_P, _P, _I64, _P, _P, _SZ, _P],
}


def _load():
    if not os.path.exists(LIB_PATH):
        raise XdfmLibraryError(
            "{0} not found: build it first (python -c 'import __graft_entry__ as g; g.build()' "
            "or make -C recommenders_amd/csrc). There is no CPU fallback.".format(LIB_PATH)
        )
    try:
        handle = ctypes.CDLL(LIB_PATH)
    except OSError as exc:  # pragma: no cover - depends on the box
        raise XdfmLibraryError("cannot load {0}: {1}".format(LIB_PATH, exc))
    for name, argtypes in SIGNATURES.items():
        try:
            fn = getattr(handle, name)
        except AttributeError:
            raise XdfmLibraryError("{0} does not export {1}".format(LIB_PATH, name))
        fn.argtypes = argtypes
        fn.restype = ctypes.c_int
    handle.xdfm_last_error.argtypes = []
    handle.xdfm_last_error.restype = ctypes.c_char_p
    return handle


lib = _load()


def check(code):
    """Raise XdfmError for a negative status."""
    if code < 0:
        msg = lib.xdfm_last_error()
        raise XdfmError(code, msg.decode("utf-8", "replace") if msg else "")
    return code


_hip_runtime = None


def hip_runtime():
    """The HIP runtime this process has loaded already (torch's; libxdfm.so resolves to the same one), for the few
    runtime calls torch does not expose: a copy node inside a capture, its node handle, its source per replay.
    Opened by the path /proc/self/maps shows, so that no second copy of the runtime is ever loaded."""
    global _hip_runtime
    if _hip_runtime is None:
        path = "libamdhip64.so"
        try:
            with open("/proc/self/maps") as f:
                for line in f:
                    if "libamdhip64" in line:
                        path = line.split()[-1]
                        break
        except OSError:
            pass
        rt = ctypes.CDLL(path)
        vp, sz = ctypes.c_void_p, ctypes.c_size_t
        rt.hipMemcpyAsync.argtypes = [vp, vp, sz, ctypes.c_int, vp]
        rt.hipMemcpyAsync.restype = ctypes.c_int
        rt.hipStreamGetCaptureInfo_v2.argtypes = [vp, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_ulonglong),
                                                  ctypes.POINTER(vp), ctypes.POINTER(ctypes.POINTER(vp)),
                                                  ctypes.POINTER(sz)]
        rt.hipStreamGetCaptureInfo_v2.restype = ctypes.c_int
        rt.hipGraphExecMemcpyNodeSetParams1D.argtypes = [vp, vp, vp, vp, sz, ctypes.c_int]
        rt.hipGraphExecMemcpyNodeSetParams1D.restype = ctypes.c_int
        _hip_runtime = rt
    return _hip_runtime


HIP_MEMCPY_DEVICE_TO_HOST = 2
HIP_MEMCPY_DEVICE_TO_DEVICE = 3
