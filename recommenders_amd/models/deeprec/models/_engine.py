"""Device engine of the xDeepFM path: owns the parameters / optimizer slots in HBM and
drives the HIP kernels of libxdfm.so through the C ABI (include/xdfm.h).

This replaces what `sess.run` does for the reference (base_model.py:379-388, :402, :416):
one forward (+ backward + optimizer) pass of the graph XDeepFMModel._build_graph builds
(xDeepFM.py:24-71).  PyTorch is used for device memory, streams and H2D copies only; all
arithmetic on the path is in the library.  There is no CPU fallback: constructing an
engine without a visible GPU raises.
"""

import ctypes
import os
import warnings

import numpy as np
import torch

from recommenders_amd import _lib

_ACT = {"identity": 0, "relu": 1, "sigmoid": 2, "tanh": 3, "elu": 4, "softmax": 5}
_SOFTMAX = 5  # engine-level code: the kernels get the identity and xdfm_softmax_fwd/bwd follow
_LOSS = {"log_loss": 0, "cross_entropy_loss": 1, "square_loss": 2}
# base_model.py:283-302.  ProximalAdagrad / ProximalGradientDescent are built with l1 = l2 = 0
# there, which makes their updates Adagrad's / plain gradient descent's.
_OPT = {"sgd": 0, "gd": 0, "pgd": 0, "adam": 1, "adagrad": 2, "padagrad": 2, "rmsprop": 3, "adadelta": 4,
        "ftrl": 5}
# (beta1, beta2, eps) as TF1 builds each optimizer from `lr` alone
_OPT_CONST = {0: (0.0, 0.0, 0.0), 1: (0.9, 0.999, 1e-8), 2: (0.0, 0.0, 0.0), 3: (0.0, 0.9, 1e-10),
              4: (0.0, 0.95, 1e-8), 5: (0.0, 0.0, 0.0)}
# initial value of the first slot (Adagrad / FTRL accumulators 0.1, RMSProp's rms 1.0)
_SLOT_M_INIT = {1: 0.0, 2: 0.1, 3: 1.0, 4: 0.0, 5: 0.1}

_DEVERR = {
    1: "a feature index is outside [1, FEATURE_COUNT]",
    2: "a field index is outside [1, FIELD_COUNT]",
    4: "dnn_feat_indices rows are not sorted",
}


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def cin_shapes(hp):
    """(H_k, L_k, n_hidden_k) per CIN layer (xDeepFM.py:191-267; field_nums bookkeeping :262)."""
    out = []
    prev = hp.FIELD_COUNT
    n_layers = len(hp.cross_layer_sizes)
    for idx, layer_size in enumerate(hp.cross_layer_sizes):
        n_hidden = int(layer_size / 2) if idx != n_layers - 1 else 0
        out.append((prev, int(layer_size), n_hidden))
        prev = int(layer_size / 2)
    return out


BN_MOMENTUM, BN_EPS = 0.95, 0.0001  # xDeepFM.py:237-238, :397-398, :500-501


def bn_state_shapes(hp):
    """Moving mean / variance of every batch-norm layer (not trainable), named like the oracle."""
    shapes = {}
    if hp.enable_BN is True:
        if hp.use_CIN_part:
            for k, (_, L, _) in enumerate(cin_shapes(hp)):
                shapes["bnstat_cin_mm_%d" % k] = (L,)
                shapes["bnstat_cin_mv_%d" % k] = (L,)
        if hp.use_DNN_part:
            for k, size in enumerate(hp.layer_sizes):
                shapes["bnstat_dnn_mm_%d" % k] = (int(size),)
                shapes["bnstat_dnn_mv_%d" % k] = (int(size),)
    return shapes


def dense_param_shapes(hp):
    """Ordered name -> shape of every dense (non table-sized) variable of the graph."""
    shapes = {}
    if hp.use_Linear_part:
        shapes["lin_b"] = (1,)
    if hp.use_CIN_part:
        for k, (H, L, _) in enumerate(cin_shapes(hp)):
            if hp.fast_CIN_d > 0:  # _build_fast_CIN filters (xDeepFM.py:330-334, :353-362)
                shapes["fcin_w_%d" % k] = (hp.FIELD_COUNT, int(hp.fast_CIN_d) * L)
                if k > 0:
                    shapes["fcin_v_%d" % k] = (H, int(hp.fast_CIN_d) * L)
            else:
                shapes["cin_f_%d" % k] = (H * hp.FIELD_COUNT, L)
            if hp.enable_BN is True:  # gamma, beta of tf.layers.batch_normalization (xDeepFM.py:235-240)
                shapes["bn_cin_g_%d" % k] = (L,)
                shapes["bn_cin_o_%d" % k] = (L,)
        shapes["cin_w_out"] = (sum(L - nh for (_, L, nh) in cin_shapes(hp)), 1)
        shapes["cin_b_out"] = (1,)
    if hp.use_DNN_part:
        last = hp.FIELD_COUNT * hp.dim
        for k, size in enumerate(hp.layer_sizes):
            shapes["dnn_w_%d" % k] = (last, int(size))
            shapes["dnn_b_%d" % k] = (int(size),)
            if hp.enable_BN is True:  # xDeepFM.py:498-503
                shapes["bn_dnn_g_%d" % k] = (int(size),)
                shapes["bn_dnn_o_%d" % k] = (int(size),)
            last = int(size)
        shapes["dnn_w_out"] = (last, 1)
        shapes["dnn_b_out"] = (1,)
    return shapes


class DeviceBatch(object):
    """One mini-batch resident in HBM (the dnn view of the iterator as CSR)."""

    __slots__ = ("B", "nnz", "labels", "ids", "weights", "row_ptr", "tok_row", "sorted_ids",
                 "sorted_tok", "sorted_row", "sorted_w", "planned", "planned_gather", "plan_event", "ready_event", "src_table", "src_lin", "src_ids", "src_N", "route",
                 "flat")


class XDeepFMEngine(object):
    def __init__(self, hp, device=None, seed=None, embedding_update="dense_exact", table_rows=None,
                 compute_dtype=None):
        if not torch.cuda.is_available():
            raise RuntimeError(
                "recommenders_amd needs a ROCm GPU: the xDeepFM path runs in HIP kernels "
                "(libxdfm.so) and has no CPU fallback"
            )
        self.hp = hp
        self.device = torch.device(device) if device is not None else torch.device(
            "cuda", torch.cuda.current_device()
        )
        self._dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.lib = _lib.lib
        # table_rows: rows held by this process (the sharded engine keeps FEATURE_COUNT / world)
        self.N = int(hp.FEATURE_COUNT) if table_rows is None else int(table_rows)
        self.F, self.D = int(hp.FIELD_COUNT), int(hp.dim)
        self.fast_d = int(hp.fast_CIN_d) if hp.fast_CIN_d > 0 else 0
        self.use_bn = hp.enable_BN is True
        self._training = False  # batch statistics (train) or moving statistics (eval / infer)
        # dropout of the DNN layers (base_model.py:335-336): keep = 1 - hparams.dropout[k] in training
        self.use_dropout = bool(hp.user_dropout and hp.use_DNN_part)
        if self.use_dropout and len(hp.dropout) < len(hp.layer_sizes):
            raise ValueError("dropout needs one rate per DNN layer")
        self.dropout_seed = 0x5DEECE66D ^ (0 if seed is None else int(seed))
        self._dropout_live = False  # set by a training forward pass, read by the backward pass
        self.cross_act = 0
        self.cross_path = False
        if hp.use_CIN_part:
            if hp.cross_activation not in _ACT:
                raise ValueError("this activations not defined {0}".format(hp.cross_activation))
            self.cross_act = _ACT[hp.cross_activation]
        self.cross_soft = self.cross_act == _SOFTMAX
        if self.cross_soft:
            self.cross_act = 0
        # activation and / or batch norm between a CIN layer's product and its split: the layer
        # kernels then write raw curr for every channel and elementwise kernels do the rest
        self.cross_path = bool(self.cross_act or self.cross_soft or (self.use_bn and hp.use_CIN_part))
        if hp.loss not in _LOSS:
            raise ValueError("this loss not defined {0}".format(hp.loss))
        if hp.method not in ("classification", "regression"):
            raise ValueError(
                "method must be regression or classification, but now is {0}".format(hp.method)
            )
        if hp.use_DNN_part:
            for k in range(len(hp.layer_sizes)):
                if hp.activation[k] not in _ACT:
                    raise ValueError("this activations not defined {0}".format(hp.activation[k]))
        if not (hp.use_Linear_part or hp.use_FM_part or hp.use_CIN_part or hp.use_DNN_part):
            raise ValueError("xDeepFM needs at least one of the Linear/FM/CIN/DNN parts")
        # compute_dtype "bf16": CIN and DNN products on the bf16 matrix cores (fp32 accumulation, fp32
        # parameters, gradients and optimizer state) -- the wide shapes of BASELINE configs[4]
        if compute_dtype is None:
            compute_dtype = hp.values().get("compute_dtype", os.environ.get("XDFM_COMPUTE_DTYPE", "fp32"))
        if compute_dtype not in ("fp32", "bf16"):
            raise ValueError("compute_dtype must be fp32 or bf16")
        self.bf16 = compute_dtype == "bf16"
        if self.bf16:
            if self.fast_d or self.cross_path or self.use_bn or self.use_dropout or any(
                    _ACT[hp.activation[k]] == _SOFTMAX for k in range(len(hp.layer_sizes)) if hp.use_DNN_part):
                raise ValueError("compute_dtype=bf16 serves the plain xDeepFM layers (no fast_CIN, cross_activation, "
                                 "batch norm, dropout or softmax)")
            if hp.use_CIN_part:
                for (H, L, nh) in cin_shapes(hp):
                    if H > 128 or nh > 128:
                        raise ValueError("compute_dtype=bf16: at most 128 hidden channels per CIN layer "
                                         "(cross_layer_sizes up to 256 with split_half)")
        if embedding_update not in ("dense_exact", "sparse_rows", "sparse_lazy"):
            raise ValueError("embedding_update must be dense_exact, sparse_rows or sparse_lazy")
        self.embedding_update = embedding_update
        self.sparse = embedding_update != "dense_exact"
        self.opt_kind = _OPT.get(hp.optimizer)
        if self.opt_kind is None:
            self.opt_kind = 0  # unknown names fall back to SGD (base_model.py:301-302)
        # sparse_rows = the reference's numbers from row updates: TF moves every row every step (Adam's
        # m keeps decaying into p without a gradient; RMSProp / Adadelta decay their accumulators), so
        # every record carries the step it is current for and whoever reads a row applies the steps it
        # missed (xdfm_row_age: the gathers in registers, the row update in its one read-modify-write;
        # xdfm_embed_catchup writes aged rows back before checkpoints).  sparse_lazy skips them (a
        # different algorithm for absent rows).
        self.catchup_on = embedding_update == "sparse_rows" and self.opt_kind in (1, 3, 4, 5)
        self._rows_stale = False
        self._age_args = {}
        if self.sparse:
            if hp.embed_l1 != 0.0 or hp.embed_l2 != 0.0:
                raise ValueError("sparse_rows needs embed_l1 = embed_l2 = 0 (their gradient is dense)")
            if hp.use_Linear_part and (hp.layer_l1 != 0.0 or hp.layer_l2 != 0.0):
                raise ValueError(
                    "sparse_rows with the linear part needs layer_l1 = layer_l2 = 0 (lin_w is a "
                    "FEATURE_COUNT-sized layer parameter)"
                )
            if hp.is_clip_norm:
                raise ValueError("sparse_rows cannot clip by the norm of a dense gradient")

        dev = self.device
        f32 = torch.float32
        # ---- parameters (values are set by load_state / init)
        need_m = self.opt_kind != 0
        need_v = self.opt_kind in (1, 4, 5)  # Adam v, Adadelta accum_update, FTRL linear
        m0 = _SLOT_M_INIT.get(self.opt_kind, 0.0)
        # sparse_rows keeps everything a row's update touches in one record of whole 128-byte
        # lines: [row | lin w, m, v | pad] [slot m | slot v | pad]; the forward gather reads the
        # first line only, the update two lines instead of six scattered ones.  The tensors
        # below are strided views into it (row strides go to the kernels).
        self.interleaved = self.sparse and os.environ.get("XDFM_TABLE_RECORDS", "1") != "0"
        self.row_step, self.step_stride = None, 1
        if self.interleaved:
            D = self.D
            line0 = (D + 4 + 31) // 32 * 32
            self.table_stride = self.lin_stride = line0 + (2 * D + 31) // 32 * 32
            self._records = torch.zeros((self.N, self.table_stride), dtype=f32, device=dev)
            rec = self._records
            self.table = rec[:, 0:D]
            self.lin_w = rec[:, D] if hp.use_Linear_part else None
            self.table_m = rec[:, line0:line0 + D] if need_m else None
            self.table_v = rec[:, line0 + D:line0 + 2 * D] if need_v else None
            self.lin_m = rec[:, D + 1] if (need_m and hp.use_Linear_part) else None
            self.lin_v = rec[:, D + 2] if (need_v and hp.use_Linear_part) else None
            if self.catchup_on:  # the step a row's (p, m, v) are current for, in its first line
                self.row_step = rec[:, D + 3].view(torch.int32)
                self.step_stride = self.table_stride
            for t in (self.table_m, self.lin_m):
                if t is not None and m0 != 0.0:
                    t.fill_(m0)
        else:
            self.table_stride, self.lin_stride = self.D, 1
            self.table = torch.zeros((self.N, self.D), dtype=f32, device=dev)
            self.lin_w = torch.zeros((self.N,), dtype=f32, device=dev) if hp.use_Linear_part else None
            if self.catchup_on:
                self.row_step = torch.zeros((self.N,), dtype=torch.int32, device=dev)
        self.dense_shapes = dense_param_shapes(hp)
        # one flat buffer; segment i = [offs[i], offs[i + 1]).  bf16 path: every parameter starts on a 16-byte
        # boundary (the matrix products load their operands 16 bytes at a time); the filler at a segment's end
        # holds zeros, takes zero gradients and stays zero under every optimizer, and adds nothing to a norm
        align = 4 if self.bf16 else 1
        offs = [0]
        sizes = []
        for shape in self.dense_shapes.values():
            sizes.append(int(np.prod(shape)))
            offs.append((offs[-1] + sizes[-1] + align - 1) // align * align)
        self.dense_off = offs
        self._dense_sizes = sizes
        n_dense = max(offs[-1], 1)
        self.dense = torch.zeros((n_dense,), dtype=f32, device=dev)
        self.dense_grad = torch.zeros((n_dense,), dtype=f32, device=dev)
        self.views, self.grad_views = {}, {}
        for i, (name, shape) in enumerate(self.dense_shapes.items()):
            self.views[name] = self.dense[offs[i] : offs[i] + sizes[i]].view(shape)
            self.grad_views[name] = self.dense_grad[offs[i] : offs[i] + sizes[i]].view(shape)
        self.bn_state = {name: torch.zeros(shape, dtype=f32, device=dev)
                         for name, shape in bn_state_shapes(hp).items()}
        # ---- optimizer slots
        def slot(like, on, value=0.0):
            return torch.full_like(like, value) if on else None

        if not self.interleaved:
            self.table_m, self.table_v = slot(self.table, need_m, m0), slot(self.table, need_v)
            self.lin_m = slot(self.lin_w, need_m, m0) if self.lin_w is not None else None
            self.lin_v = slot(self.lin_w, need_v) if self.lin_w is not None else None
        self.dense_m, self.dense_v = slot(self.dense, need_m, m0), slot(self.dense, need_v)
        self.step = 0
        self.b1p = np.float32(1.0)
        self.b2p = np.float32(1.0)
        # device copy {b1^t, b2^t, Adam step size, t}: what the kernels read, so that no host
        # scalar changes between steps (a step can be replayed from a HIP graph)
        self.opt_state = torch.tensor([1.0, 1.0, 0.0, 0.0], dtype=f32, device=dev)
        # Adam's missed steps in closed form: coefficients by (steps missed, step) -- csrc/drift.h
        self.drift_tab = None
        if self.catchup_on and self.opt_kind == 1:
            sz = ctypes.c_size_t()
            _lib.check(self.lib.xdfm_drift_tab_floats(ctypes.byref(sz)))
            self.drift_tab = torch.zeros((sz.value,), dtype=f32, device=dev)
            self._init_drift_tab()
        # dense_exact keeps a dense gradient image of the table, all zero between steps
        if embedding_update == "dense_exact":
            self.grad_table = torch.zeros_like(self.table)
            self.grad_lin = torch.zeros_like(self.lin_w) if self.lin_w is not None else None
        else:
            self.grad_table = self.grad_lin = None
        # segment tables for xdfm_dense_update (host arrays)
        self._seg = self._segment_tables()
        self.err_flag = torch.zeros((1,), dtype=torch.int32, device=dev)
        # [data_loss, reg_loss] side by side: one scan yields (data, data + reg) at the end of a step
        self._loss2 = torch.zeros((2,), dtype=f32, device=dev)
        self.data_loss = self._loss2[0:1]
        self.reg_loss = self._loss2[1:2]
        self.stats = torch.zeros((max(len(self.dense_shapes), 1) * 260 + 520,), dtype=f32, device=dev)
        self.loss_scratch = torch.zeros((1032,), dtype=f32, device=dev)  # [1024]: xdfm_head_loss's block counter
        self._cap_B = self._cap_nnz = 0
        self.ws = {}
        self._ws_token = 0  # bumped whenever the workspace is re-allocated
        self._eval_ws = None
        # name -> [(start_event, end_event)]: filled when bench.py asks for in-situ kernel times
        self.kernel_events = None
        self._side_stream = None
        self._dnn_stream = None
        # with both towers on, the DNN tower (small, latency/HBM-bound launches) runs on a second
        # stream beside the CIN (matrix-core bound): the two share nothing but emb_out and d_emb
        # regularisation loss (a function of the parameters alone) beside the forward pass
        self.early_reg = (self.overlap_dnn_wanted(hp) and self.sparse and not hp.is_clip_norm
                          and 0 < len(self.dense_shapes) <= 32)
        self.prepack = os.environ.get("XDFM_CIN_PREPACK", "1") != "0"
        # pooled channels summed over d first (S = sum_d x0*xk, pooled = S Wp): 1/D of their MACs, but
        # measured slower on configs[2] (0.629 against 0.572 ms/step: the layer kernels do not shrink
        # with the channel count and the pair kernels sit at 24/38 us) -- opt-in until they do
        self.pool_split = os.environ.get("XDFM_POOL_SPLIT", "0") != "0"
        self.pooled_layer = os.environ.get("XDFM_POOLED_LAYER", "1") != "0"
        self.head_in_layer = os.environ.get("XDFM_HEAD_IN_LAYER", "1") != "0"
        self._loss_partials = 0
        self._split = {}
        self._fused_pool = set()
        self.split_dw = os.environ.get("XDFM_SPLIT_DW", "1") != "0"
        # the side work of backward() on two streams (what the large kernels wait for | the rest).  Off: one
        # side stream carries both, the chain's part first and joined as soon as it is done (0.3695 against
        # 0.3954 ms per step with the second stream, whose kernels took room from the chain's)
        self.side_split = os.environ.get("XDFM_SIDE_SPLIT", "0") != "0"
        self._dnn_stream2 = None
        self.main_first = os.environ.get("XDFM_MAIN_FIRST", "1") != "0"
        self.defer_dw = os.environ.get("XDFM_DEFER_DW", "1") != "0"
        self.tower_serial = os.environ.get("XDFM_TOWER_SERIAL", "1") != "0"
        self._plan_pending = None
        # bf16 path: the pooled channels' pair sums on the bf16 matrix cores too (0: the exact fp32 forms)
        self.pair_bf16 = os.environ.get("XDFM_PAIR_BF16", "1") != "0"
        self._plan_at = int(os.environ.get("XDFM_PLAN_AT", "1"))  # 0: ahead of the step, 1: behind the filter images, 2: behind the gather
        self.serial_plan = False  # bench.py's per-kernel timing: the id sort on the main stream
        self._late_event = None
        self._late_event2 = None
        self._late_work = None
        self._late_multi = None
        self.dw_multi_tail = os.environ.get("XDFM_DW_MULTI_TAIL", "1") != "0"
        self._fold_dx = False
        self.overlap_dnn = self.overlap_dnn_wanted(hp) and not self.bf16
        if self.bf16:
            self.early_reg = False
        # Fewer hardware queues than the step has parallel branches (main, id sort, DNN tower / late
        # filter gradient): hipGraphLaunch of the branched capture died inside the runtime with
        # GPU_MAX_HW_QUEUES=2 (round 2).  The step is then FOLDED onto one stream -- the same kernels
        # (the DNN tower's input gradient meets the CIN's in d_emb in another order: last-bit
        # differences, test_folded_step_...) -- and its capture is a plain chain that needs one queue.
        self.folded = hw_queue_limit() is not None and hw_queue_limit() < 4
        if self.folded:
            self.overlap_dnn = False
            self.early_reg = False
            self.split_dw = False
            self.serial_plan = True
        self.mean_count = 0  # set by the sharded engine: global batch size of the step
        self.init_parameters(seed)

    # ------------------------------------------------------------------ params
    def init_parameters(self, seed=None):
        """Initialise like base_model.py:169-213 (distribution only: TF's random stream is
        not reproducible; parity runs use load_state).  Drawn on the device, so a 33M-row
        table does not pass through host memory."""
        hp = self.hp
        gen = torch.Generator(device=self.device)
        gen.manual_seed(0 if seed is None else int(seed))

        def draw(dst, shape):
            method = hp.init_method
            fan_in = shape[0]
            fan_out = shape[1] if len(shape) > 1 else shape[0]
            if method == "uniform":
                dst.uniform_(-hp.init_value, hp.init_value, generator=gen)
            elif method == "normal":
                dst.normal_(0.0, hp.init_value, generator=gen)
            elif method in ("xavier_uniform", "he_uniform"):
                scale, fan = (1.0, (fan_in + fan_out) / 2.0) if method.startswith("xavier") else (2.0, fan_in)
                lim = float(np.sqrt(3.0 * scale / fan))
                dst.uniform_(-lim, lim, generator=gen)
            else:
                if method in ("xavier_normal", "he_normal"):
                    scale, fan = (1.0, (fan_in + fan_out) / 2.0) if method.startswith("xavier") else (2.0, fan_in)
                    std = float(np.sqrt(scale / fan)) / 0.87962566103423978
                else:  # tnormal and the fallback
                    std = float(hp.init_value)
                torch.nn.init.trunc_normal_(dst, mean=0.0, std=std, a=-2.0 * std, b=2.0 * std, generator=gen)

        draw(self.table, (self.N, self.D))
        if self.lin_w is not None:
            draw(self.lin_w, (self.N, 1))
        for name, shape in self.dense_shapes.items():
            if name.startswith("bn_"):
                self.views[name].fill_(1.0 if "_g_" in name else 0.0)  # gamma = 1, beta = 0
            elif name.endswith("_b") or "_b_" in name:
                self.views[name].zero_()
            else:
                draw(self.views[name], shape)
        for name, t in self.bn_state.items():
            t.fill_(1.0 if "_mv_" in name else 0.0)  # moving mean = 0, moving variance = 1

    def state_dict(self):
        self.flush_rows()
        if self.opt_kind == 1 or self.catchup_on:  # the device copy is authoritative (graph replays do not run Python)
            b1p, b2p, _, t = self.opt_state.tolist()
            self.b1p, self.b2p, self.step = np.float32(b1p), np.float32(b2p), int(round(t))
        sd = {"embedding": self.table, "step": self.step, "b1p": float(self.b1p), "b2p": float(self.b2p)}
        if self.lin_w is not None:
            sd["lin_w"] = self.lin_w.view(-1, 1)
        for name in self.dense_shapes:
            sd[name] = self.views[name]
        sd.update(self.bn_state)
        for key, t in (("slot/table_m", self.table_m), ("slot/table_v", self.table_v),
                       ("slot/lin_m", self.lin_m), ("slot/lin_v", self.lin_v),
                       ("slot/dense_m", self.dense_m), ("slot/dense_v", self.dense_v)):
            if t is not None:
                # (the dense slots without the alignment filler of the bf16 path: one layout for every engine)
                sd[key] = self._dense_compact(t) if key.startswith("slot/dense_") else t
        return sd

    def _dense_compact(self, flat):
        offs, sizes = self.dense_off, self._dense_sizes
        if offs[-1] == sum(sizes):
            return flat
        return torch.cat([flat[offs[i]:offs[i] + sizes[i]] for i in range(len(sizes))]) if sizes else flat[:0]

    def _dense_expand(self, flat, compact):
        """compact (parameters back to back) into the flat buffer's segments."""
        offs, sizes = self.dense_off, self._dense_sizes
        compact = torch.as_tensor(compact).reshape(-1)
        if offs[-1] == sum(sizes):
            flat[:compact.numel()].copy_(compact.to(flat.device))
            return
        at = 0
        for i in range(len(sizes)):
            flat[offs[i]:offs[i] + sizes[i]].copy_(compact[at:at + sizes[i]].to(flat.device))
            at += sizes[i]

    def load_state(self, sd, strict=True):
        """Load parameters (and, when present, optimizer slots) from a dict of tensors /
        arrays keyed like state_dict()."""
        def put(dst, src):
            src = torch.as_tensor(np.asarray(src) if not torch.is_tensor(src) else src)
            dst.copy_(src.to(torch.float32).reshape(dst.shape))

        names = (["embedding"] + (["lin_w"] if self.lin_w is not None else []) + list(self.dense_shapes)
                 + list(self.bn_state))
        for name in names:
            if name not in sd:
                if strict:
                    raise KeyError("missing parameter {0}".format(name))
                continue
            if name == "embedding":
                put(self.table, sd[name])
            elif name == "lin_w":
                put(self.lin_w, sd[name])
            elif name in self.bn_state:
                put(self.bn_state[name], sd[name])
            else:
                put(self.views[name], sd[name])
        for key, t in (("slot/table_m", self.table_m), ("slot/table_v", self.table_v),
                       ("slot/lin_m", self.lin_m), ("slot/lin_v", self.lin_v),
                       ("slot/dense_m", self.dense_m), ("slot/dense_v", self.dense_v)):
            if t is not None and key in sd:
                if key.startswith("slot/dense_"):
                    self._dense_expand(t, sd[key])
                else:
                    put(t, sd[key])
        if "step" in sd:
            self.step = int(sd["step"])
            self.b1p = np.float32(sd.get("b1p", 0.9 ** self.step))
            self.b2p = np.float32(sd.get("b2p", 0.999 ** self.step))
            self.opt_state.copy_(torch.tensor([float(self.b1p), float(self.b2p), 0.0, float(self.step)]))
        if self.row_step is not None:  # a state dict holds flushed rows: all current for `step`
            self.row_step.fill_(self.step)
            self._rows_stale = False
        if self.drift_tab is not None:
            self._init_drift_tab()

    def _init_drift_tab(self):
        b1, b2, eps = _OPT_CONST[self.opt_kind]
        _lib.check(self.lib.xdfm_drift_tab_init(_ptr(self.drift_tab), _ptr(self.opt_state),
                                                float(self.hp.learning_rate), b1, b2, eps, self._stream()))

    def _age(self, step_offset=0):
        """xdfm_row_age for this engine's table (None when rows never age): step_offset = -1 once
        xdfm_opt_advance has counted the step being applied."""
        if not self.catchup_on:
            return None
        hit = self._age_args.get(step_offset)
        if hit is None:
            opt = self._opt_args()  # the kernels read the step scalars from opt_state, not from here
            age = _lib.RowAge(self.row_step.data_ptr(), self.step_stride,
                              self.table_m.data_ptr() if self.table_m is not None else None,
                              self.table_v.data_ptr() if self.table_v is not None else None,
                              self.lin_m.data_ptr() if self.lin_m is not None else None,
                              self.lin_v.data_ptr() if self.lin_v is not None else None,
                              self.drift_tab.data_ptr() if self.drift_tab is not None else None,
                              ctypes.pointer(opt), int(step_offset), 0)
            hit = self._age_args[step_offset] = (age, opt)
        return ctypes.byref(hit[0])

    @staticmethod
    def overlap_dnn_wanted(hp):
        return bool(hp.use_CIN_part and hp.use_DNN_part) and os.environ.get("XDFM_OVERLAP_DNN", "1") != "0"

    def _reg_loss_early(self):
        seg, n_seg = self._seg, len(self.dense_shapes)
        _lib.check(
            self.lib.xdfm_dense_reg_loss(
                _ptr(self.dense), n_seg, seg["off"].ctypes.data_as(ctypes.c_void_p),
                seg["l2"].ctypes.data_as(ctypes.c_void_p), seg["l1"].ctypes.data_as(ctypes.c_void_p),
                seg["nl2"].ctypes.data_as(ctypes.c_void_p), _ptr(self.stats), _ptr(self.reg_loss), 0,
                self._stream(),
            )
        )

    def _segment_tables(self):
        hp = self.hp
        n = len(self.dense_shapes)
        l2 = np.zeros(max(n, 1), dtype=np.float32)
        l1 = np.zeros(max(n, 1), dtype=np.float32)
        nl2 = np.zeros(max(n, 1), dtype=np.float32)
        for i, name in enumerate(self.dense_shapes):
            if name.startswith("bn_"):  # tf.layers variables: in no params list, no regulariser
                continue
            if name.startswith(("cin_f_", "fcin_")):  # cross_params (base_model.py:158-166)
                l1[i] = hp.cross_l1
                nl2[i] = hp.cross_l2
            else:  # layer_params (base_model.py:128-132, :143-148); biases included
                l2[i] = hp.layer_l2
                l1[i] = hp.layer_l1
        off = np.asarray(self.dense_off, dtype=np.int64)
        return {"off": off, "l2": l2, "l1": l1, "nl2": nl2}

    # --------------------------------------------------------------- workspace
    def _mark(self, name):
        """HIP events around one C-ABI call on the launch stream (bench.py only)."""
        if self.kernel_events is None:
            return None
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        self.kernel_events.setdefault(name, []).append(ev)
        ev[0].record(torch.cuda.current_stream(self.device))
        return ev

    def _done(self, ev):
        if ev is not None:
            ev[1].record(torch.cuda.current_stream(self.device))

    def _stream(self):
        # (the raw handle: torch.cuda.current_stream() builds a Stream object per call, 5 us each and a
        # dozen calls per step on the eager paths)
        return ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(self._dev_index))

    def _ensure_ws(self, B, nnz):
        if B <= self._cap_B and nnz <= self._cap_nnz:
            return
        hp, dev, f32 = self.hp, self.device, torch.float32
        B = max(B, self._cap_B, int(hp.batch_size))
        nnz = max(nnz, self._cap_nnz, B * self.F)
        F, D = self.F, self.D
        ws = {}

        def buf(name, *shape, dtype=f32):
            ws[name] = torch.zeros(shape, dtype=dtype, device=dev)

        buf("emb_out", B, F * D)
        buf("d_emb", B, F * D)
        buf("logit", B)
        buf("pred", B)
        buf("g_logit", B)
        buf("dummy", 8)
        if hp.use_Linear_part:
            buf("lin_out", B)
        if hp.use_FM_part:
            buf("fm_out", B)
            buf("fm_sum", B, D)
        sz = ctypes.c_size_t()
        _lib.check(self.lib.xdfm_dense_bwd_workspace_floats(B, 1, 1, ctypes.byref(sz)))
        part_floats = sz.value
        pool_part_floats = 1
        bf16_floats = 1
        self._split = {}
        self._fused_pool = set()
        if hp.use_CIN_part:
            shapes = cin_shapes(hp)
            final_len = sum(L - nh for (_, L, nh) in shapes)
            buf("pooled", B, final_len)
            buf("d_pooled", B, final_len)
            buf("cin_out", B)
            for k, (H, L, nh) in enumerate(shapes):
                if nh > 0:
                    buf("hid_%d" % k, B, nh * D)
                    buf("d_hid_%d" % k, B, nh * D)
                if self.cross_path:
                    # a non-identity cross_activation / batch norm: raw / activated curr of every
                    # channel is kept
                    buf("cur_%d" % k, B, L * D)
                    if self.use_bn:
                        buf("cbn_%d" % k, B, L * D)
                        buf("bn_stats_cin_%d" % k, 2 * L)
            if self.cross_path:
                buf("d_cur", B, max(L for (_, L, _) in shapes) * D)
            self._split = {}
            self._fused_pool = set()
            max_pairs = 0
            for k, (H, L, nh) in enumerate(shapes):
                if self.fast_d:
                    _lib.check(self.lib.xdfm_fast_cin_bwd_workspace_floats(B, F, H, D, L, self.fast_d,
                                                                           ctypes.byref(sz)))
                    part_floats = max(part_floats, sz.value)
                    continue
                # The pooled channels (n >= nh) take their sum over the embedding dimension first
                # (xdfm_cin_pair_sum + a plain product: 1/D of the multiply-adds); the layer kernels
                # then carry the nh hidden channels only.  Needs the packed path for that narrower
                # layer (or no hidden channel at all: the last layer).
                Lk = L
                if self.bf16:
                    # every layer: pooled channels through the pair sums and a bf16 product, hidden
                    # channels through the bf16 layer kernels
                    if L - nh > 0:
                        npair = ctypes.c_int64()
                        _lib.check(self.lib.xdfm_cin_pair_count(F, H, 1 if k == 0 else 0, ctypes.byref(npair)))
                        P, Lp = int(npair.value), L - nh
                        self._split[k] = (P, Lp)
                        max_pairs = max(max_pairs, P)
                        buf("S_%d" % k, B, P)
                        buf("wp_%d" % k, P, Lp)
                        buf("dwp_%d" % k, P, Lp)
                        buf("pool_%d" % k, B, Lp)
                        _lib.check(self.lib.xdfm_dense_bf16_workspace_floats(B, P, Lp, ctypes.byref(sz)))
                        bf16_floats = max(bf16_floats, sz.value)
                    if nh > 0:
                        fb, db = ctypes.c_size_t(), ctypes.c_size_t()
                        _lib.check(self.lib.xdfm_cin_bf16_image_bytes(F, H, nh, ctypes.byref(fb), ctypes.byref(db)))
                        buf("bimg_f_%d" % k, fb.value, dtype=torch.uint8)
                        buf("bimg_d_%d" % k, db.value, dtype=torch.uint8)
                        _lib.check(self.lib.xdfm_cin_bf16_bwd_dw_workspace_floats(B, F, H, D, nh, ctypes.byref(sz)))
                        bf16_floats = max(bf16_floats, sz.value)
                    continue
                if nh == 0 and k > 0 and self.pooled_layer and not self.cross_path:
                    # a last layer of small shape: one forward and one backward kernel on the f32
                    # matrix cores, sum over the embedding dimension first (cin_pooled_layer_kernels)
                    _lib.check(self.lib.xdfm_cin_pooled_workspace_floats(B, F, H, D, L, ctypes.byref(sz)))
                    if sz.value:
                        self._fused_pool.add(k)
                        part_floats = max(part_floats, sz.value)
                        continue
                if self.pool_split and self.prepack and not self.cross_path and L - nh > 0:
                    fsz, bsz = ctypes.c_size_t(), ctypes.c_size_t()
                    if nh > 0:
                        _lib.check(self.lib.xdfm_cin_pack_floats(B, F, H, D, nh, 1 if k == 0 else 0,
                                                                 ctypes.byref(fsz), ctypes.byref(bsz)))
                    if nh == 0 or (fsz.value and bsz.value):
                        npair = ctypes.c_int64()
                        _lib.check(self.lib.xdfm_cin_pair_count(F, H, 1 if k == 0 else 0, ctypes.byref(npair)))
                        P, Lp = int(npair.value), L - nh
                        self._split[k] = (P, Lp)
                        max_pairs = max(max_pairs, P)
                        buf("S_%d" % k, B, P)
                        buf("wp_%d" % k, P, Lp)
                        buf("dwp_%d" % k, P, Lp)
                        buf("pool_%d" % k, B, Lp)
                        _lib.check(self.lib.xdfm_dense_bwd_workspace_floats(B, P, Lp, ctypes.byref(sz)))
                        pool_part_floats = max(pool_part_floats, sz.value)
                        Lk = nh
                if Lk == 0:
                    continue
                _lib.check(self.lib.xdfm_cin_bwd_workspace_floats(B, F, H, D, Lk, ctypes.byref(sz)))
                part_floats = max(part_floats, sz.value)
                _lib.check(self.lib.xdfm_cin_fwd_workspace_floats(B, F, H, D, Lk, ctypes.byref(sz)))
                part_floats = max(part_floats, sz.value)
                # filter images packed once per step, ahead of the layer calls (where the shape
                # has a packed path)
                fsz, bsz = ctypes.c_size_t(), ctypes.c_size_t()
                _lib.check(self.lib.xdfm_cin_pack_floats(B, F, H, D, Lk, 1 if k == 0 else 0, ctypes.byref(fsz),
                                                         ctypes.byref(bsz)))
                if self.prepack and fsz.value:
                    buf("img_f_%d" % k, fsz.value)
                if self.prepack and bsz.value:
                    buf("img_b_%d" % k, bsz.value)
            if self._split:
                buf("dS", B, max_pairs)
                buf("part_pool", pool_part_floats)
                buf("zeros", max(Lp for (_, Lp) in self._split.values()))  # the bias of the plain product
                buf("db_pool", max(Lp for (_, Lp) in self._split.values()))  # its (unused) bias gradient
            _lib.check(self.lib.xdfm_dense_bwd_workspace_floats(B, final_len, 1, ctypes.byref(sz)))
            part_floats = max(part_floats, sz.value)
        dnn_part_floats = 1
        if hp.use_DNN_part:
            last = F * D
            for k, size in enumerate(hp.layer_sizes):
                buf("h_%d" % k, B, int(size))
                buf("d_h_%d" % k, B, int(size))
                if self.use_bn:
                    buf("z_%d" % k, B, int(size))
                    buf("bn_stats_dnn_%d" % k, 2 * int(size))
                if self.use_dropout:
                    buf("drop_%d" % k, B, int(size))
                _lib.check(self.lib.xdfm_dense_bwd_workspace_floats(B, last, int(size), ctypes.byref(sz)))
                dnn_part_floats = max(dnn_part_floats, sz.value)
                if self.bf16:
                    _lib.check(self.lib.xdfm_dense_bf16_workspace_floats(B, last, int(size), ctypes.byref(sz)))
                    bf16_floats = max(bf16_floats, sz.value)
                last = int(size)
            buf("dnn_out", B)
            _lib.check(self.lib.xdfm_dense_bwd_workspace_floats(B, last, 1, ctypes.byref(sz)))
            dnn_part_floats = max(dnn_part_floats, sz.value)
        buf("part", max(part_floats, 1))
        if self.bf16:
            buf("bf16_ws", bf16_floats)
        # the DNN tower runs on its own stream beside the CIN: its partial sums get their own space
        # (the lin_b / CIN head weight gradients run there too)
        _lib.check(self.lib.xdfm_dense_bwd_workspace_floats(B, 1, 1, ctypes.byref(sz)))
        dnn_part_floats = max(dnn_part_floats, sz.value)
        if hp.use_CIN_part:
            _lib.check(self.lib.xdfm_dense_bwd_workspace_floats(B, ws["pooled"].shape[1], 1, ctypes.byref(sz)))
            dnn_part_floats = max(dnn_part_floats, sz.value)
        buf("part_dnn", dnn_part_floats)
        if self._tower_fused():
            jobs = self._dw_jobs(ws, B, sizes_only=True)
            _lib.check(self.lib.xdfm_dense_bwd_dw_multi_workspace_floats(jobs, len(jobs), B, ctypes.byref(sz)))
            buf("part_multi", max(sz.value, 1))
        buf("pieces", 2 * (nnz // 64 + 1) * (D + 2))
        if self.overlap_dnn:
            # the DNN tower's gradient w.r.t. emb_out, written beside the CIN backward and added to
            # d_emb where the table gradient reads it
            buf("d_emb_dnn", B, F * D)
        _lib.check(self.lib.xdfm_embed_plan_workspace_bytes(nnz, self.N, ctypes.byref(sz)))
        buf("plan_ws", sz.value, dtype=torch.uint8)
        if self.ws:
            # the old buffers may still be in use on the side streams, and captured graphs hold
            # their addresses: wait for the device, and let every GraphedStep / local graph see that
            # it is stale (they compare the token before a replay)
            torch.cuda.synchronize(self.device)
        self.ws = ws
        self._ws_token += 1
        self._cap_B, self._cap_nnz = B, nnz

    # ------------------------------------------------------------------- batch
    def upload(self, labels, dnn_feat_indices, dnn_feat_values, dnn_feat_weights):
        """numpy arrays of the iterator (iterator.py:204-218) -> DeviceBatch."""
        dev = self.device
        labels = np.ascontiguousarray(np.asarray(labels, dtype=np.float32).reshape(-1))
        ids = np.ascontiguousarray(np.asarray(dnn_feat_values, dtype=np.int64).reshape(-1))
        w = np.ascontiguousarray(np.asarray(dnn_feat_weights, dtype=np.float32).reshape(-1))
        idx = np.ascontiguousarray(np.asarray(dnn_feat_indices, dtype=np.int64).reshape(-1, 2))
        if not (len(ids) == len(w) == len(idx)):
            raise ValueError("dnn_feat_indices / values / weights disagree in length")
        def dev_copy(a):
            # arrays the iterator keeps for later epochs are read-only; they are only read here
            with warnings.catch_warnings():
                warnings.simplefilter("ignore", UserWarning)
                return torch.from_numpy(a).to(dev, non_blocking=True)

        return self.make_batch(dev_copy(labels), dev_copy(idx), dev_copy(ids), dev_copy(w))

    def make_batch(self, labels, dnn_idx, ids, weights):
        """Device tensors -> DeviceBatch (builds the CSR offsets on the device)."""
        b = DeviceBatch()
        b.B = int(labels.numel())
        b.nnz = int(ids.numel())
        b.labels, b.ids, b.weights = labels, ids, weights
        b.row_ptr = torch.empty((b.B * self.F + 1,), dtype=torch.int32, device=self.device)
        b.tok_row = torch.empty((max(b.nnz, 1),), dtype=torch.int32, device=self.device)
        _lib.check(
            self.lib.xdfm_build_row_ptr(_ptr(dnn_idx), b.nnz, b.B * self.F, _ptr(b.row_ptr),
                                        _ptr(b.tok_row), _ptr(self.err_flag), self._stream())
        )
        b.sorted_ids = b.sorted_tok = b.sorted_row = b.sorted_w = None
        b.planned = b.planned_gather = False
        b.plan_event = None
        # what another stream waits for before it reads the batch's arrays (routing one step ahead)
        b.ready_event = torch.cuda.Event()
        b.ready_event.record(torch.cuda.current_stream(self.device))
        b.route = None
        # the sharded engine points these at the rows it fetched from their owners
        b.src_table = b.src_lin = b.src_ids = None
        b.src_N = 0
        return b

    def make_batch_csr(self, labels, row_ptr, tok_row, ids, weights):
        """Device tensors with the lookup's CSR offsets (row_ptr[B*F + 1]) and token rows already
        built (the text reader's `row_ptr`, a GraphedStep's packed input) -> DeviceBatch; no kernel."""
        b = DeviceBatch()
        b.B = int(labels.numel())
        b.nnz = int(ids.numel())
        b.labels, b.ids, b.weights = labels, ids, weights
        b.row_ptr, b.tok_row = row_ptr, tok_row
        b.sorted_ids = b.sorted_tok = b.sorted_row = b.sorted_w = None
        b.planned = b.planned_gather = False
        b.plan_event = None
        b.ready_event = torch.cuda.Event()
        b.ready_event.record(torch.cuda.current_stream(self.device))
        b.route = None
        b.src_table = b.src_lin = b.src_ids = None
        b.src_N = 0
        return b

    def _plan_complete(self, batch):
        if not batch.planned:
            return False
        if batch.tok_row is None or batch.nnz == 0:
            return True  # nothing to bring into sorted order (a batch known by its ids only)
        return batch.sorted_row is not None and batch.planned_gather

    def plan_async(self, batch):
        """Start the id sort of `batch` (and the gather of its rows / weights into sorted order) on a
        side stream so that it runs beside the forward pass (it depends on the batch only);
        update() waits for it."""
        if self._plan_complete(batch):
            return
        if self.serial_plan:  # (measurement: the sort's kernels alone on the device)
            return self.plan(batch)
        self._ensure_ws(batch.B, batch.nnz)
        main = torch.cuda.current_stream(self.device)
        if self._side_stream is None:
            self._side_stream = torch.cuda.Stream(device=self.device)
        # allocated under the main stream, which is also where they are consumed and freed
        self._plan_buffers(batch)
        ready = torch.cuda.Event()
        ready.record(main)
        self._side_stream.wait_event(ready)
        with torch.cuda.stream(self._side_stream):
            self.plan(batch)
            batch.plan_event = torch.cuda.Event()
            batch.plan_event.record(self._side_stream)

    def _plan_buffers(self, batch):
        n = max(batch.nnz, 1)
        if batch.sorted_ids is None:
            batch.sorted_ids = torch.empty((n,), dtype=torch.int32, device=self.device)
            batch.sorted_tok = torch.empty((n,), dtype=torch.int32, device=self.device)
        if batch.sorted_row is None and batch.tok_row is not None and batch.nnz:
            batch.sorted_row = torch.empty((n,), dtype=torch.int32, device=self.device)
            batch.sorted_w = torch.empty((n,), dtype=torch.float32, device=self.device)
            batch.planned_gather = False

    def plan(self, batch, sort_only=False):
        """Sort the batch's tokens by feature id, then bring tok_row / weights into that order (what
        the table gradient reads per token); both on the current stream."""
        if self._plan_complete(batch):
            return
        self._ensure_ws(batch.B, batch.nnz)
        self._plan_buffers(batch)
        if not batch.planned:
            ws = self.ws["plan_ws"]
            ev = self._mark("embed_plan")
            _lib.check(
                self.lib.xdfm_embed_plan(_ptr(batch.ids), batch.nnz, self.N, _ptr(batch.sorted_ids),
                                         _ptr(batch.sorted_tok), _ptr(ws), ws.numel(), self._stream())
            )
            self._done(ev)
            batch.planned = True
        if not sort_only and batch.sorted_row is not None and not batch.planned_gather:
            _lib.check(
                self.lib.xdfm_embed_plan_gather(_ptr(batch.sorted_tok), _ptr(batch.tok_row), _ptr(batch.weights),
                                                batch.nnz, _ptr(batch.sorted_row), _ptr(batch.sorted_w),
                                                self._stream())
            )
            batch.planned_gather = True

    def check_errors(self):
        """Read the device error flag (synchronises) and raise like the reference's
        runtime would for out-of-range indices."""
        flag = int(self.err_flag.item())
        if flag:
            self.err_flag.zero_()
            msgs = [m for bit, m in _DEVERR.items() if flag & bit]
            raise ValueError("invalid batch: " + "; ".join(msgs))

    # ----------------------------------------------------------------- forward
    def forward(self, batch, with_loss=True, with_grad=False):
        hp, lib, ws, st = self.hp, self.lib, None, self._stream()
        # layer_keeps = keep_prob_train only under train() (base_model.py:377); eval / infer feed ones
        self._dropout_live = self.use_dropout and with_grad
        self._training = bool(with_grad)  # is_train_stage (base_model.py:378, :401, :415)
        # (scoring needs no flush: the gather ages the rows it reads, like the training forward pass)
        self._ensure_ws(batch.B, batch.nnz)
        ws = self.ws
        B, F, D = batch.B, self.F, self.D
        use_lin, use_fm = bool(hp.use_Linear_part), bool(hp.use_FM_part)
        if hp.use_CIN_part and not self.fast_d:
            self._pack_filters(B, with_grad)
        # train_step's id sort starts here, behind the first launch of the chain: captured ahead of everything,
        # the replay gave it the launch's own queue and the chain began a queue hop later (0.3677 -> 0.3632 ms).
        # (Captured behind the gather instead, the replay ran it after all other side work: 0.438 ms.)
        pend, self._plan_pending = self._plan_pending, None
        if pend is not None and self._plan_at == 1:
            self.plan_async(pend)
        ev = self._mark("embed_fwd")
        _lib.check(
            lib.xdfm_embed_fwd(
                _ptr(batch.src_table if batch.src_table is not None else self.table),
                _ptr(batch.src_lin if batch.src_table is not None else self.lin_w) if use_lin else None,
                batch.src_table.stride(0) if batch.src_table is not None else self.table_stride,
                (batch.src_lin.stride(0) if use_lin else 0) if batch.src_table is not None else self.lin_stride,
                batch.src_N if batch.src_table is not None else self.N, D, F, B,
                _ptr(batch.row_ptr), _ptr(batch.src_ids if batch.src_table is not None else batch.ids),
                _ptr(batch.weights), batch.nnz,
                _ptr(ws["emb_out"]), _ptr(ws["lin_out"]) if use_lin else None,
                _ptr(ws["fm_out"]) if use_fm else None, _ptr(ws["fm_sum"]) if use_fm else None,
                _ptr(self.err_flag), None if batch.src_table is not None else self._age(0), st,
            )
        )
        self._done(ev)
        if pend is not None and self._plan_at == 2:
            self.plan_async(pend)
        dnn_done = None
        if self.overlap_dnn:
            # fork: the DNN tower starts on its stream as soon as emb_out is there
            if self._dnn_stream is None:
                self._dnn_stream = torch.cuda.Stream(device=self.device)
            side = self._dnn_stream
            emb_ready = torch.cuda.Event()
            emb_ready.record(torch.cuda.current_stream(self.device))
            # d_emb without a clear and without a second array: the CIN layer whose backward runs
            # first writes it, the one that runs last folds the DNN tower's gradient in
            self._fold_dx = bool(with_grad and hp.use_CIN_part and not self.fast_d and not self.cross_path
                                 and "img_b_0" in ws
                                 and all(("img_b_%d" % k) in ws or (k in self._split and nh == 0)
                                         or k in self._fused_pool
                                         for k, (_, _, nh) in enumerate(cin_shapes(hp)))
                                 and self._dnn_plain() and os.environ.get("XDFM_FOLD_DX", "1") != "0")
            serial_tower = bool(self.tower_serial and self._tower_fused())
            if serial_tower:
                # the one-launch tower on the chain's own stream, ahead of the CIN (13 us alone; beside the CIN
                # kernel it finishes 12-25 us AFTER it)
                self._dnn_forward(B)

            def fork_tower():
                side.wait_event(emb_ready)
                with torch.cuda.stream(side):
                    if not serial_tower:
                        self._dnn_forward(B)
                    if with_grad and not self._fold_dx:
                        ws["d_emb"][:B].zero_()  # the CIN backward accumulates into it
                    done = torch.cuda.Event()
                    done.record(side)
                    # behind the join's event: the regularisation statistics gate nothing of the forward pass
                    if with_grad and self.early_reg:
                        self._reg_loss_early()
                return done

            # main_first: the CIN kernels are issued (captured) ahead of the tower's, so that the runtime keeps
            # the chain of large kernels on one hardware queue (a replayed graph continues a queue with the
            # FIRST node captured behind a node; a hop to another queue costs ~10 us against ~3)
            dnn_done = fork_tower if self.main_first else fork_tower()
        # training step with a CIN: head, logit, loss, d(loss)/d(logit) and d_pooled in one launch
        fused_head = bool(hp.use_CIN_part and with_grad and with_loss)
        self._head_fused = fused_head
        if hp.use_CIN_part:
            ev = self._mark("cin_fwd")
            shapes = cin_shapes(hp)
            final_len = ws["pooled"].shape[1]
            off = 0
            x0 = ws["emb_out"]
            xk = x0
            for k, (H, L, nh) in enumerate(shapes):
                xnext = ws["hid_%d" % k] if nh > 0 else None
                # with a cross_activation the layer writes raw curr for all L channels; the
                # activation kernel then produces the hidden / pooled split
                out_h, out_nh = (ws["cur_%d" % k], L) if self.cross_path else (xnext, nh)
                if self.bf16:
                    masked = 1 if k == 0 else 0
                    if nh > 0:
                        _lib.check(lib.xdfm_cin_bf16_fwd(_ptr(x0), _ptr(xk), _ptr(ws["bimg_f_%d" % k]), B, F, H, D, nh,
                                                         masked, _ptr(xnext), st))
                    if k in self._split:
                        P, Lp = self._split[k]
                        pair_sum = lib.xdfm_cin_pair_sum_bf16 if self.pair_bf16 else lib.xdfm_cin_pair_sum
                        _lib.check(pair_sum(_ptr(x0), _ptr(xk), B, F, H, D, masked, _ptr(ws["S_%d" % k]), st))
                        _lib.check(lib.xdfm_dense_bf16_fwd(_ptr(ws["S_%d" % k]), _ptr(ws["wp_%d" % k]), _ptr(ws["zeros"]),
                                                           B, P, Lp, 0, _ptr(ws["pool_%d" % k]), _ptr(ws["bf16_ws"]), st))
                        ws["pooled"][:B, off:off + Lp].copy_(ws["pool_%d" % k][:B])
                    off += L - nh
                    xk = xnext
                    continue
                if k in self._fused_pool:
                    if (fused_head and self.head_in_layer and k == len(shapes) - 1 and final_len <= 64
                            and B <= 16384 and hp.loss in ("log_loss", "cross_entropy_loss")):
                        # the rest of the forward pass rides in the last layer's launch (one operation
                        # fewer on the step's chain): it needs the DNN tower's output first
                        if dnn_done is not None:
                            if callable(dnn_done):
                                dnn_done = dnn_done()
                            torch.cuda.current_stream(self.device).wait_event(dnn_done)
                            dnn_done = None
                        elif hp.use_DNN_part and not self.overlap_dnn:
                            self._dnn_forward(B)
                        n_part = ctypes.c_int32()
                        _lib.check(
                            lib.xdfm_cin_pooled_fwd_head(
                                _ptr(x0), _ptr(xk), _ptr(self.views["cin_f_%d" % k]), B, F, H, D, L, 0,
                                _ptr(ws["pooled"]), off,
                                _ptr(ws["lin_out"]) if use_lin else None, _ptr(ws["fm_out"]) if use_fm else None,
                                _ptr(ws["dnn_out"]) if hp.use_DNN_part else None,
                                _ptr(self.views["lin_b"]) if use_lin else None,
                                _ptr(self.views["cin_w_out"]), _ptr(self.views["cin_b_out"]), final_len,
                                _ptr(batch.labels), int(self.mean_count or 0),
                                1 if hp.method == "classification" else 0, _LOSS[hp.loss], _ptr(ws["logit"]),
                                _ptr(ws["pred"]), _ptr(ws["cin_out"]), _ptr(ws["g_logit"]), _ptr(ws["d_pooled"]),
                                _ptr(self.loss_scratch), ctypes.byref(n_part), st,
                            )
                        )
                        self._loss_partials = int(n_part.value)
                        if self.overlap_dnn:
                            self._loss_deferred = True  # finished on the side stream (backward())
                        else:
                            self._finish_loss(B, st)
                        self._done(ev)
                        return ws["pred"][:B]
                    _lib.check(lib.xdfm_cin_pooled_fwd(_ptr(x0), _ptr(xk), _ptr(self.views["cin_f_%d" % k]), B, F, H,
                                                       D, L, 0, _ptr(ws["pooled"]), final_len, off, st))
                    off += L
                    xk = xnext
                    continue
                if k in self._split:
                    P, Lp = self._split[k]
                    if nh > 0:
                        _lib.check(
                            lib.xdfm_cin_fwd_packed(_ptr(x0), _ptr(xk), _ptr(ws["img_f_%d" % k]), B, F, H, D,
                                                    nh, nh, 1 if k == 0 else 0, _ptr(xnext), _ptr(ws["pooled"]),
                                                    final_len, off, st)
                        )
                    _lib.check(lib.xdfm_cin_pair_sum(_ptr(x0), _ptr(xk), B, F, H, D, 1 if k == 0 else 0,
                                                     _ptr(ws["S_%d" % k]), st))
                    _lib.check(lib.xdfm_dense_fwd(_ptr(ws["S_%d" % k]), _ptr(ws["wp_%d" % k]), _ptr(ws["zeros"]), B, P,
                                                  Lp, 0, _ptr(ws["pool_%d" % k]), st))
                    ws["pooled"][:B, off:off + Lp].copy_(ws["pool_%d" % k][:B])
                    off += L - nh
                    xk = xnext
                    continue
                if self.fast_d:
                    _lib.check(
                        lib.xdfm_fast_cin_fwd(_ptr(x0), _ptr(xk), _ptr(self.views["fcin_w_%d" % k]),
                                              _ptr(self.views["fcin_v_%d" % k]) if k > 0 else None, B, F, H,
                                              D, L, self.fast_d, out_nh, 1 if k == 0 else 0, _ptr(out_h),
                                              _ptr(ws["pooled"]), final_len, off, st)
                    )
                elif ("img_f_%d" % k) in ws:
                    _lib.check(
                        lib.xdfm_cin_fwd_packed(_ptr(x0), _ptr(xk), _ptr(ws["img_f_%d" % k]), B, F, H, D,
                                                L, out_nh, 1 if k == 0 else 0, _ptr(out_h), _ptr(ws["pooled"]),
                                                final_len, off, st)
                    )
                else:
                    _lib.check(
                        lib.xdfm_cin_fwd(_ptr(x0), _ptr(xk), _ptr(self.views["cin_f_%d" % k]), B, F, H, D,
                                         L, out_nh, 1 if k == 0 else 0, _ptr(out_h), _ptr(ws["pooled"]),
                                         final_len, off, _ptr(ws["part"]), st)
                    )
                if self.cross_path:
                    act_in = ws["cur_%d" % k]
                    if self.use_bn:
                        act_in = ws["cbn_%d" % k]
                        self._bn_fwd(ws["cur_%d" % k], act_in, "cin", k, B, L, D, 0)
                    if self.cross_soft:  # over the channels of every (sample, d)
                        _lib.check(lib.xdfm_softmax_fwd(_ptr(act_in), B, L, D, _ptr(act_in), st))
                    _lib.check(
                        lib.xdfm_cin_act_fwd(_ptr(act_in), B, L, D, nh, self.cross_act, _ptr(xnext),
                                             _ptr(ws["pooled"]), final_len, off, st)
                    )
                off += L - nh
                xk = xnext
            if not fused_head:
                _lib.check(
                    lib.xdfm_cin_head_fwd(_ptr(ws["pooled"]), _ptr(self.views["cin_w_out"]),
                                          _ptr(self.views["cin_b_out"]), B, final_len,
                                          _ptr(ws["cin_out"]), st)
                )
            self._done(ev)
        if dnn_done is not None:
            if callable(dnn_done):
                dnn_done = dnn_done()
            torch.cuda.current_stream(self.device).wait_event(dnn_done)
        elif hp.use_DNN_part:
            self._dnn_forward(B)
        ev = self._mark("logit_loss")
        if fused_head:
            _lib.check(
                lib.xdfm_head_loss(
                    _ptr(ws["lin_out"]) if use_lin else None, _ptr(ws["fm_out"]) if use_fm else None,
                    _ptr(ws["dnn_out"]) if hp.use_DNN_part else None,
                    _ptr(self.views["lin_b"]) if use_lin else None,
                    _ptr(ws["pooled"]), _ptr(self.views["cin_w_out"]), _ptr(self.views["cin_b_out"]),
                    ws["pooled"].shape[1], _ptr(batch.labels), B, int(self.mean_count or 0),
                    1 if hp.method == "classification" else 0, _LOSS[hp.loss], _ptr(ws["logit"]),
                    _ptr(ws["pred"]), _ptr(ws["cin_out"]), _ptr(ws["g_logit"]), _ptr(ws["d_pooled"]),
                    None if self.overlap_dnn else _ptr(self.data_loss), _ptr(self.loss_scratch), st,
                )
            )
            # with a side stream the mean of the loss is finished there (backward()): no kernel
            # of the step waits for that value
            self._loss_deferred = self.overlap_dnn
            self._loss_partials = 0
            self._done(ev)
            return ws["pred"][:B]
        _lib.check(
            lib.xdfm_logit_loss(
                _ptr(ws["lin_out"]) if use_lin else None, _ptr(ws["fm_out"]) if use_fm else None,
                _ptr(ws["cin_out"]) if hp.use_CIN_part else None,
                _ptr(ws["dnn_out"]) if hp.use_DNN_part else None,
                _ptr(self.views["lin_b"]) if use_lin else None,
                _ptr(batch.labels) if with_loss else None, B, int(self.mean_count or 0),
                1 if hp.method == "classification" else 0, _LOSS[hp.loss], _ptr(ws["logit"]),
                _ptr(ws["pred"]), _ptr(ws["g_logit"]) if with_grad else None,
                _ptr(self.data_loss) if with_loss else None, _ptr(self.loss_scratch), st,
            )
        )
        self._done(ev)
        return ws["pred"][:B]

    def _bn_fwd(self, x, y, where, k, B, C, inner, act):
        _lib.check(
            self.lib.xdfm_bn_fwd(_ptr(x), B, C, inner, _ptr(self.views["bn_%s_g_%d" % (where, k)]),
                                 _ptr(self.views["bn_%s_o_%d" % (where, k)]),
                                 _ptr(self.bn_state["bnstat_%s_mm_%d" % (where, k)]),
                                 _ptr(self.bn_state["bnstat_%s_mv_%d" % (where, k)]), BN_MOMENTUM, BN_EPS,
                                 1 if self._training else 0, act, _ptr(self.ws["bn_stats_%s_%d" % (where, k)]),
                                 _ptr(y), self._stream())
        )

    def _bn_bwd(self, x, y, dy, where, k, B, C, inner, act):
        """dy <- gradient w.r.t. the batch norm's input (in place); gamma / beta gradients."""
        _lib.check(
            self.lib.xdfm_bn_bwd(_ptr(x), _ptr(y), _ptr(dy), B, C, inner,
                                 _ptr(self.views["bn_%s_g_%d" % (where, k)]),
                                 _ptr(self.ws["bn_stats_%s_%d" % (where, k)]), act, _ptr(dy),
                                 _ptr(self.grad_views["bn_%s_g_%d" % (where, k)]),
                                 _ptr(self.grad_views["bn_%s_o_%d" % (where, k)]), self._stream())
        )

    def _pack_filters(self, B, with_grad):
        """One launch packs the filter images of every CIN layer for this step."""
        ws = self.ws
        shapes = cin_shapes(self.hp)
        if self.bf16:
            for k, (H, L, nh) in enumerate(shapes):
                masked = 1 if k == 0 else 0
                if k in self._split:
                    _lib.check(self.lib.xdfm_cin_pool_pack(_ptr(self.views["cin_f_%d" % k]), self.F, H, L, nh, masked,
                                                           _ptr(ws["wp_%d" % k]), self._stream()))
                if nh > 0:
                    _lib.check(self.lib.xdfm_cin_bf16_pack(_ptr(self.views["cin_f_%d" % k]), self.F, H, nh, L, masked,
                                                           _ptr(ws["bimg_f_%d" % k]),
                                                           _ptr(ws["bimg_d_%d" % k]) if with_grad else None,
                                                           self._stream()))
            return
        jobs = (_lib.CinPackJob * len(shapes))()
        n = 0
        for k, (H, L, nh) in enumerate(shapes):
            if k in self._split:  # the pooled channels' filter slice in pair order
                _lib.check(self.lib.xdfm_cin_pool_pack(_ptr(self.views["cin_f_%d" % k]), self.F, H, L, nh,
                                                       1 if k == 0 else 0, _ptr(ws["wp_%d" % k]), self._stream()))
            f_img = ws.get("img_f_%d" % k)
            b_img = ws.get("img_b_%d" % k) if with_grad else None
            if f_img is None and b_img is None:
                continue
            Lk = nh if k in self._split else L  # a split layer's kernels carry its hidden channels only
            jobs[n] = _lib.CinPackJob(self.views["cin_f_%d" % k].data_ptr(), self.F, H, Lk, 1 if k == 0 else 0,
                                      f_img.data_ptr() if f_img is not None else None,
                                      b_img.data_ptr() if b_img is not None else None, L, 0)
            n += 1
        for s0 in range(0, n, 8):
            chunk = ctypes.cast(ctypes.byref(jobs, s0 * ctypes.sizeof(_lib.CinPackJob)), ctypes.c_void_p)
            _lib.check(self.lib.xdfm_cin_pack(chunk, min(8, n - s0), B, self.D, self._stream()))

    def _branch_stream(self):
        """The DNN tower's stream, made to wait for everything enqueued so far on the current one."""
        if self._dnn_stream is None:
            self._dnn_stream = torch.cuda.Stream(device=self.device)
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(self.device))
        self._dnn_stream.wait_event(ready)
        return self._dnn_stream

    def _dnn_forward(self, B):
        """DNN tower (xDeepFM.py:473-533) on the current stream."""
        hp, lib, ws, st = self.hp, self.lib, self.ws, self._stream()
        ev = self._mark("dnn_fwd")
        if self._tower_fused():
            _lib.check(lib.xdfm_tower_fwd(ctypes.byref(self._tower_desc()), _ptr(ws["emb_out"]), B,
                                          _ptr(ws["dnn_out"]), st))
            self._done(ev)
            return
        x, K = ws["emb_out"], self.F * self.D
        for k, size in enumerate(hp.layer_sizes):
            act = _ACT[hp.activation[k]]
            soft = act == _SOFTMAX
            if soft:
                act = 0
            if self.bf16 and K >= 32 and int(size) >= 32:
                _lib.check(
                    lib.xdfm_dense_bf16_fwd(_ptr(x), _ptr(self.views["dnn_w_%d" % k]), _ptr(self.views["dnn_b_%d" % k]),
                                            B, K, int(size), act, _ptr(ws["h_%d" % k]), _ptr(ws["bf16_ws"]), st)
                )
                x, K = ws["h_%d" % k], int(size)
                continue
            _lib.check(
                lib.xdfm_dense_fwd(_ptr(x), _ptr(self.views["dnn_w_%d" % k]),
                                   _ptr(self.views["dnn_b_%d" % k]), B, K, int(size),
                                   0 if (self._dropout_live or self.use_bn) else act,
                                   _ptr(ws["z_%d" % k] if self.use_bn else ws["h_%d" % k]), st)
            )
            if self.use_bn:  # xw_plus_b -> batch norm -> (dropout ->) activation
                self._bn_fwd(ws["z_%d" % k], ws["h_%d" % k], "dnn", k, B, int(size), 1,
                             0 if self._dropout_live else act)
            if self._dropout_live:
                _lib.check(
                    lib.xdfm_dropout_act_fwd(_ptr(ws["h_%d" % k]), _ptr(ws["drop_%d" % k]), B, int(size),
                                             float(1.0 - float(hp.dropout[k])), act,
                                             (self.dropout_seed + 0x9E3779B97F4A7C15 * (k + 1)) & (2 ** 64 - 1),
                                             _ptr(self.opt_state[3:]), st)
                )
            if soft:
                _lib.check(lib.xdfm_softmax_fwd(_ptr(ws["h_%d" % k]), B, int(size), 1, _ptr(ws["h_%d" % k]), st))
            x, K = ws["h_%d" % k], int(size)
        _lib.check(
            lib.xdfm_dense_fwd(_ptr(x), _ptr(self.views["dnn_w_out"]), _ptr(self.views["dnn_b_out"]),
                               B, K, 1, 0, _ptr(ws["dnn_out"]), st)
        )
        self._done(ev)

    def _tower_fused(self):
        """The tower runs as one launch per pass (xdfm_tower_fwd / xdfm_tower_bwd_dx) and the step's small
        weight gradients as one multi-job launch: fp32, no batch norm / dropout / softmax, 1..4 hidden layers,
        the first at most 32 wide, the others at most 64."""
        hp = self.hp
        if not hp.use_DNN_part or self.bf16 or os.environ.get("XDFM_TOWER_FUSED", "1") == "0":
            return False
        sizes = [int(v) for v in hp.layer_sizes]
        return bool(1 <= len(sizes) <= 4 and sizes[0] <= 32 and max(sizes) <= 64 and not self.use_dropout
                    and not self.use_bn and all(_ACT[a] != _SOFTMAX for a in hp.activation[:len(sizes)]))

    def _tower_desc(self):
        """struct xdfm_tower_desc over the current workspace (cached until it is re-allocated)."""
        hit = getattr(self, "_tower_cache", None)
        if hit is not None and hit[0] == self._ws_token:
            return hit[1]
        hp, ws = self.hp, self.ws
        d = _lib.TowerDesc()
        d.n_hidden, d.in_dim = len(hp.layer_sizes), self.F * self.D
        for k, size in enumerate(hp.layer_sizes):
            d.size[k], d.act[k] = int(size), _ACT[hp.activation[k]]
            d.w[k], d.b[k] = self.views["dnn_w_%d" % k].data_ptr(), self.views["dnn_b_%d" % k].data_ptr()
            d.h[k], d.d_h[k] = ws["h_%d" % k].data_ptr(), ws["d_h_%d" % k].data_ptr()
        d.w_out, d.b_out = self.views["dnn_w_out"].data_ptr(), self.views["dnn_b_out"].data_ptr()
        self._tower_cache = (self._ws_token, d)
        return d

    def _dw_jobs(self, ws, B, sizes_only=False, head=True):
        """xdfm_dw_job array: the tower's layers, then (head) the linear bias and the CIN head."""
        hp = self.hp
        sizes = [self.F * self.D] + [int(v) for v in hp.layer_sizes]
        n_layers = len(hp.layer_sizes)
        head = bool(head and hp.use_Linear_part and hp.use_CIN_part)
        n = n_layers + 1 + (2 if head else 0)
        jobs = (_lib.DwJob * n)()
        final_len = ws["pooled"].shape[1] if hp.use_CIN_part else 1
        dims = [(sizes[k], sizes[k + 1], _ACT[hp.activation[k]]) for k in range(n_layers)] + [(sizes[-1], 1, 0)]
        if head:
            dims += [(1, 1, 0), (final_len, 1, 0)]
        for q, (K, Nout, act) in enumerate(dims):
            jobs[q].K, jobs[q].Nout, jobs[q].act = K, Nout, act
        if sizes_only:
            return jobs
        gv, g = self.grad_views, ws["g_logit"]
        acts = [ws["emb_out"]] + [ws["h_%d" % k] for k in range(n_layers)]
        for k in range(n_layers):
            J = jobs[k]
            J.x, J.y, J.dy = acts[k].data_ptr(), acts[k + 1].data_ptr(), ws["d_h_%d" % k].data_ptr()
            J.dw, J.db = gv["dnn_w_%d" % k].data_ptr(), gv["dnn_b_%d" % k].data_ptr()
        J = jobs[n_layers]
        J.x, J.y, J.dy = acts[-1].data_ptr(), ws["dnn_out"].data_ptr(), g.data_ptr()
        J.dw, J.db = gv["dnn_w_out"].data_ptr(), gv["dnn_b_out"].data_ptr()
        if head:
            J = jobs[n_layers + 1]  # d lin_b = sum_b g: the bias row of a [1] -> [1] identity layer
            J.x = J.y = J.dy = g.data_ptr()
            J.dw, J.db = ws["dummy"].data_ptr(), gv["lin_b"].data_ptr()
            J = jobs[n_layers + 2]  # the CIN head as [n] -> [1] with dy = g
            J.x, J.y, J.dy = ws["pooled"].data_ptr(), g.data_ptr(), g.data_ptr()
            J.dw, J.db = gv["cin_w_out"].data_ptr(), gv["cin_b_out"].data_ptr()
        return jobs

    def _dnn_plain(self):
        """No dropout, batch norm or softmax in the tower: its backward splits into an input-gradient
        chain and the weight gradients."""
        hp = self.hp
        return not (self._dropout_live or self.use_bn or any(_ACT[hp.activation[k]] == _SOFTMAX
                                                              for k in range(len(hp.layer_sizes))))

    def _dnn_backward(self, B, first_layer_dx, phase="all", head_jobs=False):
        """Backward of the DNN tower on the current stream.  first_layer_dx: (buffer, accumulate)
        for the gradient w.r.t. emb_out, None to skip it.  phase (plain towers only): "dx" = the
        chain of input gradients down to emb_out alone, "dw" = the weight / bias gradients alone
        (after "dx": they read the d_h the chain left).  head_jobs: the linear bias and the CIN head's
        weight gradients ride in the same multi-job launch (fused towers; sets self._head_jobs_done)."""
        self._head_jobs_done = False
        hp, lib, ws, st = self.hp, self.lib, self.ws, self._stream()
        g, gv = ws["g_logit"], self.grad_views
        ev = self._mark("dnn_bwd")
        n_layers = len(hp.layer_sizes)
        sizes = [self.F * self.D] + [int(s) for s in hp.layer_sizes]
        acts = [ws["emb_out"]] + [ws["h_%d" % k] for k in range(n_layers)]
        d_acts = [None] + [ws["d_h_%d" % k] for k in range(n_layers)]
        want_dx, want_dw = phase != "dw", phase != "dx"
        if self._tower_fused() and (first_layer_dx is not None or phase == "dw"):
            if want_dx:
                dx, acc = first_layer_dx
                _lib.check(lib.xdfm_tower_bwd_dx(ctypes.byref(self._tower_desc()), _ptr(g), B, _ptr(dx), int(acc), st))
            if want_dw:
                jobs = self._dw_jobs(ws, B, head=head_jobs)
                _lib.check(lib.xdfm_dense_bwd_dw_multi(jobs, len(jobs), B, _ptr(ws["part_multi"]), st))
                self._head_jobs_done = len(jobs) > len(hp.layer_sizes) + 1
            self._done(ev)
            return
        if phase != "all":
            x_in, K_in = acts[-1], sizes[-1]
            layers = [(x_in, self.views["dnn_w_out"], ws["dnn_out"], g, K_in, 1, 0, d_acts[-1], 0,
                       gv["dnn_w_out"], gv["dnn_b_out"])]
            for k in range(n_layers - 1, -1, -1):
                dx, acc = (first_layer_dx if first_layer_dx is not None else (None, 0)) if k == 0 else (d_acts[k], 0)
                layers.append((acts[k], self.views["dnn_w_%d" % k], acts[k + 1], d_acts[k + 1], sizes[k], sizes[k + 1],
                               _ACT[hp.activation[k]], dx, acc, gv["dnn_w_%d" % k], gv["dnn_b_%d" % k]))
            for (x, w, y, dy, K, Nout, act, dx, acc, dw, db) in layers:
                if want_dx and dx is None:
                    continue
                _lib.check(
                    lib.xdfm_dense_bwd(_ptr(x), _ptr(w), _ptr(y), _ptr(dy), B, K, Nout, act,
                                       _ptr(dx) if want_dx else None, acc, _ptr(dw) if want_dw else None,
                                       _ptr(db) if want_dw else None, _ptr(ws["part_dnn"]) if want_dw else None, st)
                )
            self._done(ev)
            return
        _lib.check(
            lib.xdfm_dense_bwd(_ptr(acts[-1]), _ptr(self.views["dnn_w_out"]), _ptr(ws["dnn_out"]),
                               _ptr(g), B, sizes[-1], 1, 0, _ptr(d_acts[-1]), 0, _ptr(gv["dnn_w_out"]),
                               _ptr(gv["dnn_b_out"]), _ptr(ws["part_dnn"]), st)
        )
        for k in range(n_layers - 1, -1, -1):
            if k == 0:
                dx, acc = first_layer_dx if first_layer_dx is not None else (None, 0)
            else:
                dx, acc = d_acts[k], 0
            act = _ACT[hp.activation[k]]
            if act == _SOFTMAX:
                _lib.check(lib.xdfm_softmax_bwd(_ptr(acts[k + 1]), _ptr(d_acts[k + 1]), B, sizes[k + 1], 1,
                                                _ptr(d_acts[k + 1]), st))
                act = 0
            if self._dropout_live:
                # dz = dy * act'(y) * mask/keep in place; the layer itself is then linear
                _lib.check(
                    lib.xdfm_dropout_act_bwd(_ptr(acts[k + 1]), _ptr(d_acts[k + 1]), _ptr(ws["drop_%d" % k]), B,
                                             sizes[k + 1], act, _ptr(d_acts[k + 1]), st)
                )
                act = 0
            if self.use_bn:
                self._bn_bwd(ws["z_%d" % k], acts[k + 1], d_acts[k + 1], "dnn", k, B, sizes[k + 1], 1, act)
                act = 0
            if self.bf16 and sizes[k] >= 32 and sizes[k + 1] >= 32:
                _lib.check(
                    lib.xdfm_dense_bf16_bwd(_ptr(acts[k]), _ptr(self.views["dnn_w_%d" % k]), _ptr(acts[k + 1]),
                                            _ptr(d_acts[k + 1]), B, sizes[k], sizes[k + 1], act, _ptr(dx), acc,
                                            _ptr(gv["dnn_w_%d" % k]), _ptr(gv["dnn_b_%d" % k]),
                                            _ptr(ws["bf16_ws"]), st)
                )
                continue
            _lib.check(
                lib.xdfm_dense_bwd(_ptr(acts[k]), _ptr(self.views["dnn_w_%d" % k]), _ptr(acts[k + 1]),
                                   _ptr(d_acts[k + 1]), B, sizes[k], sizes[k + 1], act, _ptr(dx), acc,
                                   _ptr(gv["dnn_w_%d" % k]), _ptr(gv["dnn_b_%d" % k]),
                                   _ptr(ws["part_dnn"]), st)
            )
        self._done(ev)

    # ---------------------------------------------------------------- backward
    def backward(self, batch, advance=False, defer_late=False):
        """Gradients of the data loss w.r.t. every dense parameter (into dense_grad) and
        w.r.t. emb_out (into ws['d_emb'] [+ d_emb2()]).  forward(with_grad=True) must have run.
        advance: also move the optimizer's step scalars on (beside the backward pass when the DNN
        tower has its own stream); returns True when that was done.
        defer_late: the caller runs run_late_work() once it has issued the table gradient (train_step)."""
        hp, lib, ws, st = self.hp, self.lib, self.ws, self._stream()
        B, F, D = batch.B, self.F, self.D
        g = ws["g_logit"]
        d_emb = ws["d_emb"]
        gv = self.grad_views
        wrote_d_emb = False
        fused_head = bool(hp.use_CIN_part and getattr(self, "_head_fused", False))

        def small_grads(part, stream):
            """lin_b and CIN head weight gradients: read g and pooled only."""
            if hp.use_Linear_part:
                # d lin_b = sum_b g  -- the bias row of a [1]->[1] identity layer
                _lib.check(
                    lib.xdfm_dense_bwd(_ptr(g), _ptr(self.views["lin_b"]), _ptr(g), _ptr(g), B, 1, 1, 0, None,
                                       0, _ptr(ws["dummy"]), _ptr(gv["lin_b"]), _ptr(part), stream)
                )
            if hp.use_CIN_part:
                _lib.check(
                    lib.xdfm_cin_head_bwd(_ptr(ws["pooled"]), _ptr(self.views["cin_w_out"]), _ptr(g), B,
                                          ws["pooled"].shape[1], None if fused_head else _ptr(ws["d_pooled"]),
                                          _ptr(gv["cin_w_out"]), _ptr(gv["cin_b_out"]), _ptr(part), stream)
                )

        dnn_done = None
        if self.overlap_dnn:
            # fork: everything of the DNN tower but its gradient w.r.t. emb_out runs beside the CIN,
            # and so do the small weight gradients that hang off g_logit
            if self._dnn_stream is None:
                self._dnn_stream = torch.cuda.Stream(device=self.device)
            side = self._dnn_stream
            head_ready = torch.cuda.Event()
            head_ready.record(torch.cuda.current_stream(self.device))
            # What the chain of large kernels waits for comes first on that stream and nothing else does: the
            # optimizer's scalars (the table update reads them) and the tower's gradient w.r.t. emb_out (the
            # first CIN layer's input-gradient kernel adds it in).  Everything else that hangs off g_logit --
            # the loss, the small gradients, the tower's weight gradients: a dozen launches of ~5 us that
            # wait for room beside kernels which fill the chip (tools/stamp_timeline.py: 100-150 us for the
            # lot) -- goes to a second stream that only the dense update joins (join_late()).
            third = bool(self._fold_dx and self.side_split)
            side_state = {"done": None, "dx_done": None, "forked": False}

            def head_work():
                if getattr(self, "_loss_deferred", False):
                    self._finish_loss(B, self._stream())
                    self._loss_deferred = False
                if fused_head:
                    small_grads(ws["part_dnn"], self._stream())

            def multi_dw():
                self._dnn_backward(B, None, phase="dw", head_jobs=True)
                if not self._head_jobs_done:
                    small_grads(ws["part_dnn"], self._stream())

            def tail_work():
                """The loss, the small gradients and the tower's weight gradients (current stream): what only
                the dense update waits for."""
                if fused_head and self._tower_fused():
                    if getattr(self, "_loss_deferred", False):
                        self._finish_loss(B, self._stream())
                        self._loss_deferred = False
                    if self.dw_multi_tail and defer_late:
                        # the small weight gradients as one launch on the chain's own stream behind
                        # the table gradient, where that stream has slack (run_late_work())
                        self._late_multi = multi_dw
                    else:
                        multi_dw()
                else:
                    head_work()
                    self._dnn_backward(B, None, phase="dw")

            def fork_side():
                # (with main_first this runs behind the first CIN kernel of the backward pass: see forward())
                if side_state["forked"]:
                    return
                side_state["forked"] = True
                side.wait_event(head_ready)
                with torch.cuda.stream(side):
                    if third:
                        if advance:
                            self._advance_step()
                    elif not self._fold_dx:
                        head_work()
                    if self._fold_dx:
                        # (one side stream: still the chain the CIN waits for first, the rest behind it)
                        self._dnn_backward(B, (ws["d_emb_dnn"], 0), phase="dx")
                        side_state["dx_done"] = torch.cuda.Event()
                        side_state["dx_done"].record(side)
                        if not third:
                            # the chain joins here; the loss, the small gradients and the tower's weight
                            # gradients follow on this stream and only the dense update waits for them
                            if advance:
                                self._advance_step()
                            side_state["done"] = torch.cuda.Event()
                            side_state["done"].record(side)
                            # (launch by launch: 0.3695 ms per step against 0.376 with the one-launch form
                            # here and 0.392 with it behind the table gradient)
                            head_work()
                            self._dnn_backward(B, None, phase="dw")
                            self._late_event2 = torch.cuda.Event()
                            self._late_event2.record(side)
                    else:
                        self._dnn_backward(B, (ws["d_emb_dnn"], 0))
                        if advance:
                            self._advance_step()  # nothing reads the scalars before the join
                    if side_state["done"] is None:
                        side_state["done"] = torch.cuda.Event()
                        side_state["done"].record(side)
                if third:
                    if self._dnn_stream2 is None:
                        self._dnn_stream2 = torch.cuda.Stream(device=self.device)
                    side2 = self._dnn_stream2
                    side2.wait_event(side_state["dx_done"])  # (the weight gradients read the d_h the chain left)
                    with torch.cuda.stream(side2):
                        tail_work()
                        self._late_event2 = torch.cuda.Event()
                        self._late_event2.record(side2)

            if not (self.main_first and hp.use_CIN_part):
                fork_side()
            dnn_done = True
            st = self._stream()
            if not fused_head:
                small_grads(ws["part"], st)
                if not self._fold_dx:
                    d_emb[:B].zero_()
            wrote_d_emb = True  # zeroed beside the forward pass (or just above), or written by the CIN
        else:
            small_grads(ws["part"], st)
            if hp.use_DNN_part:
                self._dnn_backward(B, (d_emb, 0))
                wrote_d_emb = True
        if not wrote_d_emb:
            d_emb[:B].zero_()
        if hp.use_CIN_part:
            ev = self._mark("cin_bwd")
            shapes = cin_shapes(hp)
            final_len = ws["pooled"].shape[1]
            offs, off = [], 0
            for (H, L, nh) in shapes:
                offs.append(off)
                off += L - nh
            x0 = ws["emb_out"]
            for k in range(len(shapes) - 1, -1, -1):
                H, L, nh = shapes[k]
                xk = x0 if k == 0 else ws["hid_%d" % (k - 1)]
                d_xnext = ws["d_hid_%d" % k] if nh > 0 else None
                d_xk = None if k == 0 else ws["d_hid_%d" % (k - 1)]
                d_pool, up_nh = ws["d_pooled"], nh
                if self.cross_path:
                    _lib.check(
                        lib.xdfm_cin_act_bwd(_ptr(ws["cbn_%d" % k] if self.use_bn else ws["cur_%d" % k]),
                                             _ptr(d_xnext), _ptr(ws["d_pooled"]),
                                             final_len, offs[k], B, L, D, nh, self.cross_act,
                                             _ptr(ws["d_cur"]), st)
                    )
                    if self.cross_soft:
                        _lib.check(
                            lib.xdfm_softmax_bwd(_ptr(ws["cbn_%d" % k] if self.use_bn else ws["cur_%d" % k]),
                                                 _ptr(ws["d_cur"]), B, L, D, _ptr(ws["d_cur"]), st)
                        )
                    if self.use_bn:
                        self._bn_bwd(ws["cur_%d" % k], ws["cbn_%d" % k], ws["d_cur"], "cin", k, B, L, D, 0)
                    d_xnext, d_pool, up_nh = ws["d_cur"], None, L
                if self.fast_d:
                    _lib.check(
                        lib.xdfm_fast_cin_bwd(_ptr(x0), _ptr(xk), _ptr(self.views["fcin_w_%d" % k]),
                                              _ptr(self.views["fcin_v_%d" % k]) if k > 0 else None, B, F, H,
                                              D, L, self.fast_d, up_nh, 1 if k == 0 else 0, _ptr(d_xnext),
                                              _ptr(d_pool), final_len, offs[k], _ptr(d_xk),
                                              _ptr(d_emb), _ptr(gv["fcin_w_%d" % k]),
                                              _ptr(gv["fcin_v_%d" % k]) if k > 0 else None,
                                              _ptr(ws["part"]), st)
                    )
                    continue
                if self.bf16:
                    masked = 1 if k == 0 else 0
                    if k in self._split:
                        P, Lp = self._split[k]
                        ws["pool_%d" % k][:B].copy_(ws["d_pooled"][:B, offs[k]:offs[k] + Lp])
                        _lib.check(
                            lib.xdfm_dense_bf16_bwd(_ptr(ws["S_%d" % k]), _ptr(ws["wp_%d" % k]), _ptr(ws["pool_%d" % k]),
                                                    _ptr(ws["pool_%d" % k]), B, P, Lp, 0, _ptr(ws["dS"]), 0,
                                                    _ptr(ws["dwp_%d" % k]), _ptr(ws["db_pool"]), _ptr(ws["bf16_ws"]), st)
                        )
                        _lib.check(lib.xdfm_cin_pool_unpack(_ptr(ws["dwp_%d" % k]), F, H, L, nh, masked,
                                                            _ptr(gv["cin_f_%d" % k]), st))
                    if nh > 0:
                        _lib.check(lib.xdfm_cin_bf16_bwd_dx(_ptr(x0), _ptr(xk), _ptr(ws["bimg_d_%d" % k]), _ptr(d_xnext),
                                                            B, F, H, D, nh, masked, _ptr(d_xk), _ptr(d_emb), 1, st))
                        _lib.check(lib.xdfm_cin_bf16_bwd_dw(_ptr(x0), _ptr(xk), _ptr(d_xnext), B, F, H, D, nh, masked,
                                                            _ptr(gv["cin_f_%d" % k]), L, _ptr(ws["bf16_ws"]), st))
                    if k in self._split:
                        pair_bwd = lib.xdfm_cin_pair_bwd_bf16 if self.pair_bf16 else lib.xdfm_cin_pair_bwd
                        _lib.check(pair_bwd(_ptr(x0), _ptr(xk), _ptr(ws["dS"]), B, F, H, D, masked,
                                            _ptr(d_emb), 1, _ptr(d_xk), 1 if nh > 0 else 0, st))
                    continue
                fold = self.overlap_dnn and self._fold_dx
                first_writer = fold and k == len(shapes) - 1  # nothing has been put into d_emb yet
                if k in self._fused_pool:
                    _lib.check(
                        lib.xdfm_cin_pooled_bwd(_ptr(x0), _ptr(xk), _ptr(self.views["cin_f_%d" % k]),
                                                _ptr(ws["d_pooled"]), final_len, offs[k], B, F, H, D, L, 0,
                                                _ptr(d_xk), _ptr(d_emb), 0 if first_writer else 1,
                                                _ptr(gv["cin_f_%d" % k]), _ptr(ws["part"]), st)
                    )
                    if dnn_done is not None:
                        fork_side()
                    continue
                Lk, pair_after = L, False
                if k in self._split:
                    # pooled channels: dS = d_pooled Wp^T and dWp = S^T d_pooled as one dense-layer
                    # backward, then the per-sample products with dS
                    P, Lp = self._split[k]
                    ws["pool_%d" % k][:B].copy_(ws["d_pooled"][:B, offs[k]:offs[k] + Lp])
                    _lib.check(
                        lib.xdfm_dense_bwd(_ptr(ws["S_%d" % k]), _ptr(ws["wp_%d" % k]), _ptr(ws["pool_%d" % k]),
                                           _ptr(ws["pool_%d" % k]), B, P, Lp, 0, _ptr(ws["dS"]), 0,
                                           _ptr(ws["dwp_%d" % k]), _ptr(ws["db_pool"]), _ptr(ws["part_pool"]), st)
                    )
                    _lib.check(lib.xdfm_cin_pool_unpack(_ptr(ws["dwp_%d" % k]), F, H, L, nh, 1 if k == 0 else 0,
                                                        _ptr(gv["cin_f_%d" % k]), st))
                    if nh == 0:  # no hidden channel: the layer is done with the pair products
                        _lib.check(lib.xdfm_cin_pair_bwd(_ptr(x0), _ptr(xk), _ptr(ws["dS"]), B, F, H, D,
                                                         1 if k == 0 else 0, _ptr(d_emb), 0 if first_writer else 1,
                                                         _ptr(d_xk), 0, st))
                        continue
                    Lk, up_nh, d_pool, pair_after = nh, nh, None, True
                if ("img_b_%d" % k) in ws:
                    add0 = None if first_writer else d_emb
                    add1 = ws["d_emb_dnn"] if (fold and k == 0) else None
                    if add1 is not None:
                        fork_side()
                        torch.cuda.current_stream(self.device).wait_event(side_state["dx_done"])

                    def layer_bwd(phases, stream):
                        _lib.check(
                            lib.xdfm_cin_bwd_packed(_ptr(x0), _ptr(xk), _ptr(ws["img_b_%d" % k]), B, F, H, D, Lk,
                                                    up_nh, 1 if k == 0 else 0, _ptr(d_xnext), _ptr(d_pool),
                                                    final_len, offs[k], _ptr(d_xk), _ptr(d_emb),
                                                    _ptr(gv["cin_f_%d" % k]), _ptr(ws["part"]), phases,
                                                    _ptr(add0), _ptr(add1), L if Lk != L else 0, stream)
                        )

                    def pair_bwd():
                        if pair_after:  # behind the hidden channels' kernel, which wrote d_xk
                            _lib.check(lib.xdfm_cin_pair_bwd(_ptr(x0), _ptr(xk), _ptr(ws["dS"]), B, F, H, D,
                                                             1 if k == 0 else 0, _ptr(d_emb), 1, _ptr(d_xk), 1, st))

                    side_dw = dnn_done is not None and self.split_dw
                    if k == 1 and side_dw and "img_b_0" in ws:
                        # the reduction of this layer's filter-gradient partials is a short kernel
                        # that only the optimizer waits for: the side stream runs it (and is the
                        # next writer of the partials' buffer, with the first layer's gradient)
                        layer_bwd(1 | 4, st)
                        pair_bwd()
                        part_done = torch.cuda.Event()
                        part_done.record(torch.cuda.current_stream(self.device))
                        side.wait_event(part_done)
                        with torch.cuda.stream(side):
                            layer_bwd(8, self._stream())
                    elif k == 0 and side_dw:
                        # the last layer's filter gradient (matrix-core bound) leaves the critical
                        # path: it runs on the side stream beside the table gradient / update
                        # (latency bound), which only needs d_emb
                        layer_bwd(1, st)
                        pair_bwd()
                        fork_side()
                        dx_done = torch.cuda.Event()
                        dx_done.record(torch.cuda.current_stream(self.device))

                        def late_dw(layer_bwd=layer_bwd, dx_done=dx_done):
                            side.wait_event(dx_done)
                            with torch.cuda.stream(side):
                                layer_bwd(2, self._stream())
                                self._late_event = torch.cuda.Event()
                                self._late_event.record(side)

                        # main_first: issued behind the table gradient (run_late_work(), from the update)
                        if defer_late and self.main_first and self.defer_dw:
                            self._late_work = late_dw
                        else:
                            late_dw()
                    else:
                        layer_bwd(3, st)
                        pair_bwd()
                    continue
                _lib.check(
                    lib.xdfm_cin_bwd(_ptr(x0), _ptr(xk), _ptr(self.views["cin_f_%d" % k]), B, F, H, D, L,
                                     up_nh, 1 if k == 0 else 0, _ptr(d_xnext), _ptr(d_pool),
                                     final_len, offs[k], _ptr(d_xk), _ptr(d_emb),
                                     _ptr(gv["cin_f_%d" % k]), _ptr(ws["part"]), st)
                )
            self._done(ev)
        if dnn_done is not None:
            fork_side()
            torch.cuda.current_stream(self.device).wait_event(side_state["done"])  # join
            return bool(advance)
        return False

    def _finish_loss(self, B, st):
        """The mean of the data loss from the per-block sums the forward pass left in loss_scratch."""
        if self._loss_partials:
            _lib.check(self.lib.xdfm_loss_finish_partials(_ptr(self.loss_scratch), self._loss_partials,
                                                          int(self.mean_count or B), _ptr(self.data_loss), st))
        else:
            _lib.check(self.lib.xdfm_loss_finish(_ptr(self.loss_scratch), B, int(self.mean_count or 0),
                                                 _LOSS[self.hp.loss], _ptr(self.data_loss), st))

    def run_late_work(self):
        """Issue what backward(defer_late=True) held back (the first layer's filter gradient on the side
        stream)."""
        work, self._late_work = self._late_work, None
        if work is not None:
            work()
        work, self._late_multi = self._late_multi, None
        if work is not None:
            work()

    def join_late(self):
        """Make the current stream wait for what backward() left running on the side streams: the first
        layer's filter gradient, the tower's weight gradients and the loss (dense_grad and data_loss are
        complete only after this)."""
        self.run_late_work()
        for name in ("_late_event", "_late_event2"):
            ev = getattr(self, name)
            if ev is not None:
                torch.cuda.current_stream(self.device).wait_event(ev)
                setattr(self, name, None)

    def d_emb2(self):
        """Second summand of the gradient w.r.t. emb_out (None when d_emb holds all of it)."""
        return self.ws["d_emb_dnn"] if (self.overlap_dnn and not self._fold_dx) else None

    def _opt_args(self, l2=0.0, l1=0.0, clip=0.0):
        hp = self.hp
        b1, b2, eps = _OPT_CONST[self.opt_kind]
        return _lib.OptArgs(self.opt_kind, float(hp.learning_rate), b1, b2, eps,
                            float(self.b1p), float(self.b2p), float(l2), float(l1), float(clip), 0,
                            self.opt_state.data_ptr() if (self.opt_kind == 1 or self.catchup_on) else None)

    def _advance_step(self):
        self.step += 1
        # the device-side step count also seeds the dropout masks and dates the rows (catch-up)
        if self.opt_kind == 1 or self.use_dropout or self.catchup_on:
            self.b1p = np.float32(self.b1p * np.float32(0.9))  # host mirror (checkpoints)
            self.b2p = np.float32(self.b2p * np.float32(0.999))
            _lib.check(self.lib.xdfm_opt_advance(_ptr(self.opt_state), float(self.hp.learning_rate), 0.9, 0.999,
                                                 1e-8, _ptr(self.drift_tab), self._stream()))

    def update(self, batch, advanced=False):
        """apply_gradients (base_model.py:305-322) for every variable + regular_loss
        (base_model.py:264-272, value before the update) into self.reg_loss.
        advanced: backward(advance=True) has moved the step scalars on already."""
        if not advanced:
            self._advance_step()
        first = self._update_embedding(batch)
        self._update_dense(first)

    def _update_embedding(self, batch):
        """Table / lin_w step from the sorted-key segment sum.  Returns True when reg_loss has
        not been written yet (the dense update then writes instead of accumulating)."""
        hp, lib, ws, st = self.hp, self.lib, self.ws, self._stream()
        clip = float(hp.max_grad_norm) if hp.is_clip_norm else 0.0
        use_lin, use_fm = bool(hp.use_Linear_part), bool(hp.use_FM_part)
        self.plan(batch)
        if batch.plan_event is not None:
            torch.cuda.current_stream(self.device).wait_event(batch.plan_event)
            batch.plan_event = None
        sparse = self.sparse
        opt_sparse = self._opt_args()
        pre = batch.sorted_row is not None and batch.planned_gather  # tok_row / weights in sorted order
        ev = self._mark("embed_bwd")
        _lib.check(
            lib.xdfm_embed_bwd(
                0 if sparse else 1, _ptr(self.table), _ptr(self.table_m), _ptr(self.table_v),
                _ptr(self.lin_w) if use_lin else None, _ptr(self.lin_m) if use_lin else None,
                _ptr(self.lin_v) if use_lin else None, self.table_stride, self.lin_stride,
                _ptr(self.grad_table),
                _ptr(self.grad_lin) if use_lin else None, self.N, self.D, self.F, batch.B,
                _ptr(batch.sorted_ids), None if pre else _ptr(batch.sorted_tok),
                _ptr(batch.sorted_row if pre else batch.tok_row),
                _ptr(batch.sorted_w if pre else batch.weights), batch.nnz, _ptr(ws["d_emb"]),
                _ptr(ws["g_logit"]) if (use_lin or use_fm) else None,
                _ptr(ws["fm_sum"]) if use_fm else None, 1 if use_fm else 0,
                ctypes.byref(opt_sparse), _ptr(ws["pieces"]), _ptr(self.d_emb2()), 0,
                self._age(-1) if sparse else None, st,
            )
        )
        self._done(ev)
        self.run_late_work()  # (behind the table gradient in issue order: backward(defer_late=True))
        if sparse and self.catchup_on:
            self._rows_stale = True
        first = True  # the first dense_update writes reg_loss, the others accumulate
        if not sparse:
            one = np.asarray([0, self.N * self.D], dtype=np.int64)
            opt = self._opt_args(clip=clip)
            _lib.check(
                lib.xdfm_dense_update(
                    _ptr(self.table), _ptr(self.grad_table), _ptr(self.table_m), _ptr(self.table_v), 1,
                    one.ctypes.data_as(ctypes.c_void_p),
                    np.asarray([hp.embed_l2], dtype=np.float32).ctypes.data_as(ctypes.c_void_p),
                    np.asarray([hp.embed_l1], dtype=np.float32).ctypes.data_as(ctypes.c_void_p),
                    None, ctypes.byref(opt), _ptr(self.stats), _ptr(self.reg_loss), 0, st,
                )
            )
            first = False
            if use_lin:
                one = np.asarray([0, self.N], dtype=np.int64)
                _lib.check(
                    lib.xdfm_dense_update(
                        _ptr(self.lin_w), _ptr(self.grad_lin), _ptr(self.lin_m), _ptr(self.lin_v), 1,
                        one.ctypes.data_as(ctypes.c_void_p),
                        np.asarray([hp.layer_l2], dtype=np.float32).ctypes.data_as(ctypes.c_void_p),
                        np.asarray([hp.layer_l1], dtype=np.float32).ctypes.data_as(ctypes.c_void_p),
                        None, ctypes.byref(opt), _ptr(self.stats), _ptr(self.reg_loss), 1, st,
                    )
                )
            _lib.check(
                lib.xdfm_embed_clear_rows(_ptr(self.grad_table), _ptr(self.grad_lin) if use_lin else None,
                                          self.D, _ptr(batch.sorted_ids), batch.nnz, st)
            )
        return first

    def _update_dense(self, first=True):
        """Optimizer step of the dense (replicated) parameters from self.dense_grad."""
        self.join_late()
        hp, lib, st = self.hp, self.lib, self._stream()
        clip = float(hp.max_grad_norm) if hp.is_clip_norm else 0.0
        ev_upd = self._mark("dense_update")
        n_seg = len(self.dense_shapes)
        if n_seg:
            seg = self._seg
            opt = self._opt_args(clip=clip)
            for s0 in range(0, n_seg, 32):
                s1 = min(n_seg, s0 + 32)
                base = int(seg["off"][s0])
                off = (seg["off"][s0 : s1 + 1] - base).astype(np.int64)
                sl = slice(base, int(seg["off"][s1]))
                _lib.check(
                    lib.xdfm_dense_update(
                        _ptr(self.dense[sl]), _ptr(self.dense_grad[sl]),
                        _ptr(self.dense_m[sl]) if self.dense_m is not None else None,
                        _ptr(self.dense_v[sl]) if self.dense_v is not None else None, s1 - s0,
                        off.ctypes.data_as(ctypes.c_void_p),
                        np.ascontiguousarray(seg["l2"][s0:s1]).ctypes.data_as(ctypes.c_void_p),
                        np.ascontiguousarray(seg["l1"][s0:s1]).ctypes.data_as(ctypes.c_void_p),
                        np.ascontiguousarray(seg["nl2"][s0:s1]).ctypes.data_as(ctypes.c_void_p),
                        ctypes.byref(opt), _ptr(self.stats), _ptr(self.reg_loss),
                        2 if self.early_reg else (0 if first else 1), st,
                    )
                )
                first = False
        if first:
            self.reg_loss.zero_()
        self._done(ev_upd)

    def catchup(self, sorted_ids, n):
        """xdfm_embed_catchup: aged rows written back, for the rows named by sorted_ids (None: the whole
        table)."""
        use_lin = self.lin_w is not None
        _lib.check(
            self.lib.xdfm_embed_catchup(
                _ptr(self.table), _ptr(self.table_m), _ptr(self.table_v), _ptr(self.lin_w) if use_lin else None,
                _ptr(self.lin_m) if use_lin else None, _ptr(self.lin_v) if use_lin else None,
                self.table_stride, self.lin_stride, self.N, self.D, _ptr(sorted_ids), int(n), self._age(0),
                self._stream(),
            )
        )

    def flush_rows(self):
        """Write every row as of the current step (before checkpoints and table exports: whatever reads
        the table without ageing rows itself)."""
        if self.catchup_on and self._rows_stale:
            self.catchup(None, self.N)
            self._rows_stale = False

    def train_step(self, batch, next_batch=None):
        """One optimisation step; returns (total_loss, data_loss) as 0-dim device tensors.
        The id sort of the batch runs on a side stream beside its forward pass (it depends on the ids
        only); next_batch is accepted for callers of earlier versions and not needed any more (rows are
        aged where they are read, nothing has to be prepared a step ahead)."""
        if self._plan_at and self.hp.use_CIN_part and not self.fast_d:
            self._plan_pending = batch  # forward() starts the sort behind its first launch
        else:
            self.plan_async(batch)
        self.forward(batch, with_loss=True, with_grad=True)
        advanced = self.backward(batch, advance=True, defer_late=True)
        if not advanced:
            self._advance_step()
        self.update(batch, True)
        tot = torch.empty((2,), dtype=torch.float32, device=self.device)
        _lib.check(self.lib.xdfm_loss_total(_ptr(self._loss2), _ptr(tot), self._stream()))  # [data, data + reg]
        return tot[1], tot[0]

    def capture_step(self, labels, dnn_idx, ids, weights):
        """Capture CSR build + train_step for batches of exactly this shape into a HIP graph
        (launch-bound loop: ~50 kernel launches per step become one graph launch).  Returns a
        GraphedStep; its run(labels, dnn_idx, ids, weights) copies the arrays into the graph's
        static inputs and replays."""
        return GraphedStep(self, labels, dnn_idx, ids, weights)

    # -------------------------------------------------------------------- eval
    def eval_metrics(self, preds, labels):
        """Fused AUC + logloss of device vectors -> (auc, logloss) python floats."""
        n = int(preds.numel())
        sz = ctypes.c_size_t()
        _lib.check(self.lib.xdfm_eval_workspace_bytes(n, ctypes.byref(sz)))
        if self._eval_ws is None or self._eval_ws.numel() < sz.value:
            self._eval_ws = torch.empty((sz.value,), dtype=torch.uint8, device=self.device)
        out = torch.zeros((4,), dtype=torch.float64, device=self.device)
        _lib.check(
            self.lib.xdfm_eval_auc_logloss(_ptr(preds), _ptr(labels), n, _ptr(out), _ptr(self._eval_ws),
                                           self._eval_ws.numel(), self._stream())
        )
        auc, logloss, n_pos, _ = out.tolist()
        return auc, logloss, n_pos


class PackedBatch(object):
    """A batch in a GraphedStep's input layout: one flat device buffer (`buf`).  `ids_src` names the
    tensor its ids were copied from."""

    __slots__ = ("buf", "ids_src")

    def __init__(self, buf, ids_src):
        self.buf, self.ids_src = buf, ids_src


def hw_queue_limit():
    """Hardware queues this process's HIP runtime will use when that is below its default: the runtime
    takes the number from GPU_MAX_HW_QUEUES (default 4) and offers no call to ask for it, so the
    variable is the runtime's own answer.  None: the default."""
    q = os.environ.get("GPU_MAX_HW_QUEUES", "").strip()
    return int(q) if q.isdigit() and int(q) >= 1 else None


def graph_replay_refusal():
    """None, or why a captured step must not be replayed in this process.  Nothing is refused any more:
    with fewer hardware queues than the step has branches the engine folds the step onto one stream
    (XDeepFMEngine.folded) and its capture is a chain.  Kept for callers of earlier versions."""
    return None


class GraphedStep(object):
    """A captured train step.  The graph reads its batch from one flat device buffer -- labels, the CSR
    offsets and token rows of the lookup (what the text reader hands over as `row_ptr`; built once
    from the index array by pack() / run(), which also validates it), ids and weights, at 256-byte
    aligned offsets -- so handing it a batch is ONE copy and nothing but the step's own kernels sits
    on the replay's chain (every small operation in front of the forward pass costs the step about
    8 us): run_packed(pack(...)) for a batch laid out ahead of time, run(...) for four arrays.

    The id sort of the batch runs inside the graph, on a side stream beside the forward pass (rows are
    aged where they are read, so nothing of a batch has to be prepared before its step).  The
    `next_*` arguments of earlier versions, which announced the following batch, are accepted and
    ignored."""

    def __init__(self, engine, labels, dnn_idx, ids, weights, csr=None):
        """csr = (row_ptr, tok_row): the lookup's offsets / token rows when they exist already (a
        DeviceBatch); dnn_idx may then be None.  The constructor TRAINS one step on the batch it is
        given (the warm-up a capture needs); its losses are left in warmup_loss."""
        self.engine = engine
        self._csr_only = dnn_idx is None
        if engine.kernel_events is not None:
            raise RuntimeError("kernel event hooks cannot be captured")
        if graph_replay_refusal():
            raise RuntimeError(graph_replay_refusal())
        dev = engine.device
        B, nnz = int(labels.numel()), int(ids.numel())
        self._B, self._nnz = B, nnz
        self._idx_shape, self._idx_dtype = (None, None) if dnn_idx is None else (tuple(dnn_idx.shape), dnn_idx.dtype)
        fields = [("labels", labels.dtype, tuple(labels.shape)),
                  ("row_ptr", torch.int32, (B * engine.F + 1,)),
                  ("tok_row", torch.int32, (max(nnz, 1),)),
                  ("ids", ids.dtype, tuple(ids.shape)),
                  ("weights", weights.dtype, tuple(weights.shape))]
        self._layout = {}
        off = 0
        for name, dt, shape in fields:
            n = 1
            for d in shape:
                n *= int(d)
            nbytes = n * torch.empty((), dtype=dt).element_size()
            self._layout[name] = (off, nbytes, dt, shape)
            off += (nbytes + 255) // 256 * 256
        self.flat = torch.empty((max(off, 256),), dtype=torch.uint8, device=dev)
        self.views = self._views(self.flat)
        self._fill(self.views, labels, dnn_idx, ids, weights, csr)
        # warm up on a side stream (allocations, lazy module loads), as torch asks before a capture
        s = torch.cuda.Stream(device=dev)
        s.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(s):
            self.warmup_loss = engine.train_step(self._batch_of_views())
        torch.cuda.current_stream(dev).wait_stream(s)
        torch.cuda.synchronize(dev)
        engine.check_errors()
        self._ws_token = engine._ws_token
        mirrors = (engine.step, engine.b1p, engine.b2p)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.batch = self._batch_of_views()
            self.loss, self.data_loss = engine.train_step(self.batch)
        # the capture pass ran the host bookkeeping of a step without device work
        engine.step, engine.b1p, engine.b2p = mirrors

    def _views(self, flat):
        return {name: flat[o:o + n].view(dt).view(shape) for name, (o, n, dt, shape) in self._layout.items()}

    def _batch_of_views(self):
        v = self.views
        return self.engine.make_batch_csr(v["labels"], v["row_ptr"], v["tok_row"], v["ids"], v["weights"])

    def _check(self, labels, dnn_idx, ids, weights):
        for name, src in (("labels", labels), ("ids", ids), ("weights", weights)):
            _, _, dt, shape = self._layout[name]
            if tuple(src.shape) != shape or src.dtype != dt:
                raise ValueError("batch array %s/%s does not match the captured shape %s/%s"
                                 % (tuple(src.shape), src.dtype, shape, dt))
        if dnn_idx is not None and self._idx_shape is not None and (
                tuple(dnn_idx.shape) != self._idx_shape or dnn_idx.dtype != self._idx_dtype):
            raise ValueError("batch array %s/%s does not match the captured shape %s/%s"
                             % (tuple(dnn_idx.shape), dnn_idx.dtype, self._idx_shape, self._idx_dtype))

    def _fill(self, views, labels, dnn_idx, ids, weights, csr=None):
        """The arrays of a batch into the views of a flat buffer; the CSR offsets / token rows are
        built from the index array here (xdfm_build_row_ptr, which also flags bad indices) unless
        they are handed over (csr)."""
        eng = self.engine
        views["labels"].copy_(labels.reshape(views["labels"].shape))
        views["ids"].copy_(ids)
        views["weights"].copy_(weights)
        if csr is not None:
            views["row_ptr"].copy_(csr[0])
            views["tok_row"].copy_(csr[1][:views["tok_row"].numel()])
        else:
            _lib.check(eng.lib.xdfm_build_row_ptr(_ptr(dnn_idx), self._nnz, self._B * eng.F, _ptr(views["row_ptr"]),
                                                  _ptr(views["tok_row"]), _ptr(eng.err_flag), eng._stream()))

    def pack(self, labels, dnn_idx, ids, weights, next_ids=None):
        """A batch in the graph's input layout (a new device buffer)."""
        self._check(labels, dnn_idx, ids, weights)
        buf = torch.empty_like(self.flat)
        self._fill(self._views(buf), labels, dnn_idx, ids, weights)
        return PackedBatch(buf, ids)

    def pack_batch(self, batch, next_ids=None):
        """pack() for a DeviceBatch (its CSR offsets exist already)."""
        if batch.B != self._B or batch.nnz != self._nnz:
            raise ValueError("batch of %d samples / %d tokens does not match the captured %d / %d"
                             % (batch.B, batch.nnz, self._B, self._nnz))
        buf = torch.empty_like(self.flat)
        self._fill(self._views(buf), batch.labels, None, batch.ids, batch.weights,
                   csr=(batch.row_ptr, batch.tok_row))
        return PackedBatch(buf, batch.ids)

    def _replay(self):
        eng = self.engine
        if eng._ws_token != self._ws_token:
            raise RuntimeError("the engine's workspace was re-allocated after this step was captured "
                               "(a larger batch went through the engine): capture it again")
        self.graph.replay()
        eng.step += 1
        eng._rows_stale = eng.catchup_on
        return self.loss, self.data_loss

    def run_batch(self, pb, key=None, next_key=None, next_ids=None):
        return self.run_packed(pb)

    def run_packed(self, pb, next_pb=None):
        self.flat.copy_(pb.buf, non_blocking=True)
        return self._replay()

    def run(self, labels, dnn_idx, ids, weights, next_ids=None):
        self._check(labels, dnn_idx, ids, weights)
        self._fill(self.views, labels, dnn_idx, ids, weights)
        return self._replay()
