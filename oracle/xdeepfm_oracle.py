"""CPU oracle of the xDeepFM graph -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

PARITY UNPINNED at op level: the reference's arithmetic runs inside TensorFlow 2.7
(un-vendored, absent here) and its tests hold no logits / gradients.  This file restates
the graph the reference builds, op by op, on torch CPU tensors (float32 or float64), and
lets autograd play TF's autodiff.  Citations are to
/root/reference/recommenders/models/deeprec/models/{xDeepFM,base_model}.py; TF op
semantics that are not in the reference are the published ones.

It materialises Z = [B, D, F*H] exactly like the reference (xDeepFM.py:193-200), so it is
only meant for small/medium cases and for the bounded cpu_baseline sample of bench.py.
"""

import math

import numpy as np
import torch

# ---------------------------------------------------------------------------
# parameters
# ---------------------------------------------------------------------------


def cin_shapes(hp):
    """Filter shapes of _build_CIN (xDeepFM.py:191-267): list of (H_k, L_k, n_hidden_k)."""
    out = []
    prev = hp.FIELD_COUNT
    n_layers = len(hp.cross_layer_sizes)
    for idx, layer_size in enumerate(hp.cross_layer_sizes):
        n_hidden = int(layer_size / 2) if idx != n_layers - 1 else 0
        out.append((prev, layer_size, n_hidden))
        prev = int(layer_size / 2)  # xDeepFM.py:262
    return out


def cin_final_len(hp):
    return sum(L - nh for (_, L, nh) in cin_shapes(hp))


def param_shapes(hp):
    """name -> shape, in creation order of the reference graph."""
    shapes = {"embedding": (hp.FEATURE_COUNT, hp.dim)}  # xDeepFM.py:38-42
    if hp.use_Linear_part:
        shapes["lin_w"] = (hp.FEATURE_COUNT, 1)  # :114-116
        shapes["lin_b"] = (1,)  # :117-122
    if hp.use_CIN_part:
        for k, (H, L, _) in enumerate(cin_shapes(hp)):
            if hp.fast_CIN_d > 0:
                # _build_fast_CIN: fast_w [1, F, d*L] (:330-334, :353-357), fast_v [1, H, d*L] (:358-362)
                shapes["fcin_w_%d" % k] = (hp.FIELD_COUNT, hp.fast_CIN_d * L)
                if k > 0:
                    shapes["fcin_v_%d" % k] = (H, hp.fast_CIN_d * L)
            else:
                shapes["cin_f_%d" % k] = (H * hp.FIELD_COUNT, L)  # :202-206 ([1, H*F, L] squeezed)
            if hp.enable_BN is True:  # tf.layers.batch_normalization variables (:235-240, :395-400)
                for part in ("g", "o"):  # gamma, beta (offset): trainable
                    shapes["bn_cin_%s_%d" % (part, k)] = (L,)
                for part in ("mm", "mv"):  # moving mean / variance: not trainable
                    shapes["bnstat_cin_%s_%d" % (part, k)] = (L,)
        shapes["cin_w_out"] = (cin_final_len(hp), 1)  # :279-281
        shapes["cin_b_out"] = (1,)  # :282-287
    if hp.use_DNN_part:
        last = hp.FIELD_COUNT * hp.dim
        for k, size in enumerate(hp.layer_sizes):
            shapes["dnn_w_%d" % k] = (last, size)  # :474-478
            shapes["dnn_b_%d" % k] = (size,)  # :479-484
            if hp.enable_BN is True:  # :498-503
                for part in ("g", "o"):
                    shapes["bn_dnn_%s_%d" % (part, k)] = (size,)
                for part in ("mm", "mv"):
                    shapes["bnstat_dnn_%s_%d" % (part, k)] = (size,)
            last = size
        shapes["dnn_w_out"] = (last, 1)  # :514-516
        shapes["dnn_b_out"] = (1,)  # :517-522
    return shapes


def param_groups(hp):
    """(embed_params, layer_params, cross_params) name lists (base_model.py:43-45 and the
    appends at xDeepFM.py:43, :129-130, :267, :288-289, :511-512, :529-530)."""
    names = list(param_shapes(hp))
    embed = ["embedding"]
    cross = [n for n in names if n.startswith(("cin_f_", "fcin_"))]  # also :336, :364-365
    # batch-norm variables are created by tf.layers, never appended to a params list: no regulariser
    layer = [n for n in names if n not in embed and n not in cross and not n.startswith("bn")]
    return embed, layer, cross


def init_params(hp, seed=0, dtype=torch.float32):
    """Random parameters with the reference's initialiser *distributions*
    (base_model.py:169-213); biases are zeros.  TF's random streams cannot be reproduced,
    so parity runs load these same tensors into both sides."""
    gen = torch.Generator().manual_seed(seed)
    params = {}
    for name, shape in param_shapes(hp).items():
        if name.startswith("bn"):  # gamma = 1, beta = 0, moving mean = 0, moving variance = 1
            one = name.startswith(("bn_cin_g_", "bn_dnn_g_", "bnstat_cin_mv_", "bnstat_dnn_mv_"))
            t = torch.ones(shape, dtype=torch.float64) if one else torch.zeros(shape, dtype=torch.float64)
        elif name.endswith("_b") or "_b_" in name:
            t = torch.zeros(shape, dtype=torch.float64)
        else:
            t = _init_tensor(hp, shape, gen)
        params[name] = t.to(dtype)
    return params


def _init_tensor(hp, shape, gen):
    method = hp.init_method
    fan_in = shape[0]
    fan_out = shape[1] if len(shape) > 1 else shape[0]
    if method == "uniform":
        return (torch.rand(shape, generator=gen, dtype=torch.float64) * 2 - 1) * hp.init_value
    if method == "normal":
        return torch.randn(shape, generator=gen, dtype=torch.float64) * hp.init_value
    if method in ("xavier_normal", "he_normal"):
        scale, n = (1.0, (fan_in + fan_out) / 2.0) if method.startswith("xavier") else (2.0, fan_in)
        std = math.sqrt(scale / n) / 0.87962566103423978
        return _truncated_normal(shape, std, gen)
    if method in ("xavier_uniform", "he_uniform"):
        scale, n = (1.0, (fan_in + fan_out) / 2.0) if method.startswith("xavier") else (2.0, fan_in)
        limit = math.sqrt(3.0 * scale / n)
        return (torch.rand(shape, generator=gen, dtype=torch.float64) * 2 - 1) * limit
    return _truncated_normal(shape, hp.init_value, gen)  # "tnormal" and the fallback


def _truncated_normal(shape, std, gen):
    # values beyond two standard deviations are redrawn
    t = torch.randn(shape, generator=gen, dtype=torch.float64)
    bad = t.abs() > 2
    while bad.any():
        t[bad] = torch.randn(int(bad.sum()), generator=gen, dtype=torch.float64)
        bad = t.abs() > 2
    return t * std


# ---------------------------------------------------------------------------
# ops
# ---------------------------------------------------------------------------


def embedding_lookup_sparse_np(table, dnn_indices, dnn_values, dnn_weights, n_rows):
    """Bit-level oracle of tf.nn.embedding_lookup_sparse(combiner="sum") with weights
    (xDeepFM.py:92-97): per token one fp32 multiply, per row a sequential fp32 sum in column
    order; rows without tokens are zero (the build defines this for trailing rows too).
    numpy float32; returns [n_rows, D]."""
    table = np.asarray(table, dtype=np.float32)
    out = np.zeros((n_rows, table.shape[1]), dtype=np.float32)
    if len(dnn_values) == 0:
        return out
    rows = np.asarray(dnn_indices)[:, 0]
    cols = np.asarray(dnn_indices)[:, 1]
    ids = np.asarray(dnn_values)
    w = np.asarray(dnn_weights, dtype=np.float32)
    # position of a token inside its row (tokens are sorted by (row, col); cols may repeat
    # only when field indices alias, so use the rank inside the row, not the col value)
    first = np.ones(len(rows), dtype=bool)
    first[1:] = rows[1:] != rows[:-1]
    start = np.maximum.accumulate(np.where(first, np.arange(len(rows)), 0))
    rank = np.arange(len(rows)) - start
    del cols
    for level in range(int(rank.max()) + 1):
        sel = rank == level
        t = w[sel, None] * table[ids[sel]]  # one rounding per element
        if level == 0:
            out[rows[sel]] = t
        else:
            out[rows[sel]] = out[rows[sel]] + t
    return out


def _lookup_torch(table, rows, ids, w, n_rows):
    first = torch.ones(len(rows), dtype=torch.bool)
    first[1:] = rows[1:] != rows[:-1]
    pos = torch.arange(len(rows))
    start = torch.cummax(torch.where(first, pos, torch.zeros_like(pos)), 0).values
    rank = pos - start
    out = torch.zeros((n_rows, table.shape[1]), dtype=table.dtype)
    for level in range(int(rank.max()) + 1 if len(rows) else 0):
        sel = rank == level
        out = out.index_add(0, rows[sel], w[sel, None].to(table.dtype) * table[ids[sel]])
    return out


def _activate(x, name):
    """base_model.py:339-353"""
    if name == "sigmoid":
        return torch.sigmoid(x)
    if name == "softmax":
        return torch.softmax(x, dim=-1)
    if name == "relu":
        return torch.relu(x)
    if name == "tanh":
        return torch.tanh(x)
    if name == "elu":
        return torch.nn.functional.elu(x)
    if name == "identity":
        return x
    raise ValueError("this activations not defined {0}".format(name))


BN_MOMENTUM, BN_EPS = 0.95, 0.0001  # the constants at every call site (xDeepFM.py:237-238, :397-398, :500-501)


def _batch_norm(x, params, where, k, training, moving_out):
    """tf.compat.v1.layers.batch_normalization over the last axis, non-fused path (inputs of rank
    2 and 3): batch moments with the biased variance in training, the moving statistics
    otherwise; training records the moved statistics (the UPDATE_OPS of base_model.py:62-64,
    run with the train op at :381-382) in moving_out."""
    gamma, beta = params["bn_%s_g_%d" % (where, k)], params["bn_%s_o_%d" % (where, k)]
    mm, mv = params["bnstat_%s_mm_%d" % (where, k)], params["bnstat_%s_mv_%d" % (where, k)]
    if training:
        red = tuple(range(x.dim() - 1))
        mean = x.mean(red)
        var = ((x - mean) ** 2).mean(red)
        if moving_out is not None:
            with torch.no_grad():
                moving_out["bnstat_%s_mm_%d" % (where, k)] = mm - (mm - mean) * (1 - BN_MOMENTUM)
                moving_out["bnstat_%s_mv_%d" % (where, k)] = mv - (mv - var) * (1 - BN_MOMENTUM)
    else:
        mean, var = mm, mv
    return (x - mean) / torch.sqrt(var + BN_EPS) * gamma + beta


def forward(params, batch, hp, embedding_rows=None, dropout_masks=None, training=False, moving_out=None):
    """Logit [B,1] of XDeepFMModel._build_graph (xDeepFM.py:24-71).

    batch: dict with the iterator's arrays (labels, dnn_feat_*, fm_feat_*) as numpy.
    embedding_rows: optional (table_sub, ids_remap) to run on a gathered sub-table
    (used by the sparse_rows step)."""
    table = params["embedding"]
    dt = table.dtype
    B = int(batch["labels"].shape[0])
    F, D = hp.FIELD_COUNT, hp.dim
    nnz = len(batch["dnn_feat_values"])
    rows = torch.as_tensor(np.asarray(batch["dnn_feat_indices"]).reshape(-1, 2)[:, 0].copy())
    ids = torch.as_tensor(np.asarray(batch["dnn_feat_values"]).copy())
    w = torch.as_tensor(np.asarray(batch["dnn_feat_weights"]).copy())
    fm_idx = torch.as_tensor(np.asarray(batch["fm_feat_indices"]).reshape(-1, 2).copy())
    fm_val = torch.as_tensor(np.asarray(batch["fm_feat_values"]).copy()).to(dt)
    fm_ids = fm_idx[:, 1]
    if embedding_rows is not None:
        table, remap = embedding_rows
        ids = remap(ids)
        fm_ids = remap(fm_ids)
    aux = {}

    # _build_embedding  xDeepFM.py:73-102
    embed_out = _lookup_torch(table, rows, ids, w, B * F).reshape(B, F * D)
    aux["embed_out"] = embed_out
    logit = 0

    if hp.use_Linear_part:  # :104-133
        lin_w = params["lin_w"] if "lin_rows" not in params else params["lin_rows"]
        contrib = fm_val[:, None] * lin_w[fm_ids]
        lin = torch.zeros((B, 1), dtype=dt).index_add(0, fm_idx[:, 0], contrib) + params["lin_b"]
        aux["linear"] = lin
        logit = logit + lin

    if hp.use_FM_part:  # :135-159
        xe = torch.zeros((B, D), dtype=dt).index_add(0, fm_idx[:, 0], fm_val[:, None] * table[fm_ids])
        xxe2 = torch.zeros((B, D), dtype=dt).index_add(
            0, fm_idx[:, 0], (fm_val**2)[:, None] * table[fm_ids] ** 2
        )
        fm = 0.5 * torch.sum(xe**2 - xxe2, dim=1, keepdim=True)
        aux["fm"] = fm
        logit = logit + fm

    if hp.use_CIN_part:  # :161-293 with res=True, direct=False, bias=False, is_masked=True
        x0 = embed_out.reshape(B, F, D)
        hidden = x0
        finals = []
        n_layers = len(hp.cross_layer_sizes)
        fast_d = int(hp.fast_CIN_d)
        x0t = x0.permute(0, 2, 1)  # [B,D,F]  (fast CIN works on this view, xDeepFM.py:321)
        for k, (H, L, n_hidden) in enumerate(cin_shapes(hp)):
            if fast_d > 0:  # _build_fast_CIN (xDeepFM.py:295-451), res=True, direct=False, bias=False
                w = params["fcin_w_%d" % k]
                if k == 0:
                    dot1 = torch.matmul(x0t, w)  # :337-339 -- not squared in the reference
                    dot2 = torch.matmul(x0t**2, w**2)  # :340-345
                    dot = 0.5 * (dot1 - dot2)  # :346-349
                else:
                    v = params["fcin_v_%d" % k]
                    dot = torch.matmul(x0t, w) * torch.matmul(hidden.permute(0, 2, 1), v)  # :367-381
                curr = dot.reshape(B, D, L, fast_d).sum(3)  # :346-351 / :378-383  [B,D,L]
            else:
                # dot_result_m [D,B,F,H] -> [D,B,F*H] -> [B,D,F*H]   (:192-200)
                z = torch.einsum("bfd,bhd->dbfh", x0, hidden).reshape(D, B, F * H).permute(1, 0, 2)
                if k == 0:  # is_masked (:208-217): strictly upper, times 2
                    mask = torch.triu(torch.ones(F, F, dtype=dt), diagonal=1).reshape(1, 1, F * F)
                    z = z * mask * 2
                curr = torch.matmul(z, params["cin_f_%d" % k])  # conv1d width 1 (:220-222) [B,D,L]
            if hp.enable_BN is True:  # :234-240 / :394-400, on [B,D,L]: channels = L
                curr = _batch_norm(curr, params, "cin", k, training, moving_out)
            curr = _activate(curr, hp.cross_activation)  # :242 / :402
            curr = curr.permute(0, 2, 1)  # [B,L,D]  (:244; the fast branch splits axis 2 of [B,D,L], :412-414)
            if k != n_layers - 1:
                nxt, direct = torch.split(curr, 2 * [int(L / 2)], 1)  # :254-256
            else:
                nxt, direct = None, curr  # :258-261
            finals.append(direct)
            hidden = nxt
        result = torch.cat(finals, dim=1).sum(-1)  # [B, final_len]  (:269-270)
        aux["cin_pooled"] = result
        base_score = result.sum(1, keepdim=True)  # res=True (:272-275)
        cin = base_score + result @ params["cin_w_out"] + params["cin_b_out"]  # :290-292
        aux["cin"] = cin
        logit = logit + cin

    if hp.use_DNN_part:  # :453-534
        h = embed_out
        for k in range(len(hp.layer_sizes)):
            h = h @ params["dnn_w_%d" % k] + params["dnn_b_%d" % k]
            if hp.enable_BN is True:  # :497-503
                h = _batch_norm(h, params, "dnn", k, training, moving_out)
            if hp.user_dropout and dropout_masks is not None:
                # _active_layer (base_model.py:335-336): tf.nn.dropout on the pre-activation.  TF's
                # random stream is not reproducible: the caller supplies m / keep per layer
                # (training); None = keep probability 1 (eval / infer, base_model.py:400, :414)
                h = h * torch.as_tensor(dropout_masks[k], dtype=h.dtype)
            h = _activate(h, hp.activation[k])
        dnn = h @ params["dnn_w_out"] + params["dnn_b_out"]
        aux["dnn"] = dnn
        logit = logit + dnn

    del nnz
    return logit, aux


def pred_from_logit(logit, hp):
    """base_model.py:91-112"""
    if hp.method == "regression":
        return logit
    if hp.method == "classification":
        return torch.sigmoid(logit)
    raise ValueError("method must be regression or classification, but now is {0}".format(hp.method))


def data_loss(logit, pred, labels, hp):
    """base_model.py:215-262 (softmax loss is not used by xDeepFM)."""
    y = labels.reshape(-1)
    if hp.loss == "cross_entropy_loss":
        z = logit.reshape(-1)
        return torch.mean(torch.clamp(z, min=0) - z * y + torch.log1p(torch.exp(-z.abs())))
    if hp.loss == "square_loss":
        return torch.sqrt(torch.mean((pred.reshape(-1) - y) ** 2))
    if hp.loss == "log_loss":
        p = pred.reshape(-1)
        eps = 1e-7  # tf.compat.v1.losses.log_loss default
        return torch.mean(-y * torch.log(p + eps) - (1 - y) * torch.log(1 - p + eps))
    raise ValueError("this loss not defined {0}".format(hp.loss))


def regular_loss(params, hp):
    """base_model.py:121-167, :264-272"""
    embed, layer, cross = param_groups(hp)
    total = 0
    for n in embed:
        total = total + hp.embed_l2 * (params[n] ** 2).sum() / 2 + hp.embed_l1 * params[n].abs().sum()
    for n in layer:
        total = total + hp.layer_l2 * (params[n] ** 2).sum() / 2 + hp.layer_l1 * params[n].abs().sum()
    for n in cross:
        total = total + hp.cross_l1 * params[n].abs().sum()
        total = total + hp.cross_l2 * torch.sqrt((params[n] ** 2).sum())
    return total


# ---------------------------------------------------------------------------
# optimizer (TF1 semantics) and one training step
# ---------------------------------------------------------------------------


def new_opt_state(params, hp):
    st = {"step": 0}
    if hp.optimizer == "adam":
        st["m"] = {n: torch.zeros_like(p) for n, p in params.items()}
        st["v"] = {n: torch.zeros_like(p) for n, p in params.items()}
        # fp32 beta powers multiplied up step by step, like TF's slot variables
        st["b1p"] = np.float32(1.0)
        st["b2p"] = np.float32(1.0)
    elif hp.optimizer in ("adagrad", "padagrad"):
        # ProximalAdagradOptimizer(lr) has l1 = l2 = 0: its update is Adagrad's (base_model.py:295-296)
        st["acc"] = {n: torch.full_like(p, 0.1) for n, p in params.items()}  # initial_accumulator_value
    elif hp.optimizer == "rmsprop":  # RMSPropOptimizer(lr): decay .9, momentum 0, eps 1e-10, rms slot = 1
        st["ms"] = {n: torch.ones_like(p) for n, p in params.items()}
    elif hp.optimizer == "adadelta":  # AdadeltaOptimizer(lr): rho .95, eps 1e-8, both slots 0
        st["acc"] = {n: torch.zeros_like(p) for n, p in params.items()}
        st["acc_upd"] = {n: torch.zeros_like(p) for n, p in params.items()}
    elif hp.optimizer == "ftrl":  # FtrlOptimizer(lr): lr_power -.5, accum 0.1, l1 = l2 = 0
        st["acc"] = {n: torch.full_like(p, 0.1) for n, p in params.items()}
        st["lin"] = {n: torch.zeros_like(p) for n, p in params.items()}
    return st


def _apply(name, p, g, st, hp, rows=None):
    """In-place TF1 update of p (optionally only of p[rows] with g = grad of those rows)."""
    lr = hp.learning_rate
    sel = slice(None) if rows is None else rows
    if hp.optimizer == "adam":
        m, v = st["m"][name], st["v"][name]
        b1, b2, eps = 0.9, 0.999, 1e-8
        if p.dtype == torch.float32:
            # TF's ApplyAdam evaluates everything in the variable's dtype
            alpha = float(np.float32(lr) * np.sqrt(np.float32(1) - st["b2p"]) / (np.float32(1) - st["b1p"]))
            one_b1 = float(np.float32(1) - np.float32(b1))
            one_b2 = float(np.float32(1) - np.float32(b2))
            eps = float(np.float32(eps))
        else:
            alpha = lr * math.sqrt(1 - b2 ** st["step"]) / (1 - b1 ** st["step"])
            one_b1, one_b2 = 1 - b1, 1 - b2
        m[sel] = m[sel] + (g - m[sel]) * one_b1
        v[sel] = v[sel] + (g * g - v[sel]) * one_b2
        p[sel] = p[sel] - (m[sel] * alpha) / (torch.sqrt(v[sel]) + eps)
    elif hp.optimizer in ("adagrad", "padagrad"):
        acc = st["acc"][name]
        acc[sel] = acc[sel] + g * g
        p[sel] = p[sel] - (g * lr) / torch.sqrt(acc[sel])
    elif hp.optimizer == "rmsprop":
        # TF ApplyRMSProp: ms += (g^2 - ms)*(1-decay); mom = momentum*mom + lr*g*rsqrt(ms+eps); p -= mom
        ms = st["ms"][name]
        ms[sel] = ms[sel] + (g * g - ms[sel]) * (1 - 0.9)
        p[sel] = p[sel] - (g * lr) / torch.sqrt(ms[sel] + 1e-10)
    elif hp.optimizer == "adadelta":
        # TF ApplyAdadelta: accum = rho*accum + (1-rho)*g^2; update = sqrt(accum_update+eps)*rsqrt(accum+eps)*g;
        # accum_update = rho*accum_update + (1-rho)*update^2; p -= lr*update
        acc, au = st["acc"][name], st["acc_upd"][name]
        rho, eps = 0.95, 1e-8
        acc[sel] = acc[sel] * rho + g * g * (1 - rho)
        upd = torch.sqrt(au[sel] + eps) / torch.sqrt(acc[sel] + eps) * g
        au[sel] = au[sel] * rho + upd * upd * (1 - rho)
        p[sel] = p[sel] - upd * lr
    elif hp.optimizer == "ftrl":
        # TF ApplyFtrl with lr_power = -0.5, l1 = l2 = 0:
        # new = accum + g^2; linear += g - (sqrt(new) - sqrt(accum))/lr * p; p = -linear / (sqrt(new)/lr)
        acc, lin = st["acc"][name], st["lin"][name]
        new = acc[sel] + g * g
        lin[sel] = lin[sel] + g - (torch.sqrt(new) - torch.sqrt(acc[sel])) / lr * p[sel]
        p[sel] = -lin[sel] / (torch.sqrt(new) / lr)
        acc[sel] = new
    else:  # sgd / gd / pgd (l1 = l2 = 0) / unknown -> GradientDescent (base_model.py:288, :293, :297-302)
        p[sel] = p[sel] - g * lr


def train_step(params, st, batch, hp, mode="dense_exact", dropout_masks=None):
    """One `sess.run(update)` (base_model.py:367-388): returns (loss, data_loss, grads).

    dense_exact : reference semantics -- every variable (the whole table included) takes an
                  optimizer step every batch, regularisers included.
    sparse_rows : only the table / lin_w rows that occur in the batch are updated (lazy
                  optimizer); requires embed_l1 = embed_l2 = 0 and layer_l1/l2 not applied
                  to lin_w ... see DESIGN.md.  Dense parameters are updated as in dense_exact.
    """
    labels = torch.as_tensor(np.asarray(batch["labels"])).to(params["embedding"].dtype)
    st["step"] += 1
    if hp.optimizer == "adam":
        st["b1p"] = np.float32(st["b1p"] * np.float32(0.9))
        st["b2p"] = np.float32(st["b2p"] * np.float32(0.999))

    sparse = mode in ("sparse_rows", "sparse_lazy")  # one name per engine mode that skips absent rows
    leaves = {}
    moving = {}  # batch-norm moving statistics after this step
    if sparse:
        ids_all = torch.as_tensor(np.asarray(batch["dnn_feat_values"]).copy())
        uniq = torch.unique(ids_all)  # sorted
        lookup = {"uniq": uniq}

        def remap(x):
            return torch.searchsorted(lookup["uniq"], x.contiguous())

        sub = params["embedding"][uniq].clone().requires_grad_(True)
        leaves["embedding"] = sub
        work = dict(params)
        if hp.use_Linear_part:
            lin_sub = params["lin_w"][uniq].clone().requires_grad_(True)
            leaves["lin_w"] = lin_sub
            work["lin_rows"] = lin_sub
        for n, p in params.items():
            if n not in ("embedding", "lin_w") and not n.startswith("bnstat_"):
                leaves[n] = p.clone().requires_grad_(True)
                work[n] = leaves[n]
        logit, aux = forward(work, batch, hp, embedding_rows=(sub, remap), dropout_masks=dropout_masks,
                             training=True, moving_out=moving)
        reg_params = {n: leaves[n] for n in leaves if n not in ("embedding", "lin_w")}
    else:
        work = dict(params)
        for n, p in params.items():
            if not n.startswith("bnstat_"):
                leaves[n] = p.clone().requires_grad_(True)
                work[n] = leaves[n]
        logit, aux = forward(work, batch, hp, dropout_masks=dropout_masks, training=True, moving_out=moving)
        reg_params = leaves

    pred = pred_from_logit(logit, hp)
    dl = data_loss(logit, pred, labels, hp)
    if sparse:
        rl = _regular_loss_subset(reg_params, hp)
    else:
        rl = regular_loss(reg_params, hp)
    loss = dl + rl
    loss.backward()

    grads = {}
    for n, leaf in leaves.items():
        g = leaf.grad if leaf.grad is not None else torch.zeros_like(leaf)
        if hp.is_clip_norm:  # per-tensor clip_by_norm (base_model.py:315-321)
            nrm = torch.sqrt((g * g).sum())
            g = g * hp.max_grad_norm / torch.clamp(nrm, min=hp.max_grad_norm)
        grads[n] = g
        with torch.no_grad():
            if sparse and n in ("embedding", "lin_w"):
                _apply(n, params[n], g, st, hp, rows=uniq)
            else:
                _apply(n, params[n], g, st, hp)
    for n, t in moving.items():
        params[n].copy_(t)
    return float(loss.detach()), float(dl.detach()), grads, aux


def _regular_loss_subset(params, hp):
    """regular_loss over the dense parameters only (sparse_rows mode)."""
    total = 0
    for n, p in params.items():
        if n.startswith("bn"):
            continue
        if n.startswith(("cin_f_", "fcin_")):
            total = total + hp.cross_l1 * p.abs().sum() + hp.cross_l2 * torch.sqrt((p**2).sum())
        else:
            total = total + hp.layer_l2 * (p**2).sum() / 2 + hp.layer_l1 * p.abs().sum()
    return total


def predict(params, batch, hp):
    with torch.no_grad():
        logit, _ = forward(params, batch, hp)
        return pred_from_logit(logit, hp).reshape(-1)
