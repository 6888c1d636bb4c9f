"""Self-consistency of the CPU oracle of the graph (oracle/xdeepfm_oracle.py).  Op-level
parity with TensorFlow is UNPINNED (no TF here, no logits in the reference's tests); these
tests pin the restatement against independently written formulas (SURVEY appendix C and the
reference's own docstrings) and fp64."""
import math

import numpy as np
import torch

from oracle import xdeepfm_oracle as X
from tests.util import make_hp, random_batch


def _to(params, dt):
    return {k: v.to(dt) for k, v in params.items()}


def test_lookup_np_is_bit_exact_sequential_sum():
    rng = np.random.default_rng(0)
    hp = make_hp()
    b = random_batch(rng, hp, 32)
    table = rng.standard_normal((hp.FEATURE_COUNT, hp.dim)).astype(np.float32)
    out = X.embedding_lookup_sparse_np(table, b["dnn_feat_indices"], b["dnn_feat_values"],
                                       b["dnn_feat_weights"], 32 * hp.FIELD_COUNT)
    # token by token python loop, the definition
    ref = np.zeros_like(out)
    seen = set()
    for (row, _), fid, w in zip(b["dnn_feat_indices"], b["dnn_feat_values"], b["dnn_feat_weights"]):
        t = np.float32(w) * table[fid]
        ref[row] = t if row not in seen else ref[row] + t
        seen.add(row)
    assert np.array_equal(out, ref)
    assert out.dtype == np.float32


def test_forward_against_naive_loops():
    """CIN / FM / linear / DNN written as explicit loops over (b, d, f, h) in float64."""
    rng = np.random.default_rng(1)
    hp = make_hp(FEATURE_COUNT=60, FIELD_COUNT=4, dim=3, cross_layer_sizes=[4, 2], layer_sizes=[5],
                 activation=["relu"])
    B = 5
    b = random_batch(rng, hp, B)
    p = X.init_params(hp, seed=3, dtype=torch.float64)
    p["lin_b"] += 0.3
    p["cin_b_out"] += 0.1
    logit, aux = X.forward(p, b, hp)
    E = p["embedding"].numpy()
    F, D = hp.FIELD_COUNT, hp.dim
    emb = np.zeros((B * F, D))
    for (row, _), fid, w in zip(b["dnn_feat_indices"], b["dnn_feat_values"], b["dnn_feat_weights"]):
        emb[row] += float(w) * E[fid]
    x0 = emb.reshape(B, F, D)
    lin = np.zeros(B) + p["lin_b"].item()
    s = np.zeros((B, D))
    q = np.zeros((B, D))
    for (smp, fid), v in zip(b["fm_feat_indices"], b["fm_feat_values"]):
        lin[smp] += float(v) * p["lin_w"][fid, 0].item()
        s[smp] += float(v) * E[fid]
        q[smp] += float(v) ** 2 * E[fid] ** 2
    fm = 0.5 * (s * s - q).sum(1)
    hidden = x0
    pooled = []
    shapes = X.cin_shapes(hp)
    for k, (H, L, nh) in enumerate(shapes):
        W = p["cin_f_%d" % k].numpy()
        curr = np.zeros((B, L, D))
        for bb in range(B):
            for d in range(D):
                for f in range(F):
                    for h in range(H):
                        z = x0[bb, f, d] * hidden[bb, h, d]
                        if k == 0:
                            z = z * 2 if f < h else 0.0
                        curr[bb, :, d] += z * W[f * H + h]
        if k != len(shapes) - 1:
            hidden, direct = curr[:, :nh], curr[:, nh:]
        else:
            direct = curr
        pooled.append(direct.sum(-1))
    r = np.concatenate(pooled, 1)
    cin = r.sum(1) + r @ p["cin_w_out"].numpy()[:, 0] + p["cin_b_out"].item()
    h = emb.reshape(B, F * D)
    h = np.maximum(h @ p["dnn_w_0"].numpy() + p["dnn_b_0"].numpy(), 0)
    dnn = h @ p["dnn_w_out"].numpy()[:, 0] + p["dnn_b_out"].item()
    np.testing.assert_allclose(aux["linear"].numpy()[:, 0], lin, rtol=1e-12)
    np.testing.assert_allclose(aux["fm"].numpy()[:, 0], fm, rtol=1e-10, atol=1e-14)
    np.testing.assert_allclose(aux["cin"].numpy()[:, 0], cin, rtol=1e-10, atol=1e-14)
    np.testing.assert_allclose(aux["dnn"].numpy()[:, 0], dnn, rtol=1e-12)
    np.testing.assert_allclose(logit.numpy()[:, 0], lin + fm + cin + dnn, rtol=1e-10)


def test_fp32_oracle_tracks_fp64():
    rng = np.random.default_rng(2)
    hp = make_hp()
    b = random_batch(rng, hp, 48)
    p64 = X.init_params(hp, seed=5, dtype=torch.float64)
    l64, _ = X.forward(p64, b, hp)
    l32, _ = X.forward(_to(p64, torch.float32), b, hp)
    np.testing.assert_allclose(l32.numpy(), l64.numpy(), rtol=2e-5, atol=2e-6)


def test_losses_and_gradients_against_appendix_c():
    rng = np.random.default_rng(3)
    z = torch.tensor(rng.standard_normal(16) * 3, dtype=torch.float64, requires_grad=True)
    y = torch.tensor(rng.integers(0, 2, 16), dtype=torch.float64)
    for loss in ("log_loss", "cross_entropy_loss", "square_loss"):
        hp = make_hp(loss=loss)
        z.grad = None
        pred = X.pred_from_logit(z, hp)
        L = X.data_loss(z, pred, y, hp)
        L.backward()
        p = torch.sigmoid(z.detach())
        B = 16
        if loss == "log_loss":
            eps = 1e-7
            exp_l = torch.mean(-y * torch.log(p + eps) - (1 - y) * torch.log(1 - p + eps))
            g = (-y / (p + eps) + (1 - y) / (1 - p + eps)) * p * (1 - p) / B
        elif loss == "cross_entropy_loss":
            exp_l = torch.mean(-y * torch.log(p) - (1 - y) * torch.log(1 - p))
            g = (p - y) / B
        else:
            exp_l = torch.sqrt(torch.mean((p - y) ** 2))
            g = (p - y) * p * (1 - p) / (B * exp_l)
        assert math.isclose(L.item(), exp_l.item(), rel_tol=1e-9)
        np.testing.assert_allclose(z.grad.numpy(), g.numpy(), rtol=1e-8, atol=1e-14)


def test_regular_loss_definition():
    hp = make_hp(embed_l2=0.01, embed_l1=0.002, layer_l2=0.03, layer_l1=0.004, cross_l2=0.05, cross_l1=0.006)
    p = X.init_params(hp, seed=1, dtype=torch.float64)
    embed, layer, cross = X.param_groups(hp)
    assert embed == ["embedding"] and "lin_w" in layer and "dnn_b_0" in layer and cross == ["cin_f_0", "cin_f_1", "cin_f_2"]
    exp = 0.0
    for n in embed:
        exp += 0.01 * 0.5 * float((p[n] ** 2).sum()) + 0.002 * float(p[n].abs().sum())
    for n in layer:
        exp += 0.03 * 0.5 * float((p[n] ** 2).sum()) + 0.004 * float(p[n].abs().sum())
    for n in cross:  # base_model.py:158-166: un-squared Frobenius norm
        exp += 0.05 * math.sqrt(float((p[n] ** 2).sum())) + 0.006 * float(p[n].abs().sum())
    assert math.isclose(float(X.regular_loss(p, hp)), exp, rel_tol=1e-12)


def test_adam_and_adagrad_known_answer():
    """TF1 formulas by hand for two steps on a scalar."""
    hp = make_hp(optimizer="adam", learning_rate=0.1)
    p = {"w": torch.tensor([1.0], dtype=torch.float64)}
    st = X.new_opt_state(p, hp)
    m = v = 0.0
    w = 1.0
    for t, g in enumerate([0.5, -0.25], start=1):
        st["step"] = t
        X._apply("w", p["w"], torch.tensor([g], dtype=torch.float64), st, hp)
        m = 0.9 * m + 0.1 * g
        v = 0.999 * v + 0.001 * g * g
        w -= 0.1 * math.sqrt(1 - 0.999**t) / (1 - 0.9**t) * m / (math.sqrt(v) + 1e-8)
        assert math.isclose(p["w"].item(), w, rel_tol=1e-12)
    hp = make_hp(optimizer="adagrad", learning_rate=0.1)
    p = {"w": torch.tensor([1.0], dtype=torch.float64)}
    st = X.new_opt_state(p, hp)
    acc, w = 0.1, 1.0
    for g in (0.5, -0.25):
        X._apply("w", p["w"], torch.tensor([g], dtype=torch.float64), st, hp)
        acc += g * g
        w -= 0.1 * g / math.sqrt(acc)
        assert math.isclose(p["w"].item(), w, rel_tol=1e-12)


def test_train_step_dense_vs_sparse_rows_agree_for_sgd_without_regularisers():
    rng = np.random.default_rng(4)
    hp = make_hp(optimizer="sgd")
    b = random_batch(rng, hp, 32)
    pa = X.init_params(hp, seed=2, dtype=torch.float64)
    pb = {k: v.clone() for k, v in pa.items()}
    X.train_step(pa, X.new_opt_state(pa, hp), b, hp, mode="dense_exact")
    X.train_step(pb, X.new_opt_state(pb, hp), b, hp, mode="sparse_rows")
    for k in pa:
        np.testing.assert_allclose(pa[k].numpy(), pb[k].numpy(), rtol=1e-12, atol=1e-15)


def test_training_reduces_loss():
    rng = np.random.default_rng(5)
    hp = make_hp(learning_rate=0.01, embed_l2=0.001, layer_l2=0.001, cross_l2=0.001)
    b = random_batch(rng, hp, 64)
    p = X.init_params(hp, seed=2, dtype=torch.float32)
    st = X.new_opt_state(p, hp)
    losses = [X.train_step(p, st, b, hp)[1] for _ in range(15)]
    assert losses[-1] < losses[0] * 0.9
