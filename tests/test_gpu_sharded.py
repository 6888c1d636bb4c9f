"""ShardedXDeepFMEngine on one GPU: with world size 1 the exchange is the identity, so a step
must reproduce the plain engine (same kernels, rows fetched through the owner path) and the
oracle.  The cross-rank routing itself is covered on CPU by tests/test_sharded_exchange.py;
real multi-GPU runs are the driver's scaling bench."""
import numpy as np
import pytest
import torch

from oracle import xdeepfm_oracle as X
from recommenders_amd.models.deeprec.models._engine import XDeepFMEngine
from recommenders_amd.models.deeprec.models._sharded import ShardedXDeepFMEngine
from tests.util import make_hp, random_batch

pytestmark = pytest.mark.gpu


def _upload(eng, b):
    return eng.upload(b["labels"], b["dnn_feat_indices"], b["dnn_feat_values"], b["dnn_feat_weights"])


@pytest.mark.parametrize("virtual_world", [1, 3])
@pytest.mark.parametrize("optimizer", ["adam", "adagrad"])
def test_sharded_step_equals_plain_engine_and_oracle(optimizer, virtual_world):
    """virtual_world = 3: tokens are partitioned as for 3 owners (all of them this process), so
    the kernels see the permuted token order and the inverse map of a real multi-rank step."""
    hp = make_hp(optimizer=optimizer, learning_rate=0.01, FEATURE_COUNT=300, cross_l2=0.01, cross_l1=0.001)
    rng = np.random.default_rng(5)
    p32 = X.init_params(hp, seed=9)
    plain = XDeepFMEngine(hp, embedding_update="sparse_rows")
    shard = ShardedXDeepFMEngine(hp, virtual_world=virtual_world)
    assert shard.world == 1 and shard.N == 300
    plain.load_state({k: v.clone() for k, v in p32.items()})
    shard.load_global_state({k: v.clone() for k, v in p32.items()})
    p64 = {k: v.double() for k, v in p32.items()}
    st = X.new_opt_state(p64, hp)
    for step in range(3):
        b = random_batch(rng, hp, 64 if step != 1 else 33)
        l0, d0 = plain.train_step(_upload(plain, b))
        l1, d1 = shard.train_step(_upload(shard, b))
        o_loss, o_dl, _, _ = X.train_step(p64, st, b, hp, mode="sparse_rows")
        assert abs(float(d1) - o_dl) <= 1e-5 * abs(o_dl)
        assert abs(float(l1) - o_loss) <= 1e-5 * abs(o_loss)
        # same arithmetic in the same order: the two engines agree to the bit on the forward
        assert float(d0) == float(d1)
    torch.cuda.synchronize()
    shard.check_errors()
    table, lin = shard.gather_global_table()
    tol = 2e-4 if optimizer == "adam" else 2e-5
    for got, name in ((table, "embedding"), (lin.view(-1, 1), "lin_w")):
        exp = p64[name].numpy()
        err = np.abs(got.cpu().numpy() - exp).max()
        assert err <= tol * np.abs(exp).max(), (name, err)
    for k in shard.dense_shapes:
        exp = p64[k].numpy().reshape(shard.dense_shapes[k])
        err = np.abs(shard.views[k].cpu().numpy() - exp).max()
        assert err <= tol * max(np.abs(exp).max(), 1e-30), (k, err)
    # plain engine took the same steps
    assert np.abs(plain.table.cpu().numpy() - table.cpu().numpy()).max() <= 1e-6
