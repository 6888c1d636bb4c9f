import sys, ctypes; sys.path.insert(0,'/root/repo')
import numpy as np, torch
from oracle import xdeepfm_oracle as X
from recommenders_amd.models.deeprec.models._engine import XDeepFMEngine
from tests.test_gpu_kernels import CONFIGS
from tests.util import make_hp, random_batch
for B in (97, 512):
    rng = np.random.default_rng(5)
    hp = make_hp(**CONFIGS["wide_shape"])
    b = random_batch(rng, hp, B)
    p32 = X.init_params(hp, seed=7)
    for k in p32:
        if k.endswith("_b") or "_b_" in k: p32[k] = p32[k] + 0.05
    p64 = {k: v.double().requires_grad_(True) for k, v in p32.items()}
    logit, _ = X.forward(p64, b, hp); pred = X.pred_from_logit(logit, hp)
    dl = X.data_loss(logit, pred, torch.as_tensor(b["labels"]).double(), hp); dl.backward()
    for dt in ("fp32","bf16"):
        eng = XDeepFMEngine(hp, compute_dtype=dt)
        eng.load_state({k: v.detach().clone() for k, v in p32.items()})
        db = eng.upload(b["labels"], b["dnn_feat_indices"], b["dnn_feat_values"], b["dnn_feat_weights"])
        eng.forward(db, with_loss=True, with_grad=True); eng.backward(db); torch.cuda.synchronize()
        out=[]
        def rel(g,e):
            g=np.asarray(g,np.float64); e=np.asarray(e,np.float64)
            return np.abs(g-e).max()/np.abs(e).max(), np.linalg.norm(g-e)/np.linalg.norm(e)
        out.append(("logit",)+rel(eng.ws["logit"][:B].cpu().numpy(), logit.detach().numpy()[:,0]))
        for k in eng.dense_shapes:
            out.append((k,)+rel(eng.grad_views[k].cpu().numpy(), p64[k].grad.numpy().reshape(eng.dense_shapes[k])))
        print("B",B,dt," ".join("%s:%.1e/%.1e"%o for o in out))
