"""Drop-in ``XDeepFMModel`` (reference ``recommenders/models/deeprec/models/xDeepFM.py``).

``XDeepFMModel(hparams, FFMTextIterator, seed=...)`` builds the same model the reference's
``_build_graph`` (:24-71) describes -- shared embedding table, field-embedding lookup,
linear / FM / CIN / DNN parts summed into the logit -- but as an ``XDeepFMEngine`` that runs
hand-written HIP kernels on MI355X (see ``_engine.py`` and ``include/xdfm.h``).

One additional, optional hyper-parameter is understood:

``embedding_update`` (str, default ``"dense_exact"``)
    ``"dense_exact"``: reference semantics -- the table's gradient is dense (regularisers,
    TF's IndexedSlices densification) and the optimizer moves every row every step.
    ``"sparse_rows"``: only the rows present in the batch are touched by a step (the optimizer
    is fused into the sorted-key segment sum) and the result is still the reference's: the steps
    a row missed -- TF's Adam keeps moving a row on its decaying ``m`` without a gradient -- are
    applied, in closed form, by whoever reads the row next: the gather forms the aged row in
    registers, the update ages the record it rewrites anyway, and rows are written back only
    in front of a checkpoint or a table export (``xdfm_row_age``, ``xdfm_embed_catchup``).  For
    Adam the closed form is within 1e-6 of the row's movement for any gap and learning rate;
    the step count is exact to 2^24 steps.  Needs ``embed_l1 = embed_l2 = 0`` (their gradient
    is dense).
    ``"sparse_lazy"``: row updates without that catch-up (absent rows stand still: a different
    algorithm under Adam / RMSProp / Adadelta; identical for SGD / Adagrad / FTRL).
"""

import numpy as np

from recommenders_amd.models.deeprec.models._engine import XDeepFMEngine
from recommenders_amd.models.deeprec.models.base_model import BaseModel, _Handle

__all__ = ["XDeepFMModel"]


class XDeepFMModel(BaseModel):
    """xDeepFM model

    :Citation:

        J. Lian, X. Zhou, F. Zhang, Z. Chen, X. Xie, G. Sun, "xDeepFM: Combining Explicit
        and Implicit Feature Interactions for Recommender Systems", KDD 2018.
    """

    def _build_graph(self):
        hparams = self.hparams
        self.keep_prob_train = 1 - np.array(hparams.dropout)
        self.keep_prob_test = np.ones_like(hparams.dropout)

        mode = hparams.values().get("embedding_update", "dense_exact")
        self.engine = XDeepFMEngine(hparams, seed=self.seed, embedding_update=mode)
        eng = self.engine

        # the reference's parameter lists (xDeepFM.py:43, :129-130, :267, :288-289, :511-512,
        # :529-530) as views of the engine's device buffers
        self.embedding = eng.table
        self.embed_params.append(eng.table)
        if hparams.use_Linear_part:
            print("Add linear part.")
            self.layer_params.append(eng.lin_w.view(-1, 1))
            self.layer_params.append(eng.views["lin_b"])
        if hparams.use_FM_part:
            print("Add FM part.")
        if hparams.use_CIN_part:
            print("Add CIN part.")
            for name in eng.dense_shapes:
                if name.startswith(("cin_f_", "fcin_")):
                    self.cross_params.append(eng.views[name])
            self.layer_params.append(eng.views["cin_w_out"])
            self.layer_params.append(eng.views["cin_b_out"])
        if hparams.use_DNN_part:
            print("Add DNN part.")
            for name in eng.dense_shapes:
                if name.startswith("dnn_"):
                    self.layer_params.append(eng.views[name])
        return _Handle("logit")
