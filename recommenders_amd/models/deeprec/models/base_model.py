"""Drop-in ``BaseModel`` surface of the deeprec models, for the xDeepFM path.

Mirrors reference ``recommenders/models/deeprec/models/base_model.py``:
constructor (:20-73), ``train/eval/infer`` (:367-416), ``load_model`` (:418-434),
``fit`` (:436-540), ``group_labels`` (:542-566), ``run_eval`` (:568-595),
``predict`` (:597-618).  What was a TensorFlow graph + session is a device engine
(``_engine.XDeepFMEngine``) that launches the HIP kernels of libxdfm.so; ``graph`` and
``sess`` are accepted / exposed as inert handles.

Differences that are visible on purpose:
* ``train`` returns device scalars (0-dim tensors) for the losses, so the training loop does
  not synchronise with the GPU every step; ``float(x)`` / ``"{:.4f}".format(x)`` work on them.
* out-of-range field / feature indices raise ``ValueError`` at the next synchronisation
  point instead of TensorFlow's ``InvalidArgumentError``.
* checkpoints are ``torch.save`` state dicts (``epoch_<n>.pt``), not TF checkpoints.
"""

import abc
import os
import time
import weakref
from os.path import join

import numpy as np
import torch

from recommenders_amd.models.deeprec.deeprec_utils import cal_metric

__all__ = ["BaseModel"]


class _Handle(object):
    """Inert stand-in for a graph tensor / op attribute (``model.logit``, ``model.update``...)."""

    def __init__(self, name):
        self.name = name

    def __repr__(self):
        return "<xdfm handle {0}>".format(self.name)


class _Session(object):
    """Inert stand-in for ``tf.compat.v1.Session``: the engine is the runtime."""

    def __init__(self, engine):
        self.engine = engine

    def close(self):
        pass


class _Saver(object):
    """``tf.compat.v1.train.Saver`` stand-in: saves / restores every variable and the
    optimizer slots (the reference's Saver stores all global variables, base_model.py:60)."""

    def __init__(self, engine):
        self.engine = engine

    @staticmethod
    def _path(save_path):
        return save_path if save_path.endswith(".pt") else save_path + ".pt"

    def save(self, sess=None, save_path=None):
        sd = {k: (v.detach().cpu() if torch.is_tensor(v) else v) for k, v in self.engine.state_dict().items()}
        torch.save(sd, self._path(save_path))
        return save_path

    def restore(self, sess=None, save_path=None):
        sd = torch.load(self._path(save_path), map_location="cpu", weights_only=True)
        self.engine.load_state(sd)


class BaseModel:
    """Base class for models"""

    def __init__(self, hparams, iterator_creator, graph=None, seed=None):
        self.seed = seed
        np.random.seed(seed)

        self.graph = graph
        self.iterator = iterator_creator(hparams, self.graph)
        self.train_num_ngs = hparams.train_num_ngs if "train_num_ngs" in hparams.values() else None

        self.hparams = hparams
        self.layer_params = []
        self.embed_params = []
        self.cross_params = []
        self.layer_keeps = _Handle("layer_keeps")
        self.keep_prob_train = None
        self.keep_prob_test = None
        self.is_train_stage = _Handle("is_training")
        self.group = _Handle("group")

        self.initializer = hparams.init_method
        self.logit = self._build_graph()
        self.pred = self._get_pred(self.logit, self.hparams.method)
        self.loss = self._get_loss()
        self.saver = _Saver(self.engine)
        self.update = _Handle("update")
        self.extra_update_ops = []
        self.init_op = _Handle("init")
        self.merged = None
        self.sess = _Session(self.engine)
        # captured training steps by batch shape (_train_graphed)
        self._graphed = {}
        self._shape_last, self._shape_run = None, 0
        self._graph_after = int(os.environ.get("XDFM_FIT_GRAPH_AFTER", "4"))

    @abc.abstractmethod
    def _build_graph(self):
        """Subclass will implement this: create ``self.engine`` and return the logit handle."""

    def _get_loss(self):
        self.data_loss = _Handle("data_loss")
        self.regular_loss = _Handle("regular_loss")
        return _Handle("loss")

    def _get_pred(self, logit, task):
        if task not in ("regression", "classification"):
            raise ValueError(
                "method must be regression or classification, but now is {0}".format(task)
            )
        return _Handle("pred")

    # ------------------------------------------------------------ step calls
    def _batch_from_feed(self, feed_dict):
        it = self.iterator
        cached = getattr(feed_dict, "device_batch", None)
        if cached is not None:
            return cached
        batch = self.engine.upload(
            feed_dict[it.labels],
            feed_dict[it.dnn_feat_indices],
            feed_dict[it.dnn_feat_values],
            feed_dict[it.dnn_feat_weights],
        )
        # The iterator may keep this feed dict for later epochs (XDFM_FFM_CACHE_MB); the batch in
        # device memory stays attached to it only inside a budget of its own, XDFM_FFM_CACHE_DEVICE_MB
        # (default 8192, 0 = never keep device batches): beyond it the batch is uploaded again.
        nbytes = sum(t.numel() * t.element_size() for t in
                     (batch.labels, batch.ids, batch.weights, batch.row_ptr, batch.tok_row))
        nbytes += 16 * batch.nnz  # the sort plan the first training step attaches
        budget = int(os.environ.get("XDFM_FFM_CACHE_DEVICE_MB", "8192")) << 20
        held = getattr(self, "_device_cache_bytes", 0)
        if held + nbytes <= budget:
            try:
                feed_dict.device_batch = batch
            except AttributeError:
                return batch  # a plain dict: nothing to cache on
            self._device_cache_bytes = held + nbytes
            weakref.finalize(feed_dict, self._release_device_cache, nbytes)
        return batch

    def _release_device_cache(self, nbytes):
        self._device_cache_bytes = max(0, getattr(self, "_device_cache_bytes", 0) - nbytes)

    def train(self, sess, feed_dict):
        """One optimisation step on the batch in ``feed_dict``.

        Returns ``[update, extra_update_ops, loss, data_loss, summary]`` like the reference
        (:379-388); the two losses are device scalars.
        """
        feed_dict[self.layer_keeps] = self.keep_prob_train
        feed_dict[self.is_train_stage] = True
        batch = self._batch_from_feed(feed_dict)
        out = self._train_graphed(feed_dict, batch)
        if out is None:
            out = self.engine.train_step(batch)
        loss, data_loss = out
        return [None, [], loss, data_loss, None]

    def _train_graphed(self, feed_dict, batch):
        """The step as a replay of a captured HIP graph when batches of one shape (samples, tokens) keep
        coming -- Criteo-style files where every line names every field -- and as eager launches
        otherwise (the eager step is bound by the host's launch rate: 0.66 against 0.42 ms at batch 8192).
        A shape is captured after XDFM_FIT_GRAPH_AFTER consecutive batches of it (default 4; 0 = never);
        the capture trains this batch as its warm-up.
        Returns None when this step should run eagerly."""
        eng = self.engine
        after = self._graph_after
        if after <= 0 or type(eng).__name__ != "XDeepFMEngine" or eng.kernel_events is not None:
            return None
        key = (batch.B, batch.nnz)
        self._shape_run = self._shape_run + 1 if key == self._shape_last else 1
        self._shape_last = key
        gs = self._graphed.get(key)
        if gs is not None and gs._ws_token != eng._ws_token:
            # a larger batch made the engine re-allocate its workspace: the capture holds stale addresses
            del self._graphed[key]
            gs = None
        if gs is None:
            if self._shape_run < after or len(self._graphed) >= 2 or batch.nnz == 0:
                return None
            from recommenders_amd.models.deeprec.models._engine import GraphedStep

            gs = GraphedStep(eng, batch.labels, None, batch.ids, batch.weights, csr=(batch.row_ptr, batch.tok_row))
            self._graphed[key] = gs
            return gs.warmup_loss  # the capture's warm-up WAS this batch's training step
        pb = getattr(feed_dict, "packed_batch", None)
        if pb is None or pb[0] is not gs:
            pb = (gs, gs.pack_batch(batch))
            nbytes = pb[1].buf.numel()
            budget = int(os.environ.get("XDFM_FFM_CACHE_DEVICE_MB", "8192")) << 20
            held = getattr(self, "_device_cache_bytes", 0)
            if getattr(feed_dict, "device_batch", None) is batch and held + nbytes <= budget:
                # kept for later epochs like the device batch, inside the same budget
                feed_dict.packed_batch = pb
                self._device_cache_bytes = held + nbytes
                weakref.finalize(feed_dict, self._release_device_cache, nbytes)
        loss, data_loss = gs.run_packed(pb[1])
        # the graph writes the same two tensors every step: hand out copies, as the eager step hands out
        # new tensors (a caller may keep them across steps)
        pair = torch.stack([loss, data_loss])
        return pair[0], pair[1]

    def eval(self, sess, feed_dict):
        """``[pred [B,1], labels [B,1]]`` as numpy arrays (reference :390-402)."""
        feed_dict[self.layer_keeps] = self.keep_prob_test
        feed_dict[self.is_train_stage] = False
        batch = self._batch_from_feed(feed_dict)
        pred = self.engine.forward(batch, with_loss=False)
        out = pred.detach().cpu().numpy().reshape(-1, 1)
        self.engine.check_errors()
        return [out, np.asarray(feed_dict[self.iterator.labels], dtype=np.float32).reshape(-1, 1)]

    def infer(self, sess, feed_dict):
        """``[pred [B,1]]`` (reference :404-416)."""
        feed_dict[self.layer_keeps] = self.keep_prob_test
        feed_dict[self.is_train_stage] = False
        batch = self._batch_from_feed(feed_dict)
        pred = self.engine.forward(batch, with_loss=False)
        out = pred.detach().cpu().numpy().reshape(-1, 1)
        self.engine.check_errors()
        return [out]

    def load_model(self, model_path=None):
        """Load an existing model; ``IOError`` if the restore fails (reference :418-434)."""
        act_path = self.hparams.load_saved_model
        if model_path is not None:
            act_path = model_path
        self._graphed.clear()  # captured steps are rebuilt on the restored state
        try:
            self.saver.restore(self.sess, act_path)
        except Exception:
            raise IOError("Failed to find any matching files for {0}".format(act_path))

    # ------------------------------------------------------------------- fit
    def fit(self, train_file, valid_file, test_file=None):
        """Train on ``train_file``; evaluate ``valid_file`` (and ``test_file``) every epoch.

        Prints the same per-epoch lines as the reference (:513-535).  Returns ``self``.
        """
        # write_tfevents (:448-450, :467-468, :537-538): the reference hands its merged summary (the
        # scalars data_loss, regular_loss, loss of _add_summaries) to a TF FileWriter every step.
        # Without TensorFlow the same scalars go to SUMMARIES_DIR/events.jsonl, one JSON object per
        # step (the step count restarts every epoch there as well, so the epoch is recorded too);
        # they are fetched from the device once per epoch, not once per step.
        events = None
        if self.hparams.write_tfevents:
            os.makedirs(self.hparams.SUMMARIES_DIR, exist_ok=True)
            events = open(join(self.hparams.SUMMARIES_DIR, "events.jsonl"), "a")

        train_sess = self.sess
        for epoch in range(1, self.hparams.epochs + 1):
            step = 0
            self.hparams.current_epoch = epoch

            epoch_loss = 0
            logged = []
            train_start = time.time()
            for batch_data_input, impression, data_size in self.iterator.load_data_from_file(train_file):
                step_result = self.train(train_sess, batch_data_input)
                (_, _, step_loss, step_data_loss, summary) = step_result
                epoch_loss = epoch_loss + step_loss
                step += 1
                if events is not None:
                    logged.append(torch.stack([step_loss, step_data_loss]))
                if step % self.hparams.show_step == 0:
                    print(
                        "step {0:d} , total_loss: {1:.4f}, data_loss: {2:.4f}".format(
                            step, float(step_loss), float(step_data_loss)
                        )
                    )
            epoch_loss = float(epoch_loss)  # synchronises once per epoch
            self.engine.check_errors()
            if events is not None and logged:
                import json

                for k, (tot, dat) in enumerate(torch.stack(logged).cpu().tolist(), 1):
                    events.write(json.dumps({"epoch": epoch, "step": k, "loss": tot, "data_loss": dat,
                                             "regular_loss": tot - dat}) + "\n")
                events.flush()
            train_end = time.time()
            train_time = train_end - train_start

            if self.hparams.save_model:
                if not os.path.exists(self.hparams.MODEL_DIR):
                    os.makedirs(self.hparams.MODEL_DIR)
                if epoch % self.hparams.save_epoch == 0:
                    save_path_str = join(self.hparams.MODEL_DIR, "epoch_" + str(epoch))
                    self.saver.save(sess=train_sess, save_path=save_path_str)

            eval_start = time.time()
            eval_res = self.run_eval(valid_file)
            train_info = ",".join(
                [str(item[0]) + ":" + str(item[1]) for item in [("logloss loss", epoch_loss / step)]]
            )
            eval_info = ", ".join(
                [str(item[0]) + ":" + str(item[1]) for item in sorted(eval_res.items(), key=lambda x: x[0])]
            )
            if test_file is not None:
                test_res = self.run_eval(test_file)
                test_info = ", ".join(
                    [str(item[0]) + ":" + str(item[1]) for item in sorted(test_res.items(), key=lambda x: x[0])]
                )
            eval_end = time.time()
            eval_time = eval_end - eval_start

            if test_file is not None:
                print(
                    "at epoch {0:d}".format(epoch)
                    + "\ntrain info: " + train_info
                    + "\neval info: " + eval_info
                    + "\ntest info: " + test_info
                )
            else:
                print(
                    "at epoch {0:d}".format(epoch)
                    + "\ntrain info: " + train_info
                    + "\neval info: " + eval_info
                )
            print(
                "at epoch {0:d} , train time: {1:.1f} eval time: {2:.1f}".format(
                    epoch, train_time, eval_time
                )
            )
        if events is not None:
            events.close()
        return self

    def group_labels(self, labels, preds, group_keys):
        """Divide ``labels`` and ``preds`` into groups by ``group_keys`` (reference :542-566)."""
        all_keys = list(set(group_keys))
        group_labels = {k: [] for k in all_keys}
        group_preds = {k: [] for k in all_keys}
        for label, p, k in zip(labels, preds, group_keys):
            group_labels[k].append(label)
            group_preds[k].append(p)
        return [group_labels[k] for k in all_keys], [group_preds[k] for k in all_keys]

    def run_eval(self, filename):
        """Evaluate ``filename``; returns ``{metric: value rounded to 4 decimals}``.

        ``auc`` / ``logloss`` come from the fused device kernel over the whole file's
        predictions (they stay in HBM); any other requested metric goes through
        ``cal_metric`` on the host like the reference (:588).
        """
        eng = self.engine
        preds, labels, imp_indexs = [], [], []
        for batch_data_input, imp_index, data_size in self.iterator.load_data_from_file(filename):
            batch_data_input[self.layer_keeps] = self.keep_prob_test
            batch_data_input[self.is_train_stage] = False
            batch = self._batch_from_feed(batch_data_input)
            preds.append(eng.forward(batch, with_loss=False).clone())
            labels.append(batch.labels)
            imp_indexs.extend(np.reshape(imp_index, -1))
        metrics = self.hparams.metrics
        if not preds:
            raise ValueError("no samples in {0}".format(filename))
        all_preds = torch.cat(preds)
        all_labels = torch.cat(labels)
        res = {}
        host_metrics = [m for m in metrics if m not in ("auc", "logloss")]
        if any(m in ("auc", "logloss") for m in metrics):
            auc, logloss, n_pos = eng.eval_metrics(all_preds, all_labels)
            eng.check_errors()
            if "auc" in metrics:
                if n_pos == 0 or n_pos == all_preds.numel():
                    raise ValueError(
                        "Only one class present in y_true. ROC AUC score is not defined in that case."
                    )
                res["auc"] = round(auc, 4)
            if "logloss" in metrics:
                res["logloss"] = round(logloss, 4)
        else:
            eng.check_errors()
        host_preds = host_labels = None
        if host_metrics or "pairwise_metrics" in self.hparams.values():
            host_preds = list(all_preds.cpu().numpy())
            host_labels = list(all_labels.cpu().numpy())
        if host_metrics:
            res.update(cal_metric(host_labels, host_preds, host_metrics))
        res = {m: res[m] for m in res}  # insertion order: device metrics then host metrics
        if "pairwise_metrics" in self.hparams.values():
            group_labels, group_preds = self.group_labels(host_labels, host_preds, imp_indexs)
            res.update(cal_metric(group_labels, group_preds, self.hparams.pairwise_metrics))
        return res

    def predict(self, infile_name, outfile_name):
        """Write one predicted score per line of ``infile_name`` to ``outfile_name``
        (reference :597-618: ``str(np.float32)`` per line, newline after every batch)."""
        load_sess = self.sess
        with open(outfile_name, "w") as wt:
            for batch_data_input, _, data_size in self.iterator.load_data_from_file(infile_name):
                step_pred = self.infer(load_sess, batch_data_input)
                step_pred = step_pred[0][:data_size]
                step_pred = np.reshape(step_pred, -1)
                wt.write("\n".join(map(str, step_pred)))
                wt.write("\n")
        return self
