"""CPU oracle for the xDeepFM path -- TEST INFRASTRUCTURE ONLY.

Nothing under ``recommenders_amd/`` may import this package.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` use it, and
only as the checker / the timed CPU baseline, never as the thing shipped.

Pinning status ("Oracle"):
* ``ffm_iterator_oracle``  -- pinned: matches the reference's FFMTextIterator outputs
  recorded in SURVEY.md section 8(c)(3) and the fixtures in tests/golden/ffm_iterator_*.json.
* ``metrics_oracle``       -- pinned: matches the reference's ``cal_metric`` (imported from
  /root/reference in the build container) on tests/golden/cal_metric_golden.json.
* ``hparams``              -- pinned by tests/golden/hparams_golden.json (same source).
* ``xdeepfm_oracle``       -- PARITY UNPINNED at op level: the arithmetic of the reference
  lives in TensorFlow 2.7 (setup.py:68-69), which is absent from /root/reference and from
  this image, and the reference's tests hold no logits/gradient vectors.  The restatement
  follows xDeepFM.py / base_model.py line by line plus TF's published op semantics
  (SURVEY.md appendix A), and is cross-checked fp64-vs-fp32 and by autograd.
"""
