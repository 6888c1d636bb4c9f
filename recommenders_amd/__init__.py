"""MI355X-native xDeepFM training path (drop-in for recommenders.models.deeprec xDeepFM).

    from recommenders_amd.models.deeprec.deeprec_utils import prepare_hparams
    from recommenders_amd.models.deeprec.models.xDeepFM import XDeepFMModel
    from recommenders_amd.models.deeprec.io.iterator import FFMTextIterator
"""

__version__ = "0.1.0"
