"""A recommender stand-in with the fit()/recommend() surface the tuner drives, so that the search / bookkeeping /
multi-process logic of ganmf_amd.tune is testable without a GPU.  Its 'score' is a deterministic function of the
hyper-parameters."""
import numpy as np

from ganmf_amd.base import BaseRecommender


class StubGAN(BaseRecommender):
    RECOMMENDER_NAME = "GANMF"      # picks the GANMF search space

    def __init__(self, URM_train, mode="user", seed=0, is_experiment=True):
        super().__init__(URM_train)
        self.mode, self.seed = mode, seed
        self.quality = 0.0
        self.device = 0

    def fit(self, epochs=300, num_factors=10, emb_dim=32, batch_size=32, m=1, d_lr=1e-4, g_lr=1e-4, d_reg=0.0,
            recon_coefficient=0.01, allow_worse=None, freq=None, validation_evaluator=None, validation_set=None,
            sample_every=None, metrics=("MAP",)):
        # best at d_lr = 1e-3, recon = 0.2: a smooth bowl the surrogate can learn
        self.quality = float(np.exp(-(np.log10(d_lr) + 3) ** 2 - 10 * (recon_coefficient - 0.2) ** 2))
        if num_factors == 13:
            raise MemoryError("synthetic OOM")
        stopped_at = 40 + (num_factors % 7) * 5
        return stopped_at if stopped_at < epochs else epochs + 1

    def _compute_item_score(self, user_id_array, items_to_compute=None):
        # popularity ranking blended with noise: better `quality` -> closer to the test distribution
        rng = np.random.RandomState(7)
        pop = np.asarray(self.URM_train.sum(axis=0)).ravel().astype(np.float32)
        noise = rng.rand(len(user_id_array), self.n_items).astype(np.float32) * pop.max()
        return self.quality * pop[None, :] + (1 - self.quality) * noise


class CrashingGAN(StubGAN):
    """Dies the way a HIP abort / GPU fault / OOM kill does -- no Python exception, the process is gone -- on every
    trial whose num_factors is a multiple of 5."""

    def fit(self, **kw):
        if kw.get("num_factors", 10) % 5 == 0:
            import os
            os._exit(134)
        return super().fit(**kw)
