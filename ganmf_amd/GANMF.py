"""GANMF recommender — host mirror of the reference class (GANRec/GANMF.py:23-342).

Same constructor, fit() keyword arguments, return-value quirk, scoring and early-stopping hooks
as the reference, so it can stand in for `GANRec.GANMF.GANMF` under RecSysExp.py /
RunBestParameters.py.  Every number is produced by libganmf_hip.so (HIP kernels on gfx950)
through the C ABI in include/ganmf_hip.h; this file holds only the epoch loop and bookkeeping.
"""
import os
import pickle
import time
from datetime import datetime

import numpy as np
import scipy.sparse as sps

from . import _lib as L
from .base import BaseRecommender
from .early_stopping import EarlyStoppingScheduler
from .engine import Engine

try:  # progress bar is cosmetic (GANMF.py:170,234)
    import tqdm
except Exception:  # pragma: no cover
    tqdm = None


def glorot_uniform(rng, shape):
    """tf.glorot_uniform_initializer for [a, b] variables (GANMF.py:57): U(-L, L), L = sqrt(6/(a+b)).
    TF's seeded Philox stream cannot be reproduced without TF; the build's documented default is
    numpy RandomState(seed) drawing, in order, encoding/kernel, decoding/kernel, user_embeddings,
    item_embeddings (biases are zero, tf.layers.dense default)."""
    limit = np.sqrt(6.0 / (shape[0] + shape[1]))
    return rng.uniform(-limit, limit, size=shape).astype(np.float32)


class _TensorRef(object):
    """Stands where the reference keeps tf.Variable objects in model.params (GANMF.py:119-122)."""

    def __init__(self, tid, name):
        self.tid, self.name = tid, name


class _SessionShim(object):
    """model.sess.run(var) -> ndarray, the only use callers make of the session (Utils_.py:305-308)."""

    def __init__(self, model):
        self._model = model

    def run(self, fetch):
        if isinstance(fetch, (list, tuple)):
            return [self.run(f) for f in fetch]
        if not isinstance(fetch, _TensorRef):
            raise TypeError("only parameter handles can be fetched")
        return self._model._get(fetch.tid)


class GANMF(BaseRecommender):
    RECOMMENDER_NAME = 'GANMF'

    # (tensor id, reference variable name, attribute for shape)
    _D_TENSORS = ((0, 'autoencoder/encoding/kernel'), (1, 'autoencoder/encoding/bias'),
                  (2, 'autoencoder/decoding/kernel'), (3, 'autoencoder/decoding/bias'))
    _G_TENSORS = ((L.T_USER_EMB, 'generator/user_embeddings'), (L.T_ITEM_EMB, 'generator/item_embeddings'))

    SCORE_CONTRACTS = ("ganmf", "mf")

    def __init__(self, URM_train, mode='user', verbose=False, seed=1234, is_experiment=False, device=0, devices=None,
                 dist_backend=None, world_size=None, score_contract=None):
        """`score_contract` (beyond the reference's signature; also GANMF_SCORE_CONTRACT in the environment) selects what
        `_compute_item_score` / `recommend` / the device evaluator return for the two corners the reference's classes differ in:
          "ganmf" (default) = GANRec/GANMF.py:285-292 exactly: the class derives from BaseRecommender, every user -- with or
                    without a training interaction -- gets the finite scores U[ids] . V^T, and `items_to_compute` is accepted
                    and ignored;
          "mf"    = Base/BaseMatrixFactorizationRecommender.py:113-119,128-143, the contract the north star names: only
                    `items_to_compute` keep their scores (the others -inf) and users without a training interaction score
                    -inf everywhere (empty recommendation lists).
        `devices` (beyond the reference's signature; also GANMF_DEVICES="0,1,2,3" in the environment, so that the
        reference's drivers need no change): fit() shards the generator's users row-wise over these GPUs, one rank process
        per GPU over RCCL (ganmf_amd/dist.py ShardedEngine; north star: "users shard row-wise across the 8 GPUs of one
        node").  dist_backend="local" + world_size=N (GANMF_DIST_BACKEND / GANMF_WORLD_SIZE) puts N ranks on ONE GPU over the
        library's loopback communicator (tests).  Default: one GPU, `device`."""
        if mode not in ['user', 'item']:
            raise ValueError('Accepted training modes are `user` and `item`. Given was {}.', mode)
        self.mode = mode
        # the reference does not call super().__init__ (GANMF.py:26-51); keep its attributes
        URM_train = sps.csr_matrix(URM_train, dtype=np.float32)
        self._URM_eval = URM_train                       # user x item, what evaluators see
        self._URM_fit = URM_train.T.tocsr() if mode == 'item' else URM_train   # training orientation
        self.URM_train = self._URM_fit                   # GANMF.py:31-35
        self.num_users, self.num_items = self.URM_train.shape
        self.n_users, self.n_items = URM_train.shape
        self.config = None
        self.seed = seed
        self.verbose = verbose
        self.device = device
        if devices is None and os.environ.get("GANMF_DEVICES"):
            devices = [int(d) for d in os.environ["GANMF_DEVICES"].split(",") if d.strip() != ""]
        self.devices = list(devices) if devices is not None else None
        self.dist_backend = dist_backend or os.environ.get("GANMF_DIST_BACKEND") or "process"
        self.world_size = world_size if world_size is not None else (int(os.environ["GANMF_WORLD_SIZE"]) if os.environ.get("GANMF_WORLD_SIZE") else None)
        self.score_contract = score_contract or os.environ.get("GANMF_SCORE_CONTRACT") or "ganmf"
        if self.score_contract not in self.SCORE_CONTRACTS:
            raise ValueError("score_contract must be one of %r, given %r" % (self.SCORE_CONTRACTS, self.score_contract))
        self.logsdir = os.path.join('plots', self.RECOMMENDER_NAME, datetime.now().strftime("%Y%m%d-%H%M%S"))
        self.is_experiment = is_experiment
        if not self.is_experiment:
            os.makedirs(self.logsdir, exist_ok=True)
        self.items_to_ignore_flag = False
        self.items_to_ignore_ID = np.array([], dtype=int)
        self.filterTopPop = False
        self.filterTopPop_ItemsID = np.array([], dtype=int)
        self.mfma = None                # None | "f32" | "bf16" | "f16": arithmetic of the GEMM K loops (engine.Engine)
        self.initial_weights = None     # optional dict {We,be,Wd,bd,U,V}: explicit init (parity tests)
        self.schedule_rng = None        # optional numpy RandomState for the per-epoch shuffle instead of numpy's GLOBAL stream (the
                                        # reference's, GANMF.py:175): RandomState(s) draws what np.random.seed(s) would, and lets
                                        # several fits run in threads of one process (tune.py: engines_per_worker)
        self.engine = None
        self.params = None
        self.sess = None
        self._stop_training = False
        self.train_d_loss, self.train_g_loss = [], []

    # ---- engine plumbing -----------------------------------------------------------------------
    _NAME2ID = {"We": 0, "be": 1, "Wd": 2, "bd": 3, "U": L.T_USER_EMB, "V": L.T_ITEM_EMB}

    def _get(self, tid):
        a = self.engine.get_tensor(tid)
        return a[0] if tid in (1, 3) else a

    def _build(self, num_factors, emb_dim, batch_size, **hp):
        self.num_factors, self.emb_dim = num_factors, emb_dim
        if self.engine is not None:
            self.engine.close()
        self.engine = self._make_engine(num_factors, emb_dim, batch_size, mfma=self.mfma, **hp)
        self.engine.set_urm(self._URM_fit)
        self.engine.set_seen(self._URM_eval)            # for device-side recommend()
        self._reset_score_filter()
        self.params = {'D': [_TensorRef(t, n) for t, n in self._D_TENSORS],
                       'G': [_TensorRef(t, n) for t, n in self._G_TENSORS]}
        self.sess = _SessionShim(self)

    def _reset_score_filter(self):
        """no item filter; the cold-user mask only under the MF contract (the reference's GANMF scores every user)"""
        self.engine.set_score_filter(None, mask_cold=(self.score_contract == "mf"))

    def _sharded(self):
        if self.dist_backend == "local":
            return (self.world_size or 1) > 1
        return self.devices is not None and len(self.devices) > 1

    def _make_engine(self, num_factors, width, batch_size, **kw):
        """One Engine on `device`, or the row-sharded group (same methods) when several devices / ranks were asked for."""
        if not self._sharded():
            # a single entry in `devices` / GANMF_DEVICES names THE device (it used to be ignored in favour of `device`)
            dev = self.devices[0] if (self.devices is not None and len(self.devices) == 1 and self.dist_backend != "local") else self.device
            return Engine(self.num_users, self.num_items, num_factors, width, batch_size, device=dev, **kw)
        from .dist import ShardedEngine
        if self.dist_backend == "local":
            return ShardedEngine(self.num_users, self.num_items, num_factors, width, batch_size, world_size=self.world_size,
                                 devices=[self.devices[0] if self.devices else self.device], backend="local", **kw)
        return ShardedEngine(self.num_users, self.num_items, num_factors, width, batch_size, devices=self.devices,
                             backend="process", **kw)

    def _init_weights(self):
        if self.initial_weights is not None:
            w = self.initial_weights
        else:
            rng = np.random.RandomState(self.seed)
            w = {"We": glorot_uniform(rng, (self.num_items, self.emb_dim)),
                 "be": np.zeros(self.emb_dim, np.float32),
                 "Wd": glorot_uniform(rng, (self.emb_dim, self.num_items)),
                 "bd": np.zeros(self.num_items, np.float32),
                 "U": glorot_uniform(rng, (self.num_users, self.num_factors)),
                 "V": glorot_uniform(rng, (self.num_items, self.num_factors))}
        for name, tid in self._NAME2ID.items():
            self.engine.set_tensor(tid, w[name])

    # ---- fit (GANMF.py:88-244) -------------------------------------------------------------------
    def fit(self, num_factors=10, emb_dim=32, epochs=300, batch_size=32, d_lr=1e-4, g_lr=1e-4, d_steps=1, g_steps=1,
            d_reg=0, g_reg=0, m=1, recon_coefficient=1e-2, allow_worse=None, freq=None, after=0, metrics=['MAP'],
            sample_every=None, validation_evaluator=None, validation_set=None):
        self.config = dict(locals())
        del self.config['self']

        self._build(num_factors, emb_dim, batch_size, d_lr=d_lr, g_lr=g_lr, d_reg=d_reg, g_reg=g_reg, m=m,
                    recon_coefficient=recon_coefficient)
        self._init_weights()

        return self._epoch_loop(epochs, d_steps, g_steps, allow_worse, freq, after, metrics, sample_every,
                                validation_evaluator, validation_set)

    def _epoch_loop(self, epochs, d_steps, g_steps, allow_worse, freq, after, metrics, sample_every,
                    validation_evaluator, validation_set):
        """Epochs 1..`epochs` shared by GANMF and DisGANMF (schedule of GANMF.py:151-244, DisGANMF.py:152-244).

        Per epoch: ONE in-place shuffle of the row ids on numpy's global stream (cumulative across epochs, which is
        what makes a seeded run reproduce the reference's minibatch schedule, GANMF.py:175), ONE library call that
        runs `d_steps` passes of discriminator updates and then `g_steps` passes of generator updates over the same
        slices of that permutation (:176-203), then the optional progress evaluation and the early-stopping hook.
        Returns the epoch early stopping fired at, or `epochs + 1` when it never did (the reference's counter is
        incremented once more before its loop ends, :244 -- callers subtract `allow_worse * freq` from it)."""
        self._stop_training = False
        watcher = None
        if validation_evaluator is not None:
            watcher = EarlyStoppingScheduler(self, evaluator=validation_evaluator, allow_worse=allow_worse, freq=freq,
                                             metrics=metrics, after=after)
        row_ids = np.arange(self.num_users)
        self.train_d_loss, self.train_g_loss = [], []
        progress = tqdm.tqdm(total=epochs, initial=1) if (tqdm is not None and self.verbose) else None
        if self.verbose:
            print('Starting training...')
        fit_t0 = window_t0 = time.time()
        stopped_at = None
        for epoch in range(1, epochs + 1):
            (self.schedule_rng if self.schedule_rng is not None else np.random).shuffle(row_ids)
            d_losses, g_losses = self.engine.train_epoch(row_ids, d_steps, g_steps)
            self.train_d_loss.append(np.mean(d_losses) if len(d_losses) else np.nan)
            self.train_g_loss.append(np.mean(g_losses) if len(g_losses) else np.nan)

            if validation_set is not None and sample_every is not None and epoch % sample_every == 0:
                elapsed = time.time() - window_t0
                print('Epoch : {:d}. Total: {:.2f} secs, {:.2f} secs/epoch.'.format(epoch, elapsed, elapsed / sample_every))
                print(self._evaluate_in_eval_orientation(lambda: validation_evaluator.evaluateRecommender(self)[1]))
                window_t0 = time.time()
            if watcher is not None:
                self._evaluate_in_eval_orientation(lambda: watcher(epoch))
            if progress is not None:
                progress.update()
            if self._stop_training:
                print('Training stopped, epoch:', epoch)
                stopped_at = epoch
                break
        if progress is not None:
            progress.close()
        if self.verbose:
            print('Training took {:.2f} seconds'.format(time.time() - fit_t0))
        self.URM_train = self._URM_eval      # user x item again once fit() returns (GANMF.py:241-242)
        return epochs + 1 if stopped_at is None else stopped_at

    def _evaluate_in_eval_orientation(self, call):
        """Runs `call()` with self.URM_train in user x item orientation (what evaluators and remove-seen expect)
        and switches back to the training orientation afterwards."""
        self._flip_for_evaluation(True)
        try:
            return call()
        finally:
            self._flip_for_evaluation(False)

    def _flip_for_evaluation(self, to_eval):
        """The reference flips self.URM_train with .T.tocsr() around every evaluation in item mode
        (GANMF.py:215-228); both orientations are kept instead of re-transposing."""
        self.URM_train = self._URM_eval if to_eval else self._URM_fit

    # ---- hooks used by EarlyStoppingScheduler (GANMF.py:246-255) --------------------------------
    def stop_fit(self):
        self._stop_training = True

    def save_current_model(self):
        self.engine.snapshot_best()

    def load_model(self):
        self.engine.restore_best()

    def load_weights(self, best_params, weights):
        """GANMF.py:257-283: `weights` is the Utils_.saveWeights dict {'D': [...], 'G': [...]}."""
        self._build(best_params['num_factors'], best_params['emb_dim'], batch_size=32)
        for ref, w in zip(self.params['D'] + self.params['G'], list(weights['D']) + list(weights['G'])):
            self.engine.set_tensor(ref.tid, np.asarray(w, dtype=np.float32))

    # ---- scoring (GANMF.py:285-292) ---------------------------------------------------------------
    def _compute_item_score(self, user_id_array, items_to_compute=None):
        """Scores in evaluation orientation.  score_contract "ganmf" (default; GANMF.py:285-292): U[ids] . V^T for every user,
        `items_to_compute` ignored.  "mf" (BaseMatrixFactorizationRecommender.py:113-119,128-143): `items_to_compute` given ->
        every other item is -inf; users without a training interaction are -inf everywhere; both masks applied on the device."""
        self._require_engine()
        ids = np.asarray(user_id_array).reshape(-1)
        if items_to_compute is None or self.score_contract != "mf":
            return self.engine.scores(ids, transposed=(self.mode == 'item'))
        with self._item_filter(items_to_compute):
            return self.engine.scores(ids, transposed=(self.mode == 'item'))

    def _item_filter(self, items_to_compute):
        """context (MF contract only): scores / recommend / evaluate restricted to `items_to_compute`; the cold-user mask stays on"""
        eng = self.engine
        reset = self._reset_score_filter

        class _Ctx(object):
            def __enter__(self_inner):
                eng.set_score_filter(items_to_compute, mask_cold=True)

            def __exit__(self_inner, *exc):
                reset()
                return False
        return _Ctx()

    # ---- recommend (Base/BaseRecommender.py:155-247) ---------------------------------------------
    _DEVICE_TOPK_MAX = 256   # above this the k-round device selection loses to numpy's argpartition

    def recommend_topk(self, user_id_array, cutoff, remove_seen_flag=True, items_to_compute=None):
        """Top-`cutoff` item ids per user as an [n, cutoff] int32 array, -1 padded where a user has fewer
        finite scores; scores, seen-item mask and selection all stay on the device (ganmf_recommend).
        Cut-offs the device selection does not take (above _DEVICE_TOPK_MAX or above the item count) are ranked by
        the host route and padded the same way."""
        self._require_engine()
        ids = np.atleast_1d(np.asarray(user_id_array)).reshape(-1)
        if 1 <= cutoff <= min(self._DEVICE_TOPK_MAX, self.n_items):
            if items_to_compute is None or self.score_contract != "mf":
                items, _ = self.engine.recommend(ids, cutoff, transposed=(self.mode == 'item'), remove_seen=remove_seen_flag)
            else:
                with self._item_filter(items_to_compute):
                    items, _ = self.engine.recommend(ids, cutoff, transposed=(self.mode == 'item'), remove_seen=remove_seen_flag)
            return items
        lists = self.recommend(ids, cutoff=cutoff, remove_seen_flag=remove_seen_flag, items_to_compute=items_to_compute,
                               return_scores=True)[0]
        items = np.full((len(ids), cutoff), -1, dtype=np.int32)
        for i, row in enumerate(lists):
            items[i, :len(row)] = row
        return items

    def evaluate_on_device(self, evaluator_key, urm_test_sorted, gains, user_id_array, cutoffs, disc, ideal_cum,
                           remove_seen_flag=True):
        """Hold-out metric sums for EvaluatorHoldoutFast without leaving the device (ganmf_evaluate): [len(cutoffs), 9]
        float64 in the order of ganmf_amd._lib.EVAL_METRICS, or None when the device route does not apply (cut-off beyond
        the device selection, too many cut-offs).  The test matrix is uploaded once per evaluator (`evaluator_key`)."""
        from . import _lib as L
        self._require_engine()
        cutoffs = list(cutoffs)
        if not cutoffs or len(cutoffs) > L.EVAL_MAX_CUTOFFS or not (1 <= max(cutoffs) <= min(self._DEVICE_TOPK_MAX, self.n_items)):
            return None
        if min(cutoffs) < 1:
            return None
        # `evaluator_key`: a token the evaluator draws once from a process-wide counter (never id(): ids of freed objects are
        # reused); the engine is compared by identity through a strong reference, so a rebuilt engine uploads again
        held = getattr(self, "_test_on_device", None)
        if held is None or held[0] != evaluator_key or held[1] is not self.engine:
            self.engine.set_test(urm_test_sorted, gains)
            self._test_on_device = (evaluator_key, self.engine)
        return self.engine.evaluate(np.asarray(user_id_array).reshape(-1), cutoffs, disc, ideal_cum,
                                    transposed=(self.mode == 'item'), remove_seen=remove_seen_flag)

    def recommend(self, user_id_array, cutoff=None, remove_seen_flag=True, items_to_compute=None,
                  remove_top_pop_flag=False, remove_CustomItems_flag=False, return_scores=False):
        device_ok = (not return_scores and not remove_top_pop_flag
                     and not remove_CustomItems_flag and cutoff is not None and 1 <= cutoff <= self._DEVICE_TOPK_MAX
                     and cutoff <= self.n_items)
        if not device_ok:   # full score matrix needed on the host: the reference's own route
            saved = self.URM_train
            self.URM_train = self._URM_eval
            try:
                return super(GANMF, self).recommend(user_id_array, cutoff=cutoff, remove_seen_flag=remove_seen_flag,
                                                    items_to_compute=items_to_compute,
                                                    remove_top_pop_flag=remove_top_pop_flag,
                                                    remove_CustomItems_flag=remove_CustomItems_flag,
                                                    return_scores=return_scores)
            finally:
                self.URM_train = saved
        single = np.isscalar(user_id_array)
        items = self.recommend_topk(user_id_array, cutoff, remove_seen_flag, items_to_compute=items_to_compute)
        lists = [row[row >= 0].tolist() for row in items]
        return lists[0] if single else lists

    def _require_engine(self):
        if self.engine is None:
            raise RuntimeError("GANMF: model has no device state; call fit() or loadModel() first")

    def user_factors(self):
        self._require_engine()
        return self._get(L.T_USER_EMB)

    def item_factors(self):
        self._require_engine()
        return self._get(L.T_ITEM_EMB)

    # MF contract named by the north star (BaseMatrixFactorizationRecommender.py:94-143):
    # USER_factors[ids] @ ITEM_factors.T == _compute_item_score(ids), evaluation orientation
    @property
    def USER_factors(self):
        return self.item_factors() if self.mode == 'item' else self.user_factors()

    @property
    def ITEM_factors(self):
        return self.user_factors() if self.mode == 'item' else self.item_factors()

    def autoencoder_codes(self):
        """GANMF.py:304-307: encoding of every training row, URM_train . We + be (off the hot path:
        one sparse product on the host from the fetched encoder)."""
        self._require_engine()
        return np.asarray(self._URM_fit.dot(self._get(0)) + self._get(1), dtype=np.float32)

    # ---- persistence (GANMF.py:309-342) -----------------------------------------------------------
    # Same files as the reference: build_params.pkl + the tf.train.Saver bundle <name>.index /
    # <name>.data-00000-of-00001 holding params['D'] + params['G'] (ganmf_amd/tf_bundle.py), so models
    # saved by either side load on the other (AblationStudy.py:85-88, MFLearned.py:95).
    def _model_file(self, folder_path, file_name):
        return os.path.join(folder_path, self.RECOMMENDER_NAME + '_' + self.mode if file_name is None else file_name)

    def saveModel(self, folder_path, file_name=None):
        from .tf_bundle import write_bundle
        self._require_engine()
        os.makedirs(folder_path, exist_ok=True)
        build_params = {'num_factors': self.num_factors, 'emb_dim': self.emb_dim}
        with open(os.path.join(folder_path, 'build_params.pkl'), 'wb') as f:
            pickle.dump(build_params, f, pickle.HIGHEST_PROTOCOL)
        write_bundle(self._model_file(folder_path, file_name),
                     {ref.name: self.sess.run(ref) for ref in self.params['D'] + self.params['G']})

    def loadModel(self, folder_path, file_name=None):
        from .tf_bundle import read_bundle
        filepath = os.path.join(folder_path, 'build_params.pkl')
        if self.verbose:
            print(self.RECOMMENDER_NAME + ': Loading model from file ' + filepath)
        with open(filepath, 'rb') as f:
            build_params = pickle.load(f)
        data = read_bundle(self._model_file(folder_path, file_name))
        self._build(build_params['num_factors'], build_params['emb_dim'], batch_size=32)
        for ref in self.params['D'] + self.params['G']:
            if ref.name not in data:
                raise KeyError("%s: checkpoint has no tensor %r" % (self.RECOMMENDER_NAME, ref.name))
            self.engine.set_tensor(ref.tid, data[ref.name])
        if self.verbose:
            print(self.RECOMMENDER_NAME + ': Loading complete')
