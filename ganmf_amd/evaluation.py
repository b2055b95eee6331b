"""Hold-out evaluation -- the consumer of the scores (protocol of Base/Evaluation/Evaluator.py:214-414 with the
accuracy metrics of Base/Evaluation/metrics.py).  The GANMF callers read `results_dic[cutoff][metric]` (early stopping:
cut-off 5, Utils_.py:64); the reference's beyond-accuracy metrics (novelty, diversity, coverage) are out of scope.

Metric definitions (one user, a ranked list `L` cut at c, test items `T` with ratings `w`, hit flags `h_i = [L_i in T]`):
  PRECISION = sum(h)/|L|      PRECISION_RECALL_MIN_DEN = sum(h)/min(|T|,|L|)      RECALL = sum(h)/|T|
  HIT_RATE = sum(h)           MRR = 1/rank of the first hit        ARHR = sum_i h_i/i
  MAP = sum_i h_i * (hits up to i)/i / min(|T|,|L|)
  ROC_AUC = share of (hit, miss) pairs of the list ranked in the right order (1 when the list has no miss)
  NDCG = DCG(L)/DCG(best |L| of T),  DCG = sum_i (2^{w_i}-1)/ln(i+1)
  RMSE over the test items whose score is finite (seen items carry -inf and do not count)
Every user contributes the same weight; values are means over the evaluated users; F1 is formed from the means.

`RankedListMetrics` computes all of them for one list in one pass over the hit positions.  Where the reference's
evaluator works in float32 (hit counts divided in float32, the DCG sums, and therefore its running sums) this one
does too, so that the two agree to the last digits on the reference's golden outputs (tests/test_evaluator.py)."""
import itertools

import numpy as np
import scipy.sparse as sps

METRICS = ("ROC_AUC", "PRECISION", "PRECISION_RECALL_MIN_DEN", "RECALL", "MAP", "MRR", "NDCG", "F1",
           "HIT_RATE", "ARHR", "RMSE")
_SUMMED = tuple(m for m in METRICS if m != "F1")


def get_result_string(results_run, n_decimals=7):
    """'CUTOFF: c - NAME: value, NAME: value, \n' per cut-off, the line format the reference's drivers log."""
    lines = []
    for cutoff, per_metric in results_run.items():
        fields = "".join("%s: %.*f, " % (name, n_decimals, value) for name, value in per_metric.items())
        lines.append("CUTOFF: %s - %s\n" % (cutoff, fields))
    return "".join(lines)


class RankedListMetrics(object):
    """Accuracy metrics of one user's ranked list against that user's test items.

    `test_items` / `test_ratings`: the stored entries of the user's URM_test row.  `__call__(recommended, c)` returns a
    dict over the metric names for the list cut at c."""

    def __init__(self, test_items, test_ratings, max_cutoff):
        order = np.argsort(test_items, kind="stable")
        self._items = np.asarray(test_items)[order]
        self._ratings = np.asarray(test_ratings)[order]
        self.n_test = int(self._items.shape[0])
        # DCG discounts ln(rank + 1), rank = 1..max_cutoff, and the user's best possible gains, both float32
        self._ln_rank = np.log(np.arange(max_cutoff, dtype=np.float32) + 2)
        best_first = np.sort(np.asarray(test_ratings))[::-1][:max_cutoff]
        self._ideal_terms = (np.power(2, best_first.astype(np.float32)) - 1) / self._ln_rank[:best_first.shape[0]]

    def match(self, recommended):
        """(hit flags, rating of each hit else 0) for the ranked ids."""
        recommended = np.asarray(recommended, dtype=self._items.dtype if self.n_test else np.int64)
        if self.n_test == 0 or recommended.shape[0] == 0:
            return np.zeros(recommended.shape[0], dtype=bool), np.zeros(recommended.shape[0], dtype=np.float32)
        slot = np.minimum(np.searchsorted(self._items, recommended), self.n_test - 1)
        hit = self._items[slot] == recommended
        return hit, np.where(hit, self._ratings[slot], 0).astype(np.float32)

    def __call__(self, hit, gain, c):
        hit, gain = hit[:c], gain[:c]
        n = int(hit.shape[0])
        at = np.flatnonzero(hit)                       # 0-based ranks of the hits
        n_hit = int(at.shape[0])
        hits32 = np.float32(n_hit)
        out = dict.fromkeys(_SUMMED, 0.0)
        out["HIT_RATE"] = n_hit
        out["RECALL"] = hits32 / self.n_test
        if n:
            out["PRECISION"] = hits32 / n
            out["PRECISION_RECALL_MIN_DEN"] = hits32 / min(self.n_test, n)
        n_miss = n - n_hit
        if n_miss == 0:
            out["ROC_AUC"] = 1.0
        elif n_hit:
            # misses ranked below hit j (the j-th hit at rank at[j]): the (n-1-at[j]) later entries minus the later hits
            ordered_pairs = int(((n - 1 - at) - (n_hit - 1 - np.arange(n_hit))).sum())
            out["ROC_AUC"] = np.float32(ordered_pairs) / (n_hit * n_miss)
        if n_hit:
            rank = at + 1.0
            out["MRR"] = 1.0 / rank[0]
            out["ARHR"] = float((1.0 / rank).sum())
            precision_at_hit = np.arange(1, n_hit + 1, dtype=np.float32) / rank
            out["MAP"] = precision_at_hit.sum() / min(self.n_test, n)
            dcg = np.sum((np.power(2, gain) - 1) / self._ln_rank[:n], dtype=np.float32)
            if dcg != 0.0:
                out["NDCG"] = dcg / np.sum(self._ideal_terms[:n], dtype=np.float32)
        return out


def rmse_on_test_items(score_row, test_items, test_ratings):
    """Root mean squared error over the test items with a finite score; NaN when there is none."""
    sq = (score_row[test_items] - test_ratings) ** 2
    usable = np.isfinite(sq)
    count = usable.sum()
    return np.sqrt(np.sum(sq[usable]) / count) if count else np.nan


def _finish(sums, n_eval, cutoffs):
    """Means over the evaluated users + F1 of the mean precision / recall (0 when both are 0)."""
    results = {}
    for c in cutoffs:
        r = {name: sums[c][name] / n_eval for name in _SUMMED}
        p, rc = r["PRECISION"], r["RECALL"]
        r["F1"] = 2 * (p * rc) / (p + rc) if p + rc != 0 else 0.0
        results[c] = r
    return results


class EvaluatorHoldout(object):
    EVALUATOR_NAME = "EvaluatorHoldout"

    def __init__(self, URM_test_list, cutoff_list, minRatingsPerUser=1, exclude_seen=True):
        if isinstance(URM_test_list, list):
            raise ValueError("List of URM_test not supported")
        self.cutoff_list = list(cutoff_list)
        self.max_cutoff = max(self.cutoff_list)
        self.minRatingsPerUser = minRatingsPerUser
        self.exclude_seen = exclude_seen
        self.URM_test = sps.csr_matrix(URM_test_list)
        self.n_users, self.n_items = self.URM_test.shape
        n_ratings = np.ediff1d(self.URM_test.indptr)
        self.usersToEvaluate = list(np.arange(self.n_users)[n_ratings >= minRatingsPerUser])

    def get_user_relevant_items(self, user_id):
        return self.URM_test.indices[self.URM_test.indptr[user_id]:self.URM_test.indptr[user_id + 1]]

    def get_user_test_ratings(self, user_id):
        return self.URM_test.data[self.URM_test.indptr[user_id]:self.URM_test.indptr[user_id + 1]]

    def evaluateRecommender(self, recommender_object):
        """(results[cutoff][metric], text).  Users are scored in blocks of min(1000, 1e8/n_items) through
        `recommender.recommend(..., return_scores=True)` (Evaluator.py:237-277)."""
        block_size = min(1000, int(1e8 / self.n_items))
        sums = {c: dict.fromkeys(_SUMMED, 0.0) for c in self.cutoff_list}
        users = np.asarray(self.usersToEvaluate, dtype=np.int64)
        for lo in range(0, len(users), max(block_size, 1)):
            batch = users[lo:lo + block_size]
            rec_lists, scores_batch = recommender_object.recommend(
                batch, remove_seen_flag=self.exclude_seen, cutoff=self.max_cutoff, remove_top_pop_flag=False,
                remove_CustomItems_flag=False, return_scores=True)
            assert len(rec_lists) == len(batch) and scores_batch.shape == (len(batch), self.n_items)
            for user, recommended, score_row in zip(batch, rec_lists, scores_batch):
                test_items, test_ratings = self.get_user_relevant_items(user), self.get_user_test_ratings(user)
                scorer = RankedListMetrics(test_items, test_ratings, self.max_cutoff)
                hit, gain = scorer.match(recommended)
                user_rmse = rmse_on_test_items(score_row, test_items, test_ratings)
                for c in self.cutoff_list:
                    acc = sums[c]
                    for name, value in scorer(hit, gain, c).items():
                        acc[name] += value
                    acc["RMSE"] += user_rmse
        n_eval = len(users)
        if n_eval == 0:
            print("WARNING: No users had a sufficient number of relevant items")
            results = {c: dict.fromkeys(METRICS, 0.0) for c in self.cutoff_list}
        else:
            results = _finish(sums, n_eval, self.cutoff_list)
        return results, get_result_string(results)


_DEVICE_TOKENS = itertools.count(1)


class EvaluatorHoldoutFast(EvaluatorHoldout):
    """Same protocol and result dictionaries as EvaluatorHoldout (Evaluator.py:214-414), but consumes only the
    top-`max_cutoff` ids of each user — `recommender.recommend_topk(...)` when the recommender has it (device
    selection, include/ganmf_hip.h: ganmf_recommend), else `recommend(..., return_scores=False)` — and computes
    the ranking metrics for a whole block of users at once.  Sums are float64 (the per-user functions above
    follow the reference's float32 sums); the two agree to ~1e-6 relative.  RMSE needs every score of every
    user and is reported as NaN here; SURVEY §8(f) row 1."""
    EVALUATOR_NAME = "EvaluatorHoldoutFast"

    def __init__(self, URM_test_list, cutoff_list, minRatingsPerUser=1, exclude_seen=True):
        super().__init__(URM_test_list, cutoff_list, minRatingsPerUser=minRatingsPerUser, exclude_seen=exclude_seen)
        K = self.max_cutoff
        self._users = np.asarray(self.usersToEvaluate, dtype=np.int64)
        self._n_test = np.ediff1d(self.URM_test.indptr)[self._users].astype(np.int64)
        # relevance lookup: stored entries of URM_test are the relevant items (Evaluator.py:46-52), value = gain
        self._rel = sps.csr_matrix((np.ones_like(self.URM_test.data, dtype=np.float64), self.URM_test.indices,
                                    self.URM_test.indptr), shape=self.URM_test.shape)
        self._gain = sps.csr_matrix((np.power(2.0, self.URM_test.data.astype(np.float32)).astype(np.float64) - 1.0,
                                     self.URM_test.indices, self.URM_test.indptr), shape=self.URM_test.shape)
        # ideal DCG prefix sums: ratings sorted descending, first K, discounted (metrics.py ndcg/dcg)
        disc = 1.0 / np.log(np.arange(K, dtype=np.float32) + 2).astype(np.float64)
        ideal = np.zeros((len(self._users), K))
        for i, u in enumerate(self._users):
            r = np.sort(self.get_user_test_ratings(u))[::-1][:K].astype(np.float32)
            ideal[i, :len(r)] = (np.power(2.0, r).astype(np.float64) - 1.0) * disc[:len(r)]
        self._ideal_cum = np.cumsum(ideal, axis=1)
        self._disc = disc
        # device route (recommender.evaluate_on_device -> ganmf_evaluate): the test matrix with sorted rows and its DCG gains
        self._test_sorted = self.URM_test.tocsr().copy()
        self._test_sorted.sort_indices()
        self._test_gain = np.power(2.0, self._test_sorted.data.astype(np.float32)).astype(np.float64) - 1.0
        self.use_device_metrics = True
        # identifies THIS evaluator's test matrix on the device (never reused, unlike id(): CPython hands the id of a freed
        # evaluator to the next one, and a recommender keyed on it would score the new evaluator against the old test matrix)
        self._device_token = next(_DEVICE_TOKENS)

    def _topk(self, rec, batch):
        K = self.max_cutoff
        if hasattr(rec, "recommend_topk"):
            return np.asarray(rec.recommend_topk(batch, K, remove_seen_flag=self.exclude_seen))
        lists = rec.recommend(batch, remove_seen_flag=self.exclude_seen, cutoff=K, remove_top_pop_flag=False,
                              remove_CustomItems_flag=False, return_scores=False)
        out = np.full((len(batch), K), -1, dtype=np.int64)
        for i, l in enumerate(lists):
            out[i, :len(l)] = l
        return out

    def evaluateRecommender(self, recommender_object):
        K = self.max_cutoff
        block_size = max(1, min(4096, int(1e8 / self.n_items)))
        names = _SUMMED
        sums = {c: {m: 0.0 for m in names} for c in self.cutoff_list}
        n_eval = len(self._users)
        inv_rank = 1.0 / np.arange(1, K + 1, dtype=np.float64)
        if self.use_device_metrics and n_eval > 0 and hasattr(recommender_object, "evaluate_on_device"):
            # everything on the device: scores, seen mask, top-k AND the metric sums (only len(cutoffs) x 9 doubles come back)
            # in user blocks: the device forms a [block, n_items] score matrix (+ block x K doubles) per call, the same cap as
            # the host routes; the [cutoffs, 9] partial sums are added here in block order.  Out of device memory -> host route.
            dev = None
            try:
                for start in range(0, n_eval, max(1, int(1e8 / self.n_items))):
                    sl = slice(start, min(start + max(1, int(1e8 / self.n_items)), n_eval))
                    part = recommender_object.evaluate_on_device(self._device_token, self._test_sorted, self._test_gain,
                                                                 self._users[sl], self.cutoff_list, self._disc,
                                                                 self._ideal_cum[sl], remove_seen_flag=self.exclude_seen)
                    if part is None:
                        dev = None
                        break
                    dev = part if dev is None else dev + part
            except MemoryError:
                dev = None
            if dev is not None:
                from ._lib import EVAL_METRICS
                for ci, c in enumerate(self.cutoff_list):
                    for mi, name in enumerate(EVAL_METRICS):
                        sums[c][name] = float(dev[ci, mi])
                results = _finish(sums, n_eval, self.cutoff_list)
                for c in self.cutoff_list:
                    results[c] = {m: float(v) for m, v in results[c].items()}
                    results[c]["RMSE"] = float("nan")
                return results, get_result_string(results)
        for start in range(0, n_eval, block_size):
            sl = slice(start, min(start + block_size, n_eval))
            batch = self._users[sl]
            items = self._topk(recommender_object, batch)
            assert items.shape == (len(batch), K)
            valid = items >= 0
            safe = np.where(valid, items, 0)
            rows = np.repeat(np.arange(len(batch)), K)
            rel_block, gain_block = self._rel[batch], self._gain[batch]
            is_rel = np.asarray(rel_block[rows, safe.ravel()]).reshape(len(batch), K) > 0
            is_rel &= valid
            gain = np.asarray(gain_block[rows, safe.ravel()]).reshape(len(batch), K) * is_rel
            n_test = self._n_test[sl].astype(np.float64)
            length = valid.sum(axis=1)
            for c in self.cutoff_list:
                r = sums[c]
                rel = is_rel[:, :c].astype(np.float64)
                neg = (valid[:, :c] & ~is_rel[:, :c]).astype(np.float64)
                len_c = np.minimum(length, c).astype(np.float64)
                hits = rel.sum(axis=1)
                nneg = neg.sum(axis=1)
                # AUC over the list: for each hit, the negatives ranked after it (metrics.py roc_auc)
                neg_after = nneg[:, None] - np.cumsum(neg, axis=1)
                pairs = (rel * neg_after).sum(axis=1)
                auc = np.where(nneg == 0, 1.0, np.where(hits > 0, pairs / np.maximum(hits * nneg, 1.0), 0.0))
                nz = np.maximum(len_c, 1.0)
                r["ROC_AUC"] += auc.sum()
                r["PRECISION"] += np.where(len_c > 0, hits / nz, 0.0).sum()
                r["PRECISION_RECALL_MIN_DEN"] += np.where(len_c > 0, hits / np.maximum(np.minimum(n_test, len_c), 1.0), 0.0).sum()
                r["RECALL"] += (hits / n_test).sum()
                dcg_rank = (gain[:, :c] * self._disc[:c]).sum(axis=1)
                li = np.maximum(len_c.astype(np.int64) - 1, 0)
                ideal = self._ideal_cum[sl][np.arange(len(batch)), li]
                r["NDCG"] += np.where(dcg_rank > 0, dcg_rank / np.where(ideal > 0, ideal, 1.0), 0.0).sum()
                r["HIT_RATE"] += hits.sum()
                r["ARHR"] += (rel * inv_rank[:c]).sum()
                first = np.argmax(rel > 0, axis=1)
                r["MRR"] += np.where(hits > 0, inv_rank[first], 0.0).sum()
                p_at_k = rel * np.cumsum(rel, axis=1) * inv_rank[:c]
                r["MAP"] += np.where(len_c > 0, p_at_k.sum(axis=1) / np.maximum(np.minimum(n_test, len_c), 1.0), 0.0).sum()
        if n_eval > 0:
            results = _finish(sums, n_eval, self.cutoff_list)
            for c in self.cutoff_list:
                results[c] = {m: float(v) for m, v in results[c].items()}
                results[c]["RMSE"] = float("nan")
        else:
            results = {c: dict.fromkeys(METRICS, 0.0) for c in self.cutoff_list}
            print("WARNING: No users had a sufficient number of relevant items")
        return results, get_result_string(results)
