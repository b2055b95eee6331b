"""Hold-out evaluation — the consumer of the scores (Base/Evaluation/Evaluator.py:214-414,
metrics.py).  Restates the accuracy metrics the GANMF callers read (early stopping uses
results_dic[5][metric], Utils_.py:64); beyond-accuracy metrics of the reference (novelty,
diversity, coverage) are out of scope.  Accumulation dtypes follow the reference (float32 sums
inside precision/recall/AP/DCG) so results agree to the last digits with its evaluator."""
import numpy as np
import scipy.sparse as sps

METRICS = ("ROC_AUC", "PRECISION", "PRECISION_RECALL_MIN_DEN", "RECALL", "MAP", "MRR", "NDCG", "F1",
           "HIT_RATE", "ARHR", "RMSE")


def roc_auc(is_relevant):
    ranks = np.arange(len(is_relevant))
    pos_ranks = ranks[is_relevant]
    neg_ranks = ranks[~is_relevant]
    auc_score = 0.0
    if len(neg_ranks) == 0:
        return 1.0
    if len(pos_ranks) > 0:
        for pos_pred in pos_ranks:
            auc_score += np.sum(pos_pred < neg_ranks, dtype=np.float32)
        auc_score /= (pos_ranks.shape[0] * neg_ranks.shape[0])
    return auc_score


def precision(is_relevant):
    if len(is_relevant) == 0:
        return 0.0
    return np.sum(is_relevant, dtype=np.float32) / len(is_relevant)


def precision_recall_min_denominator(is_relevant, n_test_items):
    if len(is_relevant) == 0:
        return 0.0
    return np.sum(is_relevant, dtype=np.float32) / min(n_test_items, len(is_relevant))


def recall(is_relevant, pos_items):
    return np.sum(is_relevant, dtype=np.float32) / pos_items.shape[0]


def rr(is_relevant):
    ranks = np.arange(1, len(is_relevant) + 1)[is_relevant]
    return 1. / ranks[0] if len(ranks) > 0 else 0.0


def arhr(is_relevant):
    p_reciprocal = 1 / np.arange(1, len(is_relevant) + 1, 1.0, dtype=np.float64)
    return is_relevant.dot(p_reciprocal)


def average_precision(is_relevant, pos_items):
    if len(is_relevant) == 0:
        return 0.0
    p_at_k = is_relevant * np.cumsum(is_relevant, dtype=np.float32) / (1 + np.arange(is_relevant.shape[0]))
    return np.sum(p_at_k) / np.min([pos_items.shape[0], is_relevant.shape[0]])


def dcg(scores):
    return np.sum(np.divide(np.power(2, scores) - 1, np.log(np.arange(scores.shape[0], dtype=np.float32) + 2)),
                  dtype=np.float32)


def ndcg(ranked_list, pos_items, relevance=None, at=None):
    if relevance is None:
        relevance = np.ones_like(pos_items)
    it2rel = {it: r for it, r in zip(pos_items, relevance)}
    rank_scores = np.asarray([it2rel.get(it, 0.0) for it in ranked_list[:at]], dtype=np.float32)
    ideal_dcg = dcg(np.sort(relevance)[::-1][:len(ranked_list)])
    rank_dcg = dcg(rank_scores)
    if rank_dcg == 0.0:
        return 0.0
    return rank_dcg / ideal_dcg


def rmse(all_items_predicted_ratings, relevant_items, relevant_items_rating):
    err = (all_items_predicted_ratings[relevant_items] - relevant_items_rating) ** 2
    finite = np.isfinite(err)
    if finite.sum() == 0:
        return np.nan
    return np.sqrt(np.sum(err[finite]) / finite.sum())


def get_result_string(results_run, n_decimals=7):
    out = ""
    for cutoff, res in results_run.items():
        out += "CUTOFF: {} - ".format(cutoff)
        for metric, value in res.items():
            out += "{}: {:.{n}f}, ".format(metric, value, n=n_decimals)
        out += "\n"
    return out


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
        block_size = min(1000, int(1e8 / self.n_items))       # Evaluator.py:237-238
        results = {c: {m: 0.0 for m in METRICS if m != "F1"} for c in self.cutoff_list}
        n_eval = 0
        users = self.usersToEvaluate
        start = 0
        while start < len(users):
            end = min(start + block_size, len(users))
            batch = np.array(users[start:end])
            start = end
            rec_lists, scores_batch = recommender_object.recommend(
                batch, remove_seen_flag=self.exclude_seen, cutoff=self.max_cutoff, remove_top_pop_flag=False,
                remove_CustomItems_flag=False, return_scores=True)
            assert len(rec_lists) == len(batch) and scores_batch.shape == (len(batch), self.n_items)
            for bi in range(len(batch)):
                user = batch[bi]
                relevant = self.get_user_relevant_items(user)
                ratings = self.get_user_test_ratings(user)
                user_rmse = rmse(scores_batch[bi], relevant, ratings)
                recommended = rec_lists[bi]
                is_relevant = np.isin(recommended, relevant, assume_unique=True)
                n_eval += 1
                for c in self.cutoff_list:
                    r = results[c]
                    rel_c = is_relevant[0:c]
                    rec_c = recommended[0:c]
                    r["ROC_AUC"] += roc_auc(rel_c)
                    r["PRECISION"] += precision(rel_c)
                    r["PRECISION_RECALL_MIN_DEN"] += precision_recall_min_denominator(rel_c, len(relevant))
                    r["RECALL"] += recall(rel_c, relevant)
                    r["NDCG"] += ndcg(rec_c, relevant, relevance=ratings, at=c)
                    r["HIT_RATE"] += rel_c.sum()
                    r["ARHR"] += arhr(rel_c)
                    r["RMSE"] += user_rmse
                    r["MRR"] += rr(rel_c)
                    r["MAP"] += average_precision(rel_c, relevant)
        if n_eval > 0:
            for c in self.cutoff_list:
                r = results[c]
                for key in list(r.keys()):
                    r[key] = r[key] / n_eval
                p, rc = r["PRECISION"], r["RECALL"]
                if p + rc != 0:
                    r["F1"] = 2 * (p * rc) / (p + rc)
        else:
            print("WARNING: No users had a sufficient number of relevant items")
        return results, get_result_string(results)


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
        names = [m for m in METRICS if m != "F1"]
        sums = {c: {m: 0.0 for m in names} for c in self.cutoff_list}
        n_eval = len(self._users)
        inv_rank = 1.0 / np.arange(1, K + 1, dtype=np.float64)
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
        results = {c: {} for c in self.cutoff_list}
        if n_eval > 0:
            for c in self.cutoff_list:
                for m in names:
                    results[c][m] = float(sums[c][m] / n_eval)
                results[c]["RMSE"] = float("nan")
                p, rc = results[c]["PRECISION"], results[c]["RECALL"]
                if p + rc != 0:
                    results[c]["F1"] = 2 * (p * rc) / (p + rc)
        else:
            results = {c: {m: 0.0 for m in names} for c in self.cutoff_list}
            print("WARNING: No users had a sufficient number of relevant items")
        return results, get_result_string(results)
