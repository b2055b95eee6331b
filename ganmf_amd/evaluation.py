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
