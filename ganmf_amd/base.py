"""Recommender base the reference's callers rely on (Base/BaseRecommender.py): URM accessors,
remove-seen, top-k ranking.  Same method names, argument meaning and return values so that
EvaluatorHoldout / EarlyStoppingScheduler style callers work unchanged."""
import numpy as np
import scipy.sparse as sps


class BaseRecommender(object):
    RECOMMENDER_NAME = "Recommender_Base_Class"

    def __init__(self, URM_train):
        super(BaseRecommender, self).__init__()
        self.URM_train = sps.csr_matrix(URM_train, dtype=np.float32)
        self.URM_train.eliminate_zeros()
        self.n_users, self.n_items = self.URM_train.shape
        self.items_to_ignore_flag = False
        self.items_to_ignore_ID = np.array([], dtype=int)
        self.filterTopPop = False
        self.filterTopPop_ItemsID = np.array([], dtype=int)

    # Base/BaseRecommender.py:51-52
    def get_URM_train(self):
        return self.URM_train.copy()

    def set_items_to_ignore(self, items_to_ignore):
        self.items_to_ignore_flag = True
        self.items_to_ignore_ID = np.array(items_to_ignore, dtype=int)

    def reset_items_to_ignore(self):
        self.items_to_ignore_flag = False
        self.items_to_ignore_ID = np.array([], dtype=int)

    # Base/BaseRecommender.py:93-100
    def _remove_seen_on_scores(self, user_id, scores):
        assert self.URM_train.getformat() == "csr"
        seen = self.URM_train.indices[self.URM_train.indptr[user_id]:self.URM_train.indptr[user_id + 1]]
        scores[seen] = -np.inf
        return scores

    def _remove_TopPop_on_scores(self, scores_batch):
        scores_batch[:, self.filterTopPop_ItemsID] = -np.inf
        return scores_batch

    def _remove_CustomItems_on_scores(self, scores_batch):
        scores_batch[:, self.items_to_ignore_ID] = -np.inf
        return scores_batch

    def _compute_item_score(self, user_id_array, items_to_compute=None):
        raise NotImplementedError("BaseRecommender: compute_item_score not assigned for current recommender")

    # Base/BaseRecommender.py:155-247
    def recommend(self, user_id_array, cutoff=None, remove_seen_flag=True, items_to_compute=None,
                  remove_top_pop_flag=False, remove_CustomItems_flag=False, return_scores=False):
        if np.isscalar(user_id_array):
            user_id_array = np.atleast_1d(user_id_array)
            single_user = True
        else:
            single_user = False
        if cutoff is None:
            cutoff = self.URM_train.shape[1] - 1

        scores_batch = self._compute_item_score(user_id_array, items_to_compute=items_to_compute)

        if remove_seen_flag:
            for user_index in range(len(user_id_array)):
                self._remove_seen_on_scores(user_id_array[user_index], scores_batch[user_index, :])
        if remove_top_pop_flag:
            scores_batch = self._remove_TopPop_on_scores(scores_batch)
        if remove_CustomItems_flag:
            scores_batch = self._remove_CustomItems_on_scores(scores_batch)

        # partition -> sort the relevant part -> original indices (same three steps as the reference)
        relevant_items_partition = (-scores_batch).argpartition(cutoff, axis=1)[:, 0:cutoff]
        rows = np.arange(scores_batch.shape[0])[:, None]
        partition_values = scores_batch[rows, relevant_items_partition]
        partition_sorting = np.argsort(-partition_values, axis=1)
        ranking = relevant_items_partition[rows, partition_sorting]

        ranking_list = [None] * ranking.shape[0]
        for user_index in range(len(user_id_array)):
            user_recommendation_list = ranking[user_index]
            user_item_scores = scores_batch[user_index, user_recommendation_list]
            not_inf_scores_mask = np.logical_not(np.isinf(user_item_scores))
            ranking_list[user_index] = user_recommendation_list[not_inf_scores_mask].tolist()

        if single_user:
            ranking_list = ranking_list[0]
        if return_scores:
            return ranking_list, scores_batch
        return ranking_list

    def saveModel(self, folder_path, file_name=None):
        raise NotImplementedError("BaseRecommender: saveModel not implemented")
