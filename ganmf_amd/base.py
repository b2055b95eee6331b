"""Recommender base class with the surface the reference's callers use (Base/BaseRecommender.py:51-52,93-100,
155-247): URM accessor, the three score filters and `recommend()`.  Method names, arguments and return values
are the reference's, so its evaluators / early-stopping scheduler / experiment drivers can hold one of these;
the bodies are this package's own, vectorised over the whole user block:

  * seen items of every user of the block are masked with ONE fancy-index assignment built from the CSR arrays;
  * ranking = partial selection of the `cutoff` best per row, then an ordering of those `cutoff` only;
  * masked (-inf) entries are dropped from each list with one boolean matrix instead of a per-user mask.

A score of -inf means "never recommend" in all three filters, exactly as in the reference."""
import numpy as np
import scipy.sparse as sps


def _csr_block_coordinates(csr, row_ids):
    """(r, c) coordinates of every stored entry of `csr[row_ids]`, r counted inside the block."""
    row_ids = np.asarray(row_ids, dtype=np.int64)
    begin, end = csr.indptr[row_ids], csr.indptr[row_ids + 1]
    lengths = (end - begin).astype(np.int64)
    total = int(lengths.sum())
    if total == 0:
        return np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.int64)
    block_rows = np.repeat(np.arange(len(row_ids), dtype=np.int64), lengths)
    # position of each entry inside its own row, then shifted to the row's start in csr.indices
    first_of_row = np.repeat(np.cumsum(lengths) - lengths, lengths)
    within = np.arange(total, dtype=np.int64) - first_of_row
    return block_rows, csr.indices[np.repeat(begin.astype(np.int64), lengths) + within]


def rank_top(scores, cutoff):
    """Per row, the indices of the `cutoff` largest scores in descending order of score: [rows, cutoff]."""
    n_cols = scores.shape[1]
    kth = min(int(cutoff), n_cols - 1)
    neg = -scores
    candidates = np.argpartition(neg, kth, axis=1)[:, :cutoff]
    order = np.argsort(np.take_along_axis(neg, candidates, axis=1), axis=1)
    return np.take_along_axis(candidates, order, axis=1)


class BaseRecommender(object):
    RECOMMENDER_NAME = "Recommender_Base_Class"

    def __init__(self, URM_train):
        super(BaseRecommender, self).__init__()
        self.URM_train = sps.csr_matrix(URM_train, dtype=np.float32)
        self.URM_train.eliminate_zeros()
        self.n_users, self.n_items = self.URM_train.shape
        self.reset_items_to_ignore()
        self.filterTopPop = False
        self.filterTopPop_ItemsID = np.array([], dtype=int)

    def get_URM_train(self):
        return self.URM_train.copy()

    def set_items_to_ignore(self, items_to_ignore):
        self.items_to_ignore_flag = True
        self.items_to_ignore_ID = np.array(items_to_ignore, dtype=int)

    def reset_items_to_ignore(self):
        self.items_to_ignore_flag = False
        self.items_to_ignore_ID = np.array([], dtype=int)

    # ---- score filters ---------------------------------------------------------------------------
    def _remove_seen_on_scores(self, user_id, scores):
        """One user's score row with the items of that user's training profile set to -inf (in place)."""
        if self.URM_train.getformat() != "csr":
            raise AssertionError("URM_train must be CSR to mask seen items")
        lo, hi = self.URM_train.indptr[user_id], self.URM_train.indptr[user_id + 1]
        scores[self.URM_train.indices[lo:hi]] = -np.inf
        return scores

    def _remove_seen_on_scores_block(self, user_id_array, scores_batch):
        """All rows of a block at once (what recommend() uses)."""
        if self.URM_train.getformat() != "csr":
            raise AssertionError("URM_train must be CSR to mask seen items")
        r, c = _csr_block_coordinates(self.URM_train, user_id_array)
        scores_batch[r, c] = -np.inf
        return scores_batch

    def _remove_TopPop_on_scores(self, scores_batch):
        scores_batch[:, self.filterTopPop_ItemsID] = -np.inf
        return scores_batch

    def _remove_CustomItems_on_scores(self, scores_batch):
        scores_batch[:, self.items_to_ignore_ID] = -np.inf
        return scores_batch

    def _compute_item_score(self, user_id_array, items_to_compute=None):
        raise NotImplementedError("BaseRecommender: compute_item_score not assigned for current recommender")

    # ---- ranking ---------------------------------------------------------------------------------
    def recommend(self, user_id_array, cutoff=None, remove_seen_flag=True, items_to_compute=None,
                  remove_top_pop_flag=False, remove_CustomItems_flag=False, return_scores=False):
        """Ranked item ids per user (list of lists; a single list for a scalar user id); with `return_scores` also
        the filtered score matrix.  Lists are shorter than `cutoff` when fewer unmasked items exist."""
        one_user = np.isscalar(user_id_array)
        users = np.atleast_1d(user_id_array)
        if cutoff is None:
            cutoff = self.URM_train.shape[1] - 1

        scores_batch = self._compute_item_score(users, items_to_compute=items_to_compute)
        if remove_seen_flag:
            self._remove_seen_on_scores_block(users, scores_batch)
        if remove_top_pop_flag:
            scores_batch = self._remove_TopPop_on_scores(scores_batch)
        if remove_CustomItems_flag:
            scores_batch = self._remove_CustomItems_on_scores(scores_batch)

        ranking = rank_top(scores_batch, cutoff)
        keep = ~np.isinf(np.take_along_axis(scores_batch, ranking, axis=1))
        lists = [row[mask].tolist() for row, mask in zip(ranking, keep)]

        result = lists[0] if one_user else lists
        return (result, scores_batch) if return_scores else result

    def saveModel(self, folder_path, file_name=None):
        raise NotImplementedError("BaseRecommender: saveModel not implemented")
