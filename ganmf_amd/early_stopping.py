"""Patience-based early stopping on validation metrics, the protocol `fit()` drives once per epoch.

Interface and behaviour follow the scheduler the reference's recommenders are written against (Utils_.py:25-88):
the object is called with the epoch number; past `after`, every `freq`-th epoch it evaluates the model and reads
the metrics at cut-off 5 (the reference hard-codes that key, Utils_.py:64).  An evaluation that improves NONE of the
watched metrics strictly costs one unit of patience; with no patience left the model is told to stop and to restore
its best weights.  Any strict improvement re-arms the patience and asks the model to snapshot itself.
The model side of the protocol: `stop_fit()`, `load_model()`, `save_current_model()` (GANMF.py:246-255).
Public attributes other code reads (`best_scores`, `worse_left`, `allow_worse`, `freq`, `after`, `scores`) keep the
reference's names."""
import numpy as np

_CUTOFF_KEY = 5


class EarlyStoppingScheduler(object):
    def __init__(self, model, evaluator, metrics=['PRECISION', 'RECALL', 'MAP', 'NDCG'], freq=1, allow_worse=5,
                 after=0):
        self.model, self.evaluator = model, evaluator
        self.metrics = metrics
        self.freq, self.after = freq, after
        self.allow_worse = allow_worse
        self.worse_left = allow_worse
        self.best_scores = np.zeros(len(metrics))
        self.scores = []

    def _evaluate(self):
        at5 = self.evaluator.evaluateRecommender(self.model)[0][_CUTOFF_KEY]
        return np.array([at5[name] for name in self.metrics])

    def score(self, epoch):
        if epoch % self.freq:
            return
        current = self._evaluate()
        self.scores.append(current)
        # "not all <=" rather than "any >": a NaN metric counts as an improvement, as it does in the reference
        if not np.less_equal(current, self.best_scores).all():
            self.best_scores = current
            self.reset()
            self.model.save_current_model()
        elif self.worse_left > 0:
            self.worse_left -= 1
        else:
            self.model.stop_fit()
            self.load_best()

    def reset(self):
        self.worse_left = self.allow_worse

    def __call__(self, epoch):
        if epoch > self.after:
            self.score(epoch)

    def load_best(self):
        self.model.load_model()

    def get_scores(self):
        return self.scores
