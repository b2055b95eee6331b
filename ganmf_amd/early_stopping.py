"""Early stopping on validation metrics @5 (Utils_.py:25-88), same protocol with the model:
model.stop_fit(), model.load_model(), model.save_current_model()."""
import numpy as np


class EarlyStoppingScheduler(object):
    def __init__(self, model, evaluator, metrics=['PRECISION', 'RECALL', 'MAP', 'NDCG'], freq=1, allow_worse=5,
                 after=0):
        self.model = model
        self.evaluator = evaluator
        self.metrics = metrics
        self.freq = freq
        self.best_scores = np.zeros(len(metrics))
        self.allow_worse = allow_worse
        self.worse_left = allow_worse
        self.after = after
        self.scores = []

    def score(self, epoch):
        if epoch % self.freq == 0:
            results_dic, _ = self.evaluator.evaluateRecommender(self.model)
            curr_scores = np.array([results_dic[5][m] for m in self.metrics])   # hard-coded cutoff (Utils_.py:64)
            self.scores.append(curr_scores)
            if np.all(np.less_equal(curr_scores, self.best_scores)):
                if self.worse_left > 0:
                    self.worse_left -= 1
                else:
                    self.model.stop_fit()
                    self.model.load_model()
            else:
                self.best_scores = curr_scores
                self.worse_left = self.allow_worse
                self.model.save_current_model()

    def reset(self):
        self.worse_left = self.allow_worse

    def __call__(self, epoch):
        if epoch > self.after:
            self.score(epoch)

    def load_best(self):
        self.model.load_model()

    def get_scores(self):
        return self.scores
