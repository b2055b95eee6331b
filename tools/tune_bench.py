"""Trial-parallel search on ML-1M (the regime the paper ran: 50 trials x <= 300 epochs took 2 h 18 m on its GPU,
SURVEY §6).  Usage: python tools/tune_bench.py [evals] [worker processes] [trial threads per process]"""
import os
import sys
import tempfile
import time

import numpy as np
import scipy.sparse as sps

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from GANRec.GANMF import GANMF  # noqa: E402
from ganmf_amd import tune  # noqa: E402


def main():
    g = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
    train = sps.load_npz(os.path.join(g, "Movielens1M_URM_train.npz")).tocsr()
    test = sps.load_npz(os.path.join(g, "Movielens1M_URM_test.npz")).tocsr()
    # carve early-stop / validation splits out of the training matrix the way the reference's splitter does (80/10/10 of interactions)
    rng = np.random.RandomState(1)
    coo = train.tocoo()
    r = rng.rand(coo.nnz)
    mk = lambda m: sps.csr_matrix((coo.data[m], (coo.row[m], coo.col[m])), shape=train.shape)
    small, early, val = mk(r < 0.8), mk((r >= 0.8) & (r < 0.9)), mk(r >= 0.9)
    evals = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    workers = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    engines = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    t0 = time.time()
    t = tune.TrialParallelTuner(GANMF, small, early, val, tempfile.mkdtemp(), seed=1337, n_workers=workers, engines_per_worker=engines)
    best, params = t.tune(evals=evals)
    print("%d trials, %d workers x %d engines on %d device(s): %.1f s wall; best MAP@5 %.4f at %s" % (evals, workers, engines, len(t.devices), time.time() - t0, -best, params))


if __name__ == "__main__":      # worker processes are spawned: they re-import this module
    main()
