"""Trial-parallel hyper-parameter search for GANMF / DisGANMF.

The reference tunes one trial at a time (`RecSysExp.tune`, RecSysExp.py:313-412: skopt `gp_minimize` /
`dummy_minimize` over `obj_func`, :246-311; 50 trials x <= 300 epochs on ML-1M took 2 h 18 m).  On an MI355X a
whole ML-1M trial is seconds, so the useful parallelism on an 8-GPU node is ACROSS trials: one worker process per
GPU (or several per GPU: small trials do not fill a device), each running complete trials with early stopping.

Kept from the reference: the search spaces (RecSysExp.py:502-523 plus the data-dependent `emb_dim` / `d_nodes`
dimension, :340-346, and the `num_factors <= min(U, I)` clamp, :351-358), the objective (fit on URM_train_small with
the GAN early-stopping dict `allow_worse=5, freq=5` on URM_early_stop, fitness = -metric@at on URM_validation, the
`epochs = last_epoch - allow_worse*freq` correction, :266-276), and the artefacts in `logsdir` (`results.txt`,
`best_params.pkl`, `best_params.txt`, a resumable `checkpoint.pkl`).  Not kept: skopt itself (absent here) — the
random search draws from numpy `RandomState(seed)` in dimension order, and `method="bayesian"` is a batch
Gaussian-process / expected-improvement loop on scikit-learn with the constant-liar rule for trials in flight.
"""
import json
import multiprocessing as mp
import os
import pickle
import queue
import time
from collections import OrderedDict

import numpy as np

EARLY_STOPPING = {"allow_worse": 5, "freq": 5, "validation_set": None, "sample_every": None}   # RecSysExp.py:217-223


# ---- search space (skopt.space restated) ----------------------------------------------------------
class Dim(object):
    def __init__(self, name, kind, low=None, high=None, prior="uniform", choices=None):
        self.name, self.kind, self.low, self.high, self.prior, self.choices = name, kind, low, high, prior, choices

    def sample(self, rng):
        if self.kind == "categorical":
            return self.choices[rng.randint(len(self.choices))]
        if self.kind == "integer":
            return int(rng.randint(self.low, self.high + 1))
        if self.prior == "log-uniform":
            return float(np.exp(rng.uniform(np.log(self.low), np.log(self.high))))
        return float(rng.uniform(self.low, self.high))

    def to_unit(self, v):
        """position in [0, 1] for the surrogate model"""
        if self.kind == "categorical":
            return self.choices.index(v) / max(len(self.choices) - 1, 1)
        if self.prior == "log-uniform":
            return (np.log(v) - np.log(self.low)) / (np.log(self.high) - np.log(self.low))
        return (v - self.low) / max(self.high - self.low, 1e-12)


def search_space(model_name, n_users, n_items):
    """RecSysExp.py:502-523 + :340-358 (I = number of items of URM_test)."""
    width = int(n_items * 0.75) if n_items <= 1024 else 1024
    common = [Dim("batch_size", "categorical", choices=[64, 128, 256, 512, 1024]),
              Dim("d_lr", "real", 1e-4, 1e-2, "log-uniform"), Dim("g_lr", "real", 1e-4, 1e-2, "log-uniform"),
              Dim("d_reg", "real", 1e-6, 1e-4, "log-uniform"), Dim("recon_coefficient", "real", 1e-2, 0.5)]
    if model_name == "GANMF":
        dims = [Dim("epochs", "categorical", choices=[300]),
                Dim("num_factors", "integer", 1, min(250, n_users, n_items)), common[0],
                Dim("m", "integer", 1, 10)] + common[1:] + [Dim("emb_dim", "integer", 4, width)]
    elif model_name == "DisGANMF":
        dims = [Dim("epochs", "categorical", choices=[300]),
                Dim("d_hidden_act", "categorical", choices=["linear", "tanh", "relu", "sigmoid"]),
                Dim("d_layers", "integer", 1, 5), Dim("num_factors", "integer", 5, min(250, n_users, n_items))] + \
            common + [Dim("d_nodes", "integer", 4, width)]
    else:
        raise ValueError("search_space: unknown model %r" % model_name)
    return dims


# ---- one trial (RecSysExp.obj_func) -----------------------------------------------------------------
def run_trial(spec, params, device, schedule_rng=None):
    """-> dict(fitness, fit_params, results_string, seconds).  `spec` is the picklable experiment description.
    schedule_rng: a numpy RandomState for the fit's per-epoch shuffles (default: numpy's global stream, as the reference);
    trials that run in threads of one process each bring their own."""
    t0 = time.time()
    cls = spec["recommender_class"]
    model = cls(spec["URM_train_small"], mode=spec["mode"], seed=spec["seed"], is_experiment=True, **spec["model_kwargs"])
    if hasattr(model, "device"):
        model.device = device
    if schedule_rng is not None:
        model.schedule_rng = schedule_rng
    fit_params = dict(params)
    fit_kwargs = dict(fit_params)
    fit_kwargs.update(EARLY_STOPPING)
    fit_kwargs["validation_evaluator"] = spec["evaluator_class"](spec["URM_early_stop"], [spec["at"]], exclude_seen=True)
    fit_kwargs["metrics"] = [spec["metric"]]
    last_epoch = model.fit(**fit_kwargs)
    if last_epoch != fit_params["epochs"]:          # RecSysExp.py:272-276
        fit_params["epochs"] = last_epoch - EARLY_STOPPING["allow_worse"] * EARLY_STOPPING["freq"]
    evaluator = spec["evaluator_class"](spec["URM_validation"], [spec["at"]], exclude_seen=True)
    results, text = evaluator.evaluateRecommender(model)
    engine = getattr(model, "engine", None)
    if engine is not None:
        engine.close()
    return {"fitness": -float(results[spec["at"]][spec["metric"]]), "fit_params": fit_params, "results_string": text,
            "seconds": time.time() - t0}


def _worker(spec, device, tasks, results, worker_id=0, engines=1):
    """One worker PROCESS per GPU slot running `engines` trial threads: every thread owns an engine (its own HIP stream) and
    runs whole trials; ctypes releases the GIL inside the library, so one trial's launch gaps and host-side evaluation are
    filled by another trial's kernels -- unlike several PROCESSES on one GPU, which time-slice it (20 ML-1M trials: 47.8 s
    with one process, 56.4 s with two).  Thread t announces its trials as worker (worker_id, t)."""
    try:
        if spec.get("visible_devices") is not None:      # one physical GPU per worker, seen as device 0
            os.environ["HIP_VISIBLE_DEVICES"] = str(spec["visible_devices"][device])
            device = 0

        def loop(t):
            while True:
                item = tasks.get()
                if item is None:
                    return
                idx, params = item
                results.put(("start", (worker_id, t), idx))        # lets the driver attribute a hard crash to this trial
                try:
                    # threads share numpy's global stream: each trial shuffles from its own (seeded by trial, reproducible)
                    rng = np.random.RandomState((spec["seed"] * 1000003 + idx) % (1 << 31)) if engines > 1 else None
                    out = run_trial(spec, params, device, schedule_rng=rng)
                except MemoryError as e:                      # the reference maps OOM to fitness 0 (RecSysExp.py:290-291)
                    out = {"fitness": 0.0, "fit_params": dict(params), "results_string": "out of memory: %s\n" % e, "seconds": 0.0}
                except Exception as e:                        # a failed trial must not take the search down
                    out = {"fitness": 0.0, "fit_params": dict(params), "results_string": "trial failed: %r\n" % (e,), "seconds": 0.0,
                           "error": repr(e)}
                results.put(("done", (worker_id, t), idx, params, out))

        if engines <= 1:
            return loop(0)
        import threading
        threads = [threading.Thread(target=loop, args=(t,), daemon=True) for t in range(engines)]
        for th in threads:
            th.start()
        for th in threads:
            th.join()
    except KeyboardInterrupt:
        return


# ---- the driver ----------------------------------------------------------------------------------------
class TrialParallelTuner(object):
    def __init__(self, recommender_class, URM_train_small, URM_early_stop, URM_validation, logsdir, mode="user",
                 metric="MAP", at=5, seed=1337, method="random", n_workers=None, devices=None, evaluator_class=None,
                 model_kwargs=None, isolate_devices=False, engines_per_worker=1):
        """devices: device ordinals to use (default: every visible HIP device); n_workers: worker processes (default:
        one per device; more than one per device is allowed); isolate_devices: give each worker only its GPU via
        HIP_VISIBLE_DEVICES; engines_per_worker: trial threads (each with its own engine and stream) inside every worker
        process -- the way to put more than one trial on a GPU at a time (_worker)."""
        if method not in ("random", "bayesian"):
            raise ValueError("method must be 'random' or 'bayesian'")
        if evaluator_class is None:
            from .evaluation import EvaluatorHoldoutFast as evaluator_class
        if devices is None:
            from . import _lib as L
            devices = list(range(max(1, L.load_library().ganmf_device_count())))
        self.devices = list(devices)
        self.n_workers = n_workers or len(self.devices)
        self.engines = max(1, int(engines_per_worker))
        self.method, self.seed, self.logsdir = method, seed, logsdir
        self.metric, self.at = metric, at
        os.makedirs(logsdir, exist_ok=True)
        n_users, n_items = URM_validation.shape
        self.dims = search_space(recommender_class.RECOMMENDER_NAME, n_users, n_items)
        self.spec = {"recommender_class": recommender_class, "URM_train_small": URM_train_small,
                     "URM_early_stop": URM_early_stop, "URM_validation": URM_validation, "mode": mode, "seed": seed,
                     "metric": metric, "at": at, "evaluator_class": evaluator_class, "model_kwargs": model_kwargs or {},
                     "visible_devices": self.devices if isolate_devices else None}
        self.x_iters, self.func_vals = [], []
        self.best_res, self.best_params = None, None
        self.poll_seconds = 2.0      # how often the driver looks for dead workers while waiting for results
        self.max_respawns = 16       # hard worker deaths tolerated before tune() raises

    # -- bookkeeping (results.txt / best_params.* / checkpoint.pkl as the reference leaves them)
    def _checkpoint_path(self):
        return os.path.join(self.logsdir, "checkpoint.pkl")

    def _load_checkpoint(self):
        if os.path.exists(self._checkpoint_path()):
            with open(self._checkpoint_path(), "rb") as f:
                ck = pickle.load(f)
            self.x_iters, self.func_vals = list(ck["x_iters"]), list(ck["func_vals"])
            self.best_res, self.best_params = ck.get("best_res"), ck.get("best_params")

    def _record(self, params, out):
        self.x_iters.append(OrderedDict(params))
        self.func_vals.append(out["fitness"])
        with open(os.path.join(self.logsdir, "results.txt"), "a") as f:
            f.write(json.dumps(out["fit_params"]) + "\n" + out["results_string"] + "\n\n")
        if self.best_res is None or out["fitness"] < self.best_res:      # RecSysExp.py:293-298
            self.best_res, self.best_params = out["fitness"], dict(out["fit_params"])
            with open(os.path.join(self.logsdir, "best_params.pkl"), "wb") as f:
                pickle.dump(self.best_params, f, pickle.HIGHEST_PROTOCOL)
        with open(self._checkpoint_path(), "wb") as f:
            pickle.dump({"x_iters": self.x_iters, "func_vals": self.func_vals, "best_res": self.best_res,
                         "best_params": self.best_params}, f, pickle.HIGHEST_PROTOCOL)

    # -- proposals
    def _unit(self, params):
        return np.array([d.to_unit(params[d.name]) for d in self.dims], dtype=np.float64)

    def _propose(self, rng, pending):
        sample = OrderedDict((d.name, d.sample(rng)) for d in self.dims)
        n_seen = len(self.func_vals)
        if self.method == "random" or n_seen < 10:
            return sample
        from scipy.stats import norm
        from sklearn.gaussian_process import GaussianProcessRegressor
        from sklearn.gaussian_process.kernels import ConstantKernel, Matern, WhiteKernel
        X = [self._unit(x) for x in self.x_iters] + [self._unit(p) for p in pending]
        y = list(self.func_vals) + [max(self.func_vals)] * len(pending)          # constant liar: pessimistic
        gp = GaussianProcessRegressor(ConstantKernel(1.0) * Matern(length_scale=np.ones(len(self.dims)), nu=2.5)
                                      + WhiteKernel(1e-3), normalize_y=True, random_state=rng.randint(1 << 30))
        gp.fit(np.array(X), np.array(y))
        cands = [OrderedDict((d.name, d.sample(rng)) for d in self.dims) for _ in range(2000)]
        mu, sd = gp.predict(np.array([self._unit(c) for c in cands]), return_std=True)
        best = min(self.func_vals)
        z = (best - mu) / np.maximum(sd, 1e-9)
        ei = (best - mu) * norm.cdf(z) + sd * norm.pdf(z)
        return cands[int(np.argmax(ei))]

    def tune(self, evals=50, verbose=True):
        """Run until `evals` trials are recorded (resumes from logsdir/checkpoint.pkl).  Returns (best fitness,
        best fit parameters)."""
        self._load_checkpoint()
        rng = np.random.RandomState(self.seed)
        for _ in range(len(self.func_vals)):      # a resumed random search continues its own stream
            [d.sample(rng) for d in self.dims]
        t_start = time.time()
        ctx = mp.get_context("spawn")             # never fork a process that has touched the GPU
        # results travel on a SimpleQueue: its put() writes to the pipe before returning (a Queue hands the item to a
        # feeder thread, and a worker that dies right after announcing a trial would take the announcement with it)
        # Tasks travel on ONE queue PER worker process (the driver places every trial): a shared task queue is read under a
        # cross-process lock, and a worker that dies hard while one of its threads sits in get() would take that lock with it.
        results = ctx.SimpleQueue()
        workers, queues, assigned = {}, {}, {}        # worker id -> process / its task queue / trial indices it holds
        reaped, running = set(), {}                   # ids of dead workers; (worker id, thread) -> announced trial index
        backlog = []                                  # trials to place (new ones and those a dead worker never announced)
        next_wid = [0]

        def spawn():
            wid = next_wid[0]
            next_wid[0] += 1
            slot = wid % len(self.devices)
            dev = slot if self.spec["visible_devices"] is not None else self.devices[slot]
            queues[wid], assigned[wid] = ctx.Queue(), set()
            workers[wid] = ctx.Process(target=_worker, args=(self.spec, dev, queues[wid], results, wid, self.engines), daemon=True)
            workers[wid].start()

        for _ in range(self.n_workers):
            spawn()
        pending, issued, done = {}, len(self.func_vals), len(self.func_vals)

        def finish(idx, params, out):
            nonlocal done
            if pending.pop(idx, None) is None:
                return                            # already recorded
            self._record(params, out)
            done += 1
            if verbose:
                print("trial %d/%d: %s@%d = %.6f in %.1f s%s" % (done, evals, self.metric, self.at, -out["fitness"],
                                                                 out["seconds"], "  [%s]" % out["error"] if "error" in out else ""),
                      flush=True)

        def crashed(idx, wid, exitcode):
            if idx in pending:
                p_dead = pending[idx]
                finish(idx, p_dead, {"fitness": 0.0, "fit_params": dict(p_dead), "seconds": 0.0,
                                     "results_string": "worker %d died (exit code %s)\n" % (wid, exitcode),
                                     "error": "worker died, exit code %s" % exitcode})

        def place():
            """hand backlog trials to the live workers with a free trial thread, least loaded first"""
            while backlog:
                live = [w for w in workers if len(assigned[w]) < self.engines]
                if not live:
                    return
                wid = min(live, key=lambda w: len(assigned[w]))
                idx = backlog.pop(0)
                assigned[wid].add(idx)
                queues[wid].put((idx, pending[idx]))

        def next_message():
            deadline = time.time() + self.poll_seconds
            while results.empty():
                if time.time() >= deadline:
                    return None
                time.sleep(0.02)
            return results.get()

        try:
            while done < evals:
                while issued < evals and len(pending) < self.n_workers * self.engines:
                    pending[issued] = self._propose(rng, list(pending.values()))
                    backlog.append(issued)
                    issued += 1
                place()
                msg = next_message()
                if msg is not None and msg[0] == "start":
                    _, wid, idx = msg             # wid = (process, thread)
                    if wid[0] in reaped:          # announced, died and was replaced before the announcement was read
                        crashed(idx, wid[0], "unknown")
                    else:
                        running[wid] = idx
                    continue
                if msg is not None:
                    _, wid, idx, params, out = msg
                    if running.get(wid) == idx:
                        del running[wid]
                    if wid[0] in assigned:
                        assigned[wid[0]].discard(idx)
                    finish(idx, params, out)
                    continue
                # Nothing arrived for poll_seconds.  A worker that died hard (HIP abort, GPU fault, OOM kill) never
                # reports: every trial one of its threads had announced is recorded as failed with fitness 0 (what the
                # reference does for an out-of-memory trial, RecSysExp.py:290-291), the trials it held without announcing
                # them go back to the backlog, and a fresh child process takes its place.
                for wid, w in list(workers.items()):
                    if w.is_alive():
                        continue
                    del workers[wid]
                    reaped.add(wid)
                    announced = set()
                    for key in [k for k in running if k[0] == wid]:
                        announced.add(running[key])
                        crashed(running.pop(key), wid, w.exitcode)
                    backlog.extend(sorted(i for i in assigned.pop(wid) if i not in announced and i in pending))
                    queues.pop(wid)
                    if len(reaped) > self.max_respawns:
                        raise RuntimeError("TrialParallelTuner: %d worker processes died; giving up" % len(reaped))
                    spawn()
        finally:
            for wid in workers:
                for _ in range(self.engines):
                    queues[wid].put(None)
            for w in workers.values():
                w.join(timeout=30)
                if w.is_alive():
                    w.terminate()
        elapsed = time.time() - t_start
        with open(os.path.join(self.logsdir, "results.txt"), "a") as f:
            f.write("Experiment ran for {:.1f} s with {} workers x {} engines on devices {}\n".format(elapsed, self.n_workers, self.engines, self.devices))
            f.write("Best {} score: {}. Best result found at: {}\n".format(self.metric, self.best_res, self.best_params))
        if self.best_params is not None:
            with open(os.path.join(self.logsdir, "best_params.txt"), "w") as f:
                f.write(json.dumps(self.best_params))
        return self.best_res, self.best_params
