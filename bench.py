#!/usr/bin/env python3
"""bench.py — GANMF training steps/sec on MI355X (BASELINE.json metric), one process per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Both forms work for N > 1: started without a launcher (`WORLD_SIZE` unset), `--gpus N` makes this process a launcher
itself -- it starts N rank processes of this script (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT
in their environment) BEFORE it has touched torch, HIP or the library, relays rank 0's JSON line and exits non-zero if any
rank does.  No process that has initialised the GPU is ever re-executed.

A *step* is one minibatch parameter update — one `sess.run([dtrain|gtrain, loss])` of the
reference (GANRec/GANMF.py:186-187,200-201) — on a batch of 128 user rows.  K timed steps are
K/2 discriminator updates followed by K/2 generator updates over the same slices, issued exactly
as fit() issues them (ganmf_train_epoch, one C call per pass of at most floor(U/128) slices).

Workload at every N: BASELINE.json configs[1] — "GANMF --user on MovieLens-1M shape
(6040 x 3706, k=250, batch=128)" with the reference's tuned hyper-parameters
(experiments/GANMF_user_1M/best_params.txt: emb_dim=992, m=10, d_lr=1e-4, g_lr=1.653e-4,
d_reg=1e-4, alpha=0.01) on a synthetic binary user x item matrix of that shape and density
(ganmf_amd/synthetic.py), resident in HBM as CSR before the timed region.  With N > 1 GPUs each
rank holds its own 6040-user shard (global users = 6040*N, weak scaling) and the gradients of the
replicated tensors are reduce-scattered over RCCL every step (Adam on the rank's slice, parameters
all-gathered: DESIGN.md section 6); `value` counts the 128-row minibatch
updates all ranks processed per second (= N x synchronous global steps/s).

Protocol: W untimed warm-up steps, then the K-step region FIVE times; every repeat is exactly K steps between a
barrier + synchronize on both sides and is timed by hipEvents on the library's own stream (ganmf_stream_timer) and by the host's
wall clock.  `value` = N * K / the MEDIAN event time (MAX over ranks per repeat); `timing` carries all five samples, their spread,
the wall-clock median and the per-call overhead.  With --gpus N > 1 (or GANMF_BENCH_FORCE_COMM=1 on one GPU) the line also carries
`parallelism` (per replicated tensor: collective time and exposed join-wait time per step; RCCL's own world size; rows/s) and
`configs3_sharded` (BASELINE.json configs[3]: one 25 000 x 50 000 shard per rank at emb_dim 32 and 1024, same protocol).

SURVEY 8(d)'s full report: the reference runs its discriminator passes and its generator passes as separate loops (GANRec/GANMF.py:176-189
vs 191-203), so beside `value` (whole-epoch steps/s: D and G passes as fit() issues them) the line carries `d_steps_per_s` and `g_steps_per_s`
(D-only / G-only ganmf_train_epoch calls over the same slices, event-timed, median of five, right after the timed region) and
`cpu_baseline.value_1thread` beside the all-threads figure.

Extra objects on the JSON line: `roofline` (dominant DEVICE FUNCTION of the step = gemm_bf16k_mfma<false, true, 3, false>, the 16-wave split-bf16 GEMM of
gemm_bf16k.hpp with a K-major B -- fp32 in, fp32-accurate, priced against the fp32 MFMA peak -- over ALL of its launch classes: `frac` = sum of
algorithmic FLOPs / sum of launch durations; `frac_best_class` = its best class alone); `roofline_fused_adam` = the HBM-bound launch of the two fused-Adam weight-gradient GEMMs; HIP-event timed on the
library's stream in a profiled repeat of the same steps (at least 96) right after the timed region — events
stay out of the timed region so that `value` is not perturbed), `cpu_baseline` (the numpy fp32
oracle = a port of the reference's per-step procedure, timed on this box's host cores on a
bounded sample), `kernels` (per-kernel-class table) and `scoring_gemm` (the 6040x3706x250
generator/scoring GEMM the north star quotes its MFMA target on).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 FLOP/clk/CU x 256 CU x 2.4 GHz
PEAK_BF16_MFMA_TFLOPS = 2516.6  # same guide: v_mfma_f32_32x32x16_bf16, 4096 FLOP/clk/CU x 256 CU x 2.4 GHz
PEAK_HBM_GBS = 8000.0          # HBM3E spec; ~6.3 TB/s achievable
REPEATS = 5                    # timed repeats of the K-step region (value = median)

C2 = dict(U=6040, N=3706, k=250, e=992, B=128, density=0.035,
          hp=dict(d_lr=1e-4, g_lr=0.0001653241474168571, d_reg=1e-4, g_reg=0.0, m=10.0, recon_coefficient=0.01))


def run_steps(eng, perm, B, n_steps):
    """EXACTLY n_steps updates: ceil(n/2) D updates and floor(n/2) G updates, in chunks of full batches -- each chunk one
    ganmf_train_epoch call of c D updates followed by c G updates over the same slices (the call fit() makes); an odd
    n_steps ends with one call that runs a single D update (d_steps=1, g_steps=0)."""
    per_call = (len(perm) // B)
    half = n_steps // 2
    done = 0
    while done < half:
        c = min(per_call, half - done)
        eng.train_epoch(perm[:c * B], 1, 1)
        done += c
    if n_steps % 2:
        eng.train_epoch(perm[:B], 1, 0)
    return n_steps


def run_pass_steps(eng, perm, B, n_steps, kind):
    """n_steps updates of ONE kind ("D" | "G"): the passes of run_steps as calls of their own (ganmf_train_epoch with d_steps = 1, g_steps = 0
    or the reverse) -- the reference's two loops, GANRec/GANMF.py:176-189 and 191-203, timed apart."""
    per_call = len(perm) // B
    done = 0
    while done < n_steps:
        c = min(per_call, n_steps - done)
        eng.train_epoch(perm[:c * B], 1 if kind == "D" else 0, 0 if kind == "D" else 1)
        done += c
    return n_steps


# The dominant DEVICE FUNCTION of the step (most kernel time of any one symbol in the rocprofv3 trace, profiles/r0N_bench_kernel_summary.md) and
# the launch classes that run it under the default plan: the encode product of both steps and the discriminator step's decode (the generator
# step's decode runs gemm_bf16w_mfma<true>, its dE gemm_bf16k_mfma<false, false, ...>, the D-step's dE rides in de_dcoef_kernel).
DOMINANT_FN = "gemm_bf16k_mfma<false, true, 3, false>"
DOMINANT_FN_CLASSES = (("D", "gemm_encode[2B,N]x[N,e]"), ("G", "gemm_encode[2B,N]x[N,e]"), ("D", "gemm_decode[2B,e]x[e,N]"))
TRAFFIC_KEY = {"D": " (D-step)", "G": " (G-step)"}


def roofline_objects(prof_d, prof_g, traffic=None, traffic_source=None):
    """`roofline` (+ `roofline_fused_adam`, the per-class `kernels` table) from the library's profiled D-only and G-only repeats.
    `roofline.frac` is the dominant device function over ALL of its classes: sum of algorithmic FLOPs / sum of launch durations / peak;
    `frac_best_class` its best class alone; `frac_time_weighted` every launch class of the 16-wave GEMM family (combined launches included)."""
    rows = [dict(p, step="D") for p in prof_d] + [dict(p, step="G") for p in prof_g]
    kernels = []
    for p in rows:
        avg_ms = p["ms"] / max(p["launches"], 1)
        row = {"step": p["step"], "name": p["name"], "launches": p["launches"], "avg_us": round(avg_ms * 1e3, 2), "total_ms": round(p["ms"], 3)}
        if p["flops"] > 0:
            row["tflops"] = round(p["flops"] / max(p["ms"], 1e-9) / 1e9, 2)
        if p["bytes"] > 0:
            row["gbs"] = round(p["bytes"] / max(p["ms"], 1e-9) / 1e6, 1)
        kernels.append(row)
    gemms = [p for p in rows if p["flops"] > 0 and p["ms"] > 0]
    fam = [p for p in gemms if not ("gWd" in p["name"] or "gWe" in p["name"] or "gV" in p["name"])]
    fused = [p for p in gemms if "gWd" in p["name"] or "gWe" in p["name"]]
    dom = [p for p in fam if (p["step"], p["name"]) in DOMINANT_FN_CLASSES] or fam or gemms      # (another plan: the whole family)
    roofline = None
    if dom:
        fl, ms, n = sum(p["flops"] for p in dom), sum(p["ms"] for p in dom), sum(p["launches"] for p in dom)
        tf = fl / ms / 1e9
        classes = [{"step": p["step"], "name": p["name"], "launches": p["launches"], "avg_launch_us": round(p["ms"] / p["launches"] * 1e3, 2),
                    "achieved": round(p["flops"] / p["ms"] / 1e9, 2), "frac": round(p["flops"] / p["ms"] / 1e9 / PEAK_F32_MFMA_TFLOPS, 4)} for p in dom]
        best = max(classes, key=lambda c: c["frac"])
        roofline = {"kernel": DOMINANT_FN + " (gemm_bf16k.hpp; all of its launch classes)", "bound": "mfma", "achieved": round(tf, 2),
                    "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / PEAK_F32_MFMA_TFLOPS, 4), "traffic": None,
                    "flops_per_launch": fl / n, "avg_launch_us": round(ms / n * 1e3, 2), "launches": n, "classes": classes,
                    "frac_best_class": best["frac"], "best_class": "%s:%s" % (best["step"], best["name"])}
        if fam:
            fam_tf = sum(p["flops"] for p in fam) / sum(p["ms"] for p in fam) / 1e9
            roofline["frac_time_weighted"] = round(fam_tf / PEAK_F32_MFMA_TFLOPS, 4)
            roofline["achieved_time_weighted"] = round(fam_tf, 2)
            roofline["classes_time_weighted"] = ["%s:%s" % (p["step"], p["name"]) for p in fam]
        if traffic:      # HBM bytes per launch from the committed rocprofv3 PMC passes, launch-weighted over the function's classes
            hit = [(p, traffic.get(p["name"] + TRAFFIC_KEY[p["step"]])) for p in dom]
            if hit and all(v for _, v in hit):
                roofline["traffic"] = round(sum(v["hbm_bytes_per_launch"] * p["launches"] for p, v in hit) / n)
                roofline["algorithmic_bytes_per_launch"] = round(sum(v["algorithmic_bytes"] * p["launches"] for p, v in hit) / n)
                roofline["traffic_source"] = traffic_source
    roofline_fused = None
    if fused:
        fa_ms = sum(p["ms"] for p in fused)
        fa_n = max(p["launches"] for p in fused)          # per discriminator step
        fa_bytes = sum(p["bytes"] for p in fused)          # operands once + the six Adam streams (theta, m, v in and out)
        roofline_fused = {"kernel": " + ".join(p["name"] for p in fused), "bound": "hbm",
                          "achieved": round(fa_bytes / fa_ms / 1e6, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                          "frac": round(fa_bytes / fa_ms / 1e6 / PEAK_HBM_GBS, 4), "traffic": None,
                          "algorithmic_bytes_per_step": round(fa_bytes / fa_n),
                          "avg_us_per_step": round(fa_ms / fa_n * 1e3, 2)}
        if traffic:
            fc = [v for k, v in traffic.items() if any(k.startswith(p["name"].replace(" (one launch)", "")) for p in fused)]
            if fc:
                roofline_fused["traffic"] = sum(v["hbm_bytes_per_launch"] for v in fc)
                roofline_fused["traffic_source"] = traffic_source
    if roofline is not None:
        step_flops, step_ms = sum(p["flops"] for p in rows), sum(p["ms"] for p in rows)
        roofline["whole_step_tflops_kernel_time"] = round(step_flops / max(step_ms, 1e-9) / 1e9, 2)
    return roofline, roofline_fused, kernels


def cpu_baseline(urm, params, w, seconds):
    """Port of the reference's per-step procedure (densify + dense GEMMs + dense TF-Adam) in numpy
    fp32 on the host cores; bounded sample of the same workload."""
    from oracle.ganmf_oracle import GANMFOracle   # checker / baseline only, never the product path
    o = GANMFOracle(w["U"], w["N"], w["k"], w["e"], dtype=np.float32, **w["hp"])
    o.set_params(**params)
    rng = np.random.RandomState(7)
    perm = rng.permutation(w["U"])
    B = w["B"]

    def pair(i):
        uids = perm[(i * B) % (w["U"] - B):][:B]
        X = np.asarray(urm[uids].toarray(), dtype=np.float32)   # the reference's per-step densify
        o.d_step(uids, X)
        X = np.asarray(urm[uids].toarray(), dtype=np.float32)
        o.g_step(uids, X)
    pair(0)  # warm-up

    def sample(budget, first):
        n, t0 = 0, time.perf_counter()
        while True:
            pair(first + n)
            n += 1
            el = time.perf_counter() - t0
            if el >= budget or n >= 200:
                return n, el
    # two thirds of the budget on all threads of the BLAS pool (`value`, `cores`), one third on ONE thread (`value_1thread`: BASELINE.md section 2
    # quotes the reference's CPU path both ways)
    # The pool never gets more threads than this process may run on (affinity mask and cgroup CPU quota: a 64-thread OpenBLAS pool on a 16-CPU share
    # of the box ran this workload at 8 steps/s against 19 on ONE thread -- oversubscription, not the CPU's speed).
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            avail = max(1, min(avail, int(float(quota) / float(period))))
    except Exception:
        pass
    threads = avail
    out1 = {}
    try:
        from threadpoolctl import threadpool_info, threadpool_limits
        pool = max([p.get("num_threads", 1) for p in threadpool_info()] + [1])
        threads = max(1, min(pool, avail))
        with threadpool_limits(limits=threads):
            n, el = sample(seconds * 2.0 / 3.0, 1)
        with threadpool_limits(limits=1):
            n1, el1 = sample(seconds / 3.0, n + 1)
        out1 = {"value_1thread": round(2 * n1 / el1, 3),
                "sample_1thread": "%d D + %d G updates, BLAS pool limited to one thread (threadpoolctl), %.1f s" % (n1, n1, el1)}
    except Exception as ex:      # (no threadpoolctl, or it cannot steer this BLAS: the pool's own thread count, no one-thread figure -- never lose the line to it)
        n, el = sample(seconds, 1)
        threads = os.cpu_count() or 1
        out1 = {"value_1thread": None, "sample_1thread": "not measured: %s" % ex}
    out = {"value": round(2 * n / el, 3), "unit": "steps/s", "cores": int(threads), "kind": "port",
           "sample": "%d D + %d G updates (B=%d) of the same synthetic workload, numpy fp32 oracle incl. "
                     "URM[uids].toarray() densify and dense Adam, %.1f s" % (n, n, B, el)}
    out.update(out1)
    return out


def launch_ranks(n, argv, script=None):
    """`python bench.py --gpus N` without a launcher: start the N rank processes ourselves.  Runs before this process has
    imported torch or loaded the HIP library (nothing here touches a GPU); the children are fresh interpreters."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, script or os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr))
    # rank 0's stdout (the JSON line) is drained by a thread while every child is polled: a rank that dies would leave its
    # peers waiting in a collective for ever, so the first non-zero exit ends the others (their exact PIDs)
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    failed = False
    while any(p.poll() is None for p in procs):
        if any(p.poll() not in (None, 0) for p in procs):
            failed = True
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            for p in procs:
                try:
                    p.wait(timeout=20)
                except subprocess.TimeoutExpired:
                    p.kill()
            break
        time.sleep(0.2)
    reader.join(timeout=10)
    codes = [p.wait() for p in procs]
    if failed or any(codes):
        sys.stderr.write("bench.py: rank exit codes %s\n" % codes)
        raise SystemExit(1)
    line = b"".join(chunks)
    sys.stdout.write(line.decode())
    sys.stdout.flush()


C4_SHARD = dict(U=25000, N=50000, k=250, B=128, density=0.01,      # one rank's share of BASELINE.json configs[3] (200 k x 50 k over 8 GPUs)
                hp=dict(d_lr=1e-4, g_lr=1e-4, d_reg=1e-4, g_reg=0.0, m=10.0, recon_coefficient=0.01))


def parallelism_object(eng, prof, world, steps_profiled, rows_per_s):
    """What the data-parallel step spent on its collectives, from the library's profiled repeat (HIP events per class):
    per replicated tensor the reduce-scatter + all-gather time on the side lane, the Adam pass on the rank's slice, and the time the
    MAIN lane waited at the join in front of the tensor's next reader -- the part of the collective that was NOT hidden.
    Per step = per minibatch update (D and G steps counted alike, as `value` counts them)."""
    by = {}
    for p in prof:      # (the D-only and the G-only profiled repeat may both hold a class: summed)
        q = by.setdefault(p["name"], {"ms": 0.0})
        q["ms"] += p["ms"]

    def us(name):
        return round(by[name]["ms"] * 1e3 / max(steps_profiled, 1), 2) if name in by else 0.0
    comm_world, comm_rank = eng.comm_info()
    tensors = {}
    for t in ("We", "Wd", "V"):
        tensors[t] = {"collective_us_per_step": us("collective_%s (reduce-scatter + all-gather)" % t),
                      "exposed_us_per_step": us("join_wait_%s (main lane)" % t)}
    coll = sum(v["collective_us_per_step"] for v in tensors.values())
    exposed = sum(v["exposed_us_per_step"] for v in tensors.values())
    return {"scheme": "users sharded row-wise; per step reduce-scatter of the gradient, TF-Adam on the rank's slice, all-gather of the "
                      "parameter (We, Wd in D steps, V in G steps) on a side stream; 2-float all-reduce before the hinge",
            "rccl_world_size": comm_world, "rccl_rank": comm_rank, "launcher_world_size": world,
            "per_tensor": tensors, "collective_us_per_step": round(coll, 2), "exposed_collective_us_per_step": round(exposed, 2),
            "adam_slice_us_per_step": round(us("adam_dense_D") + us("adam_dense_V"), 2),
            "small_allreduce_us_per_step": us("rccl_allreduce"),
            "rows_per_s": round(rows_per_s, 1),
            "note": "timed in the profiled repeat (events around every launch and join: slower than the timed region)"}


REPLICATED = (("We", 0), ("be", 1), ("Wd", 2), ("bd", 3), ("V", 101))      # GANMF tensors every rank holds a full copy of


def replica_check(eng, world, dist=None, crc=None):
    """The invariant of the reduce-scatter -> Adam-on-the-slice -> all-gather design (csrc/lib/dataparallel.inc dp_update): after any
    number of steps every rank holds BITWISE the same replicated tensors, and RCCL's communicator has as many ranks as the launcher
    started.  CRC-32C (ganmf_crc32c) of each tensor on every rank, gathered over the gloo control plane; every rank returns the same
    verdict.  A peer-write ordering bug on the side lane shows here before it shows in a loss."""
    if crc is None:
        from ganmf_amd.tf_bundle import crc32c as crc
    mine = {name: int(crc(np.ascontiguousarray(eng.get_tensor(tid), dtype=np.float32))) for name, tid in REPLICATED}
    mine["rccl_world_size"] = int(eng.comm_info()[0])
    everyone = [mine]
    if world > 1:
        everyone = [None] * world
        dist.all_gather_object(everyone, mine)
    crcs = {}
    for name, _ in REPLICATED:
        vals = ["%08x" % v[name] for v in everyone]
        crcs[name] = vals[0] if len(set(vals)) == 1 else vals      # (one value when the ranks agree, the per-rank list when they do not)
    return {"replicas_bitwise_equal": all(isinstance(v, str) for v in crcs.values()),
            "rccl_world_size_equals_launcher_world_size": all(v["rccl_world_size"] == world for v in everyone),
            "ranks_checked": len(everyone), "crc32c": crcs}


def replicas_ok(check):
    return bool(check["replicas_bitwise_equal"] and check["rccl_world_size_equals_launcher_world_size"])


def extra_workload(w, e, world, rank, local_rank, steps, warmup, sync, torch, dist, force_comm):
    """One more data-parallel line (bench.py --gpus N > 1): `w` with emb_dim e, every rank its own shard; same protocol as the
    headline (REPEATS event-timed repeats of `steps` steps, MAX over ranks, median)."""
    from ganmf_amd.engine import Engine, comm_unique_id
    from ganmf_amd.synthetic import glorot_params, synthetic_urm
    # Phase 1 -- everything that can fail on ONE rank alone (host memory, device memory, bad shapes) -- ends with an agreement over
    # gloo: either every rank goes on to the collectives or none does (a rank that skipped ahead would leave its peers inside an
    # RCCL call for ever, and the headline line of this run with them).
    eng, err = None, None
    try:
        urm = synthetic_urm(w["U"], w["N"], w["density"], seed=4242 + rank)
        params = glorot_params(w["U"], w["N"], w["k"], e, seed=4242)
        if rank:
            params["U"] = glorot_params(w["U"], 8, w["k"], 8, seed=4242 + rank)["U"]
        eng = Engine(w["U"], w["N"], w["k"], e, w["B"], device=local_rank, world_size=world, rank=rank, row_offset=rank * w["U"], **w["hp"])
        eng.set_urm(urm)
        for name, tid in (("We", 0), ("be", 1), ("Wd", 2), ("bd", 3), ("U", 100), ("V", 101)):
            eng.set_tensor(tid, params[name])
    except BaseException as ex:      # (MemoryError included)
        err = "%s: %s" % (type(ex).__name__, ex)
    ok = 0 if err else 1
    if world > 1:
        t = torch.tensor([ok], dtype=torch.int32)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        ok = int(t.item())
    if not ok:
        if eng is not None:
            eng.close()
        raise RuntimeError(err or "another rank could not set this workload up")
    ids = [comm_unique_id() if rank == 0 else None]
    if world > 1:
        dist.broadcast_object_list(ids, src=0)
    eng.comm_init(ids[0])
    perm = np.random.RandomState(99 + rank).permutation(w["U"]).astype(np.int32)
    per_call = len(perm) // w["B"]

    def run(n):
        half, done = n // 2, 0
        while done < half:
            c = min(per_call, half - done)
            eng.train_epoch(perm[:c * w["B"]], 1, 1, steps_per_pass=c, global_batch_rows=np.full(c, world * w["B"], np.int32))
            done += c
        if n % 2:
            eng.train_epoch(perm[:w["B"]], 1, 0, steps_per_pass=1, global_batch_rows=np.full(1, world * w["B"], np.int32))
    run(max(warmup, 4))
    ev_s = []
    for _ in range(REPEATS):
        sync()
        eng.timer_start()
        run(steps)
        ev = eng.timer_stop() * 1e-3
        sync()
        if world > 1:
            t = torch.tensor([ev], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            ev = float(t.item())
        ev_s.append(ev)
    el = float(np.median(ev_s))
    eng.profile(True)
    run(max(steps, 16))
    prof = eng.profile_read()
    eng.profile(False)
    out = {"workload": "one rank's shard of BASELINE.json configs[3] per GPU: %d x %d, k=%d, emb_dim=%d, batch=%d/GPU"
                       % (w["U"], w["N"], w["k"], e, w["B"]),
           "value": round(world * steps / el, 2), "unit": "steps/s", "ms_per_step": round(el / steps * 1e3, 4),
           "value_samples": [round(world * steps / t, 2) for t in ev_s],
           "parallelism": parallelism_object(eng, prof, world, max(steps, 16), world * steps * w["B"] / el)}
    out["parallelism"].update(replica_check(eng, world, dist))
    eng.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)      # (>= 200 timed steps, median of five repeats)
    ap.add_argument("--warmup", type=int, default=96)
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args.gpus, sys.argv[1:])
    # stdout carries exactly ONE line (the JSON).  RCCL prints a version banner through C stdio that is
    # flushed at exit, after Python's own prints: park fd 1 on stderr for the whole run and write the JSON
    # line to the saved descriptor at the very end.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("GANMF_BENCH_ONE_DEVICE") == "1":      # rehearsal on a one-GPU box: every rank on device 0 (if RCCL accepts it)
        local_rank = 0
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE=%d but --gpus %d" % (world, args.gpus))
    steps = max(1, args.steps)
    warmup = max(0, args.warmup)

    # One GPU: no torch at all -- ganmf_train_epoch is blocking (it returns after its stream has drained), which is the
    # synchronisation the timed region needs.  N > 1: torch.distributed (gloo) is the control plane (unique-id broadcast,
    # barriers, the MAX over ranks) and torch.cuda.synchronize() brackets the timed region as the contract asks.
    torch = dist = None
    if world > 1:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)   # control plane only
        if torch.cuda.is_available():
            torch.cuda.set_device(local_rank)      # torch.cuda.synchronize() below must target this rank's GPU

    from ganmf_amd.engine import Engine, comm_unique_id
    from ganmf_amd.synthetic import glorot_params, synthetic_urm

    w = C2
    urm = synthetic_urm(w["U"], w["N"], w["density"], seed=1337 + rank)
    params = glorot_params(w["U"], w["N"], w["k"], w["e"], seed=1337)
    if rank:
        params["U"] = glorot_params(w["U"], 8, w["k"], 8, seed=1337 + rank)["U"]   # own user rows
    force_comm = world == 1 and os.environ.get("GANMF_BENCH_FORCE_COMM") == "1"   # exercise the RCCL path on one GPU
    if force_comm:
        os.environ["GANMF_FORCE_COLLECTIVES"] = "1"      # (read when the handle is created) the one-rank communicator issues its
                                                         # in-place ncclReduceScatter / ncclAllGather calls instead of skipping them
    eng = Engine(w["U"], w["N"], w["k"], w["e"], w["B"], device=local_rank, world_size=world, rank=rank,
                 row_offset=rank * w["U"], **w["hp"])
    eng.set_urm(urm)
    for name, tid in (("We", 0), ("be", 1), ("Wd", 2), ("bd", 3), ("U", 100), ("V", 101)):
        eng.set_tensor(tid, params[name])
    if world > 1 or force_comm:
        ids = [comm_unique_id() if rank == 0 else None]
        if world > 1:
            dist.broadcast_object_list(ids, src=0)
        eng.comm_init(ids[0])
        _train = eng.train_epoch

        def train_dp(perm, d, g):   # every slice is full on every rank: global rows = world*B
            n = len(perm) // w["B"]
            return _train(perm, d, g, steps_per_pass=n, global_batch_rows=np.full(n, world * w["B"], np.int32))
        eng.train_epoch = train_dp

    def sync():
        if world > 1:
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()

    # ---- side measurement first: the generator / scoring GEMM (every rank runs it, rank 0 reports it).  ~20 ms of sustained GPU
    # work right before the warm-up: the timed region then starts at the clocks a training run holds -- a short K would
    # otherwise be timed on the power-management ramp of a GPU that has just been handed its first kernels (20 steps: 2.83 ms
    # cold against 2.66 ms after 96 warm-up steps and 2.60 ms in a tight loop, tools/epoch_overhead.py)
    e32 = None
    try:
        e32 = Engine(w["U"], w["N"], w["k"], w["e"], w["B"], device=local_rank, mfma="f32", **w["hp"])
        e32.set_tensor(100, params["U"])
        e32.set_tensor(101, params["V"])
    except Exception as ex:   # never lose the bench line to the side measurement
        plain32 = {"error": str(ex)}
    eng.bench_scores(w["U"], transposed=False, iters=100)          # (clock ramp: the first ~20 ms of work on an idle GPU run slower)
    ms_sc = eng.bench_scores(w["U"], transposed=False, iters=100)
    sc_tf = 2.0 * w["U"] * w["N"] * w["k"] / ms_sc / 1e9
    user_tune = os.environ.get("GANMF_TUNE")          # (a caller's own plan overrides stay in force for the engines built later)
    os.environ["GANMF_TUNE"] = (user_tune + "," if user_tune else "") + "score_product=1"      # (read per call) the whole product: both split passes + the GEMM launch
    ms_prod = eng.bench_scores(w["U"], transposed=False, iters=50)
    if user_tune is None:
        del os.environ["GANMF_TUNE"]
    else:
        os.environ["GANMF_TUNE"] = user_tune
    scoring = {"shape": [w["U"], w["N"], w["k"]], "ms": round(ms_sc, 4), "achieved": round(sc_tf, 2),
               "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(sc_tf / PEAK_F32_MFMA_TFLOPS, 4),
               "kernel": "gemm_bf16p_persist: persistent 8-wave tile walk over operands split ONCE into three bf16 planes "
                         "(presplit_rows_kernel); `ms` = the GEMM launch, `product_ms` = split pass of both factors + GEMM",
               "product_ms": round(ms_prod, 4),
               "arithmetic": "f32 in/out; K loop = 3-way exact bf16 split, 6 piece products on "
                             "v_mfma_f32_32x32x16_bf16, f32 accumulate",
               "executed_bf16_tflops": round(6 * sc_tf, 1), "peak_bf16": PEAK_BF16_MFMA_TFLOPS,
               "frac_of_bf16_peak": round(6 * sc_tf / PEAK_BF16_MFMA_TFLOPS, 4)}
    # the plain fp32-MFMA kernel last: it holds the clock the step's fp32 GEMMs run at (2.0 GHz; the bf16 kernel is power-bound
    # at 1.85 GHz and would hand the warm-up steps a governor that is still ramping back up); its engine is closed after the
    # timed region so that nothing but kernels sits between this loop and the warm-up steps
    if e32 is not None:
        try:
            e32.bench_scores(w["U"], transposed=False, iters=100)
            ms32 = e32.bench_scores(w["U"], transposed=False, iters=100)
            tf32 = 2.0 * w["U"] * w["N"] * w["k"] / ms32 / 1e9
            plain32 = {"ms": round(ms32, 4), "achieved": round(tf32, 2), "frac": round(tf32 / PEAK_F32_MFMA_TFLOPS, 4)}
        except Exception as ex:
            plain32 = {"error": str(ex)}
    scoring["plain_f32_mfma"] = plain32
    scoring_pre = scoring

    perm = np.random.RandomState(1337 + rank).permutation(w["U"]).astype(np.int32)
    if warmup:
        run_steps(eng, perm, w["B"], warmup)
    # ---- the timed region, REPEATS times (hipEvent timing, median of 5).  Every repeat is EXACTLY `steps` steps,
    # bracketed by the barrier + synchronize the contract asks for; its duration is read twice: hipEvents on the library's own
    # stream (the time the device spent on the K steps, host gaps between the blocking ganmf_train_epoch calls included) and the
    # host's wall clock around the same calls.  `value` is formed from the MEDIAN event time (MAX over ranks per repeat);
    # the wall-clock median and the difference per call (launch + drain latency of a blocking call) are reported beside it.
    calls_per_repeat = (steps // 2 + (len(perm) // w["B"]) - 1) // max(len(perm) // w["B"], 1) + (steps % 2)
    ev_s, wall_s = [], []
    done = steps
    for _ in range(REPEATS):
        sync()
        t0 = time.perf_counter()
        eng.timer_start()
        done = run_steps(eng, perm, w["B"], steps)      # blocking: returns after the stream drained
        ev = eng.timer_stop() * 1e-3
        sync()
        wl = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([ev, wl], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            ev, wl = float(t[0].item()), float(t[1].item())
        ev_s.append(ev)
        wall_s.append(wl)
    el = float(np.median(ev_s))
    timing = {"repeats": REPEATS, "clock": "hipEvents on the library's stream around the K steps (median; MAX over ranks per repeat)",
              "value_samples": [round(world * done / t, 2) for t in ev_s],
              "value_spread": round((max(ev_s) - min(ev_s)) / el, 4),
              "value_wall_clock_median": round(world * done / float(np.median(wall_s)), 2),
              "calls_per_repeat": int(max(calls_per_repeat, 1)),
              "per_call_overhead_us": round(float(np.median(np.array(wall_s) - np.array(ev_s))) / max(calls_per_repeat, 1) * 1e6, 1)}

    if e32 is not None:
        e32.close()
    # ---- profiled repeat (HIP events around every launch, on the library's stream) --------------
    # ---- the two passes timed apart (D-steps/s, G-steps/s beside the whole-epoch rate): K/2 updates of one kind per repeat, as calls of
    # their own over the same slices, same clock and protocol as the timed region
    # (at least one whole pass over the shard per repeat, as the reference's loops run them: a 10-update call would mostly time its own set-up)
    half = max(steps // 2, len(perm) // w["B"], 1)
    split_rate, split_samples = {}, {}
    for kind in ("D", "G"):
        evs = []
        for _ in range(REPEATS):
            sync()
            eng.timer_start()
            run_pass_steps(eng, perm, w["B"], half, kind)
            ev = eng.timer_stop() * 1e-3
            sync()
            if world > 1:
                t = torch.tensor([ev], dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                ev = float(t.item())
            evs.append(ev)
        split_rate[kind] = world * half / float(np.median(evs))
        split_samples[kind] = [round(world * half / t, 2) for t in evs]
    # ---- profiled repeats (HIP events around every launch, on the library's stream), D passes and G passes apart so that a launch class is a
    # (step kind, class) pair -- the discriminator step's decode and the generator step's are different kernels
    prof_steps = max(steps, 96) // 2      # (at least 48 launches per class: a short K alone averages over too few)
    eng.profile(True)
    run_pass_steps(eng, perm, w["B"], prof_steps, "D")
    prof_d = eng.profile_read()
    eng.profile(True)                     # (clears the records)
    run_pass_steps(eng, perm, w["B"], prof_steps, "G")
    prof_g = eng.profile_read()
    eng.profile(False)
    prof = prof_d + prof_g

    # ---- data-parallel runs verify themselves: replicas bitwise equal after all of the above, RCCL's world = the launcher's ----
    check = replica_check(eng, world, dist) if (world > 1 or force_comm) else None
    failed_checks = [] if check is None or replicas_ok(check) else ["configs[1]"]

    out = None
    if rank == 0:
        # Dominant DEVICE FUNCTION of the step = the symbol with the most kernel time: gemm_bf16k_mfma<false, true, 3, false> (encode of both
        # steps, decode of the discriminator step); `roofline.frac` covers all of its launches, `frac_best_class` the best class alone,
        # `frac_time_weighted` every class of the 16-wave GEMM family.  The fused-Adam weight-gradient launch, HBM-bound, has its own object.
        # HBM bytes per launch from the committed rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE in separate runs, gfx950 read correction
        # applied: tools/collect_traffic.py); null when no profile matches
        traffic, traffic_source = None, None
        try:
            import glob
            tf = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")))[-1]
            traffic, traffic_source = json.load(open(tf)), os.path.basename(tf)
        except Exception:
            pass
        roofline, roofline_fused, kernels = roofline_objects(prof_d, prof_g, traffic, traffic_source)
        # generator / scoring GEMM at the shape the north star quotes (6040 x 3706, k = 250).  The library runs it
        # on the bf16 matrix cores with every fp32 operand split EXACTLY into three bf16 pieces and the six piece
        # products of weight >= 2^-16 accumulated in fp32 (fp32-accurate: tests/test_gpu_mfma_modes.py); `achieved`
        # counts the ALGORITHMIC 2MNK flops against the fp32 MFMA roof (the dtype of operands and result); the
        # executed bf16 flops (6x) against the bf16 roof and the plain fp32-MFMA kernel are reported beside it.
        scoring = scoring_pre
        out = {
            "metric": "GANMF training steps/sec", "value": round(world * done / el, 2), "unit": "steps/s",
            "n_gpus": world, "steps": done, "warmup": warmup, "ms_per_step": round(el / done * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "GANMF --user, MovieLens-1M shape %dx%d per GPU, k=%d, emb_dim=%d, batch=%d/GPU, "
                                   "tuned hyper-parameters (BASELINE.json configs[1])" % (w["U"], w["N"], w["k"], w["e"], w["B"]),
                       "step": "one 128-row minibatch update (D or G), K/2 D then K/2 G",
                       "arithmetic": "float32 tensors throughout; the K loops of the step's GEMMs run the fp32-accurate split-bf16 loop "
                                     "(3 exact bf16 pieces per operand, 6 piece products on v_mfma_f32_32x32x16_bf16, fp32 accumulate) in "
                                     "16-wave workgroups (gemm_bf16k.hpp) or, for the two fused-Adam weight-gradient GEMMs, 4-wave "
                                     "workgroups; gUb + gV on the fp32 MFMA",
                       "launches": "D-step 6, G-step 8, plus per pass: two launches in front of a discriminator pass (the CSR rows of all its full minibatches -- that launch also writes lr_t of every step of the pass -- and their generated rows: the generator is frozen during the pass) and two around a generator pass (the embeddings of the scheduled rows advanced to their step, ONE all-rows Adam over U behind the pass instead of one per step: with g_reg = 0 a row is read once and has a gradient once per pass) (dE + d_coef, gWd + gWe, gUb + gV share a launch; the generator step: generator GEMM + CSR rows, encode, its slab sum, decode on 64 x 32 tiles unsplit, dE, its slab sum, dF unsplit, gUb + gV)",
                       "global_steps_per_s": round(done / el, 2), "rows_per_s": round(world * done * w["B"] / el, 1),
                       "parallelism": "dp%d (users sharded row-wise; RCCL reduce-scatter of the D and V gradients, Adam on the rank's "
                                      "slice, all-gather of the parameters)" % world},
            "d_steps_per_s": round(split_rate["D"], 2), "g_steps_per_s": round(split_rate["G"], 2),
            "pass_rates": {"protocol": "max(K/2, one pass over the shard) = %d updates of one kind per repeat as D-only / G-only ganmf_train_epoch calls over the same slices, hipEvent-timed, "
                                       "median of %d, right after the timed region (the reference's two loops, GANRec/GANMF.py:176-189 / 191-203)" % (half, REPEATS),
                           "d_samples": split_samples["D"], "g_samples": split_samples["G"],
                           "harmonic_mean_steps_per_s": round(2.0 / (1.0 / split_rate["D"] + 1.0 / split_rate["G"]), 2),
                           "kernel_us_per_step": {"D": round(sum(p["ms"] for p in prof_d) * 1e3 / max(prof_steps, 1), 2),
                                                  "G": round(sum(p["ms"] for p in prof_g) * 1e3 / max(prof_steps, 1), 2)}},
            "timing": timing, "roofline": roofline, "roofline_fused_adam": roofline_fused, "scoring_gemm": scoring, "kernels": kernels,
            "reference_derived_steps_per_s": 84.0,
        }
        if world > 1 or force_comm:
            out["parallelism"] = parallelism_object(eng, prof, world, 2 * prof_steps, world * done * w["B"] / el)
            out["parallelism"].update(check)
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(urm, params, w, args.cpu_seconds)
    eng.close()
    # ---- N > 1: the configuration BASELINE.md defines the >= 6x target on -- configs[3], 200 000 x 50 000 users x items sharded
    # 25 000 rows per GPU -- at the paper's default emb_dim = 32 and at 1024, beside the configs[1] line above (weak scaling: every
    # rank holds one shard at every N).  GANMF_BENCH_FORCE_COMM=1 rehearses the same code on one GPU (one-rank RCCL communicator).
    if (world > 1 or force_comm) and os.environ.get("GANMF_BENCH_SKIP_C4") != "1":
        extra = []
        for e4 in (32, 1024):
            try:
                extra.append(extra_workload(C4_SHARD, e4, world, rank, local_rank, min(steps, 20), min(warmup, 6), sync, torch, dist, force_comm))
            except Exception as ex:      # never lose the headline line to the extra ones (every rank fails or succeeds alike)
                extra.append({"workload": "configs[3] shard, emb_dim=%d" % e4, "error": "%s: %s" % (type(ex).__name__, ex)})
        failed_checks += [x["workload"] for x in extra if "parallelism" in x and not replicas_ok(x["parallelism"])]
        if rank == 0:
            out["configs3_sharded"] = extra
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    os.close(json_fd)
    if failed_checks:      # (every rank reaches the same verdict: the line is written, the run fails)
        sys.stderr.write("bench.py: replicated tensors differ between ranks, or RCCL's world size is not the launcher's: %s\n" % failed_checks)
        raise SystemExit(3)


if __name__ == "__main__":
    main()
