"""bench.py -- acquisition-candidate evaluations / second on MI355X (BASELINE.json's metric).

A "step" is one pass of the hot path over one batch of synthetic candidates resident in HBM:

  headline / c2 / c3   fused EI sweep: K* generation -> W K* (f64 MFMA) -> variance / mean -> Expected Improvement
                       -> arg-max                                                      unit: candidates/s
  c4                   batch Monte-Carlo EI: joint posterior of G q-batches (f64 MFMA sweep + Gram) -> chol(cov + jitter I)
                       -> reparametrised samples -> qEI -> arg-max over the batches    unit: q-batches/s
  c5                   decoupled Thompson sampling: B trajectories (RFF features + canonical kernel sums) over the
                       candidates -> per-trajectory arg-min                             unit: candidate-trajectory evals/s

followed, for N > 1 GPUs, by ONE RCCL all-gather of the per-rank (value, index) winners, a merge kernel and one
16-byte device-to-host copy.  `update` (K assembly + Cholesky + inverse) is outside the timed region and reported in
`config`.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload headline|c2|c3|c4|c5]
                    [--scaling weak|strong] [--mode ranks|group]

`--gpus N` with N > 1 and no torch.distributed environment re-launches itself as N ranks (one process per GPU,
`python -m torch.distributed.run --nproc-per-node N ...`, backend nccl = RCCL); under torchrun it is one of the ranks.
`--mode group` instead drives all N GPUs from ONE process through the C-ABI's single-controller group
(tgp_group_*: in-process RCCL communicator).  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

I8_PEAK_TOPS = 5033.0  # int8 MFMA: 2048 ops / clk / SIMD x 1024 SIMDs x 2.4 GHz (= 2 x the bf16 dense rate)
FP64_PEAK_TFLOPS = 78.6  # MI355X fp64 matrix = fp64 vector peak (public spec; MFMA measured 78.0, profiles/r01_ubench_fp64.txt)

WORKLOADS = {
    # BASELINE.md section 4 / SURVEY.md section 8(d).  "M" = units per GPU (weak scaling)
    "headline": dict(kind="ei", objective="ackley", d=8, kernel="matern52", N=4096, M=1 << 20, noise=1e-2),
    # SURVEY 8(d) names both noise levels: sigma^2 = 1e-5 is the reference's integration-test value (a secondary line)
    "headline_lownoise": dict(kind="ei", objective="ackley", d=8, kernel="matern52", N=4096, M=1 << 20, noise=1e-5),
    "c3": dict(kind="ei", objective="ackley", d=8, kernel="matern52", N=4096, M=1_000_000, noise=1e-2),
    "c2": dict(kind="ei", objective="hartmann_6", d=6, kernel="rbf", N=1024, M=1_000_000, noise=1e-2),
    "c4": dict(kind="qei", objective="hartmann_6", d=6, kernel="matern52", N=2048, M=100_000, noise=1e-2, q=50, S=512),
    "c5": dict(kind="ts", objective="ackley", d=16, kernel="matern52", N=8192, M=2_000_000, noise=1e-2, F=2048, B=4),
}
UNITS = {"ei": "candidates/s", "qei": "q-batches/s", "ts": "candidate-trajectory evals/s"}
KERNEL_FLOPS = {"rbf": 12, "matern12": 16, "matern32": 18, "matern52": 20}  # c_k of SURVEY 8(d)
COS_FLOPS = 20  # c_cos: one RFF feature (range reduction + polynomial)


def traffic_of(key: str):
    """HBM bytes per step of a workload's dominant kernel from profiles/traffic.json (separate rocprofv3 PMC passes,
    2 * FETCH_SIZE + WRITE_SIZE), with the round THAT ENTRY was measured in -> (bytes or None, source string or None)."""
    tfile = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        tj = json.load(open(tfile))
    except Exception:
        return None, None
    val = tj.get(key)
    if not isinstance(val, (int, float)):
        return None, None
    rnd = (tj.get("_rounds") or {}).get(key, "?")
    return float(val), (f"profiles/traffic.json[{key}]: rocprofv3 PMC passes (2*FETCH_SIZE + WRITE_SIZE) of round {rnd} "
                        f"(profiles/{rnd}_rocprof_{key}.txt) -- NOT measured in this run")


def flops_per_unit(w) -> float:
    """Algorithmic flops per unit, SURVEY.md section 8(d)."""
    N, d, ck = float(w["N"]), w["d"], KERNEL_FLOPS[w["kernel"]]
    if w["kind"] == "ei":        # per candidate: N^2 + N (3d + c_k) + 2N + 60
        return N * N + N * (3 * d + ck) + 2.0 * N + 60.0
    if w["kind"] == "qei":       # per q-batch: q N^2 + q N (3d + c_k) + q^2 N + q^3/3 + q^2 S + 3 q S
        q, S = w["q"], w["S"]
        return q * N * N + q * N * (3 * d + ck) + q * q * N + q ** 3 / 3.0 + q * q * S + 3.0 * q * S
    # per candidate-trajectory: 2 F d + c_cos F + 2 F + N (3d + c_k) + 2 N for ONE trajectory.  The B
    # trajectories of a Thompson batch share the sampler's RFF basis and the kernel row k(x, X) (reference
    # acquisition/sampler.py:262-271 draws them from one sampler), so the work the algorithm NEEDS per candidate is
    # F (2d + c_cos) + N (3d + c_k) once plus 2 (F + N) per trajectory; the roofline is priced on that, per (cand, traj)
    F, B = w["F"], w["B"]
    return (F * (2.0 * d + COS_FLOPS) + N * (3 * d + ck)) / B + 2.0 * (F + N)


# ---------------------------------------------------------------------------------------------------------------
# CPU baselines (rank 0, N = 1 only): the oracle side of the repo is imported HERE and nowhere else in this file
# ---------------------------------------------------------------------------------------------------------------
def cpu_baseline(w, budget_s=4.5):
    """kind = "port": trieste's own GPflow/TF path cannot be installed here.  The EI sweeps
    use torch-CPU (MKL trsm/gemm) float64 in chunks of 16384 candidates on all host cores, in the reference's
    algorithmic shape; `improved` is the same sweep with the engine's algorithmic savings (cached alpha, one solve)
    so that the GPU/CPU ratio is not inflated by them.  c4 / c5 time the numpy oracle's batch MC-EI / trajectory."""
    from oracle import cpu_baseline as CB
    from oracle import gp_oracle as O

    X, Y = O.synthetic_problem(getattr(O, w["objective"]), w["d"], w["N"])
    st = O.gpr_update(w["kernel"], 1.0, O.default_lengthscales(w["d"]), w["noise"], float(Y.mean()), X, Y)
    d, N = w["d"], w["N"]
    if w["kind"] == "ei":
        eta = O.eta_min_mean(st)
        chunk = 16384
        rate, done, el, threads = CB.timed_sweep(st, eta, d, chunk, budget_s, improved=False)
        rate2, done2, el2, _ = CB.timed_sweep(st, eta, d, chunk, budget_s / 2.5, improved=True, threads=threads)
        return {"value": rate, "unit": UNITS["ei"], "cores": threads, "kind": "port",
                "sample": f"{done} candidates at N={N}, d={d}, {w['kernel']}: torch-CPU float64 (MKL, {threads} threads = the fastest of the power-of-two "
                          f"fractions of {os.cpu_count()} logical cores on a probe chunk), "
                          f"reference-shaped chunks of {chunk} (K* [N,{chunk}], 2 trsm, EI, arg-max), {el:.1f} s",
                "improved": {"value": rate2, "unit": UNITS["ei"], "cores": threads,
                             "sample": f"{done2} candidates, same sweep with cached alpha (gemv mean) and one trsm, {el2:.1f} s"}}
    try:
        from threadpoolctl import threadpool_info

        threads = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        threads = os.cpu_count() or 1
    rng = np.random.default_rng(5678)
    if w["kind"] == "qei":
        q, S = w["q"], w["S"]
        eps = np.random.default_rng(91011).standard_normal((q, S))
        eta = O.eta_min_mean(st)
        g, done, t0 = 8, 0, time.perf_counter()
        while True:
            O.batch_mc_ei(st, rng.uniform(size=(g, q, d)), eps, eta)
            done += g
            el = time.perf_counter() - t0
            if el > budget_s:
                break
        return {"value": done / el, "unit": UNITS["qei"], "cores": int(threads), "kind": "port",
                "sample": f"{done} q-batches (q={q}, S={S}) at N={N}, d={d}: numpy/scipy float64 restatement "
                          f"(predict_joint, chol(cov + jitter I), reparametrised samples), {el:.1f} s"}
    F, B = w["F"], w["B"]
    r2 = np.random.default_rng(7)
    Wf, b = r2.standard_t(5, size=(F, d)), r2.uniform(0, 2 * np.pi, F)
    wts, xi = r2.standard_normal((F, B)), r2.standard_normal((N, B))
    v = O.decoupled_weights(st, Wf, b, wts, xi)
    chunk, done, t0 = 4096, 0, time.perf_counter()
    while True:
        O.trajectory_eval(st, Wf, b, wts, v, rng.uniform(size=(chunk, d)))
        done += chunk * B
        el = time.perf_counter() - t0
        if el > budget_s:
            break
    return {"value": done / el, "unit": UNITS["ts"], "cores": int(threads), "kind": "port",
            "sample": f"{done} candidate-trajectory evaluations (F={F}, B={B}) at N={N}, d={d}: numpy float64 restatement "
                      f"(features [chunk, F+N] materialised per chunk of {chunk}), {el:.1f} s"}


def end_to_end_fit_ms(X, Y, w):
    """Informational (outside the timed region): ``GaussianProcessRegression.optimize`` -- the MAP fit of a BO step (prior
    draws as batched trial evaluations, then L-BFGS-B) -- through the reference-shaped host API, COLD: a fresh model from
    ``build_gpr`` defaults, its first ``optimize`` (engine buffers, task plans and scratch are allocated inside the timed
    call), then a second fresh model in the same process (allocator warm).  -> {"ms", "nfev", "second_ms", "second_nfev"}
    or a "failed: ..." string."""
    try:
        import scipy.optimize  # noqa: F401  (a ~190 ms import the first optimize() of a process would otherwise pay)

        import trieste_amd.models as M
        from trieste_amd.data import Dataset
        from trieste_amd.space import Box

        d = w["d"]
        data = Dataset(X, Y[:, None])
        out = {}
        for key in ("", "second_"):
            model = M.GaussianProcessRegression(M.build_gpr(data, Box([0.0] * d, [1.0] * d), likelihood_variance=w["noise"]))
            t0 = time.perf_counter()
            res = model.optimize(data)
            out[key + "ms"] = (time.perf_counter() - t0) * 1e3
            out[key + "nfev"] = int(getattr(res, "nfev", -1))
        out["what"] = "cold fit: fresh model from build_gpr defaults, first optimize() (90 prior draws in batched launches + L-BFGS-B)"
        return out
    except Exception as e:  # never let the informational figure break the bench line
        return f"failed: {type(e).__name__}: {e}"


def end_to_end_acquire_ms(X, Y, w):
    """Informational (outside the timed region, SURVEY 8d "end-to-end acquire time"): one
    EfficientGlobalOptimization().acquire on the same model through the reference-shaped host API."""
    try:
        import trieste_amd.models as M
        from trieste_amd import objectives as O
        from trieste_amd.acquisition import EfficientGlobalOptimization
        from trieste_amd.data import Dataset
        from trieste_amd.space import Box

        d = w["d"]
        kern = M.Kernel(variance=1.0, lengthscales=O.default_lengthscales(d), kind=w["kernel"])
        model = M.GaussianProcessRegression(M.GPR(data=(X, Y[:, None]), kernel=kern,
                                                  mean_function=M.Constant(float(Y.mean())),
                                                  likelihood_variance=w["noise"]))
        data = Dataset(X, Y[:, None])
        rule = EfficientGlobalOptimization()
        space = Box([0.0] * d, [1.0] * d)
        rule.acquire_single(space, model, dataset=data)  # warm-up (allocations)
        t0 = time.perf_counter()
        rule.acquire_single(space, model, dataset=data)
        return (time.perf_counter() - t0) * 1e3
    except Exception as e:  # never let the informational figure break the bench line
        return f"failed: {type(e).__name__}: {e}"


def small_call_ms(X, Y, w):
    """Informational (outside the timed region): the calls a BO step makes on a HANDFUL of points at this model -- since the last
    session of round 6 skinny products instead of launches of the sweep / joint kernels: predict at 128 points,
    predict_joint of 60 groups of 5, and qEI value + gradient of 300 groups of 5 (one L-BFGS-B iteration of a joint batch) --
    median of 10 host-timed calls each, ms."""
    try:
        from trieste_amd import objectives as O
        from trieste_amd.engine import GPEngine

        d = w["d"]
        eng = GPEngine(d, w["kernel"])
        eng.set_hyper(1.0, O.default_lengthscales(d), w["noise"], float(Y.mean()))
        eng.set_data(X, Y)
        rng = np.random.default_rng(3)
        xp, xg, xq = rng.uniform(size=(128, d)), rng.uniform(size=(60, 5, d)), rng.uniform(size=(300, 5, d))
        eps = rng.standard_normal((5, 512))
        eta = float(np.median(np.asarray(eng.predict(xp)[0])))

        def med(f):
            f()
            ts = []
            for _ in range(10):
                t0 = time.perf_counter()
                f()
                ts.append((time.perf_counter() - t0) * 1e3)
            return sorted(ts)[5]

        out = {"predict_128": med(lambda: eng.predict(xp)), "predict_joint_60x5": med(lambda: eng.predict_joint(xg)),
               "qei_value_grad_300x5_S512": med(lambda: eng.qei_value_grad(xq, eps, eta, 1e-6))}
        eng.close()
        return out
    except Exception as e:  # never let the informational figure break the bench line
        return f"failed: {type(e).__name__}: {e}"


def qei_eta(eng, Xq) -> float:
    """The incumbent of the qEI workload: the MEDIAN posterior mean over the first 8192 candidate points.  At the
    reference's eta = min_i mean(X_i) (function.py:1135-1147) every uniformly random q-batch of this synthetic problem
    has qEI = max(eta - min, 0) = 0 exactly, so the step's arg-max would be over zeros; the
    arithmetic and its cost do not depend on eta, the winner does."""
    import torch

    return float(torch.median(eng.predict_mean(Xq.reshape(-1, Xq.shape[-1])[:8192])))


def secondary_line(name: str, precision: str, steps: int, device: int = 0, traffic_key: str = "") -> dict:
    """One more workload inside the SAME driver-timed run (1 GPU): `steps` timed steps after one warm-up step, the
    dominant kernel timed by HIP events on its launch stream, priced exactly like the main line.  Returns
    {value, unit, ms_per_step, steps, dtype, roofline: {achieved, peak, unit, frac, kernel, kernel_ms}}."""
    import torch

    from trieste_amd import objectives as O
    from trieste_amd.engine import GPEngine

    w = dict(WORKLOADS[name])
    kind, d, kernel, N, noise, per = w["kind"], w["d"], w["kernel"], w["N"], w["noise"], w["M"]
    X, Y = O.synthetic_problem(getattr(O, w["objective"]), d, N)
    eng = GPEngine(d, kernel, device=device)
    eng.set_precision(precision)
    eng.use_torch_stream()
    eng.set_hyper(1.0, O.default_lengthscales(d), noise, float(Y.mean()))
    eng.set_data(X, Y)
    eta = eng.eta()
    dev = f"cuda:{device}"
    if kind == "ei":
        Xq = eng.sample_box(5678, 0, per, 0.0, 1.0)

        def step():
            return eng.acq_argmax("ei", eta, Xq)[:2]
    elif kind == "qei":
        q, S = w["q"], w["S"]
        eps = torch.from_numpy(np.random.default_rng(91011).standard_normal((q, S))).to(dev)
        Xq = eng.sample_box(5678, 0, per * q, 0.0, 1.0).reshape(per, q, d)
        eta = qei_eta(eng, Xq)

        def step():
            v, i = torch.max(eng.qei(Xq, eps, eta, 1e-6), 0)
            return float(v), int(i)
    else:
        F, B = w["F"], w["B"]
        r2 = np.random.default_rng(7)
        traj = eng.trajectory(r2.standard_t(5, size=(F, d)), r2.uniform(0, 2 * np.pi, F),
                              r2.standard_normal((F, B)), r2.standard_normal((N, B)))
        Xq = eng.sample_box(5678, 0, per, 0.0, 1.0)

        def step():
            v, i = traj.argmin(Xq)
            return float(v[0]), int(i[0])

    best = step()
    kms = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        best = step()
        kms.append(eng.last_kernel_ms()[0])
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    units = per * (w.get("B", 1) if kind == "ts" else 1)
    k_ms = float(np.mean(kms))
    auto_info = None
    if precision == "auto" and kind == "ei":  # the int8 sweep with the a-posteriori float64 repair: price what it ran
        _, eff, frac = eng.get_precision()
        auto_info = {"arithmetic": eff, "recomputed_in_f64_fraction": frac}
        try:  # the run-time check of the error model over the warm-up + timed sweeps: both strata of the canary, apart
            strata = eng.get_auto_strata()
            rep = eng.get_auto_report()
            auto_info["canary"] = {"uniform_1_in_4096": strata["uniform"], "adversarial_worst_bound_per_64th": strata["adversarial"],
                                   "slack_saved": strata["slack_saved"], "demotions": rep["demotions"], "rung": rep["level"]}
        except Exception as e:  # (informational: never let it break the bench line)
            auto_info["canary"] = f"failed: {type(e).__name__}: {e}"
        precision = eff
    emulated = precision != "f64" and kind == "ei"
    if emulated:
        planes = int(precision[-1])
        ops = (10.0 if planes == 4 else 15.0) * float(N) * N
        achieved, peak, unit = ops * units / (k_ms * 1e-3) * 1e-12, I8_PEAK_TOPS, "TOP/s (int8)"
        kern = "sweep_i8_kernel<KIND, DP>"
    else:
        achieved, peak, unit = flops_per_unit(w) * units / (k_ms * 1e-3) * 1e-12, FP64_PEAK_TFLOPS, "TFLOP/s"
        kern = {"ei": "sweep_dma_kernel<KIND, DP>", "qei": "joint_kernel<KIND, DP>", "ts": "traj_eval kernel"}[kind]
    out = {"value": units * steps / el, "unit": UNITS[kind], "ms_per_step": el / steps * 1e3, "steps": steps,
           "dtype": "f64" if not emulated else
                    f"f64 emulated ({precision}: int8 digit planes, Ozaki" +
                    ("; TGP_PREC_AUTO: every candidate outside the parity tolerance by its own error bound, and every candidate "
                     "that could be the float64 arg-max, recomputed in float64 inside the timed step)" if auto_info else ")"),
           "workload": f"{name}: {kind}, {w['objective']} d={d}, {kernel}, N={N}, {per} units, noise={noise:g}",
           "roofline": {"achieved": achieved, "peak": peak, "unit": unit, "frac": achieved / peak, "kernel": kern,
                        "kernel_ms": k_ms}}
    tr, tr_src = traffic_of(traffic_key) if traffic_key else (None, None)
    out["roofline"].update({"traffic": tr, "traffic_source": tr_src,
                            "hbm_GBps": (tr / (k_ms * 1e-3) * 1e-9) if tr else None,
                            "hbm_frac": (tr / (k_ms * 1e-3) / 8.0e12) if tr else None})
    if auto_info:
        out["auto"] = auto_info
        out["best"] = [float(best[0]), int(best[1])]
    eng.close()
    return out


def respawn_as_ranks(n: int) -> None:
    """`python bench.py --gpus N` (N > 1, no torch.distributed environment): become N ranks, one per GPU."""
    import torch

    have = torch.cuda.device_count()
    if have < n:
        raise SystemExit(f"bench.py --gpus {n}: only {have} GPU(s) visible -- refusing to report an {n}-GPU number")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.run(cmd, env=env).returncode)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="headline", choices=sorted(WORKLOADS))
    ap.add_argument("--scaling", default="weak", choices=("weak", "strong"),
                    help="weak: the workload's units per GPU; strong: 8 x that in total, split over the GPUs")
    ap.add_argument("--mode", default="ranks", choices=("ranks", "group"),
                    help="ranks: one process per GPU over torch.distributed/RCCL; group: one process, tgp_group_*")
    ap.add_argument("--merge", default="rccl", choices=("rccl", "peer"), help="group mode: winner exchange")
    ap.add_argument("--m-per-gpu", type=int, default=0, help="override the units per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the secondary workloads (c2 / c4 / c5 / emulated i8x5 headline) attached to the default line")
    ap.add_argument("--secondary-steps", type=int, default=3)
    ap.add_argument("--no-acquire", action="store_true", help="skip the informational end-to-end acquire timing")
    ap.add_argument("--variant", type=int, default=0, help="tgp_set_variant launch-policy bits (experiments)")
    ap.add_argument("--precision", default="f64", choices=("f64", "i8x4", "i8x5", "auto"),
                    help="arithmetic of the EI sweeps: f64 (default, the parity path) or i8x4 = W K* on the int8 matrix "
                         "cores with four digit planes per operand (emulated precision; its own line, never the headline)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")

    under_launcher = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if args.mode == "ranks" and args.gpus > 1 and not under_launcher:
        respawn_as_ranks(args.gpus)

    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: trieste_amd has no CPU fallback")
    group_mode = args.mode == "group"
    if group_mode:
        if under_launcher and int(os.environ["WORLD_SIZE"]) > 1:
            raise SystemExit("--mode group is a single process: do not launch it under torchrun with several ranks")
        if torch.cuda.device_count() < args.gpus:
            raise SystemExit(f"--mode group --gpus {args.gpus}: only {torch.cuda.device_count()} GPU(s) visible")
        world, rank, local_rank, use_dist = 1, 0, 0, False
        nshards = args.gpus
    else:
        world = int(os.environ.get("WORLD_SIZE", "1")) if under_launcher else 1
        rank = int(os.environ.get("RANK", "0")) if under_launcher else 0
        local_rank = int(os.environ.get("LOCAL_RANK", "0")) if under_launcher else 0
        if world != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: refusing to report a {args.gpus}-GPU number "
                             f"from {world} rank(s)")
        use_dist = under_launcher  # also with --nproc-per-node 1: the RCCL path at world size 1
        nshards = world
    torch.cuda.set_device(local_rank)
    rccl_ranks = 0
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
        rccl_ranks = dist.get_world_size()

    from trieste_amd import objectives as O  # the seeded synthetic problem (inputs)
    from trieste_amd.distributed import all_gather_winners
    from trieste_amd.engine import GPEngine

    w = dict(WORKLOADS[args.workload])
    kind, d, kernel, N, noise = w["kind"], w["d"], w["kernel"], w["N"], w["noise"]
    per_gpu = args.m_per_gpu or w["M"]
    if args.scaling == "strong":  # fixed total = the config's 8-GPU job, split over the GPUs in use
        total_units = 8 * per_gpu
        per = -(-total_units // nshards)
    else:
        total_units, per = per_gpu * nshards, per_gpu
    X, Y = O.synthetic_problem(getattr(O, w["objective"]), d, N)  # seed 1234, standardised
    ls = O.default_lengthscales(d)
    dev = f"cuda:{local_rank}"

    if group_mode:
        from trieste_amd.group import GPEngineGroup

        if kind == "qei":
            raise SystemExit("--mode group benches the resident-candidate sweeps (headline, c2, c3, c5)")
        grp = GPEngineGroup(d, kernel, devices=list(range(args.gpus)), merge=args.merge)
        info = grp.info()
        rccl_ranks = info["rccl_ranks"]
        for m in grp.members:
            m.set_variant(args.variant)
            m.set_precision(args.precision)
        grp.set_hyper(1.0, ls, noise, float(Y.mean()))
        grp.set_data(X, Y)  # warm-up (allocations)
        t0 = time.perf_counter()
        grp.set_data(X, Y)  # every member runs the same deterministic update, concurrently
        update_ms = (time.perf_counter() - t0) * 1e3
        eta = grp.eta()
        grp.sample_candidates(5678, total_units, 0.0, 1.0)  # ONE logical Philox sample, sharded
        eng = grp.primary
        if kind == "ts":
            r2 = np.random.default_rng(7)
            traj = grp.trajectory(r2.standard_t(5, size=(w["F"], d)), r2.uniform(0, 2 * np.pi, w["F"]),
                                  r2.standard_normal((w["F"], w["B"])), r2.standard_normal((N, w["B"])))

            def step():
                v, i = traj.argmin()
                return float(v[0]), int(i[0])
        else:
            def step():
                v, i, _ = grp.acq_argmax("ei", eta)
                return float(v), int(i)

        def kernel_ms_of_step():
            return grp.last_kernel_ms()
    else:
        eng = GPEngine(d, kernel, device=local_rank)
        eng.set_variant(args.variant)
        eng.set_precision(args.precision)
        eng.use_torch_stream()
        eng.set_hyper(1.0, ls, noise, float(Y.mean()))
        eng.set_data(X, Y)  # warm-up (allocations)
        torch.cuda.synchronize()
        samples = []
        for _ in range(5):  # every rank runs the same deterministic update (replicated model state); the median of five calls
            t0 = time.perf_counter()
            eng.set_data(X, Y)
            torch.cuda.synchronize()
            samples.append((time.perf_counter() - t0) * 1e3)
        update_ms = sorted(samples)[2]
        eta = eng.eta()
        lo = min(rank * per, total_units)
        mine = max(0, min(lo + per, total_units) - lo)
        if mine == 0:
            raise SystemExit("empty shard: fewer units than ranks")
        if kind == "ei":
            # rank r owns global rows [lo, lo + mine) of ONE logical Philox sample (seed 5678), generated on the device
            Xq = eng.sample_box(5678, lo, mine, 0.0, 1.0)

            def step():
                pair = eng.acq_argmax_pair("ei", eta, Xq, index_base=lo)      # enqueue only
                v, i = all_gather_winners(eng, pair)                          # all-gather + merge + ONE sync
                return float(v[0]), int(i[0])
        elif kind == "qei":
            q, S = w["q"], w["S"]
            eps = torch.from_numpy(np.random.default_rng(91011).standard_normal((q, S))).to(dev)
            Xq = eng.sample_box(5678, lo * q, mine * q, 0.0, 1.0).reshape(mine, q, d)
            eta = qei_eta(eng, eng.sample_box(5678, 0, 8192, 0.0, 1.0))  # the same eta on every rank

            def step():
                vals = eng.qei(Xq, eps, eta, 1e-6)                             # [G] on the device
                v, i = torch.max(vals, 0)
                pair = torch.stack([v, (i + lo).view(1).view(torch.float64)[0]])
                gv, gi = all_gather_winners(eng, pair)
                return float(gv[0]), int(gi[0])
        else:
            F, B = w["F"], w["B"]
            r2 = np.random.default_rng(7)
            traj = eng.trajectory(r2.standard_t(5, size=(F, d)), r2.uniform(0, 2 * np.pi, F),
                                  r2.standard_normal((F, B)), r2.standard_normal((N, B)))
            Xq = eng.sample_box(5678, lo, mine, 0.0, 1.0)

            def step():
                pairs = traj.argmin_pairs(Xq, index_base=lo)
                v, i = all_gather_winners(eng, pairs, minimize=True)
                return float(v[0]), int(i[0])

        def kernel_ms_of_step():
            return eng.last_kernel_ms()[0]  # HIP events on the launch stream, this step's dominant kernel

    best = step()  # first call: allocations and (lazy in RCCL) communicator set-up, never in the timed region
    for _ in range(args.warmup):
        best = step()
    kernel_ms = []
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        best = step()
        kernel_ms.append(kernel_ms_of_step())
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    per_rank_ms = [float(np.mean(kernel_ms))]
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # every rank's mean kernel time: a future SCALE line shows the load balance, not only the slowest rank
        mine_ms = torch.tensor([float(np.mean(kernel_ms))], dtype=torch.float64, device=dev)
        all_ms = torch.empty(world, dtype=torch.float64, device=dev)
        dist.all_gather_into_tensor(all_ms, mine_ms)
        per_rank_ms = [float(x) for x in all_ms.cpu()]
    elif group_mode:
        per_rank_ms = [float(m.last_kernel_ms()[0]) for m in grp.members]  # the last step's, per member

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        units_per_unit = w.get("B", 1) if kind == "ts" else 1
        value = total_units * units_per_unit * args.steps / elapsed
        k_ms = float(np.mean(kernel_ms))
        fl = flops_per_unit(w)
        my_units = (per if not group_mode else -(-total_units // nshards)) * units_per_unit
        achieved = fl * my_units / (k_ms * 1e-3) * 1e-12 if k_ms > 0 else float("nan")
        traffic, traffic_src = None, None
        if not args.m_per_gpu and args.scaling == "weak":
            traffic, traffic_src = traffic_of({"i8x4": "i8", "i8x5": "i8x5", "auto": "auto"}.get(args.precision, args.workload)
                                              if kind == "ei" else args.workload)
        kern_name = {"ei": "sweep_dma_kernel<KIND, DP>" if d <= 16 else "sweep_kernel<KIND, DP, JOINT=false, SPLIT=false>", "qei": "joint_kernel<KIND, DP>",
                     "ts": "traj_eval_kernel"}[kind]
        eff_precision = args.precision
        if args.precision == "auto" and kind == "ei":
            eff_precision = (grp.primary if group_mode else eng).get_precision()[1]
        emulated = eff_precision in ("i8x4", "i8x5") and kind == "ei"
        if emulated:  # priced on the int8 work the scheme NEEDS: 10 digit-plane products of N^2 ops per candidate
            kern_name = "sweep_i8_kernel<KIND, DP>"
            i8_ops = (10.0 if eff_precision == "i8x4" else 15.0) * float(N) * N
            i8_achieved = i8_ops * my_units / (k_ms * 1e-3) * 1e-12 if k_ms > 0 else float("nan")
        par = (f"single-controller group x{nshards} (tgp_group_*, merge={args.merge})" if group_mode else
               f"candidate-sharded x{world}, one process per GPU, replicated model, (val,idx) all-gather")
        out = {
            "metric": "acquisition-candidate evals/sec (N train, d dim)",
            "value": value,
            "unit": UNITS[kind],
            "n_gpus": nshards,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f64" if not (args.precision != "f64" and kind == "ei") else
                     (f"f64 emulated ({args.precision}): W K* as {eff_precision[-1]} x {eff_precision[-1]} int8 digit planes (Ozaki, "
                      f"{10 if eff_precision == 'i8x4' else 15} int8 MFMA products, exact int32 sums); K*, mean, norms, EI, "
                      "arg-max in f64" + ("; a-posteriori float64 repair" if args.precision == "auto" else "")) if emulated else "f64",
            "data": "synthetic",
            "config": {
                "workload": f"{args.workload}: {kind} step, {w['objective']} d={d}, {kernel}, N={N} train, "
                            f"{per} units/GPU ({total_units} total), noise={noise:g}"
                            + (f", q={w['q']}, S={w['S']}" if kind == "qei" else "")
                            + (f", F={w['F']}, B={w['B']}" if kind == "ts" else ""),
                "N": N, "d": d, "kernel": kernel, "units_per_gpu": per, "total_units": total_units,
                "parallelism": par, "rccl_ranks": rccl_ranks, "kernel_ms_per_rank": per_rank_ms,
                "update_ms": update_ms, "eta": float(eta), "best_value": best[0], "best_index": best[1],
            },
            "roofline": {
                "bound": "mfma" if kind != "ts" else "valu_f64", "achieved": achieved, "peak": FP64_PEAK_TFLOPS,
                "unit": "TFLOP/s", "frac": achieved / FP64_PEAK_TFLOPS, "traffic": traffic, "traffic_source": traffic_src,
                "kernel": kern_name, "kernel_ms": k_ms, "flops_per_unit": fl,
                # the other roofline (SURVEY 8d asks for both): PMC traffic / kernel time against 8 TB/s HBM3E
                "hbm_GBps": (traffic / (k_ms * 1e-3) * 1e-9) if traffic else None,
                "hbm_frac": (traffic / (k_ms * 1e-3) / 8.0e12) if traffic else None,
            },
        }
        if emulated:
            I8_PEAK = I8_PEAK_TOPS
            out["roofline"].update({"bound": "mfma", "achieved": i8_achieved, "peak": I8_PEAK, "unit": "TOP/s (int8)",
                                    "frac": i8_achieved / I8_PEAK, "int8_ops_per_unit": i8_ops,
                                    "f64_equivalent_TFLOPs": achieved})
        if nshards == 1 and not args.no_acquire and kind == "ei":
            out["config"]["acquire_ms"] = end_to_end_acquire_ms(X, Y, w)
            out["config"]["fit"] = end_to_end_fit_ms(X, Y, w)
            out["config"]["small_calls_ms"] = small_call_ms(X, Y, w)
        if (nshards == 1 and not args.no_secondary and args.workload == "headline" and args.precision == "f64"
                and not args.m_per_gpu and not group_mode):
            # every other workload of BASELINE.json's configs in the same driver-timed run: few steps each, same pricing
            if not group_mode:
                eng.close()
            sec = {}
            for name, prec, tkey in (("c2", "f64", "c2"), ("c4", "f64", "c4"), ("c5", "f64", "c5"), ("headline", "auto", "auto"),
                                     ("headline", "i8x5", "i8x5"), ("headline_lownoise", "f64", "headline")):
                key = name if prec == "f64" else f"{name}_{prec}"
                try:
                    sec[key] = secondary_line(name, prec, args.secondary_steps, local_rank, tkey)
                except Exception as e:  # a secondary line never breaks the graded one
                    sec[key] = {"error": f"{type(e).__name__}: {e}"}
            if isinstance(sec.get("headline_auto"), dict) and "best" in sec["headline_auto"]:
                # the repaired int8 sweep ran over the same Philox candidates as the float64 headline steps above
                sec["headline_auto"]["same_winner_as_f64_headline"] = bool(
                    sec["headline_auto"]["best"][1] == out["config"]["best_index"]
                    and abs(sec["headline_auto"]["best"][0] - out["config"]["best_value"]) <= 1e-12 * abs(out["config"]["best_value"]))
            out["secondary"] = sec
        if not args.no_cpu_baseline and nshards == 1:
            out["cpu_baseline"] = cpu_baseline(w)
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
