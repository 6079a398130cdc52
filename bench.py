"""bench.py -- acquisition-candidate evaluations / second on MI355X (BASELINE.json's metric).

A "step" is one fused sweep of the hot path over one batch of synthetic candidates:
    K* generation -> W K* (f64 MFMA) -> variance/mean -> Expected Improvement -> arg-max
followed, for N > 1 GPUs, by the RCCL all-gather of the per-rank (value, index) winners.
Inputs (model state, candidates) are resident in HBM when the timed region starts; `update`
(K assembly + Cholesky + inverse) is outside it and reported separately in `config`.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload headline|c2|c3]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP64_MFMA_PEAK_TFLOPS = 78.6  # MI355X fp64 matrix peak (public spec; measured 78.0, profiles/r01_ubench_fp64.txt)

WORKLOADS = {
    # name: (objective, d, kernel, N, M per GPU, noise)        -- BASELINE.md section 4
    "headline": ("ackley", 8, "matern52", 4096, 1 << 20, 1e-2),  # north-star: N=4096, d=8, 1 GPU
    "c3": ("ackley", 8, "matern52", 4096, 1_000_000, 1e-2),      # Ackley-8, 10^6 candidates / GPU
    "c2": ("hartmann_6", 6, "rbf", 1024, 1_000_000, 1e-2),       # Hartmann-6 RBF N=1024
}
KERNEL_FLOPS = {"rbf": 12, "matern12": 16, "matern32": 18, "matern52": 20}  # c_k of SURVEY 8(d)


def algorithmic_flops_per_candidate(N: int, d: int, kernel: str) -> float:
    """N^2 + N (3d + c_k) + 2N + 60."""
    return float(N) * N + N * (3 * d + KERNEL_FLOPS[kernel]) + 2.0 * N + 60.0


def cpu_baseline(obj_name, d, kernel, N, noise, budget_s=15.0):
    """The oracle's reference-shaped sweep (materialise K*, two triangular solves, column norms,
    EI, arg-max per chunk) on the host cores.  kind = "port": trieste's own GPflow/TF path cannot
    be installed here."""
    from oracle import gp_oracle as O

    try:
        from threadpoolctl import threadpool_info

        threads = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        threads = os.cpu_count() or 1
    X, Y = O.synthetic_problem(getattr(O, obj_name), d, N)
    st = O.gpr_update(kernel, 1.0, O.default_lengthscales(d), noise, float(Y.mean()), X, Y)
    eta = O.eta_min_mean(st)
    rng = np.random.default_rng(5678)
    chunk = 2000
    done, t0 = 0, time.perf_counter()
    while True:
        Xq = rng.uniform(size=(chunk, d))
        O.ei_sweep_reference_shape(st, Xq, eta, chunk=chunk)
        done += chunk
        el = time.perf_counter() - t0
        if el > budget_s or done >= 60 * chunk:
            break
    return {"value": done / el, "unit": "candidates/s", "cores": int(threads), "kind": "port",
            "sample": f"{done} candidates at N={N}, d={d}, {kernel}: numpy/scipy fp64 restatement of the "
                      f"reference algorithm (K* [N,{chunk}], 2 triangular solves, EI, arg-max), {el:.1f} s"}


def end_to_end_acquire_ms(X, Y, d, kernel, noise):
    """Informational (outside the timed region, SURVEY 8d "end-to-end acquire time"): one
    EfficientGlobalOptimization().acquire on the same model through the reference-shaped host API --
    eta, max(5000, 1000 d) random candidates swept + top-k on the device, 10 d greenlet-batched
    L-BFGS-B runs on the analytic EI gradient (the reference's default for a Box)."""
    try:
        import trieste_amd.models as M
        from trieste_amd import objectives as O
        from trieste_amd.acquisition import EfficientGlobalOptimization
        from trieste_amd.data import Dataset
        from trieste_amd.space import Box

        kern = M.Kernel(variance=1.0, lengthscales=O.default_lengthscales(d), kind=kernel)
        model = M.GaussianProcessRegression(M.GPR(data=(X, Y[:, None]), kernel=kern,
                                                  mean_function=M.Constant(float(Y.mean())),
                                                  likelihood_variance=noise))
        data = Dataset(X, Y[:, None])
        rule = EfficientGlobalOptimization()
        space = Box([0.0] * d, [1.0] * d)
        rule.acquire_single(space, model, dataset=data)  # warm-up (allocations)
        t0 = time.perf_counter()
        rule.acquire_single(space, model, dataset=data)
        return (time.perf_counter() - t0) * 1e3
    except Exception as e:  # never let the informational figure break the bench line
        return f"failed: {type(e).__name__}: {e}"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="headline", choices=sorted(WORKLOADS))
    ap.add_argument("--m-per-gpu", type=int, default=0, help="override candidates per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-acquire", action="store_true", help="skip the informational end-to-end acquire timing")
    ap.add_argument("--variant", type=int, default=0)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: trieste_amd has no CPU fallback")
    torch.cuda.set_device(local_rank)
    # launched by torch.distributed.run (also with --nproc-per-node 1): one process per GPU over RCCL
    use_dist = world > 1 or ("RANK" in os.environ and "MASTER_PORT" in os.environ)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))

    from trieste_amd import objectives as O  # the seeded synthetic problem (inputs)
    from trieste_amd.distributed import all_gather_best
    from trieste_amd.engine import GPEngine

    obj_name, d, kernel, N, M, noise = WORKLOADS[args.workload]
    if args.m_per_gpu:
        M = args.m_per_gpu
    X, Y = O.synthetic_problem(getattr(O, obj_name), d, N)  # seed 1234, standardised
    eng = GPEngine(d, kernel, device=local_rank)
    eng.set_variant(args.variant)
    eng.use_torch_stream()
    eng.set_hyper(1.0, O.default_lengthscales(d), noise, float(Y.mean()))
    eng.set_data(X, Y)  # warm-up (allocations)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.set_data(X, Y)  # every rank runs the same deterministic update (replicated model state)
    torch.cuda.synchronize()
    update_ms = (time.perf_counter() - t0) * 1e3
    eta = eng.eta()
    # weak scaling: M candidates per GPU; rank r owns global rows [r*M, (r+1)*M) of ONE logical
    # Philox sample (seed 5678), generated on the device
    Xq = eng.sample_box(5678, rank * M, M, 0.0, 1.0)

    def step():
        val, idx, _ = eng.acq_argmax("ei", eta, Xq, index_base=rank * M)
        gv, gi = all_gather_best(val, idx, device=f"cuda:{local_rank}", force=use_dist)
        return float(gv[0]), int(gi[0])

    if use_dist:  # communicator set-up (lazy in RCCL) must not land in the timed region even with --warmup 0
        all_gather_best(0.0, rank * M, device=f"cuda:{local_rank}", force=True)
    best = (float("nan"), -1)
    for _ in range(args.warmup):
        best = step()
    kernel_ms = []
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        best = step()
        kernel_ms.append(eng.last_kernel_ms()[0])  # HIP events on the launch stream, this launch
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = world * M * args.steps / elapsed
        k_ms = float(np.mean(kernel_ms))
        flops = algorithmic_flops_per_candidate(N, d, kernel) * M
        achieved = flops / (k_ms * 1e-3) * 1e-12
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tfile) and not args.m_per_gpu:  # measured for the workload's own candidate count
            try:
                traffic = json.load(open(tfile)).get(args.workload)
            except Exception:
                traffic = None
        out = {
            "metric": "acquisition-candidate evals/sec (N train, d dim)",
            "value": value,
            "unit": "candidates/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"{args.workload}: EI sweep + arg-max, {obj_name} d={d}, {kernel}, N={N} train, "
                            f"{M} candidates/GPU, noise={noise:g}",
                "N": N, "d": d, "kernel": kernel, "candidates_per_gpu": M,
                "parallelism": f"candidate-sharded x{world}, replicated model, (val,idx) all-gather",
                "update_ms": update_ms, "best_ei": best[0], "best_index": best[1],
            },
            "roofline": {
                "bound": "mfma", "achieved": achieved, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved / FP64_MFMA_PEAK_TFLOPS, "traffic": traffic,
                "kernel": {0: "sweep_u16_kernel", 1: "sweep_kernel", 2: "sweep_ws_kernel"}.get(args.variant & 0xff, "?"),
                "kernel_ms": k_ms, "flops_per_candidate": algorithmic_flops_per_candidate(N, d, kernel),
                # the other roofline (SURVEY 8d asks for both): PMC traffic / kernel time against 8 TB/s HBM3E
                "hbm_GBps": (traffic / (k_ms * 1e-3) * 1e-9) if traffic else None,
                "hbm_frac": (traffic / (k_ms * 1e-3) / 8.0e12) if traffic else None,
            },
        }
        if world == 1 and not args.no_acquire:
            out["config"]["acquire_ms"] = end_to_end_acquire_ms(X, Y, d, kernel, noise)
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(obj_name, d, kernel, N, noise)
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
