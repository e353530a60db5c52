#!/usr/bin/env python3
"""bench.py -- leapfrog steps/s of the HMC hot path on BASELINE configs[1].

  python bench.py --gpus N --steps K --warmup W
  (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one HMC transition (epoch) = L=50 leapfrog steps of the
5->50->50->50->1 Relu BNN over the 100k-row synthetic regression matrix
(SURVEY.md section 8(d)); X and Y are resident in HBM
before the timed region.  One independent chain per GPU (chain_id = rank), no
data-path collective; an RCCL all-gather of the sampled state (theta, eta)
runs every --sampling-step epochs inside the timed region when N > 1
(checkpoint-time gather, the path's only exchange).  value = leapfrog steps of
all ranks / max-over-ranks wall time.

roofline: dominant kernel = k_fwd_bwd_fast.  achieved = algorithmic matmul
FLOP of one launch (2n(3S - in1*out1)) / its mean duration,
measured with hipEvent pairs on the chain's own stream around every 10th
launch inside the timed region.  peak = 157.3 TFLOP/s (FP32 MFMA, dense,
MI355X_MICROARCH.md).  traffic = HBM bytes per launch from the committed
rocprofv3 PMC pass (profiles/), or null.
cpu_baseline: the oracle's C restatement (oracle/c, kind "port") timed on the
host cores on a bounded sample of the same workload (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_TFLOPS = 157.3
HYPER_EPS = 2e-5       # configs[4] hyper transition (L_h = 100): scan 3e-5 -> 0.85..0.13, 1e-5 -> 0.99 (tools/hyperprobe.py)
# --workload: c2 = BASELINE configs[1] (the metric's config, default); c4 / c5 = configs[3] / configs[4]
# (extra lines for the wide-layer path, same JSON contract)
WORKLOADS = {
    "c2": dict(dims=[5, 50, 50, 50, 1], n=100_000, L=50, lik="gaussian", steps=200, warmup=20, cpu_epochs=4, cpu_L=50,
               text="BASELINE configs[1]: 5->50->50->50->1 Relu BNN (Cauchy DenseLayer, GaussianLikelihood sd=0.1), "
                    "100k-row fp32 synthetic regression, L=50 leapfrog, 1 chain per GPU"),
    "c4": dict(dims=[10, 200, 200, 200, 1], n=1_000_000, L=100, lik="gaussian", steps=20, warmup=20, cpu_epochs=1, cpu_L=3,
               text="BASELINE configs[3]: 10->200->200->200->1 Relu BNN (Cauchy DenseLayer, GaussianLikelihood sd=0.1), "
                    "1M-row fp32 synthetic regression, L=100 leapfrog, 1 chain per GPU"),
    "c5": dict(dims=[20, 100, 100, 2], n=500_000, L=50, lik="bernoulli", steps=60, warmup=20, cpu_epochs=1, cpu_L=20,
               text="BASELINE configs[4]: 20->100->100->2 Relu/Sigmoid BNN (Cauchy DenseLayer, BernoulliLikelihood), "
                    "500k-row fp32 synthetic classification, L=50 leapfrog + hyper-HMC (L_h=100) per epoch, 1 chain per GPU"),
}


def algorithmic_flops(dims, n):
    S = sum(dims[i] * dims[i + 1] for i in range(len(dims) - 1))
    return 2.0 * n * (3 * S - dims[0] * dims[1])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--workload", default="c2", choices=list(WORKLOADS))
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--eps", type=float, default=None, help="leapfrog step size (default: fixture value)")
    ap.add_argument("--sampling-step", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-epochs", type=int, default=None)
    ap.add_argument("--kernel", default="auto", choices=["auto", "generic", "fast"])
    args = ap.parse_args()
    wl = WORKLOADS[args.workload]
    DIMS, N_ROWS, L = wl["dims"], wl["n"], wl["L"]
    if args.steps is None:
        args.steps = wl["steps"]
    if args.warmup is None:
        args.warmup = wl["warmup"]
    if args.cpu_epochs is None:
        args.cpu_epochs = wl["cpu_epochs"]

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import numpy as np
    import torch
    import torch.distributed as dist

    if not os.path.exists(os.path.join(ROOT, "tensorbnn_amd", "libtbnn.so")) and rank == 0:
        from tensorbnn_amd import build as _b
        _b.build(verbose=False)
    from tensorbnn_amd import _native as nat
    from tensorbnn_amd.workloads import synth_problem, bench_eps

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    # TBNN_BENCH_SINGLE_GPU=1 (test hook): every rank uses GPU 0 and the collectives run over gloo on host
    # copies -- exercises the N > 1 control flow on a 1-GPU box.  Normal runs: one rank per GPU over RCCL.
    single_gpu = os.environ.get("TBNN_BENCH_SINGLE_GPU", "0") == "1"
    dev = 0 if single_gpu else local_rank
    torch.cuda.set_device(dev)
    cdev = "cpu" if single_gpu else "cuda"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if single_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev))

    layers, lik, X, Y, theta0, eta0 = synth_problem(
        DIMS, N_ROWS, likelihood=nat.LIK_BERNOULLI if wl["lik"] == "bernoulli" else nat.LIK_GAUSSIAN)
    eps_warm, eps = bench_eps(args.workload)
    hyper = args.workload == "c5"          # configs[4]: hyper-HMC on the priors enabled (network.py:414-471)
    if args.eps is not None:
        eps_warm = eps = args.eps
    kern = {"auto": nat.KERNEL_AUTO, "generic": nat.KERNEL_GENERIC, "fast": nat.KERNEL_FAST}[args.kernel]
    ch = nat.Chain(layers, likelihood=lik, device=dev, seed=50, chain_id=rank, kernel=kern)
    # inputs resident in HBM before the timed region (torch owns the buffers)
    dX = torch.from_numpy(X).cuda()
    dY = torch.from_numpy(Y).cuda()
    torch.cuda.synchronize()
    ch.set_data_device(dX.data_ptr(), dY.data_ptr(), N_ROWS)
    ch.set_state(theta0)
    ch.set_hypers(eta0)
    sample = torch.empty(ch.P + ch.H, dtype=torch.float32, device="cuda")
    gathered = torch.empty(world * (ch.P + ch.H), dtype=torch.float32, device=cdev) if world > 1 else None

    hyp_acc = []

    def run(epochs, profile_stride, eps):
        ch.set_profiling(profile_stride)
        outs = []
        done = 0
        while done < epochs:
            k = min(args.sampling_step, epochs - done)
            if hyper:                          # weight transition + hyper transition per epoch (hyper steps not counted)
                for _ in range(k):
                    outs.append(ch.hmc_step(eps, L))
                    hyp_acc.append(ch.hyper_step(HYPER_EPS, 100)["accept_prob"])
            else:
                outs += ch.hmc_run(eps, L, k)
            done += k
            if world > 1:                      # checkpoint-time gather over RCCL/xGMI
                ch.export_sample_device(sample.data_ptr())
                if single_gpu:
                    dist.all_gather(list(gathered.chunk(world)), sample.cpu())
                else:
                    dist.all_gather_into_tensor(gathered, sample)
        return outs

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    run(args.warmup, 0, eps_warm)
    hyp_acc.clear()
    fence()
    t0 = time.perf_counter()
    outs = run(args.steps, 10, eps)
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        acc = torch.tensor([float(np.mean([o["accept_prob"] for o in outs])),
                            float(np.mean([o["accepted"] for o in outs]))], dtype=torch.float64, device=cdev)
        dist.all_reduce(acc, op=dist.ReduceOp.SUM)
        acc_prob, acc_frac = (acc / world).tolist()
    else:
        acc_prob = float(np.mean([o["accept_prob"] for o in outs]))
        acc_frac = float(np.mean([o["accepted"] for o in outs]))

    total_leapfrog = world * args.steps * L
    value = total_leapfrog / dt
    prof = [o["fwdbwd_us"] for o in outs if o["fwdbwd_us"] > 0]
    k_us = float(np.mean(prof)) if prof else None
    flops = algorithmic_flops(DIMS, N_ROWS)

    if rank == 0:
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                traffic = tj.get("hbm_bytes_per_launch") if args.workload == "c2" else tj.get(args.workload, {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        roofline = None
        if k_us:
            achieved = flops / (k_us * 1e-6) / 1e12
            roofline = {"bound": "mfma", "kernel": ch.kernel_name, "achieved": round(achieved, 3),
                        "peak": PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / PEAK_TFLOPS, 4),
                        "traffic": traffic, "kernel_us": round(k_us, 2), "flop_per_launch": flops,
                        "hbm_gbps_algorithmic": round((4.0 * N_ROWS * (DIMS[0] + DIMS[-1]) + 12.0 * ch.P)
                                                      / (k_us * 1e-6) / 1e9, 2)}
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import c_oracle
            import tbnn_oracle as o
            bern = wl["lik"] == "bernoulli"
            spec = o.make_spec(DIMS, likelihood=o.LIK_BERNOULLI if bern else o.LIK_GAUSSIAN,
                               final_act=o.ACT_SIGMOID if bern else o.ACT_NONE)
            co = c_oracle.COracle(spec, X, Y)
            th = theta0.copy()
            rng = np.random.default_rng(0)
            if args.workload == "c2":
                co.logp_grad(th, eta0)      # warm the thread pool
            tc = time.perf_counter()
            for e in range(args.cpu_epochs):
                p0 = rng.standard_normal(ch.P).astype(np.float32)
                # from the initial state only the warm-up step size is stable; the arithmetic per epoch is identical
                th, _, _, _, _ = co.hmc_step(th, eta0, eps_warm, wl["cpu_L"], p0, float(np.log(rng.random())))
            tc = time.perf_counter() - tc
            cpu = {"value": round(args.cpu_epochs * wl["cpu_L"] / tc, 3), "unit": "leapfrog steps/s", "cores": co.threads,
                   "kind": "port",
                   "sample": f"{args.cpu_epochs} epochs x L={wl['cpu_L']} (+1 bootstrap gradient per epoch, as the reference "
                             f"pays, Q10) of the same {N_ROWS}-row workload, OpenMP C restatement oracle/c"}
        line = {
            "metric": "leapfrog steps/sec (whole node)", "value": round(value, 2), "unit": "leapfrog steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": wl["text"], "leapfrog_per_step": L, "eps": eps, "eps_warmup": eps_warm,
                       "rows": N_ROWS, "chains": world, "parallelism": f"{world} independent chains",
                       "kernel": ch.kernel_name},
            "accept_ratio": round(acc_prob, 4), "accepted_fraction": round(acc_frac, 4),
            "roofline": roofline, "cpu_baseline": cpu,
        }
        if hyper:
            line["hyper_accept_ratio"] = round(float(np.mean(hyp_acc)), 4) if hyp_acc else None
        print(json.dumps(line), flush=True)
    ch.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
