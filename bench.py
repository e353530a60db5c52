#!/usr/bin/env python3
"""bench.py -- leapfrog steps/s of the HMC hot path on BASELINE configs[1] (+ the other configs as `secondary`).

  python bench.py --gpus N --steps K --warmup W
  (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one HMC transition (epoch) = L=50 leapfrog steps of the 5->50->50->50->1 Relu BNN over the 100k-row
synthetic regression matrix; X and Y are resident in HBM before the
timed region.  The chain starts from the committed burned-in state tests/golden/c2_burned.npz (tools/make_burned.py)
with the step size recorded there, so the accept ratio is in [0.6, 0.9] whatever --steps / --warmup are.  One
independent chain per GPU (chain_id = rank), no data-path collective; the RCCL all-gather of the sampled state
(theta, eta) -- tbnn_gather_samples, the path's only exchange -- runs every --sampling-step epochs inside the timed
region when N > 1.  value = leapfrog steps of all ranks / max-over-ranks wall time of one timed region of exactly --steps
epochs; the region is measured --repeats times back to back (default 5, at most --max-region-seconds in all) and the median
region is reported, every region's time under "timed_regions_ms".

At N = 1 the same invocation also measures BASELINE configs[3], configs[4] (weight transition + hyper transition with
the reference's dual averaging per epoch) and configs[0], each with its own roofline and cpu_baseline, under
"secondary" (--workload X measures X alone; --no-secondary skips them).

roofline: dominant kernel = the fused forward+backward pass (k_fwd_bwd_fast3; k_chain_wide + k_dw_wide for the wide
configs).  achieved = algorithmic matmul FLOP of one pass (2n(3S - in1*out1)) / its mean duration,
measured with hipEvent pairs on the chain's own stream around every 47th pass inside the timed region.  peak = 157.3
TFLOP/s (FP32 MFMA, dense, MI355X_MICROARCH.md).  traffic = HBM bytes per pass from the committed rocprofv3 PMC pass
(profiles/) -- quoted, like frac_rocprof, only when that pass was measured on the build that is loaded (tbnn_build_id), else null.
cpu_baseline: the oracle's C restatement (oracle/c, kind "port") timed on the host cores on a bounded sample of the
same workload from the same state (rank 0, N = 1 only, outside the timed GPU region): best thread count of a short
scan, a 1-thread figure, and a PyTorch-CPU value+grad cross-check (oracle/torch_ref.py).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_TFLOPS = 157.3          # FP32 MFMA, dense (MI355X_MICROARCH.md)
PEAK_HBM_GBPS = 8000.0       # HBM3E spec; a float4 copy measures 6.29 TB/s (same guide) -- frac_hbm is quoted against the measured figure
MEASURED_HBM_GBPS = 6290.0
CONFIG_KEY = {"c1": "configs[0]", "c2": "configs[1]", "c4": "configs[3]", "c5": "configs[4]",
              "c5g": "configs[4] with GaussianDenseLayer priors", "mn": "docs example 784-20-20-1",
              "w300": "8-300-300-1", "mc10": "784-100-100-10", "wm10": "10-200-200-10", "oh100": "1-100-1", "wf50": "50-100-100-1"}
# CPU sample per workload: (max epochs, leapfrog steps per epoch, wall cap in s) for the all-threads run
CPU_SAMPLE = {"c1": (20, 100, 10.0), "c2": (20, 50, 30.0), "c4": (1, 3, 30.0), "c5": (1, 20, 30.0), "mn": (4, 50, 15.0),
              "w300": (1, 5, 15.0), "mc10": (1, 5, 15.0), "wm10": (1, 5, 15.0), "oh100": (2, 20, 10.0), "wf50": (1, 5, 15.0)}


def algorithmic_flops(dims, n):
    S = sum(dims[i] * dims[i + 1] for i in range(len(dims) - 1))
    return 2.0 * n * (3 * S - dims[0] * dims[1])


def host_cpus():
    """(logical CPUs this process may run on, physical cores among them)"""
    aff = sorted(os.sched_getaffinity(0))
    cores = set()
    try:
        for c in aff:
            base = f"/sys/devices/system/cpu/cpu{c}/topology/"
            cores.add((open(base + "physical_package_id").read().strip(), open(base + "core_id").read().strip()))
    except OSError:
        cores = set(aff)
    return len(aff), len(cores)


def cgroup_throttled_ms():
    """milliseconds this container's cgroup has spent throttled by its CPU quota so far (cgroup v2 cpu.stat), or None.  A math
    library that starts a thread per LOGICAL CPU (256 on the test boxes) under a 16-CPU quota burns the 100-ms period's quota in a
    few ms of spinning; the kernel then freezes every thread of the container -- the launch thread too -- for the rest of the period:
    measured as single 77-80 ms stalls inside timed regions that follow a NumPy matmul (NOTES.md, round 4)."""
    try:
        for ln in open("/sys/fs/cgroup/cpu.stat"):
            if ln.startswith("throttled_usec"):
                return int(ln.split()[1]) / 1000.0
    except (OSError, ValueError):
        pass
    return None


def cgroup_cpu_max():
    """the CPU bandwidth limit of this process's cgroup ("max 100000" = none; "1600000 100000" = 16 CPUs), or None"""
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            return open(path).read().strip()
        except OSError:
            pass
    return None


def cpu_baseline(name, wl, X, Y, theta, eta, eps, P):
    """the C restatement (and the torch cross-check) timed on the host; returns the cpu_baseline object"""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import c_oracle
    import tbnn_oracle as o
    from tensorbnn_amd import _native as nat
    bern = wl["lik"] == nat.LIK_BERNOULLI
    spec = o.make_spec(wl["dims"], prior=wl["prior"], likelihood=o.LIK_BERNOULLI if bern else o.LIK_GAUSSIAN,
                       final_act=o.ACT_SIGMOID if bern else o.ACT_NONE)
    co = c_oracle.COracle(spec, X, Y)
    logical, physical = host_cpus()
    max_thr = co.max_threads
    flop_row = algorithmic_flops(wl["dims"], 1)
    # thread-count scan: SUSTAINED gradient evaluations (>= 1 s per candidate -- the box shows 256 logical CPUs but its
    # cgroup schedules a fraction of them, and a 5-ms burst does not see the throttle) on a row block of ~2e10 FLOP
    n_scan = int(min(wl["n"], max(4096, 2e10 / flop_row)))
    cs = c_oracle.COracle(spec, X[:n_scan], Y[:n_scan])
    scan = {}
    for t in sorted({t for t in (8, 16, 32, 64) if t <= max_thr} or {max_thr}):
        cs.set_threads(t)
        cs.logp_grad(theta, eta)                      # warm the pool
        n_ev, t0 = 0, time.perf_counter()
        while n_ev < 3 or time.perf_counter() - t0 < 1.0:
            cs.logp_grad(theta, eta)
            n_ev += 1
        scan[t] = n_ev / (time.perf_counter() - t0) * n_scan / wl["n"]     # full-size gradient evaluations per second
    best = max(scan, key=scan.get)
    max_ep, cL, cap = CPU_SAMPLE[name]
    rng = np.random.default_rng(0)

    def timed(orc, threads, epochs, L, cap_s):
        orc.set_threads(threads)
        th = theta.copy()
        acc, done = [], 0
        t0 = time.perf_counter()
        while done < epochs and (done == 0 or time.perf_counter() - t0 < cap_s):
            p0 = rng.standard_normal(P).astype(np.float32)
            th, a, lar, _, _ = orc.hmc_step(th, eta, eps, L, p0, float(np.log(rng.random())))
            acc.append(min(1.0, float(np.exp(min(lar, 0.0)))))
            done += 1
        dt = time.perf_counter() - t0
        return done * L / dt, done, float(np.mean(acc)), dt

    v, ep, acc, dt = timed(co, best, max_ep, cL, cap)
    lp_c = co.logp_grad(theta, eta)[0]                 # (still at `best` threads) reference value for the torch cross-check
    # one thread: the same arithmetic on a row block sized to ~2.5e10 FLOP per gradient, scaled to the full row count
    n_one = int(min(wl["n"], max(1024, 2.5e10 / flop_row)))
    c1t = c_oracle.COracle(spec, X[:n_one], Y[:n_one])
    L1 = cL if name == "c1" else 5
    v1, ep1, _, dt1 = timed(c1t, 1, 1, L1, cap)
    v1 *= n_one / wl["n"]
    out = {"value": round(v, 3), "unit": "leapfrog steps/s", "cores": int(best), "kind": "port",
           "sample": f"{ep} epochs x L={cL} (+1 bootstrap gradient per epoch, as the reference pays, Q10) of the same "
                     f"{wl['n']}-row workload from the same chain state at the timed eps, OpenMP C restatement oracle/c, "
                     f"{dt:.1f} s",
           "accept_ratio": round(acc, 4), "logical_cpus": logical, "physical_cores": physical, "cgroup_cpu_max": cgroup_cpu_max(),
           "thread_scan_grad_evals_per_s": {str(k): round(val, 3) for k, val in scan.items()},
           "one_thread": {"value": round(v1, 4), "unit": "leapfrog steps/s", "cores": 1,
                          "sample": f"{ep1} epoch x L={L1} on the first {n_one} rows, scaled by rows to {wl['n']}; {dt1:.1f} s"}}
    try:                                               # PyTorch-CPU value+grad (secondary cross-check)
        import torch_ref
        tt = torch_ref.TorchTarget(spec, X, Y, threads=best)     # (OpenMP's thread count is process-global: set it back)
        tt.value_and_grad(theta, eta)
        n_ev, t0 = 0, time.perf_counter()
        while n_ev < 20 and (n_ev == 0 or time.perf_counter() - t0 < 10.0):
            lp_t, _ = tt.value_and_grad(theta, eta)
            n_ev += 1
        dt_t = time.perf_counter() - t0
        out["torch_cpu"] = {"value": round(n_ev / dt_t, 3), "unit": "value+grad evaluations/s (= leapfrog steps/s)",
                            "cores": tt.threads, "kind": "port", "sample": f"{n_ev} evaluations, torch {tt.threads} threads, fp32 autograd",
                            "logp_rel_diff_vs_c_port": abs(lp_t - lp_c) / abs(lp_c)}
    except Exception as e:                             # the cross-check is optional; the C port is the baseline
        out["torch_cpu"] = {"error": repr(e)[:200]}
    return out


def committed_profile(name, build_id, profiles_dir=None):
    """What the committed rocprofv3 summaries under profiles/ say about workload `name` -- but only when they were measured on THIS
    build: rocprof_kernel_us.json / pmc_traffic.json carry the tbnn_build_id() of the library that was profiled
    (tools/rocprof_summary.py), and a kernel changed without re-profiling must not be quoted with the old kernel's numbers.
    Returns {"kernel_us", "source", "traffic", "match", "profiled_build"}; kernel_us / traffic are None on a mismatch."""
    d = profiles_dir or os.path.join(ROOT, "profiles")
    key = "c5" if name == "c5g" else name               # (c5g: configs[4]'s kernels on other priors)
    out = {"kernel_us": None, "source": None, "traffic": None, "match": False, "profiled_build": None}
    try:
        ku = json.load(open(os.path.join(d, "rocprof_kernel_us.json")))
    except Exception:
        return out
    out["profiled_build"] = (ku.get("_build") or {}).get("build_id")
    out["match"] = bool(build_id) and out["profiled_build"] == build_id
    if not out["match"]:
        return out
    e = ku.get(key)
    if e:
        out["kernel_us"], out["source"] = float(e["us"]), e["source"]
    try:
        tj = json.load(open(os.path.join(d, "pmc_traffic.json")))
        if (tj.get("_build") or {}).get("build_id") == build_id:
            out["traffic"] = tj.get(key, {}).get("hbm_bytes_per_launch")
    except Exception:
        pass
    return out


def roofline_bound(flops, alg_bytes):
    """which roof the algorithmic work of one pass sits under: MFMA floor flops / 157.3 TF/s against the HBM floor at the measured
    6.29 TB/s (machine balance 25 FLOP/B)"""
    return "hbm" if alg_bytes / (MEASURED_HBM_GBPS * 1e9) > flops / (PEAK_TFLOPS * 1e12) else "mfma"


def run_workload(name, steps, warmup, args, rank, world, dev, ctx, repeats=1):
    """one workload on this rank's GPU; returns the result dict (rank 0) or None"""
    import numpy as np
    import torch
    import torch.distributed as dist
    from tensorbnn_amd import _native as nat
    from tensorbnn_amd import parallel
    from tensorbnn_amd.network import DualAveraging
    from tensorbnn_amd.workloads import WORKLOADS, bench_eps, burned_state, synth_problem

    wl = WORKLOADS[name]
    DIMS, N_ROWS, L = wl["dims"], wl["n"], wl["L"]
    layers, lik, X, Y, theta0, eta0 = synth_problem(DIMS, N_ROWS, prior=wl["prior"], likelihood=wl["lik"], x_scale=wl.get("x_scale"))
    burned = None if args.from_initial else burned_state(name, os.path.join(ROOT, "tests", "golden"))
    hyper = wl["hyper"]                   # configs[4]: hyper-HMC on the priors enabled (network.py:414-471)
    da = None
    if burned is not None:
        theta0, eta0 = burned["theta"].astype(np.float32), burned["eta"].astype(np.float32)
        eps_warm = eps = float(burned["eps"])
        state = f"burned-in ({int(burned['epochs'])} epochs, tests/golden/{name}_burned.npz)"
    else:
        eps_warm, eps = bench_eps(name) if name in ("c2", "c4", "c5") else ((1e-4, 1e-4) if name == "c1" else (1e-3, 1e-3) if name == "mn" else (1e-6, 1e-6))
        state = "initial state (no burned-in fixture): warm-up at eps_warmup"
    if args.eps is not None:
        eps_warm = eps = args.eps
    if hyper:
        da = DualAveraging(1e-2, burnin=10 ** 9)          # setupMCMC's default hyperStepSize; adapting throughout
        if burned is not None and "da_h" in burned:
            da.h, da.logEpsilonBar, da.step_size = np.float32(burned["da_h"]), np.float32(burned["da_logEpsilonBar"]), np.float32(burned["da_step"])
    da_epoch = int(burned["da_epoch"]) if (hyper and burned is not None and "da_epoch" in burned) else 0
    kern = {"auto": nat.KERNEL_AUTO, "generic": nat.KERNEL_GENERIC, "fast": nat.KERNEL_FAST}[args.kernel]
    ch = nat.Chain(layers, likelihood=lik, device=dev, seed=50, chain_id=rank, kernel=kern)
    # inputs resident in HBM before the timed region (torch owns the buffers)
    dX = torch.from_numpy(X).cuda()
    dY = torch.from_numpy(Y).cuda()
    torch.cuda.synchronize()
    ch.set_data_device(dX.data_ptr(), dY.data_ptr(), N_ROWS)
    ch.set_state(theta0)
    ch.set_hypers(eta0)

    # checkpoint-time gather: the native RCCL all-gather of the C ABI (tbnn_gather_samples); every rank must agree on
    # the route, so a rank that cannot load / join says so through torch.distributed before anyone enters a collective
    comm, gather_kind = None, "none"
    sample = gathered = None
    dist_on = ctx["dist"]                 # world > 1, or TBNN_BENCH_FORCE_DIST=1: the N > 1 code path at world = 1 (real RCCL on one GPU)
    if dist_on:
        cdev = ctx["cdev"]
        ok = 1
        try:
            nat.comm_unique_id()                       # loads the collective library (non-collective)
        except Exception as e:
            ok = 0
            print(f"[rank {rank}] native collective library unavailable: {e}", file=sys.stderr, flush=True)
        t = torch.tensor([ok], dtype=torch.int32, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        if int(t.item()) == 1:
            try:
                comm = parallel.make_comm(ch)
                ch.gather_samples(comm)
            except Exception as e:
                ok, comm = 0, None
                print(f"[rank {rank}] native gather failed: {e}", file=sys.stderr, flush=True)
            t = torch.tensor([ok], dtype=torch.int32, device=cdev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
        if int(t.item()) == 1:
            gather_kind = "tbnn_gather_samples (RCCL all-gather on the chain's stream)"
        else:                                          # torch.distributed carries the same P+H floats per rank
            comm = None
            gather_kind = "torch.distributed all_gather (native route unavailable, see stderr)"
            sample = torch.empty(ch.P + ch.H, dtype=torch.float32, device="cuda")
            gathered = torch.empty(world * (ch.P + ch.H), dtype=torch.float32, device=cdev)

    hyp_acc, hyp_eps = [], []

    def run(epochs, eps):
        nonlocal da_epoch
        outs = []
        done = 0
        while done < epochs:
            k = min(args.sampling_step, epochs - done)
            if hyper:                          # weight transition + hyper transition + dual averaging per epoch (hyper steps not counted)
                for _ in range(k):
                    outs.append(ch.hmc_step(eps, L))
                    h = ch.hyper_step(float(da.step_size), 100)
                    hyp_eps.append(float(da.step_size))
                    hyp_acc.append(float(da.update(da_epoch, h["log_accept_ratio"])))
                    da_epoch += 1
            else:
                outs += ch.hmc_run(eps, L, k)
            done += k
            if dist_on:                        # checkpoint-time gather over RCCL/xGMI
                if comm is not None:
                    ch.gather_samples(comm)
                else:
                    ch.export_sample_device(sample.data_ptr())
                    if ctx["single_gpu"]:
                        dist.all_gather(list(gathered.chunk(world)), sample.cpu())
                    else:
                        dist.all_gather_into_tensor(gathered, sample)
        return outs

    def fence():
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    run(warmup, eps_warm)
    hyp_acc.clear(); hyp_eps.clear()
    # hipEvent pairs around a sample of the fused passes: the event pool is created HERE, outside the timed region
    # hipEvent pair around every 47th fused pass inside the timed region (coprime to the trajectory lengths: the samples walk through
    # all positions of a trajectory).  Every pair costs the stream a bubble: at every 10th pass (rounds 1-3) the pairs themselves took
    # 1.2 % off the measured rate of configs[1] (19.75 k against 19.98 k leapfrog steps/s)
    ch.set_profiling(int(os.environ.get("TBNN_BENCH_PROFILE_STRIDE", "47")))
    # The timed region: EXACTLY `steps` epochs between two fences (barrier + device synchronisation), max over ranks.  A region of a
    # few tens of ms (the driver's --steps 20 is 0.05 s at configs[1]) is at the mercy of one scheduling hiccup, so the region is
    # measured `repeats` times back to back -- each again exactly `steps` epochs of the free-running chain -- and the MEDIAN region is
    # the one reported (every region's time is in the full record); regions stop early once `--max-region-seconds` are spent.
    regions, outs, dt_own = [], [], None
    throttled = 0.0
    t_all = time.perf_counter()
    for rep in range(max(1, repeats)):
        fence()
        thr0 = cgroup_throttled_ms()
        t0 = time.perf_counter()
        o_r = run(steps, eps)
        own = time.perf_counter() - t0         # this rank's own clock: its last epoch's record has been read back (hmc_run / gather return)
        if os.environ.get("TBNN_BENCH_FAIL_RANK") == str(rank):      # test hook: a rank that dies mid-run must take the job down, non-zero
            print(f"[rank {rank}] TBNN_BENCH_FAIL_RANK: exiting with code 3", file=sys.stderr, flush=True)
            os._exit(3)
        fence()
        dt_r = time.perf_counter() - t0
        thr1 = cgroup_throttled_ms()
        if dist_on:
            t = torch.tensor([dt_r], dtype=torch.float64, device=ctx["cdev"])
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt_r = float(t.item())
        regions.append((dt_r, own, o_r, (thr1 - thr0) if (thr0 is not None and thr1 is not None) else None))
        stop = time.perf_counter() - t_all > args.max_region_seconds
        if dist_on:                                    # every rank takes the same decision
            t = torch.tensor([1.0 if stop else 0.0], dtype=torch.float64, device=ctx["cdev"])
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            stop = bool(t.item() > 0)
        if stop:
            break
    order = sorted(range(len(regions)), key=lambda i: regions[i][0])
    med = order[(len(order) - 1) // 2]                  # the median region (the lower one of an even count: a region that was measured)
    dt, dt_own, _, thr = regions[med]
    outs = [o for r in regions for o in r[2]]           # accept ratio / kernel-time samples: every region's epochs
    throttled = round(thr, 1) if thr is not None else None
    region_ms = [round(r[0] * 1e3, 3) for r in regions]
    ranks = None
    if dist_on:
        cdev = ctx["cdev"]
        # what every rank saw: its own rate (a starved launch thread shows here), the communicator size its collective library
        # reports (ncclCommCount), the gather route it took
        mine = torch.tensor([steps * L / dt_own, float(comm.count()) if comm is not None else -1.0,
                             1.0 if comm is not None else 0.0, float(len(os.sched_getaffinity(0)))], dtype=torch.float64, device=cdev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per = [a.tolist() for a in allr]
        ranks = {"steps_per_s": [round(a[0], 1) for a in per], "steps_per_s_min": round(min(a[0] for a in per), 1),
                 "steps_per_s_max": round(max(a[0] for a in per), 1), "nccl_comm_count": [int(a[1]) for a in per],
                 "native_gather": [bool(a[2]) for a in per], "cpus_allowed": [int(a[3]) for a in per],
                 "omp_num_threads": os.environ.get("OMP_NUM_THREADS"), "collective_library": os.environ.get("TBNN_RCCL_LIB", "librccl (RCCL)")}
        acc = torch.tensor([float(np.mean([o["accept_prob"] for o in outs])),
                            float(np.mean([o["accepted"] for o in outs]))], dtype=torch.float64, device=cdev)
        dist.all_reduce(acc, op=dist.ReduceOp.SUM)
        acc_prob, acc_frac = (acc / world).tolist()
    else:
        acc_prob = float(np.mean([o["accept_prob"] for o in outs]))
        acc_frac = float(np.mean([o["accepted"] for o in outs]))

    total_leapfrog = world * steps * L
    value = total_leapfrog / dt
    prof = [o["fwdbwd_us"] for o in outs if o["fwdbwd_us"] > 0]
    k_pair_us = float(np.mean(prof)) if prof else None
    # next to it: a burst of passes at the state the timed region ended in, back to back on the chain's stream between ONE pair of events
    # (tbnn_debug_fused_burst, ~20 ms).  It is a THROUGHPUT figure: consecutive launches of the same kernel overlap head and tail (41.9 us per
    # pass at configs[1] where rocprofv3 sees 42.6 us per launch), while an event pair around a single launch inside the timed region carries
    # ~3 us of its own cost (45.4).  `roofline.frac` stays on the event pairs (the conservative one); rocprofv3's duration is frac_rocprof.
    k_us = k_pair_us
    k_burst_us = None
    if rank == 0 and k_pair_us:
        try:
            k_burst_us = ch.fused_burst_us(int(min(2000, max(5, 20000.0 / k_pair_us))))
        except Exception as e:               # (a diagnostic build without the entry point)
            print(f"bench: fused burst not measured ({e})", file=sys.stderr)
    flops = algorithmic_flops(DIMS, N_ROWS)
    kernel_name = ch.kernel_name
    try:                                     # which kernels ran the leapfrog steps of the timed transitions (small narrow problems: one launch for all L)
        if ch.last_transition_path == "trajectory":
            kernel_name = "traj:" + kernel_name
    except Exception:
        pass
    theta_end, eta_end = (ch.get_state(), ch.get_hypers()) if rank == 0 else (None, None)
    P = ch.P
    if comm is not None:
        comm.close()
    ch.close()
    del dX, dY
    if rank != 0:
        return None

    prof_c = committed_profile(name, nat.build_id())
    traffic = prof_c["traffic"]
    roofline = None
    if k_us and name != "c1":
        alg_bytes = 4.0 * N_ROWS * (DIMS[0] + DIMS[-1]) + 12.0 * P        # X, Y read once; q, p, g read / written
        bound = roofline_bound(flops, alg_bytes)
        tf = flops / (k_us * 1e-6) / 1e12
        gbps = alg_bytes / (k_us * 1e-6) / 1e9
        rp_us, rp_src = prof_c["kernel_us"], prof_c["source"]
        if bound == "mfma":
            head = {"bound": "mfma", "kernel": kernel_name, "achieved": round(tf, 3), "peak": PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(tf / PEAK_TFLOPS, 4),
                    "frac_rocprof": round(flops / (rp_us * 1e-6) / 1e12 / PEAK_TFLOPS, 4) if rp_us else None}
        else:                                  # at or under the ridge (docs784: 20.7 FLOP/B against 25): the HBM floor is the longer one
            head = {"bound": "hbm", "kernel": kernel_name, "achieved": round(gbps, 1), "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                    "frac": round(gbps / PEAK_HBM_GBPS, 4),
                    "frac_rocprof": round(alg_bytes / (rp_us * 1e-6) / 1e9 / PEAK_HBM_GBPS, 4) if rp_us else None}
        roofline = dict(head)
        roofline.update({"frac_mfma": round(tf / PEAK_TFLOPS, 4), "frac_hbm": round(gbps / MEASURED_HBM_GBPS, 4),
                    "frac_hbm_note": "algorithmic bytes / kernel time / 6.29 TB/s (measured copy rate); frac_mfma: algorithmic FLOP / kernel time / 157.3 TF/s",
                    "rocprof_kernel_us": rp_us, "rocprof_source": rp_src,
                    "profile_build_match": prof_c["match"], "build_id": nat.build_id(), "profiled_build": prof_c["profiled_build"],
                    "hazard_check": nat.lint_status(),
                    "traffic": traffic, "kernel_us": round(k_us, 2), "kernel_us_burst": round(k_burst_us, 2) if k_burst_us else None,
                    "kernel_us_note": "kernel_us: mean of the event pairs around single launches inside the timed region (+~3 us of pair cost); "
                                      "kernel_us_burst: passes back to back between one event pair after it (throughput: launches overlap head and tail)",
                    "flop_per_launch": flops, "algorithmic_bytes_per_launch": alg_bytes,
                    "hbm_gbps_algorithmic": round(gbps, 2),
                    "end_to_end_frac": round(value / world * flops / 1e12 / PEAK_TFLOPS, 4) if bound == "mfma"
                    else round(value / world * alg_bytes / 1e9 / PEAK_HBM_GBPS, 4)})
    elif name == "c1":
        roofline = {"bound": "launch latency (7e5 FLOP per step: SURVEY 8(d) reports steps/s only)", "kernel": kernel_name,
                    "kernel_us": round(k_us, 2) if k_us else None, "frac": None}
    cpu = None
    if world == 1 and not args.no_cpu_baseline and name != "c5g":      # c5g: configs[4]'s arithmetic, see its cpu_baseline
        cpu = cpu_baseline(name, wl, X, Y, theta0, eta0, eps, P)
    line = {
        "metric": "leapfrog steps/sec (whole node)", "value": round(value, 2), "unit": "leapfrog steps/s",
        "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": round(dt / steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": wl["text"], "leapfrog_per_step": L, "eps": eps, "eps_warmup": eps_warm,
                   "rows": N_ROWS, "chains": world, "parallelism": f"{world} independent chains",
                   "kernel": kernel_name, "start_state": state, "sample_gather": gather_kind},
        "accept_ratio": round(acc_prob, 4), "accepted_fraction": round(acc_frac, 4),
        "roofline": roofline, "cpu_baseline": cpu, "cgroup_throttled_ms_in_timed_region": throttled,
        "timed_regions_ms": region_ms, "timed_region": f"median of {len(region_ms)} regions of exactly {steps} epochs each",
    }
    if ranks is not None:
        line["ranks"] = ranks
    if hyper:
        line["hyper_accept_ratio"] = round(float(np.mean(hyp_acc)), 4) if hyp_acc else None
        line["hyper_step_size"] = {"first": hyp_eps[0], "last": hyp_eps[-1], "rule": "dual averaging, network.py:457-469"} if hyp_eps else None
    return line


def run_chain_group(name, n_chains, steps, warmup, dev):
    """`n_chains` independent chains of workload `name` on ONE GPU behind one handle (tbnn_create_multi: the per-chain kernels of all
    chains are one launch each, gridDim.y = chain; every chain is bit for bit its solo self, tests/test_gpu_multichain.py).  A
    SECONDARY figure: the headline configs are one chain per GPU by definition.  Small problems -- the reference's own examples --
    are bound by launch latency with most of the GPU idle; this is what a user who wants several chains gets on one card."""
    import numpy as np
    import torch
    from tensorbnn_amd import _native as nat
    from tensorbnn_amd.workloads import WORKLOADS, burned_state, synth_problem
    wl = WORKLOADS[name]
    layers, lik, X, Y, theta0, eta0 = synth_problem(wl["dims"], wl["n"], prior=wl["prior"], likelihood=wl["lik"], x_scale=wl.get("x_scale"))
    b = burned_state(name, os.path.join(ROOT, "tests", "golden"))
    theta0, eta0, eps = b["theta"].astype(np.float32), b["eta"].astype(np.float32), float(b["eps"])
    g = nat.ChainGroup(layers, n_chains, likelihood=lik, device=dev, seed=50, chain_id=0)
    dX, dY = torch.from_numpy(X).cuda(), torch.from_numpy(Y).cuda()
    torch.cuda.synchronize()
    g.set_data_device(dX.data_ptr(), dY.data_ptr(), wl["n"]); g.set_state(theta0); g.set_hypers(eta0)
    g.hmc_run(eps, wl["L"], warmup)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    outs = g.hmc_run(eps, wl["L"], steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    kname = g.kernel_name
    g.close()
    del dX, dY
    return {"value": round(n_chains * steps * wl["L"] / dt, 1), "unit": "leapfrog steps/s (all chains of the group)", "chains_on_one_gpu": n_chains,
            "per_chain": round(steps * wl["L"] / dt, 1), "steps": steps, "warmup": warmup, "ms_per_step": round(dt / steps * 1e3, 4),
            "accept_ratio": round(float(np.mean([o["accept_prob"] for c in outs for o in c])), 4),
            "config": {"workload": wl["text"], "leapfrog_per_step": wl["L"], "eps": eps, "kernel": kname,
                       "how": "tbnn_create_multi: one handle, gridDim.y = chain; chain c == the solo chain with chain_id c"}}


def _short_cpu(c):
    if not c:
        return None
    out = {"value": c["value"], "unit": c["unit"], "cores": c["cores"], "kind": c["kind"],
           "sample": c["sample"].split(" (+1")[0] + ", oracle/c OpenMP"}
    if "one_thread" in c:
        out["one_thread"] = c["one_thread"]["value"]
    return out                                 # (the PyTorch-CPU cross-check is in the full record)


def _short_roof(r):
    if not r:
        return None
    keep = ("bound", "achieved", "peak", "unit", "frac", "frac_rocprof", "frac_mfma", "frac_hbm", "rocprof_source", "profile_build_match", "traffic",
            "kernel_us", "end_to_end_frac")          # ("kernel" is config.kernel: not repeated on the short line)
    out = {k: r[k] for k in keep if k in r}
    if r.get("hazard_check"):                      # the build-time MFMA hazard check, in short: units checked / pairs repaired in their listings
        import re
        hc = r["hazard_check"]
        fixed = sum(int(x) for x in re.findall(r"[(](\d+) repaired", hc))
        out["hazard_check"] = f"{hc.count('listing checked')} units checked, {fixed} pairs repaired"
    if out.get("rocprof_source"):
        out["rocprof_source"] = out["rocprof_source"].replace("profiles/", "").split(" + per-dispatch trace")[0] + " (medians)"
    if "launch latency" in str(out.get("bound", "")):
        out["bound"] = "launch latency"
    return out


def compact_line(line):
    """the stdout line: the contract's keys in full for the headline config, a few numbers per secondary config"""
    out = dict(line)
    out["roofline"] = _short_roof(line.get("roofline"))
    out["cpu_baseline"] = _short_cpu(line.get("cpu_baseline"))
    cfg = dict(line["config"])
    cfg["workload"] = cfg["workload"].split(":")[0] + ":" + cfg["workload"].split(":", 1)[1].split("(")[0].rstrip() \
        if ":" in cfg["workload"] else cfg["workload"]
    cfg.pop("eps_warmup", None)
    cfg.pop("parallelism", None)
    out["config"] = cfg
    out.pop("accepted_fraction", None)
    if not out.get("cgroup_throttled_ms_in_timed_region"):
        out.pop("cgroup_throttled_ms_in_timed_region", None)
    if "ranks" in out:                                   # per-rank diagnostics: min / max and what the collective library counted
        r = out["ranks"]
        out["ranks"] = {"steps_per_s_min": r["steps_per_s_min"], "steps_per_s_max": r["steps_per_s_max"],
                        "nccl_comm_count": sorted(set(r["nccl_comm_count"])), "native_gather": all(r["native_gather"])}
    if "secondary" in line:
        sec = {}
        for key, r in line["secondary"].items():
            if "error" in r:
                sec[key] = r
                continue
            if "chains_on_one_gpu" in r:
                sec["configs[0]x64"] = {"value": r["value"], "chains": r["chains_on_one_gpu"], "per_chain": r["per_chain"], "accept": r["accept_ratio"]}
                continue
            e = {"value": r["value"], "ms_per_step": round(r["ms_per_step"], 3), "accept": round(r["accept_ratio"], 3)}      # (L = value x ms_per_step / 1000)
            rf = r.get("roofline") or {}
            if rf.get("bound") == "hbm":
                e["bound"] = "hbm"
            for k in ("frac", "frac_rocprof", "traffic"):
                if rf.get(k) is not None and not (k != "frac" and key.endswith("GaussianDenseLayer priors")):      # ([4]g: [4]'s kernel, [4]'s profile)
                    e[k] = rf[k]
            c = r.get("cpu_baseline")
            if c:
                e["cpu"] = c["value"]
            if r.get("cgroup_throttled_ms_in_timed_region"):
                e["throttled_ms"] = r["cgroup_throttled_ms_in_timed_region"]
            if "hyper_accept_ratio" in r:
                e["hyper_accept"] = r["hyper_accept_ratio"]
                e["hyper_step"] = float("%.3g" % r["hyper_step_size"]["last"]) if r.get("hyper_step_size") else None
            sec[key.replace(" with GaussianDenseLayer priors", "g").replace("docs example 784-20-20-1", "docs784")] = e
        out["secondary"] = sec
        out["secondary_note"] = "[4]: Cauchy priors (Q1); [4]g: Gaussian priors; [0]x64: 64 chains, 1 GPU; docs784: tutorial shape; all: gpurun_out/bench_full.json"
    return out


def visible_gpus():
    """GPUs a rank could use, found WITHOUT touching the HIP runtime in this process (the parent of the ranks must stay
    GPU-free): a short-lived CHILD asks the native library (tbnn_device_count = hipGetDeviceCount).  sysfs is not enough: a
    container may show the host's whole KFD topology (8 nodes) while one GPU is usable, and eight ranks started on that
    evidence all open the one card.  Falls back to the KFD node count (cut down by *_VISIBLE_DEVICES) when the child cannot
    run; None when nothing says."""
    import subprocess
    try:
        r = subprocess.run([sys.executable, "-c",
                            "import ctypes, sys; l = ctypes.CDLL(sys.argv[1]); l.tbnn_device_count.restype = ctypes.c_int; print(l.tbnn_device_count())",
                            os.path.join(ROOT, "tensorbnn_amd", "libtbnn.so")], capture_output=True, text=True, timeout=120)
        n = int(r.stdout.strip().splitlines()[-1])
        if r.returncode == 0 and n >= 0:
            return n
    except Exception:
        pass
    return _kfd_gpu_nodes()


def _kfd_gpu_nodes():
    try:
        base = "/sys/class/kfd/kfd/topology/nodes"
        cnt = 0
        for node in os.listdir(base):
            for ln in open(os.path.join(base, node, "properties")):
                if ln.startswith("simd_count") and int(ln.split()[1]) > 0:
                    cnt += 1
    except (OSError, ValueError):
        return None
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            cnt = min(cnt, len([x for x in v.split(",") if x.strip() != ""]))
    return cnt


def spawn_ranks(n):
    """`python bench.py --gpus N` run bare: launch N ranks with torch.distributed.run as a child process, pass their output
    through (rank 0 prints the one JSON line), return the child's exit code"""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # one launch thread per rank does the work; the math libraries of N ranks must not each start a thread per logical CPU of a
    # box whose cgroup schedules 2 CPUs per rank
    thr = str(max(1, min(4, (len(os.sched_getaffinity(0)) or 1) // n)))
    for var in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
        env.setdefault(var, thr)
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--workload", default=None, choices=["c1", "c2", "c4", "c5", "c5g", "mn", "w300", "mc10", "wm10", "oh100", "wf50"],
                    help="measure this workload alone (default: c2 = BASELINE configs[1], + the others as `secondary` at N = 1)")
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--eps", type=float, default=None, help="leapfrog step size (default: fixture value)")
    ap.add_argument("--sampling-step", type=int, default=10)
    ap.add_argument("--repeats", type=int, default=5, help="timed regions of exactly --steps epochs each for the headline workload; the median region is reported")
    ap.add_argument("--max-region-seconds", type=float, default=20.0, help="no further timed region is started once this much time went into them")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--from-initial", action="store_true", help="start from the initial state instead of the burned-in fixture")
    ap.add_argument("--kernel", default="auto", choices=["auto", "generic", "fast"])
    args = ap.parse_args()

    # --gpus N > 1 without a launcher around us: start the N ranks ourselves (one process per GPU) as a CHILD
    # torch.distributed.run and relay rank 0's line.  Nothing in this process has touched the GPU (or imported torch) yet.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        have = None if os.environ.get("TBNN_BENCH_SINGLE_GPU", "0") == "1" else visible_gpus()
        if have is not None and have < args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but only {have} GPU(s) visible on this node (one rank per GPU; nothing was started)")
        raise SystemExit(spawn_ranks(args.gpus))
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:      # ranks started by a launcher: same thread caps as spawn_ranks sets
        for var in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
            os.environ.setdefault(var, "2")
    else:
        # one process: NumPy's BLAS (the synthetic problems' teacher networks) must not start a thread per LOGICAL CPU under a CPU quota
        # (see cgroup_throttled_ms); the CPU baseline sets its own thread counts through the C library / torch
        os.environ.setdefault("OPENBLAS_NUM_THREADS", "8")

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import torch
    import torch.distributed as dist

    # the library is built before the ranks need it: local rank 0 builds (atomic rename), the others wait for the file
    lib_path = os.path.join(ROOT, "tensorbnn_amd", "libtbnn.so")
    if not os.path.exists(lib_path):
        if local_rank == 0:
            from tensorbnn_amd import build as _b
            _b.build(verbose=False)
        else:
            t0 = time.time()
            while not os.path.exists(lib_path):
                if time.time() - t0 > 900:
                    raise SystemExit("libtbnn.so did not appear (build it first: python -m tensorbnn_amd.build)")
                time.sleep(0.5)
    from tensorbnn_amd.workloads import WORKLOADS

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    # TBNN_BENCH_SINGLE_GPU=1 (test hook): every rank uses GPU 0 and torch's collectives run over gloo on host
    # copies -- exercises the N > 1 control flow on a 1-GPU box.  Normal runs: one rank per GPU over RCCL.
    single_gpu = os.environ.get("TBNN_BENCH_SINGLE_GPU", "0") == "1"
    if not single_gpu and torch.cuda.device_count() < world:
        raise SystemExit(f"--gpus {world} but only {torch.cuda.device_count()} GPU(s) visible")
    dev = 0 if single_gpu else local_rank
    torch.cuda.set_device(dev)
    # TBNN_BENCH_FORCE_DIST=1 (test hook): run the N > 1 code path -- process group, native communicator, gather inside the timed
    # region, per-rank diagnostics -- at world = 1, i.e. against the REAL RCCL on a one-GPU box
    force_dist = os.environ.get("TBNN_BENCH_FORCE_DIST", "0") == "1"
    ctx = {"single_gpu": single_gpu, "cdev": "cpu" if single_gpu else "cuda", "dist": world > 1 or force_dist}
    if ctx["dist"]:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if single_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev))

    main_wl = args.workload or "c2"
    wl = WORKLOADS[main_wl]
    steps = args.steps if args.steps is not None else wl["steps"]
    warmup = args.warmup if args.warmup is not None else wl["warmup"]
    line = run_workload(main_wl, steps, warmup, args, rank, world, dev, ctx, repeats=args.repeats)
    if args.workload is None and world == 1 and not args.no_secondary:
        sec = {}
        for name in ("c4", "c5", "c5g", "c1", "mn"):
            w2 = WORKLOADS[name]
            try:
                r = run_workload(name, w2["steps"], w2["warmup"], args, rank, world, dev, ctx)
                for k in ("metric", "higher_is_better", "scaling", "vs_baseline", "data", "n_gpus"):
                    r.pop(k, None)
                sec[CONFIG_KEY[name]] = r
            except Exception as e:           # a secondary line must never take the headline line down with it
                sec[CONFIG_KEY[name]] = {"error": repr(e)[:300]}
        try:                                 # several chains on one GPU (secondary: small problems are launch-bound)
            sec["configs[0] x64 chains on one GPU"] = run_chain_group("c1", 64, 100, 10, dev)
        except Exception as e:
            sec["configs[0] x64 chains on one GPU"] = {"error": repr(e)[:300]}
        line["secondary"] = sec
    if rank == 0:
        # the whole record (long `sample` / `workload` texts, thread scans, torch cross-check) goes to stderr and to
        # gpurun_out/bench_full.json; stdout carries ONE line short enough that every config's value / roofline fraction /
        # CPU baseline survives a 2-KB tail of it
        full = json.dumps(line)
        print(full, file=sys.stderr, flush=True)
        try:
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            with open(os.path.join(ROOT, "gpurun_out", "bench_full.json"), "w") as fh:
                fh.write(full + "\n")
        except OSError:
            pass
        print(json.dumps(compact_line(line)), flush=True)
    if ctx["dist"]:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
